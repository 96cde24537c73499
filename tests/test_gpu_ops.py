"""GPU parity of every C-ABI kernel against the CPU oracle (oracle/pero_oracle.py) on seeded inputs.
Integer / index outputs are compared bit-exact; f32-mode kernels to 1e-5 relative (different f32
summation order than the CPU), bf16-mode kernels against an f32 evaluation of the same bf16-rounded
inputs with a tolerance set by the bf16 output rounding (2^-8 relative)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pero_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    from pero_pretraining_amd import ops as _ops
    return _ops


def dev(x, dtype=None):
    t = torch.as_tensor(x).cuda()
    return t if dtype is None else t.to(dtype)


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ------------------------------------------------------------------------------------------------ gemm
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(70, 45, 33), (128, 64, 16), (1, 1, 1), (200, 130, 300)])
def test_gemm_generic_f32(ops, ta, tb, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((K, N) if tb else (N, K), generator=g)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    gate = torch.randn(M, N, generator=g)
    A = a.t() if ta else a
    Bm = b.t() if tb else b
    ref = torch.relu(0.5 * (A.double() @ Bm.double().t()) + bias.double() + res.double()) * (gate > 0)
    out = ops.gemm(dev(a), dev(b), bias=dev(bias), residual=dev(res), gate=dev(gate), trans_a=ta, trans_b=tb,
                   relu=True, alpha=0.5)
    assert rel_err(out, ref) < 1e-5


def test_gemm_generic_exact_integers_and_accumulate(ops):
    g = torch.Generator().manual_seed(0)
    a = torch.randint(-4, 5, (96, 40), generator=g).float()
    b = torch.randint(-4, 5, (72, 40), generator=g).float()
    ref = a @ b.t()
    c = torch.ones(96, 72).cuda()
    ops.gemm(dev(a), dev(b), out=c, accum=True)
    assert torch.equal(c.cpu(), ref + 1)
    ops.gemm(dev(a), dev(b), out=c, atomic=True)
    assert torch.equal(c.cpu(), 2 * ref + 1)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_bf16_fast_layouts_exact_integers(ops, ta, tb, out_dtype):
    """Small integers: every product and partial sum is exact in bf16 x bf16 -> f32, so the fast MFMA
    kernel (all four operand layouts, swizzled LDS images, transposed reads) must be bit-exact."""
    M, N, K = 256, 384, 192
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-3, 4, (K, M) if ta else (M, K), generator=g).float()
    b = torch.randint(-3, 4, (K, N) if tb else (N, K), generator=g).float()
    A = a.t() if ta else a
    Bm = b.t() if tb else b
    ref = A @ Bm.t()
    out = ops.gemm(dev(a, torch.bfloat16), dev(b, torch.bfloat16), trans_a=ta, trans_b=tb, out_dtype=out_dtype)
    assert out.dtype == out_dtype
    # f32 accumulators are exact; a bf16 output rounds integers above 256 -> compare through that rounding
    assert torch.equal(out.float().cpu(), ref.to(out_dtype).float())


def test_gemm_bf16_fast_epilogues_and_splitk(ops):
    M, N, K = 384, 256, 512
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g).bfloat16()
    b = torch.randn(N, K, generator=g).bfloat16()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).bfloat16()
    gate = torch.randn(M, N, generator=g).bfloat16()
    acc = a.double() @ b.double().t()
    ref = torch.relu(0.25 * acc + bias.double() + res.double()) * (gate.double() > 0)
    out = ops.gemm(dev(a), dev(b), bias=dev(bias), residual=dev(res), gate=dev(gate), relu=True, alpha=0.25)
    assert rel_err(out, ref) < 2 ** -8
    # fast kernel == generic exact-f32 kernel on the same bf16 inputs, up to f32 summation order + output rounding
    out_g = ops.gemm(dev(a), dev(b), bias=dev(bias), residual=dev(res), gate=dev(gate), relu=True, alpha=0.25,
                     force_generic=True)
    assert rel_err(out, out_g.float()) < 2 ** -7
    # split-K with f32 atomics (weight-gradient form: both operands K-major)
    at, bt = a.t().contiguous(), b.t().contiguous()  # stored [K][M], [K][N]
    c = torch.zeros(M, N, device="cuda")
    ops.gemm(dev(at), dev(bt), out=c, trans_a=True, trans_b=True, atomic=True, k_split=4)
    assert rel_err(c, acc) < 1e-5
    ops.gemm(dev(at), dev(bt), out=c, trans_a=True, trans_b=True, atomic=True, k_split=3)
    assert rel_err(c, 2 * acc) < 1e-5


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_bf16_tile256_layouts_exact_integers(ops, ta, tb, out_dtype):
    """PERO_GEMM_TILE256 (the eight-phase 256x256x64 kernel for bf16 outputs; f32 outputs fall through to the other tile
    kernels): all operand layouts, an odd number of K-tiles, exact integers."""
    from pero_pretraining_amd._lib import GEMM_TILE256
    M, N, K = 512, 768, 320
    g = torch.Generator().manual_seed(11)
    a = torch.randint(-3, 4, (K, M) if ta else (M, K), generator=g).float()
    b = torch.randint(-3, 4, (K, N) if tb else (N, K), generator=g).float()
    ref = (a.t() if ta else a) @ (b.t() if tb else b).t()
    out = ops.gemm(dev(a, torch.bfloat16), dev(b, torch.bfloat16), trans_a=ta, trans_b=tb, out_dtype=out_dtype,
                   extra_flags=GEMM_TILE256)
    assert torch.equal(out.float().cpu(), ref.to(out_dtype).float())


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_bf16_t128_layouts_and_epilogues(ops, ta, tb):
    """persistent 128x128x64 kernel, forced (PERO_GEMM_TILE128): exact integers in all layouts + the fused epilogue."""
    M, N, K = 384, 256, 192
    g = torch.Generator().manual_seed(13)
    a = torch.randint(-3, 4, (K, M) if ta else (M, K), generator=g).float()
    b = torch.randint(-3, 4, (K, N) if tb else (N, K), generator=g).float()
    ref = (a.t() if ta else a) @ (b.t() if tb else b).t()
    for od in (torch.bfloat16, torch.float32):
        out = ops.gemm(dev(a, torch.bfloat16), dev(b, torch.bfloat16), trans_a=ta, trans_b=tb, out_dtype=od, extra_flags=64)
        assert torch.equal(out.float().cpu(), ref.to(od).float())
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).bfloat16()
    gate = torch.randn(M, N, generator=g).bfloat16()
    full = torch.relu(0.5 * ref.double() + bias.double() + res.double()) * (gate.double() > 0)
    out = ops.gemm(dev(a, torch.bfloat16), dev(b, torch.bfloat16), trans_a=ta, trans_b=tb, bias=dev(bias), residual=dev(res),
                   gate=dev(gate), relu=True, alpha=0.5, extra_flags=64)
    assert rel_err(out, full) < 2 ** -8


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("policy", [7, 4])
def test_gemm_bf16_r256_o128_layouts_and_epilogues(ops, ta, tb, policy):
    """256x128x32 eight-wave (policy 7) and one-tile-per-workgroup 128x128x64 (policy 4) kernels: exact integers in all
    layouts + the fused epilogue."""
    from pero_pretraining_amd import _lib
    M, N, K = 768, 512, 192
    g = torch.Generator().manual_seed(14)
    a = torch.randint(-3, 4, (K, M) if ta else (M, K), generator=g).float()
    b = torch.randint(-3, 4, (K, N) if tb else (N, K), generator=g).float()
    ref = (a.t() if ta else a) @ (b.t() if tb else b).t()
    _lib.lib().pero_set_option(b"gemm_policy", policy)
    try:
        for od in (torch.bfloat16, torch.float32):
            out = ops.gemm(dev(a, torch.bfloat16), dev(b, torch.bfloat16), trans_a=ta, trans_b=tb, out_dtype=od)
            assert torch.equal(out.float().cpu(), ref.to(od).float())
        bias = torch.randn(N, generator=g)
        res = torch.randn(M, N, generator=g).bfloat16()
        gate = torch.randn(M, N, generator=g).bfloat16()
        full = torch.relu(0.5 * ref.double() + bias.double() + res.double()) * (gate.double() > 0)
        out = ops.gemm(dev(a, torch.bfloat16), dev(b, torch.bfloat16), trans_a=ta, trans_b=tb, bias=dev(bias), residual=dev(res),
                       gate=dev(gate), relu=True, alpha=0.5)
        assert rel_err(out, full) < 2 ** -8
    finally:
        _lib.lib().pero_set_option(b"gemm_policy", 0)


def test_gemm_bf16_tile256_epilogues_splitk_and_persistence(ops):
    from pero_pretraining_amd._lib import GEMM_TILE256
    M, N, K = 256 * 70, 512, 256   # 140 tiles... still < CUs; plus a > CU-count case below
    g = torch.Generator().manual_seed(12)
    a = torch.randn(M, K, generator=g).bfloat16()
    b = torch.randn(N, K, generator=g).bfloat16()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).bfloat16()
    gate = torch.randn(M, N, generator=g).bfloat16()
    acc = a.double() @ b.double().t()
    ref = torch.relu(0.25 * acc + bias.double() + res.double()) * (gate.double() > 0)
    out = ops.gemm(dev(a), dev(b), bias=dev(bias), residual=dev(res), gate=dev(gate), relu=True, alpha=0.25,
                   extra_flags=GEMM_TILE256)
    assert rel_err(out, ref) < 2 ** -8
    # more work items than workgroups (persistent loop + cross-item prefetch): 1200 tiles
    M2 = 256 * 300
    a2 = torch.randn(M2, 64, generator=g).bfloat16()
    b2 = torch.randn(1024, 64, generator=g).bfloat16()
    out2 = ops.gemm(dev(a2), dev(b2), extra_flags=GEMM_TILE256)
    assert rel_err(out2, a2.double() @ b2.double().t()) < 2 ** -8
    out3 = ops.gemm(dev(a2), dev(b2))  # default policy picks a kernel itself
    assert torch.equal(out2, out3) or rel_err(out3, out2.float()) < 2 ** -8
    # split-K atomics on the 256 tile (weight-gradient form) incl. automatic split
    at, bt = a.t().contiguous(), b.t().contiguous()  # (K=256, M) , (K, N): treat as [Kred][Mout]
    x = torch.randn(4096, 512, generator=g).bfloat16()
    dy = torch.randn(4096, 768, generator=g).bfloat16()
    refw = dy.double().t() @ x.double()
    c = torch.zeros(768, 512, device="cuda")
    ops.gemm(dev(dy), dev(x), out=c, trans_a=True, trans_b=True, atomic=True, k_split=4, extra_flags=GEMM_TILE256)
    assert rel_err(c, refw) < 1e-5
    ops.gemm(dev(dy), dev(x), out=c, trans_a=True, trans_b=True, atomic=True, k_split=0)
    assert rel_err(c, 2 * refw) < 1e-5


def test_gemm_batched_strided_attention_shapes(ops):
    """Q K^T and P V as batched GEMMs straight out of the packed qkv tensor (two-level batch strides)."""
    n, s, h, hd = 2, 256, 2, 128
    d = h * hd
    g = torch.Generator().manual_seed(9)
    qkv = (torch.randn(n * s, 3 * d, generator=g) * 0.5).bfloat16()
    qd = dev(qkv)
    scores = torch.empty(n * h, s, s, device="cuda", dtype=torch.float32)
    ops.gemm_raw(qd, qd[:, d:], scores, s, s, hd, 3 * d, 3 * d, s, batch=n * h, batch_inner=h,
                 sA=(s * 3 * d, hd), sB=(s * 3 * d, hd), sC=(h * s * s, s * s))
    q, k, v = qkv.double().reshape(n, s, 3, h, hd).permute(2, 0, 3, 1, 4)
    ref = (q @ k.transpose(-1, -2)).reshape(n * h, s, s)
    assert rel_err(scores, ref) < 1e-5
    p = torch.softmax(ref / math.sqrt(hd), -1).bfloat16()
    out = torch.empty(n * s, d, device="cuda", dtype=torch.bfloat16)
    from pero_pretraining_amd._lib import GEMM_TRANS_B
    ops.gemm_raw(dev(p), qd[:, 2 * d:], out, s, hd, s, s, 3 * d, d, batch=n * h, batch_inner=h,
                 sA=(h * s * s, s * s), sB=(s * 3 * d, hd), sC=(s * d, hd), flags=GEMM_TRANS_B)
    ref_o = (p.double().reshape(n, h, s, s) @ v).permute(0, 2, 1, 3).reshape(n * s, d)
    assert rel_err(out, ref_o) < 2 ** -8


@pytest.mark.parametrize("n,s,h", [(2, 256, 2), (1, 128, 4), (2, 384, 1)])
def test_fused_attention_fwd_bwd(ops, n, s, h):
    """Flash-style kernels (scores never stored) vs the oracle's attention on the same bf16 inputs (f64), and
    the gradients through it; exercises 1, 2 and 3 key tiles (online softmax rescale)."""
    hd = 128
    d = h * hd
    g = torch.Generator().manual_seed(n * 1000 + s + h)
    qkv = (torch.randn(n * s, 3 * d, generator=g) * 0.7).bfloat16()
    qkv[5, :d] *= 6.0          # a spiked query row: forces a large running-max jump between key tiles
    qkv[s // 2 + 3, d:2 * d] *= 6.0
    dout = torch.randn(n * s, d, generator=g).bfloat16()
    ref_in = qkv.double().requires_grad_(True)
    ref = O.attention(ref_in, n, s, h)
    ref.backward(dout.double())
    out, lse = ops.attention_fwd_fused(dev(qkv), n, s, h)
    assert rel_err(out, ref.detach()) < 2 ** -7
    # base-2 LSE of the scaled scores
    q, k, _ = qkv.double().reshape(n, s, 3, h, hd).permute(2, 0, 3, 1, 4)
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    lse_ref = torch.logsumexp(sc, -1) / math.log(2.0)
    assert float((lse.cpu().double().reshape(n, h, s) - lse_ref).abs().max()) < 2e-3
    dqkv = ops.attention_bwd_fused(dev(qkv), out, dev(dout), lse, n, s, h)
    gref = ref_in.grad
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        err = rel_err(dqkv[:, sl], gref[:, sl])
        assert err < 3e-2, (name, err)
        a, b = dqkv[:, sl].double().cpu().flatten(), gref[:, sl].flatten()
        assert float(a @ b / (a.norm() * b.norm())) > 0.9995, name
    # fused in_proj bias gradient: the column sums of the stored dqkv, accumulated into the given vector
    dbias = torch.full((3 * d,), 2.0, device="cuda")
    dqkv2 = ops.attention_bwd_fused(dev(qkv), out, dev(dout), lse, n, s, h, dbias=dbias)
    assert torch.equal(dqkv2, dqkv)
    want = 2.0 + dqkv.float().sum(0)
    assert float((dbias - want).abs().max()) <= 1e-3 * max(1.0, float(want.abs().max()))
    # D = rowsum(dO * O) handed in (the step takes it from the epilogue of the product that made dO): `out` is not read and both
    # backward kernels run as ONE launch with the four workgroups of a (line, head) side by side - same numbers, bit for bit
    # (D computed here exactly as the dQ kernel computes it: f32 products of the bf16 values, summed per head)
    from pero_pretraining_amd._lib import call
    dvec = (out.float() * dev(dout).float()).reshape(n * s, h, hd).sum(-1).contiguous()
    db_pair, db_two = torch.zeros(3 * d, device="cuda"), torch.zeros(3 * d, device="cuda")
    got_pair = ops.attention_bwd_fused(dev(qkv), None, dev(dout), lse, n, s, h, dbias=db_pair, dvec=dvec)
    call("pero_set_option", b"attn_bwd_pair", 0)
    try:
        got_two = ops.attention_bwd_fused(dev(qkv), None, dev(dout), lse, n, s, h, dbias=db_two, dvec=dvec)
    finally:
        call("pero_set_option", b"attn_bwd_pair", 1)
    assert torch.equal(got_pair, got_two)
    assert float((db_pair - db_two).abs().max()) <= 1e-3 * max(1.0, float(db_two.abs().max()))
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        assert rel_err(got_pair[:, sl], gref[:, sl]) < 3e-2, name


@pytest.mark.parametrize("n", [1, 3, 70])
def test_attention_backward_variants_give_the_same_bits(ops, n):
    """S = 256, D handed in, bias gradient wanted: the paired kernels with software-pipelined operand reads (default), their
    compiler-scheduled form (attn_pipe 0) and the persistent one-(line, head)-per-pass kernel (attn_lh 1; 70 lines x 4 heads = more
    units than CUs: the workgroups loop) must agree bit for bit in dqkv; the bias gradient up to the order of its f32 sums."""
    from pero_pretraining_amd._lib import call
    s, h, hd = 256, 4, 128
    d = h * hd
    g = torch.Generator().manual_seed(40 + n)
    qkv = dev((torch.randn(n * s, 3 * d, generator=g) * 0.7).bfloat16())
    dout = dev(torch.randn(n * s, d, generator=g).bfloat16())
    out, lse = ops.attention_fwd_fused(qkv, n, s, h)
    dvec = (out.float() * dout.float()).reshape(n * s, h, hd).sum(-1).contiguous()
    res = {}
    try:
        for name, pipe, lh in (("pipelined", 1, 0), ("compiler", 0, 0), ("line_head", 1, 1)):
            call("pero_set_option", b"attn_pipe", pipe)
            call("pero_set_option", b"attn_lh", lh)
            db = torch.zeros(3 * d, device="cuda")
            res[name] = (ops.attention_bwd_fused(qkv, None, dout, lse, n, s, h, dbias=db, dvec=dvec), db)
            o2, l2 = ops.attention_fwd_fused(qkv, n, s, h)
            assert torch.equal(o2, out) and torch.equal(l2, lse)
    finally:
        call("pero_set_option", b"attn_pipe", 1)
        call("pero_set_option", b"attn_lh", 0)
    ref, dbr = res["pipelined"]
    assert bool(torch.isfinite(ref.float()).all())
    for name in ("compiler", "line_head"):
        got, db = res[name]
        assert torch.equal(got, ref), name
        assert float((db - dbr).abs().max()) <= 1e-5 * max(1.0, float(dbr.abs().max())), name
    want = ref.float().sum(0)
    assert float((dbr - want).abs().max()) <= 1e-3 * max(1.0, float(want.abs().max()))


# ------------------------------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,d", [(37, 64), (64, 256), (130, 512)])
def test_layernorm_fwd_bwd(ops, dtype, rows, d):
    g = torch.Generator().manual_seed(rows + d)
    S = 1 if rows % 2 else rows // 2
    x = (torch.randn(rows, d, generator=g) * 2 + 0.3).to(dtype)
    gamma = torch.randn(d, generator=g)
    beta = torch.randn(d, generator=g)
    pe = O.positional_table(d, 512)
    offsets = torch.randint(0, 512 - S, (rows // S,), generator=g)
    xr = x.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    y_ref = O.add_positional(O.layer_norm(xr, gr, br), rows // S, S, pe, offsets)
    dy = torch.randn(rows, d, generator=g).to(dtype)
    y_ref.backward(dy.float())
    y, mean, rstd = ops.layernorm_fwd(dev(x), dev(gamma), dev(beta), 1e-5, pe=dev(pe), offsets=dev(offsets), S=S)
    tol = 1e-5 if dtype == torch.float32 else 2 ** -7
    assert rel_err(y, y_ref.detach()) < tol
    dg = torch.zeros(d, device="cuda")
    db = torch.zeros(d, device="cuda")
    dxs = torch.zeros(d, device="cuda")
    dx = ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(gamma), dg, db, dxs)
    assert rel_err(dx, xr.grad) < (2e-5 if dtype == torch.float32 else 2 ** -6)
    assert rel_err(dg, gr.grad) < 1e-4
    assert rel_err(db, br.grad) < 1e-4
    assert rel_err(dxs, xr.grad.sum(0)) < 1e-4  # column sums are taken in f32 before dx is rounded


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,d", [(37, 64), (300, 512), (4096 + 6, 512), (50, 1024)])
def test_layernorm_bwd_from_output(ops, dtype, rows, d):
    """pero_layernorm_bwd_out: the backward from the layer's OUTPUT t = xhat * gamma + beta and rstd (the input rows are not kept).  f32: the
    gradients of the autograd oracle to rounding; bf16: t carries one more 2^-9 rounding than the input-row form, so dx is held to a few
    bf16 steps and the column sums to 1e-2 of their range.  A column with gamma == 0 has lost its xhat (t = beta there): its gamma gradient
    and its own dx column (the -xhat * c2 term) are those of xhat = 0, every other column is unaffected, nothing is NaN - the documented limit
    of the form (pero_hip.h)."""
    g = torch.Generator().manual_seed(7 * rows + d)
    x = (torch.randn(rows, d, generator=g) * 2 + 0.3).to(dtype)
    gamma = torch.rand(d, generator=g) + 0.5
    gamma[1::7] *= -1.0
    beta = torch.randn(d, generator=g) * 0.3
    xr, gr, br = x.float().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = O.layer_norm(xr, gr, br)
    dy = torch.randn(rows, d, generator=g).to(dtype)
    y_ref.backward(dy.float())
    t, mean, rstd = ops.layernorm_fwd(dev(x), dev(gamma), dev(beta), 1e-5)
    dg, db, dxs = (torch.zeros(d, device="cuda") for _ in range(3))
    dx = ops.layernorm_bwd_out(dev(dy), t, rstd, dev(gamma), dev(beta), dg, db, dxs)
    f32 = dtype == torch.float32
    assert rel_err(dx, xr.grad) < (1e-4 if f32 else 2 ** -5)
    assert rel_err(dg, gr.grad) < (1e-4 if f32 else 1e-2)
    assert rel_err(db, br.grad) < 1e-4
    assert rel_err(dxs, xr.grad.sum(0)) < (1e-3 if f32 else 2e-2)
    # ... and next to the input-row form of the same library
    dg2, db2, dxs2 = (torch.zeros(d, device="cuda") for _ in range(3))
    dx2 = ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(gamma), dg2, db2, dxs2)
    assert rel_err(dx, dx2.float().cpu()) < (1e-4 if f32 else 2 ** -5)
    assert torch.equal(db, db2)
    if f32:
        gz = gamma.clone()
        gz[3] = 0.0
        t, mean, rstd = ops.layernorm_fwd(dev(x), dev(gz), dev(beta), 1e-5)
        dg, db, dxs = (torch.zeros(d, device="cuda") for _ in range(3))
        dx = ops.layernorm_bwd_out(dev(dy), t, rstd, dev(gz), dev(beta), dg, db, dxs)
        dg2, db2, dxs2 = (torch.zeros(d, device="cuda") for _ in range(3))
        dx2 = ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(gz), dg2, db2, dxs2)
        keep = torch.ones(d, dtype=torch.bool)
        keep[3] = False
        assert bool(torch.isfinite(dx).all()) and rel_err(dx[:, keep.cuda()], dx2[:, keep.cuda()].cpu()) < 1e-4
        assert rel_err(dg[keep.cuda()], dg2[keep.cuda()].cpu()) < 1e-4 and float(dg[3]) == 0.0


@pytest.mark.parametrize("rows,d", [(4096 + 13, 512), (8192, 256), (5000, 504)])
def test_layernorm_fwd_four_rows_per_wave_equals_one_row_per_wave(ops, rows, d):
    """bf16, d <= 512, >= 4096 rows, no positional table: the four-rows-per-wave kernel - the rows, means and rstds it writes are those
    of the one-row-per-wave kernel (which a call on < 4096 rows takes), bit for bit; and they match the oracle."""
    g = torch.Generator().manual_seed(rows + d)
    x = dev((torch.randn(rows, d, generator=g) * 2 + 0.3).bfloat16())
    gamma, beta = dev(torch.randn(d, generator=g)), dev(torch.randn(d, generator=g))
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    for lo in range(0, rows, 4000):   # chunks below the threshold
        y1, m1, r1 = ops.layernorm_fwd(x[lo:lo + 4000].contiguous(), gamma, beta, 1e-5)
        assert torch.equal(y[lo:lo + 4000], y1) and torch.equal(mean[lo:lo + 4000], m1) and torch.equal(rstd[lo:lo + 4000], r1)
    y_ref = O.layer_norm(x.float().cpu(), gamma.cpu(), beta.cpu())
    assert rel_err(y, y_ref) < 2 ** -7


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_softmax_fwd_bwd(ops, dtype):
    g = torch.Generator().manual_seed(1)
    s = torch.randn(50, 100, generator=g) * 4
    sr = s.clone().requires_grad_(True)
    p_ref = torch.softmax(sr * 0.3, -1)
    dp = torch.randn(50, 100, generator=g)
    p_ref.backward(dp)
    p = ops.softmax_fwd(dev(s), 0.3, dtype)
    assert rel_err(p, p_ref.detach()) < (1e-6 if dtype == torch.float32 else 2 ** -8)
    ds = ops.softmax_bwd(p, dev(dp), 0.3)
    assert rel_err(ds, sr.grad) < (1e-5 if dtype == torch.float32 else 2 ** -6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("uw", [None, 0.25])
def test_masked_ce(ops, dtype, uw):
    g = torch.Generator().manual_seed(2)
    rows, V = 60, 333
    logits = (torch.randn(rows, V, generator=g) * 3).to(dtype)
    labels = torch.randint(0, V, (rows,), generator=g)
    labels[50:] = -1
    mask = (torch.rand(rows, generator=g) < 0.3).long() * (labels >= 0)
    lr = logits.float().requires_grad_(True)
    loss_ref = O.masked_cross_entropy(lr[None], labels[None], mask[None], uw)
    loss_ref.backward()
    loss, work = ops.masked_ce_fwd(dev(logits), dev(labels), dev(mask), uw)
    assert abs(float(loss) - float(loss_ref)) < 2e-6 * abs(float(loss_ref))
    dl = ops.masked_ce_bwd(dev(logits), dev(labels), dev(mask), work, uw)
    assert rel_err(dl, lr.grad) < (1e-5 if dtype == torch.float32 else 2 ** -7)
    dl2 = ops.masked_ce_bwd(dev(logits), dev(labels), dev(mask), work, uw, dloss=torch.full((1,), 0.5, device="cuda"))
    assert rel_err(dl2, 0.5 * lr.grad) < (1e-5 if dtype == torch.float32 else 2 ** -7)
    if uw is None:   # the compact forward (pero_masked_ce_fwd_rows: only the listed mask == 1 rows are visited): the same bits
        loss_c, work_c = ops.masked_ce_fwd_rows(dev(logits), dev(labels), dev(mask), dev(torch.nonzero(mask == 1).reshape(-1)))
        assert torch.equal(loss_c, loss)
        assert torch.equal(ops.masked_ce_bwd(dev(logits), dev(labels), dev(mask), work_c, uw), dl)
    # compact form (pero_masked_ce_bwd_rows): the listed rows bit for bit, zero padding rows behind; a listed row that takes no
    # part in the loss is a zero row
    sel = torch.nonzero(mask == 1).reshape(-1)
    extra = torch.nonzero((mask == 0) & (labels < 0)).reshape(-1)[:2]          # take no part under either setting
    index = torch.cat([sel, extra])
    n_out = ((index.numel() + 15) // 16) * 16 + 16
    rows_c = ops.masked_ce_bwd_rows(dev(logits), dev(labels), dev(mask), work, dev(index), n_out, uw)
    assert rows_c.shape == (n_out, V)
    assert torch.equal(rows_c[:sel.numel()], dl[dev(sel)])
    assert float(rows_c[sel.numel():].float().abs().max()) == 0.0


def test_masked_ce_empty_mask_is_nan(ops):
    logits = torch.randn(8, 16).cuda()
    labels = torch.zeros(8, dtype=torch.long).cuda()
    mask = torch.zeros(8, dtype=torch.long).cuda()
    loss, _ = ops.masked_ce_fwd(logits, labels, mask)
    assert math.isnan(float(loss))  # reference: cross_entropy over an empty selection is NaN


def test_masked_ce_label_outside_the_head_is_nan_not_a_fault(ops):
    """8192-code labels against a 4096-way head (or a negative label under a set mask bit): F.cross_entropy raises in the
    reference (masked_pretraining/model.py:82); here the loss and the row's gradient are NaN - no out-of-bounds read."""
    g = torch.Generator().manual_seed(9)
    logits = torch.randn(12, 4096, generator=g).bfloat16().cuda()
    labels = torch.randint(0, 4096, (12,), generator=g).cuda()
    mask = torch.ones(12, dtype=torch.long).cuda()
    ok, work = ops.masked_ce_fwd(logits, labels, mask)
    assert math.isfinite(float(ok))
    for bad in (5000, -1):
        lab = labels.clone()
        lab[3] = bad
        loss, work = ops.masked_ce_fwd(logits, lab, mask)
        assert math.isnan(float(loss))
        dl = ops.masked_ce_bwd(logits, lab, mask, work)
        assert torch.isnan(dl[3].float()).all() and torch.isfinite(dl[2].float()).all()


def test_colsum_cast_scale(ops):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(300, 1000, generator=g)
    out = torch.ones(1000, device="cuda")
    ops.colsum(dev(x), out)
    assert rel_err(out, x.double().sum(0) + 1) < 1e-5
    xb = dev(x.bfloat16())
    out.zero_()
    ops.colsum(xb[:, :512], out[:512])
    assert rel_err(out[:512], x.bfloat16().double()[:, :512].sum(0)) < 1e-5
    flat = torch.randn(100003, generator=g)
    padded = torch.zeros(100008)
    padded[:100003] = flat
    dst = torch.empty(100008, device="cuda", dtype=torch.bfloat16)
    ops.cast_to_bf16(dev(padded), dst)
    assert torch.equal(dst.cpu(), padded.bfloat16())
    y = dev(flat.clone())
    ops.scale_(y, 0.5)
    assert torch.equal(y.cpu(), flat * 0.5)


def test_adam_matches_oracle(ops):
    g = torch.Generator().manual_seed(6)
    n = 10007
    p = torch.randn(n, generator=g)
    pd, md, vd = dev(p.clone()), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pb = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    po, mo, vo = p.clone(), torch.zeros(n), torch.zeros(n)
    for step in range(1, 5):
        gr = torch.randn(n, generator=g) * (0.1 if step % 2 else 10)
        O.adam_update(po, gr, mo, vo, step, 1e-3 * step)
        ops.adam_step(pd, dev(gr), md, vd, pb, 1e-3 * step, 0.9, 0.999, 1e-8, step)
    assert rel_err(pd, po) < 1e-6
    assert rel_err(md, mo) < 1e-6 and rel_err(vd, vo) < 2e-6
    assert torch.equal(pb.cpu(), pd.cpu().bfloat16())


# ------------------------------------------------------------------------------------------ front end
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,W", [(3, 128), (2, 200), (1, 8)])
def test_patches(ops, dtype, N, W):
    rng = np.random.default_rng(N * 100 + W)
    images = rng.integers(0, 256, (N, 40, W, 3), dtype=np.uint8)
    mask = (rng.random((N, W // 8)) < 0.3).astype(np.int64)
    tile = O.mask_tile()
    x = O.prepare_images(torch.from_numpy(images))
    ref = O.patches(O.apply_mask(x, mask, tile))
    out = ops.patches_from_u8(dev(images), dev(mask), dev(tile), 8, dtype)
    if dtype == torch.float32:
        assert torch.equal(out.cpu(), ref)  # u8/255 in f32: bit exact
    else:
        assert torch.equal(out.cpu(), ref.bfloat16())
    out2 = ops.patches_from_f32(dev(x), dev(mask), dev(tile), 8, dtype)
    assert torch.equal(out2.cpu(), ref.to(dtype))
    out3 = ops.patches_from_u8(dev(images), None, None, 8, torch.float32)
    assert torch.equal(out3.cpu(), O.patches(x))
    outp = ops.patches_from_u8(dev(images), dev(mask), dev(tile), 8, dtype, pitch=1024)
    assert torch.equal(outp[:, :960].cpu(), ref.to(dtype)) and float(outp[:, 960:].float().abs().sum()) == 0
    outp = ops.patches_from_f32(dev(x), dev(mask), dev(tile), 8, dtype, pitch=1024)
    assert torch.equal(outp[:, :960].cpu(), ref.to(dtype)) and float(outp[:, 960:].float().abs().sum()) == 0
    xm = dev(x.contiguous())
    ops.apply_mask_(xm, dev(mask), dev(tile), 8)
    assert torch.equal(xm.cpu(), O.apply_mask(x, mask, tile))


# ------------------------------------------------------------------------------------------ quantizer
def test_vq_argmin_small_and_ragged(ops, golden):
    g = golden("g6_quantizers.npz")
    flat = np.ascontiguousarray(g["small.features"].transpose(0, 2, 3, 1)).reshape(-1, 32)
    idx = ops.vq_argmin(dev(flat), dev(g["small.codebook"]))
    assert np.array_equal(idx.cpu().numpy(), g["small.indices"])  # reference-generated indices, bit exact
    q = ops.vq_gather(dev(flat), dev(g["small.codebook"]), idx)
    qo, _ = O.vq_quantize(g["small.features"], g["small.codebook"])
    assert np.array_equal(q.cpu().numpy().reshape(2, 1, 100, 32).transpose(0, 3, 1, 2), qo)
    rng = np.random.default_rng(0)  # ragged sizes, duplicate codes (ties -> first index)
    x = rng.standard_normal((77, 19)).astype(np.float32)
    e = rng.standard_normal((130, 19)).astype(np.float32)
    e[100] = e[3]
    x[5] = e[3]
    idx, best = ops.vq_argmin(dev(x), dev(e), want_dist=True)
    ref, dist = O.vq_nearest(x, e)
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert int(idx[5]) == 3


def test_vq_argmin_codebook_8192(ops, golden):
    g = golden("g6_quantizers.npz")
    torch.manual_seed(5)
    w = torch.nn.Embedding(8192, 512).weight.data
    w.normal_()
    flat = np.ascontiguousarray(g["cb8192.features"].transpose(0, 2, 3, 1)).reshape(-1, 512)
    idx, best = ops.vq_argmin(dev(flat), dev(w), want_dist=True)
    # (round 1 exempted relative margins below 1e-4; the kernel reproduces the reference's decisions down to 1e-6 - g17 pins 2e-6)
    near_tie = (g["cb8192.second"] - g["cb8192.best"]) < 1e-6 * np.abs(g["cb8192.best"])
    got = idx.cpu().numpy()
    assert np.array_equal(got[~near_tie], g["cb8192.indices"][~near_tie])
    assert (got != g["cb8192.indices"]).sum() <= near_tie.sum()
    assert np.abs(best.cpu().numpy() - g["cb8192.best"]).max() < 1e-3


def test_vq_argmin_fast_path_matches_generic_and_oracle(ops):
    """LDS-DMA / b128-fragment kernel (M % 64, K % 128, D % 32 == 0) vs the f32 oracle; ties -> first index."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((192, 96)).astype(np.float32)
    e = rng.standard_normal((384, 96)).astype(np.float32)
    e[300] = e[7]
    x[9] = e[7]
    idx, best = ops.vq_argmin(dev(x), dev(e), want_dist=True)
    ref, dist = O.vq_nearest(x, e)
    b, s2 = O.margins(dist)
    near = (s2 - b) < 1e-5 * np.abs(b) + 1e-6
    assert np.array_equal(idx.cpu().numpy()[~near], ref[~near])
    assert int(idx[9]) == 7
    assert np.abs(best.cpu().numpy() - b).max() < 1e-3
    # ragged sizes still take the generic kernel and agree on the common prefix
    idx2 = ops.vq_argmin(dev(x[:100]), dev(e))
    assert np.array_equal(idx2.cpu().numpy()[~near[:100]], ref[:100][~near[:100]])


def test_gather_scatter(ops):
    g = torch.Generator().manual_seed(8)
    for dtype in (torch.float32, torch.bfloat16):
        src = torch.randn(50, 96, generator=g).to(dtype)
        index = torch.randperm(50, generator=g)[:20]
        out = ops.gather_rows(dev(src), dev(index), 24)
        assert torch.equal(out[:20].cpu(), src[index]) and float(out[20:].float().abs().sum()) == 0
        dst = dev(src.clone())
        ops.scatter_add_rows(out, dev(index), dst)
        ref = src.clone().float()
        ref[index] += src[index].float()
        assert torch.equal(dst.cpu(), ref.to(dtype))


@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (256, 128, 64), (192, 136, 72)])
@pytest.mark.parametrize("with_gate", [False, True])
def test_gemm_fused_column_sums(M, N, K, with_gate):
    """PERO_GEMM_COLSUM: the bias gradient of the upstream Linear out of the dX product's epilogue (tile kernels) or the
    library's fallback pass (other shapes); accumulates into the output vector."""
    from pero_pretraining_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    dy = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(K, N, device="cuda", generator=g) * 0.5).bfloat16()
    gate = (torch.randn(M, N, device="cuda", generator=g)).bfloat16() if with_gate else None
    acc = torch.full((N,), 3.0, device="cuda")
    out = ops.gemm(dy, w, trans_b=True, gate=gate, colsum_into=acc)
    ref = ops.gemm(dy, w, trans_b=True, gate=gate)
    assert torch.equal(out, ref)                                  # the stored result is unchanged by the fused reduction
    want = 3.0 + ref.float().sum(0)
    err = (acc - want).abs().max().item()
    assert err <= 2e-2 * max(1.0, want.abs().max().item()), err   # f32 sums of pre-rounding values vs sums of bf16 values


def test_patches_divide_every_byte_value_exactly():
    """The front end's x / 255 (reciprocal multiply + one fma correction) equals the reference's IEEE division for all 256
    byte values, in f32 and after the bf16 rounding."""
    from pero_pretraining_amd import ops
    vals = torch.arange(256, dtype=torch.uint8)
    img = vals.repeat(40 * 8 * 3 * 2)[: 40 * 16 * 3].reshape(1, 40, 16, 3).contiguous()   # (N=1, H=40, W=16, C=3): 2 patches
    img[0, :, :, 0] = vals[:16][None, :]; img[0, 0, :, 1] = vals[100:116]; img[0, 1, :, 2] = vals[240:256]
    for rows in range(40):
        img[0, rows, :, 1] = vals[(rows * 16) % 256:(rows * 16) % 256 + 16] if (rows * 16) % 256 + 16 <= 256 else vals[:16]
    want = (img.float().permute(0, 3, 1, 2) / 255.0)                                      # reference arithmetic (f32 division)
    ref_rows = want.reshape(1, 3, 40, 2, 8).permute(0, 3, 1, 2, 4).reshape(2, 960)         # (token, c*h*p)
    got = ops.patches_from_u8(img.cuda(), None, None, 8, torch.float32)
    assert torch.equal(got.cpu(), ref_rows)
    got16 = ops.patches_from_u8(img.cuda(), None, None, 8, torch.bfloat16, pitch=1024)
    assert torch.equal(got16[:, :960].cpu(), ref_rows.bfloat16()) and float(got16[:, 960:].abs().max()) == 0.0
    assert set(img.reshape(-1).tolist()) == set(range(256))


@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (256, 512, 64), (192, 128, 72)])
def test_gemm_fused_row_dots(M, N, K):
    """PERO_GEMM_ROWDOT: per-128-column-block row dots of the stored result with a second matrix (the attention backward's D out
    of the epilogue of the product that writes dO), tile kernel and fallback pass."""
    from pero_pretraining_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M * 3 + N + K)
    dy = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(K, N, device="cuda", generator=g) * 0.5).bfloat16()
    y = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    dots = torch.full((M, N // 128), 7.0, device="cuda")
    out = ops.gemm(dy, w, trans_b=True, rowdot=(y, dots))
    ref = ops.gemm(dy, w, trans_b=True)
    assert torch.equal(out, ref)
    want = (ref.float() * y.float()).view(M, N // 128, 128).sum(-1)
    assert float((dots - want).abs().max()) <= 1e-3 * max(1.0, float(want.abs().max()))


def test_transpose_multi_and_transposed_weight_copies():
    """pero_transpose_multi: every matrix of a flat bf16 buffer transposed in one launch (aligned 16-byte path, ragged
    scalar path, sizes that are not multiples of the 64 x 64 tile); lowp.weight_t follows parameter updates."""
    from pero_pretraining_amd import lowp, ops
    shapes = [(1536, 512), (64, 64), (100, 36), (8, 1000), (3, 5), (520, 72)]
    offs, total = [], 0
    for r, c in shapes:
        offs.append(total)
        total += ((r * c + 7) // 8) * 8
    flat = torch.randn(total, device="cuda").to(torch.bfloat16)
    out = torch.zeros_like(flat)
    table, tiles = ops.transpose_table([(o, o, r, c) for o, (r, c) in zip(offs, shapes)], flat.device)
    ops.transpose_multi(flat, out, table, tiles)
    for o, (r, c) in zip(offs, shapes):
        assert torch.equal(out[o:o + r * c].view(c, r), flat[o:o + r * c].view(r, c).t()), (r, c)
    p = torch.nn.Parameter(torch.randn(256, 3, 8, 8, device="cuda"))
    wt = lowp.weight_t(p)
    assert wt.shape == (192, 256) and torch.equal(wt, p.detach().to(torch.bfloat16).view(256, 192).t())
    with torch.no_grad():
        p.mul_(2.0)
    assert torch.equal(lowp.weight_t(p), p.detach().to(torch.bfloat16).view(256, 192).t())


@pytest.mark.parametrize("M,N,K", [(512, 256, 192), (256, 128, 64), (1024, 2048, 512)])
def test_gemm_relu_gate_as_bit_mask(M, N, K):
    """PERO_GEMM_RELU_BITS: the ReLU product leaves bit (n & 7) of byte [m][n / 8] = (stored value > 0); the gated product
    reads that mask instead of the bf16 activation - same bytes out as with the bf16 gate (e256 / r256 epilogues)."""
    from pero_pretraining_amd import _lib, ops
    torch.manual_seed(M + N)
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.2).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    dy = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    wt = (torch.randn(N, K, device="cuda") * 0.2).to(torch.bfloat16)  # (in = N, out = K) transposed copy of the next weight
    # (policy 20 sends every stored product to the eight-phase kernel whatever the tile count, policy 7 to the 256x128x32 one)
    for tile_flags, policy in ((0, 0), (0, 7), (0, 20)):
        _lib.lib().pero_set_option(b"gemm_policy", policy)
        bits = torch.zeros((M, N // 8), device="cuda", dtype=torch.uint8)
        h = ops.gemm(x, w, bias=b, relu=True, relu_bits=bits, extra_flags=tile_flags)
        h_ref = ops.gemm(x, w, bias=b, relu=True, extra_flags=tile_flags)
        assert torch.equal(h, h_ref)
        want = np.packbits((h.float() > 0).cpu().numpy(), axis=1, bitorder="little")
        assert np.array_equal(bits.cpu().numpy(), want)
        cs_a = torch.zeros(N, device="cuda"); cs_b = torch.zeros(N, device="cuda")
        for kw_a, kw_b in (({}, {}), ({"colsum_into": cs_a}, {"colsum_into": cs_b})):
            got = ops.gemm(dy, wt, relu_bits=bits, extra_flags=tile_flags, **kw_a)
            ref = ops.gemm(dy, wt, gate=h, extra_flags=tile_flags, **kw_b)
            assert torch.equal(got, ref)
        # (some epilogues add the f32 values before the rounding, others - and the fall-back pass - the stored bf16 ones: which
        #  one runs depends on the policy; both are within the rounding of the stored result of its exact column sums)
        want_cs = got.float().sum(0)
        tol = 4e-3 * float(got.float().abs().sum(0).max())
        assert float((cs_a - want_cs).abs().max()) <= tol and float((cs_b - want_cs).abs().max()) <= tol
        _lib.lib().pero_set_option(b"gemm_policy", 0)
        ref32 = torch.relu(x.float() @ w.float().t() + b)
        assert float((h.float() - ref32).abs().max()) <= 2e-2 * float(ref32.abs().max())  # (bias and ReLU are in it)
    with pytest.raises(RuntimeError):  # shapes outside the 256-row tile kernels are refused, not silently ungated
        ops.gemm(dy[:128], wt, relu_bits=bits[:128])
