"""The eight-phase persistent 256x256x64 bf16 GEMM (csrc/gemm_e.hip) against exact / f32 references and against the other
tile kernels (bit for bit): every operand layout, every fused epilogue, the split-K mode, many tiles per workgroup."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def e256():
    from pero_pretraining_amd import _lib, ops
    lib = _lib.lib()
    lib.pero_set_option(b"gemm_policy", 20)
    yield ops
    lib.pero_set_option(b"gemm_policy", 0)


def _setpol(p):
    from pero_pretraining_amd import _lib
    _lib.lib().pero_set_option(b"gemm_policy", p)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("K", [128, 192, 512])
def test_layouts_exact_integers(e256, ta, tb, K):
    """small integers: every partial sum is exact in f32 and the bf16 result is exact - any wrong fragment, tile or k-slice shows."""
    M, N = 768, 512
    g = torch.Generator().manual_seed(11 + K)
    a = torch.randint(-3, 4, (K, M) if ta else (M, K), generator=g).float()
    b = torch.randint(-3, 4, (K, N) if tb else (N, K), generator=g).float()
    ref = (a.t() if ta else a) @ (b.t() if tb else b).t()
    out = e256.gemm(a.cuda().bfloat16(), b.cuda().bfloat16(), trans_a=ta, trans_b=tb)
    assert torch.equal(out.float().cpu(), ref.bfloat16().float())


@pytest.mark.parametrize("M,N,K", [(1024, 512, 512), (1024, 1536, 192), (2048, 256, 128)])
def test_epilogues_match_the_other_tile_kernel_bit_for_bit(e256, M, N, K):
    """bias / ReLU / residual / bit mask out and in / column sums / row dots: same bytes as gemm_bf16_r256 (policy 7), whose
    epilogue is the f32 acc * alpha + bias (+ residual) -> one rounding that every tile kernel of the library implements."""
    ops = e256
    torch.manual_seed(M + N + K)
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").bfloat16()

    def both(fn):
        _setpol(20)
        a = fn()
        _setpol(7)
        b = fn()
        _setpol(20)
        return a, b

    for kw in ({}, {"bias": bias}, {"bias": bias, "relu": True}, {"bias": bias, "residual": res}, {"residual": res}):
        a, b = both(lambda: ops.gemm(x, w, **kw))
        assert torch.equal(a, b), kw
    ref = x.float() @ w.float().t()
    y = ops.gemm(x, w, bias=bias, residual=res)
    assert float((y.float() - (ref + bias + res.float())).abs().max()) <= 2e-2 * float(ref.abs().max())
    # ReLU bit mask out
    bits = torch.zeros((M, N // 8), device="cuda", dtype=torch.uint8)
    h = ops.gemm(x, w, bias=bias, relu=True, relu_bits=bits)
    assert torch.equal(h, ops.gemm(x, w, bias=bias, relu=True))
    want = np.packbits((h.float() > 0).cpu().numpy(), axis=1, bitorder="little")
    assert np.array_equal(bits.cpu().numpy(), want)
    # ... in, with and without the fused column sums
    cs = torch.full((N,), 3.0, device="cuda")
    gd = ops.gemm(x, w, relu_bits=bits, colsum_into=cs)
    _setpol(7)
    gr = ops.gemm(x, w, gate=h)
    _setpol(20)
    assert torch.equal(gd, gr) and torch.equal(gd, ops.gemm(x, w, relu_bits=bits))
    # ... and WITH an input bias (round-2 advisor finding: the e256 gate epilogue has no input-bias path and once dropped it silently; such a
    # product now goes to the 256x128x32 kernel at every tile count): the same bits under both policies, and the bias is really in them
    gb = ops.gemm(x, w, bias=bias, relu_bits=bits)
    _setpol(7)
    gb7 = ops.gemm(x, w, bias=bias, gate=h)
    _setpol(20)
    assert torch.equal(gb, gb7)
    keep = h.float() > 0
    assert float(((gb.float() - (ref + bias)) * keep).abs().max()) <= 2e-2 * float(ref.abs().max()) and float((gb.float() * ~keep).abs().max()) == 0.0
    want_cs = 3.0 + gd.float().sum(0)
    assert float((cs - want_cs).abs().max()) <= 1e-3 * max(1.0, float(want_cs.abs().max()))
    # row dots
    if N % 128 == 0:
        dots = torch.full((M, N // 128), 7.0, device="cuda")
        yd = ops.gemm(x, w, rowdot=(res, dots))
        assert torch.equal(yd, ops.gemm(x, w))
        dref = (yd.float() * res.float()).view(M, N // 128, 128).sum(-1)
        assert float((dots - dref).abs().max()) <= 1e-3 * max(1.0, float(dref.abs().max()))


def test_many_tiles_per_workgroup_and_repeatability(e256):
    """3072 tiles over 256 persistent workgroups (the cross-tile LDS-DMA stream, the counted waits past the previous epilogue's
    stores); ten launches give identical bytes (a missed wait shows as run-to-run differences)."""
    ops = e256
    torch.manual_seed(5)
    x = (torch.randn(131072, 512, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(1536, 512, device="cuda") * 0.5).bfloat16()
    res = torch.randn(131072, 1536, device="cuda").bfloat16()
    for kw in ({}, {"residual": res}):
        y0 = ops.gemm(x, w, **kw)
        for sl in (slice(0, 2048), slice(65536 - 1024, 65536 + 1024), slice(131072 - 2048, 131072)):
            ref = x[sl].float() @ w.float().t() + (res[sl].float() if kw else 0)
            assert float((y0[sl].float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
        for _ in range(10):
            assert torch.equal(ops.gemm(x, w, **kw), y0)
        _setpol(7)
        assert torch.equal(ops.gemm(x[:4096], w, **({"residual": res[:4096]} if kw else {})), y0[:4096])
        _setpol(20)


@pytest.mark.parametrize("N,K,rows", [(512, 512, 8192), (1536, 512, 65536), (768, 256, 16384)])
def test_split_k_weight_gradient_mode(e256, N, K, rows):
    """TT split-K (f32 atomics through the LDS-staged epilogue): XCD-aligned and plain slice counts, uneven slices, alpha."""
    ops = e256
    g = torch.Generator(device="cuda").manual_seed(N + K)
    dy = (torch.randn(rows, N, device="cuda", generator=g) * 0.5).bfloat16()
    x = (torch.randn(rows, K, device="cuda", generator=g) * 0.5).bfloat16()
    ref = dy[:, :].double().t() @ x.double()
    for ks in (0, 1, 3, 8):
        c = torch.zeros(N, K, device="cuda")
        ops.gemm(dy, x, out=c, trans_a=True, trans_b=True, atomic=True, k_split=ks)
        assert float((c.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) * max(1, rows // 8192), ks
    c = torch.ones(N, K, device="cuda")
    ops.gemm(dy, x, out=c, trans_a=True, trans_b=True, atomic=True, k_split=0, alpha=0.5)
    assert float((c.double() - (1.0 + 0.5 * ref)).abs().max()) <= 1e-5 * float(ref.abs().max()) * max(1, rows // 8192)


def test_split_k_partial_tiles_are_summed_in_slice_order(e256):
    """The split-K products leave their partial tiles in a workspace and a second kernel adds them in slice order: two runs give the
    same bits (f32 atomics do not), the result is added to what C held, and it agrees with the atomic epilogue (option
    "splitk_workspace" 0) to rounding."""
    from pero_pretraining_amd._lib import call
    ops = e256
    torch.manual_seed(3)
    rows, N, K = 65536, 768, 512
    dy = (torch.randn(rows, N, device="cuda") * 0.5).bfloat16()
    x = (torch.randn(rows, K, device="cuda") * 0.5).bfloat16()
    runs = []
    for _ in range(3):
        c = torch.full((N, K), 0.25, device="cuda")
        ops.gemm(dy, x, out=c, trans_a=True, trans_b=True, atomic=True, k_split=0)
        runs.append(c)
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    call("pero_set_option", b"splitk_workspace", 0)
    try:
        ca = torch.full((N, K), 0.25, device="cuda")
        ops.gemm(dy, x, out=ca, trans_a=True, trans_b=True, atomic=True, k_split=0)
    finally:
        call("pero_set_option", b"splitk_workspace", 1)
    ref = 0.25 + dy[:, :64].float().t() @ x.float()
    assert float((runs[0][:64] - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    assert float((runs[0] - ca).abs().max()) < 1e-3 * float(ca.abs().max())


def test_strided_views_of_every_operand(e256):
    """Column slices of wider tensors as A, B, C, the residual and the row-dot matrix (leading dimensions larger than the row
    length): the epilogue's transposed lane layout addresses rows by their pitch - same bytes as the other tile kernel, and
    nothing outside the slice is touched."""
    ops = e256
    torch.manual_seed(5)
    M, N, K = 1024, 512, 256
    xa = (torch.randn(M, K + 64, device="cuda") * 0.5).bfloat16()
    wa = (torch.randn(N, K + 128, device="cuda") * 0.5).bfloat16()
    ra = torch.randn(M, N + 256, device="cuda").bfloat16()
    x, w, res = xa[:, 64:], wa[:, :K], ra[:, 128:128 + N]
    bias = torch.randn(N, device="cuda")
    outs = {}
    for pol in (20, 7):
        _setpol(pol)
        ca = torch.full((M, N + 512), 7.0, device="cuda").bfloat16()
        ops.gemm(x, w, out=ca[:, 256:256 + N], bias=bias, residual=res)
        dots = torch.zeros((M, N // 128), device="cuda")
        yd = ops.gemm(x, w, rowdot=(res, dots))
        outs[pol] = (ca, yd, dots)
    _setpol(20)
    assert torch.equal(outs[20][0], outs[7][0]) and torch.equal(outs[20][1], outs[7][1])
    assert float((outs[20][2] - outs[7][2]).abs().max()) <= 1e-3 * float(outs[7][2].abs().max())
    ca = outs[20][0]
    assert torch.all(ca[:, :256] == 7.0) and torch.all(ca[:, 256 + N:] == 7.0)
    ref = x.float() @ w.float().t() + bias + res.float()
    assert float((ca[:, 256:256 + N].float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def _tile_mask(bits, M, N):
    """row-major [M][N/8] -> PERO_GEMM_MASK_TILED [N/256][M][32]"""
    return bits.view(M, N // 256, 32).permute(1, 0, 2).contiguous().view(M, N // 8)


@pytest.mark.parametrize("M,N,K,policy", [(65536, 2048, 512, 20), (16384, 2048, 512, 20), (16384, 1024, 256, 20), (1024, 2048, 512, 7), (512, 512, 128, 7)])
def test_bit_mask_per_256_column_block_is_the_row_mask_rearranged(M, N, K, policy):
    """PERO_GEMM_MASK_TILED (round 4): the ReLU bit mask stored per 256-column block, [N/256][M][32 bytes], so that a tile's mask is whole cache lines.  Producer
    (ReLU + bias, mask out) and consumer (mask as the gate, with the column sums) on both 256-row tile kernels (policy 20: gemm_bf16_e256 - with its sequential
    walk at K <= 512 -, 7: gemm_bf16_r256): C bit-identical to the row-major form, the mask the same bits in the other order."""
    from pero_pretraining_amd import _lib, ops
    L = _lib.lib()
    torch.manual_seed(13)
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    L.pero_set_option(b"gemm_policy", policy)
    try:
        b0 = torch.zeros(M, N // 8, device="cuda", dtype=torch.uint8)
        b1 = torch.zeros(M, N // 8, device="cuda", dtype=torch.uint8)
        c0 = ops.gemm(x, w, bias=bias, relu=True, relu_bits=b0)
        c1 = ops.gemm(x, w, bias=bias, relu=True, relu_bits=b1, bits_tiled=True)
        assert torch.equal(c0, c1)
        assert torch.equal(_tile_mask(b0, M, N), b1)
        want = torch.zeros_like(b0)
        pos = (c0 > 0).view(M, N // 8, 8).to(torch.int32)
        want = (pos * (2 ** torch.arange(8, device="cuda", dtype=torch.int32))).sum(-1).to(torch.uint8)
        assert torch.equal(b0, want)
        g = torch.randint(0, 256, (M, N // 8), device="cuda", dtype=torch.uint8)
        cs0, cs1 = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        d0 = ops.gemm(x, w, relu_bits=g, colsum_into=cs0)
        d1 = ops.gemm(x, w, relu_bits=_tile_mask(g, M, N), colsum_into=cs1, bits_tiled=True)
        assert torch.equal(d0, d1)
        assert float((cs0 - cs1).abs().max()) <= 1e-4 * float(cs0.abs().max()) + 1e-2
        for _ in range(3):
            assert torch.equal(ops.gemm(x, w, relu_bits=_tile_mask(g, M, N), bits_tiled=True), d0)
    finally:
        L.pero_set_option(b"gemm_policy", 0)


@pytest.mark.parametrize("M,N,K", [(16384, 2048, 512), (32768, 1536, 512), (16384, 4096, 256), (65536, 2048, 512)])
def test_sequential_walk_of_a_row_panel_gives_the_same_bits(e256, M, N, K):
    """Stored K <= 512 products walk the N-tiles of a 256-row panel partly one after the other (round 4, pero_launch_gemm_e256: `gemm_e_walk`, default on for the
    plain / ReLU / bit-mask-gate epilogues): a different tile -> workgroup order and nothing else, so the output must equal the side-by-side order's bit for bit,
    every tile written exactly once (a guard band around the output view stays untouched), column sums equal up to the order of the atomics."""
    ops = e256
    from pero_pretraining_amd._lib import call
    torch.manual_seed(12)
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    gbits = torch.randint(0, 256, (M, N // 8), device="cuda", dtype=torch.uint8)

    def run(kind, walk):
        call("pero_set_option", b"gemm_e_walk", walk)
        ca = torch.full((M, N + 256), 7.0, device="cuda").bfloat16()
        out = ca[:, 128:128 + N]
        cs = torch.zeros(N, device="cuda")
        if kind == "plain":
            ops.gemm(x, w, out=out)
        elif kind == "bias":
            ops.gemm(x, w, out=out, bias=bias)
        elif kind == "relu":
            ops.gemm(x, w, out=out, bias=bias, relu=True)
        elif kind == "gate":
            ops.gemm(x, w, out=out, relu_bits=gbits)
        else:
            ops.gemm(x, w, out=out, relu_bits=gbits, colsum_into=cs)
        assert torch.all(ca[:, :128] == 7.0) and torch.all(ca[:, 128 + N:] == 7.0)
        return out.clone(), cs

    try:
        for kind in ("plain", "bias", "relu", "gate", "gate_colsum"):
            want, wcs = run(kind, 0)
            for _ in range(2):
                got, gcs = run(kind, 1)
                assert torch.equal(got, want), kind
                assert float((gcs - wcs).abs().max()) <= 1e-4 * float(wcs.abs().max()) + 1e-2, kind
        ref = x[:512].float() @ w.float().t()
        assert float((run("plain", 1)[0][:512].float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    finally:
        call("pero_set_option", b"gemm_e_walk", 1)


@pytest.mark.parametrize("M,N,K", [(32768, 512, 192), (16384, 1024, 512), (65536, 2048, 512), (32768, 1536, 1024)])
def test_two_workgroups_per_cu_tile_gives_the_bits_of_the_256x256_tile(e256, M, N, K):
    """gemm_bf16_d128 (opt-in "gemm_d128": 256 x 128 tiles, four waves per workgroup, two independent workgroups per CU - round 4's measurement of
    'the epilogue of one tile under the main loop of another', slower than gemm_bf16_e256 and therefore off): same K-tile and MFMA order per
    output, so plain / bias / ReLU / bit-mask-out / bit-mask-gate products must agree bit for bit with the 256 x 256 tile; repeated launches give
    identical bytes (its counted vmcnt waits); strided views of every operand."""
    ops = e256
    from pero_pretraining_amd._lib import call
    torch.manual_seed(11)
    xa = (torch.randn(M, K + 64, device="cuda") * 0.5).bfloat16()
    wa = (torch.randn(N, K + 128, device="cuda") * 0.5).bfloat16()
    x, w = xa[:, 64:], wa[:, :K]
    bias = torch.randn(N, device="cuda")
    gbits = torch.randint(0, 256, (M, N // 8), device="cuda", dtype=torch.uint8)

    def run(kind):
        ca = torch.full((M, N + 256), 7.0, device="cuda").bfloat16()
        out = ca[:, 128:128 + N]
        bits = torch.zeros(M, N // 8, device="cuda", dtype=torch.uint8)
        if kind == "plain":
            ops.gemm(x, w, out=out)
        elif kind == "bias":
            ops.gemm(x, w, out=out, bias=bias)
        elif kind == "relu":
            ops.gemm(x, w, out=out, bias=bias, relu=True)
        elif kind == "relu_bits":
            ops.gemm(x, w, out=out, bias=bias, relu=True, relu_bits=bits)
        else:
            ops.gemm(x, w, out=out, relu_bits=gbits)
        assert torch.all(ca[:, :128] == 7.0) and torch.all(ca[:, 128 + N:] == 7.0)
        return out.clone(), bits

    try:
        for kind in ("plain", "bias", "relu", "relu_bits", "gate"):
            call("pero_set_option", b"gemm_d128", 0)
            want, wbits = run(kind)
            call("pero_set_option", b"gemm_d128", 64)
            for _ in range(3):
                got, gb = run(kind)
                assert torch.equal(got, want), kind
                assert torch.equal(gb, wbits), kind
        # the column sums of the gated product (linear1's bias gradient): f32 atomics in both kernels, equal up to the order of the additions
        cs0, cs1 = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        call("pero_set_option", b"gemm_d128", 0)
        a = ops.gemm(x, w, relu_bits=gbits, colsum_into=cs0)
        call("pero_set_option", b"gemm_d128", 64)
        b = ops.gemm(x, w, relu_bits=gbits, colsum_into=cs1)
        assert torch.equal(a, b)
        ref = a.float().sum(0)
        assert float((cs1 - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-2
        assert float((cs1 - cs0).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-2
    finally:
        call("pero_set_option", b"gemm_d128", 0)


@pytest.mark.parametrize("M,K", [(1024, 192), (2176, 512), (65536, 1536), (131072, 2048)])
def test_row_complete_tile_gives_the_bits_of_the_256x256_tile(e256, M, K):
    """gemm_bf16_n512 (opt-in "gemm_nw": one workgroup = 128 rows x all 512 columns, the tile the fused LayerNorm epilogues need): the same
    K-tile order and the same MFMA order per output as gemm_bf16_e256, so plain / bias / residual products must agree bit for bit (M % 256
    != 0, which the 256-row tile does not take: against the f32 product); repeated launches give identical bytes (a missed counted wait shows
    as run-to-run differences); operands and output as column-slice views (leading dimensions larger than the rows)."""
    ops = e256
    from pero_pretraining_amd import _lib
    L = _lib.lib()
    torch.manual_seed(9)
    N = 512
    xa = (torch.randn(M, K + 64, device="cuda") * 0.5).bfloat16()
    wa = (torch.randn(N, K + 128, device="cuda") * 0.5).bfloat16()
    ra = torch.randn(M, N + 256, device="cuda").bfloat16()
    x, w, res = xa[:, 64:], wa[:, :K], ra[:, 128:128 + N]
    bias = torch.randn(N, device="cuda")
    try:
        for kw in ({}, {"bias": bias}, {"residual": res}, {"bias": bias, "residual": res}):
            L.pero_set_option(b"gemm_nw", 0)
            want = ops.gemm(x, w, **kw) if M % 256 == 0 else None
            L.pero_set_option(b"gemm_nw", 1)
            ca = torch.full((M, N + 512), 7.0, device="cuda").bfloat16()
            got = ops.gemm(x, w, out=ca[:, 256:256 + N], **kw)
            if want is not None:
                assert torch.equal(got, want), kw
            else:
                ref = x.float() @ w.float().t() + (bias if "bias" in kw else 0) + (res.float() if "residual" in kw else 0)
                assert float((got.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max()), kw
            assert torch.all(ca[:, :256] == 7.0) and torch.all(ca[:, 256 + N:] == 7.0)
            for _ in range(5):
                assert torch.equal(ops.gemm(x, w, **kw), got), kw
    finally:
        L.pero_set_option(b"gemm_nw", 0)


@pytest.mark.parametrize("M,K", [(1024, 192), (2176, 512), (131072, 2048)])
def test_fused_linear_residual_layernorm(e256, M, K):
    """pero_gemm_resid_layernorm (gemm_bf16_n512 with the LayerNorm epilogue): y = the bits of the residual product, mean / rstd / t = what
    pero_layernorm_fwd computes from those rows (two-pass variance over the ROUNDED rows; other summation order: 1e-6 on the statistics, at
    most one bf16 step on a few t per million), repeatable; with and without the Linear's bias."""
    ops = e256
    from pero_pretraining_amd import _lib
    L = _lib.lib()
    torch.manual_seed(13)
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(512, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(512, device="cuda")
    res = torch.randn(M, 512, device="cuda").bfloat16()
    gamma = torch.rand(512, device="cuda") + 0.5
    beta = torch.randn(512, device="cuda") * 0.1
    assert ops.gemm_resid_layernorm_ok(x, w, res)
    for b in (bias, None):
        try:
            L.pero_set_option(b"gemm_nw", 1)      # the unfused pair with the product on the same tile kernel (it takes M % 128 == 0)
            y0 = ops.gemm(x, w, bias=b, residual=res)
        finally:
            L.pero_set_option(b"gemm_nw", 0)
        t0, m0, r0 = ops.layernorm_fwd(y0, gamma, beta, 1e-5)
        y, t, mean, rstd = ops.gemm_resid_layernorm(x, w, b, res, gamma, beta, 1e-5)
        assert torch.equal(y, y0)
        assert float((mean - m0).abs().max()) <= 1e-6 and float(((rstd - r0) / r0).abs().max()) <= 1e-5
        d = (t.float() - t0.float()).abs()
        assert float(d.max()) <= 2 ** -7 * float(t0.float().abs().max())          # one bf16 step at the top of the range
        assert int((d > 0).sum()) <= 1e-4 * t.numel()
        for _ in range(3):
            again = ops.gemm_resid_layernorm(x, w, b, res, gamma, beta, 1e-5)
            assert all(torch.equal(a, c) for a, c in zip(again, (y, t, mean, rstd)))
        # the launch that does not store y (the backward then reads t: pero_layernorm_bwd_out): the same t, mean, rstd, bit for bit
        yn, tn, mn, rn = ops.gemm_resid_layernorm(x, w, b, res, gamma, beta, 1e-5, store_y=False)
        assert yn is None and torch.equal(tn, t) and torch.equal(mn, mean) and torch.equal(rn, rstd)


@pytest.mark.parametrize("M,K", [(1024, 192), (2176, 1536), (131072 + 384, 2048), (65536, 512)])
def test_fused_input_gradient_layernorm_backward(e256, M, K):
    """pero_gemm_resid_layernorm_bwd (gemm_bf16_n512 with the LayerNorm-BACKWARD epilogue): dx, dgamma, dbeta, dxsum = what the pair
    pero_gemm (residual epilogue; dt stored in bf16) + pero_layernorm_bwd_out computes - the fused launch rounds dt the same way and uses the
    same formulas, only the summation orders of the row and column sums differ (a few dx per thousand by one bf16 step; column sums 1e-5 of
    their range); repeatable; strided operands; the column sums ACCUMULATE into their destinations."""
    ops = e256
    from pero_pretraining_amd import _lib
    L = _lib.lib()
    torch.manual_seed(17 + K)
    dy = (torch.randn(M, K + 64, device="cuda") * 0.5).bfloat16()[:, 64:]          # a column-slice view: lda > K
    wt = (torch.randn(512, K, device="cuda") * 0.05).bfloat16()
    res = torch.randn(M, 512 + 128, device="cuda").bfloat16()[:, :512]
    gamma = torch.rand(512, device="cuda") + 0.5
    gamma[5::11] *= -1.0
    beta = torch.randn(512, device="cuda") * 0.2
    y = (torch.randn(M, 512, device="cuda") * 1.5 + 0.2).bfloat16()
    t, mean, rstd = ops.layernorm_fwd(y, gamma, beta, 1e-5)
    assert ops.gemm_resid_layernorm_bwd_ok(dy, wt, res, t)
    try:
        L.pero_set_option(b"gemm_nw", 1)      # the unfused product on the same tile kernel (it takes M % 128 == 0)
        dt = ops.gemm(dy, wt, residual=res)
    finally:
        L.pero_set_option(b"gemm_nw", 0)
    dg0, db0, dx0s = (torch.zeros(512, device="cuda") for _ in range(3))
    dx0 = ops.layernorm_bwd_out(dt, t, rstd, gamma, beta, dg0, db0, dx0s)
    init = torch.arange(512, device="cuda", dtype=torch.float32) * 0.25
    dg, db, dxs = init.clone(), init.clone(), init.clone()
    dx = ops.gemm_resid_layernorm_bwd(dy, wt, res, t, rstd, gamma, beta, dg, db, dxs)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dx.float()).all())
    d = (dx.float() - dx0.float()).abs()
    top = float(dx0.float().abs().max())
    assert float(d.max()) <= 2 ** -7 * top, (float(d.max()), top)
    assert int((d > 0).sum()) <= 2e-2 * dx.numel(), int((d > 0).sum())
    for got, want, name in ((dg - init, dg0, "dgamma"), (db - init, db0, "dbeta"), (dxs - init, dx0s, "dxsum")):
        assert float((got - want).abs().max()) <= 2e-4 * max(1.0, float(want.abs().max())), name
    # ... and against f64 arithmetic on the stored dt
    xh = ((t.double() - beta.double()) / gamma.double())
    g = dt.double() * gamma.double()
    ref = rstd.double()[:, None] * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
    assert float((dx.double() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())
    assert float((dg - init - (dt.double() * xh).sum(0).float()).abs().max()) <= 1e-3 * float((dt.double() * xh).sum(0).abs().max())
    for _ in range(3):
        dg2, db2, dxs2 = init.clone(), init.clone(), init.clone()
        again = ops.gemm_resid_layernorm_bwd(dy, wt, res, t, rstd, gamma, beta, dg2, db2, dxs2)
        assert torch.equal(again, dx) and torch.equal(dg2, dg) and torch.equal(db2, db) and torch.equal(dxs2, dxs)


def test_fused_linear_residual_layernorm_on_strided_operands(e256):
    """pero_gemm_resid_layernorm through the C ABI with every leading dimension larger than its row: A and the residual as column slices of
    wider matrices, Y and T written into column slices of wider buffers - the same bits as the contiguous call, the neighbouring columns
    untouched (the LayerNorm epilogue's own stores and side loads use ldy / ldr / ldt, the row statistics stay contiguous)."""
    ops = e256
    from pero_pretraining_amd import _lib
    M, K = 1152, 640
    torch.manual_seed(23)
    xa = (torch.randn(M, K + 192, device="cuda") * 0.5).bfloat16()
    wa = (torch.randn(512, K + 64, device="cuda") * 0.05).bfloat16()
    ra = torch.randn(M, 512 + 256, device="cuda").bfloat16()
    x, w, res = xa[:, 128:128 + K], wa[:, :K], ra[:, 64:64 + 512]
    bias = torch.randn(512, device="cuda")
    gamma = torch.rand(512, device="cuda") + 0.5
    beta = torch.randn(512, device="cuda") * 0.1
    y0, t0, m0, r0 = ops.gemm_resid_layernorm(x.contiguous(), w.contiguous(), bias, res.contiguous(), gamma, beta, 1e-5)
    ya = torch.full((M, 512 + 384), 7.0, device="cuda").bfloat16()
    ta = torch.full((M, 512 + 128), 7.0, device="cuda").bfloat16()
    y, t = ya[:, 256:256 + 512], ta[:, 64:64 + 512]
    mean = torch.empty(M, device="cuda")
    rstd = torch.empty(M, device="cuda")
    for store_y in (True, False):
        ya.fill_(7.0); ta.fill_(7.0)
        _lib.call("pero_gemm_resid_layernorm", x.data_ptr(), w.data_ptr(), bias.data_ptr(), res.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                  y.data_ptr() if store_y else None, t.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, 512, K, x.stride(0), w.stride(0),
                  y.stride(0), res.stride(0), t.stride(0), 1e-5, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(t, t0) and torch.equal(mean, m0) and torch.equal(rstd, r0)
        assert torch.equal(y, y0) if store_y else bool(torch.all(y == 7.0))
        assert bool(torch.all(ya[:, :256] == 7.0)) and bool(torch.all(ya[:, 768:] == 7.0))
        assert bool(torch.all(ta[:, :64] == 7.0)) and bool(torch.all(ta[:, 576:] == 7.0))
