"""Tensor-level wrappers over the C ABI (libpero_hip.so).  PyTorch supplies device memory and the
current HIP stream; all arithmetic happens in the hand-written kernels.  No CPU fallback."""
import threading

import numpy as np
import torch

from . import _lib
from ._lib import (GEMM_ACCUM, GEMM_ATOMIC, GEMM_COLSUM, GEMM_FORCE_GENERIC, GEMM_MASK_TILED, GEMM_ROWDOT, GEMM_RELU, GEMM_RELU_BITS, GEMM_TRANS_A, GEMM_TRANS_B,
                   PERO_BF16, PERO_F32, call)


def dt(t_or_dtype):
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.float32:
        return PERO_F32
    if d == torch.bfloat16:
        return PERO_BF16
    raise TypeError(f"unsupported dtype {d}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.PeroHipError("pero_pretraining_amd ops need CUDA/HIP tensors: the HIP kernels are the only "
                                    "implementation (no CPU fallback)")


# optional launch timeline for bench.py's roofline leg: list of (start_event, end_event, flops, kernel tag)
gemm_timeline = None

# Split-K workspaces (partial tiles of the weight-gradient products, pero_gemm's `workspace`): the C ABI never allocates, so the
# caller - this module - keeps one buffer per (device, stream) from PyTorch's caching allocator, grown on demand.  Products on one
# stream run one after the other and may share it; the weight-gradient side stream and the autograd threads' streams get their own.
_ws_lock = threading.Lock()
_ws_cache = {}


def gemm_workspace(nbytes, device):
    """uint8 device buffer of at least `nbytes` for the current stream (None for 0)."""
    if nbytes <= 0:
        return None
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    with _ws_lock:
        buf = _ws_cache.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = _ws_cache[key] = torch.empty(max(int(nbytes), 64 << 20), device=device, dtype=torch.uint8)
    return buf


def release_workspaces():
    """Drop the cached split-K workspaces (they return to PyTorch's allocator once the queued work has run)."""
    with _ws_lock:
        _ws_cache.clear()


def gemm_raw(A, B, C, M, N, K, lda, ldb, ldc, *, bias=None, residual=None, gate=None, ldr=0, ldg=0, batch=1,
             batch_inner=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), alpha=1.0, flags=0, k_split=1, in_dtype=None,
             out_dtype=None):
    """Direct pero_gemm call; A/B/C may be tensors (pointer taken at storage offset) or raw ints."""
    _req_cuda(A, B, C)
    idt = dt(A) if in_dtype is None else in_dtype
    odt = dt(C) if out_dtype is None else out_dtype
    ws = None
    if flags & GEMM_ATOMIC:
        need = _lib.lib().pero_gemm_workspace_bytes(M, N, K, batch, int(flags), int(k_split), idt, odt)
        ws = gemm_workspace(need, A.device if isinstance(A, torch.Tensor) else torch.device("cuda", torch.cuda.current_device()))
    if gemm_timeline is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("pero_gemm", ptr(A), ptr(B), ptr(C), ptr(bias), ptr(residual), ptr(gate), M, N, K, lda, ldb, ldc, ldr, ldg,
         batch, batch_inner, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], float(alpha), int(flags), int(k_split),
         idt, odt, ptr(ws), ws.numel() if ws is not None else 0, stream())
    if gemm_timeline is not None:
        e1.record()
        fast = idt == PERO_BF16 and M % 128 == 0 and N % 128 == 0 and K % 64 == 0 and not (flags & GEMM_FORCE_GENERIC)
        lay = ("T" if flags & GEMM_TRANS_A else "N") + ("T" if flags & GEMM_TRANS_B else "N")
        gemm_timeline.append((e0, e1, 2.0 * M * N * K * batch, ("gemm_bf16_tile" if fast else "gemm_generic") + ":" + lay))


def gemm(a, b, out=None, *, bias=None, residual=None, gate=None, trans_a=False, trans_b=False, relu=False,
         alpha=1.0, out_dtype=None, atomic=False, accum=False, k_split=1, force_generic=False, extra_flags=0,
         colsum_into=None, rowdot=None, relu_bits=None, bits_tiled=False):
    """out[M,N] = alpha * op(a) @ op(b)^T ...   a: [M,K] ([K,M] if trans_a); b: [N,K] ([K,N] if trans_b).
    Row-strided 2-D views are fine (unit stride in the last dim).  colsum_into (f32 [N]): the column sums of the stored
    result are accumulated into it (PERO_GEMM_COLSUM; no input bias in that mode).  rowdot = (y, out): out (f32 [M][N/128])
    receives, per 128-column block, the row dots of the stored bf16 result with y (bf16 [M][N]) - PERO_GEMM_ROWDOT.
    relu_bits (uint8 [M][N/8]): with relu=True it RECEIVES the bit mask (stored result > 0); otherwise it is applied as the
    ReLU gate in place of a bf16 gate matrix - PERO_GEMM_RELU_BITS.  bits_tiled: the mask is laid out per 256-column block ([N/256][M][32 bytes],
    PERO_GEMM_MASK_TILED; N % 256 == 0, relu_bits contiguous) - producer and consumer must pass the same value."""
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N, Kb = (b.shape[1], b.shape[0]) if trans_b else b.shape
    assert K == Kb, (a.shape, b.shape, trans_a, trans_b)
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=out_dtype or a.dtype)
    assert out.shape == (M, N) and out.stride(1) == 1
    flags = (GEMM_RELU if relu else 0) | (GEMM_TRANS_A if trans_a else 0) | (GEMM_TRANS_B if trans_b else 0) | \
        (GEMM_ATOMIC if atomic else 0) | (GEMM_ACCUM if accum else 0) | (GEMM_FORCE_GENERIC if force_generic else 0) | extra_flags
    if colsum_into is not None:
        assert bias is None and colsum_into.dtype == torch.float32 and colsum_into.numel() == N
        bias, flags = colsum_into, flags | GEMM_COLSUM
    if relu_bits is not None:
        assert gate is None and relu_bits.dtype == torch.uint8 and relu_bits.shape == (M, N // 8) and relu_bits.stride(1) == 1
        gate, flags = relu_bits, flags | GEMM_RELU_BITS
        if bits_tiled:
            assert N % 256 == 0 and relu_bits.is_contiguous()
            flags |= GEMM_MASK_TILED
    if rowdot is not None:
        y, dots = rowdot
        assert bias is None and gate is None and N % 128 == 0 and y.shape == (M, N) and y.stride(1) == 1
        assert dots.dtype == torch.float32 and dots.numel() == M * (N // 128) and dots.is_contiguous()
        bias, gate, flags = dots, y, flags | GEMM_ROWDOT
    gemm_raw(a, b, out, M, N, K, a.stride(0), b.stride(0), out.stride(0), bias=bias, residual=residual, gate=gate,
             ldr=residual.stride(0) if residual is not None else 0, ldg=gate.stride(0) if gate is not None else 0,
             alpha=alpha, flags=flags, k_split=k_split)
    return out


def layernorm_fwd(x, gamma, beta, eps, pe=None, offsets=None, S=1):
    _req_cuda(x)
    rows, d = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    call("pero_layernorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(pe), ptr(offsets), ptr(y), ptr(mean), ptr(rstd),
         rows, d, S, float(eps), dt(x), stream())
    return y, mean, rstd


def gemm_resid_layernorm_ok(a, w, residual):
    """Shapes the fused Linear + residual + LayerNorm launch takes (csrc/gemm_e.hip gemm_bf16_n512: bf16, 512 output columns)."""
    return (a.dtype == torch.bfloat16 and a.dim() == 2 and w.dim() == 2 and w.shape[0] == 512 and a.shape[0] % 128 == 0 and a.shape[1] % 64 == 0 and
            a.shape[1] >= 192 and a.shape[1] == w.shape[1] and residual is not None and residual.shape == (a.shape[0], 512) and
            a.stride(1) == 1 and w.stride(1) == 1 and residual.stride(1) == 1 and a.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and residual.stride(0) % 8 == 0)


def gemm_resid_layernorm(a, w, bias, residual, gamma, beta, eps, store_y=True):
    """y = a @ w^T + bias + residual (bf16, stored unless store_y=False: then None), t = LayerNorm(y) * gamma + beta, mean, rstd - ONE launch
    (pero_gemm_resid_layernorm)."""
    _req_cuda(a)
    M, K = a.shape
    y = torch.empty((M, 512), device=a.device, dtype=torch.bfloat16) if store_y else None
    t = torch.empty((M, 512), device=a.device, dtype=torch.bfloat16)
    mean = torch.empty(M, device=a.device, dtype=torch.float32)
    rstd = torch.empty(M, device=a.device, dtype=torch.float32)
    if gemm_timeline is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("pero_gemm_resid_layernorm", ptr(a), ptr(w), ptr(bias), ptr(residual), ptr(gamma), ptr(beta), ptr(y), ptr(t), ptr(mean), ptr(rstd),
         M, 512, K, a.stride(0), w.stride(0), y.stride(0) if y is not None else 0, residual.stride(0), t.stride(0), float(eps), stream())
    if gemm_timeline is not None:   # counted with the tile GEMMs of bench.py's roofline: the product's flops over the WHOLE launch (LayerNorm included)
        e1.record()
        gemm_timeline.append((e0, e1, 2.0 * M * 512 * K, "gemm_bf16_tile_ln_fwd:NN"))
    return y, t, mean, rstd


def gemm_resid_layernorm_bwd_ok(a, w_t, residual, t):
    """Shapes the fused input-gradient + LayerNorm-backward launch takes (csrc/gemm_e.hip gemm_bf16_n512, EP_RESID_LNB)."""
    return (a.dtype == torch.bfloat16 and a.dim() == 2 and w_t.dim() == 2 and w_t.shape[0] == 512 and a.shape[0] % 128 == 0 and a.shape[1] % 64 == 0 and
            a.shape[1] >= 192 and a.shape[1] == w_t.shape[1] and residual is not None and residual.shape == (a.shape[0], 512) and
            t is not None and t.shape == (a.shape[0], 512) and a.stride(1) == 1 and w_t.stride(1) == 1 and residual.stride(1) == 1 and t.stride(1) == 1 and
            a.stride(0) % 8 == 0 and w_t.stride(0) % 8 == 0 and residual.stride(0) % 8 == 0 and t.stride(0) % 8 == 0)


def gemm_resid_layernorm_bwd(a, w_t, residual, t, rstd, gamma, beta, dgamma, dbeta, dxsum=None):
    """dx = LayerNorm backward (from the norm's output t and rstd) of dt = a @ w_t^T + residual, dt never stored; dgamma / dbeta / dxsum are
    accumulated into - ONE launch + the small column-sum reduce (pero_gemm_resid_layernorm_bwd)."""
    _req_cuda(a)
    M, K = a.shape
    dx = torch.empty((M, 512), device=a.device, dtype=torch.bfloat16)
    work = torch.empty(3 * _lib.LN_BWD_BLOCKS * 512, device=a.device, dtype=torch.float32)
    if gemm_timeline is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("pero_gemm_resid_layernorm_bwd", ptr(a), ptr(w_t), ptr(residual), ptr(t), ptr(rstd), ptr(gamma), ptr(beta), ptr(dx), ptr(dgamma), ptr(dbeta),
         ptr(dxsum), ptr(work), M, 512, K, a.stride(0), w_t.stride(0), residual.stride(0), t.stride(0), dx.stride(0), stream())
    if gemm_timeline is not None:   # the product's flops over the WHOLE launch (LayerNorm backward and the reduce included)
        e1.record()
        gemm_timeline.append((e0, e1, 2.0 * M * 512 * K, "gemm_bf16_tile_ln_bwd:NN"))
    return dx


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, dxsum=None):
    rows, d = x.shape
    dx = torch.empty_like(x)
    work = torch.empty(3 * _lib.LN_BWD_BLOCKS * d, device=x.device, dtype=torch.float32)
    call("pero_layernorm_bwd", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dx), ptr(dgamma), ptr(dbeta),
         ptr(dxsum), ptr(work), rows, d, dt(x), stream())
    return dx


def layernorm_bwd_out(dy, t, rstd, gamma, beta, dgamma, dbeta, dxsum=None):
    """LayerNorm backward from the layer's OUTPUT t (pero_layernorm_bwd_out): the forward kept t and rstd only."""
    rows, d = t.shape
    dx = torch.empty_like(t)
    work = torch.empty(3 * _lib.LN_BWD_BLOCKS * d, device=t.device, dtype=torch.float32)
    call("pero_layernorm_bwd_out", ptr(dy), ptr(t), ptr(rstd), ptr(gamma), ptr(beta), ptr(dx), ptr(dgamma), ptr(dbeta),
         ptr(dxsum), ptr(work), rows, d, dt(t), stream())
    return dx


def bn_fwd(x, weight, bias, running_mean, running_var, eps, momentum, training, relu):
    """BatchNorm1d (+ ReLU) over the rows of x (rows, d): returns (y, save_mean, save_rstd); running statistics updated in place in training."""
    _req_cuda(x)
    rows, d = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(d, device=x.device, dtype=torch.float32)
    rstd = torch.empty(d, device=x.device, dtype=torch.float32)
    work = torch.empty(2 * d, device=x.device, dtype=torch.float32)
    call("pero_bn_fwd", ptr(x), ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), ptr(y), ptr(mean), ptr(rstd), ptr(work),
         rows, d, float(eps), float(momentum), int(bool(training)), int(bool(relu)), dt(x), stream())
    return y, mean, rstd


def bn_bwd(dy, x, y, weight, mean, rstd, dweight, dbias, relu):
    rows, d = x.shape
    dx = torch.empty_like(x)
    work = torch.empty(2 * d, device=x.device, dtype=torch.float32)
    call("pero_bn_bwd", ptr(dy), ptr(x), ptr(y), ptr(weight), ptr(mean), ptr(rstd), ptr(dx), ptr(dweight), ptr(dbias), ptr(work),
         rows, d, int(bool(relu)), dt(x), stream())
    return dx


def softmax_fwd(scores, scale, out_dtype):
    rows, cols = scores.numel() // scores.shape[-1], scores.shape[-1]
    p = torch.empty(scores.shape, device=scores.device, dtype=out_dtype)
    call("pero_softmax_fwd", ptr(scores), ptr(p), rows, cols, float(scale), dt(out_dtype), stream())
    return p


def softmax_bwd(p, dp, scale):
    rows, cols = p.numel() // p.shape[-1], p.shape[-1]
    ds = torch.empty_like(p)
    call("pero_softmax_bwd", ptr(p), ptr(dp), ptr(ds), rows, cols, float(scale), dt(p), stream())
    return ds


def attention_fused_ok(qkv, s, h):
    d = qkv.shape[1] // 3
    return qkv.dtype == torch.bfloat16 and d // h == 128 and d % h == 0 and s % 128 == 0


def attention_fwd_fused(qkv, n, s, h):
    d = qkv.shape[1] // 3
    out = torch.empty((n * s, d), device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty((n * h, s), device=qkv.device, dtype=torch.float32)
    call("pero_attention_fwd", ptr(qkv), ptr(out), ptr(lse), n, s, h, d // h, dt(qkv), stream())
    return out, lse


def attention_bwd_fused(qkv, out, dout, lse, n, s, h, dbias=None, dvec=None):
    """dbias (f32 [3d], optional): in_proj's bias gradient (column sums of dqkv) is accumulated into it by the kernels.
    dvec (f32 [n*s][h], optional): D = per-head row sums of dout * out, already computed (then `out` is not read)."""
    d = qkv.shape[1] // 3
    dqkv = torch.empty_like(qkv)
    if dvec is None:
        dvec = torch.empty((n * s, h), device=qkv.device, dtype=torch.float32)
    else:
        out = None
    work = torch.empty(3 * n * h * (s // 128) * 128, device=qkv.device, dtype=torch.float32) if dbias is not None else None
    call("pero_attention_bwd", ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dvec), ptr(dqkv), ptr(dbias), ptr(work), n, s, h, d // h,
         dt(qkv), stream())
    return dqkv


def masked_ce_fwd(logits, labels, mask, unmasked_weight=None):
    """logits (rows,V); labels/mask int64 (rows).  Returns (loss: f32 tensor of 1 element, work buffer)."""
    _req_cuda(logits, labels, mask)
    rows, V = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    work = torch.empty(2 * rows + 8 + 256, device=logits.device, dtype=torch.float32)  # PERO_CE_WORK(rows)
    call("pero_masked_ce_fwd", ptr(logits), ptr(labels), ptr(mask), -1.0 if unmasked_weight is None else float(unmasked_weight),
         ptr(loss), ptr(work), rows, V, dt(logits), stream())
    return loss, work


def masked_ce_fwd_rows(logits, labels, mask, index):
    """masked_ce_fwd with unmasked_weight None for a caller that lists the rows with mask == 1 (int64 device tensor): same bits."""
    rows, V = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    work = torch.empty(2 * rows + 8 + 256, device=logits.device, dtype=torch.float32)
    call("pero_masked_ce_fwd_rows", ptr(logits), ptr(labels), ptr(mask), ptr(index), index.numel(), ptr(loss), ptr(work), rows, V,
         dt(logits), stream())
    return loss, work


def zeros(shape, device, dtype):
    """A zero tensor filled by the library's linear fill kernel (torch's elementwise fill runs at a third of its rate)."""
    t = torch.empty(shape, device=device, dtype=dtype)
    if t.numel():
        call("pero_zero_fill", ptr(t), t.numel() * t.element_size(), stream())
    return t


def masked_ce_bwd(logits, labels, mask, work, unmasked_weight=None, dloss=None):
    rows, V = logits.shape
    dlogits = torch.empty_like(logits)
    call("pero_masked_ce_bwd", ptr(logits), ptr(labels), ptr(mask), -1.0 if unmasked_weight is None else float(unmasked_weight),
         ptr(dloss), ptr(work), ptr(dlogits), rows, V, dt(logits), stream())
    return dlogits


def masked_ce_bwd_rows(logits, labels, mask, work, index, n_rows_out, unmasked_weight=None, dloss=None):
    """Compact CE gradient: (n_rows_out, V), row i = gradient row of logits row index[i], zero rows behind index.numel()."""
    rows, V = logits.shape
    out = torch.empty((n_rows_out, V), device=logits.device, dtype=logits.dtype)
    call("pero_masked_ce_bwd_rows", ptr(logits), ptr(labels), ptr(mask), -1.0 if unmasked_weight is None else float(unmasked_weight),
         ptr(dloss), ptr(work), ptr(index), index.numel(), n_rows_out, ptr(out), rows, V, dt(logits), stream())
    return out


def host_mask(m):
    """The mask's values on the HOST when they are known there without a device sync: numpy arrays, CPU tensors, and device
    tensors that a batch operator / collator / prefetcher uploaded from host arrays (they carry a private copy of the original as
    `_pero_host`; such a device mask must not be modified in place after the upload).  None: the mask lives on the device only."""
    if isinstance(m, np.ndarray):
        return m
    if isinstance(m, torch.Tensor):
        if not m.is_cuda:
            return m.numpy()
        return getattr(m, "_pero_host", None)
    return np.asarray(m)


def colsum(x, out):
    rows, cols = x.shape
    call("pero_colsum", ptr(x), ptr(out), rows, cols, x.stride(0), dt(x), stream())
    return out


def transpose_table(entries, device):
    """entries: [(src offset, dst offset, rows, cols)] -> (device table for transpose_multi, total 64x64 tiles)."""
    rows_, tiles = [], 0
    for so, do, r, c in entries:
        rows_.append([so, do, r, c, tiles])
        tiles += ((r + 63) // 64) * ((c + 63) // 64)
    return torch.tensor(rows_, dtype=torch.int64).to(device), tiles


def transpose_multi(src, dst, table, tiles):
    """dst matrix t = src matrix t transposed, for every row of `table` (2-byte elements, one launch)."""
    assert src.element_size() == 2 and dst.element_size() == 2 and table.dtype == torch.int64 and table.is_cuda
    call("pero_transpose_multi", ptr(src), ptr(dst), ptr(table), table.shape[0], tiles, stream())
    return dst


def cast_to_bf16(src, dst):
    call("pero_cast_f32_bf16", ptr(src), ptr(dst), src.numel(), stream())
    return dst


def scale_(x, s):
    call("pero_scale", ptr(x), x.numel(), float(s), dt(x), stream())
    return x


def patches_from_u8(images, mask, tile, P, dtype, pitch=None):
    """(N*S, pitch) buffer whose first C*H*P columns are the patch rows (rest zero)."""
    _req_cuda(images)
    images = images.contiguous()
    N, H, W, C = images.shape
    pitch = pitch or C * H * P
    out = torch.empty((N * (W // P), pitch), device=images.device, dtype=dtype)
    call("pero_patches_from_u8", ptr(images), ptr(mask), ptr(tile), ptr(out), N, H, W, C, P, pitch, dt(dtype), stream())
    return out


def patches_from_f32(images, mask, tile, P, dtype, pitch=None):
    _req_cuda(images)
    images = images.contiguous()
    N, C, H, W = images.shape
    pitch = pitch or C * H * P
    out = torch.empty((N * (W // P), pitch), device=images.device, dtype=dtype)
    call("pero_patches_from_f32", ptr(images), ptr(mask), ptr(tile), ptr(out), N, H, W, C, P, pitch, dt(dtype), stream())
    return out


def cast_pad_to_bf16(src2d, pitch):
    rows, cols = src2d.shape
    dst = torch.empty((rows, pitch), device=src2d.device, dtype=torch.bfloat16)
    call("pero_cast_pad_f32_bf16", ptr(src2d), ptr(dst), rows, cols, pitch, stream())
    return dst


def add_rows2d(dst2d, src2d, cols):
    call("pero_add_rows2d", ptr(dst2d), ptr(src2d), dst2d.shape[0], cols, dst2d.stride(0), src2d.stride(0), stream())
    return dst2d


def apply_mask_(images, mask, tile, P):
    _req_cuda(images)
    N, C, H, W = images.shape
    call("pero_apply_mask_f32", ptr(images), ptr(mask), ptr(tile), N, H, W, C, P, stream())
    return images


def adam_step(p, g, m, v, p_bf16, lr, beta1, beta2, eps, step, grad_scale=1.0):
    call("pero_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_bf16), p.numel(), float(lr), float(beta1), float(beta2),
         float(eps), int(step), float(grad_scale), stream())


def vq_argmin(x, codebook, want_dist=False):
    _req_cuda(x, codebook)
    M, D = x.shape
    K = codebook.shape[0]
    idx = torch.empty(M, device=x.device, dtype=torch.int64)
    best = torch.empty(M, device=x.device, dtype=torch.float32) if want_dist else None
    work = torch.empty(M + K, device=x.device, dtype=torch.float32)
    call("pero_vq_argmin", ptr(x), ptr(codebook), ptr(idx), ptr(best), ptr(work), M, K, D, stream())
    return (idx, best) if want_dist else idx


def vq_gather(x, codebook, idx):
    q = torch.empty_like(x)
    call("pero_vq_gather", ptr(x), ptr(codebook), ptr(idx), ptr(q), x.shape[0], x.shape[1], stream())
    return q


def vq_ema_update(x, idx, ema_cluster_size, ema_w, codebook, decay, epsilon):
    """In-place EMA codebook update (models/autoencoders.py:225-237); all tensors f32 contiguous on the device."""
    M, D = x.shape
    K = codebook.shape[0]
    work = torch.empty(K + K * D, device=x.device, dtype=torch.float32)
    call("pero_vq_ema_update", ptr(x), ptr(idx), ptr(ema_cluster_size), ptr(ema_w), ptr(codebook), ptr(work), M, K, D,
         float(decay), float(epsilon), stream())


def gather_rows(src, index, n_rows_out=None, out=None):
    n_idx = index.numel()
    n_out = n_idx if n_rows_out is None else n_rows_out
    d = src.shape[-1]
    dst = torch.empty((n_out, d), device=src.device, dtype=src.dtype) if out is None else out
    assert dst.shape == (n_out, d) and dst.is_contiguous()
    call("pero_gather_rows", ptr(src), ptr(index), ptr(dst), n_idx, n_out, d, dt(src), stream())
    return dst


def scatter_add_rows(src, index, dst):
    if index.numel():
        call("pero_scatter_add_rows", ptr(src), ptr(index), ptr(dst), index.numel(), dst.shape[-1], dt(src), stream())
    return dst


# ---- joint-embedding loss reductions ----------------------------------------------------------------------
def sqdiff_rows(x, ix, y, iy, scale):
    n, d = ix.numel(), x.shape[-1]
    partial = torch.empty(n, device=x.device, dtype=torch.float32)
    out = torch.empty(1, device=x.device, dtype=torch.float32)
    call("pero_sqdiff_rows", ptr(x), ptr(ix), ptr(y), ptr(iy), ptr(partial), ptr(out), n, d, float(scale), dt(x), stream())
    return out


def sqdiff_rows_bwd(x, ix, y, iy, dx, dy, g, coef):
    call("pero_sqdiff_rows_bwd", ptr(x), ptr(ix), ptr(y), ptr(iy), ptr(dx), ptr(dy), ptr(g), float(coef), ix.numel(),
         x.shape[-1], dt(x), stream())


def center_cols(z, colsum_, m):
    m_pad, d = z.shape
    zc = torch.empty_like(z)
    sumsq = zeros((d,), z.device, torch.float32)
    call("pero_center_cols", ptr(z), ptr(colsum_), ptr(zc), ptr(sumsq), m, m_pad, d, dt(z), stream())
    return zc, sumsq


def vicreg_var(sumsq, m, threshold, eps):
    d = sumsq.numel()
    cvar = torch.empty(d, device=sumsq.device, dtype=torch.float32)
    loss = torch.empty(1, device=sumsq.device, dtype=torch.float32)
    call("pero_vicreg_var", ptr(sumsq), ptr(cvar), ptr(loss), m, d, float(threshold), float(eps), stream())
    return cvar, loss


def vicreg_cov(cov, cvar, m, wv, wc, dtype):
    d = cov.shape[0]
    G = torch.empty((d, d), device=cov.device, dtype=dtype)
    rowpart = torch.empty(d, device=cov.device, dtype=torch.float32)
    loss = torch.empty(1, device=cov.device, dtype=torch.float32)
    call("pero_vicreg_cov", ptr(cov), ptr(cvar), ptr(G), ptr(rowpart), ptr(loss), d, m, float(wv), float(wc), dt(dtype), stream())
    return G, loss


def scatter_add_rows_scaled(src, index, dst, g):
    if index.numel():
        call("pero_scatter_add_rows_scaled", ptr(src), ptr(index), ptr(dst), ptr(g), index.numel(), dst.shape[-1], dt(src), stream())
    return dst


def rownorm_fwd(x, out=None):
    rows, d = x.shape
    if out is not None:
        xn, inv = out
        assert xn.shape == x.shape and xn.is_contiguous() and inv.numel() == rows and inv.is_contiguous()
    else:
        xn = torch.empty_like(x)
        inv = torch.empty(rows, device=x.device, dtype=torch.float32)
    call("pero_rownorm_fwd", ptr(x), ptr(xn), ptr(inv), rows, d, dt(x), stream())
    return xn, inv


def rownorm_bwd(xn, dxn, inv, g=None):
    dx = torch.empty_like(xn)
    call("pero_rownorm_bwd", ptr(xn), ptr(dxn), ptr(inv), ptr(g), ptr(dx), xn.shape[0], xn.shape[1], dt(xn), stream())
    return dx


def ntxent_cols(sim, want_grad_dtype=None):
    lines, S, _ = sim.shape
    line_loss = torch.empty(lines, device=sim.device, dtype=torch.float32)
    loss = torch.empty(1, device=sim.device, dtype=torch.float32)
    dsim = torch.empty(sim.shape, device=sim.device, dtype=want_grad_dtype) if want_grad_dtype is not None else None
    call("pero_ntxent_cols", ptr(sim), ptr(line_loss), ptr(loss), ptr(dsim), lines, S,
         dt(want_grad_dtype) if want_grad_dtype is not None else PERO_F32, stream())
    return loss, line_loss, dsim


def ntxent_cols_cross(sim, cross, own0, grad_dtype):
    """sim (lines, S, S) f32, cross (lines*S, L) f32 -> (loss [1], line_loss, dsim, dcross) - see pero_ntxent_cols_cross."""
    lines, S, _ = sim.shape
    L = cross.shape[1]
    line_loss = torch.empty(lines, device=sim.device, dtype=torch.float32)
    loss = torch.empty(1, device=sim.device, dtype=torch.float32)
    dsim = torch.empty(sim.shape, device=sim.device, dtype=grad_dtype)
    dcross = torch.empty(cross.shape, device=sim.device, dtype=grad_dtype)
    call("pero_ntxent_cols_cross", ptr(sim), ptr(cross), ptr(line_loss), ptr(loss), ptr(dsim), ptr(dcross), lines, S, L, int(own0),
         dt(grad_dtype), stream())
    return loss, line_loss, dsim, dcross


def line_mean(x2, lines, S):
    """x2 (lines*S, d) -> f32 (lines, d): mean over the rows of each line."""
    out = torch.empty((lines, x2.shape[1]), device=x2.device, dtype=torch.float32)
    call("pero_line_mean", ptr(x2), ptr(out), lines, S, x2.shape[1], dt(x2), stream())
    return out


def add_line_rows_(dst2, src, lines, S, scale):
    """dst2 (lines*S, d) += scale * src (lines, d) f32, row-broadcast per line; in place."""
    call("pero_add_line_rows", ptr(dst2), ptr(src), lines, S, dst2.shape[1], float(scale), dt(dst2), stream())
    return dst2


MAX_TOPK = 8


def label_rank(logits, labels, mask, ks=None, counters=None, want_ranks=False):
    """Rank of each masked row's label inside its logit row + accumulated top-k error counters
    (masked_pretraining/tester.py:72-113 on the device).  logits (rows, V) f32/bf16 (row pitch = stride(0));
    labels, mask (rows,) int64; ks: int32 device tensor of measured errors; counters: int64 device tensor
    [1 + len(ks)] that is ACCUMULATED into.  Returns (counters, ranks or None)."""
    rows, V = logits.shape
    assert logits.stride(1) == 1 and labels.numel() == rows and mask.numel() == rows
    nk = 0 if ks is None else int(ks.numel())
    ranks = torch.empty((rows, 3), device=logits.device, dtype=torch.int32) if want_ranks else None
    call("pero_label_rank", ptr(logits), logits.stride(0), ptr(labels), ptr(mask), rows, V, ptr(ks), nk, ptr(counters),
         ptr(ranks), dt(logits.dtype), stream())
    return counters, ranks


def stack_lines(packed, offsets, widths, left_px, B, H, Wt, C):
    """Ragged uint8 lines (device, back to back, readable 8 bytes past the end) -> zero-padded (B, H, Wt, C) uint8 batch
    (common/dataloader.py:80-100).  offsets int64, widths / left_px int32 device tensors."""
    out = torch.empty((B, H, Wt, C), device=packed.device, dtype=torch.uint8)
    call("pero_stack_lines", ptr(packed), ptr(offsets), ptr(widths), ptr(left_px), ptr(out), B, H, Wt, C, stream())
    return out


def line_masks(widths1, left1, S, subsampling, widths2=None, left2=None, crop_shifts=None):
    """Image masks, shifts and three-valued shift masks (common/dataloader.py:92-96, 124-138) from the per-line widths (px)
    and left paddings (label positions): int32 device tensors.  Returns (im1, im2, sm1, sm2, shifts); the last four are
    None for an unpaired batch."""
    B = widths1.numel()
    dev = widths1.device
    im1 = torch.empty((B, S), device=dev, dtype=torch.uint8)
    im2 = sm1 = sm2 = shifts = None
    if widths2 is not None:
        im2, sm1, sm2 = (torch.empty((B, S), device=dev, dtype=torch.uint8) for _ in range(3))
        shifts = torch.empty(B, device=dev, dtype=torch.int32)
    call("pero_line_masks", ptr(widths1), ptr(widths2), ptr(left1), ptr(left2), ptr(crop_shifts), ptr(im1), ptr(im2), ptr(sm1),
         ptr(sm2), ptr(shifts), B, S, subsampling, stream())
    return im1, im2, sm1, sm2, shifts
