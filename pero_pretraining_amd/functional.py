"""Forward / backward of the transformer hot path, sequenced on the host as HIP kernel launches.

Token layout: every activation is a row-major (N*S, d) matrix of token rows (line-major), which is
what the reference reaches after its `n d s -> s n d` / `s n d -> n d s` transposes
(models/transformers.py:82-89) - those are pure layout and vanish here.

Attention: bf16 with head_dim 128 and S % 128 == 0 runs the fused flash-style kernels of csrc/attention.hip on the
packed qkv tensor (scores never stored; one f32 log-sum-exp per query saved for the backward); f32 parity mode and other
shapes take the unfused form of torch SDPA - per (line, head) batched GEMMs + a row softmax kernel, scores in f32.
Every product of the backward pass is pero_gemm with a different operand-layout flag: input gradients as K-contiguous
products on transposed bf16 weight copies, weight gradients as split-K products into the parameters' f32 `.grad`
buffers - partial tiles summed in slice order through a caller-owned workspace (ops.gemm_workspace; run-to-run
reproducible), f32 atomics only for short reductions.
"""
import math

import torch

from . import lowp, ops
from ._lib import GEMM_ATOMIC, GEMM_TRANS_A, GEMM_TRANS_B

LN_EPS = 1e-5


def _ksplit(rows, out_elems):
    """Split the token (reduction) dimension of a weight-gradient GEMM so that the grid fills 256 CUs."""
    tiles = max(1, out_elems // (128 * 128))
    want = max(1, 512 // tiles)
    return int(max(1, min(want, rows // 512 if rows >= 512 else 1)))


def ensure_grad(p):
    """Parameter gradient buffer the kernels accumulate into (zero-filled when freshly created)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


FWD_TILE_FLAGS = 0
FUSE_B1_COLSUM = True    # linear1's bias gradient from the epilogue of linear2's input-gradient product (else: a column-sum pass on the side stream)
DX_ON_WT = True          # input gradients on transposed weight copies (lowp.weight_t)
DX_TILE_FLAGS = 0
RELU_GATE_BITS = True    # linear1's ReLU leaves a bit mask (free in its epilogue); linear2's input gradient reads a byte per 8 columns, all of a
                         # tile's bytes ahead of the staging barriers, instead of eight dependent 16-byte gate rows: step -0.7 %


RELU_BITS_TILED = True   # the mask per 256-column block ([N/256][M][32 B]) when N % 256 == 0: a tile's mask is whole cache lines, and linear1's product may walk its
                         # tiles like the other K = 512 products (ops.gemm bits_tiled; csrc/gemm_e.hip pero_launch_gemm_e256)


def bits_tiled(n):
    return RELU_BITS_TILED and n % 256 == 0


def relu_bits_ok(m, n, k, dtype):
    """Shapes for which the ReLU of a Linear can leave its gate as a bit mask (PERO_GEMM_RELU_BITS: the 256-row tile kernels)."""
    return RELU_GATE_BITS and dtype == torch.bfloat16 and m % 256 == 0 and n % 128 == 0 and k % 32 == 0


def linear_fwd(x, w, b, dtype, residual=None, relu=False, out_dtype=None, relu_bits=None):
    return ops.gemm(x, lowp.weight(w, dtype), relu_bits=relu_bits, bias=None if b is None else b.detach(), residual=residual, relu=relu,
                    out_dtype=out_dtype, extra_flags=FWD_TILE_FLAGS, bits_tiled=relu_bits is not None and bits_tiled(w.shape[0]))


FUSE_ROWDOT = True     # attention backward's D = rowsum(dO * O) from the epilogue of the out-projection's dX product
FUSE_BIAS_GRAD = True  # bias gradients that are column sums of a dX product's output come out of that product's epilogue
SIDE_STREAM_DW = False  # weight / bias gradients on a second HIP stream (they are off the backward critical path).  Off since round 3: every
                        # product is a whole-chip kernel now (persistent dX tiles, 256 split-K work items), two of them side by side only
                        # share the CUs - B = 1536 step 115.6 ms with, 114.6 ms without (same box, interleaved); rounds 1-2 gained 0.5-1 ms
SIDE_STREAMS = 2       # side streams used round-robin (the library's split-K aims at 512 work items per product)
_side_streams = {}


class SideStream:
    """Runs the weight-gradient products of the backward pass on a second HIP stream: nothing downstream in the
    backward chain reads them, so they can fill the tails / stalls of the dX chain's kernels.  Inputs are protected
    from the caching allocator with record_stream; `join()` makes the main stream wait at the end."""

    def __init__(self, device):
        self.enabled = SIDE_STREAM_DW
        self.keep = []
        self.main = torch.cuda.current_stream(device)
        if self.enabled:
            self.sides = []
            for i in range(SIDE_STREAMS):
                key = (device.index, self.main.cuda_stream, i)
                if key not in _side_streams:
                    _side_streams[key] = torch.cuda.Stream(device=device)
                self.sides.append(_side_streams[key])
                self.sides[-1].wait_stream(self.main)  # the gradient buffers were zeroed / touched on the main stream
            self.side = self.sides[0]
            self.rr = 0

    def run(self, fn, *tensors):
        """fn() launches kernels that read `tensors` (already produced on the main stream).  The tensors are kept
        alive until join() so that the caching allocator cannot hand their memory to a later main-stream allocation
        while the side stream still reads it (this also holds under hipGraph capture, where record_stream does not)."""
        if not self.enabled:
            fn()
            return
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side = self.sides[self.rr % len(self.sides)]
        self.rr += 1
        self.side.wait_event(ev)
        self.keep.extend(tensors)
        with torch.cuda.stream(self.side):
            fn()

    def comm_context(self):
        """Stream context for work that must see every gradient enqueued so far (the data-parallel bucket hooks): a
        dedicated stream that waits for the main stream and all side streams.  The side streams themselves are NOT joined -
        making side stream 0 wait for the others once per layer cost ~0.9 ms per step (tools/dp_ab.py)."""
        key = (self.main.device.index, self.main.cuda_stream, "comm")
        if key not in _side_streams:
            _side_streams[key] = torch.cuda.Stream(device=self.main.device)
        c = _side_streams[key]
        c.wait_stream(self.main)
        if self.enabled:
            for st in self.sides:
                c.wait_stream(st)
        return torch.cuda.stream(c)

    def join(self):
        if self.enabled:
            for st in self.sides:
                self.main.wait_stream(st)
        self.keep = []


def linear_bwd(dy, x, w, b, dtype, need_dx=True, gate=None, residual=None, bias_grad_done=False, side=None, dx_colsum_into=None,
               dx_rowdot=None, gate_bits=None):
    """dy (M,N), x (M,K), w (N,K).  Accumulates dW (+db) into .grad; returns dx = dy @ W (+residual)(*gate>0).
    dx_colsum_into: f32 [K] that receives the column sums of dx (the bias gradient of the Linear that produced x), fused
    into the dX product's epilogue.  dx_rowdot = (y, out): per-128-column-block row dots of dx with y (the attention backward's D
    when dx is the attention output's gradient and y the attention output), same epilogue."""
    def grads():
        if w.requires_grad:
            ops.gemm(dy, x, out=ensure_grad(w).view(w.shape[0], -1), trans_a=True, trans_b=True, atomic=True, k_split=0)
        if b is not None and b.requires_grad and not bias_grad_done:
            ops.colsum(dy, ensure_grad(b))
    if side is not None:
        ensure_grad(w)
        if b is not None:
            ensure_grad(b)
        side.run(grads, dy, x)
    else:
        grads()
    if not need_dx:
        return None
    if DX_ON_WT and dtype == torch.bfloat16:
        # dX = dY (W^T)^T on a transposed bf16 weight copy: both operands K-contiguous, 256x256x64 tiles
        return ops.gemm(dy, lowp.weight_t(w), residual=residual, gate=None if gate_bits is not None else gate,
                        relu_bits=gate_bits, colsum_into=dx_colsum_into, rowdot=dx_rowdot, extra_flags=DX_TILE_FLAGS,
                        bits_tiled=gate_bits is not None and bits_tiled(w.shape[1]))
    if gate_bits is not None:
        raise RuntimeError("linear_bwd: a bit-mask gate needs the bf16 transposed-weight path")
    return ops.gemm(dy, lowp.weight(w, dtype).view(w.shape[0], -1), trans_b=True, residual=residual, gate=gate,
                    colsum_into=dx_colsum_into, rowdot=dx_rowdot)


# ---------------------------------------------------------------------------------------------
# attention on the packed qkv tensor
# ---------------------------------------------------------------------------------------------
def attention_fwd(qkv, n, s, h, need_p=True):
    d = qkv.shape[1] // 3
    hd = d // h
    scores = torch.empty((n * h, s, s), device=qkv.device, dtype=torch.float32)
    ops.gemm_raw(qkv, qkv[:, d:], scores, s, s, hd, 3 * d, 3 * d, s, batch=n * h, batch_inner=h,
                 sA=(s * 3 * d, hd), sB=(s * 3 * d, hd), sC=(h * s * s, s * s))
    p = ops.softmax_fwd(scores, 1.0 / math.sqrt(hd), qkv.dtype)
    out = torch.empty((n * s, d), device=qkv.device, dtype=qkv.dtype)
    ops.gemm_raw(p, qkv[:, 2 * d:], out, s, hd, s, s, 3 * d, d, batch=n * h, batch_inner=h,
                 sA=(h * s * s, s * s), sB=(s * 3 * d, hd), sC=(s * d, hd), flags=GEMM_TRANS_B)
    return out, p


def attention_bwd(qkv, p, dout, n, s, h):
    d = qkv.shape[1] // 3
    hd = d // h
    dqkv = torch.empty_like(qkv)
    bq = dict(batch=n * h, batch_inner=h)
    sP, sQ, sO = (h * s * s, s * s), (s * 3 * d, hd), (s * d, hd)
    # dV = P^T dO
    ops.gemm_raw(p, dout, dqkv[:, 2 * d:], s, hd, s, s, d, 3 * d, sA=sP, sB=sO, sC=sQ,
                 flags=GEMM_TRANS_A | GEMM_TRANS_B, **bq)
    # dP = dO V^T  (f32)
    dp = torch.empty((n * h, s, s), device=qkv.device, dtype=torch.float32)
    ops.gemm_raw(dout, qkv[:, 2 * d:], dp, s, s, hd, d, 3 * d, s, sA=sO, sB=sQ, sC=sP, **bq)
    ds = ops.softmax_bwd(p, dp, 1.0 / math.sqrt(hd))
    # dQ = dS K ; dK = dS^T Q
    ops.gemm_raw(ds, qkv[:, d:], dqkv, s, hd, s, s, 3 * d, 3 * d, sA=sP, sB=sQ, sC=sQ, flags=GEMM_TRANS_B, **bq)
    ops.gemm_raw(ds, qkv, dqkv[:, d:], s, hd, s, s, 3 * d, 3 * d, sA=sP, sB=sQ, sC=sQ,
                 flags=GEMM_TRANS_A | GEMM_TRANS_B, **bq)
    return dqkv


# ---------------------------------------------------------------------------------------------
# one post-norm encoder layer (torch.nn.TransformerEncoderLayer semantics, models/transformers.py:36-43)
# ---------------------------------------------------------------------------------------------
FUSED_ATTENTION = True  # bf16, head_dim 128, S % 128 == 0: flash-style HIP kernels; else batched GEMM + softmax


FUSE_LN_FWD_MAX_K = 4096  # Linear + residual + LayerNorm as ONE launch (csrc/gemm_e.hip gemm_bf16_n512, pero_gemm_resid_layernorm) for reductions up to this
                          # length: at K = 512 (out-projection) the fused launch takes 509 us against 352 + 195 for the pair (524 288 rows), at
                          # K = 2048 (linear2) 1 133 against 931 + 190 - the row-complete tile runs 5 % behind the 256 x 256 one there; in the
                          # 2048-line step: never 147.7 ms, out-projection only 147.0, both 146.8.  0: never


LN_BWD_FROM_OUT = True    # bf16 mode: the encoder layers' LayerNorms keep their OUTPUT t (the next Linear's input, saved anyway) and rstd for the backward,
                          # not their input rows y ("memory-efficient" LayerNorm: xhat = (t - beta) / gamma, pero_layernorm_bwd_out).  The fused
                          # Linear + residual + LayerNorm launch then does not store y at all: 24 x 537 MB per 2048-line step less written and kept.
                          # f32 parity mode always keeps y (the reference's arithmetic: torch.nn.LayerNorm saves its input)


def ln_from_out(dtype):
    return LN_BWD_FROM_OUT and dtype == torch.bfloat16


def linear_resid_ln_fwd(x, lin_w, lin_b, resid, norm, dtype, keep_y=True):
    """y = x @ W^T + b + resid ; t, mean, rstd = LayerNorm(y): fused where the shape and the reduction length allow, else the pair.
    keep_y=False: y is not needed afterwards (returned as None; the fused launch does not even store it)."""
    if (dtype == torch.bfloat16 and FUSE_LN_FWD_MAX_K and x.shape[1] <= FUSE_LN_FWD_MAX_K and lin_b is not None and
            ops.gemm_resid_layernorm_ok(x, lowp.weight(lin_w, dtype), resid)):
        return ops.gemm_resid_layernorm(x, lowp.weight(lin_w, dtype), lin_b.detach(), resid, norm.weight.detach(), norm.bias.detach(), norm.eps,
                                        store_y=keep_y)
    y = linear_fwd(x, lin_w, lin_b, dtype, residual=resid)
    t, mean, rstd = ops.layernorm_fwd(y, norm.weight.detach(), norm.bias.detach(), norm.eps)
    return (y if keep_y else None), t, mean, rstd


FUSE_LN_BWD = True    # bf16 mode with LN_BWD_FROM_OUT: the LayerNorm backward of norm1 / norm2 runs in the epilogue of the input-gradient product that
                      # produces its upstream gradient (linear1's dX -> norm1, in_proj's dX -> the PREVIOUS layer's norm2) - one launch on the
                      # row-complete tile (pero_gemm_resid_layernorm_bwd), dt never stored


def linear_bwd_ln(dy, x, w, b, dtype, residual, t_ln, rstd_ln, norm, dxsum, side=None, bias_grad_done=False):
    """linear_bwd(dy, x, w, b, residual=residual) followed by the LayerNorm backward (from its output t_ln and rstd_ln) of `norm`, whose output
    x is: returns dy_norm.  Fused into one launch where the shape allows; None if it does not (the caller runs the pair)."""
    if not (FUSE_LN_BWD and DX_ON_WT and dtype == torch.bfloat16 and ops.gemm_resid_layernorm_bwd_ok(dy, lowp.weight_t(w), residual, t_ln)):
        return None
    linear_bwd(dy, x, w, b, dtype, need_dx=False, side=side, bias_grad_done=bias_grad_done)   # weight (and bias) gradients
    return ops.gemm_resid_layernorm_bwd(dy, lowp.weight_t(w), residual, t_ln, rstd_ln, norm.weight.detach(), norm.bias.detach(),
                                        ensure_grad(norm.weight), ensure_grad(norm.bias), dxsum)


def ln_bwd(dt, y, t, mean, rstd, norm, dxsum):
    """LayerNorm backward of an encoder layer's norm: from the saved input rows y, or - when the forward did not keep them - from its output t."""
    if y is None:
        return ops.layernorm_bwd_out(dt, t, rstd, norm.weight.detach(), norm.bias.detach(), ensure_grad(norm.weight), ensure_grad(norm.bias), dxsum)
    return ops.layernorm_bwd(dt, y, mean, rstd, norm.weight.detach(), ensure_grad(norm.weight), ensure_grad(norm.bias), dxsum)


def layer_fwd(t, L, n, s, h, dtype, save):
    at = L.self_attn
    qkv = linear_fwd(t, at.in_proj_weight, at.in_proj_bias, dtype)
    if FUSED_ATTENTION and ops.attention_fused_ok(qkv, s, h):
        a, p = ops.attention_fwd_fused(qkv, n, s, h)   # p = base-2 log-sum-exp rows (N*h, S)
    else:
        a, p = attention_fwd(qkv, n, s, h)             # p = probabilities (N*h, S, S)
    keep_y = not (save and ln_from_out(dtype))
    y1, t1, mean1, rstd1 = linear_resid_ln_fwd(a, at.out_proj.weight, at.out_proj.bias, t, L.norm1, dtype, keep_y=keep_y)
    bits = None
    if save and DX_ON_WT and relu_bits_ok(t1.shape[0], L.linear1.weight.shape[0], t1.shape[1], dtype) and \
            relu_bits_ok(t1.shape[0], L.linear1.weight.shape[0], L.linear2.weight.shape[0], dtype):
        bits = torch.empty((t1.shape[0], L.linear1.weight.shape[0] // 8), device=t1.device, dtype=torch.uint8)
    hdn = linear_fwd(t1, L.linear1.weight, L.linear1.bias, dtype, relu=True, relu_bits=bits)
    y2, t2, mean2, rstd2 = linear_resid_ln_fwd(hdn, L.linear2.weight, L.linear2.bias, t1, L.norm2, dtype, keep_y=keep_y)
    saved = (t, qkv, p, a, y1, mean1, rstd1, t1, hdn, y2, mean2, rstd2, bits, t2) if save else None
    return t2, saved


def layer_bwd(dt2, L, saved, n, s, h, dtype, side=None, dt2_is_dy2=False, prev=None):
    """dt2: gradient of the layer's output - or, with dt2_is_dy2, already the gradient behind norm2's backward (the layer above ran it in the
    epilogue of its last product).  prev = (rstd2, layer) of the layer BELOW: its norm2's backward is then fused into this layer's last product
    where the shape allows.  Returns (gradient for the layer below, whether that is already behind the lower layer's norm2 backward)."""
    t, qkv, p, a, y1, mean1, rstd1, t1, hdn, y2, mean2, rstd2, bits, t2 = saved
    at = L.self_attn
    # LN2 (its dx column sums are linear2's bias gradient)
    dy2 = dt2 if dt2_is_dy2 else ln_bwd(dt2, y2, t2, mean2, rstd2, L.norm2, ensure_grad(L.linear2.bias))
    # linear1's bias gradient = column sums of dpre1: accumulated by the epilogue of the product that writes dpre1
    fuse_b1 = FUSE_BIAS_GRAD and FUSE_B1_COLSUM and dtype == torch.bfloat16 and L.linear1.bias is not None and L.linear1.bias.requires_grad
    dpre1 = linear_bwd(dy2, hdn, L.linear2.weight, L.linear2.bias, dtype, gate=hdn, gate_bits=bits, bias_grad_done=True, side=side,
                       dx_colsum_into=ensure_grad(L.linear1.bias) if fuse_b1 else None)
    dy1 = None
    if y1 is None:   # norm1's backward in the epilogue of linear1's input-gradient product
        dy1 = linear_bwd_ln(dpre1, t1, L.linear1.weight, L.linear1.bias, dtype, dy2, t1, rstd1, L.norm1, ensure_grad(at.out_proj.bias),
                            side=side, bias_grad_done=fuse_b1)
    if dy1 is None:
        dt1 = linear_bwd(dpre1, t1, L.linear1.weight, L.linear1.bias, dtype, residual=dy2, side=side, bias_grad_done=fuse_b1)
        dy1 = ln_bwd(dt1, y1, t1, mean1, rstd1, L.norm1, ensure_grad(at.out_proj.bias))
    fused_attn = p.dim() == 2
    dvec = None
    if fused_attn and FUSE_ROWDOT and a.shape[1] % 128 == 0:
        # D = rowsum(dO * O) per head out of the epilogue of the product that writes dO (the dQ kernel then skips the O rows)
        dvec = torch.empty((a.shape[0], a.shape[1] // 128), device=a.device, dtype=torch.float32)
    da = linear_bwd(dy1, a, at.out_proj.weight, at.out_proj.bias, dtype, bias_grad_done=True, side=side,
                    dx_rowdot=(a, dvec) if dvec is not None else None)
    return _layer_bwd_attn(da, dvec, dy1, L, saved, n, s, h, dtype, side, prev)


def _layer_bwd_attn(da, dvec, dy1, L, saved, n, s, h, dtype, side, prev):
    """The rest of a layer's backward behind the out-projection: attention and in_proj (da: gradient of the attention output, dvec: its row dots with
    the output per head or None, dy1: the gradient that joins as the residual)."""
    t, qkv, p, a = saved[0], saved[1], saved[2], saved[3]
    at = L.self_attn
    fuse_bq = False
    if p.dim() == 2:
        # in_proj's bias gradient = column sums of dqkv: out of the attention kernels' staged output tiles
        fuse_bq = FUSE_BIAS_GRAD and at.in_proj_bias is not None and at.in_proj_bias.requires_grad
        dqkv = ops.attention_bwd_fused(qkv, a, da, p, n, s, h, dbias=ensure_grad(at.in_proj_bias) if fuse_bq else None, dvec=dvec)
    else:
        dqkv = attention_bwd(qkv, p, da, n, s, h)
    if prev is not None:   # the lower layer's norm2 (its output is this layer's input t) in the epilogue of in_proj's input-gradient product
        rstd2_prev, Lp = prev
        dy2_prev = linear_bwd_ln(dqkv, t, at.in_proj_weight, at.in_proj_bias, dtype, dy1, t, rstd2_prev, Lp.norm2, ensure_grad(Lp.linear2.bias),
                                 side=side, bias_grad_done=fuse_bq)
        if dy2_prev is not None:
            return dy2_prev, True
    return linear_bwd(dqkv, t, at.in_proj_weight, at.in_proj_bias, dtype, residual=dy1, side=side, bias_grad_done=fuse_bq), False


ROW_SPARSE_LAST_LAYER = True   # bf16 mode: when the gradient of the backbone's output is nonzero on a KNOWN list of rows only (the masked positions of the head's
                               # loss: masked_pretraining/model.py _HeadCEFn, ~15 % of the positions), the last layer's row-wise part - norm2, linear2, linear1,
                               # norm1, the out-projection - runs its backward on those rows alone (gathered into whole 256-row tiles) and scatters into zero
                               # matrices in front of the attention backward, which mixes the rows.  The dropped terms are products with exact zeros.

_row_grad_hint = None
row_sparse_steps = 0   # backward passes whose last layer ran on the listed rows (tests read it)


def set_row_grad_hint(dense, index, compact, nrows):
    """The producer of a row-sparse gradient announces it: `dense` (the tensor it returns to autograd) is zero outside the rows `index` (int64, device), whose
    values are compact[:nrows] (rows nrows.. of compact are zero)."""
    global _row_grad_hint
    _row_grad_hint = (dense.data_ptr(), dense.numel(), index, compact, nrows)


def drop_row_grad_hint():
    global _row_grad_hint
    _row_grad_hint = None


def take_row_grad_hint(dense):
    """(index, compact, nrows) if `dense` is the announced tensor, else None; the announcement is consumed either way."""
    global _row_grad_hint
    h, _row_grad_hint = _row_grad_hint, None
    if h is None or not ROW_SPARSE_LAST_LAYER or h[0] != dense.data_ptr() or h[1] != dense.numel():
        return None
    return h[2], h[3], h[4]


def layer_bwd_rows(dtc, index, nrows, L, saved, n, s, h, dtype, side=None, prev=None):
    """layer_bwd for a gradient of the layer's output that is nonzero on the rows `index` only: dtc = those rows (then zero rows up to a multiple of 256).
    Returns what layer_bwd returns, or None when this layer's saved state does not allow it (the caller then runs layer_bwd on the dense gradient)."""
    t, qkv, p, a, y1, mean1, rstd1, t1, hdn, y2, mean2, rstd2, bits, t2 = saved
    if dtype != torch.bfloat16 or y1 is not None or y2 is not None or p.dim() != 2 or dtc.shape[0] % 256 or not DX_ON_WT:
        return None
    at = L.self_attn
    n_pad, M, d = dtc.shape[0], t.shape[0], t.shape[1]

    def rows_of(x):
        return ops.gather_rows(x, index, n_rows_out=n_pad)

    def vec_of(v):   # a per-row f32 statistic of the listed rows (pad rows: 1)
        out = torch.ones(n_pad, device=v.device, dtype=v.dtype)
        out[:nrows] = v[index]
        return out

    # norm2 (its dx column sums are linear2's bias gradient)
    t2c = rows_of(t2)
    dy2c = ops.layernorm_bwd_out(dtc, t2c, vec_of(rstd2), L.norm2.weight.detach(), L.norm2.bias.detach(), ensure_grad(L.norm2.weight), ensure_grad(L.norm2.bias),
                                 ensure_grad(L.linear2.bias))
    # linear2: the ReLU gate from the hidden rows themselves (their bit mask is laid out for whole tiles)
    hdnc = rows_of(hdn)
    fuse_b1 = FUSE_BIAS_GRAD and FUSE_B1_COLSUM and L.linear1.bias is not None and L.linear1.bias.requires_grad
    dpre1c = linear_bwd(dy2c, hdnc, L.linear2.weight, L.linear2.bias, dtype, gate=hdnc, bias_grad_done=True, side=side,
                        dx_colsum_into=ensure_grad(L.linear1.bias) if fuse_b1 else None)
    del hdnc
    # linear1 + norm1
    t1c, r1c = rows_of(t1), vec_of(rstd1)
    dy1c = linear_bwd_ln(dpre1c, t1c, L.linear1.weight, L.linear1.bias, dtype, dy2c, t1c, r1c, L.norm1, ensure_grad(at.out_proj.bias), side=side,
                         bias_grad_done=fuse_b1)
    if dy1c is None:
        dt1c = linear_bwd(dpre1c, t1c, L.linear1.weight, L.linear1.bias, dtype, residual=dy2c, side=side, bias_grad_done=fuse_b1)
        dy1c = ops.layernorm_bwd_out(dt1c, t1c, r1c, L.norm1.weight.detach(), L.norm1.bias.detach(), ensure_grad(L.norm1.weight), ensure_grad(L.norm1.bias),
                                     ensure_grad(at.out_proj.bias))
    del dpre1c
    # out-projection (+ D = rowsum(dO * O) per head for the attention backward)
    ac = rows_of(a)
    nh = a.shape[1] // 128
    dvecc = torch.empty((n_pad, nh), device=a.device, dtype=torch.float32) if FUSE_ROWDOT and a.shape[1] % 128 == 0 else None
    dac = linear_bwd(dy1c, ac, at.out_proj.weight, at.out_proj.bias, dtype, bias_grad_done=True, side=side, dx_rowdot=(ac, dvecc) if dvecc is not None else None)
    # back to all positions: exact zeros elsewhere
    da = ops.scatter_add_rows(dac, index, ops.zeros((M, a.shape[1]), a.device, dtype))
    dy1 = ops.scatter_add_rows(dy1c, index, ops.zeros((M, d), a.device, dtype))
    dvec = None
    if dvecc is not None:
        dvec = ops.zeros((M, nh), a.device, torch.float32)
        dvec[index] = dvecc[:nrows]
    global row_sparse_steps
    row_sparse_steps += 1
    return _layer_bwd_attn(da, dvec, dy1, L, saved, n, s, h, dtype, side, prev)


# ---------------------------------------------------------------------------------------------
# whole backbone: front end + L layers
# ---------------------------------------------------------------------------------------------
def backbone_fwd(mod, x, mask, offsets, dtype, save):
    """x: uint8 (N,H,W,C) line images, or float32 (N,C,H,W) (the reference's model input).
    Returns (tokens (N*S, d) in `dtype`, saved activations or None)."""
    P = mod.patch_size[1]
    tile = mod.mask_tile_device(x.device)
    if mask is not None:
        mask = torch.as_tensor(mask).to(device=x.device, dtype=torch.int64).contiguous()
    d = mod.model_dim
    kp = mod.conv_layer.weight[0].numel()                      # C*H*P = 960
    # bf16 path: pad the patch rows to a pitch that is a multiple of 128 so that the patch-embedding weight
    # gradient (an N = C*H*P GEMM) runs on the fast tile kernel; the pad columns are zeros
    pitch = kp if dtype == torch.float32 else ((kp + 127) // 128) * 128
    if x.dtype == torch.uint8:
        n, _, w, _ = x.shape
        a0 = ops.patches_from_u8(x.contiguous(), mask, tile, P, dtype, pitch)
    else:
        n, _, _, w = x.shape
        xc = x.detach()
        if xc.dtype != torch.float32 or not xc.is_contiguous():
            xc = xc.float().contiguous()
        a0 = ops.patches_from_f32(xc, mask, tile, P, dtype, pitch)
    s = w // P
    y0 = ops.gemm(a0[:, :kp], lowp.weight(mod.conv_layer.weight, dtype).view(d, -1), bias=mod.conv_layer.bias.detach(),
                  extra_flags=FWD_TILE_FLAGS)
    pe = mod.position_model.pe_table(x.device)
    t, mean0, rstd0 = ops.layernorm_fwd(y0, mod.intermediate_norm.weight.detach(), mod.intermediate_norm.bias.detach(),
                                        mod.intermediate_norm.eps, pe=pe, offsets=offsets, S=s)
    layers = []
    for L in mod.encoder_layers.layers:
        t, sv = layer_fwd(t, L, n, s, mod.num_heads, dtype, save)
        layers.append(sv)
    saved = (a0, y0, mean0, rstd0, layers, n, s) if save else None
    return t, saved


def backbone_bwd(mod, saved, dt, dtype, on_layer_done=None, rows=None):
    """rows = (index, compact, nrows): dt is zero outside the rows `index`, whose values are compact[:nrows] (take_row_grad_hint)."""
    a0, y0, mean0, rstd0, layers, n, s = saved
    nl = len(layers)
    side = SideStream(dt.device)
    is_dy2 = False
    for i in range(nl - 1, -1, -1):
        prev = None
        if i > 0 and layers[i - 1][9] is None:   # the lower layer kept no pre-norm rows: its norm2 backward runs from its output (= this layer's input)
            prev = (layers[i - 1][11], mod.encoder_layers.layers[i - 1])
        res = None
        if i == nl - 1 and rows is not None:
            res = layer_bwd_rows(rows[1], rows[0], rows[2], mod.encoder_layers.layers[i], layers[i], n, s, mod.num_heads, dtype, side, prev=prev)
        if res is None:
            res = layer_bwd(dt, mod.encoder_layers.layers[i], layers[i], n, s, mod.num_heads, dtype, side, dt2_is_dy2=is_dy2, prev=prev)
        dt, is_dy2 = res
        layers[i] = None
        if on_layer_done is not None:
            with side.comm_context():  # sees the layer's gradient kernels on the main and the side streams
                on_layer_done(i)
    nrm = mod.intermediate_norm
    dy0 = ops.layernorm_bwd(dt, y0, mean0, rstd0, nrm.weight.detach(), ensure_grad(nrm.weight), ensure_grad(nrm.bias),
                            ensure_grad(mod.conv_layer.bias))
    cw = mod.conv_layer.weight
    kp = cw[0].numel()
    if a0.shape[1] == kp:
        linear_bwd(dy0, a0, cw, mod.conv_layer.bias, dtype, need_dx=False, bias_grad_done=True)
    elif cw.requires_grad:  # padded pitch: dW into a (d, pitch) scratch, then add the real columns into .grad
        tmp = ops.zeros((cw.shape[0], a0.shape[1]), a0.device, torch.float32)
        ops.gemm(dy0, a0, out=tmp, trans_a=True, trans_b=True, atomic=True, k_split=0)
        ops.add_rows2d(ensure_grad(cw).view(cw.shape[0], kp), tmp, kp)
    side.join()
    if on_layer_done is not None:
        on_layer_done(-1)
