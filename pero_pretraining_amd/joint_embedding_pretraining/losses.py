"""VICRegLoss and NTXentLoss with the interface of the reference's joint_embedding_pretraining/losses.py,
computed by HIP kernels: boolean-mask row selections become index gathers, the covariance (a D x D SYRK)
and the per-line similarity matrices run on pero_gemm, the statistics are f32 reductions."""
import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from .._lib import GEMM_TRANS_A, GEMM_TRANS_B
from ..precision import compute_dtype


def _rows(t, dtype):
    t2 = t.detach().reshape(-1, t.shape[-1])
    if t2.dtype != dtype or not t2.is_contiguous():
        t2 = t2.to(dtype).contiguous()
    return t2


host_mask = ops.host_mask


def _nz(mask, device, value=1):
    """Index list (int64, on `device`) of the positions with mask == value.  With a host copy of the mask the list is built on the
    host and uploaded asynchronously - no device sync; a mask that exists on the device only costs one torch.nonzero sync, like
    the reference's boolean indexing (joint_embedding_pretraining/losses.py:14-22)."""
    h = host_mask(mask)
    if h is not None:
        idx = np.flatnonzero(np.asarray(h).reshape(-1) == value).astype(np.int64)
        return torch.from_numpy(idx).to(device, non_blocking=True)
    return torch.nonzero(torch.as_tensor(mask).to(device).reshape(-1) == value, as_tuple=False).reshape(-1).contiguous()


class _VICRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, ix, iy, jx, jy, wv, wi, wc, thr, eps, dtype, group=None):
        # ix, iy, jx, jy: index lists of the invariance rows (shift masks == 1) and of the statistics rows (image masks == 1)
        # group: a torch.distributed process group -> statistics over the lines of ALL its ranks (see VICRegLoss)
        D = x.shape[-1]
        x2, y2 = _rows(x, dtype), _rows(y, dtype)
        if ix.numel() != iy.numel():
            raise RuntimeError(f"The size of tensor a ({ix.numel()}) must match the size of tensor b ({iy.numel()}) "
                               "at non-singleton dimension 0")  # what mse_loss reports in the reference
        n_inv = ix.numel()
        n1, m_loc = jx.numel(), jx.numel() + jy.numel()
        m, world = m_loc, 1
        if group is not None:  # row counts of the whole batch (exact integers: int64 sum)
            counts = torch.tensor([n_inv, m_loc], device=x2.device, dtype=torch.int64)
            dist.all_reduce(counts, group=group)
            n_inv, m = (int(v) for v in counts.tolist())
            world = dist.get_world_size(group)
        inv = ops.sqdiff_rows(x2, ix, y2, iy, 1.0 / (n_inv * D))
        m_pad = ((m_loc + 63) // 64) * 64
        z = torch.empty((m_pad, D), device=x2.device, dtype=dtype)
        ops.gather_rows(x2, jx, out=z[:n1])
        ops.gather_rows(y2, jy, n_rows_out=m_pad - n1, out=z[n1:])
        cs = torch.zeros(D, device=x2.device, dtype=torch.float32)
        ops.colsum(z, cs)
        if group is not None:
            # exchange 1: column sums (D floats) -> the GLOBAL mean; pero_center_cols divides by its row count, so the
            # global sums are rescaled by m_loc / m.  The invariance partial sum rides along.
            pack = torch.cat([cs, inv])
            dist.all_reduce(pack, group=group)
            cs, inv = pack[:D] * (m_loc / m), pack[D:].clone()
        zc, sumsq = ops.center_cols(z, cs.contiguous(), m_loc)
        # rows centred with the global mean: sum over ranks of zc^T zc IS the global scatter matrix
        if dtype == torch.bfloat16:
            # the D x D SYRK as a split-K product into a zeroed f32 matrix: the long-reduction mode of the 256x256x64 kernel
            cov = torch.zeros((D, D), device=x2.device, dtype=torch.float32)
            ops.gemm(zc, zc, out=cov, trans_a=True, trans_b=True, alpha=1.0 / (m - 1), atomic=True, k_split=0)
        else:
            cov = ops.gemm(zc, zc, trans_a=True, trans_b=True, alpha=1.0 / (m - 1), out_dtype=torch.float32)
        if group is not None:
            # exchange 2: the D x D scatter matrix (f32: 64 MiB at D = 4096) and the D squared column norms
            dist.all_reduce(cov, group=group)
            dist.all_reduce(sumsq, group=group)
        cvar, var = ops.vicreg_var(sumsq, m, thr, eps)
        G, covl = ops.vicreg_cov(cov, cvar, m, wv, wc, dtype)
        loss = wv * var + wi * inv + wc * covl
        ctx.save_for_backward(x2, y2, ix, iy, jx, jy, zc, G)
        ctx.meta = (x.shape, y.shape, n1, m_loc, wi * 2.0 / (n_inv * D), dtype)
        ctx.seed = float(world)
        ctx.mark_non_differentiable(var, inv, covl)
        return loss[0], var[0], inv[0], covl[0]

    @staticmethod
    def backward(ctx, g, _gv, _gi, _gc):
        x2, y2, ix, iy, jx, jy, zc, G = ctx.saved_tensors
        xs, ys, n1, m, inv_coef, dtype = ctx.meta
        # global statistics: every rank holds the SAME loss and differentiates it w.r.t. its own rows; the data-parallel
        # gradient AVERAGE over ranks would divide the sum of those parts by world, so the seed is multiplied by world
        gdev = g.detach().reshape(1).to(torch.float32) * ctx.seed
        dzc = ops.gemm(zc, G)  # (m_pad, D): d(wv*var + wc*cov)/d zc
        dx = torch.zeros_like(x2)
        dy = torch.zeros_like(y2)
        ops.scatter_add_rows_scaled(dzc[:n1], jx, dx, gdev)
        ops.scatter_add_rows_scaled(dzc[n1:m], jy, dy, gdev)
        ops.sqdiff_rows_bwd(x2, ix, y2, iy, dx, dy, gdev, inv_coef)
        return (dx.view(xs), dy.view(ys)) + (None,) * 11


class VICRegLoss(torch.nn.Module):
    """joint_embedding_pretraining/losses.py:3-47.

    global_statistics=True (no reference counterpart on one device): mean, variance, covariance
    and the invariance mean are taken over the lines of ALL ranks of `process_group` - two exchanges per step (D + 1
    floats; D x D + D floats).  Every rank then returns the loss the reference computes on the concatenated batch, and
    the backward is seeded with world_size, so that data-parallel gradient averaging yields exactly the single-process
    gradient of that batch.  Default False: per-rank statistics, i.e. the reference's loss on each rank's shard."""

    def __init__(self, variance_weight=1.0, invariance_weight=1.0, covariance_weight=1.0, variance_threshold=1.0,
                 global_statistics=False, process_group=None):
        super().__init__()
        self.global_statistics = global_statistics
        self.process_group = process_group
        self.variance_weight = variance_weight
        self.invariance_weight = invariance_weight
        self.covariance_weight = covariance_weight
        self.variance_threshold = variance_threshold
        self.eps = 1e-5

    def forward(self, x, y, image_masks1, image_masks2, shift_masks1, shift_masks2):
        if not x.is_cuda:
            raise RuntimeError("pero_pretraining_amd losses run on the GPU only (HIP kernels, no CPU fallback)")
        dev = x.device
        sel = [_nz(m, dev) for m in (shift_masks1, shift_masks2, image_masks1, image_masks2)]
        loss, var, inv, cov = _VICRegFn.apply(x, y, *sel, float(self.variance_weight), float(self.invariance_weight),
                                              float(self.covariance_weight), float(self.variance_threshold), self.eps,
                                              compute_dtype(), self._group())
        return {"loss": loss, "loss.variance": var, "loss.invariance": inv, "loss.covariance": cov}


    def _group(self):
        if not self.global_statistics:
            return None
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("VICRegLoss(global_statistics=True) needs an initialised torch.distributed process group")
        return self.process_group if self.process_group is not None else dist.group.WORLD


class _NTXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, temperature, dtype):
        n, s, D = x.shape
        xn, invx = ops.rownorm_fwd(_rows(x, dtype))
        yn, invy = ops.rownorm_fwd(_rows(y, dtype))
        sim = torch.empty((n, s, s), device=x.device, dtype=torch.float32)
        ops.gemm_raw(xn, yn, sim, s, s, D, D, D, s, batch=n, sA=(s * D, 0), sB=(s * D, 0), sC=(s * s, 0),
                     alpha=1.0 / temperature)
        loss, _, dsim = ops.ntxent_cols(sim, dtype)
        ctx.save_for_backward(xn, yn, invx, invy, dsim)
        ctx.meta = (x.shape, temperature)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        xn, yn, invx, invy, dsim = ctx.saved_tensors
        (n, s, D), temperature = ctx.meta
        gdev = g.detach().reshape(1).to(torch.float32)
        dxn = torch.empty_like(xn)
        dyn = torch.empty_like(yn)
        # d xn = dsim @ yn / T ; d yn = dsim^T @ xn / T   (per line)
        ops.gemm_raw(dsim, yn, dxn, s, D, s, s, D, D, batch=n, sA=(s * s, 0), sB=(s * D, 0), sC=(s * D, 0),
                     alpha=1.0 / temperature, flags=GEMM_TRANS_B)
        ops.gemm_raw(dsim, xn, dyn, s, D, s, s, D, D, batch=n, sA=(s * s, 0), sB=(s * D, 0), sC=(s * D, 0),
                     alpha=1.0 / temperature, flags=GEMM_TRANS_A | GEMM_TRANS_B)
        dx = ops.rownorm_bwd(xn, dxn, invx, gdev)
        dy = ops.rownorm_bwd(yn, dyn, invy, gdev)
        return dx.view(n, s, D), dy.view(n, s, D), None, None


class _BmmNT(torch.autograd.Function):
    """per line: out[l] = a[l] @ b[l]^T * alpha   (N, S, D) x (N, T, D) -> (N, S, T) f32, on batched pero_gemm"""

    @staticmethod
    def forward(ctx, a, b, alpha):
        n, s, D = a.shape
        t = b.shape[1]
        out = torch.empty((n, s, t), device=a.device, dtype=torch.float32)
        ops.gemm_raw(a, b, out, s, t, D, D, D, t, batch=n, sA=(s * D, 0), sB=(t * D, 0), sC=(s * t, 0), alpha=alpha)
        ctx.save_for_backward(a, b)
        ctx.alpha = alpha
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        n, s, D = a.shape
        t = b.shape[1]
        gd = g.contiguous().to(a.dtype)
        da, db = torch.empty_like(a), torch.empty_like(b)
        ops.gemm_raw(gd, b, da, s, D, t, t, D, D, batch=n, sA=(s * t, 0), sB=(t * D, 0), sC=(s * D, 0), alpha=ctx.alpha, flags=GEMM_TRANS_B)
        ops.gemm_raw(gd, a, db, t, D, s, t, D, D, batch=n, sA=(s * t, 0), sB=(s * D, 0), sC=(t * D, 0), alpha=ctx.alpha,
                     flags=GEMM_TRANS_A | GEMM_TRANS_B)
        return da, db, None


class _MmNT(torch.autograd.Function):
    """out = a @ b^T * alpha   (M, D) x (L, D) -> (M, L) f32, on pero_gemm"""

    @staticmethod
    def forward(ctx, a, b, alpha):
        out = ops.gemm(a, b, alpha=alpha, out_dtype=torch.float32)
        ctx.save_for_backward(a, b)
        ctx.alpha = alpha
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        gd = g.contiguous().to(a.dtype)
        da = ops.gemm(gd, b, trans_b=True, alpha=ctx.alpha)                  # (M, L) x (L, D)
        db = ops.gemm(gd, a, trans_a=True, trans_b=True, alpha=ctx.alpha)    # (L, M) x (M, D)
        return da, db, None


class _AllGatherRows(torch.autograd.Function):
    """all-gather of equally shaped row blocks over the ranks; backward: every rank's gradient block summed back to its owner"""

    @staticmethod
    def forward(ctx, t, group):
        ctx.group, ctx.world, ctx.rank = group, dist.get_world_size(group), dist.get_rank(group)
        parts = [torch.empty_like(t) for _ in range(ctx.world)]
        dist.all_gather(parts, t.contiguous(), group=group)
        return torch.cat(parts, dim=0)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)   # (reduce-scatter semantics; all-reduce keeps gloo and RCCL on one path)
        n = g.shape[0] // ctx.world
        return g[ctx.rank * n:(ctx.rank + 1) * n].clone(), None


class NTXentLoss(torch.nn.Module):
    """joint_embedding_pretraining/losses.py:51-83.  Like the reference, only all-ones masks are valid: the
    reference indexes the shift-reduced similarity matrix with the full-length image masks (losses.py:78) and
    raises IndexError for every other input; the same exception is raised here.

    cross_rank_negatives=True (BASELINE.json configs[4] / north_star; NO reference counterpart - the reference's loss is per line,
    SURVEY.md section 8e): every line additionally contributes ONE pooled embedding p = normalize(mean over its positions of the
    normalised view-1 rows); the pooled embeddings of all ranks are all-gathered (a bounded exchange: world x N x D values per
    step - 33.5 MB in bf16 for 8 ranks x 512 lines x 4096, not the 1 GB per rank of the full embeddings) and enter every column's
    normaliser as negatives, the line's own pooled embedding excepted:
        loss_line = mean_j [ log( sum_i exp(x_i . y_j / T) + sum_{lines l' != line, all ranks} exp(p_l' . y_j / T) ) - x_j . y_j / T ].
    Gradients flow back through the gathered rows to the rank that owns them (sum over ranks), so the data-parallel average of
    the parameter gradients is the gradient of the mean loss over the global batch.  The restatement on the concatenated batch is
    oracle/pero_oracle.py::ntxent_cross_loss; the heavy products run on pero_gemm, the exponentials on torch elementwise kernels
    (an extension outside the reference-parity path)."""

    def __init__(self, temperature=0.1, cross_rank_negatives=False, process_group=None):
        super().__init__()
        self.temperature = temperature
        self.cross_rank_negatives = cross_rank_negatives
        self.process_group = process_group

    def forward(self, x, y, image_masks1, image_masks2, shift_masks1, shift_masks2):
        if not x.is_cuda:
            raise RuntimeError("pero_pretraining_amd losses run on the GPU only (HIP kernels, no CPU fallback)")
        for m in (shift_masks1, shift_masks2, image_masks1, image_masks2):
            h = host_mask(m)   # (checked on the host copy when there is one: no device sync)
            if bool((np.asarray(h) != 1).any()) if h is not None else bool((torch.as_tensor(m) != 1).any()):
                raise IndexError("The shape of the mask at index 0 does not match the shape of the indexed tensor "
                                 "(NT-Xent of the reference is only defined for all-ones masks)")
        if not self.cross_rank_negatives:
            return {"loss": _NTXentFn.apply(x, y, float(self.temperature), compute_dtype())}
        return {"loss": self._cross(x, y)}

    def _cross(self, x, y):
        dtype, T = compute_dtype(), float(self.temperature)
        n, s, D = x.shape
        group = None
        if dist.is_available() and dist.is_initialized():
            group = self.process_group if self.process_group is not None else dist.group.WORLD
        xn = torch.nn.functional.normalize(x.float(), dim=-1, eps=1e-12)
        yn = torch.nn.functional.normalize(y.float(), dim=-1, eps=1e-12)
        pooled = torch.nn.functional.normalize(xn.mean(dim=1), dim=-1, eps=1e-12)          # (N, D): one bounded negative per line
        gathered = _AllGatherRows.apply(pooled, group) if group is not None else pooled        # (L, D), L = world * N
        rank = dist.get_rank(group) if group is not None else 0
        xl, yl = xn.to(dtype).contiguous(), yn.to(dtype).contiguous()
        sim = _BmmNT.apply(xl, yl, 1.0 / T)                                                    # (N, S, S): sim[l, i, j] = x_i . y_j / T
        cross = _MmNT.apply(yl.view(n * s, D), gathered.to(dtype).contiguous(), 1.0 / T)       # (N*S, L): y_j . p_l' / T
        cross = cross.view(n, s, -1)
        own = torch.arange(n, device=x.device) + rank * n
        cross = cross.masked_fill(torch.nn.functional.one_hot(own, cross.shape[-1]).bool()[:, None, :], float("-inf"))
        lse = torch.logsumexp(torch.cat([sim.transpose(1, 2), cross], dim=-1), dim=-1)         # per (line, column j): over i and l'
        diag = torch.diagonal(sim, dim1=1, dim2=2)
        return (lse - diag).mean(dim=1).mean()
