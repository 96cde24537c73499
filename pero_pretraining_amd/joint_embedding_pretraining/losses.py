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


_host_groups = {}


def _host_count_sum(group, *counts):
    """Sums of a few HOST integers over the ranks of `group` without touching the GPU queue: the counts are all-reduced as a CPU tensor over
    a gloo companion group of the same ranks (created once per group; a gloo group serves itself).  Round 3 all-reduced them on the device
    and read them back with .tolist(): one device sync per step in the global-statistics VICReg."""
    t = torch.tensor(list(counts), dtype=torch.int64)
    g = group
    if dist.get_backend(group) != "gloo":
        key = id(group)
        if key not in _host_groups:
            _host_groups[key] = dist.new_group(ranks=dist.get_process_group_ranks(group), backend="gloo")
        g = _host_groups[key]
    dist.all_reduce(t, group=g)
    return tuple(int(v) for v in t.tolist())


class _VICRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, ix, iy, jx, jy, wv, wi, wc, thr, eps, dtype, group=None):
        # ix, iy, jx, jy: index lists of the invariance rows (shift masks == 1) and of the statistics rows (image masks == 1)
        # group: a torch.distributed process group -> statistics over the lines of ALL its ranks (see VICRegLoss)
        D = x.shape[-1]
        stacked = y is None   # x = both views stacked along dim 0 (one tensor, ONE gradient: no slice-backward fill / copy / add kernels)
        if stacked:
            xy = _rows(x, dtype)
            x2, y2 = xy[:xy.shape[0] // 2], xy[xy.shape[0] // 2:]
        else:
            x2, y2 = _rows(x, dtype), _rows(y, dtype)
        if ix.numel() != iy.numel():
            raise RuntimeError(f"The size of tensor a ({ix.numel()}) must match the size of tensor b ({iy.numel()}) "
                               "at non-singleton dimension 0")  # what mse_loss reports in the reference
        n_inv = ix.numel()
        n1, m_loc = jx.numel(), jx.numel() + jy.numel()
        m, world = m_loc, 1
        if group is not None:  # row counts of the whole batch (exact integers: int64 sum)
            world = dist.get_world_size(group)
            n_inv, m = _host_count_sum(group, n_inv, m_loc)
        inv = ops.sqdiff_rows(x2, ix, y2, iy, 1.0 / (n_inv * D))
        m_pad = ((m_loc + 63) // 64) * 64
        z = torch.empty((m_pad, D), device=x2.device, dtype=dtype)
        ops.gather_rows(x2, jx, out=z[:n1])
        ops.gather_rows(y2, jy, n_rows_out=m_pad - n1, out=z[n1:])
        cs = ops.zeros((D,), x2.device, torch.float32)
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
            cov = ops.zeros((D, D), x2.device, torch.float32)   # (64 MiB at D = 4096: the library's linear fill, not torch's elementwise one)
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
        ctx.meta = (x.shape, None if stacked else y.shape, n1, m_loc, wi * 2.0 / (n_inv * D), dtype)
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
        if ys is None:   # stacked views: one gradient buffer, the two halves are the views' gradients
            dxy = ops.zeros((2 * x2.shape[0], x2.shape[1]), x2.device, x2.dtype)
            dx, dy = dxy[:x2.shape[0]], dxy[x2.shape[0]:]
        else:
            dx = ops.zeros(x2.shape, x2.device, x2.dtype)
            dy = ops.zeros(y2.shape, y2.device, y2.dtype)
        ops.scatter_add_rows_scaled(dzc[:n1], jx, dx, gdev)
        ops.scatter_add_rows_scaled(dzc[n1:m], jy, dy, gdev)
        ops.sqdiff_rows_bwd(x2, ix, y2, iy, dx, dy, gdev, inv_coef)
        if ys is None:
            return (dxy.view(xs), None) + (None,) * 11
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

    def forward_stacked(self, xy, image_masks1, image_masks2, shift_masks1, shift_masks2):
        """forward(xy[:n], xy[n:], ...) for the two views' head outputs stacked along dim 0 (what the batched 2N-line encode produces): the
        same values, and ONE gradient tensor for xy - slicing the views apart made autograd fill, copy and add three tensors of xy's size."""
        return self.forward(xy, None, image_masks1, image_masks2, shift_masks1, shift_masks2)


    def _group(self):
        if not self.global_statistics:
            return None
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("VICRegLoss(global_statistics=True) needs an initialised torch.distributed process group")
        return self.process_group if self.process_group is not None else dist.group.WORLD


class _NTXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, temperature, dtype):
        stacked = y is None   # x = both views stacked along dim 0: one normalisation launch, ONE gradient tensor
        if stacked:
            n, s, D = x.shape[0] // 2, x.shape[1], x.shape[2]
            xyn, invxy = ops.rownorm_fwd(_rows(x, dtype))
        else:
            n, s, D = x.shape
            # (the two views normalised into the halves of one buffer: the backward then is the stacked one's)
            xyn = torch.empty((2 * n * s, D), device=x.device, dtype=dtype)
            invxy = torch.empty(2 * n * s, device=x.device, dtype=torch.float32)
            ops.rownorm_fwd(_rows(x, dtype), out=(xyn[:n * s], invxy[:n * s]))
            ops.rownorm_fwd(_rows(y, dtype), out=(xyn[n * s:], invxy[n * s:]))
        xn, yn = xyn[:n * s], xyn[n * s:]
        sim = torch.empty((n, s, s), device=x.device, dtype=torch.float32)
        ops.gemm_raw(xn, yn, sim, s, s, D, D, D, s, batch=n, sA=(s * D, 0), sB=(s * D, 0), sC=(s * s, 0),
                     alpha=1.0 / temperature)
        loss, _, dsim = ops.ntxent_cols(sim, dtype)
        ctx.save_for_backward(xyn, invxy, dsim)
        ctx.meta = ((n, s, D), temperature, stacked)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        xyn, invxy, dsim = ctx.saved_tensors
        (n, s, D), temperature, stacked = ctx.meta
        xn, yn = xyn[:n * s], xyn[n * s:]
        gdev = g.detach().reshape(1).to(torch.float32)
        dxyn = torch.empty_like(xyn)
        dxn, dyn = dxyn[:n * s], dxyn[n * s:]
        # d xn = dsim @ yn / T ; d yn = dsim^T @ xn / T   (per line)
        ops.gemm_raw(dsim, yn, dxn, s, D, s, s, D, D, batch=n, sA=(s * s, 0), sB=(s * D, 0), sC=(s * D, 0),
                     alpha=1.0 / temperature, flags=GEMM_TRANS_B)
        ops.gemm_raw(dsim, xn, dyn, s, D, s, s, D, D, batch=n, sA=(s * s, 0), sB=(s * D, 0), sC=(s * D, 0),
                     alpha=1.0 / temperature, flags=GEMM_TRANS_A | GEMM_TRANS_B)
        dxy = ops.rownorm_bwd(xyn, dxyn, invxy, gdev)     # both views: one launch, one tensor
        if stacked:
            return dxy.view(2 * n, s, D), None, None, None
        return dxy[:n * s].view(n, s, D), dxy[n * s:].view(n, s, D), None, None


def _is_nccl(group):
    try:
        return dist.get_backend(group) == "nccl"
    except Exception:  # noqa: BLE001
        return False


class _NTXentCrossFn(torch.autograd.Function):
    """NT-Xent with cross-rank negatives as ONE autograd node on HIP kernels (no torch elementwise / cat / logsumexp kernels, no host
    sync): row normalisation (pero_rownorm_*), the per-line pooled embedding (pero_line_mean / pero_add_line_rows), the two similarity
    products and their backward (pero_gemm), the joint column log-sum-exp over [own rows | gathered negatives] and its gradients
    (pero_ntxent_cols_cross).  The only exchange: the pooled embeddings, all-gathered forward (world x N x D f32), their gradients
    reduce-scattered back to the owning rank in the backward (RCCL: reduce_scatter_tensor into a fresh tensor; gloo: all-reduce of a
    private copy).  The upstream gradient is applied once, at the end (it is the same scalar on every rank of a data-parallel step)."""

    @staticmethod
    def forward(ctx, x, y, temperature, dtype, group):
        n, s, D = x.shape
        T = float(temperature)
        xn, invx = ops.rownorm_fwd(_rows(x, dtype))
        yn, invy = ops.rownorm_fwd(_rows(y, dtype))
        pm = ops.line_mean(xn, n, s)                                   # (N, D) f32: mean of the normalised view-1 rows of a line
        p, invp = ops.rownorm_fwd(pm)                                  # one bounded negative per line
        world, rank = 1, 0
        gathered = p
        if group is not None:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            gathered = torch.empty((world * n, D), device=p.device, dtype=p.dtype)
            if _is_nccl(group):
                dist.all_gather_into_tensor(gathered, p, group=group)
            else:
                parts = list(gathered.view(world, n, D).unbind(0))
                dist.all_gather(parts, p, group=group)
        L = gathered.shape[0]
        if dtype == torch.bfloat16:
            glp = ops.cast_to_bf16(gathered, torch.empty((L, D), device=p.device, dtype=torch.bfloat16))
        else:
            glp = gathered
        sim = torch.empty((n, s, s), device=x.device, dtype=torch.float32)
        ops.gemm_raw(xn, yn, sim, s, s, D, D, D, s, batch=n, sA=(s * D, 0), sB=(s * D, 0), sC=(s * s, 0), alpha=1.0 / T)
        cross = ops.gemm(yn, glp, alpha=1.0 / T, out_dtype=torch.float32)          # (N*S, L): y_j . p_l' / T
        loss, _, dsim, dcross = ops.ntxent_cols_cross(sim, cross, rank * n, dtype)
        ctx.save_for_backward(xn, yn, invx, invy, dsim, dcross, p, invp, glp)
        ctx.meta = (n, s, D, T, group, world, rank)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        xn, yn, invx, invy, dsim, dcross, p, invp, glp = ctx.saved_tensors
        n, s, D, T, group, world, rank = ctx.meta
        gdev = g.detach().reshape(1).to(torch.float32)
        dxn = torch.empty_like(xn)
        dyn = torch.empty_like(yn)
        ops.gemm_raw(dsim, yn, dxn, s, D, s, s, D, D, batch=n, sA=(s * s, 0), sB=(s * D, 0), sC=(s * D, 0), alpha=1.0 / T, flags=GEMM_TRANS_B)
        ops.gemm_raw(dsim, xn, dyn, s, D, s, s, D, D, batch=n, sA=(s * s, 0), sB=(s * D, 0), sC=(s * D, 0), alpha=1.0 / T,
                     flags=GEMM_TRANS_A | GEMM_TRANS_B)
        # through the negatives: d yn += dcross @ P / T ;  d P = dcross^T @ yn / T  (f32: it is summed over the ranks)
        dyn = ops.gemm(dcross, glp, trans_b=True, alpha=1.0 / T, residual=dyn)
        dgath = ops.gemm(dcross, yn, trans_a=True, trans_b=True, alpha=1.0 / T, out_dtype=torch.float32)      # (L, D)
        if group is not None:
            if _is_nccl(group):
                dp = torch.empty((n, D), device=dgath.device, dtype=torch.float32)
                dist.reduce_scatter_tensor(dp, dgath, op=dist.ReduceOp.SUM, group=group)
            else:
                dist.all_reduce(dgath, op=dist.ReduceOp.SUM, group=group)      # (dgath is this function's own tensor)
                dp = dgath[rank * n:(rank + 1) * n].contiguous()
        else:
            dp = dgath
        dpm = ops.rownorm_bwd(p, dp, invp)                              # through p = normalize(pm)
        ops.add_line_rows_(dxn, dpm, n, s, 1.0 / s)                     # through pm = mean over the line's rows of xn
        dx = ops.rownorm_bwd(xn, dxn, invx, gdev)
        dy = ops.rownorm_bwd(yn, dyn, invy, gdev)
        return dx.view(n, s, D), dy.view(n, s, D), None, None, None


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
    oracle/pero_oracle.py::ntxent_cross_loss; everything runs on HIP kernels (_NTXentCrossFn)."""

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
        if y is None:
            n = x.shape[0] // 2
            x, y = x[:n], x[n:]
        return {"loss": self._cross(x, y)}

    def forward_stacked(self, xy, image_masks1, image_masks2, shift_masks1, shift_masks2):
        """forward(xy[:n], xy[n:], ...) for the two views stacked along dim 0: same values, ONE gradient tensor for xy (see VICRegLoss)."""
        return self.forward(xy, None, image_masks1, image_masks2, shift_masks1, shift_masks2)

    def _cross(self, x, y):
        group = None
        if dist.is_available() and dist.is_initialized():
            group = self.process_group if self.process_group is not None else dist.group.WORLD
        return _NTXentCrossFn.apply(x, y, float(self.temperature), compute_dtype(), group)
