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


def host_mask(m):
    """The mask's values on the HOST when they are known there without a device sync: numpy arrays, CPU tensors, and device
    tensors that the batch operator / collator uploaded from host arrays (they carry the original as `_pero_host`)."""
    if isinstance(m, np.ndarray):
        return m
    if isinstance(m, torch.Tensor):
        if not m.is_cuda:
            return m.numpy()
        return getattr(m, "_pero_host", None)
    return np.asarray(m)


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


class NTXentLoss(torch.nn.Module):
    """joint_embedding_pretraining/losses.py:51-83.  Like the reference, only all-ones masks are valid: the
    reference indexes the shift-reduced similarity matrix with the full-length image masks (losses.py:78) and
    raises IndexError for every other input; the same exception is raised here."""

    def __init__(self, temperature=0.1):
        super().__init__()
        self.temperature = temperature

    def forward(self, x, y, image_masks1, image_masks2, shift_masks1, shift_masks2):
        if not x.is_cuda:
            raise RuntimeError("pero_pretraining_amd losses run on the GPU only (HIP kernels, no CPU fallback)")
        for m in (shift_masks1, shift_masks2, image_masks1, image_masks2):
            h = host_mask(m)   # (checked on the host copy when there is one: no device sync)
            if bool((np.asarray(h) != 1).any()) if h is not None else bool((torch.as_tensor(m) != 1).any()):
                raise IndexError("The shape of the mask at index 0 does not match the shape of the indexed tensor "
                                 "(NT-Xent of the reference is only defined for all-ones masks)")
        return {"loss": _NTXentFn.apply(x, y, float(self.temperature), compute_dtype())}
