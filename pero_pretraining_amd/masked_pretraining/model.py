"""Masked pre-training model with the API of the reference's masked_pretraining/model.py
(init_backbone, init_head, MaskedTransformerEncoder, MaskedCrossEntropyLoss, LinearHead), computing
through the HIP kernels."""
import numpy as np
import torch

from .. import functional as F
from .. import lowp, ops
from ..models.transformers import VisionTransformerEncoder
from ..precision import compute_dtype


def init_backbone(backbone_definition):
    """masked_pretraining/model.py:7-17 (the whole dict is splatted into the constructor)."""
    backbone_type = backbone_definition.get("type", "vit")
    if backbone_type == "vit":
        return VisionTransformerEncoder(**backbone_definition)
    if backbone_type == "vggt":
        raise NotImplementedError("vggt (VGG convolutional front end) is outside the HIP hot path (SURVEY.md section 2)")
    raise ValueError(f"Unknown backbone type: {backbone_type}")


def init_head(head_definition):
    """masked_pretraining/model.py:20-30 (pops "type" from the caller's dict, like the reference)."""
    head_type = head_definition.get("type", "linear")
    if "type" in head_definition:
        del head_definition["type"]
    if head_type == "linear":
        return LinearHead(**head_definition)
    raise ValueError(f"Unknown head type: {head_type}")


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on token rows.  relu_out fuses a ReLU into the GEMM epilogue; gate_in declares that x
    itself is the output of such a fused ReLU, so dx is gated by (x > 0) in the backward GEMM epilogue
    (that IS the ReLU backward) - used by the MLP head chain."""

    @staticmethod
    def forward(ctx, x, weight, bias, dtype, relu_out, gate_in):
        x2 = x.detach().reshape(-1, x.shape[-1])
        if x2.dtype != dtype or not x2.is_contiguous():
            x2 = x2.to(dtype).contiguous()
        y = F.linear_fwd(x2, weight, bias, dtype, relu=relu_out)
        ctx.dtype, ctx.gate_in, ctx.x_shape = dtype, gate_in, x.shape
        ctx.weight, ctx.bias = weight, bias
        ctx.save_for_backward(x2)
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != ctx.dtype or not dy2.is_contiguous():
            dy2 = dy2.to(ctx.dtype).contiguous()
        dx = F.linear_bwd(dy2, x2, ctx.weight, ctx.bias, ctx.dtype, need_dx=ctx.needs_input_grad[0],
                          gate=x2 if ctx.gate_in else None)
        return (dx.view(ctx.x_shape) if dx is not None else None), None, None, None, None, None


def linear(x, weight, bias, relu_out=False, gate_in=False):
    if not x.is_cuda:
        raise RuntimeError("pero_pretraining_amd layers run on the GPU only (HIP kernels, no CPU fallback)")
    return _LinearFn.apply(x, weight, bias, compute_dtype(), relu_out, gate_in)


class LinearHead(torch.nn.Module):
    def __init__(self, in_features=512, out_features=4096):
        super().__init__()
        self.linear = torch.nn.Linear(in_features, out_features)  # parameter container (same init / keys)

    def forward(self, x):
        return linear(x, self.linear.weight, self.linear.bias)


class _MaskedCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, labels, mask, unmasked_weight):
        lg = output.detach().reshape(-1, output.shape[-1])
        if not lg.is_contiguous():
            lg = lg.contiguous()
        lab = labels.reshape(-1).to(device=lg.device, dtype=torch.int64).contiguous()
        msk = mask.reshape(-1).to(device=lg.device, dtype=torch.int64).contiguous()
        loss, work = ops.masked_ce_fwd(lg, lab, msk, unmasked_weight)
        ctx.save_for_backward(lg, lab, msk, work)
        ctx.shape, ctx.uw = output.shape, unmasked_weight
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        lg, lab, msk, work = ctx.saved_tensors
        dl = dloss.detach().reshape(1).to(torch.float32)  # stays on the device: no host sync
        return ops.masked_ce_bwd(lg, lab, msk, work, ctx.uw, dloss=dl).view(ctx.shape), None, None, None


class _GatherTokensFn(torch.autograd.Function):
    """rows[i] = tokens[index[i]] for i < index.numel(), zero rows up to n_out; the backward scatters the row gradients
    into a zero matrix (the index entries are distinct positions)."""

    @staticmethod
    def forward(ctx, tokens, index, n_out):
        ctx.save_for_backward(index)
        ctx.shape = tokens.shape
        return ops.gather_rows(tokens.detach(), index, n_rows_out=n_out)

    @staticmethod
    def backward(ctx, drows):
        (index,) = ctx.saved_tensors
        dx = ops.zeros(ctx.shape, drows.device, drows.dtype)
        ops.scatter_add_rows(drows.contiguous(), index, dx)
        return dx, None, None


def masked_row_list(mask, rows=None):
    """Flat positions with mask == 1 as an int64 list - WITHOUT a device sync, or None when that is impossible.  The list is known
    on the host when the caller hands it over (`rows`), or when the mask is a host array / a device tensor that carries its host
    original (the reference's BatchOperator draws the mask in numpy: batch_operator.py:27-32).  A mask that exists on the device
    only would cost a torch.nonzero round trip per step: callers fall back to their dense path instead."""
    if rows is not None:
        return rows.reshape(-1)
    h = ops.host_mask(mask)
    if h is None:
        return None
    return torch.from_numpy(np.flatnonzero(np.asarray(h).reshape(-1) == 1).astype(np.int64))


class _HeadCEFn(torch.autograd.Function):
    """LinearHead + MaskedCrossEntropyLoss of a TRAINING step as one autograd node (masked_pretraining/model.py:60-61,78-82).
    Forward: exactly the two calls of the separate path - logits of EVERY position (the result dict carries them, like the
    reference's) and the masked mean CE.  Backward: with unmasked_weight None the dense dlogits has an exact zero row for every
    position outside the mask (~85 %), so the head's weight-gradient and input-gradient products, the bias gradient and the CE
    gradient itself run on the listed masked rows alone (gathered into whole 256-row tiles) and the token gradient is scattered
    into a zero matrix: same values (the dropped terms are products with exact zeros), no 2 GB zero fill, no column-sum pass over
    it.  The returned logits are marked non-differentiable: the loss is the step's only differentiable output."""

    @staticmethod
    def forward(ctx, tokens, weight, bias, labels, mask, index, dtype):
        x2 = tokens.detach().reshape(-1, tokens.shape[-1])
        if x2.dtype != dtype or not x2.is_contiguous():
            x2 = x2.to(dtype).contiguous()
        logits = F.linear_fwd(x2, weight, bias, dtype)
        lab = labels.reshape(-1).to(device=x2.device, dtype=torch.int64).contiguous()
        msk = mask.reshape(-1).to(device=x2.device, dtype=torch.int64).contiguous()
        loss, work = ops.masked_ce_fwd_rows(logits, lab, msk, index)     # the listed rows only: same bits as the all-rows form
        ctx.save_for_backward(x2, logits, lab, msk, work, index)
        ctx.weight, ctx.bias, ctx.dtype, ctx.shape = weight, bias, dtype, tokens.shape
        ctx.mark_non_differentiable(logits)
        # (without this autograd hands the backward a ZERO gradient of the logits' shape - a 2.1 GB fill per step at 1024 lines)
        ctx.set_materialize_grads(False)
        return logits, loss[0]

    @staticmethod
    def backward(ctx, _dlogits, dloss):
        if dloss is None:
            return None, None, None, None, None, None, None
        x2, logits, lab, msk, work, index = ctx.saved_tensors
        n = index.numel()
        n_pad = ((n + 255) // 256) * 256                                   # whole 256-row GEMM tiles; pad rows are zero
        dl = dloss.detach().reshape(1).to(torch.float32)                   # stays on the device: no host sync
        dlog = ops.masked_ce_bwd_rows(logits, lab, msk, work, index, n_pad, dloss=dl)
        xm = ops.gather_rows(x2, index, n_rows_out=n_pad)
        dxm = F.linear_bwd(dlog, xm, ctx.weight, ctx.bias, ctx.dtype, need_dx=ctx.needs_input_grad[0])
        dx = None
        if dxm is not None:
            dx = ops.zeros(x2.shape, x2.device, x2.dtype)
            ops.scatter_add_rows(dxm, index, dx)                            # reads the first n rows of dxm
            dx = dx.view(ctx.shape)
            if dxm.dtype == torch.bfloat16:
                F.set_row_grad_hint(dx, index, dxm, n)                      # the backbone's last layer may work from the listed rows (functional.layer_bwd_rows)
        return dx, None, None, None, None, None, None


class MaskedCrossEntropyLoss(torch.nn.Module):
    """masked_pretraining/model.py:72-95."""

    def __init__(self, unmasked_weight=None):
        super().__init__()
        self.unmasked_weight = unmasked_weight

    def forward(self, output, labels, mask):
        if not isinstance(mask, torch.Tensor):
            mask = torch.as_tensor(mask)
        return _MaskedCEFn.apply(output, labels, mask, self.unmasked_weight)


class MaskedTransformerEncoder(torch.nn.Module):
    """masked_pretraining/model.py:33-69."""

    # "all": the head is evaluated on every position, as the reference does (model.py:41-63).
    # "masked": in TRAINING steps whose loss reads the masked positions only (unmasked_weight None) the head, the loss and
    # their backward run on those rows alone (about 15 % of them: SURVEY.md section 8 a8/a9) - same loss and gradients,
    # result["output"] is None and result["output_rows"] / result["rows"] hold the logits of the masked positions.
    head_rows = "all"
    # "masked" (default): in those same training steps with head_rows == "all" the head's BACKWARD runs on the masked rows only
    # (_HeadCEFn; forward, result["output"] and every value unchanged).  "dense": backward over all positions through the separate
    # head / loss nodes (result["output"] then stays differentiable).  Needs the mask's row list without a device sync
    # (masked_row_list); otherwise the dense path runs.
    head_backward = "masked"
    # Debug switch for both row-list modes (costs one device sync per step): the list the step works from - the caller's `rows`, or the one
    # built from the host twin of an uploaded mask - is compared with the mask ON THE DEVICE; a mask edited in place after its upload, or a
    # caller's list that misses positions, raises instead of giving a loss over the wrong rows.
    check_masked_rows = False

    def _check_rows(self, mask, index):
        m = torch.as_tensor(mask)
        want = torch.nonzero(m.reshape(-1).to(index.device) == 1, as_tuple=False).reshape(-1)
        if want.numel() != index.numel() or not torch.equal(want, index.reshape(-1).to(want.dtype)):
            raise ValueError(f"masked row list ({index.numel()} rows) does not match the mask on the device ({want.numel()} positions with mask == 1): "
                             "the mask was modified after its upload, or `rows` is not its list of masked positions")

    def __init__(self, backbone, head, loss=None):
        super().__init__()
        self.backbone = backbone
        self.head = head
        self.loss = MaskedCrossEntropyLoss() if loss is None else loss

    def _forward_masked_rows(self, x, labels, mask, rows=None):
        index = masked_row_list(mask, rows)   # caller's list, or from the mask's host original: never a device sync
        if index is None:
            return None  # a device-only mask: the dense path (a torch.nonzero round trip per step would cost more than it saves)
        n = index.numel()
        if n == 0:
            return None  # the dense path reproduces the reference's NaN for an empty selection
        tokens = self.backbone.encode_tokens(x, mask)                # (N*S, d)
        index = index.to(tokens.device)
        if self.check_masked_rows:
            self._check_rows(mask, index)
        n_pad = ((n + 255) // 256) * 256                              # whole 256-row GEMM tiles; pad rows are zero
        rows = _GatherTokensFn.apply(tokens, index, n_pad)
        logits = self.head(rows)                                      # (n_pad, V)
        row_labels = torch.zeros(n_pad, device=tokens.device, dtype=torch.int64)
        row_labels[:n] = torch.as_tensor(labels).to(tokens.device).reshape(-1)[index]
        row_mask = torch.zeros(n_pad, device=tokens.device, dtype=torch.int64)
        row_mask[:n] = 1
        loss = self.loss(logits.view(1, n_pad, -1), row_labels.view(1, n_pad), row_mask.view(1, n_pad))
        return {"output": None, "loss": loss, "output_rows": logits[:n], "rows": index}

    def forward(self, x, labels=None, mask=None, rows=None):
        if self.head_rows == "masked" and self.training and labels is not None and mask is not None \
                and getattr(self.loss, "unmasked_weight", None) is None:
            result = self._forward_masked_rows(x, labels, mask, rows)
            if result is not None:
                return result
        elif self.head_rows not in ("all", "masked"):
            raise ValueError(f"Unknown head_rows: {self.head_rows}")
        if self.head_backward not in ("masked", "dense"):
            raise ValueError(f"Unknown head_backward: {self.head_backward}")
        if self.head_backward == "masked" and self.training and torch.is_grad_enabled() and labels is not None and mask is not None \
                and type(self.loss) is MaskedCrossEntropyLoss and self.loss.unmasked_weight is None and isinstance(self.head, LinearHead):
            index = masked_row_list(mask, rows)
            if index is not None and index.numel() > 0:
                n = x.shape[0]
                tokens = self.backbone.encode_tokens(x, mask)
                if not isinstance(mask, torch.Tensor):
                    mask = torch.from_numpy(np.asarray(mask))
                if self.check_masked_rows:
                    self._check_rows(mask, index.to(tokens.device))
                logits, loss = _HeadCEFn.apply(tokens, self.head.linear.weight, self.head.linear.bias, torch.as_tensor(labels), mask,
                                               index.to(tokens.device, non_blocking=True), compute_dtype())
                return {"output": logits.view(n, -1, logits.shape[-1]), "loss": loss}
        output = self.encode(x, mask)
        if mask is not None and not isinstance(mask, torch.Tensor):
            mask = torch.from_numpy(mask).to(output.device)
        loss = None
        if mask is not None and labels is not None:
            loss = self.loss(output, labels, mask)
        return {"output": output, "loss": loss}

    def encode(self, images, mask=None):
        n = images.shape[0]
        tokens = self.backbone.encode_tokens(images, mask)          # (N*S, d) row-major
        return self.head(tokens.view(n, -1, tokens.shape[-1]))     # == head(rearrange(backbone(x), 'n c w -> n w c'))

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        """masked_pretraining/model.py:68-69; the file is a plain state_dict (reference format) and is read with the
        loader that executes nothing from it."""
        device = next(self.parameters()).device
        self.load_state_dict(torch.load(path, map_location=device, weights_only=True))
