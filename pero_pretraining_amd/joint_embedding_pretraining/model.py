"""joint_embedding_pretraining/model.py of the reference on the HIP kernels: init_backbone, init_head,
JointEmbeddingTransformerEncoder, LinearHead, MLPHead."""
import torch

from .. import functional as F
from .. import ops
from ..masked_pretraining.model import LinearHead as _LinearHead
from ..masked_pretraining.model import linear
from ..models.transformers import VisionTransformerEncoder
from ..precision import compute_dtype


def init_backbone(backbone_definition):
    """The reference ignores the definition and always builds the default 6-layer d=512 ViT
    (joint_embedding_pretraining/model.py:10-13); here the definition is honoured - with an empty / type-only
    definition the result is identical to the reference's."""
    backbone_type = backbone_definition.get("type", "vit")
    if backbone_type == "vit":
        return VisionTransformerEncoder(**backbone_definition)
    if backbone_type == "vggt":
        raise NotImplementedError("vggt (VGG convolutional front end) is outside the HIP hot path (SURVEY.md section 2)")
    raise ValueError(f"Unknown backbone type: {backbone_type}")


def init_head(head_definition):
    head_type = head_definition.get("type", "linear")
    kwargs = {k: v for k, v in head_definition.items() if k != "type"}
    if head_type == "linear":
        return LinearHead(**kwargs)
    if head_type == "mlp":
        return MLPHead(**kwargs)
    raise ValueError(f"Unknown head type: {head_type}")


class LinearHead(_LinearHead):
    pass


class _BatchNormReluFn(torch.autograd.Function):
    """torch.nn.BatchNorm1d followed by ReLU on (rows, d) rows as one node on pero_bn_fwd / pero_bn_bwd (csrc/bnorm.hip): batch statistics in
    training (running statistics updated in place, momentum and the unbiased running variance as torch does), running statistics in eval."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, dtype):
        x2 = x.detach()
        if x2.dtype != dtype or not x2.is_contiguous():
            x2 = x2.to(dtype).contiguous()
        training = bn.training or bn.running_mean is None
        momentum = 0.1 if bn.momentum is None else bn.momentum
        if training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
            if bn.momentum is None:   # torch: cumulative moving average
                momentum = 1.0 / float(bn.num_batches_tracked)
        y, mean, rstd = ops.bn_fwd(x2, weight.detach(), bias.detach(), bn.running_mean if bn.track_running_stats else None,
                                   bn.running_var if bn.track_running_stats else None, bn.eps, momentum, training, True)
        ctx.save_for_backward(x2, y, mean, rstd)
        ctx.weight, ctx.bias, ctx.training = weight, bias, training
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, y, mean, rstd = ctx.saved_tensors
        g = dy.detach()
        if g.dtype != x2.dtype or not g.is_contiguous():
            g = g.to(x2.dtype).contiguous()
        if not ctx.training:
            raise RuntimeError("BatchNorm backward in evaluation mode is not implemented (training-mode statistics only)")
        dw = F.ensure_grad(ctx.weight) if ctx.weight.requires_grad else None
        db = F.ensure_grad(ctx.bias) if ctx.bias.requires_grad else None
        dx = ops.bn_bwd(g, x2, y, ctx.weight.detach(), mean, rstd, dw, db, True)
        return dx, None, None, None, None


class MLPHead(torch.nn.Module):
    """Linear(in,h) [BatchNorm1d(h)] ReLU [Linear(h,h) [BatchNorm1d(h)] ReLU]* Linear(h,h); parameters and buffers under `layers.{i}` exactly as
    the reference's torch.nn.Sequential (joint_embedding_pretraining/model.py:79-115).  Without BatchNorm the ReLUs are fused into the GEMM
    epilogues; with `use_bn=True` (round 4; reference default False) BatchNorm + ReLU run as one HIP node between the products.  BatchNorm
    statistics are those of the rows THIS process sees: under data-parallel training they are per rank, exactly like torch.nn.BatchNorm1d
    under DistributedDataParallel without SyncBatchNorm (gradients are still averaged; the running statistics differ slightly between ranks)."""

    def __init__(self, in_dim=512, hidden_dim=8192, num_layers=3, use_bn=False):
        super().__init__()
        self.in_dim, self.hidden_dim, self.num_layers, self.use_bn = in_dim, hidden_dim, num_layers, use_bn
        layers, d = [], in_dim
        for _ in range(num_layers - 1):
            layers.append(torch.nn.Linear(d, hidden_dim))
            d = hidden_dim
            if use_bn:
                layers.append(torch.nn.BatchNorm1d(hidden_dim))
            layers.append(torch.nn.ReLU())
        layers.append(torch.nn.Linear(d, hidden_dim))
        self.layers = torch.nn.Sequential(*layers)

    def forward(self, x):
        N, S, D = x.shape
        y = x.reshape(N * S, D)
        mods = list(self.layers)
        lin = [m for m in mods if isinstance(m, torch.nn.Linear)]
        if not self.use_bn:
            for i, m in enumerate(lin):
                y = linear(y, m.weight, m.bias, relu_out=i < len(lin) - 1, gate_in=i > 0)
            return y.reshape(N, S, -1)
        if not y.is_cuda:
            raise RuntimeError("pero_pretraining_amd layers run on the GPU only (HIP kernels, no CPU fallback)")
        bns = [m for m in mods if isinstance(m, torch.nn.BatchNorm1d)]
        for i, m in enumerate(lin):
            y = linear(y, m.weight, m.bias)
            if i < len(lin) - 1:
                y = _BatchNormReluFn.apply(y, bns[i].weight, bns[i].bias, bns[i], compute_dtype())
        return y.reshape(N, S, -1)


BATCH_VIEWS = True  # encode the two views as one 2N-line batch


class JointEmbeddingTransformerEncoder(torch.nn.Module):
    """joint_embedding_pretraining/model.py:33-66."""

    def __init__(self, backbone, head, loss):
        super().__init__()
        self.backbone, self.head, self.loss = backbone, head, loss

    def forward(self, images1, images2, image_masks1, image_masks2, shift_masks1, shift_masks2):
        if BATCH_VIEWS and images1.shape == images2.shape and images1.dtype == images2.dtype and images1.is_cuda:
            # both views through one pass of 2N lines; same weights, same per-view RNG draws
            n = images1.shape[0]
            tokens = self.backbone.encode_tokens_views([images1, images2])
            out = self.head(tokens.view(2 * n, -1, tokens.shape[-1]))
            if hasattr(self.loss, "forward_stacked"):
                # the loss reads both views from the ONE tensor the head wrote and returns one gradient for it (slicing them apart made
                # autograd fill, copy and add three tensors of the output's size per step)
                loss = self.loss.forward_stacked(out, image_masks1, image_masks2, shift_masks1, shift_masks2)
                return {"output1": out[:n].detach(), "output2": out[n:].detach(), **loss}
            output1, output2 = out[:n], out[n:]
        else:
            output1 = self.encode(images1)
            output2 = self.encode(images2)
        loss = self.loss(output1, output2, image_masks1, image_masks2, shift_masks1, shift_masks2)
        return {"output1": output1.detach(), "output2": output2.detach(), **loss}

    def encode(self, images):
        n = images.shape[0]
        tokens = self.backbone.encode_tokens(images, None)
        return self.head(tokens.view(n, -1, tokens.shape[-1]))

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        device = next(self.parameters()).device
        self.load_state_dict(torch.load(path, map_location=device, weights_only=True))
