"""joint_embedding_pretraining/model.py of the reference on the HIP kernels: init_backbone, init_head,
JointEmbeddingTransformerEncoder, LinearHead, MLPHead."""
import torch

from ..masked_pretraining.model import LinearHead as _LinearHead
from ..masked_pretraining.model import linear
from ..models.transformers import VisionTransformerEncoder


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


class MLPHead(torch.nn.Module):
    """Linear(in,h) ReLU [Linear(h,h) ReLU]* Linear(h,h); parameters under `layers.{i}` exactly as the reference's
    torch.nn.Sequential (joint_embedding_pretraining/model.py:79-115).  ReLUs are fused into the GEMM epilogues."""

    def __init__(self, in_dim=512, hidden_dim=8192, num_layers=3, use_bn=False):
        super().__init__()
        if use_bn:
            raise NotImplementedError("MLPHead(use_bn=True) is not implemented in the HIP path (reference default: False)")
        self.in_dim, self.hidden_dim, self.num_layers, self.use_bn = in_dim, hidden_dim, num_layers, use_bn
        layers, d = [], in_dim
        for _ in range(num_layers - 1):
            layers += [torch.nn.Linear(d, hidden_dim), torch.nn.ReLU()]
            d = hidden_dim
        layers.append(torch.nn.Linear(d, hidden_dim))
        self.layers = torch.nn.Sequential(*layers)

    def forward(self, x):
        N, S, D = x.shape
        y = x.reshape(N * S, D)
        lin = [m for m in self.layers if isinstance(m, torch.nn.Linear)]
        for i, m in enumerate(lin):
            y = linear(y, m.weight, m.bias, relu_out=i < len(lin) - 1, gate_in=i > 0)
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
