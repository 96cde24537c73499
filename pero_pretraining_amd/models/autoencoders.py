"""VectorQuantizer with the interface of the reference's models/autoencoders.py:170-241 - the tokenizer step that
produces the labels masked pre-training predicts (config 3: codebook 8192 x 512).  Nearest-code search by the fused
exact-f32 HIP kernel (distances and one-hot matrices are never materialised), quantized output with the reference's
straight-through arithmetic and gradient; in training mode with decay > 0 the EMA codebook update
(autoencoders.py:225-237) runs as scatter kernels instead of two one-hot GEMMs.  The VGG encoder / decoder around the
quantizer are outside the hot path."""
import torch

from .. import ops


class VectorQuantizer(torch.nn.Module):
    def __init__(self, num_embeddings, embeddings_dim, commitment_cost, decay, epsilon=1e-5):
        super().__init__()
        self.embeddings_dim = embeddings_dim
        self.num_embeddings = num_embeddings
        self.embedding = torch.nn.Embedding(self.num_embeddings, self.embeddings_dim)
        if decay > 0.0:  # same parameter / buffer set and the same RNG draws as the reference constructor
            self.embedding.weight.data.normal_()
            self.register_buffer("ema_cluster_size", torch.zeros(num_embeddings))
            self.ema_w = torch.nn.Parameter(torch.Tensor(num_embeddings, self.embeddings_dim))
            self.ema_w.data.normal_()
        else:
            self.embedding.weight.data.uniform_(-1 / self.num_embeddings, 1 / self.num_embeddings)
        self.commitment_cost = commitment_cost
        self.decay = decay
        self.epsilon = epsilon

    def calculate_loss(self, tokens, features):
        q_latent_loss = 0 if self.decay > 0.0 else torch.nn.functional.mse_loss(tokens, features.detach())
        e_latent_loss = torch.nn.functional.mse_loss(tokens.detach(), features)
        return q_latent_loss + self.commitment_cost * e_latent_loss

    @torch.no_grad()
    def nearest(self, flat_input):
        """(M, D) float32 rows -> int64 indices of the nearest code (first minimum), bit-exact f32 arithmetic."""
        if not flat_input.is_cuda:
            raise RuntimeError("pero_pretraining_amd VectorQuantizer runs on the GPU only (HIP kernel, no CPU fallback)")
        return ops.vq_argmin(flat_input.float().contiguous(), self.embedding.weight.detach().float().contiguous())

    def forward(self, inputs):
        """inputs (N, D, 1, T) -> (quantized (N, D, 1, T), indices (N*T,)) like autoencoders.py:204-241.
        The gradient of `quantized` flows straight through to `inputs` (autoencoders.py:239).  Training mode with
        decay > 0 updates ema_cluster_size / ema_w / embedding.weight IN PLACE after quantizing with the old codebook
        (the reference re-creates the two nn.Parameters every step, which detaches them from any optimizer and from
        DDP - not reproduced; values are the same)."""
        x = inputs.permute(0, 2, 3, 1).contiguous()
        flat = x.reshape(-1, self.embeddings_dim).float()
        q, idx = _QuantizeFn.apply(flat, self)
        return q.view(x.shape).permute(0, 3, 1, 2).contiguous(), idx


class _QuantizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, vq):
        flat = flat.detach().contiguous()
        weight = vq.embedding.weight.detach()
        idx = vq.nearest(flat)
        q = ops.vq_gather(flat, weight.float().contiguous(), idx)
        if vq.training and vq.decay > 0.0:
            ops.vq_ema_update(flat, idx, vq.ema_cluster_size, vq.ema_w.data, vq.embedding.weight.data, vq.decay, vq.epsilon)
        ctx.mark_non_differentiable(idx)
        return q, idx

    @staticmethod
    def backward(ctx, dq, _didx):
        return dq, None


class _Project1x1Fn(torch.autograd.Function):
    """Conv2d(C_in -> C_out, kernel 1) on (N, C_in, 1, T) features as ONE exact-f32 pero_gemm over the token rows (N*T, C_in): the
    quantizer behind it decides by f32 distances, so the projection keeps f32 operands whatever the autocast mode."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        n, c, h, t = x.shape
        rows = x.detach().permute(0, 2, 3, 1).reshape(-1, c).float().contiguous()
        w2 = weight.detach().reshape(weight.shape[0], c).float().contiguous()
        y = ops.gemm(rows, w2, bias=None if bias is None else bias.detach().float().contiguous())
        ctx.save_for_backward(rows, w2)
        ctx.shape, ctx.has_bias = x.shape, bias is not None
        return y.view(n, h, t, -1).permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def backward(ctx, dy):
        rows, w2 = ctx.saved_tensors
        n, c, h, t = ctx.shape
        d2 = dy.permute(0, 2, 3, 1).reshape(-1, dy.shape[1]).float().contiguous()
        dx = ops.gemm(d2, w2, trans_b=True).view(n, h, t, c).permute(0, 3, 1, 2).contiguous()
        dw = ops.gemm(d2, rows, trans_a=True, trans_b=True).view(w2.shape[0], c, 1, 1)
        db = d2.sum(0) if ctx.has_bias else None
        return dx, dw, db


class VQVAEQuantizer(torch.nn.Module):
    """The quantizing middle of the reference's VQVAE (models/autoencoders.py:107-146): `encoder_projection_layer` (1x1 conv,
    encoder channels -> embeddings_dim), `vq`, `decoder_projection_layer` (1x1 conv, embeddings_dim -> decoder channels), with the
    reference's parameter names - a VQVAE checkpoint's `encoder_projection_layer.*`, `vq.*`, `decoder_projection_layer.*` entries
    load with `load_state_dict(..., strict=False)` on the full file or strictly on the filtered one.  `quantize(x)` is the
    reference's method: encoder features (N, C_enc, 1, T) -> (projected tokens (N, C_dec, 1, T), labels (N*T,)); the labels are
    what masked pre-training predicts (scripts/produce_vqvae_labels.py:27-46).  The VGG encoder / decoder around it are outside the
    hot path."""

    def __init__(self, encoder_channels, decoder_channels, num_embeddings, embeddings_dim, commitment_cost=0.25, decay=0.99):
        super().__init__()
        self.encoder_projection_layer = torch.nn.Conv2d(encoder_channels, embeddings_dim, 1)     # parameter containers (same init / keys)
        self.decoder_projection_layer = torch.nn.Conv2d(embeddings_dim, decoder_channels, 1)
        self.num_embeddings = num_embeddings
        self.embeddings_dim = embeddings_dim
        self.vq = VectorQuantizer(num_embeddings, embeddings_dim, commitment_cost, decay)

    def quantize(self, x):
        if not x.is_cuda:
            raise RuntimeError("pero_pretraining_amd VQVAEQuantizer runs on the GPU only (HIP kernels, no CPU fallback)")
        x = _Project1x1Fn.apply(x, self.encoder_projection_layer.weight, self.encoder_projection_layer.bias)
        tokens, labels = self.vq(x)
        projected_tokens = _Project1x1Fn.apply(tokens, self.decoder_projection_layer.weight, self.decoder_projection_layer.bias)
        return projected_tokens, labels

    def labels(self, x):
        """Labels only (the label-production path: no decoder projection)."""
        with torch.no_grad():
            x = _Project1x1Fn.apply(x, self.encoder_projection_layer.weight, self.encoder_projection_layer.bias)
            return self.vq(x)[1]


def kmeans_labels(features, centroids):
    """Feature-Quantization labels (scripts/produce_kmeans_labels.py:72-79): features (N, F, T), centroids (K, F)
    -> (N, T) int64 assignments.  argmin of the true L2 distance == argmin of the squared expanded form used by the
    kernel (up to exact ties)."""
    n, f, t = features.shape
    flat = features.permute(0, 2, 1).reshape(-1, f).float().contiguous()
    return ops.vq_argmin(flat, centroids.float().contiguous()).view(n, t)
