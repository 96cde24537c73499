"""CPU oracle for the pero-pretraining hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in primitive tensor arithmetic (matmul / exp / sum / sqrt, no torch.nn
layers, no fused library ops), the algorithm of the reference's masked / joint-embedding
pre-training step.  It is the *checker* for the HIP kernels: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.  The product
package (`pero_pretraining_amd`) never does, and fails loudly without its HIP library.

Pinning: the reference has no tests or golden vectors of its own, all of its
arithmetic lives in an un-pinned `torch`.  The oracle is therefore pinned against outputs of the
reference itself, run in the build container by `oracle/make_golden.py` (which imports
/root/reference) and committed as small `.npz` fixtures under `tests/golden/`;
`tests/test_oracle_golden.py` checks every function here against them.

All functions take/return torch CPU tensors (float32 by default; pass float64 inputs for a
higher-precision check) so that `torch.autograd` can supply the reference gradients of the same
restated arithmetic.  Integer outputs (quantizer indices) are int64.

Reference citations are relative to /root/reference/pero_pretraining/.
"""
import math

import numpy as np
import torch


# --------------------------------------------------------------------------------------------
# constants / tables
# --------------------------------------------------------------------------------------------
def mask_tile(in_channels=3, patch_h=40, patch_w=8):
    """Fixed noise tile that overwrites masked patches.  models/transformers.py:29-34
    (np.random.seed(42); np.random.rand(1, C, ph, pw)) -> float32 (C, ph, pw)."""
    rs = np.random.RandomState(42)  # same MT19937 stream as np.random.seed(42); np.random.rand
    tile = rs.rand(1, in_channels, patch_h, patch_w)
    return torch.tensor(tile, dtype=torch.float32)[0]


def positional_table(d_model, max_len=4096):
    """Sinusoid table pe[pos, 2i] = sin(pos * w_i), pe[pos, 2i+1] = cos(pos * w_i),
    w_i = exp(-2i ln(1e4) / d).  models/transformers.py:164-170."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def warmup_lr(iteration, base_lr, warm_up_iterations, order):
    """common/lr_scheduler.py:14-24."""
    if warm_up_iterations is not None and order is not None and \
            iteration <= warm_up_iterations and warm_up_iterations > 0:
        return ((iteration / warm_up_iterations) ** order) * base_lr
    return base_lr


# --------------------------------------------------------------------------------------------
# front end: batch preparation, masking, patch embedding
# --------------------------------------------------------------------------------------------
def prepare_images(images_u8_nhwc):
    """uint8 (N,H,W,C) -> float32 (N,C,H,W) / 255.  masked_pretraining/batch_operator.py:17-20."""
    return images_u8_nhwc.to(torch.float32).permute(0, 3, 1, 2) / 255.0


def apply_mask(x_nchw, mask, tile, patch_w=8):
    """Overwrite every masked patch_w-pixel column group with the noise tile (the tile has
    period patch_w along W).  models/transformers.py:53-68.  Returns a new tensor."""
    n, c, h, w = x_nchw.shape
    m = torch.as_tensor(mask).to(torch.int64)
    m = m[:, None, None, :].expand(n, c, h, w // patch_w).repeat_interleave(patch_w, dim=3)
    pattern = tile[None].repeat(n, 1, 1, w // patch_w)
    return torch.where(m == 1, pattern.to(x_nchw.dtype), x_nchw)


def patches(x_nchw, patch_w=8):
    """(N,C,H,W) -> (N*S, C*H*patch_w): the rows a Conv2d(kernel=stride=(H,patch_w)) contracts,
    ordered (c, h, p) like conv weight.reshape(d, -1).  models/transformers.py:99-107."""
    n, c, h, w = x_nchw.shape
    s = w // patch_w
    return x_nchw.reshape(n, c, h, s, patch_w).permute(0, 3, 1, 2, 4).reshape(n * s, c * h * patch_w)


def patch_embed(x_nchw, conv_w, conv_b, patch_w=8):
    """Conv2d(C->d, kernel=stride=(H, patch_w)) as a GEMM; returns (N*S, d) token rows."""
    d = conv_w.shape[0]
    return patches(x_nchw, patch_w) @ conv_w.reshape(d, -1).t() + conv_b


def layer_norm(x, w, b, eps=1e-5):
    """torch.nn.LayerNorm over the last dim (biased variance).  models/transformers.py:28."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def add_positional(t, n, s, pe, offsets=None):
    """t: (N*S, d) rows ordered (line, position).  eval: + pe[0:S]; train with random_shift:
    line i gets pe[offset_i : offset_i + S].  models/transformers.py:174-188."""
    d = t.shape[-1]
    t = t.reshape(n, s, d)
    if offsets is None:
        return (t + pe[:s].to(t.dtype)[None]).reshape(n * s, d)
    idx = torch.as_tensor(offsets, dtype=torch.int64)[:, None] + torch.arange(s)[None, :]
    return (t + pe.to(t.dtype)[idx]).reshape(n * s, d)


# --------------------------------------------------------------------------------------------
# transformer encoder layer (torch.nn.TransformerEncoderLayer, post-norm, ReLU, no masks)
# --------------------------------------------------------------------------------------------
def attention(qkv, n, s, num_heads):
    """qkv (N*S, 3d) with column blocks [q | k | v], heads = contiguous hd-slices.
    softmax(q k^T / sqrt(hd)) v over all S keys of the same line.  models/transformers.py:37-43,86
    (torch.nn.MultiheadAttention semantics)."""
    d = qkv.shape[1] // 3
    hd = d // num_heads
    q, k, v = qkv.reshape(n, s, 3, num_heads, hd).permute(2, 0, 3, 1, 4)  # (N,h,S,hd) each
    scores = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    scores = scores - scores.max(dim=-1, keepdim=True).values
    p = torch.exp(scores)
    p = p / p.sum(dim=-1, keepdim=True)
    o = p @ v  # (N,h,S,hd)
    return o.permute(0, 2, 1, 3).reshape(n * s, d)


def encoder_layer(t, p, n, s, num_heads):
    """One post-norm layer on token rows t (N*S, d).  p: dict with in_proj_weight/bias,
    out_proj_weight/bias, linear1_weight/bias, linear2_weight/bias, norm1_weight/bias,
    norm2_weight/bias."""
    qkv = t @ p["in_proj_weight"].t() + p["in_proj_bias"]
    a = attention(qkv, n, s, num_heads)
    t = layer_norm(t + a @ p["out_proj_weight"].t() + p["out_proj_bias"], p["norm1_weight"], p["norm1_bias"])
    hdn = torch.relu(t @ p["linear1_weight"].t() + p["linear1_bias"])
    t = layer_norm(t + hdn @ p["linear2_weight"].t() + p["linear2_bias"], p["norm2_weight"], p["norm2_bias"])
    return t


def layer_params(sd, i, prefix="backbone.encoder_layers.layers."):
    """Pick layer i's tensors out of a reference-format state dict."""
    b = f"{prefix}{i}."
    return {
        "in_proj_weight": sd[b + "self_attn.in_proj_weight"], "in_proj_bias": sd[b + "self_attn.in_proj_bias"],
        "out_proj_weight": sd[b + "self_attn.out_proj.weight"], "out_proj_bias": sd[b + "self_attn.out_proj.bias"],
        "linear1_weight": sd[b + "linear1.weight"], "linear1_bias": sd[b + "linear1.bias"],
        "linear2_weight": sd[b + "linear2.weight"], "linear2_bias": sd[b + "linear2.bias"],
        "norm1_weight": sd[b + "norm1.weight"], "norm1_bias": sd[b + "norm1.bias"],
        "norm2_weight": sd[b + "norm2.weight"], "norm2_bias": sd[b + "norm2.bias"],
    }


def backbone_tokens(sd, x_nchw, num_heads, mask=None, offsets=None, max_len=4096, prefix="backbone."):
    """VisionTransformerEncoder.forward on float NCHW input -> token rows (N*S, d) (the reference
    returns the same numbers as (N, d, S)).  models/transformers.py:45-51,71-89."""
    n, c, h, w = x_nchw.shape
    conv_w = sd[prefix + "conv_layer.weight"]
    pw = conv_w.shape[-1]
    s = w // pw
    if mask is not None:
        x_nchw = apply_mask(x_nchw, mask, mask_tile(c, conv_w.shape[-2], pw).to(x_nchw.dtype), pw)
    t = patch_embed(x_nchw, conv_w, sd[prefix + "conv_layer.bias"], pw)
    t = layer_norm(t, sd[prefix + "intermediate_norm.weight"], sd[prefix + "intermediate_norm.bias"])
    t = add_positional(t, n, s, positional_table(t.shape[-1], max_len), offsets)
    i = 0
    while f"{prefix}encoder_layers.layers.{i}.linear1.weight" in sd:
        t = encoder_layer(t, layer_params(sd, i, prefix + "encoder_layers.layers."), n, s, num_heads)
        i += 1
    return t


# --------------------------------------------------------------------------------------------
# masked pre-training head + loss
# --------------------------------------------------------------------------------------------
def cross_entropy_rows(logits, labels):
    """mean over rows of (logsumexp(logits) - logits[label])."""
    m = logits.max(dim=-1, keepdim=True).values
    lse = torch.log(torch.exp(logits - m).sum(dim=-1)) + m[:, 0]
    picked = logits.gather(1, labels[:, None])[:, 0]
    return (lse - picked).mean()


def masked_cross_entropy(output, labels, mask, unmasked_weight=None):
    """masked_pretraining/model.py:78-95.  output (N,S,V); labels (N,S) int64 (-1 = padding);
    mask (N,S) in {0,1}."""
    mask = torch.as_tensor(mask)
    loss = cross_entropy_rows(output[mask == 1], labels[mask == 1])
    if unmasked_weight is not None:
        sel = (mask == 0) & (labels >= 0)
        loss = loss + unmasked_weight * cross_entropy_rows(output[sel], labels[sel])
    return loss


def masked_model_forward(sd, images_nchw, labels, mask, num_heads, offsets=None, max_len=4096,
                         unmasked_weight=None):
    """MaskedTransformerEncoder.forward.  masked_pretraining/model.py:41-63.
    Returns (output (N,S,V), loss or None)."""
    n, _, _, w = images_nchw.shape
    t = backbone_tokens(sd, images_nchw, num_heads, mask, offsets, max_len)
    s = t.shape[0] // n
    out = (t @ sd["head.linear.weight"].t() + sd["head.linear.bias"]).reshape(n, s, -1)
    loss = None
    if mask is not None and labels is not None:
        loss = masked_cross_entropy(out, labels, mask, unmasked_weight)
    return out, loss


def adam_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (no weight decay, no amsgrad), in place; `step` is 1-based.
    masked_pretraining/train.py:146."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------------------------
# quantizers (integer outputs: graded bit-exact)
# --------------------------------------------------------------------------------------------
def vq_nearest(flat_input, codebook):
    """VectorQuantizer.forward's index computation, models/autoencoders.py:212-217:
    dist = sum(x^2, 1, keepdim) + sum(e^2, 1) - 2 x e^T ; argmin over codes (first minimum).
    float32 numpy arithmetic.  Returns (indices int64 (M,), distances float32 (M,K))."""
    x = np.asarray(flat_input, dtype=np.float32)
    e = np.asarray(codebook, dtype=np.float32)
    sx = np.sum(x * x, axis=1, keepdims=True, dtype=np.float32)
    se = np.sum(e * e, axis=1, dtype=np.float32)
    dist = (sx + se) - np.float32(2.0) * (x @ e.T)
    return np.argmin(dist, axis=1).astype(np.int64), dist


def vq_quantize(inputs_ncht, codebook):
    """Eval-mode VectorQuantizer.forward: (N,D,1,T) -> (quantized (N,D,1,T), indices (N*T,)).
    models/autoencoders.py:204-241 (training-mode EMA branch: vq_ema_update)."""
    x = np.asarray(inputs_ncht, dtype=np.float32)
    n, d, one, t = x.shape
    flat = np.ascontiguousarray(x.transpose(0, 2, 3, 1)).reshape(-1, d)
    idx, _ = vq_nearest(flat, codebook)
    q = np.asarray(codebook, dtype=np.float32)[idx]
    q = flat + (q - flat)  # straight-through estimator arithmetic, autoencoders.py:239 (rounds in fp32)
    q = q.reshape(n, one, t, d).transpose(0, 3, 1, 2)
    return np.ascontiguousarray(q), idx


def vq_ema_update(flat_input, indices, ema_cluster_size, ema_w, decay=0.99, epsilon=1e-5):
    """models/autoencoders.py:225-237 without the one-hot matrix, float32 numpy: returns (ema_cluster_size, ema_w, codebook)."""
    f32 = np.float32
    x = np.asarray(flat_input, dtype=f32)
    ema_w = np.asarray(ema_w, dtype=f32)
    K, D = ema_w.shape
    idx = np.asarray(indices, dtype=np.int64)
    counts = np.bincount(idx, minlength=K).astype(f32)                       # == torch.sum(encodings, 0)
    cluster = np.asarray(ema_cluster_size, dtype=f32) * f32(decay) + f32(1 - decay) * counts
    n = np.sum(cluster, dtype=f32)
    cluster = (cluster + f32(epsilon)) / (n + f32(K * epsilon)) * n
    dw = np.zeros((K, D), dtype=f32)
    np.add.at(dw, idx, x)                                                    # == encodings.t() @ flat_input
    ema_w = ema_w * f32(decay) + f32(1 - decay) * dw
    return cluster.astype(f32), ema_w.astype(f32), (ema_w / cluster[:, None]).astype(f32)


def kmeans_assign(features, centroids):
    """Feature-Quantization labels: argmin of the true L2 distance (torch.cdist) between
    (M,F) features and (K,F) centroids.  scripts/produce_kmeans_labels.py:72-76."""
    x = np.asarray(features, dtype=np.float32)
    c = np.asarray(centroids, dtype=np.float32)
    diff = x[:, None, :] - c[None, :, :]
    dist = np.sqrt(np.sum(diff * diff, axis=2, dtype=np.float32))
    return np.argmin(dist, axis=1).astype(np.int64), dist


def margins(dist):
    """(best, second best) distance per row - used to flag near-ties in index parity tests."""
    part = np.partition(dist, 1, axis=1)
    return part[:, 0], part[:, 1]


# --------------------------------------------------------------------------------------------
# joint-embedding losses
# --------------------------------------------------------------------------------------------
def vicreg_loss(x, y, image_masks1, image_masks2, shift_masks1, shift_masks2,
                variance_weight=1.0, invariance_weight=1.0, covariance_weight=1.0,
                variance_threshold=1.0, eps=1e-5):
    """joint_embedding_pretraining/losses.py:13-47.  x, y (N,S,D); masks (N,S) in {0,1,2}."""
    im1, im2 = torch.as_tensor(image_masks1), torch.as_tensor(image_masks2)
    sm1, sm2 = torch.as_tensor(shift_masks1), torch.as_tensor(shift_masks2)
    inv_x, inv_y = x[sm1 == 1], y[sm2 == 1]
    diff = inv_x - inv_y
    invariance = (diff * diff).mean()
    z = torch.cat([x[im1 == 1], y[im2 == 1]], dim=0)
    m, d = z.shape
    mu = z.mean(dim=0)
    zc = z - mu
    var = (zc * zc).sum(dim=0) / (m - 1)
    variance = torch.relu(variance_threshold - torch.sqrt(var + eps)).mean()
    cov = (zc.t() @ zc) / (m - 1)
    off = cov - torch.diag(torch.diag(cov))
    covariance = (off * off).sum() / d
    loss = variance_weight * variance + invariance_weight * invariance + covariance_weight * covariance
    return {"loss": loss, "loss.variance": variance, "loss.invariance": invariance,
            "loss.covariance": covariance}


def mlp_head(x, sd, use_bn=False, training=True, momentum=0.1, eps=1e-5):
    """joint_embedding_pretraining/model.py:79-115 restated: MLPHead on (N, S, D) -> Linear [BatchNorm1d] ReLU ... Linear over the (N*S, D)
    rows.  sd: the head's state dict (`layers.{i}.*`, torch tensors).  BatchNorm1d (torch.nn.BatchNorm1d semantics, model.py:99-100): in
    training the batch mean and BIASED variance normalise, the running statistics move by `momentum` towards the batch mean and the
    UNBIASED batch variance; in evaluation the running statistics normalise.  Returns (y (N, S, hidden), {buffer name: new value})."""
    n, s_, d = x.shape
    y = x.reshape(n * s_, d)
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("layers.")})
    lin = [i for i in idx if sd[f"layers.{i}.weight"].dim() == 2]
    new_buffers = {}
    for j, i in enumerate(lin):
        y = y @ sd[f"layers.{i}.weight"].t() + sd[f"layers.{i}.bias"]
        if j == len(lin) - 1:
            break
        if use_bn:
            b = i + 1
            if training:
                m = y.shape[0]
                mean = y.sum(dim=0) / m
                cen = y - mean
                var = (cen * cen).sum(dim=0) / m
                new_buffers[f"layers.{b}.running_mean"] = (1 - momentum) * sd[f"layers.{b}.running_mean"] + momentum * mean.detach()
                new_buffers[f"layers.{b}.running_var"] = (1 - momentum) * sd[f"layers.{b}.running_var"] + momentum * (var.detach() * m / (m - 1))
                new_buffers[f"layers.{b}.num_batches_tracked"] = sd[f"layers.{b}.num_batches_tracked"] + 1
            else:
                mean, var = sd[f"layers.{b}.running_mean"], sd[f"layers.{b}.running_var"]
                cen = y - mean
            y = cen / torch.sqrt(var + eps) * sd[f"layers.{b}.weight"] + sd[f"layers.{b}.bias"]
        y = torch.clamp_min(y, 0.0)
    return y.reshape(n, s_, -1), new_buffers


def ntxent_loss(x, y, image_masks1, image_masks2, shift_masks1, shift_masks2, temperature=0.1):
    """joint_embedding_pretraining/losses.py:56-83.  Per line: rows selected by shift masks, cosine
    similarities / T, softmax normalised over dim 0 (columns), -log of the diagonal, mean.
    The reference then re-indexes the (reduced) similarity matrix with the full-length image masks
    (losses.py:78), which raises IndexError unless every shift mask is all ones; like the reference
    this restatement is defined for masks for which that indexing is valid."""
    xn = x / torch.sqrt((x * x).sum(dim=-1, keepdim=True)).clamp_min(1e-12)
    yn = y / torch.sqrt((y * y).sum(dim=-1, keepdim=True)).clamp_min(1e-12)
    losses = []
    for i in range(x.shape[0]):
        lx = xn[i][torch.as_tensor(shift_masks1[i]) == 1]
        ly = yn[i][torch.as_tensor(shift_masks2[i]) == 1]
        sim = (lx @ ly.t()) / temperature
        sim = sim[torch.as_tensor(image_masks1[i]) == 1, :][:, torch.as_tensor(image_masks2[i]) == 1]
        e = torch.exp(sim)
        losses.append((-torch.log(torch.diag(e) / e.sum(dim=0))).mean())
    # the reference accumulates per-line losses into a float32 CPU tensor (losses.py:57,63)
    return {"loss": torch.stack(losses).mean()}


def vqvae_quantize(features, sd):
    """VQVAE.quantize (reference models/autoencoders.py:142-146) in eval mode: 1x1 encoder projection (a matmul over the channel axis,
    Conv2d semantics autoencoders.py:113), VectorQuantizer.forward (autoencoders.py:204-241: expanded squared distance, first argmin,
    straight-through output), 1x1 decoder projection (autoencoders.py:114).  features (N, C, 1, T) f32, sd: the layer's tensors under the
    reference's names.  Returns (projected rows (N*T, D), labels (N*T,), projected tokens (N, C_dec, 1, T))."""
    x = torch.as_tensor(features, dtype=torch.float32)
    n, c, _, t = x.shape
    we = torch.as_tensor(sd["encoder_projection_layer.weight"]).reshape(-1, c)
    be = torch.as_tensor(sd["encoder_projection_layer.bias"])
    rows = x.permute(0, 2, 3, 1).reshape(-1, c) @ we.t() + be
    e = torch.as_tensor(sd["vq.embedding.weight"])
    dist = (rows ** 2).sum(dim=1, keepdim=True) + (e ** 2).sum(dim=1) - 2 * rows @ e.t()
    labels = torch.argmin(dist, dim=1)
    q = rows + (e[labels] - rows)
    wd = torch.as_tensor(sd["decoder_projection_layer.weight"]).reshape(-1, e.shape[1])
    bd = torch.as_tensor(sd["decoder_projection_layer.bias"])
    tok = (q @ wd.t() + bd).reshape(n, 1, t, -1).permute(0, 3, 1, 2)
    return rows, labels, tok


def ntxent_cross_loss(x, y, lines_per_rank, temperature=0.1):
    """The cross-rank-negatives EXTENSION of NT-Xent (no reference counterpart; pero_pretraining_amd NTXentLoss(cross_rank_negatives=
    True)) restated on the CONCATENATED batch x, y (L, S, D) of all ranks: per line the reference's column-normalised softmax
    (joint_embedding_pretraining/losses.py:73-83) whose normaliser additionally holds one pooled view-1 embedding of every OTHER
    line, p_l = normalize(mean_i normalize(x_l,i)).  Returns (mean loss over all lines, per-rank mean losses)."""
    xn = x / torch.sqrt((x * x).sum(dim=-1, keepdim=True)).clamp_min(1e-12)
    yn = y / torch.sqrt((y * y).sum(dim=-1, keepdim=True)).clamp_min(1e-12)
    pm = xn.mean(dim=1)
    p = pm / torch.sqrt((pm * pm).sum(dim=-1, keepdim=True)).clamp_min(1e-12)
    losses = []
    for l in range(x.shape[0]):
        sim = (xn[l] @ yn[l].t()) / temperature                      # [i, j]
        neg = (p @ yn[l].t()) / temperature                          # [l', j]
        keep = torch.ones(x.shape[0], dtype=torch.bool, device=x.device)
        keep[l] = False
        denom = torch.exp(sim).sum(dim=0) + torch.exp(neg[keep]).sum(dim=0)
        losses.append((torch.log(denom) - torch.diag(sim)).mean())
    losses = torch.stack(losses)
    return losses.mean(), losses.view(-1, lines_per_rank).mean(dim=1)


def joint_model_forward(sd, images1_nchw, images2_nchw, masks, num_heads, loss="vicreg",
                        offsets1=None, offsets2=None, max_len=4096):
    """JointEmbeddingTransformerEncoder.forward with a LinearHead.
    joint_embedding_pretraining/model.py:41-60.  masks = (im1, im2, sm1, sm2)."""
    outs = []
    for img, off in ((images1_nchw, offsets1), (images2_nchw, offsets2)):
        n = img.shape[0]
        t = backbone_tokens(sd, img, num_heads, None, off, max_len)
        outs.append((t @ sd["head.linear.weight"].t() + sd["head.linear.bias"]).reshape(n, t.shape[0] // n, -1))
    fn = vicreg_loss if loss == "vicreg" else ntxent_loss
    return outs[0], outs[1], fn(outs[0], outs[1], *masks)


# --------------------------------------------------------------------------------------------
# whole training step (used for trajectory parity and as the bench's CPU baseline "port")
# --------------------------------------------------------------------------------------------
# -------------------------------------------------------------------------------------------------
# Evaluation
# -------------------------------------------------------------------------------------------------
def topk_error_counts(output, labels, mask, measured_errors=(1, 3, 10)):
    """masked_pretraining/tester.py:72-113 (_update_errors, _topk, _calculate_errors) restated:
    output (N, S, V) numpy, labels / mask (N, S).  Returns ({'errors_k': count}, length)."""
    output, labels, mask = np.asarray(output), np.asarray(labels), np.asarray(mask)
    masked_output = output[mask == 1]
    masked_labels = labels[mask == 1]
    counts = {}
    for k in measured_errors:
        if k == 1:
            pred = np.argmax(masked_output, axis=1)                        # tester.py:86 (first maximum)
            counts[f"errors_{k}"] = int(np.sum(pred != masked_labels))
        else:
            top = np.argsort(masked_output, axis=1, kind="stable")[:, -k:]  # tester.py:94-96 (ties: see eval.hip)
            counts[f"errors_{k}"] = int(sum(r not in h for h, r in zip(top, masked_labels)))
    return counts, int(masked_labels.shape[0])


def label_ranks(output2d, labels, mask):
    """(rows, 3) int32: #{logit > x}, #{logit == x, j < label}, #{logit == x, j > label} per row with mask == 1
    (x = logit[label]); -1 elsewhere.  The quantity pero_label_rank returns."""
    output2d = np.asarray(output2d, dtype=np.float32)
    rows, V = output2d.shape
    out = np.full((rows, 3), -1, dtype=np.int32)
    j = np.arange(V)
    for r in range(rows):
        if mask[r] != 1:
            continue
        lab = int(labels[r])
        if not 0 <= lab < V:
            out[r] = (V, 0, 0)
            continue
        x = output2d[r, lab]
        eq = output2d[r] == x
        out[r] = (int(np.sum(output2d[r] > x)), int(np.sum(eq & (j < lab))), int(np.sum(eq & (j > lab))))
    return out


def errors_from_ranks(ranks, measured_errors=(1, 3, 10)):
    act = ranks[:, 0] >= 0
    gt, lo, hi = ranks[act, 0], ranks[act, 1], ranks[act, 2]
    return {f"errors_{k}": int(np.sum((gt + lo) >= 1) if k == 1 else np.sum((gt + hi) >= k)) for k in measured_errors}, int(act.sum())


# -------------------------------------------------------------------------------------------------
# Batch collation
# -------------------------------------------------------------------------------------------------
def padded_width(max_width, padding_coefficient=32):
    """common/dataloader.py:197-198."""
    return int(np.ceil(max_width / padding_coefficient) * padding_coefficient) + padding_coefficient


def collate_view(lines, left_paddings, target_width, sub=8):
    """common/dataloader.py:80-100 for GIVEN left paddings (label positions): padded uint8 batch + image mask."""
    h, c = lines[0].shape[0], lines[0].shape[2]
    images = np.zeros((len(lines), h, target_width, c), dtype=np.uint8)
    masks = np.ones((len(lines), target_width // sub), dtype=np.uint8)
    for img, m, line, lp in zip(images, masks, lines, left_paddings):
        img[:, lp * sub:lp * sub + line.shape[1]] = line
        m[:lp] = 0
        m[lp + int(np.ceil(line.shape[1] / sub)):] = 0
    return images, masks


def shift_masks(shifts, image_masks1, image_masks2):
    """common/dataloader.py:128-138: three-valued masks (1 = shared position, 2 = shared but padding in that view)."""
    sm1 = np.zeros_like(image_masks1)
    for row, shift in zip(sm1, shifts):
        if shift < 0:
            row[:shift] = 1
        else:
            row[shift:] = 1
    sm2 = np.copy(sm1[:, ::-1])
    sm1[np.bitwise_and(sm1 == 1, image_masks1 == 0)] = 2
    sm2[np.bitwise_and(sm2 == 1, image_masks2 == 0)] = 2
    return sm1, sm2


def draw_left_paddings(lines, target_width, sub=8):
    """The reference's RNG consumption for one view (common/dataloader.py:86-89), from the global numpy stream."""
    return [0 if l.shape[1] == target_width else np.random.randint(0, target_width - l.shape[1]) // sub for l in lines]


class MaskedStepOracle:
    """Plain-tensor restatement of Trainer.train_step (masked_pretraining/trainer.py:53-68):
    prepare_batch -> forward -> backward (autograd over the restated arithmetic) -> Adam."""

    def __init__(self, state_dict, num_heads, max_len=4096):
        self.sd = {k: v.clone().requires_grad_(True) for k, v in state_dict.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.sd.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.sd.items()}
        self.num_heads = num_heads
        self.max_len = max_len
        self.steps = 0

    def step(self, images_u8_nhwc, labels, mask, lr, offsets=None):
        x = prepare_images(torch.as_tensor(images_u8_nhwc))
        _, loss = masked_model_forward(self.sd, x, torch.as_tensor(labels).long(), mask, self.num_heads,
                                       offsets, self.max_len)
        grads = torch.autograd.grad(loss, list(self.sd.values()))
        self.steps += 1
        with torch.no_grad():
            for (k, p), g in zip(self.sd.items(), grads):
                adam_update(p, g, self.m[k], self.v[k], self.steps, lr)
        return float(loss.detach())
