"""Backbones with the construction API, attribute names and state_dict keys of the reference's
pero_pretraining/models/transformers.py, computing through the HIP kernels.

* `VisionTransformerEncoder(height, patch_size, in_channels, model_dim, num_heads, num_blocks,
  feedforward_dim, dropout, max_len)`; `backbone(x, mask=None) -> (N, model_dim, S)`;
  `backbone.mask(x, mask)` (in place, like the reference: models/transformers.py:53-68).
* parameters are held in the same torch.nn containers the reference builds (Conv2d, LayerNorm,
  TransformerEncoder) so initialisation (same RNG draws, same clone-of-one-layer quirk) and checkpoint
  keys are identical; their torch forward methods are never called - `forward` launches HIP kernels.
* deliberate differences: the mask tile lives on the input's device (the reference hard-codes "cuda" in
  the constructor, models/transformers.py:34, which breaks CPU construction); positional offsets for
  training (`random_shift`) are drawn with the same `torch.randint(0, max_len - S, (N,), device=x.device)`
  call as the reference (models/transformers.py:182) but may also be injected (`set_offsets`) so that
  parity tests are deterministic; dropout != 0 is rejected (the reference scripts always use 0.0).
"""
import math
from abc import ABC, abstractmethod

import numpy as np
import torch

from .. import functional as F
from .. import ops
from ..precision import compute_dtype


class PositionalEncoding(torch.nn.Module):
    """models/transformers.py:154-192.  `pe` is a non-persistent buffer of shape [max_len, 1, d_model]."""

    def __init__(self, d_model, max_len=1024, random_shift=True, **kwargs):
        super().__init__()
        self.d_model = d_model
        self.max_len = max_len
        self.random_shift = random_shift
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(0, 1), persistent=False)
        self._next_offsets = None

    def pe_table(self, device):
        if self.pe.device != device:
            self.pe = self.pe.to(device)
        return self.pe.view(self.max_len, self.d_model)

    def draw_offsets(self, batch_size, seq_len, device):
        """Per-line start rows into the table (None = 0 for every line), reference lines 178-188."""
        if self._next_offsets:
            off = self._next_offsets.pop(0)
            return torch.as_tensor(off, dtype=torch.int64).to(device)
        if self.random_shift and self.training:
            max_shift = self.max_len - seq_len
            if max_shift > 0:
                return torch.randint(0, max_shift, (batch_size,), device=device)
        return None

    def extra_repr(self):
        return f"d_model={self.d_model}, max_len={self.max_len}, random_shift={self.random_shift}"


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, mask, offsets, dtype, *params):
        save = any(ctx.needs_input_grad)
        if save:
            mod._grad_forwards = getattr(mod, "_grad_forwards", 0) + 1  # parallel.DataParallel overlaps only single-pass steps
        F.drop_row_grad_hint()   # (an announcement a failed or abandoned backward left behind must not meet this pass's gradient)
        tokens, saved = F.backbone_fwd(mod, x, mask, offsets, dtype, save)
        ctx.mod, ctx.saved, ctx.dtype = mod, saved, dtype
        return tokens

    @staticmethod
    def backward(ctx, dt):
        if ctx.saved is None:
            raise RuntimeError("backbone backward called twice or without saved activations")
        rows = F.take_row_grad_hint(dt)   # the head's loss announces a gradient that is zero outside its masked rows (functional.ROW_SPARSE_LAST_LAYER)
        dt = dt.to(ctx.dtype).contiguous()
        F.backbone_bwd(ctx.mod, ctx.saved, dt, ctx.dtype, ctx.mod._on_layer_grads_ready, rows=rows)
        ctx.saved = None
        return (None,) * (5 + len(ctx.mod._param_list))


class TransformerEncoder(ABC, torch.nn.Module):
    def __init__(self, height=40, patch_size=(40, 8), in_channels=3, model_dim=512, num_heads=4, num_blocks=6,
                 feedforward_dim=2048, dropout=0.0, max_len=4096, *args, **kwargs):
        super().__init__()
        if dropout != 0.0:
            raise ValueError("pero_pretraining_amd: dropout != 0.0 is not implemented in the HIP path")
        self.height = height
        self.patch_size = tuple(patch_size)
        self.in_channels = in_channels
        self.model_dim = model_dim
        self.num_heads = num_heads
        self.num_blocks = num_blocks
        self.feedforward_dim = feedforward_dim
        self.dropout = dropout
        self.max_len = max_len

        self.position_model = PositionalEncoding(self.model_dim, self.max_len)
        self.encoder_layers = self.create_layers()
        self.intermediate_norm = torch.nn.LayerNorm(self.model_dim, eps=1e-05)
        # same global-RNG side effect and the same tile values as the reference (transformers.py:30-32)
        np.random.seed(42)
        mask_tile = np.random.rand(1, self.in_channels, self.patch_size[0], self.patch_size[1])
        self._mask_tile = torch.tensor(mask_tile, dtype=torch.float32)[0].contiguous()
        self._on_layer_grads_ready = None  # set by parallel.DataParallel for comm/compute overlap

    def create_layers(self):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            layer = torch.nn.TransformerEncoderLayer(d_model=self.model_dim, nhead=self.num_heads,
                                                     dim_feedforward=self.feedforward_dim, dropout=self.dropout)
            return torch.nn.TransformerEncoder(layer, num_layers=self.num_blocks)

    @property
    def mask_pattern(self):
        """(1, C, H, 512*P) tiled noise pattern, as the reference attribute of the same name."""
        return self._mask_tile[None].repeat(1, 1, 1, 512)

    def mask_tile_device(self, device):
        if self._mask_tile.device != device:
            self._mask_tile = self._mask_tile.to(device)
        return self._mask_tile

    def set_offsets(self, *offsets):
        """Inject the positional start rows of the next encode(s), one array per encode in call order (the joint step makes
        two): deterministic parity tests replay the reference's torch.randint draws this way."""
        self.position_model._next_offsets = [o for o in offsets if o is not None]

    @property
    def _param_list(self):
        return list(self.parameters())

    def forward(self, x, mask=None):
        tokens = self.encode_tokens(x, mask)
        n = x.shape[0]
        return tokens.view(n, -1, self.model_dim).permute(0, 2, 1)

    def encode_tokens(self, x, mask=None):
        """(N*S, model_dim) token rows (row-major, line-major)."""
        if not x.is_cuda:
            raise RuntimeError("pero_pretraining_amd backbones run on the GPU only (HIP kernels, no CPU fallback)")
        if x.dtype != torch.uint8 and mask is not None:
            # reference semantic: the caller's tensor is overwritten in place.  The reference BatchOperator's
            # `.float().permute(0, 3, 1, 2) / 255` keeps NHWC strides: mask a contiguous copy, write it back, go on with the copy
            if x.dtype == torch.float32 and not x.is_contiguous():
                xc = x.contiguous()
                self.mask(xc, mask)
                x.copy_(xc)
                x = xc
            else:
                self.mask(x, mask)
            mask = None
        w = x.shape[2] if x.dtype == torch.uint8 else x.shape[3]
        offsets = self.position_model.draw_offsets(x.shape[0], w // self.patch_size[1], x.device)
        return _BackboneFn.apply(self, x, mask, offsets, compute_dtype(), *self._param_list)

    def encode_tokens_views(self, views):
        """Several equally shaped image batches (the two views of the joint-embedding step) through ONE pass of 2N lines:
        half the launches and twice the rows per product.  The positional offsets are drawn per view, in view order, with the
        calls a sequence of separate encodes would make - the same device RNG stream as the reference's two encodes."""
        if any(v.shape != views[0].shape or v.dtype != views[0].dtype for v in views):
            raise ValueError("encode_tokens_views: the views must have one shape and dtype")
        n, seq = views[0].shape[0], views[0].shape[-1 if views[0].dtype != torch.uint8 else 2] // self.patch_size[1]
        drawn = [self.position_model.draw_offsets(n, seq, views[0].device) for _ in views]
        offsets = None if any(o is None for o in drawn) else torch.cat(drawn)
        # (two device copies into one buffer: torch.cat of the uint8 views is an at::native gather kernel at a third of the copy rate)
        x = torch.empty((len(views) * n,) + tuple(views[0].shape[1:]), device=views[0].device, dtype=views[0].dtype)
        for k, v in enumerate(views):
            x[k * n:(k + 1) * n].copy_(v)
        tokens = _BackboneFn.apply(self, x, None, offsets, compute_dtype(), *self._param_list)
        return tokens  # (len(views) * N * S, model_dim): view v, line i at rows (v * N + i) * S ...

    def mask(self, x, mask):
        m = torch.as_tensor(mask).to(device=x.device, dtype=torch.int64).contiguous()
        if x.dtype != torch.float32 or not x.is_contiguous():
            raise TypeError("mask(): x must be a contiguous float32 (N,C,H,W) tensor")
        return ops.apply_mask_(x, m, self.mask_tile_device(x.device), self.patch_size[1])

    def encode(self, x):
        return self.forward(x)

    @abstractmethod
    def _conv(self, x):
        pass


class VisionTransformerEncoder(TransformerEncoder):
    def __init__(self, height=40, patch_size=(40, 8), in_channels=3, model_dim=512, num_heads=4, num_blocks=6,
                 feedforward_dim=2048, dropout=0.0, *args, **kwargs):
        super().__init__(height=height, patch_size=patch_size, in_channels=in_channels, model_dim=model_dim,
                         num_heads=num_heads, num_blocks=num_blocks, feedforward_dim=feedforward_dim,
                         dropout=dropout, *args, **kwargs)
        self.conv_layer = torch.nn.Conv2d(in_channels=self.in_channels, out_channels=self.model_dim,
                                          kernel_size=self.patch_size, stride=self.patch_size)

    def _conv(self, x):  # part of the reference's class surface; the HIP path fuses it into encode_tokens
        raise NotImplementedError("use forward(): patch embedding is fused into the HIP front end")
