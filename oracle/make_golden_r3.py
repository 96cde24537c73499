#!/usr/bin/env python3
"""Round-3 golden vectors, produced by the reference itself.  TEST INFRASTRUCTURE.

Runs ONLY in the build container (imports /root/reference read-only on CPU); writes under tests/golden/:

  g20_vqvae_quantize.npz   `VQVAE.quantize` (models/autoencoders.py:142-146: the 1x1 `encoder_projection_layer` in front of the
                           VectorQuantizer, then the 1x1 `decoder_projection_layer`) of a reference
                           `VQVAE(VGGEncoder(pretrained_vgg_layers=0), VGGDecoder(), 256, 64)` - local code only, no pretrained weights
                            - in eval mode on seeded encoder features: the three layers' weights, the input
                           features, the projected rows, labels, best / second-best distance of every row, projected tokens.

usage:  python oracle/make_golden_r3.py [--out tests/golden]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import REFERENCE_ROOT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    out = os.path.abspath(ap.parse_args().out)
    sys.path.insert(0, REFERENCE_ROOT)
    torch.set_num_threads(8)
    from pero_pretraining.models import autoencoders as R_ae

    K, D = 256, 64
    torch.manual_seed(20)
    enc = R_ae.VGGEncoder(pretrained_vgg_layers=0)
    dec = R_ae.VGGDecoder()
    model = R_ae.VQVAE(enc, dec, K, D).eval()
    C = enc.out_channels
    g = np.random.default_rng(2020)
    feats = g.standard_normal((3, C, 1, 40)).astype(np.float32)          # (N, C_enc, 1, T): what VQVAE.encode hands to quantize
    with torch.no_grad():
        x = torch.from_numpy(feats)
        tokens, labels = model.quantize(x)
        proj = model.encoder_projection_layer(x)                          # the quantizer's input
        flat = proj.permute(0, 2, 3, 1).reshape(-1, D)
        w = model.vq.embedding.weight
        dist = (torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(w ** 2, dim=1) - 2 * torch.matmul(flat, w.t()))
        two = torch.topk(dist, 2, dim=1, largest=False).values
        assert torch.equal(torch.argmin(dist, dim=1), labels)
    sd = {k: v.numpy() for k, v in model.state_dict().items()
          if k.startswith(("encoder_projection_layer.", "decoder_projection_layer.", "vq."))}
    np.savez_compressed(os.path.join(out, "g20_vqvae_quantize.npz"), features=feats, projected=proj.numpy(), labels=labels.numpy(),
                        best=two[:, 0].numpy(), second=two[:, 1].numpy(), tokens=tokens.numpy(),
                        num_embeddings=np.int64(K), embeddings_dim=np.int64(D), encoder_channels=np.int64(C),
                        decoder_channels=np.int64(dec.base_channels), **{"sd." + k: v for k, v in sd.items()})
    print("g20: features", feats.shape, "labels", tuple(labels.shape), "tokens", tuple(tokens.shape), "min relative margin",
          float(((two[:, 1] - two[:, 0]) / two[:, 0].abs()).min()), "keys", sorted(sd))


if __name__ == "__main__":
    main()
