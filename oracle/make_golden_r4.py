#!/usr/bin/env python3
"""Round-4 golden vectors, produced by the reference itself.  TEST INFRASTRUCTURE.

Runs ONLY in the build container (imports /root/reference read-only on CPU); writes under tests/golden/:

  g21_mlp_head_bn.npz   the reference's `MLPHead(use_bn=True)` (joint_embedding_pretraining/model.py:79-115: Linear, BatchNorm1d, ReLU, Linear,
                        BatchNorm1d, ReLU, Linear on the (N*S, D) rows) at small dims with non-trivial BatchNorm scales / shifts: the state dict
                        before the step, TRAINING-mode output and every gradient of sum(y * gy) for two consecutive steps' inputs (the second
                        step sees the running statistics of the first), the BatchNorm buffers after each step, and the EVAL-mode output after them.

usage:  python oracle/make_golden_r4.py [--out tests/golden]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import REFERENCE_ROOT, np_sd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    out = os.path.abspath(ap.parse_args().out)
    sys.path.insert(0, REFERENCE_ROOT)
    torch.set_num_threads(8)
    from pero_pretraining.joint_embedding_pretraining import model as R_jm

    torch.manual_seed(43)
    head = R_jm.MLPHead(in_dim=48, hidden_dim=96, num_layers=3, use_bn=True)
    with torch.no_grad():
        for m in head.layers:
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.rand(96) + 0.5)
                m.bias.copy_(torch.randn(96) * 0.2)
    head.train()
    fix = {}
    for k, v in np_sd(head).items():
        fix["sd0." + k] = v
    for step in range(2):
        x = (torch.randn(3, 20, 48) * (1.0 + step) + 0.3 * step).requires_grad_(True)
        head.zero_grad()
        y = head(x)
        gy = torch.randn_like(y)
        (y * gy).sum().backward()
        fix[f"s{step}.x"] = x.detach().numpy().copy()
        fix[f"s{step}.y"] = y.detach().numpy().copy()
        fix[f"s{step}.gy"] = gy.numpy().copy()
        fix[f"s{step}.grad_x"] = x.grad.numpy().copy()
        for k, p in head.named_parameters():
            fix[f"s{step}.grad.{k}"] = p.grad.numpy().copy()
        for k, b in head.named_buffers():
            fix[f"s{step}.buf.{k}"] = b.detach().numpy().copy()
    head.eval()
    with torch.no_grad():
        xe = torch.randn(2, 20, 48)
        fix["eval.x"] = xe.numpy().copy()
        fix["eval.y"] = head(xe).numpy().copy()
    np.savez_compressed(os.path.join(out, "g21_mlp_head_bn.npz"), **fix)
    print("g21:", sorted(k for k in fix if k.startswith("sd0.")), "bytes", os.path.getsize(os.path.join(out, "g21_mlp_head_bn.npz")))


if __name__ == "__main__":
    main()
