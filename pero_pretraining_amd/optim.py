"""Fused Adam over one flat parameter buffer (replaces the ~150 per-tensor kernels of torch.optim.Adam,
masked_pretraining/train.py:146, by ONE HIP launch per step that also refreshes the bf16 weight copies).

Drop-in: same constructor arguments and param_groups protocol as torch.optim.Adam (the reference's
WarmupSchleduler writes param_group["lr"]); parameters keep their identity, names and shapes - only their
storage is moved into the flat buffer, so state_dict()/load_state_dict() of the model are unaffected.
"""
import torch

from . import lowp, ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if weight_decay != 0:
            raise ValueError("FusedAdam: weight_decay is not implemented (the reference uses 0)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = []  # per group: dict(p, g, m, v, lp, step)
        self.grad_scale = 1.0
        self._flatten()

    def _flatten(self):
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise RuntimeError("FusedAdam needs the parameters on the GPU (move the model first)")
            offs, total = [], 0
            for p in ps:
                offs.append(total)
                total += ((p.numel() + 7) // 8) * 8  # 16-byte aligned bf16 views
            fp = torch.zeros(total, device=dev, dtype=torch.float32)
            fg = torch.zeros(total, device=dev, dtype=torch.float32)
            flp = torch.empty(total, device=dev, dtype=torch.bfloat16)
            for p, o in zip(ps, offs):
                n = p.numel()
                fp[o:o + n].copy_(p.detach().reshape(-1))
                if p.grad is not None:
                    fg[o:o + n].copy_(p.grad.reshape(-1))
                p.data = fp[o:o + n].view(p.shape)
                p.grad = fg[o:o + n].view(p.shape)
            ops.cast_to_bf16(fp, flp)
            for p, o in zip(ps, offs):
                lowp.put(p, flp[o:o + p.numel()].view(p.shape))
            self._flat.append(dict(p=fp, g=fg, m=torch.zeros_like(fp), v=torch.zeros_like(fp), lp=flp, step=0,
                                   params=ps, offsets=offs))

    def refresh_lowp(self):
        """Re-cast the bf16 copies after the flat parameters were written from outside (broadcast, load)."""
        for f in self._flat:
            if f is not None:
                ops.cast_to_bf16(f["p"], f["lp"])

    # flat views for data-parallel gradient reduction
    def flat_grads(self):
        return [f["g"] for f in self._flat if f is not None]

    def param_offsets(self):
        """{id(param): (group index, start, numel)} inside the flat buffers."""
        out = {}
        for gi, f in enumerate(self._flat):
            if f is not None:
                for p, o in zip(f["params"], f["offsets"]):
                    out[id(p)] = (gi, o, p.numel())
        return out

    def zero_grad(self, set_to_none=False):
        for f in self._flat:
            if f is None:
                continue
            f["g"].zero_()
            for p, o in zip(f["params"], f["offsets"]):
                if p.grad is None or p.grad.data_ptr() != f["g"].data_ptr() + 4 * o:
                    p.grad = f["g"][o:o + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            f["step"] += 1
            b1, b2 = group["betas"]
            ops.adam_step(f["p"], f["g"], f["m"], f["v"], f["lp"], group["lr"], b1, b2, group["eps"], f["step"],
                          self.grad_scale)
        return loss
