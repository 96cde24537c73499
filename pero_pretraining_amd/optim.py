"""Fused Adam over one flat parameter buffer (replaces the ~150 per-tensor kernels of torch.optim.Adam,
masked_pretraining/train.py:146, by ONE HIP launch per step that also refreshes the bf16 weight copies).

Drop-in: same constructor arguments and param_groups protocol as torch.optim.Adam (the reference's
WarmupSchleduler writes param_group["lr"]); parameters keep their identity, names and shapes - only their
storage is moved into the flat buffer, so state_dict()/load_state_dict() of the model are unaffected.
"""
import torch

from . import lowp, ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False):
        if weight_decay != 0 or amsgrad or maximize:
            raise ValueError("FusedAdam: weight_decay / amsgrad / maximize are not implemented (the reference uses none)")
        # the remaining keys are torch.optim.Adam's group defaults, carried so that state_dict()s are interchangeable
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
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
            fp = ops.zeros((total,), dev, torch.float32)   # (the library's linear fill: torch's elementwise one runs at a third of its rate)
            fg = ops.zeros((total,), dev, torch.float32)
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
            # transposed bf16 copies of the matrices (lowp.weight_t: the input-gradient products read them), same offsets
            mats = [(o, o, p.shape[0], p.numel() // p.shape[0]) for p, o in zip(ps, offs) if p.dim() >= 2]
            flpt, ttable, ttiles = None, None, 0
            if mats:
                flpt = ops.zeros((total,), dev, torch.bfloat16)
                ttable, ttiles = ops.transpose_table(mats, dev)
                ops.transpose_multi(flp, flpt, ttable, ttiles)
                for p, o in zip(ps, offs):
                    if p.dim() >= 2:
                        lowp.put_t(p, flpt[o:o + p.numel()].view(p.numel() // p.shape[0], p.shape[0]))
            self._flat.append(dict(p=fp, g=fg, m=ops.zeros((total,), dev, torch.float32), v=ops.zeros((total,), dev, torch.float32), lp=flp, step=0,
                                   params=ps, offsets=offs, lpt=flpt, ttable=ttable, ttiles=ttiles))

    def refresh_lowp(self):
        """Re-cast the bf16 copies after the flat parameters were written from outside (broadcast, load)."""
        for f in self._flat:
            if f is not None:
                ops.cast_to_bf16(f["p"], f["lp"])
                self._transpose(f)

    @staticmethod
    def _transpose(f):
        if f["lpt"] is not None:
            ops.transpose_multi(f["lp"], f["lpt"], f["ttable"], f["ttiles"])

    # flat views for data-parallel gradient reduction
    def flat_grads(self):
        """{parameter-group index: flat f32 gradient buffer} (groups without parameters have none)"""
        return {gi: f["g"] for gi, f in enumerate(self._flat) if f is not None}

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

    # ---- checkpointing: the layout of torch.optim.Adam's state_dict (per-parameter 'step', 'exp_avg', 'exp_avg_sq'),
    # so optimizer state moves between this class and torch.optim.Adam (the reference's optimizer,
    # masked_pretraining/train.py:146) in both directions.  The reference itself never saves it.
    def state_dict(self):
        state, groups, idx = {}, [], 0
        for group, f in zip(self.param_groups, self._flat):
            offs = {id(p): o for p, o in zip(f["params"], f["offsets"])} if f is not None else {}
            ids = []
            for p in group["params"]:
                if id(p) in offs and f["step"] > 0:
                    o, n = offs[id(p)], p.numel()
                    state[idx] = {"step": torch.tensor(float(f["step"])),
                                  "exp_avg": f["m"][o:o + n].view(p.shape).clone(),
                                  "exp_avg_sq": f["v"][o:o + n].view(p.shape).clone()}
                ids.append(idx)
                idx += 1
            g = {k: v for k, v in group.items() if k != "params"}
            g["params"] = ids
            groups.append(g)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        for group, saved, f in zip(self.param_groups, groups, self._flat):
            if len(saved["params"]) != len(group["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
            if saved.get("weight_decay", 0) != 0 or saved.get("amsgrad", False) or saved.get("maximize", False):
                raise ValueError("FusedAdam: the loaded group uses weight_decay / amsgrad / maximize")
            for k, v in saved.items():
                if k != "params":
                    group[k] = v
            if f is None:
                continue
            offs = {id(p): o for p, o in zip(f["params"], f["offsets"])}
            f["m"].zero_(); f["v"].zero_()
            steps = set()
            for p, idx in zip(group["params"], saved["params"]):
                st = state_dict["state"].get(idx)
                if st is None or id(p) not in offs:
                    continue
                o, n = offs[id(p)], p.numel()
                f["m"][o:o + n].copy_(st["exp_avg"].reshape(-1))
                f["v"][o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise ValueError(f"FusedAdam keeps one step counter per group, got {sorted(steps)}")
            f["step"] = steps.pop() if steps else 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            # a gradient that no longer aliases the flat buffer (model.zero_grad(set_to_none=True) or `p.grad = ...` made the
            # backward write a fresh tensor) is copied in and relinked: stepping on the stale flat slice would be silent
            for p, o in zip(f["params"], f["offsets"]):
                if p.grad is not None and p.grad.data_ptr() != f["g"].data_ptr() + 4 * o:
                    if self.grad_scale != 1.0:
                        # a data-parallel driver reduced the FLAT buffer during the backward: this detached gradient is this rank's
                        # alone - copying it in would step every rank on its own gradient, scaled by 1 / world, silently diverging
                        raise RuntimeError("FusedAdam: a parameter's .grad no longer aliases the flat gradient buffer although a data-parallel "
                                           "driver is attached (grad_scale != 1): use optimizer.zero_grad() (it keeps the views), not "
                                           "model.zero_grad(set_to_none=True) / `p.grad = ...`")
                    f["g"][o:o + p.numel()].copy_(p.grad.reshape(-1))
                    p.grad = f["g"][o:o + p.numel()].view(p.shape)
            f["step"] += 1
            b1, b2 = group["betas"]
            ops.adam_step(f["p"], f["g"], f["m"], f["v"], f["lp"], group["lr"], b1, b2, group["eps"], f["step"],
                          self.grad_scale)
            self._transpose(f)
        return loss
