"""Compute precision of the HIP path.

* float32 (default): exact-f32 MFMA / VALU kernels - the parity mode checked against the CPU oracle.
* bfloat16: bf16 storage + bf16 MFMA with f32 accumulation and f32 statistics - the mode the reference
  reaches with `Trainer(bfloat16=True)` (torch.autocast, masked_pretraining/trainer.py:57-59).

The bf16 mode is selected either by this module's `autocast(True)` context or by an enclosing
`torch.autocast(device_type="cuda", dtype=torch.bfloat16)` (what the reference's Trainer enters), so
the reference's own training loop drives the same switch.
"""
import contextlib
import threading

import torch

_state = threading.local()


def compute_dtype():
    forced = getattr(_state, "dtype", None)
    if forced is not None:
        return forced
    if torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16:
        return torch.bfloat16
    return torch.float32


@contextlib.contextmanager
def autocast(enabled=True, dtype=torch.bfloat16):
    prev = getattr(_state, "dtype", None)
    _state.dtype = dtype if enabled else torch.float32
    try:
        yield
    finally:
        _state.dtype = prev
