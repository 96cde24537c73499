"""pero_pretraining_amd - MI355X-native (gfx950 HIP) implementation of the pero-pretraining
per-step hot path behind the reference's own Python API (models / masked_pretraining /
joint_embedding_pretraining).  All arithmetic runs in libpero_hip.so (see include/pero_hip.h);
PyTorch provides device memory, streams, autograd bookkeeping and torch.distributed (RCCL)."""
from .precision import autocast, compute_dtype  # noqa: F401

__all__ = ["autocast", "compute_dtype"]
