"""bf16 copies of f32 master parameters for the bf16 MFMA path.

`get(p)` returns a bf16 tensor mirroring parameter `p`; it is re-cast (one HIP cast kernel) whenever
the parameter changed (`_version` / storage) since the copy was made.  The fused optimizer
(`optim.FusedAdam`) writes the bf16 copy of the whole flat parameter buffer inside its own kernel and
registers the fresh views through `put`, so a training step issues no separate cast.
"""
import weakref

import torch

from . import ops

_cache = {}  # id(p) -> (key, bf16 copy, is_optimizer_view, weakref to p): ids are reused once a parameter dies, so an entry only
             # counts while its weak reference still points at the SAME object (a stale hit served another model's weights)


def _key(p):
    return (p._version, p.data_ptr(), tuple(p.shape))


def _ref(p):
    key = id(p)
    return weakref.ref(p, lambda _r: _cache.pop(key, None))  # runs at the parameter's death, before its id can be reused


def _entry(p):
    ent = _cache.get(id(p))
    if ent is not None and ent[3]() is not p:
        del _cache[id(p)]
        return None
    return ent


def get(p):
    ent = _entry(p)
    k = _key(p)
    if ent is not None and ent[0] == k:
        return ent[1]
    if ent is not None and ent[2]:      # view into a flat buffer owned by the optimizer: refresh in place
        dst = ent[1]
    else:
        dst = torch.empty(p.shape, device=p.device, dtype=torch.bfloat16)
    src = p.detach()
    if src.numel() % 8 == 0 and src.is_contiguous() and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0:
        ops.cast_to_bf16(src, dst)
    else:  # tiny / unaligned tensors: pad through a scratch buffer
        n = src.numel()
        pad = torch.zeros(((n + 7) // 8) * 8, device=p.device, dtype=torch.float32)
        pad[:n] = src.reshape(-1)
        tmp = torch.empty(pad.numel(), device=p.device, dtype=torch.bfloat16)
        ops.cast_to_bf16(pad, tmp)
        dst.reshape(-1).copy_(tmp[:n])
    _cache[id(p)] = (k, dst, ent[2] if ent is not None else False, _ref(p))
    return dst


def put(p, lp_view):
    """Register `lp_view` (bf16, already holding the current value of p) as p's copy."""
    _cache[id(p)] = (_key(p), lp_view, True, _ref(p))


def weight(p, dtype):
    """Parameter tensor in the requested compute dtype."""
    return p.detach() if dtype == torch.float32 else get(p)


_cache_t = {}  # id(p) -> [key, transposed bf16 copy (in, out), device table, tiles, weakref]


def _ref_t(p):
    key = id(p)
    return weakref.ref(p, lambda _r: _cache_t.pop(key, None))


def put_t(p, lp_t_view):
    """Register `lp_t_view` (bf16 (in, out), already holding the transposed current value of p's 2-D view) - the fused
    optimizer refreshes all of them with one pero_transpose_multi launch per step."""
    _cache_t[id(p)] = [_key(p), lp_t_view, None, 0, _ref_t(p)]


def weight_t(p):
    """bf16 copy of the 2-D view (out, in) of weight p, TRANSPOSED to (in, out): the input gradient dX = dY W then runs as a
    K-contiguous ("NT") product, 10-20 % faster on the 256x256x64 tile kernels than reading W k-major.
    Cached per parameter version; parameters owned by FusedAdam are refreshed by the optimizer step itself."""
    ent = _cache_t.get(id(p))
    if ent is not None and ent[4]() is not p:
        del _cache_t[id(p)]
        ent = None
    k = _key(p)
    if ent is not None and ent[0] == k:
        return ent[1]
    src = get(p)  # current bf16 copy (re-cast if p changed)
    rows = p.shape[0]
    cols = p.numel() // rows
    if ent is None:
        ent = _cache_t[id(p)] = [k, torch.empty((cols, rows), device=p.device, dtype=torch.bfloat16), None, 0, _ref_t(p)]
    if src.data_ptr() % 16 or ent[1].data_ptr() % 16:  # unaligned view: plumbing fallback
        ent[1].copy_(src.view(rows, cols).t())
    else:
        if ent[2] is None:
            ent[2], ent[3] = ops.transpose_table([(0, 0, rows, cols)], p.device)
        ops.transpose_multi(src, ent[1], ent[2], ent[3])
    ent[0] = k
    return ent[1]
