"""CPU: the C-ABI library loads and exports every symbol include/pero_hip.h declares (no compute calls),
the ctypes signature table covers the header, host-side mirrors of the reference interface behave like
the reference (construction, state_dict keys, scheduler, batch operator, error behaviour)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "pero_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pero_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    from pero_pretraining_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    h = _lib.lib()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/pero_hip.h but not exported"
    assert set(_lib.SIGNATURES) | {"pero_last_error", "pero_abi_version", "pero_set_option"} == set(names)
    assert h.pero_abi_version() == 1


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch (error code + thread-local message)."""
    from pero_pretraining_amd import _lib
    h = _lib.lib()
    rc = h.pero_layernorm_fwd(None, None, None, None, None, None, None, None, 4, 64, 1, 1e-5, 0, None)
    assert rc == -1 and b"null pointer" in h.pero_last_error()
    with pytest.raises(_lib.PeroHipError):
        _lib.call("pero_colsum", None, None, 0, 0, 0, 0, None)


def test_no_cpu_fallback():
    from pero_pretraining_amd._lib import PeroHipError
    from pero_pretraining_amd.masked_pretraining import model as M
    bb = M.init_backbone({"num_blocks": 1, "model_dim": 64, "num_heads": 4, "feedforward_dim": 128})
    with pytest.raises(RuntimeError):
        bb(torch.zeros(1, 3, 40, 64))
    with pytest.raises((RuntimeError, PeroHipError)):
        M.LinearHead(64, 32)(torch.zeros(1, 8, 64))


def test_reference_construction_api_and_errors(golden):
    from pero_pretraining_amd.masked_pretraining import model as M
    g = golden("g4_masked_tiny.npz")
    head_def = {"type": "linear", "in_features": 64, "out_features": 96}
    hd = M.init_head(head_def)
    assert "type" not in head_def  # the reference pops it from the caller's dict (model.py:22-23)
    bb = M.init_backbone({"type": "vit", "num_blocks": 2, "model_dim": 64, "num_heads": 4, "feedforward_dim": 128})
    model = M.MaskedTransformerEncoder(bb, hd)
    ref_keys = [k[3:] for k in g.files if k.startswith("sd.")]
    assert list(model.state_dict().keys()) == ref_keys
    assert "pe" not in "".join(ref_keys) and "mask_pattern" not in "".join(ref_keys)
    for attr in ("height", "patch_size", "in_channels", "model_dim", "num_heads", "num_blocks", "feedforward_dim",
                 "dropout", "max_len", "position_model", "encoder_layers", "intermediate_norm", "mask_pattern", "conv_layer"):
        assert hasattr(bb, attr), attr
    assert bb.mask_pattern.shape == (1, 3, 40, 4096)
    assert np.array_equal(bb.mask_pattern[0, :, :, :8].numpy(), g_tile(golden))
    assert model.head.linear.out_features == 96
    assert isinstance(model.loss, M.MaskedCrossEntropyLoss) and model.loss.unmasked_weight is None
    with pytest.raises(ValueError, match="Unknown backbone type"):
        M.init_backbone({"type": "resnet"})
    with pytest.raises(ValueError, match="Unknown head type"):
        M.init_head({"type": "mlp2"})


def g_tile(golden):
    return golden("g1_tables.npz")["mask_tile"]


def test_constructor_reseeds_numpy_like_reference():
    """models/transformers.py:30 reseeds the global numpy RNG; masks drawn afterwards are therefore the same
    sequence in both implementations."""
    from pero_pretraining_amd.masked_pretraining import model as M
    from pero_pretraining_amd.masked_pretraining.batch_operator import BatchOperator
    np.random.seed(999)
    M.init_backbone({"num_blocks": 1, "model_dim": 64, "num_heads": 4, "feedforward_dim": 128})
    labels = np.zeros((2, 16), dtype=np.int64)
    labels[1, 10:] = -1
    got = BatchOperator(torch.device("cpu"), 0.3)._create_mask({"labels": labels})
    rs = np.random.RandomState(42)
    rs.rand(1, 3, 40, 8)
    expect = (rs.rand(2, 16) < 0.3).astype(int) * (labels >= 0)
    assert np.array_equal(got, expect) and got.dtype == expect.dtype


def test_scheduler_matches_golden(golden):
    from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler
    g = golden("g10_lr.npz")
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    s = WarmupSchleduler(opt, 2e-4, 10000, 1)
    for it, lr in zip(g["iterations"], g["lr"]):
        s.update_learning_rate(int(it))
        assert opt.param_groups[0]["lr"] == lr and s.current_lr == lr
    s2 = WarmupSchleduler(opt, 1e-3, 100, 2)
    for it, lr in zip(g["iterations2"], g["lr2"]):
        s2.update_learning_rate(int(it))
        assert opt.param_groups[0]["lr"] == lr


def test_positional_table_and_offsets(golden):
    from pero_pretraining_amd.models.transformers import PositionalEncoding
    g = golden("g1_tables.npz")
    pm = PositionalEncoding(64, 4096)
    assert pm.pe.shape == (4096, 1, 64) and "pe" not in pm.state_dict()
    assert np.array_equal(pm.pe[:, 0, :].numpy()[g["pe64_row_index"]], g["pe64_rows"])
    pm.eval()
    assert pm.draw_offsets(4, 16, torch.device("cpu")) is None
    pm.train()
    torch.manual_seed(3)
    expect = torch.randint(0, 4096 - 16, (4,))
    torch.manual_seed(3)
    assert torch.equal(pm.draw_offsets(4, 16, torch.device("cpu")), expect)


def test_lowp_cache_entries_die_with_their_parameter():
    """The bf16 weight cache is keyed by id(parameter); ids are reused after death, so an entry must not outlive its parameter
    (a stale hit once served another model's weights to a freshly built one)."""
    import gc
    import torch
    from pero_pretraining_amd import lowp
    p = torch.nn.Parameter(torch.zeros(8))
    lowp.put(p, torch.zeros(8, dtype=torch.bfloat16))
    key = id(p)
    assert key in lowp._cache
    del p
    gc.collect()
    assert key not in lowp._cache
    q = torch.nn.Parameter(torch.ones(8))
    lowp._cache[id(q)] = (lowp._key(q), torch.zeros(8, dtype=torch.bfloat16), True, lowp.weakref.ref(torch.nn.Parameter(torch.zeros(1))))
    assert lowp._entry(q) is None          # an entry whose weak reference is not this very object is discarded


def test_new_entry_points_validate_arguments_without_gpu():
    """pero_transpose_multi and the PERO_GEMM_RELU_BITS contract are refused on the host side before any launch."""
    import ctypes
    from pero_pretraining_amd import _lib
    L = _lib.lib()
    assert L.pero_transpose_multi(None, None, None, 1, 1, None) < 0
    assert b"pero_transpose_multi" in L.pero_last_error()
    buf = (ctypes.c_uint16 * 64)()
    tab = (ctypes.c_int64 * 5)(0, 0, 8, 8, 0)
    assert L.pero_transpose_multi(buf, buf, tab, 0, 1, None) < 0          # no matrices
    assert L.pero_transpose_multi(buf, buf, tab, 1, 0, None) < 0          # no tiles


def test_masked_head_mode_is_validated_on_the_host():
    from pero_pretraining_amd.masked_pretraining import model as M
    bb = M.init_backbone({"type": "vit", "num_blocks": 1, "model_dim": 64, "num_heads": 1, "feedforward_dim": 64})
    model = M.MaskedTransformerEncoder(bb, M.init_head({"type": "linear", "in_features": 64, "out_features": 32}))
    assert model.head_rows == "all"
    model.head_rows = "some"
    with pytest.raises(ValueError, match="Unknown head_rows"):
        model(torch.zeros(1, 3, 40, 16), None, None)
