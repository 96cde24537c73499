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
    assert set(_lib.SIGNATURES) | {"pero_last_error", "pero_abi_version", "pero_set_option", "pero_gemm_workspace_bytes"} == set(names)
    assert h.pero_abi_version() == _lib.ABI_VERSION == 2


def test_library_never_allocates():
    """include/pero_hip.h: the caller owns every buffer.  No allocation, free, blocking copy or device synchronisation anywhere in csrc/."""
    import glob
    import re
    csrc = os.path.join(ROOT, "pero_pretraining_amd", "csrc")
    banned = re.compile(r"\b(hipMalloc\w*|hipFree\w*|hipMemcpy(?!Async)\w*|hipDeviceSynchronize|hipStreamSynchronize|hipHostMalloc|hipMallocAsync)\s*\(")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp"))):
        for i, line in enumerate(open(f), 1):
            code = line.split("//")[0]
            assert not banned.search(code), f"{f}:{i}: {line.strip()}"


def test_gemm_workspace_query_needs_no_gpu():
    """pero_gemm_workspace_bytes depends on the shape only: split-K weight gradients of long reductions on 256x256 tiles get
    tiles x slices x 256 KiB, everything else 0."""
    from pero_pretraining_amd import _lib
    h = _lib.lib()
    A = _lib.GEMM_ATOMIC | _lib.GEMM_TRANS_A | _lib.GEMM_TRANS_B
    # linear1's weight gradient at 1024 lines: 2048 x 512 outputs = 16 tiles, 16 slices
    assert h.pero_gemm_workspace_bytes(2048, 512, 262144, 1, A, 0, _lib.PERO_BF16, _lib.PERO_F32) == 16 * 16 * 256 * 256 * 4
    # in_proj's: 1536 x 512 = 12 tiles, 21 slices (not XCD-aligned)
    assert h.pero_gemm_workspace_bytes(1536, 512, 262144, 1, A, 0, _lib.PERO_BF16, _lib.PERO_F32) == 12 * 21 * 256 * 256 * 4
    assert h.pero_gemm_workspace_bytes(1536, 512, 262144, 1, A, 4, _lib.PERO_BF16, _lib.PERO_F32) == 12 * 4 * 256 * 256 * 4
    for args in [(2048, 512, 262144, 1, A & ~_lib.GEMM_ATOMIC, 1, _lib.PERO_BF16, _lib.PERO_BF16),   # stored product
                 (2048, 512, 4096, 1, A, 0, _lib.PERO_BF16, _lib.PERO_F32),                          # short reduction: 128x128 kernel, atomics
                 (2048, 500, 262144, 1, A, 0, _lib.PERO_BF16, _lib.PERO_F32),                        # ragged
                 (2048, 512, 262144, 2, A, 0, _lib.PERO_BF16, _lib.PERO_F32),                        # batched
                 (2048, 512, 262144, 1, A, 0, _lib.PERO_F32, _lib.PERO_F32)]:                        # parity mode
        assert h.pero_gemm_workspace_bytes(*args) == 0, args
    # a workspace size without a pointer is rejected before any launch
    rc = h.pero_gemm(1, 1, 1, None, None, None, 256, 256, 256, 256, 256, 256, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, 0, 1, 1, 1, None, 64, None)
    assert rc == -1 and b"workspace" in h.pero_last_error()


@pytest.mark.gpu
def test_splitk_from_two_threads_on_two_streams_is_bit_identical():
    """The C ABI is re-entrant: the split-K weight-gradient path called concurrently from two Python threads, each on its own stream
    with its own caller-owned workspace, gives bit-identical results - to each other, to a single-threaded call and from run to run
    (forward runs on the main thread, weight gradients on autograd worker threads: include/pero_hip.h conventions)."""
    import threading
    from pero_pretraining_amd import ops
    torch.manual_seed(0)
    dev = torch.device("cuda")
    K, M, N = 65536, 1536, 512     # 12 tiles x 21 slices: the unaligned slice count (work items XCD by XCD) and a long reduction
    a = torch.randn(K, M, device=dev).bfloat16()
    b = torch.randn(K, N, device=dev).bfloat16()
    from pero_pretraining_amd import _lib
    need = _lib.lib().pero_gemm_workspace_bytes(M, N, K, 1, _lib.GEMM_ATOMIC | _lib.GEMM_TRANS_A | _lib.GEMM_TRANS_B, 0, _lib.PERO_BF16, _lib.PERO_F32)
    assert need == 12 * 21 * 256 * 256 * 4
    ref = torch.zeros(M, N, device=dev)
    ops.gemm(a, b, ref, trans_a=True, trans_b=True, atomic=True, k_split=0)
    torch.cuda.synchronize()
    exact = a.float().T @ b.float()
    assert float((ref - exact).abs().max()) < 2e-3 * float(exact.abs().max())
    outs, errs = [None, None], []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                res = []
                for _ in range(6):
                    c = torch.zeros(M, N, device=dev)
                    ops.gemm(a, b, c, trans_a=True, trans_b=True, atomic=True, k_split=0)
                    res.append(c)
                st.synchronize()
            outs[i] = res
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for res in outs:
        for c in res:
            assert torch.equal(c, ref)
    # two distinct workspaces were handed out (one per stream), none by the library
    assert len({v.data_ptr() for v in ops._ws_cache.values()}) >= 2


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


def test_inline_asm_vector_memory_of_the_tile_gemm_passes_the_isa_lint(tmp_path):
    """tools/check_async_loads.py on the ISA of csrc/gemm_e.hip (hipcc cross-compiles without a GPU): no instruction touches the destination of
    an inline-asm load before a counted s_waitcnt has retired it, and no vector-memory instruction reads an SGPR that a v_readlane_b32 (a
    restored spill) wrote fewer than five wait states earlier - the compiler pads neither for instructions inside an asm string (round 3: the
    LayerNorm epilogue's stores took stale row offsets; one store of the production ReLU-bits kernel had the same exposure)."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "gemm_e.s")
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffp-contract=off", "-Wno-unused-value",
                    "-S", "--cuda-device-only", os.path.join(root, "pero_pretraining_amd", "csrc", "gemm_e.hip"), "-o", out],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_async_loads.py"), out, "gemm_bf16"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "gemm_bf16_n512" in r.stdout and "gemm_bf16_e256" in r.stdout


def test_inline_asm_vector_memory_of_the_attention_kernels_passes_the_sgpr_lint(tmp_path):
    """The same SGPR hazard check on csrc/attention.hip (round-3 advice): its asm global_load_dword* / global_store_dword* / global_load_lds
    statements take an SGPR base, the kernels hold 20-38 v_readlane / v_readfirstlane each, and nothing but the `s_nop`s inside the asm strings
    keeps a restored SGPR five wait states away from the vector-memory instruction that reads it.  (Only the SGPR check: the in-flight-register
    check models straight-line code and the attention loops are compiler-scheduled.)"""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "attention.s")
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffp-contract=off", "-Wno-unused-value",
                    "-S", "--cuda-device-only", os.path.join(root, "pero_pretraining_amd", "csrc", "attention.hip"), "-o", out],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_async_loads.py"), out, "attn", "--sgpr-only"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "attn_fwd_p_k" in r.stdout and "attn_bwd_pair_k" in r.stdout and "attn_bwd_lh_k" in r.stdout
    src = open(os.path.join(root, "pero_pretraining_amd", "csrc", "attention.hip")).read()
    import re
    for m in re.finditer(r'asm volatile\("([^"]*(?:global_load_dword|global_store_dword)[^"]*)"', src):
        assert m.group(1).startswith("s_nop 4"), "asm vector-memory statement without leading wait states: " + m.group(1)[:60]
