"""ctypes binding of libpero_hip.so (the C ABI declared in include/pero_hip.h).

The product path has no fallback: if the shared library is missing this module raises at import of
the first op, and every non-zero return code becomes a Python exception carrying pero_last_error().
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpero_hip.so")

PERO_F32, PERO_BF16 = 0, 1
LN_BWD_BLOCKS = 512
GEMM_RELU, GEMM_ATOMIC, GEMM_ACCUM, GEMM_TRANS_A, GEMM_TRANS_B, GEMM_FORCE_GENERIC = 1, 2, 4, 8, 16, 32
GEMM_ROWDOT = 4096   # `gate` = second matrix, `bias` = f32 [M][N/128] output of the 128-column-block row dots of the stored result
GEMM_RELU_BITS = 8192
GEMM_MASK_TILED = 16384
GEMM_COLSUM = 1024  # `bias` is an OUTPUT: column sums of the stored result (bias gradient of the upstream Linear)
GEMM_TILE_V = 512  # prefer the 256x256x64 kernel: products that have the GPU to themselves (forward pass)
GEMM_TILE128, GEMM_TILE256 = 64, 128

_vp, _i64, _i32, _f32, _f64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_double

# name -> argtypes (all functions return int unless noted)
SIGNATURES = {
    "pero_patches_from_u8": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp],
    "pero_patches_from_f32": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp],
    "pero_cast_pad_f32_bf16": [_vp, _vp, _i64, _i64, _i64, _vp],
    "pero_add_rows2d": [_vp, _vp, _i64, _i64, _i64, _i64, _vp],
    "pero_apply_mask_f32": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp],
    "pero_gemm": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                  _i64, _i64, _i64, _i64, _i64, _i64, _f32, _i32, _i32, _i32, _i32, _vp, _i64, _vp],
    "pero_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32, _i32, _vp],
    "pero_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "pero_layernorm_bwd_out": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "pero_attention_fwd": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "pero_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "pero_softmax_fwd": [_vp, _vp, _i64, _i64, _f32, _i32, _vp],
    "pero_softmax_bwd": [_vp, _vp, _vp, _i64, _i64, _f32, _i32, _vp],
    "pero_masked_ce_fwd": [_vp, _vp, _vp, _f32, _vp, _vp, _i64, _i64, _i32, _vp],
    "pero_masked_ce_bwd": [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "pero_masked_ce_fwd_rows": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp],
    "pero_zero_fill": [_vp, _i64, _vp],
    "pero_masked_ce_bwd_rows": [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _vp],
    "pero_colsum": [_vp, _vp, _i64, _i64, _i64, _i32, _vp],
    "pero_cast_f32_bf16": [_vp, _vp, _i64, _vp],
    "pero_transpose_multi": [_vp, _vp, _vp, _i64, _i64, _vp],
    "pero_cast_bf16_f32": [_vp, _vp, _i64, _vp],
    "pero_scale": [_vp, _i64, _f32, _i32, _vp],
    "pero_adam_step": [_vp, _vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64, _i64, _f64, _vp],
    "pero_vq_argmin": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp],
    "pero_vq_gather": [_vp, _vp, _vp, _vp, _i64, _i64, _vp],
    "pero_gather_rows": [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp],
    "pero_scatter_add_rows": [_vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "pero_sqdiff_rows": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _i32, _vp],
    "pero_sqdiff_rows_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _i64, _i32, _vp],
    "pero_sum_scale": [_vp, _vp, _i64, _f32, _vp],
    "pero_center_cols": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp],
    "pero_vicreg_var": [_vp, _vp, _vp, _i64, _i64, _f32, _f32, _vp],
    "pero_vicreg_cov": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _i32, _vp],
    "pero_scatter_add_rows_scaled": [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "pero_rownorm_fwd": [_vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "pero_rownorm_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "pero_ntxent_cols": [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "pero_ntxent_cols_cross": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "pero_line_mean": [_vp, _vp, _i64, _i64, _i64, _i32, _vp],
    "pero_add_line_rows": [_vp, _vp, _i64, _i64, _i64, _f32, _i32, _vp],
    "pero_vq_ema_update": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f64, _f64, _vp],
    "pero_rowdot_blocks": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
    "pero_gemm_resid_layernorm": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp],
    "pero_gemm_resid_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp],
    "pero_bn_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _i32, _i32, _i32, _vp],
    "pero_bn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp],
    "pero_label_rank": [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _i32, _vp, _vp, _i32, _vp],
    "pero_stack_lines": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
    "pero_line_masks": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp],
}


class PeroHipError(RuntimeError):
    pass


ABI_VERSION = 2
_lib = None
_lib_lock = threading.Lock()


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PeroHipError(
                f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        h = ctypes.CDLL(LIB_PATH)
        h.pero_abi_version.restype = ctypes.c_int
        h.pero_abi_version.argtypes = []
        if h.pero_abi_version() != ABI_VERSION:
            raise PeroHipError(f"{LIB_PATH} has ABI version {h.pero_abi_version()}, this package binds version {ABI_VERSION}: rebuild it "
                               f"(`make -C {os.path.join(_HERE, 'csrc')}`)")
        h.pero_last_error.restype = ctypes.c_char_p
        h.pero_last_error.argtypes = []
        h.pero_set_option.restype = ctypes.c_int
        h.pero_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
        h.pero_gemm_workspace_bytes.restype = ctypes.c_int64
        h.pero_gemm_workspace_bytes.argtypes = [_i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32]
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = ctypes.c_int
            fn.argtypes = args
        # PERO_OPTIONS="name=value,name=value": pero_set_option knobs for a whole process (A/B runs of a tool or of bench.py under a profiler); unknown names raise
        for item in filter(None, os.environ.get("PERO_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            if h.pero_set_option(name.strip().encode(), int(value)) != 0:
                raise PeroHipError(f"PERO_OPTIONS: {h.pero_last_error().decode()}")
        _lib = h
    return _lib


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise PeroHipError(f"{name} failed ({rc}): {lib().pero_last_error().decode()}")
