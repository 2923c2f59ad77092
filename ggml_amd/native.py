"""ctypes binding of libcdna4_kernels.so (include/ggml_cdna4.h).  Loading is lazy and LOUD: there is no
fallback path — if the library is missing or there is no GPU, calls raise."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libcdna4_kernels.so")
if os.environ.get("CDNA4_KERNELS_LIB"):                # measurement only: the -DCDNA4_ABLATIONS twin of tools/microbench (timing-only / instrumented instantiations)
    LIB_PATH = os.environ["CDNA4_KERNELS_LIB"]
BACKEND_PATH = os.path.join(_PKG, "lib", "libggml-cdna4.so")

# every symbol include/ggml_cdna4.h declares: (name, restype, argtypes)
_i64, _vp, _sz, _int = C.c_int64, C.c_void_p, C.c_size_t, C.c_int
SYMBOLS = [
    ("ggml_cdna4_api_version", _int, []),
    ("ggml_cdna4_last_error", C.c_char_p, []),
    ("ggml_cdna4_device_count", _int, []),
    ("ggml_cdna4_set_shared_device", _int, [_int]),
    ("ggml_cdna4_device_fault", _int, [_int]),
    ("ggml_cdna4_debug_occupy", _int, [_int, _int, _vp, _int, _vp]),
    ("ggml_cdna4_mul_mat_route", _int, [_int, _i64, _i64, _i64]),
    ("ggml_cdna4_mul_mat_route_of", _int, [_int, _vp, _i64, _i64, _i64, _i64]),
    ("ggml_cdna4_set_device", _int, [_int]),
    ("ggml_cdna4_debug_trace", None, [_vp]),
    ("ggml_cdna4_scratch_generation", C.c_uint64, []),
    ("ggml_cdna4_row_size", _sz, [_int, _i64]),
    ("ggml_cdna4_mul_mat_workspace_size", _sz, [_int, _i64, _i64]),
    ("ggml_cdna4_mul_mat", _int, [_int, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _sz, _int, _int, _int, _vp]),
    ("ggml_cdna4_mul_mat_fused", _int, [_int, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp, _i64, _vp, _sz, _vp]),
    ("ggml_cdna4_mul_mat_fused_residual_may_alias", _int, [_int, _i64, _i64, _i64]),
    ("ggml_cdna4_mul_mat_group", _int, [_int, _int, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_vp), C.POINTER(_vp), _vp, _i64, _vp]),
    ("ggml_cdna4_act_image_key", C.c_uint32, [_int, _i64, _i64, _i64]),
    ("ggml_cdna4_act_image_key_of", C.c_uint32, [_int, _vp, _i64, _i64, _i64, _i64]),
    ("ggml_cdna4_mul_mat_prepared_fused", _int, [_int, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp, _i64, _vp, _sz, _vp]),
    ("ggml_cdna4_prepare_act", _int, [_int, _vp, _i64, _i64, _i64, _vp, _sz, _int, _vp]),
    ("ggml_cdna4_mul_mat_prepared", _int, [_int, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _sz, _int, _int, _int, _vp]),
    ("ggml_cdna4_mul_mat_id_workspace_size", _sz, [_int, _i64, _i64, _i64, _i64, _i64]),
    ("ggml_cdna4_mul_mat_id", _int, [_int, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64,
                                     _i64, _i64, _i64, _i64, _i64, _i64, _vp, _sz, _vp]),
    ("ggml_cdna4_mul_mat_id_front_key", C.c_uint32, [_int, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _sz]),
    ("ggml_cdna4_mul_mat_id_prepared", _int, [_int, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64,
                                              _i64, _i64, _i64, _i64, _i64, _i64, _vp, _sz, _vp]),
    ("ggml_cdna4_quantize_q8_K", _int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    ("ggml_cdna4_quantize_q8_0", _int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _vp]),
    ("ggml_cdna4_quantize_q8_1", _int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    # supporting ops: tensors are POINTER(Tensor) descriptors
    ("ggml_cdna4_op_binary", _int, [_int, _vp, _vp, _vp, _vp]),
    ("ggml_cdna4_sum_partials", _int, [_vp, _vp, _int, _i64, _vp]),
    ("ggml_cdna4_op_scale", _int, [_vp, _vp, C.c_float, _vp]),
    ("ggml_cdna4_op_norm", _int, [_vp, _vp, C.c_float, _int, _vp]),
    ("ggml_cdna4_op_norm_affine", _int, [_vp, _vp, _vp, _vp, C.c_float, _int, _vp]),
    ("ggml_cdna4_op_norm_affine_q8_K", _int, [_vp, _vp, _vp, _vp, C.c_float, _int, _int, _vp, _sz, _vp]),
    ("ggml_cdna4_op_norm_affine_q8_0", _int, [_vp, _vp, _vp, _vp, C.c_float, _int, _int, _vp, _sz, _vp]),
    ("ggml_cdna4_op_soft_max", _int, [_vp, _vp, _vp, C.c_float, C.c_float, _vp]),
    ("ggml_cdna4_op_soft_max_ext", _int, [_vp, _vp, _vp, C.c_float, C.c_float, _int, C.c_float, _int, _vp]),
    ("ggml_cdna4_op_diag_mask_inf", _int, [_vp, _vp, _int, _vp]),
    ("ggml_cdna4_op_unary", _int, [_int, _vp, _vp, _vp]),
    ("ggml_cdna4_op_get_rows", _int, [_vp, _vp, _vp, _vp]),
    ("ggml_cdna4_op_cpy", _int, [_vp, _vp, _int, _vp]),
    ("ggml_cdna4_op_mul_mat_f", _int, [_vp, _vp, _vp, _vp]),
    ("ggml_cdna4_mul_mat_exact_supported", _int, [_int, _i64]),
    ("ggml_cdna4_mul_mat_exact_workspace_size", _sz, [_int, _i64, _i64]),
    ("ggml_cdna4_mul_mat_exact", _int, [_int, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _sz, _vp]),
    ("ggml_cdna4_op_mul_mat_f_exact", _int, [_vp, _vp, _vp, _vp]),
    ("ggml_cdna4_op_norm_exact", _int, [_vp, _vp, C.c_float, _vp]),
    ("ggml_cdna4_op_rms_norm_exact", _int, [_vp, _vp, C.c_float, _vp]),
    ("ggml_cdna4_op_silu_exact", _int, [_vp, _vp, _vp]),
    ("ggml_cdna4_op_soft_max_exact", _int, [_vp, _vp, C.c_float, _vp]),
    ("ggml_cdna4_op_rope", _int, [_vp, _vp, _vp, _vp, _int, _int, _int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    ("ggml_cdna4_op_flash_attn_ext_supported", _int, [_i64, _int]),
    ("ggml_cdna4_op_flash_attn_ext", _int, [_vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, _vp]),
    ("ggml_cdna4_dequantize_row", _int, [_int, _vp, _vp, _i64, _vp]),
    ("ggml_cdna4_convert_weights_target", _int, [_int]),
    ("ggml_cdna4_convert_weights_size", _sz, [_int, _i64, _i64]),
    ("ggml_cdna4_convert_weights", _int, [_int, _vp, _i64, _i64, _i64, _vp, _vp]),
    ("ggml_cdna4_resident_image_size", _sz, [_int, _i64, _i64]),
    ("ggml_cdna4_resident_image_register", _int, [_int, _vp, _i64, _i64, _i64, _vp, _int, _vp]),
    ("ggml_cdna4_resident_image_unregister", _int, [_vp]),
    ("ggml_cdna4_resident_image_lookup", _int, [_int, _vp, _i64, _i64, _i64, C.POINTER(_vp)]),
]


class Tensor(C.Structure):
    """struct ggml_cdna4_tensor (include/ggml_cdna4.h)"""
    _fields_ = [("data", C.c_void_p), ("type", C.c_int32), ("reserved", C.c_int32), ("ne", C.c_int64 * 4), ("nb", C.c_int64 * 4)]

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """the loaded kernel library; raises NativeError if it was not built"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError("%s not found — run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)" % LIB_PATH)
        # torch ships its own libamdhip64 (SONAME libamdhip64.so.7).  Map it FIRST so our library binds to the
        # same HIP runtime instance torch allocates memory / creates streams with; loading ours first would pull
        # in /opt/rocm's copy as a second runtime in the process.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        if l.ggml_cdna4_api_version() != 1:
            raise NativeError("libcdna4_kernels.so API version mismatch")
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise NativeError("cdna4 kernel library: %s (status %d)" % (lib().ggml_cdna4_last_error().decode(), rc))
