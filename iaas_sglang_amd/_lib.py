"""ctypes binding of libmi_hotpath.so (the C ABI declared in include/mi_hotpath.h).

There is NO fallback: if the shared library is missing or a symbol is absent the
import fails loudly -- the product path never computes on the CPU or through torch ops.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi_hotpath.so")

ABI_VERSION = 9          # MI_ABI_VERSION of include/mi_hotpath.h this table was written against
MI_BF16, MI_FP16, MI_F32 = 0, 1, 2
MI_SCALE_TENSOR, MI_SCALE_ROW = 0, 1
MI_W4_AWQ, MI_W4_GPTQ = 0, 1

_p, _i64, _int, _f = C.c_void_p, C.c_int64, C.c_int, C.c_float

# name -> (restype, argtypes); mirrors include/mi_hotpath.h one to one
SIGNATURES = {
    "mi_abi_version": (_int, []),
    "mi_last_error": (C.c_char_p, []),
    "mi_device_cu_count": (_int, []),
    "mi_kv_indptr": (_int, [_p, _int, _p, _i64, _p]),
    "mi_kv_page_indptr": (_int, [_p, _int, _i64, _p, _i64, _p]),
    "mi_kv_page_indices": (_int, [_p, _i64, _p, _p, _int, _p, _p, _i64, _i64, _p]),
    "mi_kv_indices": (_int, [_p, _i64, _p, _p, _int, _p, _p, _p, _i64, _p]),
    "mi_kv_write": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p]),
    "mi_kv_write_fp8": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _f, _int, _p]),
    "mi_alloc_extend": (_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _p]),
    "mi_alloc_decode": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p]),
    "mi_write_req_to_token": (_int, [_p, _i64, _p, _p, _p, _p, _p, _i64, _p]),
    "mi_get_last_loc": (_int, [_p, _i64, _p, _p, _p, _i64, _p]),
    "mi_compute_position": (_int, [_p, _p, _int, _p, _p, _i64, _p]),
    "mi_decode_attn_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "mi_decode_attn": (_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                              _f, _f, _i64, _i64, _p, _i64, _p, _int, _p]),
    "mi_decode_attn_paged": (_int, [_p, _p, _p, _p, _p, _p, _int, _f, _f, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64,
                                    _i64, _i64, _i64, _i64, _f, _f, _i64, _i64, _p, _i64, _p, _int, _p]),
    "mi_decode_attn_fp8out": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                     _f, _f, _i64, _i64, _p, _i64, _p, _int, _p]),
    "mi_decode_attn_fp8kv": (_int, [_p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64,
                                    _i64, _i64, _f, _f, _i64, _i64, _p, _i64, _p, _int, _p]),
    "mi_extend_attn": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                              _i64, _i64, _i64, _i64, _f, _f, _int, _i64, _int, _p]),
    "mi_extend_attn_paged": (_int, [_p] * 11 + [_i64] * 12 + [_f, _f, _int, _i64, _int, _p]),
    "mi_extend_attn_fp8out": (_int, [_p] * 13 + [_i64] * 13 + [_f, _f, _int, _i64, _p, _p, _int, _p]),
    "mi_extend_attn_fp8kv": (_int, [_p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                    _i64, _i64, _i64, _i64, _f, _f, _int, _i64, _int, _p]),
    "mi_extend_attn_splitkv": (_int, [_p] * 6 + [_int, _f, _f] + [_p] * 5 + [_int] + [_i64] * 11 + [_f, _f, _int, _i64, _p, _i64, _i64,
                                      _int, _p]),
    "mi_extend_attn_masked": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _int, _i64, _i64, _i64, _i64, _i64, _i64,
                                     _i64, _i64, _i64, _i64, _i64, _f, _f, _i64, _int, _p]),
    "mi_merge_state": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _int, _p]),
    "mi_fp8_quant_per_tensor": (_int, [_p, _p, _p, _i64, _i64, _i64, _int, _int, _p]),
    "mi_fp8_quant_per_token": (_int, [_p, _p, _p, _i64, _i64, _i64, _int, _p]),
    "mi_fp8_gemm": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _p, _i64, _p]),
    "mi_fp8_gemm_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "mi_w4_repack": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _int, _int, _p]),
    "mi_w4a16_gemm": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p, _i64, _p]),
    "mi_w4a16_gemm_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "mi_w4a16_fused_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "mi_w4a16_gemm_add_rmsnorm": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f, _int, _p, _i64, _p]),
    "mi_w4a16_gemm_rope_kvwrite": (_int, [_p] * 9 + [_i64] * 10 + [_int, _p, _i64, _p]),
    "mi_w4a16_gemm_silu_mul": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _int, _p, _i64, _p]),
    "mi_ar_shared_bytes": (_i64, [_i64]),
    "mi_ar_alloc_shared": (_int, [_i64, C.POINTER(C.c_void_p)]),
    "mi_ar_free_shared": (_int, [_p]),
    "mi_ar_ipc_get": (_int, [_p, _p]),
    "mi_ar_ipc_open": (_int, [_p, C.POINTER(C.c_void_p)]),
    "mi_ar_ipc_close": (_int, [_p]),
    "mi_ar_create": (_p, [C.POINTER(C.c_void_p), _i64, _int, _int]),
    "mi_ar_destroy": (_int, [_p]),
    "mi_ar_error": (_int, [_p]),
    "mi_ar_all_reduce": (_int, [_p, _p, _p, _i64, _int, _p]),
    "mi_ar_staging": (_p, [_p]),
    "mi_ar_all_reduce_add_rmsnorm": (_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _f, _int, _p]),
    "mi_rmsnorm": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f, _int, _p]),
    "mi_rmsnorm_fp8": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f, _int, _p]),
    "mi_silu_and_mul_fp8": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _int, _p]),
    "mi_rope_neox": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p]),
    "mi_silu_and_mul": (_int, [_p, _p, _i64, _i64, _i64, _i64, _int, _p]),
    "mi_fp8_gemm_fused_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "mi_fp8_gemm_add_rmsnorm_fp8": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f, _int,
                                           _p, _i64, _p]),
    "mi_fp8_gemm_rope_kvwrite": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64,
                                        _i64, _i64, _i64, _i64, _int, _p, _i64, _p]),
    "mi_fp8_gemm_silu_mul_fp8": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _int, _p, _i64, _p]),
    "mi_gather_columns": (_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "mi_w4_dequantize_native": (_int, [_p, _p, _p, _i64, _i64, _i64, _int, _p]),
    "mi_w4_dequantize": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _int, _int, _p]),
}


class MiHotpathError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU/torch fallback for the hot path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise ImportError(f"{LIB_PATH} does not export {name} (ABI mismatch with include/mi_hotpath.h)")
        fn.restype = res
        fn.argtypes = args
    if lib.mi_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.mi_abi_version()} != {ABI_VERSION}")
    return lib


lib = _load()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise MiHotpathError(f"{what} failed (rc={rc}): {lib.mi_last_error().decode()}")
