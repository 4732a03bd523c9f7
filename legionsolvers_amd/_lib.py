"""ctypes binding of liblsk.so (the C ABI declared in include/lsk.h).

There is no CPU fallback: if the library is missing, importing the product path raises.
torch is imported first on purpose: liblsk.so needs libamdhip64.so.7 / librccl.so.1, and the
dynamic loader must hand it the copies torch already mapped, so that pointers and streams made
by torch are valid inside the library (one HIP runtime per process).
"""
from __future__ import annotations

import ctypes
import os
import re

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSK_LIBRARY") or os.path.join(PKG_DIR, "liblsk.so")  # LSK_LIBRARY: another build of it
HEADER_PATH = os.path.join(os.path.dirname(PKG_DIR), "include", "lsk.h")

LSK_OK = 0
LSK_ERR_ARG = -2
LSK_ERR_UNSUPPORTED = -4
UNIQUE_ID_BYTES = 128

SOP_NEG, SOP_ADD, SOP_SUB, SOP_MUL, SOP_DIV, SOP_SQRT, SOP_RSQRT, SOP_COPY = range(8)


class LskError(RuntimeError):
    """A C-ABI call returned a non-zero status; the message is lsk_last_error()."""

    def __init__(self, code: int, message: str):
        super().__init__(f"lsk error {code}: {message}")
        self.code = code


_i = ctypes.c_int
_i64 = ctypes.c_int64
_sz = ctypes.c_size_t
_p = ctypes.c_void_p
_f32 = ctypes.c_float
_f64 = ctypes.c_double


def _typed(suffix_types, make):
    out = {}
    for suf, ct in suffix_types:
        out.update(make(suf, ct))
    return out


_FT = (("f32", _f32), ("f64", _f64))

# name -> (restype, argtypes)
PROTOTYPES = {
    "lsk_version": (ctypes.c_char_p, []),
    "lsk_last_error": (ctypes.c_char_p, []),
    "lsk_set_abort_on_error": (None, [_i]),
    "lsk_set_option": (_i, [ctypes.c_char_p, _i]),
    "lsk_get_option": (_i, [ctypes.c_char_p, ctypes.POINTER(_i)]),
    "lsk_device_count": (_i, [ctypes.POINTER(_i)]),
    "lsk_init": (_i, [_i]),
    "lsk_malloc": (_i, [ctypes.POINTER(_p), _sz]),
    "lsk_free": (_i, [_p]),
    "lsk_mem_info": (_i, [ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "lsk_stream_create": (_i, [ctypes.POINTER(_p)]),
    "lsk_stream_destroy": (_i, [_p]),
    "lsk_stream_synchronize": (_i, [_p]),
    "lsk_stream_check": (_i, [_p]),
    "lsk_stream_begin_capture": (_i, [_p]),
    "lsk_stream_end_capture": (_i, [_p, ctypes.POINTER(_p)]),
    "lsk_stream_is_recording": (_i, [_p, ctypes.POINTER(ctypes.c_int)]),
    "lsk_trace_replay": (_i, [_p, _p]),
    "lsk_trace_info": (_i, [_p, _p]),
    "lsk_trace_destroy": (_i, [_p]),
    "lsk_memcpy_h2d": (_i, [_p, _p, _sz, _p]),
    "lsk_memcpy_d2h": (_i, [_p, _p, _sz, _p]),
    "lsk_memcpy_d2d": (_i, [_p, _p, _sz, _p]),
    "lsk_dot_workspace_bytes": (_sz, [_i64]),
    "lsk_multi_dot_workspace_bytes": (_sz, [_i]),
    "lsk_dot_noalloc_f64": (_i, [_i64, _p, _p, _p, _p]),
    "lsk_csr_spmv_dot_partials": (_i64, [_i64, _i64, _i]),
    "lsk_rects_to_indptr_i64": (_i, [_i64, _p, _p, _p]),
    "lsk_index_bounds_i64": (_i, [_i64, _p, _p, _p]),
    "lsk_csr_halo_rows_i64": (_i, [_i64, _i64, _p, _i64, _p, _i64, _i64, _p, _p]),
    "lsk_csr_compress_cols_u32": (_i, [_i64, _p, _i64, _i64, _p, _p]),
    "lsk_csr_compress_rows_u32": (_i, [_i64, _i64, _p, _i64, _p, _p]),
    "lsk_stencil_count": (_i, [_i, _p, _i, _p, _i64, _i64, _p, _p]),
    "lsk_coo_block_order_u32": (_i, [_i64, _p, _i64, _i64, _p, _p]),
    "lsk_gather_u32": (_i, [_i64, _p, _p, _p, _p]),
    "lsk_gather_f32": (_i, [_i64, _p, _p, _p, _p]),
    "lsk_gather_f64": (_i, [_i64, _p, _p, _p, _p]),
    "lsk_csr_sliced_create_f32": (_i, [_i64, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _p, _p]),
    "lsk_csr_sliced_create_f64": (_i, [_i64, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _p, _p]),
    "lsk_csr_sliced_update_values_f32": (_i, [_p, _i64, _i64, _p, _i64, _p, _p]),
    "lsk_csr_sliced_update_values_f64": (_i, [_p, _i64, _i64, _p, _i64, _p, _p]),
    "lsk_csr_sliced_spmv_f32": (_i, [_p, _i64, _i64, _p, _p, _i, _p]),
    "lsk_csr_sliced_spmv_f64": (_i, [_p, _i64, _i64, _p, _p, _i, _p]),
    "lsk_csr_sliced_spmv_dot_f32": (_i, [_p, _i64, _i64, _p, _p, _i, _p, _p, _i64, _p, _p]),
    "lsk_csr_sliced_spmv_dot_f64": (_i, [_p, _i64, _i64, _p, _p, _i, _p, _p, _i64, _p, _p]),
    "lsk_csr_sliced_spmv2_f32": (_i, [_p, _i64, _i64, _p, _p, _p, _p, _i, _p]),
    "lsk_csr_sliced_spmv2_f64": (_i, [_p, _i64, _i64, _p, _p, _p, _p, _i, _p]),
    "lsk_csr_sliced_spmv2_dot_f32": (_i, [_p, _i64, _i64, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i64, _p, _p]),
    "lsk_csr_sliced_spmv2_dot_f64": (_i, [_p, _i64, _i64, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i64, _p, _p]),
    "lsk_csr_sliced_tune_f32": (_i, [_p, _p, _p, _p, _p, _p]),
    "lsk_csr_sliced_tune_f64": (_i, [_p, _p, _p, _p, _p, _p]),
    "lsk_csr_sliced_uniform_values": (_i, [_p]),
    "lsk_csr_sliced_tune2_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p]),
    "lsk_csr_sliced_tune2_f64": (_i, [_p, _p, _p, _p, _p, _p, _p, _p]),
    "lsk_csr_sliced_info": (_i, [_p, _p]),
    "lsk_csr_sliced_dot_partials": (_i64, [_p, _i64, _i64]),
    "lsk_csr_sliced_destroy": (_i, [_p]),
    "lsk_coo_binned_create_f32": (_i, [_i64, _p, _p, _p, _i64, _i64, _i64, _i, _i, _p, _p]),
    "lsk_coo_binned_create_f64": (_i, [_i64, _p, _p, _p, _i64, _i64, _i64, _i, _i, _p, _p]),
    "lsk_coo_binned_spmv_f32": (_i, [_p, _p, _p, _i, _p]),
    "lsk_coo_binned_spmv_f64": (_i, [_p, _p, _p, _i, _p]),
    "lsk_coo_binned_spmv_phase_f32": (_i, [_p, _p, _p, _i, _i, _i64, _i64, _p]),
    "lsk_coo_binned_spmv_phase_f64": (_i, [_p, _p, _p, _i, _i, _i64, _i64, _p]),
    "lsk_coo_binned_update_values_f32": (_i, [_p, _i64, _p, _p]),
    "lsk_coo_binned_update_values_f64": (_i, [_p, _i64, _p, _p]),
    "lsk_coo_binned_info": (_i, [_p, _p]),
    "lsk_coo_binned_destroy": (_i, [_p]),
    "lsk_csr_long_rows_create": (_i, [_i64, _p, _i64, _p, _p]),
    "lsk_csr_long_rows_info": (_i, [_p, _p]),
    "lsk_csr_long_rows_destroy": (_i, [_p]),
    "lsk_csr_spmv_c32_split_f32": (_i, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _i64, _p, _i, _p]),
    "lsk_csr_spmv_c32_split_f64": (_i, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _i64, _p, _i, _p]),
    "lsk_csr_entry_rows_u32": (_i, [_i64, _i64, _p, _i64, _p, _p]),
    "lsk_sorted_key_offsets_u32": (_i, [_i64, _p, _i64, _p, _p]),
    "lsk_inclusive_sum_i64": (_i, [_i64, _p, _p, _p]),
    "lsk_rects_from_counts_i64": (_i, [_i64, _p, _p, _i64, _p, _p]),
    "lsk_task_id_t": (_i64, [_i, _i]),
    "lsk_task_id_di": (_i64, [_i, _i, _i]),
    "lsk_task_id_tdi": (_i64, [_i, _i, _i, _i]),
    "lsk_task_id_tdddiii": (_i64, [_i] * 8),
    "lsk_comm_unique_id": (_i, [_p]),
    "lsk_comm_init_rank": (_i, [ctypes.POINTER(_p), _i, _i, _p]),
    "lsk_comm_destroy": (_i, [_p]),
    "lsk_comm_abort": (_i, [_p]),
    "lsk_comm_rank": (_i, [_p, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
}
PROTOTYPES.update(_typed(_FT, lambda s, t: {
    f"lsk_get_alpha_{s}": (_i, [_i, _p, _p]),
    f"lsk_scal_{s}": (_i, [_i64, t, _p, _p]),
    f"lsk_axpy_{s}": (_i, [_i64, t, _p, _p, _p]),
    f"lsk_xpay_{s}": (_i, [_i64, t, _p, _p, _p]),
    f"lsk_xpay_nd_{s}": (_i, [_i, _p, t, _p, _p, _p, _p, _p]),
    f"lsk_fill_{s}": (_i, [_i64, t, _p, _p]),
    f"lsk_copy_{s}": (_i, [_i64, _p, _p, _p]),
    f"lsk_dot_{s}": (_i, [_i64, _p, _p, _p, _p, _sz, _p]),
    f"lsk_dot_host_{s}": (_i, [_i64, _p, _p, _p, _p]),
    f"lsk_nrm2_{s}": (_i, [_i64, _p, _p, _p, _sz, _p]),
    f"lsk_scalar_op_{s}": (_i, [_i, _p, _p, _p, _p]),
    f"lsk_sum_scalars_{s}": (_i, [_i, _p, _p, _p]),
    f"lsk_scal_dev_{s}": (_i, [_i64, _i, _p, _p, _p, _p, _p, _p]),
    f"lsk_axpy_dev_{s}": (_i, [_i64, _i, _p, _p, _p, _p, _p, _p, _p]),
    f"lsk_xpay_dev_{s}": (_i, [_i64, _i, _p, _p, _p, _p, _p, _p, _p]),
    f"lsk_csr_spmv_{s}": (_i, [_i64, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _p, _i, _p]),
    f"lsk_csr_spmv_dot_{s}": (_i, [_i64, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _p, _i, _p, _p, _i64,
                                   ctypes.POINTER(_i64), _p]),
    f"lsk_axpy_sqnorm_{s}": (_i, [_i64, t, _p, _p, _p, _i64, ctypes.POINTER(_i), _p]),
    f"lsk_axpy_sqnorm_dev_{s}": (_i, [_i64, _i, _p, _p, _p, _p, _p, _p, _p, _i64, ctypes.POINTER(_i), _p]),
    f"lsk_axpy_dot_dev_{s}": (_i, [_i64, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i64, ctypes.POINTER(_i), _p]),
    f"lsk_axpy_xpay_dev_{s}": (_i, [_i64, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    f"lsk_sum_partials_{s}": (_i, [_i64, _p, _p, _p, _sz, _p]),
    f"lsk_multi_dot_{s}": (_i, [_i64, _i, _p, _p, _p, _p, _sz, _p]),
    f"lsk_multi_axpy_{s}": (_i, [_i64, _i, _p, _p, _i, _p, _p, _i64, ctypes.POINTER(_i), _p]),
    f"lsk_csr_spmv_tune_{s}": (_i, [_i64, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _p, _p,
                                    ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_float)]),
    f"lsk_csr_spmv_c32_{s}": (_i, [_i64, _i64, _p, _p, _p, _p, _i64, _p, _i, _p]),
    f"lsk_csr_spmv_c32_dot_{s}": (_i, [_i64, _i64, _p, _p, _p, _p, _i64, _p, _i, _p, _p, _i64,
                                       ctypes.POINTER(_i64), _p]),
    f"lsk_csr_spmv_c32_tune_{s}": (_i, [_i64, _i64, _p, _p, _p, _p, _i64, _p, _p,
                                        ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_float)]),
    f"lsk_coo_spmv_{s}": (_i, [_i64, _p, _p, _p, _p, _i64, _i64, _p, _i64, _i64, _p]),
    f"lsk_random_fill_{s}": (_i, [_i64, ctypes.c_uint64, _i64, _p, _p]),
    f"lsk_coo_spmv_c32_{s}": (_i, [_i64, _p, _p, _p, _p, _i64, _p, _i64, _i64, _p]),
    f"lsk_stencil_fill_csr_{s}": (_i, [_i, _p, _i, _p, _p, _i64, _i64, _p, _i64, _p, _p, _p]),
    f"lsk_stencil_fill_coo_{s}": (_i, [_i, _p, _i, _p, _p, _i64, _i64, _p, _i64, _p, _p, _p, _p]),
    f"lsk_allreduce_sum_{s}": (_i, [_p, _p, _i, _p]),
    f"lsk_halo_exchange_{s}": (_i, [_p, _i, _p, _p, _p, _i, _p, _p, _p, _p]),
    f"lsk_allgather_{s}": (_i, [_p, _p, _p, _i64, _p]),
}))


def declared_symbols(header_path: str = HEADER_PATH) -> list:
    """Names of every function include/lsk.h declares."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lsk_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib() -> ctypes.CDLL:
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built and there is no CPU "
                "fallback. Run `python -m legionsolvers_amd.build` (or __graft_entry__.build()).")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(code: int) -> None:
    if code != LSK_OK:
        raise LskError(code, lib().lsk_last_error().decode(errors="replace"))


def suffix(dtype: torch.dtype) -> str:
    if dtype == torch.float64:
        return "f64"
    if dtype == torch.float32:
        return "f32"
    raise TypeError(f"entry type {dtype} is not supported (float32 / float64 only)")


def call(name: str, *args) -> None:
    """Call a status-returning entry point and raise on failure."""
    check(getattr(lib(), name)(*args))
