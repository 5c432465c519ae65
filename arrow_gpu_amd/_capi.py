"""ctypes binding of the C ABI in include/arrow_gpu.h (libarrow_gpu_hip.so).

This is the Python-side equivalent of the `extern "C"` block a Rust shim would declare.  There is
no CPU fallback: if the shared library is missing this module raises at import of the first symbol, and on a box
without a gfx950 device `agpu_device_create` returns AGPU_ERR_NO_DEVICE, surfaced as ArrowErrorGPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
MAILBOX_MAX_BYTES = 3840  # AGPU_MAILBOX_MAX_BYTES (include/arrow_gpu.h)
LIB_PATH = os.environ.get("AGPU_LIB") or os.path.join(_HERE, "lib", "libarrow_gpu_hip.so")  # AGPU_LIB: A/B builds in tools/probe

# status codes
OK, ERR_UNSUPPORTED, ERR_SHAPE, ERR_HIP, ERR_ARG, ERR_NO_DEVICE = range(6)

# agpu_dtype
BOOL, F32, U32, U16, U8, I32, I16, I8, DATE32 = range(9)
# agpu_binary_op
OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_REM, OP_MIN, OP_MAX, OP_AND, OP_OR, OP_XOR, OP_SHL, OP_SHR, OP_POW = range(13)
# agpu_unary_op
(UN_NEG, UN_ABS, UN_NOT, UN_SQRT, UN_CBRT, UN_EXP, UN_EXP2, UN_LOG, UN_LOG2, UN_SIN, UN_COS, UN_ACOS, UN_SINH,
 UN_POPCOUNT) = range(14)
# agpu_cmp_op
CMP_GT, CMP_GTEQ, CMP_LT, CMP_LTEQ, CMP_EQ = range(5)
# agpu_reduce_op
RED_SUM, RED_MIN, RED_MAX = range(3)
# agpu_comm_dtype
COMM_F32, COMM_F64, COMM_I32, COMM_U32, COMM_I64, COMM_U64 = range(6)
COMM_ID_BYTES = 128

_vp, _u64, _i32, _u32, _i64, _sz = C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32, C.c_int64, C.c_size_t
_pp = C.POINTER(C.c_void_p)

# name -> argtypes (restype is agpu_status = int32 unless listed in _RESTYPES)
SIGNATURES = {
    "agpu_abi_version": [],
    "agpu_last_error": [],
    "agpu_build_info": [],
    "agpu_dtype_size": [_i32],
    "agpu_bitmap_bytes": [_u64],
    "agpu_device_count": [C.POINTER(_i32)],
    "agpu_device_create": [_i32, _pp],
    "agpu_device_destroy": [_vp],
    "agpu_device_sync": [_vp],
    "agpu_device_download": [_vp, _vp, _vp, _sz],
    "agpu_device_name": [_vp, C.c_char_p, _sz],
    "agpu_device_ordinal": [_vp, C.POINTER(_i32)],
    "agpu_device_mem_info": [_vp, C.POINTER(_u64), C.POINTER(_u64)],
    "agpu_device_trim": [_vp],
    "agpu_device_pool_info": [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)],
    "agpu_device_small_pool_info": [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)],
    "agpu_malloc": [_vp, _sz, _i32, _pp],
    "agpu_malloc_like": [_vp, _sz, _i32, _pp, _i32, _pp],
    "agpu_free": [_vp, _vp],
    "agpu_upload": [_vp, _vp, _vp, _sz],
    "agpu_download": [_vp, _vp, _vp, _sz],
    "agpu_copy": [_vp, _vp, _vp, _sz],
    "agpu_memset": [_vp, _vp, _i32, _sz],
    "agpu_host_alloc": [_vp, _sz, _pp],
    "agpu_host_free": [_vp, _vp],
    "agpu_upload_async": [_vp, _vp, _vp, _sz],
    "agpu_download_async": [_vp, _vp, _vp, _sz],
    "agpu_bitmap_copy_bits": [_vp, _vp, _u64, _vp, _u64],
    "agpu_pipeline_create": [_vp, _pp],
    "agpu_pipeline_wrap_stream": [_vp, _vp, _pp],
    "agpu_pipeline_finish": [_vp],
    "agpu_pipeline_sync": [_vp],
    "agpu_pipeline_destroy": [_vp],
    "agpu_pipeline_device": [_vp, _pp],
    "agpu_pipeline_stream": [_vp, _pp],
    "agpu_pipeline_begin_capture": [_vp],
    "agpu_pipeline_end_capture": [_vp, _pp],
    "agpu_graph_launch": [_vp, _vp],
    "agpu_graph_destroy": [_vp],
    "agpu_event_create": [_vp, _pp],
    "agpu_event_record": [_vp, _vp],
    "agpu_event_elapsed_ms": [_vp, _vp, C.POINTER(C.c_float)],
    "agpu_event_destroy": [_vp],
    "agpu_set_tuning": [C.c_char_p, _i64],
    "agpu_get_tuning": [C.c_char_p, C.POINTER(_i64)],
    "agpu_pipeline_set_tuning": [_vp, C.c_char_p, _i64],
    "agpu_pipeline_get_tuning": [_vp, C.c_char_p, C.POINTER(_i64)],
    "agpu_pipeline_wait_pipeline": [_vp, _vp],
    "agpu_pipeline_enable_timing": [_vp, _i32],
    "agpu_pipeline_last_kernel_ns": [_vp, C.POINTER(_u64), C.POINTER(C.c_char_p)],
    "agpu_comm_get_unique_id": [_vp],
    "agpu_comm_init_rank": [_vp, _vp, _i32, _i32, _pp],
    "agpu_comm_init_rank_timeout": [_vp, _vp, _i32, _i32, _i64, _pp],
    "agpu_comm_runtime_info": [C.c_char_p, _sz],
    "agpu_take_validity": [_vp, _i32, _vp, _u64, _vp, _vp, _vp, _vp, _u64],
    "agpu_compare_validity_count": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp],
    "agpu_bitmap_binary_count": [_vp, _i32, _vp, _vp, _vp, _u64, _vp],
    "agpu_bitmap_merge_validity_count": [_vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp],
    "agpu_shader_key_for_source": [C.c_char_p, _sz, C.c_char_p, _sz],
    "agpu_shader_key_for_hash": [_u64, _u64, C.c_char_p, _sz],
    "agpu_comm_destroy": [_vp],
    "agpu_comm_rank": [_vp, C.POINTER(_i32), C.POINTER(_i32)],
    "agpu_comm_reduce": [_vp, _vp, _i32, _i32, _vp, _vp, _u64, _vp],
    "agpu_comm_reduce_sum_f64": [_vp, _vp, _vp, _vp, _u64, _vp],
    "agpu_comm_reduce_stats_f32": [_vp, _vp, _vp, _vp, _u64, _vp],
    "agpu_comm_final_reduce": [_vp, _vp, _i32, _i32, _i32, _vp, _u64, _vp],
    "agpu_reduce_combine": [_vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "agpu_comm_all_reduce": [_vp, _vp, _i32, _i32, _vp, _u64],
    "agpu_comm_barrier": [_vp, _vp],
    "agpu_comm_sync": [_vp, _vp],
    "agpu_comm_size": [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)],
    "agpu_comm_is_local": [_vp, C.POINTER(_i32)],
    "agpu_device_identity": [_vp, _vp],
    "agpu_comm_peers": [_vp, _vp, _vp, _i32, C.POINTER(_i32)],
    "agpu_import_arrow": [_vp, _vp, _vp, _vp],
    "agpu_export_arrow": [_vp, _vp, _vp, _vp],
    "agpu_arrow_column_free": [_vp, _vp],
    "agpu_import_arrow_table": [_vp, _i32, _vp, _vp, _vp],
    "agpu_import_arrow_stream_next": [_vp, _vp, C.POINTER(_i32), _i32, _vp, C.POINTER(C.c_int64)],
    "agpu_ipc_read_batch": [_vp, C.c_int64, C.POINTER(_i32), _i32, _vp, _vp],
    "agpu_staged_copy": [_vp, _vp, _vp, _sz, _i32],
    "agpu_malloc_table": [_vp, _i32, C.POINTER(_u64), _i32, _pp],
    "agpu_ipc_open": [_vp, _u64, _pp],
    "agpu_ipc_close": [_vp],
    "agpu_ipc_num_fields": [_vp, C.POINTER(_i32)],
    "agpu_ipc_field_info": [_vp, _i32, _vp],
    "agpu_ipc_num_batches": [_vp, C.POINTER(C.c_int64)],
    "agpu_ipc_batch_rows": [_vp, C.c_int64, C.POINTER(C.c_int64)],
    "agpu_ipc_column_view": [_vp, C.c_int64, _i32, _vp, _vp],
    "agpu_ipc_read_column": [_vp, C.c_int64, _i32, _vp, _vp],
    "agpu_ipc_writer_create": [_vp, _i32, _i32, _i32, _pp],
    "agpu_ipc_writer_set_compression": [_vp, _i32],
    "agpu_ipc_writer_write_batch": [_vp, _vp],
    "agpu_ipc_writer_write_device_batch": [_vp, _vp, _vp],
    "agpu_ipc_writer_finish": [_vp, _pp, C.POINTER(_u64)],
    "agpu_ipc_writer_destroy": [_vp],
    "agpu_binary": [_vp, _i32, _i32, _vp, _vp, _vp, _u64],
    "agpu_scalar": [_vp, _i32, _i32, _vp, _vp, _vp, _u64],
    "agpu_unary": [_vp, _i32, _i32, _vp, _vp, _u64],
    "agpu_selftest_unary_f32": [_vp, _i32, _u64, _u64, _vp, _vp],
    "agpu_selftest_pow_f32": [_vp, _u64, _u64, _i32, _vp, _vp, _vp],
    "agpu_cast": [_vp, _i32, _i32, _vp, _vp, _u64],
    "agpu_broadcast": [_vp, _i32, _u32, _vp, _u64],
    "agpu_broadcast_from_device": [_vp, _i32, _vp, _vp, _u64],
    "agpu_fused_chain": [_vp, _i32, _vp, _vp, _i32, _vp, _u64],
    "agpu_fused_chain_compare": [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _u64],
    "agpu_fused_cast_chain": [_vp, _i32, _vp, _vp, _i32, _vp, _u64],
    "agpu_compare": [_vp, _i32, _i32, _vp, _vp, _vp, _u64],
    "agpu_compare_validity": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _u64],
    "agpu_bitmap_binary": [_vp, _i32, _vp, _vp, _vp, _u64],
    "agpu_bitmap_not": [_vp, _vp, _vp, _u64],
    "agpu_bitmap_popcount": [_vp, _vp, _u64, _vp],
    "agpu_bitmap_any": [_vp, _vp, _u64, _vp],
    "agpu_bitmap_merge_validity": [_vp, _vp, _vp, _vp, _vp, _vp, _u64],
    "agpu_reduce": [_vp, _i32, _i32, _vp, _vp, _u64, _vp],
    "agpu_reduce_sum_f64": [_vp, _vp, _vp, _u64, _vp],
    "agpu_reduce_stats_f32": [_vp, _vp, _vp, _u64, _vp],
    "agpu_take": [_vp, _i32, _vp, _u64, _vp, _vp, _u64],
    "agpu_take_columns": [_vp, _i32, _vp, _vp, _u64, _vp, _vp, _u64],
    "agpu_take_columns_validity": [_vp, _i32, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _u64],
    "agpu_take_bits": [_vp, _vp, _u64, _vp, _vp, _u64],
    "agpu_put_bounded": [_vp, _i32, _vp, _u64, _vp, _vp, _u64, _vp, _u64],
    "agpu_put_bits_bounded": [_vp, _vp, _u64, _vp, _vp, _u64, _vp, _u64],
    "agpu_put": [_vp, _i32, _vp, _vp, _vp, _vp, _u64],
    "agpu_put_bits": [_vp, _vp, _vp, _vp, _vp, _u64],
    "agpu_merge": [_vp, _i32, _vp, _vp, _vp, _vp, _u64],
    "agpu_merge_bits": [_vp, _vp, _vp, _vp, _vp, _u64],
    "agpu_index_max": [_vp, _vp, _u64, _vp],
    "agpu_launch_by_name": [_vp, C.c_char_p, C.c_char_p, _pp, _i32, _vp, _u64],
    "agpu_launch_by_name_sized": [_vp, C.c_char_p, C.c_char_p, _pp, C.POINTER(_u64), _i32, _vp, _u64, _u32],
    "agpu_synth_f32": [_vp, _vp, _u64, _u64, _u64, C.c_float, C.c_float],
    "agpu_synth_i32": [_vp, _vp, _u64, _u64, _u64, _u32],
    "agpu_synth_u8": [_vp, _vp, _u64, _u64, _u64],
    "agpu_synth_bits": [_vp, _vp, _u64, _u64, _u64, C.c_double],
    "agpu_checksum": [_vp, _vp, _u64, _vp],
}
_RESTYPES = {
    "agpu_abi_version": _i32,
    "agpu_last_error": C.c_char_p,
    "agpu_build_info": C.c_char_p,
    "agpu_dtype_size": _sz,
    "agpu_bitmap_bytes": _sz,
    "agpu_ipc_close": None,
    "agpu_ipc_writer_destroy": None,
}
# functions whose int result is NOT an agpu_status
_NOT_STATUS = set(_RESTYPES)


# ---- Arrow C Data Interface structs (include/arrow_gpu.h declares the same two, as the Arrow specification does)
class ArrowSchemaStruct(C.Structure):
    pass


class ArrowArrayStruct(C.Structure):
    pass


ArrowSchemaStruct._fields_ = [
    ("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64), ("n_children", C.c_int64),
    ("children", C.POINTER(C.POINTER(ArrowSchemaStruct))), ("dictionary", C.POINTER(ArrowSchemaStruct)),
    ("release", C.CFUNCTYPE(None, C.POINTER(ArrowSchemaStruct))), ("private_data", C.c_void_p)]
ArrowArrayStruct._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64), ("n_children", C.c_int64),
    ("buffers", C.POINTER(C.c_void_p)), ("children", C.POINTER(C.POINTER(ArrowArrayStruct))),
    ("dictionary", C.POINTER(ArrowArrayStruct)), ("release", C.CFUNCTYPE(None, C.POINTER(ArrowArrayStruct))),
    ("private_data", C.c_void_p)]


class ArrowArrayStreamStruct(C.Structure):  # struct ArrowArrayStream: five pointers
    _fields_ = [("get_schema", C.c_void_p), ("get_next", C.c_void_p), ("get_last_error", C.c_void_p),
                ("release", C.CFUNCTYPE(None, C.c_void_p)), ("private_data", C.c_void_p)]


class ArrowColumnStruct(C.Structure):  # agpu_arrow_column
    _fields_ = [("dtype", C.c_int32), ("length", C.c_uint64), ("null_count", C.c_int64), ("values", C.c_void_p),
                ("validity", C.c_void_p), ("values_bytes", C.c_uint64), ("validity_bytes", C.c_uint64)]


class IpcFieldStruct(C.Structure):  # agpu_ipc_field
    _fields_ = [("name", C.c_char_p), ("format", C.c_char_p), ("dtype", C.c_int32), ("nullable", C.c_int32)]


class ArrowErrorGPU(RuntimeError):
    """Mirror of `enum ArrowErrorGPU` (crates/array/src/lib.rs:11-14) plus device/runtime failures."""

    def __init__(self, kind: str, message: str, status: int = -1):
        super().__init__(f"{kind}: {message}")
        self.kind = kind
        self.status = status


class OperationNotSupported(ArrowErrorGPU):
    def __init__(self, message, status=ERR_UNSUPPORTED):
        super().__init__("OperationNotSupported", message, status)


class CastingNotSupported(ArrowErrorGPU):
    def __init__(self, message, status=ERR_UNSUPPORTED):
        super().__init__("CastingNotSupported", message, status)


_lib = None


def lib() -> C.CDLL:
    """Load libarrow_gpu_hip.so; fail loudly if it was not built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ArrowErrorGPU(
                "LibraryMissing",
                f"{LIB_PATH} not found: build it with `make -C arrow_gpu_amd/csrc` — there is no CPU fallback",
            )
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, _i32)
        _lib = handle
    return _lib


def last_error() -> str:
    return lib().agpu_last_error().decode()


def check(status: int, what: str = "") -> None:
    if status == OK:
        return
    msg = last_error()
    if status == ERR_UNSUPPORTED:
        raise OperationNotSupported(msg or what, status)
    kinds = {ERR_SHAPE: "ShapeError", ERR_HIP: "HipError", ERR_ARG: "ArgumentError", ERR_NO_DEVICE: "NoDevice"}
    raise ArrowErrorGPU(kinds.get(status, "Error"), f"{what}: {msg}", status)


def call(name: str, *args) -> None:
    """Call a status-returning entry point and raise on failure."""
    check(getattr(lib(), name)(*args), name)
