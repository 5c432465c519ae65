"""CPU oracle for the hot path — TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``oracle/agpu_oracle.c`` (the plain-C restatement of psvri/arrow-gpu's WGSL kernels) plus
small numpy helpers that mirror the reference's host-side array construction
(``PrimitiveArrayGpu::from_optional_slice`` crates/array/src/array/primitive_array_gpu.rs:22-53,
``BooleanArrayGPU::from_optional_slice`` crates/array/src/array/boolean_gpu.rs:24-51).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.
The product package (``arrow_gpu_amd``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

# enum mirrors of include/arrow_gpu.h
BOOL, F32, U32, U16, U8, I32, I16, I8, DATE32 = range(9)
OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_REM, OP_MIN, OP_MAX, OP_AND, OP_OR, OP_XOR, OP_SHL, OP_SHR, OP_POW = range(13)
(UN_NEG, UN_ABS, UN_NOT, UN_SQRT, UN_CBRT, UN_EXP, UN_EXP2, UN_LOG, UN_LOG2, UN_SIN, UN_COS, UN_ACOS, UN_SINH,
 UN_POPCOUNT) = range(14)
CMP_GT, CMP_GTEQ, CMP_LT, CMP_LTEQ, CMP_EQ = range(5)
RED_SUM, RED_MIN, RED_MAX = range(3)

NP_DTYPE = {
    F32: np.float32, U32: np.uint32, U16: np.uint16, U8: np.uint8,
    I32: np.int32, I16: np.int16, I8: np.int8, DATE32: np.int32,
}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile). Returns the path of liboracle.so.
    ORACLE_LIB=<path> loads another build instead (the sanitizer job: oracle/Makefile `asan`)."""
    if os.environ.get("ORACLE_LIB"):
        return os.environ["ORACLE_LIB"]
    so = os.path.join(_BUILD, "liboracle.so")
    src = os.path.join(_HERE, "agpu_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "_build/liboracle.so"], check=True, capture_output=True)
    return so


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_bitmap_bytes.restype = C.c_size_t
        _lib.orc_bitmap_bytes.argtypes = [C.c_uint64]
        _lib.orc_dtype_size.restype = C.c_size_t
    return _lib


def _p(a):
    if a is None:
        return C.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def _chk(rc, what):
    if rc != 0:
        raise OracleError(rc, what)


class OracleError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"oracle {what}: status {code}")
        self.code = code


def bitmap_bytes(n_bits: int) -> int:
    return (n_bits + 63) // 64 * 8


def pack_bits(bools) -> np.ndarray:
    """LSB-first bitmap padded to a multiple of 8 bytes (BooleanBufferBuilder, null_bit_buffer.rs:21-61)."""
    b = np.asarray(bools, dtype=bool)
    out = np.zeros(bitmap_bytes(len(b)), dtype=np.uint8)
    packed = np.packbits(b, bitorder="little")
    out[: len(packed)] = packed
    return out


def unpack_bits(bitmap: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(np.asarray(bitmap, dtype=np.uint8), bitorder="little")[:n].astype(bool)


def from_optional(values, dtype):
    """(data, validity) like from_optional_slice: null slots hold default()=0; validity always materialised."""
    valid = np.array([v is not None for v in values], dtype=bool)
    if dtype == BOOL:
        data = pack_bits([bool(v) if v is not None else False for v in values])
    else:
        data = np.array([v if v is not None else 0 for v in values]).astype(NP_DTYPE[dtype])
    return data, pack_bits(valid)


# ---------------------------------------------------------------- kernels
def binary(op, dtype, a, b):
    out_dt = NP_DTYPE[dtype]
    a = np.ascontiguousarray(a, dtype=out_dt)
    b = np.ascontiguousarray(b, dtype=np.uint32 if op in (OP_SHL, OP_SHR) else out_dt)
    out = np.empty(len(a), dtype=out_dt)
    _chk(lib().orc_binary(op, dtype, _p(a), _p(b), _p(out), C.c_uint64(len(a))), "binary")
    return out


def scalar(op, dtype, a, s):
    out_dt = NP_DTYPE[dtype]
    a = np.ascontiguousarray(a, dtype=out_dt)
    s = np.ascontiguousarray(np.atleast_1d(s), dtype=np.uint32 if op in (OP_SHL, OP_SHR) else out_dt)
    out = np.empty(len(a), dtype=out_dt)
    _chk(lib().orc_scalar(op, dtype, _p(a), _p(s), _p(out), C.c_uint64(len(a))), "scalar")
    return out


def unary(op, dtype, a):
    in_dt = NP_DTYPE[dtype]
    a = np.ascontiguousarray(a, dtype=in_dt)
    fused = dtype in (U8, I8, U16, I16) and op in (UN_SIN, UN_COS, UN_SINH)
    out = np.empty(len(a), dtype=np.float32 if fused else in_dt)
    _chk(lib().orc_unary(op, dtype, _p(a), _p(out), C.c_uint64(len(a))), "unary")
    return out


def cast(frm, to, a, n=None):
    if frm == BOOL:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        assert n is not None
    else:
        a = np.ascontiguousarray(a, dtype=NP_DTYPE[frm])
        n = len(a)
    out = np.empty(n, dtype=NP_DTYPE[to])
    _chk(lib().orc_cast(frm, to, _p(a), _p(out), C.c_uint64(n)), "cast")
    return out


def bitcast(frm, to, a):
    a = np.ascontiguousarray(a, dtype=NP_DTYPE[frm])
    out = np.empty(len(a), dtype=NP_DTYPE[to])
    _chk(lib().orc_bitcast(frm, to, _p(a), _p(out), C.c_uint64(len(a))), "bitcast")
    return out


def broadcast(dtype, value, n):
    if dtype == BOOL:
        out = np.empty(bitmap_bytes(n), dtype=np.uint8)
        bits = 1 if value else 0
    else:
        out = np.empty(n, dtype=NP_DTYPE[dtype])
        raw = np.zeros(4, dtype=np.uint8)
        vb = np.array([value]).astype(NP_DTYPE[dtype]).view(np.uint8)
        raw[: len(vb)] = vb
        bits = int(raw.view(np.uint32)[0])
    _chk(lib().orc_broadcast(dtype, C.c_uint32(bits), _p(out), C.c_uint64(n)), "broadcast")
    return out


def compare(op, dtype, a, b):
    a = np.ascontiguousarray(a, dtype=NP_DTYPE[dtype])
    b = np.ascontiguousarray(b, dtype=NP_DTYPE[dtype])
    out = np.empty(bitmap_bytes(len(a)), dtype=np.uint8)
    _chk(lib().orc_compare(op, dtype, _p(a), _p(b), _p(out), C.c_uint64(len(a))), "compare")
    return out


def bitmap_binary(op, a, b, n_bits):
    out = np.empty(bitmap_bytes(n_bits), dtype=np.uint8)
    _chk(lib().orc_bitmap_binary(op, _p(a), _p(b), _p(out), C.c_uint64(n_bits)), "bitmap_binary")
    return out


def bitmap_copy_bits(src, bit_offset, n_bits):
    out = np.empty(bitmap_bytes(n_bits), dtype=np.uint8)
    _chk(lib().orc_bitmap_copy_bits(_p(src), C.c_uint64(bit_offset), _p(out), C.c_uint64(n_bits)), "bitmap_copy_bits")
    return out


def bitmap_not(a, n_bits):
    out = np.empty(bitmap_bytes(n_bits), dtype=np.uint8)
    _chk(lib().orc_bitmap_not(_p(a), _p(out), C.c_uint64(n_bits)), "bitmap_not")
    return out


def bitmap_popcount(a, n_bits) -> int:
    out = C.c_uint64(0)
    _chk(lib().orc_bitmap_popcount(_p(a), C.c_uint64(n_bits), C.byref(out)), "popcount")
    return out.value


def bitmap_any(a, n_bits) -> bool:
    out = C.c_uint32(0)
    _chk(lib().orc_bitmap_any(_p(a), C.c_uint64(n_bits), C.byref(out)), "any")
    return bool(out.value)


def merge_validity(va, vb, mask, vmask, n_bits):
    """NullBitBufferGpu::merge_null_bit_buffer semantics are in validity_and(); this is the Swizzle::merge validity."""
    out = np.empty(bitmap_bytes(n_bits), dtype=np.uint8)
    _chk(lib().orc_bitmap_merge_validity(_p(va), _p(vb), _p(mask), _p(vmask), _p(out), C.c_uint64(n_bits)), "merge_validity")
    return out


def validity_and(va, vb, n_bits):
    """merge_null_bit_buffer (null_bit_buffer.rs:168-204): (None,None)→None; one side → copy; both → AND."""
    if va is None and vb is None:
        return None
    if va is None:
        return vb.copy()
    if vb is None:
        return va.copy()
    return bitmap_binary(OP_AND, va, vb, n_bits)


def reduce(op, dtype, a, validity=None):
    a = np.ascontiguousarray(a, dtype=NP_DTYPE[dtype])
    out = np.empty(1, dtype=NP_DTYPE[dtype])
    _chk(lib().orc_reduce(op, dtype, _p(a), _p(validity), C.c_uint64(len(a)), _p(out)), "reduce")
    return out[0]


def reduce_sum_f64(a, validity=None) -> float:
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = C.c_double(0)
    _chk(lib().orc_reduce_sum_f64(_p(a), _p(validity), C.c_uint64(len(a)), C.byref(out)), "reduce_sum_f64")
    return out.value


def take(width, values, idx):
    values = np.ascontiguousarray(values)
    assert values.dtype.itemsize == width
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    out = np.empty(len(idx), dtype=values.dtype)
    _chk(lib().orc_take(width, _p(values), C.c_uint64(len(values)), _p(idx), _p(out), C.c_uint64(len(idx))), "take")
    return out


def take_bits(bits, n_bits, idx):
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    out = np.empty(bitmap_bytes(len(idx)), dtype=np.uint8)
    _chk(lib().orc_take_bits(_p(bits), C.c_uint64(n_bits), _p(idx), _p(out), C.c_uint64(len(idx))), "take_bits")
    return out


def put(width, src, src_idx, dst, dst_idx):
    """Returns the updated copy of dst."""
    src = np.ascontiguousarray(src)
    dst = np.array(dst, copy=True)
    assert src.dtype.itemsize == width and dst.dtype.itemsize == width
    si = np.ascontiguousarray(src_idx, dtype=np.uint32)
    di = np.ascontiguousarray(dst_idx, dtype=np.uint32)
    _chk(lib().orc_put(width, _p(src), _p(si), _p(dst), _p(di), C.c_uint64(len(si))), "put")
    return dst


def put_bits(src_bits, src_idx, dst_bits, dst_idx):
    dst = np.array(dst_bits, copy=True)
    si = np.ascontiguousarray(src_idx, dtype=np.uint32)
    di = np.ascontiguousarray(dst_idx, dtype=np.uint32)
    _chk(lib().orc_put_bits(_p(src_bits), _p(si), _p(dst), _p(di), C.c_uint64(len(si))), "put_bits")
    return dst


def merge(width, a, b, mask_bits):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b, dtype=a.dtype)
    out = np.empty(len(a), dtype=a.dtype)
    _chk(lib().orc_merge(width, _p(a), _p(b), _p(mask_bits), _p(out), C.c_uint64(len(a))), "merge")
    return out


def merge_bits(a, b, mask_bits, n_bits):
    out = np.empty(bitmap_bytes(n_bits), dtype=np.uint8)
    _chk(lib().orc_merge_bits(_p(a), _p(b), _p(mask_bits), _p(out), C.c_uint64(n_bits)), "merge_bits")
    return out


def index_max(idx) -> int:
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    out = C.c_uint32(0)
    _chk(lib().orc_index_max(_p(idx), C.c_uint64(len(idx)), C.byref(out)), "index_max")
    return out.value


# ---------------------------------------------------------------- synthetic columns (shared generator)
def synth_f32(n, seed, row0=0, lo=-1000.0, hi=1000.0):
    out = np.empty(n, dtype=np.float32)
    _chk(lib().orc_synth_f32(_p(out), C.c_uint64(n), C.c_uint64(seed), C.c_uint64(row0), C.c_float(lo), C.c_float(hi)), "synth_f32")
    return out


def synth_i32(n, seed, row0=0, modulus=0):
    out = np.empty(n, dtype=np.int32)
    _chk(lib().orc_synth_i32(_p(out), C.c_uint64(n), C.c_uint64(seed), C.c_uint64(row0), C.c_uint32(modulus)), "synth_i32")
    return out


def synth_u8(n, seed, row0=0):
    out = np.empty(n, dtype=np.uint8)
    _chk(lib().orc_synth_u8(_p(out), C.c_uint64(n), C.c_uint64(seed), C.c_uint64(row0)), "synth_u8")
    return out


def synth_bits(n_bits, seed, row0=0, p_set=0.9):
    out = np.empty(bitmap_bytes(n_bits), dtype=np.uint8)
    _chk(lib().orc_synth_bits(_p(out), C.c_uint64(n_bits), C.c_uint64(seed), C.c_uint64(row0), C.c_double(p_set)), "synth_bits")
    return out


def checksum(arr) -> int:
    a = np.ascontiguousarray(arr)
    out = C.c_uint64(0)
    _chk(lib().orc_checksum(_p(a), C.c_uint64(a.nbytes), C.byref(out)), "checksum")
    return out.value


def sharded_reduce(op, dtype, shards, validities=None):
    """Spec of the multi-GPU final reduce (include/arrow_gpu.h agpu_comm_reduce; nothing in the reference): every shard
    is reduced on its own (the reference's tree order for f32 Sum), then the per-shard results are combined IN RANK
    ORDER — f32 Sum by one more adjacent-pair tree level over the shard sums [aggregate.wgsl:21-41, i.e. `reduce` of
    the vector of shard sums], integer sums wrapping, min/max with Arrow semantics (NaN ignored unless all NaN); an
    empty shard contributes the identity."""
    npd = NP_DTYPE[dtype]
    validities = validities or [None] * len(shards)
    parts = [(reduce(op, dtype, s, v), len(s)) for s, v in zip(shards, validities)]
    if op == RED_SUM:
        vec = np.array([p for p, _ in parts], dtype=npd)
        return reduce(RED_SUM, dtype, vec)
    live = np.array([p for p, n in parts if n > 0], dtype=npd)
    return reduce(op, dtype, live)


def sharded_reduce_sum_f64(shards, validities=None) -> float:
    validities = validities or [None] * len(shards)
    acc = 0.0
    for s, v in zip(shards, validities):
        if len(s):
            acc = acc + reduce_sum_f64(s, v)
    return acc


def combine_records(op, dtype, records):
    """The rank-ordered combine of the multi-GPU final reduce on the gathered records [(statistic, n_local), …] — what
    comm_finish_*_kernel (arrow_gpu_amd/csrc/reduce.hip) does after the all-gather.  Same rule as sharded_reduce."""
    npd = NP_DTYPE[dtype]
    if op == RED_SUM:
        return reduce(RED_SUM, dtype, np.array([v for v, _ in records], dtype=npd))
    return reduce(op, dtype, np.array([v for v, n in records if n > 0], dtype=npd))
