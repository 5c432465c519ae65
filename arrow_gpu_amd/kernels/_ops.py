"""Shared launch helpers for the op "traits" (the Python analogue of the reference's impl macros:
impl_arithmetic_op!/impl_arithmetic_array_op! crates/arithmetic/src/lib.rs:11-96, apply_function!
crates/compare/src/lib.rs:85-140, apply_unary_function_op! crates/math/src/lib.rs:138-193).

Every launcher: allocates the output buffer (the reference's apply_*_function allocates a fresh one per call),
enqueues ONE C-ABI kernel call on the pipeline's stream, derives the output validity with the reference's rules, and
returns a new array object without waiting.
"""
from __future__ import annotations

import ctypes as C
import weakref

from .. import _capi as capi
from .._capi import OperationNotSupported
from ..array import (ArrowArrayGPU, BooleanArrayGPU, NullBitBufferGpu, PrimitiveArrayGpu, bitmap_bytes)
from ..gpu_utils import ArrowComputePipeline


def vp(buf):
    return C.c_void_p(buf.ptr) if buf is not None else None


def default_impl(op_name: str):
    """`fn x(&self, ..) { let mut pipeline = new(); let r = self.x_op(.., &mut pipeline); pipeline.finish(); r }`"""

    def method(self, *args):
        pipeline = ArrowComputePipeline(self.get_gpu_device(), op_name)
        out = getattr(self, op_name + "_op")(*args, pipeline)
        pipeline.finish()
        return out

    method.__name__ = op_name
    return method


def impl(classes, name: str, op_fn) -> None:
    """`impl Trait for T`: attach `<name>_op` and the default `<name>` to each class."""
    for cls in classes:
        setattr(cls, name + "_op", op_fn)
        setattr(cls, name, default_impl(name))


def check_same_len(a, b, what: str) -> None:
    if a.len != b.len:
        raise capi.ArrowErrorGPU("ShapeError", f"{what}: arrays of different length {a.len} vs {b.len}", capi.ERR_SHAPE)


_FUSABLE_DTYPES = (capi.F32, capi.I32, capi.U32, capi.DATE32)
_FUSABLE_BINARY_F32 = (capi.OP_ADD, capi.OP_SUB, capi.OP_MUL, capi.OP_DIV, capi.OP_REM, capi.OP_MIN, capi.OP_MAX)
_FUSABLE_BINARY_INT = _FUSABLE_BINARY_F32 + (capi.OP_AND, capi.OP_OR, capi.OP_XOR)
_FUSABLE_UNARY_F32 = (capi.UN_NEG, capi.UN_ABS, capi.UN_SQRT, capi.UN_CBRT, capi.UN_EXP, capi.UN_EXP2, capi.UN_LOG, capi.UN_LOG2,
                      capi.UN_SIN, capi.UN_COS, capi.UN_ACOS, capi.UN_SINH)
_FUSABLE_UNARY_INT = (capi.UN_NEG, capi.UN_ABS, capi.UN_NOT)
_KIND_UNARY, _KIND_SCALAR, _KIND_ARRAY = 0, 1, 2


def _recordable(pipeline, kind: int, op: int, dtype: int, same_class: bool) -> bool:
    """Can this op wait in a fusing pipeline (ArrowComputePipeline(fuse=True)) instead of launching now?"""
    if not getattr(pipeline, "fuse", False) or not same_class or dtype not in _FUSABLE_DTYPES:
        return False
    is_f = dtype == capi.F32
    if kind == _KIND_UNARY:
        return op in (_FUSABLE_UNARY_F32 if is_f else _FUSABLE_UNARY_INT)
    return op in (_FUSABLE_BINARY_F32 if is_f else _FUSABLE_BINARY_INT)


def _finish_record(node, result):
    """Tie a recorded node to the array object handed to the caller: when that object is gone by flush time the
    intermediate is dead and the chain may skip materialising it."""
    if node is not None:
        node.out_ref = weakref.ref(result)
    return result


def binary_values(pipeline, op: int, dtype: int, a, b, out_cls, n: int):
    dev = a.gpu_device
    out = dev.create_empty_buffer(max(n * out_cls.ITEM_SIZE, 1), like=(a.data, b.data))
    capi.call("agpu_binary", pipeline._handle, op, dtype, vp(a.data), vp(b.data), vp(out), n)
    pipeline.keep(a.data, b.data, out)
    return out


def array_op(op: int, out_cls=None):
    """array ∘ array, validity = AND  [impl_arithmetic_array_op!, apply_function_min_max!, Logical binary]"""

    def fn(self, value, pipeline: ArrowComputePipeline):
        check_same_len(self, value, "binary op")
        cls = out_cls or type(self)
        node = None
        if _recordable(pipeline, _KIND_ARRAY, op, self.DTYPE, cls is type(self) and value.ITEM_SIZE == self.ITEM_SIZE):
            out = self.gpu_device.create_empty_buffer(max(self.len * cls.ITEM_SIZE, 1), like=(self.data, value.data))
            node = pipeline.record_elementwise(_KIND_ARRAY, op, self.DTYPE, self.data, value.data, out, self.len)
        else:
            out = binary_values(pipeline, op, self.DTYPE, self, value, cls, self.len)
        nulls = NullBitBufferGpu.merge_null_bit_buffer_op(self.null_buffer, value.null_buffer, pipeline)
        return _finish_record(node, cls(out, self.gpu_device, self.len, nulls))

    return fn


def scalar_op(op: int):
    """array ∘ 1-element array, validity cloned  [impl_arithmetic_op!]"""

    def fn(self, value, pipeline: ArrowComputePipeline):
        dev = self.gpu_device
        out = dev.create_empty_buffer(max(self.len * self.ITEM_SIZE, 1), like=(self.data,))
        node = None
        if _recordable(pipeline, _KIND_SCALAR, op, self.DTYPE, value.ITEM_SIZE == self.ITEM_SIZE):
            node = pipeline.record_elementwise(_KIND_SCALAR, op, self.DTYPE, self.data, value.data, out, self.len)
        else:
            capi.call("agpu_scalar", pipeline._handle, op, self.DTYPE, vp(self.data), vp(value.data), vp(out), self.len)
            pipeline.keep(self.data, value.data, out)
        nulls = NullBitBufferGpu.clone_null_bit_buffer_op(self.null_buffer, pipeline)
        return _finish_record(node, type(self)(out, dev, self.len, nulls))

    return fn


def unary_op(op: int, out_cls=None):
    """unary map, validity cloned  [apply_unary_function_op!, Neg, bitwise_not]"""

    def fn(self, pipeline: ArrowComputePipeline):
        dev = self.gpu_device
        cls = out_cls or type(self)
        out = dev.create_empty_buffer(max(self.len * cls.ITEM_SIZE, 1), like=(self.data,))
        node = None
        if _recordable(pipeline, _KIND_UNARY, op, self.DTYPE, cls is type(self)):
            node = pipeline.record_elementwise(_KIND_UNARY, op, self.DTYPE, self.data, None, out, self.len)
        else:
            capi.call("agpu_unary", pipeline._handle, op, self.DTYPE, vp(self.data), vp(out), self.len)
            pipeline.keep(self.data, out)
        nulls = NullBitBufferGpu.clone_null_bit_buffer_op(self.null_buffer, pipeline)
        return _finish_record(node, cls(out, dev, self.len, nulls))

    return fn


def dyn_binary(name: str, op_attr: str, same_type, mixed=()):
    """dyn_fn!: match on (variant, variant); panic → OperationNotSupported."""
    same = tuple(same_type)
    mixed = tuple(mixed)

    def fn_op(data_1: ArrowArrayGPU, data_2: ArrowArrayGPU, pipeline: ArrowComputePipeline):
        t1, t2 = type(data_1), type(data_2)
        if (t1 is t2 and t1 in same) or (t1, t2) in mixed:
            return getattr(data_1, op_attr)(data_2, pipeline)
        raise OperationNotSupported(
            f"Operation {name} not supported for type {data_1.get_dtype().name} {data_2.get_dtype().name}")

    def fn(data_1, data_2):
        pipeline = ArrowComputePipeline(data_1.get_gpu_device(), name)
        out = fn_op(data_1, data_2, pipeline)
        pipeline.finish()
        return out

    fn.__name__ = name
    fn_op.__name__ = name.replace("_dyn", "_op_dyn")
    return fn, fn_op


def dyn_unary(name: str, op_attr: str, types):
    types = tuple(types)

    def fn_op(data: ArrowArrayGPU, pipeline: ArrowComputePipeline):
        if type(data) in types:
            return getattr(data, op_attr)(pipeline)
        raise OperationNotSupported(f"Operation {name} not supported for type {data.get_dtype().name}")

    def fn(data):
        pipeline = ArrowComputePipeline(data.get_gpu_device(), name)
        out = fn_op(data, pipeline)
        pipeline.finish()
        return out

    fn.__name__ = name
    fn_op.__name__ = name.replace("_dyn", "_op_dyn")
    return fn, fn_op


__all__ = ["vp", "impl", "default_impl", "array_op", "scalar_op", "unary_op", "dyn_binary", "dyn_unary",
           "binary_values", "check_same_len", "bitmap_bytes", "BooleanArrayGPU", "PrimitiveArrayGpu"]
