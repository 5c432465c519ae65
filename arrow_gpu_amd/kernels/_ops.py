"""Shared launch helpers for the op "traits" (the Python analogue of the reference's impl macros:
impl_arithmetic_op!/impl_arithmetic_array_op! crates/arithmetic/src/lib.rs:11-96, apply_function!
crates/compare/src/lib.rs:85-140, apply_unary_function_op! crates/math/src/lib.rs:138-193).

Every launcher: allocates the output buffer (the reference's apply_*_function allocates a fresh one per call),
enqueues ONE C-ABI kernel call on the pipeline's stream, derives the output validity with the reference's rules, and
returns a new array object without waiting.
"""
from __future__ import annotations

import ctypes as C

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


def binary_values(pipeline, op: int, dtype: int, a, b, out_cls, n: int):
    dev = a.gpu_device
    out = dev.create_empty_buffer(max(n * out_cls.ITEM_SIZE, 1))
    capi.call("agpu_binary", pipeline._handle, op, dtype, vp(a.data), vp(b.data), vp(out), n)
    pipeline.keep(a.data, b.data, out)
    return out


def array_op(op: int, out_cls=None):
    """array ∘ array, validity = AND  [impl_arithmetic_array_op!, apply_function_min_max!, Logical binary]"""

    def fn(self, value, pipeline: ArrowComputePipeline):
        check_same_len(self, value, "binary op")
        cls = out_cls or type(self)
        out = binary_values(pipeline, op, self.DTYPE, self, value, cls, self.len)
        nulls = NullBitBufferGpu.merge_null_bit_buffer_op(self.null_buffer, value.null_buffer, pipeline)
        return cls(out, self.gpu_device, self.len, nulls)

    return fn


def scalar_op(op: int):
    """array ∘ 1-element array, validity cloned  [impl_arithmetic_op!]"""

    def fn(self, value, pipeline: ArrowComputePipeline):
        dev = self.gpu_device
        out = dev.create_empty_buffer(max(self.len * self.ITEM_SIZE, 1))
        capi.call("agpu_scalar", pipeline._handle, op, self.DTYPE, vp(self.data), vp(value.data), vp(out), self.len)
        pipeline.keep(self.data, value.data, out)
        nulls = NullBitBufferGpu.clone_null_bit_buffer_op(self.null_buffer, pipeline)
        return type(self)(out, dev, self.len, nulls)

    return fn


def unary_op(op: int, out_cls=None):
    """unary map, validity cloned  [apply_unary_function_op!, Neg, bitwise_not]"""

    def fn(self, pipeline: ArrowComputePipeline):
        dev = self.gpu_device
        cls = out_cls or type(self)
        out = dev.create_empty_buffer(max(self.len * cls.ITEM_SIZE, 1))
        capi.call("agpu_unary", pipeline._handle, op, self.DTYPE, vp(self.data), vp(out), self.len)
        pipeline.keep(self.data, out)
        nulls = NullBitBufferGpu.clone_null_bit_buffer_op(self.null_buffer, pipeline)
        return cls(out, dev, self.len, nulls)

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
