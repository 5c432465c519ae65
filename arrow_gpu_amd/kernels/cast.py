"""arrow_gpu_cast: Cast<T>, BitCast<T>, cast_dyn, bitcast_dyn.

Mirror of crates/cast/src/lib.rs (traits :15-38, impl_cast! :40-87, table :135-161, bitcast :187-192).  In Rust the
target is a type parameter (`<UInt8ArrayGPU as Cast<Float32ArrayGPU>>::cast(&a)`); here it is an argument:
`a.cast(Float32ArrayGPU)` / `a.cast_op(Float32ArrayGPU, pipeline)`.
"""
from __future__ import annotations

from .. import _capi as capi
from .._capi import CastingNotSupported
from ..array import (ARRAY_OF_TYPE, ArrowArrayGPU, ArrowType, BooleanArrayGPU, Float32ArrayGPU, Int8ArrayGPU,
                     Int16ArrayGPU, Int32ArrayGPU, NullBitBufferGpu, UInt8ArrayGPU, UInt16ArrayGPU, UInt32ArrayGPU)
from ..gpu_utils import ArrowComputePipeline
from ._ops import _finish_record, vp

# cast_dyn's table [crates/cast/src/lib.rs:135-161]
CAST_TABLE = {
    Int8ArrayGPU: (UInt8ArrayGPU, UInt16ArrayGPU, UInt32ArrayGPU, Int16ArrayGPU, Int32ArrayGPU, Float32ArrayGPU),
    Int16ArrayGPU: (Int32ArrayGPU, UInt16ArrayGPU, UInt32ArrayGPU, Float32ArrayGPU),
    UInt8ArrayGPU: (UInt16ArrayGPU, UInt32ArrayGPU, Int8ArrayGPU, Int16ArrayGPU, Int32ArrayGPU, Float32ArrayGPU),
    UInt16ArrayGPU: (UInt32ArrayGPU, Int16ArrayGPU, Int32ArrayGPU, Float32ArrayGPU),
    # f32 → u8 is the reference's only narrowing cast [cast/src/f32_cast.rs:8-31]; the other five are REFERENCE-ABSENT
    # (north_star "i8/i16/u8/u16 <-> f32"), defined by analogy with cast_u8.wgsl — see include/arrow_gpu.h agpu_cast
    Float32ArrayGPU: (UInt8ArrayGPU, Int8ArrayGPU, Int16ArrayGPU, UInt16ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU),
    BooleanArrayGPU: (Float32ArrayGPU,),
}
BITCAST_TABLE = {UInt32ArrayGPU: (Float32ArrayGPU,)}
_WIDENING_HEADS = (UInt8ArrayGPU, Int8ArrayGPU, UInt16ArrayGPU, Int16ArrayGPU)  # casts a fusing pipeline may record


def _cast_op(self: ArrowArrayGPU, into, pipeline: ArrowComputePipeline):
    if into not in CAST_TABLE.get(type(self), ()):
        raise CastingNotSupported(f"Casting not supported for type {self.get_dtype().name} {into.ARROW_TYPE.name}")
    dev = self.gpu_device
    out = dev.create_empty_buffer(max(self.len * into.ITEM_SIZE, 1))
    node = None
    if getattr(pipeline, "fuse", False) and into is Float32ArrayGPU and type(self) in _WIDENING_HEADS:
        # a fusing pipeline records the widening cast: `cast_op → sin_op`, `cast_op → mul_scalar_op → …` become ONE launch at
        # finish() (agpu_fused_cast_chain) when the caller dropped the f32 intermediate  [SURVEY §8f-2: "cast → sin"]
        node = pipeline.record_cast(self.DTYPE, self.data, out, self.len)
    else:
        capi.call("agpu_cast", pipeline._handle, self.DTYPE, into.DTYPE, vp(self.data), vp(out), self.len)
        pipeline.keep(self.data, out)
    nulls = NullBitBufferGpu.clone_null_bit_buffer_op(self.null_buffer, pipeline)
    return _finish_record(node, into(out, dev, self.len, nulls))


def apply_boolean_unary_function(gpu_device, original_values, new_buffer_size: int, output_item_size: int, shader: str, entry_point: str,
                                 pipeline: ArrowComputePipeline):
    """`cast::apply_boolean_unary_function(gpu_device, original_values, new_buffer_size, output_item_size, shader, entry_point,
    pipeline) -> Buffer` [ref: crates/cast/src/boolean_cast.rs:8-55]: a Boolean bitmap in, `new_buffer_size / output_item_size`
    elements out, one invocation per OUTPUT element — the reference's literal call shape, served by the by-name seam
    (`agpu_launch_by_name_sized`: `shader` is the WGSL text, its `#hash:len` name or its path key, e.g. "cast/boolean/cast_f32")."""
    import ctypes as C

    out = gpu_device.create_empty_buffer(max(int(new_buffer_size), 1), zero_fill=True)
    dispatch = -(-(-(-int(new_buffer_size) // int(output_item_size))) // 256)  # ceil(ceil(size / item) / 256)
    ins = (C.c_void_p * 1)(original_values.ptr)
    sizes = (C.c_uint64 * 1)(original_values.nbytes)
    key = shader.encode() if isinstance(shader, str) else shader
    capi.call("agpu_launch_by_name_sized", pipeline._handle, key, entry_point.encode(), ins, sizes, 1, vp(out), int(new_buffer_size), dispatch)
    pipeline.keep(original_values, out)
    return out


def _cast(self, into):
    p = ArrowComputePipeline(self.get_gpu_device(), "cast")
    out = _cast_op(self, into, p)
    p.finish()
    return out


def _bitcast_op(self, into, pipeline: ArrowComputePipeline):
    """Reinterpret: a device copy of the buffer [crates/cast/src/lib.rs:90-107]."""
    if into not in BITCAST_TABLE.get(type(self), ()):
        raise CastingNotSupported(f"Casting not supported for type {self.get_dtype().name} {into.ARROW_TYPE.name}")
    data = pipeline.clone_buffer(self.data)
    return into(data, self.gpu_device, self.len, NullBitBufferGpu.clone_null_bit_buffer_op(self.null_buffer, pipeline))


def _bitcast(self, into):
    p = ArrowComputePipeline(self.get_gpu_device(), "bitcast")
    out = _bitcast_op(self, into, p)
    p.finish()
    return out


for _cls in CAST_TABLE:
    _cls.cast = _cast
    _cls.cast_op = _cast_op
for _cls in BITCAST_TABLE:
    _cls.bitcast = _bitcast
    _cls.bitcast_op = _bitcast_op


def cast_op_dyn(from_: ArrowArrayGPU, into: ArrowType, pipeline: ArrowComputePipeline) -> ArrowArrayGPU:
    target = ARRAY_OF_TYPE[into]
    if target not in CAST_TABLE.get(type(from_), ()):
        raise CastingNotSupported(f"Casting not supported for type {from_.get_dtype().name} {into.name}")
    return from_.cast_op(target, pipeline)


def cast_dyn(from_: ArrowArrayGPU, into: ArrowType) -> ArrowArrayGPU:
    p = ArrowComputePipeline(from_.get_gpu_device(), "cast_dyn")
    out = cast_op_dyn(from_, into, p)
    p.finish()
    return out


def bitcast_op_dyn(from_: ArrowArrayGPU, into: ArrowType, pipeline: ArrowComputePipeline) -> ArrowArrayGPU:
    target = ARRAY_OF_TYPE[into]
    if target not in BITCAST_TABLE.get(type(from_), ()):
        raise CastingNotSupported(f"Casting not supported for type {from_.get_dtype().name} {into.name}")
    return from_.bitcast_op(target, pipeline)


def bitcast_dyn(from_: ArrowArrayGPU, into: ArrowType) -> ArrowArrayGPU:
    p = ArrowComputePipeline(from_.get_gpu_device(), "bitcast_dyn")
    out = bitcast_op_dyn(from_, into, p)
    p.finish()
    return out


__all__ = ["cast_dyn", "cast_op_dyn", "bitcast_dyn", "bitcast_op_dyn", "apply_boolean_unary_function", "CAST_TABLE", "BITCAST_TABLE"]
