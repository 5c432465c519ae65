"""arrow_gpu_logical: bitwise and/or/xor/not, shl/shr, Boolean any/all.

Mirror of crates/logical/src/{lib,boolean}.rs.  Integer arrays use the element-wise kernels; BooleanArrayGPU runs
the bitmap kernels (the reference reuses u32/logical.wgsl on the packed words, boolean.rs:12-75).  Validity: AND
for binary ops (not Kleene logic), clone for `not` and — as in the reference's apply_binary_function_op! — AND with
the u32 shift-amount array's validity for shifts.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _capi as capi
from ..array import (BooleanArrayGPU, Int8ArrayGPU, Int16ArrayGPU, Int32ArrayGPU, NullBitBufferGpu, UInt8ArrayGPU,
                     UInt16ArrayGPU, UInt32ArrayGPU, bitmap_bytes)
from ..gpu_utils import ArrowComputePipeline
from ._ops import array_op, check_same_len, dyn_binary, dyn_unary, impl, unary_op, vp

_INTS = (Int32ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU, Int16ArrayGPU, UInt8ArrayGPU, Int8ArrayGPU)

impl(_INTS, "bitwise_and", array_op(capi.OP_AND))
impl(_INTS, "bitwise_or", array_op(capi.OP_OR))
impl(_INTS, "bitwise_xor", array_op(capi.OP_XOR))
impl(_INTS, "bitwise_not", unary_op(capi.UN_NOT))
impl(_INTS, "bitwise_shl", array_op(capi.OP_SHL))  # operand: UInt32ArrayGPU of shift amounts
impl(_INTS, "bitwise_shr", array_op(capi.OP_SHR))


def _bool_binary(op: int):
    def fn(self: BooleanArrayGPU, operand: BooleanArrayGPU, pipeline: ArrowComputePipeline):
        check_same_len(self, operand, "boolean logical op")
        dev = self.gpu_device
        out = dev.create_empty_buffer(max(bitmap_bytes(self.len), 8))
        capi.call("agpu_bitmap_binary", pipeline._handle, op, vp(self.data), vp(operand.data), vp(out), self.len)
        pipeline.keep(self.data, operand.data, out)
        nulls = NullBitBufferGpu.merge_null_bit_buffer_op(self.null_buffer, operand.null_buffer, pipeline)
        return BooleanArrayGPU(out, dev, self.len, nulls)

    return fn


def _bool_not(self: BooleanArrayGPU, pipeline: ArrowComputePipeline):
    dev = self.gpu_device
    out = dev.create_empty_buffer(max(bitmap_bytes(self.len), 8))
    capi.call("agpu_bitmap_not", pipeline._handle, vp(self.data), vp(out), self.len)
    pipeline.keep(self.data, out)
    return BooleanArrayGPU(out, dev, self.len, NullBitBufferGpu.clone_null_bit_buffer_op(self.null_buffer, pipeline))


impl((BooleanArrayGPU,), "bitwise_and", _bool_binary(capi.OP_AND))
impl((BooleanArrayGPU,), "bitwise_or", _bool_binary(capi.OP_OR))
impl((BooleanArrayGPU,), "bitwise_xor", _bool_binary(capi.OP_XOR))
impl((BooleanArrayGPU,), "bitwise_not", _bool_not)


# ---- LogicalContains  [boolean.rs:106-147]  (blocking: they return a host bool, like the reference)
def _any(self: BooleanArrayGPU) -> bool:
    dev = self.gpu_device
    p = ArrowComputePipeline(dev, "any")
    out = dev.create_empty_buffer(16)
    capi.call("agpu_bitmap_any", p._handle, vp(self.data), self.len, vp(out))
    res = dev.retrive_data(out, 4, pipeline=p).view(np.uint32)[0]
    return bool(res > 0)


def _all(self: BooleanArrayGPU) -> bool:
    """popcount(first len bits) == len.  (The reference counts whole words including padding bits, boolean.rs:120-146;
    padding is unspecified there, so only the first `len` bits are counted here.)"""
    dev = self.gpu_device
    p = ArrowComputePipeline(dev, "all")
    out = dev.create_empty_buffer(16)
    capi.call("agpu_bitmap_popcount", p._handle, vp(self.data), self.len, vp(out))
    res = dev.retrive_data(out, 8, pipeline=p).view(np.uint64)[0]
    return int(res) == self.len


def _count_set_bits(self: BooleanArrayGPU) -> int:
    dev = self.gpu_device
    p = ArrowComputePipeline(dev, "popcount")
    out = dev.create_empty_buffer(16)
    capi.call("agpu_bitmap_popcount", p._handle, vp(self.data), self.len, vp(out))
    return int(dev.retrive_data(out, 8, pipeline=p).view(np.uint64)[0])


BooleanArrayGPU.any = _any
BooleanArrayGPU.all = _all
BooleanArrayGPU.count_set_bits = _count_set_bits

_LOGICAL = _INTS + (BooleanArrayGPU,)
bitwise_and_dyn, bitwise_and_op_dyn = dyn_binary("bitwise_and_dyn", "bitwise_and_op", _LOGICAL)
bitwise_or_dyn, bitwise_or_op_dyn = dyn_binary("bitwise_or_dyn", "bitwise_or_op", _LOGICAL)
bitwise_xor_dyn, bitwise_xor_op_dyn = dyn_binary("bitwise_xor_dyn", "bitwise_xor_op", _LOGICAL)
bitwise_not_dyn, bitwise_not_op_dyn = dyn_unary("bitwise_not_dyn", "bitwise_not_op", _LOGICAL)
_SH = tuple((t, UInt32ArrayGPU) for t in _INTS if t is not UInt32ArrayGPU)
bitwise_shl_dyn, bitwise_shl_op_dyn = dyn_binary("bitwise_shl_dyn", "bitwise_shl_op", (UInt32ArrayGPU,), _SH)
bitwise_shr_dyn, bitwise_shr_op_dyn = dyn_binary("bitwise_shr_dyn", "bitwise_shr_op", (UInt32ArrayGPU,), _SH)

__all__ = [n for n in dir() if n.endswith("_dyn")]
