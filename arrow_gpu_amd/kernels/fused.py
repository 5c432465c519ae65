"""Fused element-wise chains — one kernel, one pass over HBM for a whole `*_op` chain.

The reference batches the dispatches of a chain into one command buffer (`examples/simple.rs:45-72`:
`add_scalar_op_dyn` → `mul_scalar_op_dyn` → one `finish()`), but each op still reads and writes the full column.
`FusedChain` records the same chain explicitly and runs it through `agpu_fused_chain`:

    out = ag.FusedChain(a).add_scalar(s).mul_scalar(s).finish()          # (a + s) * s, 8 B/row instead of 16
    out = ag.FusedChain(a).mul(b).add(c).abs().sqrt().finish_op(pipeline)

Each step applies the same scalar operation with the same rounding as the stand-alone kernel, so the result is
bit-identical to `a.add_scalar(s).mul_scalar(s)`; validity follows the reference's rules step by step (clone for
unary/scalar steps, AND with every array operand's validity).  f32 / i32 / u32 / Date32 columns, ≤ 8 steps.

A u8 / i8 / u16 / i16 column may START a chain: the widening cast to f32 is its head (`agpu_fused_cast_chain`) and the steps are
f32 steps — `ag.FusedChain(u8_col).sin().finish()` is the reference's fused `sin_u8` (trigonometry/src/u8_kernel.rs:34-38),
`ag.FusedChain(u8_col).mul_scalar(s).add_scalar(o).finish()` reads 1 B/row and writes 4 instead of 5 + 8 + 8.
"""
from __future__ import annotations

import ctypes as C

from .. import _capi as capi
from .._capi import ArrowErrorGPU, OperationNotSupported
from ..array import (Date32ArrayGPU, Float32ArrayGPU, Int8ArrayGPU, Int16ArrayGPU, Int32ArrayGPU, NullBitBufferGpu, UInt8ArrayGPU,
                     UInt16ArrayGPU, UInt32ArrayGPU)
from ..gpu_utils import ArrowComputePipeline
from ._ops import vp

MAX_STEPS = 8
UNARY, SCALAR, ARRAY = 0, 1, 2


class _Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]


_BINARY = {"add": capi.OP_ADD, "sub": capi.OP_SUB, "mul": capi.OP_MUL, "div": capi.OP_DIV, "rem": capi.OP_REM,
           "min": capi.OP_MIN, "max": capi.OP_MAX, "bitwise_and": capi.OP_AND, "bitwise_or": capi.OP_OR,
           "bitwise_xor": capi.OP_XOR}
_UNARY = {"neg": capi.UN_NEG, "abs": capi.UN_ABS, "bitwise_not": capi.UN_NOT, "sqrt": capi.UN_SQRT, "cbrt": capi.UN_CBRT,
          "exp": capi.UN_EXP, "exp2": capi.UN_EXP2, "log": capi.UN_LOG, "log2": capi.UN_LOG2, "sin": capi.UN_SIN,
          "cos": capi.UN_COS, "acos": capi.UN_ACOS, "sinh": capi.UN_SINH}
_TYPES = (Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU, Date32ArrayGPU)
_CAST_HEADS = (UInt8ArrayGPU, Int8ArrayGPU, UInt16ArrayGPU, Int16ArrayGPU)  # the chain computes in f32 behind a widening cast


class FusedChain:
    def __init__(self, array):
        if type(array) not in _TYPES + _CAST_HEADS:
            raise OperationNotSupported(f"FusedChain not supported for type {array.get_dtype().name}")
        self.src = array
        self.cast_head = type(array) in _CAST_HEADS
        self.acc_cls = Float32ArrayGPU if self.cast_head else type(array)  # the type the steps compute in / the result type
        self.steps = []  # (op, kind, operand array or None)

    def _push(self, op, kind, operand):
        if len(self.steps) >= MAX_STEPS:
            raise ArrowErrorGPU("ShapeError", f"a fused chain holds at most {MAX_STEPS} steps", capi.ERR_SHAPE)
        self.steps.append((op, kind, operand))
        return self

    def _binary(self, name, other, force_scalar=False):
        if type(other).NP_DTYPE != self.acc_cls.NP_DTYPE or (self.cast_head and name.startswith("bitwise")):
            raise OperationNotSupported(
                f"Operation {name} not supported for type {self.acc_cls.ARROW_TYPE.name} {other.get_dtype().name}")
        scalar = force_scalar or (other.len == 1 and self.src.len != 1)
        if not scalar and other.len != self.src.len:
            raise ArrowErrorGPU("ShapeError", f"{name}: arrays of different length", capi.ERR_SHAPE)
        return self._push(_BINARY[name], SCALAR if scalar else ARRAY, other)

    def finish_op(self, pipeline: ArrowComputePipeline):
        a = self.src
        dev = a.gpu_device
        out = dev.create_empty_buffer(max(a.len * self.acc_cls.ITEM_SIZE, 1))
        steps = (_Step * max(len(self.steps), 1))()
        nulls = a.null_buffer  # validity: AND with every ARRAY operand's bitmap; unary / scalar steps clone
        for i, (op, kind, operand) in enumerate(self.steps):
            steps[i].op, steps[i].kind = op, kind
            steps[i].operand = operand.data.ptr if operand is not None else None
            if kind == ARRAY and operand.null_buffer is not None:
                nulls = NullBitBufferGpu.merge_null_bit_buffer_op(nulls, operand.null_buffer, pipeline)
            if operand is not None:
                pipeline.keep(operand.data)
        if nulls is a.null_buffer:  # never merged: the output owns a copy, like every unary/scalar op
            nulls = NullBitBufferGpu.clone_null_bit_buffer_op(nulls, pipeline)
        if self.cast_head:
            capi.call("agpu_fused_cast_chain", pipeline._handle, a.DTYPE, vp(a.data), C.cast(steps, C.c_void_p), len(self.steps),
                      vp(out), a.len)
        else:
            capi.call("agpu_fused_chain", pipeline._handle, a.DTYPE, vp(a.data), C.cast(steps, C.c_void_p), len(self.steps),
                      vp(out), a.len)
        pipeline.keep(a.data, out)
        return self.acc_cls(out, dev, a.len, nulls)

    def finish(self):
        p = ArrowComputePipeline(self.src.get_gpu_device(), "fused_chain")
        out = self.finish_op(p)
        p.finish()
        return out

    # ---- terminal compare: the predicate (chain(a) cmp other) → BooleanArrayGPU without storing the chain's result
    def _compare_op(self, name: str, op: int, other, pipeline: ArrowComputePipeline):
        from ..array import BooleanArrayGPU, bitmap_bytes

        a = self.src
        if self.cast_head or type(other).NP_DTYPE != type(a).NP_DTYPE:  # (a cast-headed chain stores its result; compare it afterwards)
            raise OperationNotSupported(f"Operation {name} not supported for type {a.get_dtype().name} {other.get_dtype().name}")
        if len(self.steps) >= MAX_STEPS:
            raise ArrowErrorGPU("ShapeError", f"a fused chain holds at most {MAX_STEPS - 1} steps before a compare", capi.ERR_SHAPE)
        scalar = other.len == 1 and a.len != 1
        if not scalar and other.len != a.len:
            raise ArrowErrorGPU("ShapeError", f"{name}: arrays of different length", capi.ERR_SHAPE)
        dev = a.gpu_device
        out = dev.create_empty_buffer(max(bitmap_bytes(a.len), 8))
        steps = (_Step * max(len(self.steps), 1))()
        nulls, merged = a.null_buffer, False
        for i, (sop, kind, operand) in enumerate(self.steps):
            steps[i].op, steps[i].kind = sop, kind
            steps[i].operand = operand.data.ptr if operand is not None else None
            if kind == ARRAY and operand.null_buffer is not None:
                nulls, merged = NullBitBufferGpu.merge_null_bit_buffer_op(nulls, operand.null_buffer, pipeline), True
            if operand is not None:
                pipeline.keep(operand.data)
        if not scalar and other.null_buffer is not None:  # compare: validity AND with the array operand's
            nulls, merged = NullBitBufferGpu.merge_null_bit_buffer_op(nulls, other.null_buffer, pipeline), True
        if not merged:
            nulls = NullBitBufferGpu.clone_null_bit_buffer_op(nulls, pipeline)
        capi.call("agpu_fused_chain_compare", pipeline._handle, a.DTYPE, vp(a.data), C.cast(steps, C.c_void_p), len(self.steps),
                  op, SCALAR if scalar else ARRAY, vp(other.data), vp(out), a.len)
        pipeline.keep(a.data, other.data, out)
        return BooleanArrayGPU(out, dev, a.len, nulls)

    def _compare(self, name: str, op: int, other):
        p = ArrowComputePipeline(self.src.get_gpu_device(), "fused_chain_compare")
        out = self._compare_op(name, op, other, p)
        p.finish()
        return out


def _make_binary(name):
    def method(self, other):
        return self._binary(name, other)

    method.__name__ = name
    return method


def _make_scalar(name):
    def method(self, other):
        return self._binary(name, other, force_scalar=True)

    method.__name__ = name + "_scalar"
    return method


def _make_unary(name):
    def method(self):
        if name == "bitwise_not" and self.acc_cls is Float32ArrayGPU:
            raise OperationNotSupported("Operation bitwise_not not supported for type Float32Type")
        if name not in ("neg", "abs", "bitwise_not") and self.acc_cls is not Float32ArrayGPU:
            raise OperationNotSupported(f"Operation {name} not supported for type {self.src.get_dtype().name}")
        return self._push(_UNARY[name], UNARY, None)

    method.__name__ = name
    return method


for _n in _BINARY:
    setattr(FusedChain, _n, _make_binary(_n))
for _n in ("add", "sub", "mul", "div", "rem"):
    setattr(FusedChain, _n + "_scalar", _make_scalar(_n))
for _n in _UNARY:
    setattr(FusedChain, _n, _make_unary(_n))


def _make_compare(name, op):
    def method(self, other):
        return self._compare(name, op, other)

    def method_op(self, other, pipeline):
        return self._compare_op(name, op, other, pipeline)

    method.__name__, method_op.__name__ = name, name + "_op"
    return method, method_op


for _n, _op in (("gt", capi.CMP_GT), ("gteq", capi.CMP_GTEQ), ("lt", capi.CMP_LT), ("lteq", capi.CMP_LTEQ), ("eq", capi.CMP_EQ)):
    _m, _mop = _make_compare(_n, _op)
    setattr(FusedChain, _n, _m)
    setattr(FusedChain, _n + "_op", _mop)

__all__ = ["FusedChain"]
