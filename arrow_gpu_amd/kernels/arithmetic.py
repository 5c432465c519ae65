"""arrow_gpu_arithmetic: + − × ÷ % (array∘array, array∘scalar), neg, Sum.

Mirror of crates/arithmetic/src/{arithmetic_kernels,aggregate_kernels,f32,i32,u32,u16}.rs.  Trait methods are attached
to the array classes (`impl ArrowAdd for Float32ArrayGPU` → `Float32ArrayGPU.add/add_op`); the `*_dyn` free
functions keep the reference's dispatch tables exactly (arithmetic_kernels.rs:122-175,225-267), including the
array-vs-scalar choice by `len == 1` (:101-119).
"""
from __future__ import annotations

from .. import _capi as capi
from ..array import (Date32ArrayGPU, Float32ArrayGPU, Int32ArrayGPU, UInt16ArrayGPU, UInt32ArrayGPU)
from ..gpu_utils import ArrowComputePipeline
from ._ops import array_op, dyn_binary, dyn_unary, impl, scalar_op, unary_op, vp

_F32 = (Float32ArrayGPU,)
_I32ish = (Int32ArrayGPU, Date32ArrayGPU, UInt32ArrayGPU)

# ---- ArrowScalar{Add,Sub,Mul,Div,Rem}  [f32.rs:18-61, i32.rs:13-101, u32.rs:9-52, u16.rs:10-17]
impl(_F32 + _I32ish + (UInt16ArrayGPU,), "add_scalar", scalar_op(capi.OP_ADD))
impl(_F32 + _I32ish, "sub_scalar", scalar_op(capi.OP_SUB))
impl(_F32 + _I32ish, "mul_scalar", scalar_op(capi.OP_MUL))
impl(_F32 + _I32ish, "div_scalar", scalar_op(capi.OP_DIV))
impl(_F32 + _I32ish, "rem_scalar", scalar_op(capi.OP_REM))

# ---- ArrowAdd/Sub/Mul/Div (array ∘ array)  [f32.rs:63-97, i32.rs:103-119, u32.rs:54-61]
# add: f32, u32, i32, Date32 (+ i32↔Date32 → Date32); sub/mul/div: f32 only in the reference.  The int variants of
# sub/mul exist in the C ABI and are exposed on the typed classes as a superset; the *_dyn tables stay the reference's.
impl(_F32 + _I32ish, "add", array_op(capi.OP_ADD))
impl(_F32 + _I32ish, "sub", array_op(capi.OP_SUB))
impl(_F32 + _I32ish, "mul", array_op(capi.OP_MUL))
impl(_F32, "div", array_op(capi.OP_DIV))

# ---- Neg  [arithmetic_kernels.rs:270-343: f32 only]
impl(_F32, "neg", unary_op(capi.UN_NEG))


# ---- Sum  [aggregate_kernels.rs:7-52: f32, i32, u32 → 1-element array, null_buffer None, validity ignored]
def _sum_op(self, pipeline: ArrowComputePipeline):
    dev = self.gpu_device
    out = dev.create_empty_buffer(16)
    capi.call("agpu_reduce", pipeline._handle, capi.RED_SUM, self.DTYPE, vp(self.data), None, self.len, vp(out))
    pipeline.keep(self.data, out)
    return type(self)(out, dev, 1, None)


impl((Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU), "sum", _sum_op)


# ---- whole-column statistics in ONE pass (no counterpart in the reference: its only reduction is Sum; north_star config 5 names sum / min /
# max of one f32 column).  `col.stats_op(p)` / `col.stats()` → Float32Stats: the 24-byte record stays on the device until values() asks.
class Float32Stats:
    """agpu_reduce_stats_f32's record {sum (the reference's tree order), min, max (Arrow's NaN rule), sum_f64}: every field bit-identical
    to the separate reduction of the same column; the column is read once."""

    def __init__(self, buf, gpu_device):
        self.data, self.gpu_device = buf, gpu_device

    def values(self) -> dict:
        import numpy as np

        raw = self.gpu_device.retrive_data(self.data, 24)
        f = raw[:12].view(np.float32)
        return {"sum": f[0], "min": f[1], "max": f[2], "sum_f64": raw[16:24].view(np.float64)[0]}


def _stats_op(self, pipeline: ArrowComputePipeline):
    dev = self.gpu_device
    out = dev.create_empty_buffer(32)
    validity = vp(self.null_buffer.bit_buffer) if getattr(self, "null_buffer", None) is not None else None
    capi.call("agpu_reduce_stats_f32", pipeline._handle, vp(self.data), validity, self.len, vp(out))
    pipeline.keep(self.data, out)
    return Float32Stats(out, dev)


impl(_F32, "stats", _stats_op)

# ---- dyn dispatch  [arithmetic_kernels.rs:122-175]
_date_mix = ((Int32ArrayGPU, Date32ArrayGPU), (Date32ArrayGPU, Int32ArrayGPU))
add_scalar_dyn, add_scalar_op_dyn = dyn_binary(
    "add_scalar_dyn", "add_scalar_op", (Float32ArrayGPU, Int32ArrayGPU, Date32ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU))
sub_scalar_dyn, sub_scalar_op_dyn = dyn_binary("sub_scalar_dyn", "sub_scalar_op", (Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU))
mul_scalar_dyn, mul_scalar_op_dyn = dyn_binary("mul_scalar_dyn", "mul_scalar_op", (Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU))
div_scalar_dyn, div_scalar_op_dyn = dyn_binary("div_scalar_dyn", "div_scalar_op", (Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU))
rem_scalar_dyn, rem_scalar_op_dyn = dyn_binary(
    "rem_scalar_dyn", "rem_scalar_op", (Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU, Date32ArrayGPU), _date_mix)

# [arithmetic_kernels.rs:225-260]
add_array_dyn, add_array_op_dyn = dyn_binary(
    "add_array_dyn", "add_op", (Float32ArrayGPU, UInt32ArrayGPU, Int32ArrayGPU, Date32ArrayGPU), _date_mix)
sub_array_dyn, sub_array_op_dyn = dyn_binary("sub_array_dyn", "sub_op", (Float32ArrayGPU,))
mul_array_dyn, mul_array_op_dyn = dyn_binary("mul_array_dyn", "mul_op", (Float32ArrayGPU,))
div_array_dyn, div_array_op_dyn = dyn_binary("div_array_dyn", "div_op", (Float32ArrayGPU,))


# NB: mixed i32/Date32 operands keep `type Output = Self` (the LEFT operand's type), as in the reference's
# impl_arithmetic_op!/impl_arithmetic_array_op! (crates/arithmetic/src/lib.rs:16,59; tests i32.rs:188-255).


def _len_dispatch(name, array_op_dyn, scalar_op_dyn):
    """add_dyn & co: array∘array when both or neither have len 1, else scalar with the len-1 side as the scalar
    [arithmetic_kernels.rs:101-119]."""

    def fn_op(input1, input2, pipeline: ArrowComputePipeline):
        x, y = input1.len, input2.len
        if (x == 1 and y == 1) or (x != 1 and y != 1):
            return array_op_dyn(input1, input2, pipeline)
        if y == 1:
            return scalar_op_dyn(input1, input2, pipeline)
        return scalar_op_dyn(input2, input1, pipeline)

    def fn(input1, input2):
        pipeline = ArrowComputePipeline(input1.get_gpu_device(), name)
        out = fn_op(input1, input2, pipeline)
        pipeline.finish()
        return out

    fn.__name__ = name
    fn_op.__name__ = name.replace("_dyn", "_op_dyn")
    return fn, fn_op


add_dyn, add_op_dyn = _len_dispatch("add_dyn", add_array_op_dyn, add_scalar_op_dyn)
sub_dyn, sub_op_dyn = _len_dispatch("sub_dyn", sub_array_op_dyn, sub_scalar_op_dyn)
mul_dyn, mul_op_dyn = _len_dispatch("mul_dyn", mul_array_op_dyn, mul_scalar_op_dyn)
div_dyn, div_op_dyn = _len_dispatch("div_dyn", div_array_op_dyn, div_scalar_op_dyn)

neg_dyn, neg_op_dyn = dyn_unary("neg_dyn", "neg_op", (Float32ArrayGPU,))

__all__ = [n for n in dir() if n.endswith("_dyn")]
