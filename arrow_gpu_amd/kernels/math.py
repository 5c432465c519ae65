"""arrow_gpu_math: abs, sqrt, cbrt, exp, exp2, log, log2, power.

Mirror of crates/math/src/lib.rs (MathUnary/FloatMathUnary/MathBinary :37-136, dyn tables :259-348).
"""
from __future__ import annotations

from .. import _capi as capi
from ..array import Float32ArrayGPU, Int32ArrayGPU
from ._ops import array_op, dyn_binary, dyn_unary, impl, unary_op

_F32 = (Float32ArrayGPU,)
impl((Float32ArrayGPU, Int32ArrayGPU), "abs", unary_op(capi.UN_ABS))
impl(_F32, "sqrt", unary_op(capi.UN_SQRT))
impl(_F32, "cbrt", unary_op(capi.UN_CBRT))
impl(_F32, "exp", unary_op(capi.UN_EXP))
impl(_F32, "exp2", unary_op(capi.UN_EXP2))
impl(_F32, "log", unary_op(capi.UN_LOG))
impl(_F32, "log2", unary_op(capi.UN_LOG2))
impl((Float32ArrayGPU, Int32ArrayGPU), "power", array_op(capi.OP_POW))

abs_dyn, abs_op_dyn = dyn_unary("abs_dyn", "abs_op", (Float32ArrayGPU, Int32ArrayGPU))
sqrt_dyn, sqrt_op_dyn = dyn_unary("sqrt_dyn", "sqrt_op", _F32)
cbrt_dyn, cbrt_op_dyn = dyn_unary("cbrt_dyn", "cbrt_op", _F32)
exp_dyn, exp_op_dyn = dyn_unary("exp_dyn", "exp_op", _F32)
exp2_dyn, exp2_op_dyn = dyn_unary("exp2_dyn", "exp2_op", _F32)
log_dyn, log_op_dyn = dyn_unary("log_dyn", "log_op", _F32)
log2_dyn, log2_op_dyn = dyn_unary("log2_dyn", "log2_op", _F32)
power_dyn, power_op_dyn = dyn_binary("power_dyn", "power_op", (Int32ArrayGPU, Float32ArrayGPU))

__all__ = [n for n in dir() if n.endswith("_dyn")]
