"""arrow_gpu_trigonometry: sin, cos, acos, sinh on f32 and the fused cast+trig kernels on u8/i8/u16/i16 → f32.

Mirror of crates/trigonometry/src/lib.rs (Trigonometric/Hyperbolic :22-83, launch :85-137 with entry point
"sin_{TYPE_STR}", dyn tables :161-202) and src/{f32,u8,i8,u16,i16}_kernel.rs (OutputType = Float32ArrayGPU).
"""
from __future__ import annotations

from .. import _capi as capi
from ..array import Float32ArrayGPU, Int8ArrayGPU, Int16ArrayGPU, UInt8ArrayGPU, UInt16ArrayGPU
from ._ops import dyn_unary, impl, unary_op

_TRIG = (Float32ArrayGPU, UInt16ArrayGPU, UInt8ArrayGPU, Int16ArrayGPU, Int8ArrayGPU)
impl(_TRIG, "sin", unary_op(capi.UN_SIN, Float32ArrayGPU))
impl(_TRIG, "cos", unary_op(capi.UN_COS, Float32ArrayGPU))
impl(_TRIG, "sinh", unary_op(capi.UN_SINH, Float32ArrayGPU))
impl((Float32ArrayGPU,), "acos", unary_op(capi.UN_ACOS, Float32ArrayGPU))

sinh_dyn, sinh_op_dyn = dyn_unary("sinh_dyn", "sinh_op", _TRIG)
cos_dyn, cos_op_dyn = dyn_unary("cos_dyn", "cos_op", _TRIG)
sin_dyn, sin_op_dyn = dyn_unary("sin_dyn", "sin_op", _TRIG)
acos_dyn, acos_op_dyn = dyn_unary("acos_dyn", "acos_op", (Float32ArrayGPU,))

__all__ = [n for n in dir() if n.endswith("_dyn")]
