#!/usr/bin/env python3
"""DEV TOOL (GPU box, under rocprofv3 --pmc): sin f32 standalone, the 1-step chain [sin] and the 2-step chain [mul_scalar, sin], three launches each."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
n = 1 << 28
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "pmc"); h = p._handle
p.set_tuning("tile_auto", 1)
f, g = dev.create_table_buffers([4 * n, 4 * n])
capi.call("agpu_synth_f32", h, C.c_void_p(f.ptr), n, 1, 0, C.c_float(-30.0), C.c_float(30.0)); p.sync()
S = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]
def chain(*items):
    arr = (Step * len(items))()
    for k, (op, kind, operand) in enumerate(items):
        arr[k].op, arr[k].kind, arr[k].operand = op, kind, (operand.ptr if operand is not None else None)
    return arr, len(items)
vp = lambda b: C.c_void_p(b.ptr)
c1, n1 = chain((capi.UN_SIN, 0, None))
c2, n2 = chain((capi.OP_MUL, 1, S), (capi.UN_SIN, 0, None))
c3, n3 = chain((capi.OP_MUL, 1, S), (capi.OP_ADD, 1, S))
for _ in range(3):
    capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(g), n)
    capi.call("agpu_fused_chain", h, capi.F32, vp(f), C.cast(c1, C.c_void_p), n1, vp(g), n)
    capi.call("agpu_fused_chain", h, capi.F32, vp(f), C.cast(c2, C.c_void_p), n2, vp(g), n)
    capi.call("agpu_fused_chain", h, capi.F32, vp(f), C.cast(c3, C.c_void_p), n3, vp(g), n)
    capi.call("agpu_scalar", h, capi.OP_MUL, capi.F32, vp(f), vp(S), vp(g), n)
    p.sync()
