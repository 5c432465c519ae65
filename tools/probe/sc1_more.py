#!/usr/bin/env python
"""DEV TOOL (round 5): sc1 nt stores in the 16-bit table kernel (sin_u16 / cos_i16) and in the ×4 cast chain (cast(u8)*s+y), A/B by AGPU_LIB."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "s1"); q = CmpQuery(dev); h = p._handle
p.set_tuning("tile_auto", 1)
u16, y, g = dev.create_table_buffers([2 * n, 4 * n, 4 * n])
capi.call("agpu_synth_u8", h, C.c_void_p(u16.ptr), 2 * n, 7, 0)
capi.call("agpu_synth_f32", h, C.c_void_p(y.ptr), n, 2, 0, C.c_float(-3.0), C.c_float(3.0)); p.sync()
S = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]
def chain(*items):
    arr = (Step * len(items))()
    for k, (op, kind, operand) in enumerate(items):
        arr[k].op, arr[k].kind, arr[k].operand = op, kind, (operand.ptr if operand is not None else None)
    return arr, len(items)
vp = lambda b: C.c_void_p(b.ptr)
c1, n1 = chain((capi.OP_MUL, 1, S), (capi.OP_ADD, 1, S))
c2, n2 = chain((capi.OP_MUL, 1, S), (capi.OP_ADD, 2, y))
def med(fn, bpr):
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return bpr * n / sorted(ts)[4] / 1e6 / 8000
K = {
    "sin_u16": (6.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U16, vp(u16), vp(g), n)),
    "cos_i16": (6.0, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.I16, vp(u16), vp(g), n)),
    "cast(u8)*s+s": (5.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u16), C.cast(c1, C.c_void_p), n1, vp(g), n)),
    "cast(u8)*s+y": (9.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u16), C.cast(c2, C.c_void_p), n2, vp(g), n)),
    "cast(u16)*s+s": (6.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U16, vp(u16), C.cast(c1, C.c_void_p), n1, vp(g), n)),
}
print(" ".join(f"{k} {med(fn, b):.3f}/{med(fn, b):.3f}" for k, (b, fn) in K.items()), flush=True)
