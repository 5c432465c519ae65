#!/usr/bin/env python3
"""DEV TOOL (GPU box): the occupancy cap (tuning wave_lds) on fused chains with a transcendental step — (x·s).sin(), (x·s + y).cos(), cast(u16)·s → sin"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "cc"); q = CmpQuery(dev); h = p._handle
u16, f, y, g = dev.create_table_buffers([2 * n, 4 * n, 4 * n, 4 * n])
capi.call("agpu_synth_u8", h, C.c_void_p(u16.ptr), 2 * n, 7, 0)
capi.call("agpu_synth_f32", h, C.c_void_p(f.ptr), n, 1, 0, C.c_float(-30.0), C.c_float(30.0))
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
c1, n1 = chain((capi.OP_MUL, 1, S), (capi.UN_SIN, 0, None))
c2, n2 = chain((capi.OP_MUL, 1, S), (capi.OP_ADD, 2, y), (capi.UN_COS, 0, None))
c3, n3 = chain((capi.OP_MUL, 1, S), (capi.UN_SIN, 0, None))
c4, n4 = chain((capi.OP_MUL, 1, S), (capi.UN_EXP, 0, None))
c5, n5 = chain((capi.OP_ADD, 2, y), (capi.UN_ABS, 0, None), (capi.UN_LOG2, 0, None), (capi.OP_MUL, 1, S))
K = {"(x*s).sin() f32": (8.0, lambda: capi.call("agpu_fused_chain", h, capi.F32, vp(f), C.cast(c1, C.c_void_p), n1, vp(g), n)),
     "(x*s+y).cos() f32": (12.0, lambda: capi.call("agpu_fused_chain", h, capi.F32, vp(f), C.cast(c2, C.c_void_p), n2, vp(g), n)),
     "(x*s).exp() f32": (8.0, lambda: capi.call("agpu_fused_chain", h, capi.F32, vp(f), C.cast(c4, C.c_void_p), n4, vp(g), n)),
     "log2(|x+y|)*s f32": (12.0, lambda: capi.call("agpu_fused_chain", h, capi.F32, vp(f), C.cast(c5, C.c_void_p), n5, vp(g), n)),
     "cast(u16)*s -> sin": (6.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U16, vp(u16), C.cast(c3, C.c_void_p), n3, vp(g), n))}
def med(fn, bpr):
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return bpr * n / sorted(ts)[4] / 1e6 / 8000
for name, (bpr, fn) in K.items():
    row = []
    for cap in [int(x) for x in os.environ.get("CHAIN_CAPS", "-1,4200,5600,6800,8000,10240,13600,-1,6800").split(",")]:
        p.set_tuning("wave_lds", cap); row.append(f"{cap}:{med(fn, bpr):.3f}")
    print(name.ljust(22), " ".join(row), flush=True)
