#!/usr/bin/env python3
"""DEV TOOL (GPU box): the LIGHT fused chains (no transcendental step) — (a + s)·t, a·b + c, (a·b + c) > d — for the A/B of a chain-kernel change (AGPU_LIB)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "cl"); q = CmpQuery(dev); h = p._handle
a, b, c, d, o = dev.create_table_buffers([4 * n] * 5)
ob = dev.create_empty_buffer((n + 63) // 64 * 8)
for k, x in enumerate((a, b, c, d)):
    capi.call("agpu_synth_f32", h, C.c_void_p(x.ptr), n, k + 1, 0, C.c_float(-3.0), C.c_float(3.0))
p.sync()
S = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]
def chain(*items):
    arr = (Step * len(items))()
    for k, (op, kind, operand) in enumerate(items):
        arr[k].op, arr[k].kind, arr[k].operand = op, kind, (operand.ptr if operand is not None else None)
    return arr, len(items)
vp = lambda x: C.c_void_p(x.ptr)
c1, n1 = chain((capi.OP_ADD, 1, S), (capi.OP_MUL, 1, S))
c2, n2 = chain((capi.OP_MUL, 2, b), (capi.OP_ADD, 2, c))
K = {"(a+s)*t": (8.0, lambda: capi.call("agpu_fused_chain", h, capi.F32, vp(a), C.cast(c1, C.c_void_p), n1, vp(o), n)),
     "a*b+c": (16.0, lambda: capi.call("agpu_fused_chain", h, capi.F32, vp(a), C.cast(c2, C.c_void_p), n2, vp(o), n)),
     "(a*b+c)>d": (16.125, lambda: capi.call("agpu_fused_chain_compare", h, capi.F32, vp(a), C.cast(c2, C.c_void_p), n2, capi.CMP_GT, 2, vp(d), vp(ob), n)),
     "i32 (a+s)*t": (8.0, lambda: capi.call("agpu_fused_chain", h, capi.I32, vp(a), C.cast(c1, C.c_void_p), n1, vp(o), n))}
def med(fn, bpr):
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(11):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return bpr * n / sorted(ts)[5] / 1e6 / 8000
print("  ".join(f"{name} {med(fn, bpr):.3f} {med(fn, bpr):.3f}" for name, (bpr, fn) in K.items()), flush=True)
