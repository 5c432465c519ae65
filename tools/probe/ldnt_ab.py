#!/usr/bin/env python
"""DEV TOOL (round 5): nontemporal against plain LOADS in the f32 element-wise kernels under the occupancy cap (build-time AGPU_STREAM_NT = 3 / 2,
A/B by AGPU_LIB): sin / cos (capped), exp, neg, add (32 waves)."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "ld"); q = CmpQuery(dev); h = p._handle
p.set_tuning("tile_auto", 1)
a, b, g = dev.create_table_buffers([4 * n, 4 * n, 4 * n])
vp = lambda x: C.c_void_p(x.ptr)
capi.call("agpu_synth_f32", h, vp(a), n, 1, 0, C.c_float(-3.0), C.c_float(3.0)); capi.call("agpu_synth_f32", h, vp(b), n, 2, 0, C.c_float(-3.0), C.c_float(3.0)); p.sync()
def med(fn, bpr):
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return bpr * n / sorted(ts)[4] / 1e6 / 8000
K = {"sin": (8, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(a), vp(g), n)),
     "cos": (8, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.F32, vp(a), vp(g), n)),
     "exp": (8, lambda: capi.call("agpu_unary", h, capi.UN_EXP, capi.F32, vp(a), vp(g), n)),
     "neg": (8, lambda: capi.call("agpu_unary", h, capi.UN_NEG, capi.F32, vp(a), vp(g), n)),
     "add": (12, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(a), vp(b), vp(g), n))}
print(" ".join(f"{k} {med(fn, bpr):.3f}/{med(fn, bpr):.3f}" for k, (bpr, fn) in K.items()), flush=True)
