#!/usr/bin/env python
"""DEV TOOL (round 5): sc1 nt against plain nt for the ONE 16-byte store per four loads of the narrowing casts (f32 -> u8 / i8 / i16 / u16), A/B by AGPU_LIB."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "nw"); q = CmpQuery(dev); h = p._handle
a, g = dev.create_table_buffers([4 * n, 2 * n])
vp = lambda x: C.c_void_p(x.ptr)
capi.call("agpu_synth_f32", h, vp(a), n, 1, 0, C.c_float(0.0), C.c_float(200.0)); p.sync()
def med(fn, bpr):
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return bpr * n / sorted(ts)[4] / 1e6 / 8000
K = {"f32->u8": (5, capi.U8), "f32->i8": (5, capi.I8), "f32->u16": (6, capi.U16), "f32->i16": (6, capi.I16)}
print(" ".join(f"{k} {med(lambda: capi.call('agpu_cast', h, capi.F32, dt, vp(a), vp(g), n), b):.3f}/{med(lambda: capi.call('agpu_cast', h, capi.F32, dt, vp(a), vp(g), n), b):.3f}" for k, (b, dt) in K.items()), flush=True)
