#!/usr/bin/env python
"""DEV TOOL (round 5): u8 / i8 eq → bitmap (word-parallel equality) against the byte-select compares, A/B by AGPU_LIB; u8 lt as the control."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = int(os.environ.get("N", 1_000_000_000))
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "c8"); q = CmpQuery(dev); h = p._handle
na = int(os.environ.get("NALLOC", n))   # allocate for NALLOC rows, run on the first n
A, B, OB = dev.create_table_buffers([na, na, na // 8 + 64])
capi.call("agpu_synth_u8", h, C.c_void_p(A.ptr), na, 6, 0); capi.call("agpu_synth_u8", h, C.c_void_p(B.ptr), na, 7, 0); p.sync()
vp = lambda b: C.c_void_p(b.ptr)
def med(fn):
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return 2.125 * n / sorted(ts)[4] / 1e6 / 8000
K = {"u8 eq": (capi.CMP_EQ, capi.U8), "i8 eq": (capi.CMP_EQ, capi.I8), "u8 lt": (capi.CMP_LT, capi.U8)}
print(" ".join(f"{k} {med(lambda: capi.call('agpu_compare', h, op, dt, vp(A), vp(B), vp(OB), n)):.3f}/{med(lambda: capi.call('agpu_compare', h, op, dt, vp(A), vp(B), vp(OB), n)):.3f}" for k, (op, dt) in K.items()), flush=True)
