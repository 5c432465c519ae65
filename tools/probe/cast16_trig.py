#!/usr/bin/env python
"""DEV TOOL (round 5): cast(i16 / u16) -> sin / cos in one launch (cvt_wide_kernel over CvtThenF32): the i16 form carried 8 out-of-line slow-path calls per 8 rows."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "c16"); q = CmpQuery(dev); h = p._handle
u16, g = dev.create_table_buffers([2 * n, 4 * n])
capi.call("agpu_synth_u8", h, C.c_void_p(u16.ptr), 2 * n, 7, 0); p.sync()
class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]
def chain(*items):
    arr = (Step * len(items))()
    for k, (op, kind, operand) in enumerate(items):
        arr[k].op, arr[k].kind, arr[k].operand = op, kind, None
    return arr, len(items)
vp = lambda b: C.c_void_p(b.ptr)
def med(fn, bpr=6.0):
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return bpr * n / sorted(ts)[4] / 1e6 / 8000
row = []
for dt, nm in ((capi.I16, "i16"), (capi.U16, "u16")):
    for op, on in ((capi.UN_SIN, "sin"), (capi.UN_COS, "cos")):
        c, nc = chain((op, 0, None))
        row.append(f"cast({nm})->{on} {med(lambda: capi.call('agpu_fused_cast_chain', h, dt, vp(u16), C.cast(c, C.c_void_p), nc, vp(g), n)):.3f}")
print(" ".join(row), flush=True)
