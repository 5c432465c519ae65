#!/usr/bin/env python3
"""DEV TOOL: the widening casts (u8 / i16 → f32, u8 → u32) against the threads per block of cvt_wide_kernel (build-time
AGPU_CVTW_BLOCK, AGPU_LIB selects the build), 1e9 rows."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "cs")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
A, O = dev.create_table_buffers([4 * n] * 2)
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-1000), C.c_float(1000))
p.sync()
out = []
for name, frm, to, bpr in (("u8→f32", capi.U8, capi.F32, 5), ("i16→f32", capi.I16, capi.F32, 6), ("u8→u32", capi.U8, capi.U32, 5),
                           ("u8→u16", capi.U8, capi.U16, 3), ("f32→u8", capi.F32, capi.U8, 5)):
    f = lambda: capi.call("agpu_cast", h, frm, to, vp(A), vp(O), n)  # noqa: E731
    for _ in range(5):
        f()
    p.sync()
    ts = []
    for _ in range(11):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    ms = float(np.median(ts))
    out.append(f"{name} {ms:.4f} {bpr * n / ms / 8e9:.3f}")
print("   ".join(out), flush=True)
