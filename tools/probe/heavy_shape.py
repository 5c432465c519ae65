#!/usr/bin/env python3
"""DEV TOOL: the VALU-heavy f32 unaries (sin / cos / sinh / log / exp) at 1e9 rows; block shape and packs per lane are
build-time (AGPU_HEAVY_BLK / AGPU_HEAVY_U), AGPU_LIB selects the build.   AGPU_LIB=… python tools/probe/heavy_shape.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "hs")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
A, P, O = dev.create_table_buffers([4 * n] * 3)
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-1000), C.c_float(1000))
capi.call("agpu_synth_f32", h, vp(P), n, 2, 0, C.c_float(0.001), C.c_float(1000))
p.sync()
out = []
for name, op, src in (("sin", capi.UN_SIN, A), ("cos", capi.UN_COS, A), ("sinh", capi.UN_SINH, A), ("log+", capi.UN_LOG, P),
                      ("log±", capi.UN_LOG, A), ("exp", capi.UN_EXP, A), ("neg", capi.UN_NEG, A)):
    f = lambda: capi.call("agpu_unary", h, op, capi.F32, vp(src), vp(O), n)  # noqa: E731
    for _ in range(6):
        f()
    p.sync()
    ts = []
    for _ in range(11):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    ms = float(np.median(ts))
    out.append(f"{name} {ms:.4f} {8 * n / ms / 8e9:.3f}")
print("   ".join(out), flush=True)
