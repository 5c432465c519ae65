#!/usr/bin/env python3
"""DEV TOOL: reduction grid size (tuning key "reduce_grid"), one process, same buffer."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "rs")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
A = dev.create_empty_buffer(4 * n)
VA = dev.create_empty_buffer((n + 63) // 64 * 8)
R = dev.create_empty_buffer(64)
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-1), C.c_float(1))
capi.call("agpu_synth_bits", h, vp(VA), n, 3, 0, C.c_double(0.9))
p.sync()
cases = {"f32 min": (4, lambda: capi.call("agpu_reduce", h, capi.RED_MIN, capi.F32, vp(A), None, n, vp(R))),
         "f32 sum tree": (4, lambda: capi.call("agpu_reduce", h, capi.RED_SUM, capi.F32, vp(A), None, n, vp(R))),
         "f32 sum null-aware": (4.125, lambda: capi.call("agpu_reduce", h, capi.RED_SUM, capi.F32, vp(A), vp(VA), n, vp(R))),
         "f32 sum f64": (4, lambda: capi.call("agpu_reduce_sum_f64", h, vp(A), None, n, vp(R))),
         "i32 sum": (4, lambda: capi.call("agpu_reduce", h, capi.RED_SUM, capi.I32, vp(A), None, n, vp(R))),
         "popcount": (0.125, lambda: capi.call("agpu_bitmap_popcount", h, vp(VA), n, vp(R)))}
for rep in range(2):
    for g in (0, 256 * 8, 256 * 16, 256 * 32, 256 * 128, 256 * 256, 256 * 1024):
        capi.call("agpu_pipeline_set_tuning", h, b"reduce_grid", g)
        for name, (bpr, f) in cases.items():
            f(); p.sync()
            ts = []
            for _ in range(7):
                q.begin(p); f(); q.end(p)
                ts.append(q.wait_for_results())
            ms = float(np.median(ts))
            print({"kernel": name, "grid": g, "ms": round(ms, 4), "TBps": round(bpr * n / ms / 1e9, 3)}, flush=True)
