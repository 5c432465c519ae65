#!/usr/bin/env python3
"""DEV TOOL: tiles per block for the LDS-table kernels (lut8 / trig16 / pow), swept inside ONE process on the same
buffers (separate processes differ by 5-10 % through buffer placement alone).   python tools/probe/table_tiles.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "tt")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
A, B, O = (dev.create_empty_buffer(4 * n) for _ in range(3))
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(0.001), C.c_float(1000))
capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-3), C.c_float(3))
p.sync()
cases = {"sin_u8": (5, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(B), vp(O), n)),
         "sin_u16": (6, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U16, vp(B), vp(O), n)),
         "pow_f32": (12, lambda: capi.call("agpu_binary", h, capi.OP_POW, capi.F32, vp(A), vp(B), vp(O), n))}
rows = []
for rep in range(3):
    for k in (1, 2, 3, 4, 6, 8):
        capi.call("agpu_pipeline_set_tuning", h, b"table_tiles", k)
        for name, (bpr, f) in cases.items():
            f(); p.sync()
            ts = []
            for _ in range(7):
                q.begin(p); f(); q.end(p)
                ts.append(q.wait_for_results())
            ms = float(np.median(ts))
            rows.append({"kernel": name, "table_tiles": k, "rep": rep, "ms": round(ms, 4), "TBps": round(bpr * n / ms / 1e9, 3)})
            print(rows[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/table_tiles.json", "w"), indent=1)
