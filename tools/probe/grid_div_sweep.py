#!/usr/bin/env python3
"""DEV TOOL: tiles per block (grid = tiles / k through the "stream_grid" knob) for the narrow-type kernels, swept
inside one process on the same buffers.   python tools/probe/grid_div_sweep.py"""
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
p = ArrowComputePipeline(dev, "gd")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
A, B, O = (dev.create_empty_buffer(4 * n) for _ in range(3))
OB = dev.create_empty_buffer((n + 63) // 64 * 8)
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(0.0), C.c_float(255))
capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-3), C.c_float(3))
p.sync()
U8, F32 = capi.U8, capi.F32
# (bytes/row, rows per tile of the tiled kernel, call)
cases = {
    "cast u8->f32": (5, 1024, lambda: capi.call("agpu_cast", h, U8, F32, vp(B), vp(O), n)),
    "cast i16->f32": (6, 512, lambda: capi.call("agpu_cast", h, capi.I16, F32, vp(B), vp(O), n)),
    "cast f32->u8": (5, 4096, lambda: capi.call("agpu_cast", h, F32, U8, vp(A), vp(O), n)),
    "u8 add": (3, 1024, lambda: capi.call("agpu_binary", h, capi.OP_ADD, U8, vp(A), vp(B), vp(O), n)),
    "u8 eq": (2.125, 4096, lambda: capi.call("agpu_compare", h, capi.CMP_EQ, U8, vp(A), vp(B), vp(OB), n)),
    "f32 neg": (8, 256, lambda: capi.call("agpu_unary", h, capi.UN_NEG, F32, vp(A), vp(O), n)),
    "bool->f32": (4.125, 1024, lambda: capi.call("agpu_cast", h, capi.BOOL, F32, vp(OB), vp(O), n)),
}
rows = []
for rep in range(2):
    for k in (1, 2, 4, 8):
        for name, (bpr, tile_rows, f) in cases.items():
            capi.call("agpu_pipeline_set_tuning", h, b"stream_grid", 0 if k == 1 else max(1, n // tile_rows // k))
            f(); p.sync()
            ts = []
            for _ in range(7):
                q.begin(p); f(); q.end(p)
                ts.append(q.wait_for_results())
            ms = float(np.median(ts))
            rows.append({"kernel": name, "tiles_per_block": k, "rep": rep, "ms": round(ms, 4), "TBps": round(bpr * n / ms / 1e9, 3)})
            print(rows[-1], flush=True)
capi.call("agpu_pipeline_set_tuning", h, b"stream_grid", 0)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/grid_div_sweep.json", "w"), indent=1)
