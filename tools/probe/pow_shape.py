#!/usr/bin/env python3
"""DEV TOOL: f32 pow / log at 1e9 rows against the tiles-per-block tuning key, inside one process (the block shape and
the packs per lane are build-time: AGPU_LIB selects the build).   AGPU_LIB=… python tools/probe/pow_shape.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "ps")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
A, B, O = dev.create_table_buffers([4 * n] * 3)
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(0.001), C.c_float(1000))
capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-8), C.c_float(8))
p.sync()
cases = {"pow": (12, lambda: capi.call("agpu_binary", h, capi.OP_POW, capi.F32, vp(A), vp(B), vp(O), n)),
         "log": (8, lambda: capi.call("agpu_unary", h, capi.UN_LOG, capi.F32, vp(A), vp(O), n)),
         "sin_u8": (5, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(B), vp(O), n)),
         "sin_u16": (6, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U16, vp(B), vp(O), n)),
         "add": (12, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(A), vp(B), vp(O), n))}
ks = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]
for k in ks:
    capi.call("agpu_pipeline_set_tuning", h, b"table_tiles", k)
    out = []
    for name, (bpr, f) in cases.items():
        for _ in range(6):
            f()
        p.sync()
        ts = []
        for _ in range(11):
            q.begin(p); f(); q.end(p)
            ts.append(q.wait_for_results())
        ms = float(np.median(ts))
        out.append(f"{name} {ms:.4f} ms {bpr * n / ms / 8e9:.3f}")
    print(f"table_tiles={k:3d}  " + "   ".join(out), flush=True)
