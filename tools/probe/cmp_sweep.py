#!/usr/bin/env python3
"""DEV TOOL: compare kernels — ballot vs vector variant with 1/2/4 packs per lane in flight, one process, same buffers."""
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
A, B = dev.create_empty_buffer(4 * n), dev.create_empty_buffer(4 * n)
nb = (n + 63) // 64 * 8
VA, VB, OB, OV = (dev.create_empty_buffer(nb) for _ in range(4))
capi.call("agpu_synth_i32", h, vp(A), n, 1, 0, 1024)
capi.call("agpu_synth_i32", h, vp(B), n, 2, 0, 1024)
capi.call("agpu_synth_bits", h, vp(VA), n, 3, 0, C.c_double(0.9))
capi.call("agpu_synth_bits", h, vp(VB), n, 4, 0, C.c_double(0.9))
p.sync()
cases = {"i32 eq": (8.125, lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(A), vp(B), vp(OB), n)),
         "i32 eq + validity": (8.5, lambda: capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, vp(A), vp(B), vp(VA), vp(VB), vp(OB), vp(OV), n)),
         "u8 eq": (2.125, lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.U8, vp(A), vp(B), vp(OB), n)),
         "u16 lt": (4.125, lambda: capi.call("agpu_compare", h, capi.CMP_LT, capi.U16, vp(A), vp(B), vp(OB), n))}
for rep in range(2):
    for variant, u in ((0, 1), (1, 1), (1, 2), (1, 4)):
        capi.call("agpu_pipeline_set_tuning", h, b"cmp_variant", variant)
        capi.call("agpu_pipeline_set_tuning", h, b"stream_unroll", u)
        for name, (bpr, f) in cases.items():
            f(); p.sync()
            ts = []
            for _ in range(7):
                q.begin(p); f(); q.end(p)
                ts.append(q.wait_for_results())
            ms = float(np.median(ts))
            print({"kernel": name, "variant": "ballot" if variant == 0 else f"vec U={u}", "ms": round(ms, 4), "TBps": round(bpr * n / ms / 1e9, 3)}, flush=True)
