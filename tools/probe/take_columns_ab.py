#!/usr/bin/env python3
"""DEV TOOL (round 6b): k columns of one table by one random index column — agpu_take_columns against k agpu_take calls, 2^28 rows."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

m = 1 << 28
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "s", fuse=False)
h = p._handle
q = CmpQuery(dev)
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
K = 4
V = [dev.create_empty_buffer(4 * m) for _ in range(K)]
OUT = [dev.create_empty_buffer(4 * m) for _ in range(K)]
I = dev.create_empty_buffer(4 * m)
for k, v in enumerate(V):
    capi.call("agpu_synth_i32", h, vp(v), m, 70 + k, 0, 0)
capi.call("agpu_synth_i32", h, vp(I), m, 8, 0, m)
p.sync()


def med(f, reps=5):
    f(); p.sync()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


for k in (1, 2, 3, 4):
    widths = (C.c_int32 * k)(*([4] * k))
    vals = (C.c_void_p * k)(*[v.ptr for v in V[:k]])
    outs = (C.c_void_p * k)(*[o.ptr for o in OUT[:k]])
    one = med(lambda: capi.call("agpu_take_columns", h, k, widths, vals, m, vp(I), outs, m))
    sep = med(lambda: [capi.call("agpu_take", h, 4, vp(V[c]), m, vp(I), vp(OUT[c]), m) for c in range(k)])
    print(json.dumps({"columns": k, "take_columns_ms": round(one, 4), "k_takes_ms": round(sep, 4), "per_column_ms": round(one / k, 4),
                      "G_rows_per_s_per_column": round(m / (one / k) / 1e6, 1), "speedup": round(sep / one, 3)}), flush=True)
