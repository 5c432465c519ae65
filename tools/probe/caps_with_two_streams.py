#!/usr/bin/env python3
"""DEV TOOL (round 6b): do sin / cos / log still want their occupancy caps now that they walk the column as two lock-step streams?
tuning wave_lds: 0 = each kernel's default cap, -1 = none; alternated in one process, 1e9 rows."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "s", fuse=False)
h = p._handle
q = CmpQuery(dev)
A, O = dev.create_table_buffers([4 * n] * 2)
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(0.001), C.c_float(1000))
p.sync()


def med(f, reps=9):
    for _ in range(3):
        f()
    p.sync()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


for name, op in (("sin", capi.UN_SIN), ("cos", capi.UN_COS), ("log", capi.UN_LOG)):
    r = {"kernel": name}
    for rnd in range(3):
        for cap, val in (("default cap", 0), ("no cap", -1), ("6800 B", 6800), ("10240 B", 10240)):
            p.set_tuning("wave_lds", val)
            r.setdefault(cap, []).append(round(8 * n / med(lambda: capi.call("agpu_unary", h, op, capi.F32, vp(A), vp(O), n)) / 1e6 / 8000, 4))
    p.set_tuning("wave_lds", 0)
    print(json.dumps(r), flush=True)
