#!/usr/bin/env python3
"""i32 eq → bitmap, ballot variant vs vector variant (tuning cmp_variant), as a function of the relative placement of the
two input streams (b shifted by k × 4 KiB inside one allocation).  1e9 rows, medians of 10."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "col")
q = CmpQuery(dev)
h = p._handle
G, K = 1 << 30, 1 << 10
n = 1_000_000_000
pre = dev.create_table_buffers([4 * n] * 3) if "--quick" in sys.argv else None  # what bench.py allocates before its compare table
big = dev.create_empty_buffer(14 * G)
base = big.ptr
capi.call("agpu_synth_i32", h, C.c_void_p(base), 3 * G, 1, 0, 1024)
p.sync()
vp = C.c_void_p


def t(f, reps=10):
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


rows = []
QUICK = "--quick" in sys.argv  # the default variant only: is the best colour the same in every process / on every box?
for variant, unroll in (((0, 1),) if QUICK else ((0, 1), (1, 1), (1, 2), (1, 4))):
    for k1 in (0, 1, 2, 3, 4, 8, 16, 64, 0):
        p.set_tuning("cmp_variant", variant)
        p.set_tuning("stream_unroll", unroll)
        a, b, o = base, base + 4 * G + k1 * 4 * K, base + 9 * G
        ms = t(lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(a), vp(b), vp(o), n))
        r = {"cmp_variant": variant, "unroll": unroll, "k1": k1, "ms": round(ms, 4), "frac": round(8.125 * n / ms / 1e6 / 8000, 4)}
        rows.append(r)
        print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
if not QUICK:
    json.dump({"what": __doc__, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "cmp_colour.json"), "w"), indent=1)
