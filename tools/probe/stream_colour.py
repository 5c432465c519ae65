#!/usr/bin/env python3
"""Follow-up to stream_offsets.py: does staggering the streams of a kernel by small multiples of 4 KiB ("colouring")
help systematically?  One 14 GiB allocation; f32 add (a, b, out), f32 neg (a, out), i32 eq → bitmap (a, b) at 1e9 rows
with the second / third stream shifted by k × 4 KiB.  Medians of 10 launches; writes gpurun_out/stream_colour.json."""
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
big = dev.create_empty_buffer(14 * G)
base = big.ptr
capi.call("agpu_synth_f32", h, C.c_void_p(base), 3 * G, 1, 0, C.c_float(-8), C.c_float(8))
p.sync()


def t(f, reps=10):
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


rows = []
vp = C.c_void_p
for rep in range(2):
    for k1, k2 in ((0, 0), (1, 2), (1, 0), (0, 1), (2, 4), (3, 6), (4, 8), (5, 10), (8, 16), (1, 3), (2, 1), (7, 14), (16, 32), (0, 0)):
        d1, d2 = k1 * 4 * K, k2 * 4 * K
        a, b, o = base, base + 4 * G + d1, base + 9 * G + d2
        ms_add = t(lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(a), vp(b), vp(o), n))
        ms_neg = t(lambda: capi.call("agpu_unary", h, capi.UN_NEG, capi.F32, vp(a), vp(b), n))
        ms_eq = t(lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(a), vp(b), vp(o), n))
        r = {"k1": k1, "k2": k2, "add_ms": round(ms_add, 4), "add_frac": round(12 * n / ms_add / 1e6 / 8000, 4),
             "neg_ms": round(ms_neg, 4), "neg_frac": round(8 * n / ms_neg / 1e6 / 8000, 4),
             "eq_ms": round(ms_eq, 4), "eq_frac": round(8.125 * n / ms_eq / 1e6 / 8000, 4)}
        rows.append(r)
        print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": __doc__, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "stream_colour.json"), "w"), indent=1)
