#!/usr/bin/env python3
"""f32 add (a, b read; out written) at 1e9 rows as a function of the distances D1 = b − a and D2 = out − a inside ONE
20 GiB allocation.  Medians of 8 launches."""
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
G, M, K = 1 << 30, 1 << 20, 1 << 10
n = 1_000_000_000
big = dev.create_empty_buffer(20 * G)
base = big.ptr
capi.call("agpu_synth_f32", h, C.c_void_p(base), 2 * G + 64 * M, 1, 0, C.c_float(-8), C.c_float(8))   # a and b regions (8.25 GiB)
p.sync()
vp = C.c_void_p


def t(f, reps=8):
    for _ in range(2):
        f()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


rows = []
S = 4001366016
cases = [(4 * G, 8 * G), (4 * G + 8 * K, 8 * G), (4 * G, 8 * G + 8 * K), (4 * G + 8 * K, 8 * G + 8 * K), (4 * G + 8 * K, 8 * G + 4 * K),
         (4 * G + 4 * K, 8 * G + 8 * K), (4 * G + 8 * K, 8 * G + 16 * K), (4 * G + 8 * K, 8 * G + 12 * K), (4 * G + 4 * K, 8 * G + 12 * K),
         (4 * G + 16 * K, 8 * G + 8 * K), (4 * G + 2 * M, 8 * G + 8 * K), (4 * G + 8 * K, 8 * G + 2 * M), (4 * G + 8 * K, 8 * G + 2 * M + 4 * K),
         (S, 2 * S), (S + 8 * K, 2 * S), (S, 2 * S + 8 * K), (S + 8 * K, 2 * S + 4 * K), (S + 4 * K, 2 * S + 8 * K), (S + 16 * K, 2 * S + 8 * K),
         (4 * G, 8 * G)]
for D1, D2 in cases:
    ms = t(lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(base), vp(base + D1), vp(base + D2), n))
    ms2 = t(lambda: capi.call("agpu_unary", h, capi.UN_NEG, capi.F32, vp(base), vp(base + D2), n))
    r = {"D1": hex(D1), "D2": hex(D2), "add_ms": round(ms, 4), "add_frac": round(12 * n / ms / 1e6 / 8000, 4),
         "neg_a_to_out_ms": round(ms2, 4), "neg_frac": round(8 * n / ms2 / 1e6 / 8000, 4)}
    rows.append(r)
    print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": __doc__, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "add_spacing.json"), "w"), indent=1)
