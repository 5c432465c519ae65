#!/usr/bin/env python3
"""Does the RELATIVE placement of the three streams of f32 add (a, b, out) matter?  One 14 GiB allocation; a at 0, b at
4 GiB + d1, out at 8 GiB + d2; 1e9 rows; 10 launches each, medians.  Writes gpurun_out/stream_offsets.json."""
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
p = ArrowComputePipeline(dev, "off")
q = CmpQuery(dev)
h = p._handle
G = 1 << 30
n = 1_000_000_000
big = dev.create_empty_buffer(14 * G)
base = big.ptr
print("base address 0x%x (mod 4 GiB: 0x%x)" % (base, base % (4 * G)))
capi.call("agpu_synth_f32", h, C.c_void_p(base), 3 * G, 1, 0, C.c_float(-8), C.c_float(8))  # 12 GiB of values
p.sync()
K, M = 1 << 10, 1 << 20


def t(a, b, o, rows=n, reps=10):
    f = lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, C.c_void_p(a), C.c_void_p(b), C.c_void_p(o), rows)  # noqa: E731
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


rows = []
pairs = [(0, 0), (-280 * M, -560 * M), (0, 0)]
for d in (256, 4 * K, 16 * K, 64 * K, 256 * K, M, 2 * M, 4 * M, 16 * M, 64 * M, 256 * M):
    pairs += [(d, 2 * d), (d, 0), (0, d)]
pairs += [(-280 * M, 0), (0, -560 * M), (-280 * M, -280 * M), (2 * M + 4 * K, 6 * M + 12 * K), (0, 0)]
for d1, d2 in pairs:
    ms = t(base, base + 4 * G + d1, base + 8 * G + 512 * M + d2)  # out region starts past b's span for every d1 used here
    r = {"d1": d1, "d2": d2, "ms": round(ms, 4), "frac_8TBs": round(12 * n / ms / 1e6 / 8000, 4)}
    rows.append(r)
    print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": __doc__, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "stream_offsets.json"), "w"), indent=1)
