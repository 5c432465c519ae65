#!/usr/bin/env python3
"""The bench's second kernel (i32 eq → bitmap with fused validity AND, 8.5 B/row) at 1e9 rows inside ONE 20 GiB allocation:
distance D between the two value columns × placement of the four bitmaps.  Medians of 8 launches."""
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
capi.call("agpu_synth_i32", h, C.c_void_p(base), 4 * G, 1, 0, 1024)
capi.call("agpu_synth_bits", h, C.c_void_p(base + 16 * G), 8 * n, 5, 0, C.c_double(0.9))  # 1 GB of validity bits
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
B = base + 16 * G
for D in (4 * G, 4 * G + 8 * K, S, 4 * G + 4 * K, 5 * G + 8 * K):
    for label, (va, vb, ob, ov) in (("bitmaps 128 MiB apart", (B, B + 128 * M, B + 256 * M, B + 384 * M)),
                                    ("bitmaps 128 MiB + 8K/4K/12K", (B, B + 128 * M + 8 * K, B + 256 * M + 4 * K, B + 384 * M + 12 * K)),
                                    ("bitmaps back to back (125000192 B)", (B, B + 125000192, B + 2 * 125000192, B + 3 * 125000192))):
        ms = t(lambda: capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, vp(base), vp(base + D), vp(va), vp(vb), vp(ob), vp(ov), n))
        r = {"D": hex(D), "bitmaps": label, "ms": round(ms, 4), "frac": round(8.5 * n / ms / 1e6 / 8000, 4)}
        rows.append(r)
        print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": __doc__, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "cmpv_spacing.json"), "w"), indent=1)
