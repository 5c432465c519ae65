#!/usr/bin/env python3
"""i32 eq → bitmap (two read streams) at 1e9 rows as a function of the distance D between the two input columns inside
ONE 20 GiB allocation: which bits of D decide between the 0.78 and the 0.85 regime?  Medians of 8 launches."""
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
print("base 0x%x" % base)
capi.call("agpu_synth_i32", h, C.c_void_p(base), 4 * G, 1, 0, 1024)   # 16 GiB of values
p.sync()
vp = C.c_void_p
out = base + 19 * G


def t(f, reps=8):
    for _ in range(2):
        f()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


rows = []
S = 4001366016  # what two consecutive 4e9-byte pool blocks are apart (1908 granules of 2 MiB)
cases = [("4 GiB", 4 * G)] + [(f"4 GiB + 2^{j}", 4 * G + (1 << j)) for j in range(8, 32)]
cases += [(f"S + {k}*4K", S + k * 4 * K) for k in range(0, 9)]
cases += [(f"{g} GiB + 8K", int(g * G) + 8 * K) for g in (3.75, 4.5, 5, 6, 7, 8, 10)]
cases += [(f"{g} GiB", int(g * G)) for g in (3.75, 4.5, 5, 6, 8)]
cases += [("4 GiB + 8K + 2^%d" % j, 4 * G + 8 * K + (1 << j)) for j in (21, 24, 27, 30)]
for label, D in cases:
    ms = t(lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(base), vp(base + D), vp(out), n))
    r = {"D": label, "D_hex": hex(D), "ms": round(ms, 4), "frac": round(8.125 * n / ms / 1e6 / 8000, 4)}
    rows.append(r)
    print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": __doc__, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "cmp_spacing.json"), "w"), indent=1)
