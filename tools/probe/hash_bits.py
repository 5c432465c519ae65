#!/usr/bin/env python3
"""Which address bits feed the memory-channel hash, and with which weight?  i32 eq → bitmap (two read streams, no big
write stream) at 1e9 rows inside ONE fresh 22 GiB allocation; the distance D between the two columns is
  (a) 2^32 XOR 2^j            — from the fully conflicting distance, what does flipping bit j alone buy?
  (b) 2^32 + 2^13 XOR 2^j     — from the best distance, which bits cancel it?
Medians of 6 launches.  Writes gpurun_out/hash_bits.json."""
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
p = ArrowComputePipeline(dev, "hash")
q = CmpQuery(dev)
h = p._handle
G = 1 << 30
n = 1_000_000_000
big = dev.create_empty_buffer(22 * G)
base = big.ptr
print("base 0x%x" % base)
capi.call("agpu_synth_i32", h, C.c_void_p(base), 5 * G + (G >> 1), 1, 0, 1024)  # 22 GiB of values
p.sync()
vp = C.c_void_p
out = base + 21 * G + (G >> 1)


def t(D, reps=6):
    f = lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(base), vp(base + D), vp(out), n)  # noqa: E731
    for _ in range(2):
        f()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


rows = []
for label, D0 in (("from 2^32", 1 << 32), ("from 2^32 + 2^13", (1 << 32) + (1 << 13))):
    ms0 = t(D0)
    rows.append({"series": label, "bit": None, "D": hex(D0), "ms": round(ms0, 4), "frac": round(8.125 * n / ms0 / 1e6 / 8000, 4)})
    print(json.dumps(rows[-1]), flush=True)
    for j in list(range(8, 32)) + [33, 34]:
        D = D0 ^ (1 << j)
        if D + 4 * n > 21 * G:
            continue
        ms = t(D)
        rows.append({"series": label, "bit": j, "D": hex(D), "ms": round(ms, 4), "frac": round(8.125 * n / ms / 1e6 / 8000, 4)})
        print(json.dumps(rows[-1]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": __doc__, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "hash_bits.json"), "w"), indent=1)
