#!/usr/bin/env python3
"""i32 eq + validity at 1e9 rows: the fused launch (agpu_compare_validity) against compare + bitmap AND as two launches,
for two placements of the value columns (slab, D = 4 GiB and 4 GiB + 8 KiB) and for pool-allocated buffers."""
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
nb = (n + 63) // 64 * 8
vp = C.c_void_p


def t(f, reps=10):
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


def run(label, a, b, va, vb, ob, ov):
    fused = t(lambda: capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, vp(a), vp(b), vp(va), vp(vb), vp(ob), vp(ov), n))
    cmp_only = t(lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(a), vp(b), vp(ob), n))
    and_only = t(lambda: capi.call("agpu_bitmap_binary", h, capi.OP_AND, vp(va), vp(vb), vp(ov), n))

    def two():
        capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(a), vp(b), vp(ob), n)
        capi.call("agpu_bitmap_binary", h, capi.OP_AND, vp(va), vp(vb), vp(ov), n)
    split = t(two)
    r = {"placement": label, "fused_ms": round(fused, 4), "fused_frac": round(8.5 * n / fused / 1e6 / 8000, 4),
         "compare_ms": round(cmp_only, 4), "and_ms": round(and_only, 4), "two_launches_ms": round(split, 4),
         "two_launches_frac": round(8.5 * n / split / 1e6 / 8000, 4)}
    print(json.dumps(r), flush=True)
    return r


rows = []
big = dev.create_empty_buffer(20 * G)
base = big.ptr
capi.call("agpu_synth_i32", h, vp(base), 4 * G, 1, 0, 1024)
B = base + 16 * G
capi.call("agpu_synth_bits", h, vp(B), 8 * n, 5, 0, C.c_double(0.9))
p.sync()
for D in (4 * G, 4 * G + 8 * K):
    rows.append(run(f"slab, D = {hex(D)}", base, base + D, B, B + 128 * M, B + 256 * M, B + 384 * M))
del big
capi.call("agpu_device_trim", dev._handle)
ia, ib = dev.create_empty_buffer(4 * n), dev.create_empty_buffer(4 * n)
va, vb, ob, ov = (dev.create_empty_buffer(nb) for _ in range(4))
capi.call("agpu_synth_i32", h, vp(ia.ptr), n, 3, 0, 1024)
capi.call("agpu_synth_i32", h, vp(ib.ptr), n, 4, 0, 1024)
capi.call("agpu_synth_bits", h, vp(va.ptr), n, 5, 0, C.c_double(0.9))
capi.call("agpu_synth_bits", h, vp(vb.ptr), n, 6, 0, C.c_double(0.9))
p.sync()
rows.append(run("six pool blocks", ia.ptr, ib.ptr, va.ptr, vb.ptr, ob.ptr, ov.ptr))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": __doc__, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "cmpv_split.json"), "w"), indent=1)
