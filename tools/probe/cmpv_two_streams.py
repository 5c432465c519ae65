#!/usr/bin/env python3
"""i32 eq + validity at 1e9 rows: the fused launch against compare and validity AND on TWO streams (joined before the end
event).  Table-allocated buffers, one process."""
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
p = ArrowComputePipeline(dev, "main")
p2 = ArrowComputePipeline(dev, "aux")
q = CmpQuery(dev)
h, h2 = p._handle, p2._handle
n = 1_000_000_000
nb = (n + 63) // 64 * 8
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
ia, ib, va, vb, ob, ov = dev.create_table_buffers([4 * n] * 2 + [nb] * 4)
capi.call("agpu_synth_i32", h, vp(ia), n, 3, 0, 1024)
capi.call("agpu_synth_i32", h, vp(ib), n, 4, 0, 1024)
capi.call("agpu_synth_bits", h, vp(va), n, 5, 0, C.c_double(0.9))
capi.call("agpu_synth_bits", h, vp(vb), n, 6, 0, C.c_double(0.9))
p.sync()


def t(f, reps=10):
    for _ in range(3):
        f()
    p.sync(); p2.sync()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


def fused():
    capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(va), vp(vb), vp(ob), vp(ov), n)


def two_streams():
    capi.call("agpu_bitmap_binary", h2, capi.OP_AND, vp(va), vp(vb), vp(ov), n)
    capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(ob), n)
    p2.finish()
    p.wait_pipeline(p2)


def two_streams_and_last():
    capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(ob), n)
    capi.call("agpu_bitmap_binary", h2, capi.OP_AND, vp(va), vp(vb), vp(ov), n)
    p2.finish()
    p.wait_pipeline(p2)


for rep in range(3):
    for label, f in (("fused (validity blocks first)", fused), ("two streams, AND issued first", two_streams), ("two streams, AND issued second", two_streams_and_last),
                     ("compare alone", lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(ob), n))):
        ms = t(f)
        print(json.dumps({"form": label, "ms": round(ms, 4), "frac_of_8.5B": round(8.5 * n / ms / 1e6 / 8000, 4)}), flush=True)
