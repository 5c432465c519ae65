#!/usr/bin/env python3
"""DEV TOOL: does the compare kernel's 0.81 ↔ 0.87 follow the ALLOCATION?  One process allocates bench.py's compare table
(two i32 columns + four bitmaps) several times — earlier ones are kept, so each lands on different physical memory — and
times eq + validity and the plain eq on each."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
nb = (n + 63) // 64 * 8
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "pl")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
first = dev.create_table_buffers([4 * n] * 3)  # bench.py's f32 table comes first
keep = []
sep = dev.create_empty_buffer(nb)


def med(f, reps=9):
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    ia, ib, va, vb, ob, ov = t = dev.create_table_buffers([4 * n] * 2 + [nb] * 4)
    keep.append(t)
    capi.call("agpu_synth_i32", h, vp(ia), n, 1, 0, 1024)
    capi.call("agpu_synth_i32", h, vp(ib), n, 2, 0, 1024)
    capi.call("agpu_synth_bits", h, vp(va), n, 3, 0, C.c_double(0.9))
    capi.call("agpu_synth_bits", h, vp(vb), n, 4, 0, C.c_double(0.9))
    p.sync()
    ev = med(lambda: capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(va), vp(vb), vp(ob), vp(ov), n))
    e = med(lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(ob), n))
    outs = []
    for name, o in (("ov", ov), ("vb", vb), ("va", va), ("separate", sep)):  # the same two columns, the result bitmap elsewhere
        outs.append(f"{name} {8.125 * n / med(lambda: capi.call('agpu_compare', h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(o), n)) / 8e9:.3f}")
    print(f"allocation {trial}: ia={ia.ptr:#x}  eq+validity {8.5 * n / ev / 8e9:.3f}   eq→ob {8.125 * n / e / 8e9:.3f}   eq→ " + "  ".join(outs), flush=True)
