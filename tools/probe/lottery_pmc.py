#!/usr/bin/env python3
"""DEV TOOL: the compare's allocation lottery under hardware counters.  One process allocates bench.py's compare table N times (all kept, so
each lands on different physical memory), launches `eq → bitmap` 11 times on each (2 warm-up + 9 timed by HIP events) and prints the median
fraction of the HBM roof per allocation.  Under `rocprofv3 --pmc …` the compare dispatches appear in the same order, 11 per allocation:
tools/probe/lottery_pmc_join.py joins the two."""
import ctypes as C
import json
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
rows = []
REPS = 11  # 2 warm-up + 9 timed
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    ia, ib, va, vb, ob, ov = t = dev.create_table_buffers([4 * n] * 2 + [nb] * 4)
    keep.append(t)
    capi.call("agpu_synth_i32", h, vp(ia), n, 1, 0, 1024)
    capi.call("agpu_synth_i32", h, vp(ib), n, 2, 0, 1024)
    p.sync()
    ts = []
    for k in range(REPS):
        q.begin(p)
        capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(ob), n)
        q.end(p)
        t_ms = q.wait_for_results()
        if k >= 2:
            ts.append(t_ms)
    frac = 8.125 * n / float(np.median(ts)) / 8e9
    rows.append({"allocation": trial, "ia": hex(ia.ptr), "ob": hex(ob.ptr), "frac": round(frac, 4), "ms": [round(x, 4) for x in ts]})
    print(f"allocation {trial}: ia={ia.ptr:#x} ob={ob.ptr:#x} eq→ob {frac:.3f}", flush=True)
out = os.environ.get("LOTTERY_OUT")
if out:
    json.dump(rows, open(out, "w"), indent=1)
