#!/usr/bin/env python3
"""put at 2^28 random rows, pair pipeline: range starts by global atomics (gather_offsets = 1) vs from the column scan of
per-tile counts for P only (= 3) and for P and G (= 2: G's starts from a count pass over P's output; measured a wash, not the default), all with
XCD-contiguous tiles.  One process, same buffers."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "put")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
n = 1 << 28
values, out, idx, idx2 = (dev.create_empty_buffer(4 * n) for _ in range(4))
capi.call("agpu_synth_i32", h, vp(values), n, 1, 0, 0)
capi.call("agpu_synth_i32", h, vp(idx), n, 2, 0, n)
capi.call("agpu_synth_i32", h, vp(idx2), n, 3, 0, n)
p.sync()
p.set_tuning("gather_bucket", 2)
for off in [int(x) for x in os.environ.get("AB_MODES", "1,3,2,1,3,2").split(",")]:
    p.set_tuning("gather_offsets", off)
    for what in ("put", "take_pairs"):
        if what == "take_pairs":
            p.set_tuning("gather_bucket", 3)
            f = lambda: capi.call("agpu_take", h, 4, vp(values), n, vp(idx), vp(out), n)  # noqa: E731
        else:
            p.set_tuning("gather_bucket", 2)
            f = lambda: capi.call("agpu_put_bounded", h, 4, vp(values), n, vp(idx), vp(out), n, vp(idx2), n)  # noqa: E731
        f(), f()
        p.sync()
        ts = []
        for _ in range(5):
            q.begin(p)
            f()
            q.end(p)
            ts.append(q.wait_for_results())
        ms = float(np.median(ts))
        print(f"gather_offsets={off} {what}: {ms:.4f} ms = {n / ms / 1e6:.1f} G rows/s", flush=True)
