#!/usr/bin/env python3
"""DEV TOOL: put with both index columns uniformly random at 2^23 … 2^28 rows (arrays of as many elements): the direct scatter, round 3's pair
pipeline (16 Ki-pair tiles, `gather_offsets` 4) and round 4's (32 Ki-pair tiles in P and G, paired reservations: 0).  One process, alternating."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "sweep")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
nmax = 1 << 28
values, out, idx, idx2 = (dev.create_empty_buffer(4 * nmax) for _ in range(4))
capi.call("agpu_synth_i32", h, vp(values), nmax, 1, 0, 0)
for lg in range(23, 29):
    n = 1 << lg
    capi.call("agpu_synth_i32", h, vp(idx), n, 2, 0, n)
    capi.call("agpu_synth_i32", h, vp(idx2), n, 3, 0, n)
    p.sync()
    res = {}
    for rnd in range(2):
        for name, bucket, offs in (("direct", 1, 0), ("r3 16Ki", 2, 4), ("r4 32Ki", 2, 0)):
            p.set_tuning("gather_bucket", bucket)
            p.set_tuning("gather_offsets", offs)
            f = lambda: capi.call("agpu_put_bounded", h, 4, vp(values), n, vp(idx), vp(out), n, vp(idx2), n)  # noqa: E731
            f(), f()
            p.sync()
            ts = []
            for _ in range(7):
                q.begin(p); f(); q.end(p)
                ts.append(q.wait_for_results())
            res.setdefault(name, []).append(float(np.median(ts)))
    print(f"2^{lg} rows: " + " | ".join(f"{k} {min(v):.4f} ms = {n / min(v) / 1e6:.1f} G rows/s" for k, v in res.items()), flush=True)
p.set_tuning("gather_bucket", 0)
p.set_tuning("gather_offsets", 0)
