#!/usr/bin/env python3
"""DEV TOOL: put (scatter) by index pattern — which side costs what (2^26 rows into / out of 1 GiB columns)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n_col, m = 1 << 28, 1 << 26
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "put")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
SRC, DST = dev.create_empty_buffer(4 * n_col), dev.create_empty_buffer(4 * n_col)
capi.call("agpu_synth_i32", h, vp(SRC), n_col, 1, 0, 0)
rng = np.random.default_rng(3)
seq = (np.arange(m, dtype=np.uint32) * 4) % n_col      # distinct lines? no: consecutive packs of 4 rows — streaming
rand = rng.permutation(n_col)[:m].astype(np.uint32)    # unique random rows
pats = {"seq": dev.create_gpu_buffer_with_data(np.arange(m, dtype=np.uint32)), "rand": dev.create_gpu_buffer_with_data(rand)}
p.sync()
for s_name in ("seq", "rand"):
    for d_name in ("seq", "rand"):
        def f():
            capi.call("agpu_put_bounded", h, 4, vp(SRC), n_col, vp(pats[s_name]), vp(DST), n_col, vp(pats[d_name]), m)
        f(); p.sync()
        ts = []
        for _ in range(7):
            q.begin(p); f(); q.end(p)
            ts.append(q.wait_for_results())
        ms = float(np.median(ts))
        print({"src_idx": s_name, "dst_idx": d_name, "ms": round(ms, 4), "Grows_per_s": round(m / ms / 1e6, 2)}, flush=True)
assert capi.lib().agpu_pipeline_sync(h) == 0
