#!/usr/bin/env python3
"""take / Boolean take / put at 2^27 rows by INDEX DISTRIBUTION: uniform random, sorted ascending (take after a filter), sequential,
sorted with gaps, 90 % of the rows inside one 64-element window, few distinct values — direct kernels against the pipelines and
against what the auto policy picks.  One process, same buffers."""
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
p = ArrowComputePipeline(dev, "dist")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
n = 1 << 27
n_src = 1 << 28
values, out = dev.create_empty_buffer(4 * n_src), dev.create_empty_buffer(4 * n_src)
capi.call("agpu_synth_i32", h, vp(values), n_src, 1, 0, 0)
bits, obits = dev.create_empty_buffer(n_src // 8 + 64), dev.create_empty_buffer(n_src // 8 + 64)
capi.call("agpu_synth_bits", h, vp(bits), n_src, 7, 0, C.c_double(0.5))
rng = np.random.default_rng(1)
dists = {
    "uniform": lambda: rng.integers(0, n_src, n, dtype=np.uint32),
    "sorted": lambda: np.sort(rng.integers(0, n_src, n, dtype=np.uint32)),
    "sequential": lambda: np.arange(n, dtype=np.uint32),
    "sorted_in_blocks_of_4096": lambda: np.sort(rng.integers(0, n_src, n, dtype=np.uint32).reshape(-1, 4096), axis=1).reshape(-1),
    "hot_window_90pct": lambda: np.where(rng.random(n) < 0.9, rng.integers(1000, 1064, n), rng.integers(0, n_src, n)).astype(np.uint32),
    "256_distinct": lambda: (rng.integers(0, 256, n) * 1048573 % n_src).astype(np.uint32),
}
perm = None
res = {}


def med(f, iters=5):
    f(), f()
    p.sync()
    ts = []
    for _ in range(iters):
        q.begin(p)
        f()
        q.end(p)
        ts.append(q.wait_for_results())
    return round(float(np.median(ts)), 4)


for name, gen in dists.items():
    idx = gen()
    didx = dev.create_gpu_buffer_with_data(idx)
    row = {}
    for label, mode in (("direct", 1), ("pipeline", 2), ("auto", 0)):
        p.set_tuning("gather_bucket", mode)
        row[f"take_{label}_ms"] = med(lambda: capi.call("agpu_take", h, 4, vp(values), n_src, vp(didx), vp(out), n))
        row[f"take_bits_{label}_ms"] = med(lambda: capi.call("agpu_take_bits", h, vp(bits), n_src, vp(didx), vp(obits), n))
    res[name] = row
    print(name, row, flush=True)
    del didx
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"rows": n, "source_elements": n_src, "distributions": res}, open(os.path.join(ROOT, "gpurun_out", "r03_take_distributions.json"), "w"), indent=1)
