#!/usr/bin/env python3
"""put / Boolean put at 2^26 rows by INDEX DISTRIBUTION on each side (uniform random / sorted / sequential): direct kernels against the
bucketed forms and against what the auto policy picks.  One process, same buffers."""
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
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
n_col = 1 << 28
src, dst = dev.create_empty_buffer(4 * n_col), dev.create_empty_buffer(4 * n_col)
capi.call("agpu_synth_i32", h, vp(src), n_col, 1, 0, 0)
sb, db = dev.create_empty_buffer(n_col // 8 + 64), dev.create_empty_buffer(n_col // 8 + 64)
capi.call("agpu_synth_bits", h, vp(sb), n_col, 7, 0, C.c_double(0.5))
capi.call("agpu_synth_bits", h, vp(db), n_col, 8, 0, C.c_double(0.5))
rng = np.random.default_rng(2)
gens = {"uniform": lambda: rng.integers(0, n_col, n, dtype=np.uint32),
        "sorted": lambda: np.sort(rng.integers(0, n_col, n, dtype=np.uint32)),
        "sequential": lambda: np.arange(n, dtype=np.uint32)}


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


res = {}
for sname in gens:
    for dname in gens:
        si, di = dev.create_gpu_buffer_with_data(gens[sname]()), dev.create_gpu_buffer_with_data(gens[dname]())
        row = {}
        for label, mode in (("direct", 1), ("bucketed", 2), ("auto", 0)):
            p.set_tuning("gather_bucket", mode)
            row[f"put_{label}_ms"] = med(lambda: capi.call("agpu_put_bounded", h, 4, vp(src), n_col, vp(si), vp(dst), n_col, vp(di), n))
            row[f"put_bits_{label}_ms"] = med(lambda: capi.call("agpu_put_bits_bounded", h, vp(sb), n_col, vp(si), vp(db), n_col, vp(di), n))
        res[f"src {sname}, dst {dname}"] = row
        print(f"src {sname}, dst {dname}", row, flush=True)
        del si, di
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"rows": n, "column_elements": n_col, "distributions": res}, open(os.path.join(ROOT, "gpurun_out", "r03_put_distributions.json"), "w"), indent=1)
