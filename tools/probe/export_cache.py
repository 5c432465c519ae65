#!/usr/bin/env python3
"""to_arrow of a 1 GiB f32 column with 10 % nulls, repeated: the first export pays for fresh host pages, later ones land in
the export cache's pages (released by pyarrow when the previous result is dropped)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import arrow_gpu_amd as ag  # noqa: E402
import pyarrow as pa  # noqa: E402

dev = ag.GPU_DEVICE()
n = 1 << 28
rng = np.random.default_rng(0)
arr = pa.array(rng.standard_normal(n).astype(np.float32), mask=rng.random(n) < 0.1)
g = ag.from_arrow(arr, dev)
for i in range(5):
    t0 = time.perf_counter()
    back = ag.to_arrow(g)
    dt = time.perf_counter() - t0
    ok = back.null_count == arr.null_count and (i > 0 or back.equals(arr))
    print(f"to_arrow #{i}: {4 * n / dt / 1e9:.1f} GB/s, ok={ok}", flush=True)
    del back
