#!/usr/bin/env python3
"""DEV TOOL (GPU box): the reference's test- and example-sized flow (examples/simple.rs: two 5-element arrays, add, values()) through the Python
host: µs per step, median of 300.  SYNC_SPIN=-1: the waits and small copies without the mailbox (R5.10)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import arrow_gpu_amd as ag
from arrow_gpu_amd import _capi as capi
if os.environ.get("SYNC_SPIN"): capi.call("agpu_set_tuning", b"sync_spin", int(os.environ["SYNC_SPIN"]))
dev = ag.GPU_DEVICE()
def med(fn, k=300):
    for _ in range(30): fn()
    ts = []
    for _ in range(k):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e6)
    return sorted(ts)[k // 2]
for n in (5, 100, 900, 10_000, 60_000):
    x = np.arange(n, dtype=np.float32); y = x + 1
    a = ag.Float32ArrayGPU.from_slice(x, dev); b = ag.Float32ArrayGPU.from_slice(y, dev)
    r = a.add(b)
    print(f"n={n}: from_slice {med(lambda: ag.Float32ArrayGPU.from_slice(x, dev)):.1f}  add {med(lambda: a.add(b)):.1f}  raw_values {med(lambda: r.raw_values()):.1f}  "
          f"from_slice x2 + add + raw_values {med(lambda: ag.Float32ArrayGPU.from_slice(x, dev).add(ag.Float32ArrayGPU.from_slice(y, dev)).raw_values()):.1f} us", flush=True)
    assert np.array_equal(np.asarray(a.add(b).raw_values()), x + y)
