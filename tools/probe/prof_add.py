#!/usr/bin/env python
"""DEV TOOL (GPU box): cProfile of the Python host's immediate `a.add(b)` at 100 elements — where its ≈ 10 µs go (four C-ABI calls and two frees at 1.5–2 µs each; Python's own share is ≈ 2 µs)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np
import arrow_gpu_amd as ag
dev = ag.GPU_DEVICE()
x = np.arange(100, dtype=np.float32)
a = ag.Float32ArrayGPU.from_slice(x, dev); b = ag.Float32ArrayGPU.from_slice(x + 1, dev)
for _ in range(200): a.add(b)
dev.sync()
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): a.add(b)
pr.disable(); dev.sync()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
