#!/usr/bin/env python3
"""Phase breakdown of the bucketed take/put passes: needs the BKT_PROFILE variant library (AGPU_LIB=…/libagpu_bktprof.so).
Workgroup 64's thread 0 stamps the cycle counter between phases; printed as deltas in microseconds at 100 MHz ticks."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice  # noqa: E402

dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "phases")
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
n = 1 << 28
values, out, idx, idx2 = (dev.create_empty_buffer(4 * n) for _ in range(4))
capi.call("agpu_synth_i32", h, vp(values), n, 1, 0, 0)
capi.call("agpu_synth_i32", h, vp(idx), n, 2, 0, n)
capi.call("agpu_synth_i32", h, vp(idx2), n, 3, 0, n)
p.set_tuning("gather_bucket", 2)
lib = capi.lib()
names = {0: "P partition", 1: "G gather (second sort)", 2: "F store"}
labels = ["start→rows in regs", "ranks (LDS atomics)", "scan", "LDS scatter", "reserve (global atomics)", "copy-out issue"]
for what in ("take", "put"):
    for _ in range(2):
        if what == "take":
            capi.call("agpu_take", h, 4, vp(values), n, vp(idx), vp(out), n)
        else:
            capi.call("agpu_put_bounded", h, 4, vp(values), n, vp(idx), vp(out), n, vp(idx2), n)
        p.sync()
    st = np.zeros(64, np.uint64)
    assert lib.agpu_debug_bkt_stamps(C.c_void_p(st.ctypes.data)) == 0
    st = st.reshape(4, 16).astype(np.int64)
    print(what)
    for k in range(3):
        s = st[k]
        d = [(s[i + 1] - s[i]) for i in range(0, 6)]
        print("  ", names[k], {lab: int(x) for lab, x in zip(labels, d)}, "ticks (s_memtime, 100 MHz = 10 ns)")
