#!/usr/bin/env python3
"""DEV TOOL: per-phase shader-clock stamps of the put's gather pass (bkt_gather2_kernel, 32 Ki-pair tiles) for workgroup 64's thread 0, and of round
3's G (gather_offsets 4, its second half only).  Needs the BKT_PROFILE variant library: hipcc … -DBKT_PROFILE -c swizzle.hip, linked like the
product, AGPU_LIB=<that .so>."""
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
L2 = ["load+ranks1", "scan1+pos", "scatter1", "gather", "own values+ranks2", "scan2", "reserve+pos2", "values scatter+deltas", "vj read", "dst scatter", "stores issue"]
L1 = ["gathers issued→rows", "ranks", "scan", "LDS scatter", "reserve", "copy-out issue"]
for mode in (0, 4, 0, 4):
    p.set_tuning("gather_offsets", mode)
    for _ in range(2):
        capi.call("agpu_put_bounded", h, 4, vp(values), n, vp(idx), vp(out), n, vp(idx2), n)
        p.sync()
    st = np.zeros(64, np.uint64)
    assert lib.agpu_debug_bkt_stamps(C.c_void_p(st.ctypes.data)) == 0
    s = st.reshape(4, 16).astype(np.int64)[1]
    if mode == 0:
        print("G 32 Ki pairs:", dict(zip(L2, [int(s[i + 1] - s[i]) for i in range(11)])), "total", int(s[11] - s[0]))
    else:
        print("G 16 Ki pairs (second half):", dict(zip(L1, [int(s[i + 1] - s[i]) for i in range(6)])), "total", int(s[6] - s[0]))
