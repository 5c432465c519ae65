#!/usr/bin/env python3
"""Boolean put (agpu_put_bits_bounded) at 2^24 … 2^28 random rows into / out of 2^28-bit bitmaps: the direct kernel (one device-scope
atomic per row, tuning gather_bucket = 1) against the form bucketed by destination region (= 2).  One process, same buffers."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "pb"); q = CmpQuery(dev); h = p._handle
vp = lambda b: C.c_void_p(b.ptr)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 28
idx, idx2 = dev.create_empty_buffer(4*n), dev.create_empty_buffer(4*n)
capi.call("agpu_synth_i32", h, vp(idx), n, 2, 0, n); capi.call("agpu_synth_i32", h, vp(idx2), n, 3, 0, n)
vb, ov = dev.create_empty_buffer(n // 8 + 64), dev.create_empty_buffer(n // 8 + 64)
capi.call("agpu_synth_bits", h, vp(vb), n, 7, 0, C.c_double(0.5)); capi.call("agpu_synth_bits", h, vp(ov), n, 8, 0, C.c_double(0.5))
p.sync()
for lg, mode in ((24, 1), (24, 2), (25, 1), (25, 2), (26, 1), (26, 2), (28, 1), (28, 2), (0, 1), (0, 2)):  # 0: all n rows
    m = 1 << lg if lg else n
    if m > n:
        continue
    p.set_tuning("gather_bucket", mode)
    f = lambda: capi.call("agpu_put_bits_bounded", h, vp(vb), n, vp(idx), vp(ov), n, vp(idx2), m)
    f(); p.sync(); ts = []
    for _ in range(5):
        q.begin(p); f(); q.end(p); ts.append(q.wait_for_results())
    print("put_bits mode %d rows %d: %.3f ms = %.1f G rows/s" % (mode, m, np.median(ts), m / np.median(ts) / 1e6), flush=True)
