#!/usr/bin/env python3
"""put of 2^28 4-byte rows at the four corners of index locality (sequential / uniformly random source and destination columns):
what the auto policy's device-side choice costs against the full pair pipeline and the direct scatter.  One process, same buffers."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "lr"); q = CmpQuery(dev); h = p._handle
vp = lambda b: C.c_void_p(b.ptr)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 28
src, dst, si, di = (dev.create_empty_buffer(4*n) for _ in range(4))
capi.call("agpu_synth_i32", h, vp(src), n, 1, 0, 0)
capi.call("agpu_synth_i32", h, vp(di), n, 3, 0, n)
si_np = np.arange(n, dtype=np.uint32)
capi.call("agpu_upload", h, vp(si), C.c_void_p(si_np.ctypes.data), 4*n)
p.sync()
sq = dev.create_empty_buffer(4*n)
capi.call("agpu_upload", h, vp(sq), C.c_void_p(si_np.ctypes.data), 4*n)
rnd2 = dev.create_empty_buffer(4*n)
capi.call("agpu_synth_i32", h, vp(rnd2), n, 2, 0, n)
p.sync()
for label, a, b in (("src sequential, dst uniform", si, di), ("src uniform, dst sequential", rnd2, sq), ("src uniform, dst uniform", rnd2, di), ("src sequential, dst sequential", si, sq)):
    for mode in (0, 2, 1):
        p.set_tuning("gather_bucket", mode)
        f = lambda: capi.call("agpu_put_bounded", h, 4, vp(src), n, vp(a), vp(dst), n, vp(b), n)
        f(); p.sync(); ts=[]
        for _ in range(3):
            q.begin(p); f(); q.end(p); ts.append(q.wait_for_results())
        print(label, "| mode", {0: "auto", 2: "full pipeline", 1: "direct"}[mode], "ms", round(float(np.median(ts)),4), "G rows/s", round(n/np.median(ts)/1e6,1), flush=True)
