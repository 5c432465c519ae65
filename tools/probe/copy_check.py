#!/usr/bin/env python3
"""DEV TOOL: agpu_copy (clone_buffer) — the stream kernel for big aligned copies against the runtime path, checksum-checked."""
import ctypes as C, numpy as np, sys
sys.path.insert(0, '/root/repo')
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "c"); q = CmpQuery(dev); h = p._handle
n = 1_000_000_000
A, O = dev.create_table_buffers([4 * n, 4 * n])
capi.call("agpu_synth_f32", h, C.c_void_p(A.ptr), n, 1, 0, C.c_float(-8), C.c_float(8))
for nbytes, soff, doff in ((4 * n, 0, 0), (4 * n - 8, 0, 0), (4 * n - 64, 16, 32), (1 << 20, 0, 0), (12344, 8, 16), (4 * n - 64, 8, 0)):
    capi.call("agpu_memset", h, C.c_void_p(O.ptr), 0, 4 * n)
    f = lambda: capi.call("agpu_copy", h, C.c_void_p(O.ptr + doff), C.c_void_p(A.ptr + soff), nbytes)
    f(); p.sync(); ts = []
    for _ in range(5):
        q.begin(p); f(); q.end(p); ts.append(q.wait_for_results())
    c1, c2 = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
    capi.call("agpu_checksum", h, C.c_void_p(A.ptr + soff), nbytes // 8 * 8, C.c_void_p(c1.ptr)); capi.call("agpu_checksum", h, C.c_void_p(O.ptr + doff), nbytes // 8 * 8, C.c_void_p(c2.ptr))
    a = dev.retrive_data(c1, 8, pipeline=p).view(np.uint64)[0]; b = dev.retrive_data(c2, 8, pipeline=p).view(np.uint64)[0]
    ms = float(np.median(ts))
    print(nbytes, soff, doff, "ms", round(ms, 4), "TB/s", round(2 * nbytes / ms / 1e9, 3), "same:", a == b)
