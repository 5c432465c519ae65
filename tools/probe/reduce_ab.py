#!/usr/bin/env python3
"""DEV TOOL: the whole-column reductions at 1e9 rows, a dozen calls each — run under `rocprofv3 --kernel-trace` and fed to
reduce_ab_join.py for the per-launch durations and the gaps between the launches of one call."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "r", fuse=False)
h = p._handle
A, = dev.create_table_buffers([4 * n])
R = dev.create_empty_buffer(64)
capi.call("agpu_synth_f32", h, C.c_void_p(A.ptr), n, 1, 0, C.c_float(-1), C.c_float(1))
p.sync()
for op, dt in ((capi.RED_SUM, capi.F32), (capi.RED_MIN, capi.F32), (capi.RED_MAX, capi.F32), (capi.RED_SUM, capi.I32), (capi.RED_MIN, capi.I32)):
    for _ in range(12):
        capi.call("agpu_reduce", h, op, dt, C.c_void_p(A.ptr), None, n, C.c_void_p(R.ptr))
    p.sync()
for _ in range(12):
    capi.call("agpu_reduce_sum_f64", h, C.c_void_p(A.ptr), None, n, C.c_void_p(R.ptr))
p.sync()
