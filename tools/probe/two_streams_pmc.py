#!/usr/bin/env python3
"""DEV TOOL (round 6b): f32 neg at 1e9 rows, 2 K launches alternating between the sequential order (tuning stream_grid = 2^30) and two lock-step streams —
run under `rocprofv3 --pmc <counters>` by two_streams_pmc.sh, which averages the counters per order."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice  # noqa: E402

n = 1_000_000_000
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "s", fuse=False)
h = p._handle
A, O = dev.create_table_buffers([4 * n] * 2)
capi.call("agpu_synth_f32", h, C.c_void_p(A.ptr), n, 1, 0, C.c_float(-1000), C.c_float(1000))
p.sync()
for k in range(2 * K):  # alternating, so that clock ramps and thermal drift hit both orders alike: even launches sequential, odd two streams
    p.set_tuning("stream_grid", (1 << 30) if k % 2 == 0 else 0)
    capi.call("agpu_unary", h, capi.UN_NEG, capi.F32, C.c_void_p(A.ptr), C.c_void_p(O.ptr), n)
    p.sync()
