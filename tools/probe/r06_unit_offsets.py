#!/usr/bin/env python3
"""Round 6: does the 512 MiB-unit position of an arena block matter?  One 40 GiB block (the arena's stand-in), three 4e9-byte columns of an f32
add placed at unit offsets (ua, ub, uo) with the arena's colours 0 / 8 KiB / 4 KiB; fraction of the 8 TB/s roof per placement."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
capi.call("agpu_set_tuning", b"pool_arena", 0)
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "uo"); q = CmpQuery(dev); h = p._handle
U = 512 << 20
big = dev.create_empty_buffer(80 * U)
base = (big.ptr + U - 1) // U * U
def med(fn):
    for _ in range(2): fn()
    p.sync(); ts = []
    for _ in range(7):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return 12.0 * n / sorted(ts)[3] / 1e6 / 8000
col = (0, 8192, 4096)
cases = [(0, 8, 16), (0, 9, 18), (0, 10, 20), (0, 11, 22), (0, 12, 24), (0, 14, 28), (0, 16, 32), (3, 11, 19), (3, 12, 22), (1, 9, 17), (5, 14, 23), (0, 8, 17), (0, 8, 18), (0, 8, 20), (0, 24, 48), (0, 32, 64), (0, 8, 16)]
for ua, ub, uo in cases:
    pa, pb, po = (base + u * U + c for u, c in zip((ua, ub, uo), col))
    capi.call("agpu_synth_f32", h, C.c_void_p(pa), n, 1, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_synth_f32", h, C.c_void_p(pb), n, 2, 0, C.c_float(-1000.0), C.c_float(1000.0))
    f = med(lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, C.c_void_p(pa), C.c_void_p(pb), C.c_void_p(po), n))
    print(f"units {ua:2d} {ub:2d} {uo:2d}  (strides {ub-ua:2d} {uo-ub:2d})  add {f:.4f}", flush=True)
