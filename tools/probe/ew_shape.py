#!/usr/bin/env python3
"""DEV TOOL: the plain streaming kernels (ew_kernel) against the block size (build-time AGPU_EW_DEFAULT_BLK, AGPU_LIB
selects the build): f32 add (12 B/row), add_scalar / neg (8), u8 add (3), i32 sum is not touched."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "es")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
A, B, O = dev.create_table_buffers([4 * n] * 3)
S = dev.create_gpu_buffer_with_data(np.array([3.0], np.float32))
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-1000), C.c_float(1000))
capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-1000), C.c_float(1000))
p.sync()
cases = (("add", 12, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(A), vp(B), vp(O), n)),
         ("add_scalar", 8, lambda: capi.call("agpu_scalar", h, capi.OP_ADD, capi.F32, vp(A), vp(S), vp(O), n)),
         ("neg", 8, lambda: capi.call("agpu_unary", h, capi.UN_NEG, capi.F32, vp(A), vp(O), n)),
         ("exp", 8, lambda: capi.call("agpu_unary", h, capi.UN_EXP, capi.F32, vp(A), vp(O), n)),
         ("u8 add", 3, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.U8, vp(A), vp(B), vp(O), n)),
         ("u16 add", 6, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.U16, vp(A), vp(B), vp(O), n)))
out = []
for name, bpr, f in cases:
    for _ in range(6):
        f()
    p.sync()
    ts = []
    for _ in range(11):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    ms = float(np.median(ts))
    out.append(f"{name} {ms:.4f} {bpr * n / ms / 8e9:.3f}")
print("   ".join(out), flush=True)
