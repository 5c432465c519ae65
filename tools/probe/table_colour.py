#!/usr/bin/env python3
"""DEV TOOL: is the colour agpu_malloc_table gives the second column right on THIS box?  bench.py's two tables, the
second i32 column over-allocated so it can be shifted by k × 4 KiB; eq + validity at every shift (k = 0 is the
allocator's placement).  Boxes of the pool differ (0.81 vs 0.87 for the same binary) — this tells whether placement is why."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
nb = (n + 63) // 64 * 8
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "tc")
q = CmpQuery(dev)
h = p._handle
fa, fb, fo = dev.create_table_buffers([4 * n] * 3)
ia, ib, va, vb, ob, ov = dev.create_table_buffers([4 * n, 4 * n + (1 << 16)] + [nb] * 4)
capi.call("agpu_synth_i32", h, C.c_void_p(ia.ptr), n, 1, 0, 1024)
capi.call("agpu_synth_i32", h, C.c_void_p(ib.ptr), n + (1 << 14), 2, 0, 1024)
for b, s in ((va, 3), (vb, 4)):
    capi.call("agpu_synth_bits", h, C.c_void_p(b.ptr), n, s, 0, C.c_double(0.9))
p.sync()
print("ia %x ib %x va %x vb %x ob %x ov %x" % tuple(x.ptr & 0xffffffffff for x in (ia, ib, va, vb, ob, ov)))
out = []
for k in list(range(8)) + [0]:
    f = lambda: capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, C.c_void_p(ia.ptr), C.c_void_p(ib.ptr + k * 4096),  # noqa: E731
                          C.c_void_p(va.ptr), C.c_void_p(vb.ptr), C.c_void_p(ob.ptr), C.c_void_p(ov.ptr), n)
    for _ in range(3):
        f()
    ts = []
    for _ in range(9):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    out.append(f"k{k}:{8.5 * n / float(np.median(ts)) / 8e9:.3f}")
print("  ".join(out))
f = lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, C.c_void_p(fa.ptr), C.c_void_p(fb.ptr), C.c_void_p(fo.ptr), n)  # noqa: E731
for _ in range(3):
    f()
ts = []
for _ in range(9):
    q.begin(p); f(); q.end(p)
    ts.append(q.wait_for_results())
print(f"add {12 * n / float(np.median(ts)) / 8e9:.3f}")
# the same eq + validity at k = 0, each launch preceded by an (untimed) f32 add — bench.py's step order
g = lambda: capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, C.c_void_p(ia.ptr), C.c_void_p(ib.ptr), C.c_void_p(va.ptr),  # noqa: E731
                      C.c_void_p(vb.ptr), C.c_void_p(ob.ptr), C.c_void_p(ov.ptr), n)
for pre in ("add", "none", "add", "none"):
    ts = []
    for _ in range(12):
        if pre == "add":
            f()
        q.begin(p); g(); q.end(p)
        ts.append(q.wait_for_results())
    print(f"eq+v after {pre}: {8.5 * n / float(np.median(ts[3:])) / 8e9:.3f}")
