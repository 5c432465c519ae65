#!/usr/bin/env python3
"""DEV TOOL: i32 eq → bitmap with the two input columns fixed and the RESULT bitmap moved: which address bits of the
(small, 1/64 of the traffic) output stream decide 0.84 vs 0.89?"""
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
p = ArrowComputePipeline(dev, "oo")
q = CmpQuery(dev)
h = p._handle
ia, ib = dev.create_table_buffers([4 * n] * 2)
big = dev.create_empty_buffer(nb + (1 << 30))
capi.call("agpu_synth_i32", h, C.c_void_p(ia.ptr), n, 1, 0, 1024)
capi.call("agpu_synth_i32", h, C.c_void_p(ib.ptr), n, 2, 0, 1024)
p.sync()


def med(off, reps=7):
    f = lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, C.c_void_p(ia.ptr), C.c_void_p(ib.ptr), C.c_void_p(big.ptr + off), n)  # noqa: E731
    for _ in range(2):
        f()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return 8.125 * n / float(np.median(ts)) / 8e9


print(f"ia {ia.ptr:#x} ib {ib.ptr:#x} out base {big.ptr:#x}")
for bit in range(7, 30):
    print(f"offset 2^{bit:<2d}: {med(1 << bit):.3f}", end="   ")
    if bit % 4 == 2:
        print()
print()
print("multiples of 4 KiB:", " ".join(f"{med(k << 12):.3f}" for k in range(16)))
print("multiples of 2 MiB:", " ".join(f"{med(k << 21):.3f}" for k in range(16)))
print("offset 0 again:", f"{med(0):.3f}")
