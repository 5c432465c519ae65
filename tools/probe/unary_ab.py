#!/usr/bin/env python3
"""A/B helper: time one f32 unary op (and add as the memory-bound reference) at 1e9 rows, 20 launches, in this process.
Usage: [AGPU_LIB=variant.so] python tools/probe/unary_ab.py log|sin|..."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

op = {"log": capi.UN_LOG, "sin": capi.UN_SIN, "cos": capi.UN_COS, "sinh": capi.UN_SINH, "exp": capi.UN_EXP}[sys.argv[1]]
n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "ab")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
A, B, O = (dev.create_empty_buffer(4 * n) for _ in range(3))
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(0.001), C.c_float(1000))
capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-8), C.c_float(8))


def t(f):
    q.begin(p); f(); q.end(p)
    return q.wait_for_results()


un = lambda: capi.call("agpu_unary", h, op, capi.F32, vp(A), vp(O), n)  # noqa: E731
add = lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(A), vp(B), vp(O), n)  # noqa: E731
for _ in range(6):
    un()
ts = [t(un) for _ in range(20)]
ta = [t(add) for _ in range(10)]
print(sys.argv[1], "ms min/median/max:", round(min(ts), 4), round(float(np.median(ts)), 4), round(max(ts), 4), "| add median", round(float(np.median(ta)), 4))
