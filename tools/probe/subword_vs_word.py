#!/usr/bin/env python3
"""Same bytes, same buffers, same process: u8 / u16 element-wise kernels against the i32 kernel over the identical byte
ranges (n/4 i32 rows) — separates the memory system from the sub-word VALU code.  Also sizes 1 GB and 4 GB per operand."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "ab")
q = CmpQuery(dev)
h = p._handle
vp = lambda b, off=0: C.c_void_p(b.ptr + off)  # noqa: E731
G = 1 << 30
A, B, O = (dev.create_empty_buffer(4 * G) for _ in range(3))
capi.call("agpu_synth_u8", h, vp(A), 4 * G, 1, 0)
capi.call("agpu_synth_u8", h, vp(B), 4 * G, 2, 0)
p.sync()


def t(f, reps=12):
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


for nbytes in (G, 4 * G):
    for label, dt, w in (("i32 add", capi.I32, 4), ("u8 add", capi.U8, 1), ("u16 add", capi.U16, 2), ("i32 add", capi.I32, 4),
                         ("u8 and", capi.U8, 1), ("u8 min", capi.U8, 1)):
        op = {"add": capi.OP_ADD, "and": capi.OP_AND, "min": capi.OP_MIN}[label.split()[1]]
        ms = t(lambda: capi.call("agpu_binary", h, op, dt, vp(A), vp(B), vp(O), nbytes // w))
        print(json.dumps({"bytes_per_operand": nbytes, "kernel": label, "ms": round(ms, 4), "TBps": round(3 * nbytes / ms / 1e9, 3)}), flush=True)
