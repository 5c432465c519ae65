#!/usr/bin/env python3
"""DEV TOOL: fused chains by number of input streams, with and without a terminal compare (2^28 rows)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1 << 28
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "cp")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
bufs = [dev.create_empty_buffer(4 * n + 4096 * i) for i in range(6)]
for i, b in enumerate(bufs[:5]):
    capi.call("agpu_synth_f32", h, vp(b), n, i + 1, 0, C.c_float(-1), C.c_float(1))
out = bufs[5]
ob = dev.create_empty_buffer(n // 8)


class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]


def steps(k):
    s = (Step * max(k, 1))()
    for i in range(k):
        s[i].op, s[i].kind, s[i].operand = (capi.OP_MUL if i % 2 == 0 else capi.OP_ADD), 2, bufs[1 + i].ptr
    return s


def timeit(label, f, nbytes):
    f(); p.sync()
    ts = []
    for _ in range(7):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    ms = float(np.median(ts))
    print(f"{label:<46} {ms:.4f} ms  {nbytes / ms / 1e9:.2f} TB/s", flush=True)


for k in (0, 1, 2, 3):
    st = steps(k)
    if k:
        timeit(f"chain, {k} array steps ({k + 1} in, 1 out)", lambda: capi.call(
            "agpu_fused_chain", h, capi.F32, vp(bufs[0]), C.cast(st, C.c_void_p), k, vp(out), n), (4 * (k + 2)) * n)
    timeit(f"chain+cmp, {k} array steps ({k + 2} in, bits out)", lambda: capi.call(
        "agpu_fused_chain_compare", h, capi.F32, vp(bufs[0]), C.cast(st, C.c_void_p), k, capi.CMP_GT, 2, vp(bufs[4]), vp(ob), n),
        (4 * (k + 2) + 0.125) * n)
timeit("agpu_compare (reference point, 2 in, bits out)", lambda: capi.call(
    "agpu_compare", h, capi.CMP_GT, capi.F32, vp(bufs[0]), vp(bufs[4]), vp(ob), n), 8.125 * n)
