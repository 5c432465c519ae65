#!/usr/bin/env python3
"""DEV TOOL (round 6b): two lock-step streams (the default) against the sequential order — tuning stream_grid = 2^30 keeps one tile per
block and switches the two-stream order off (common.hpp two_streams_half) — for the VALU-heavy kernels, alternating in one process."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "s", fuse=False)
h = p._handle
q = CmpQuery(dev)
A, B, O = dev.create_table_buffers([4 * n] * 3)
S2 = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-30), C.c_float(30))
capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-1000), C.c_float(1000))
p.sync()
F32, U8, I16, U16 = capi.F32, capi.U8, capi.I16, capi.U16


class _Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]


def chain(*items):
    arr = (_Step * len(items))()
    for k_, (op_, kind_, operand_) in enumerate(items):
        arr[k_].op, arr[k_].kind, arr[k_].operand = op_, kind_, (operand_.ptr if operand_ is not None else None)
    return arr, len(items)


c_sin, n_sin = chain((capi.UN_SIN, 0, None))
c_hv, n_hv = chain((capi.OP_MUL, 1, S2), (capi.UN_SIN, 0, None))
ops = {
    "f32 sin": (8, lambda: capi.call("agpu_unary", h, capi.UN_SIN, F32, vp(A), vp(O), n)),
    "f32 cos": (8, lambda: capi.call("agpu_unary", h, capi.UN_COS, F32, vp(A), vp(O), n)),
    "f32 sinh": (8, lambda: capi.call("agpu_unary", h, capi.UN_SINH, F32, vp(A), vp(O), n)),
    "f32 log": (8, lambda: capi.call("agpu_unary", h, capi.UN_LOG, F32, vp(A), vp(O), n)),
    "f32 exp": (8, lambda: capi.call("agpu_unary", h, capi.UN_EXP, F32, vp(A), vp(O), n)),
    "sin_u16": (6, lambda: capi.call("agpu_unary", h, capi.UN_SIN, U16, vp(B), vp(O), n)),
    "cos_i16": (6, lambda: capi.call("agpu_unary", h, capi.UN_COS, I16, vp(B), vp(O), n)),
    "cast i16 → sin one launch": (6, lambda: capi.call("agpu_fused_cast_chain", h, I16, vp(B), C.cast(c_sin, C.c_void_p), n_sin, vp(O), n)),
    "(x·s).sin() chain": (8, lambda: capi.call("agpu_fused_chain", h, F32, vp(A), C.cast(c_hv, C.c_void_p), n_hv, vp(O), n)),
    "clone_buffer 4 GB": (8, lambda: capi.call("agpu_copy", h, vp(O), vp(A), 4 * n)),
}


def med(f, reps=9):
    for _ in range(3):
        f()
    p.sync()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


rows = []
for name, (bpr, f) in ops.items():
    r = {"kernel": name}
    for rnd in range(4):
        for mode in ("seq", "two"):
            p.set_tuning("stream_grid", (1 << 30) if mode == "seq" else 0)
            ms = med(f)
            r.setdefault(mode, []).append(round(bpr * n / ms / 1e6 / 8000, 4))
    p.set_tuning("stream_grid", 0)
    rows.append(r)
    print(json.dumps(r), flush=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r06b_two_streams_ab3.json"), "w"), indent=1)
