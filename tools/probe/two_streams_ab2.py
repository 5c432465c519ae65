#!/usr/bin/env python3
"""DEV TOOL (round 6b): the width-changing and chain kernels with the tile order cut into TWO lock-step streams (AGPU_EXP_TWO read at every
launch) against the sequential order, alternating in one process on the same table-placed buffers; plus the unary kernel at smaller sizes."""
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
S = dev.create_gpu_buffer_with_data(np.array([1.5], np.float32))
S2 = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-100), C.c_float(100))
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
c_so, n_so = chain((capi.OP_MUL, 1, S2), (capi.OP_ADD, 1, S))
c_hv, n_hv = chain((capi.OP_MUL, 1, S2), (capi.UN_SIN, 0, None))
ops = {
    "cast u8→f32": (5, lambda: capi.call("agpu_cast", h, U8, F32, vp(B), vp(O), n)),
    "cast i16→f32": (6, lambda: capi.call("agpu_cast", h, I16, F32, vp(B), vp(O), n)),
    "cast u8→u16": (3, lambda: capi.call("agpu_cast", h, U8, U16, vp(B), vp(O), n)),
    "cast f32→u8": (5, lambda: capi.call("agpu_cast", h, F32, U8, vp(A), vp(O), n)),
    "cast f32→i16": (6, lambda: capi.call("agpu_cast", h, F32, I16, vp(A), vp(O), n)),
    "sin_u8": (5, lambda: capi.call("agpu_unary", h, capi.UN_SIN, U8, vp(B), vp(O), n)),
    "sin_u16": (6, lambda: capi.call("agpu_unary", h, capi.UN_SIN, U16, vp(B), vp(O), n)),
    "cast u8 → sin one launch": (5, lambda: capi.call("agpu_fused_cast_chain", h, U8, vp(B), C.cast(c_sin, C.c_void_p), n_sin, vp(O), n)),
    "cast u8 ·s+s one launch": (5, lambda: capi.call("agpu_fused_cast_chain", h, U8, vp(B), C.cast(c_so, C.c_void_p), n_so, vp(O), n)),
    "cast i16 → sin one launch": (6, lambda: capi.call("agpu_fused_cast_chain", h, I16, vp(B), C.cast(c_sin, C.c_void_p), n_sin, vp(O), n)),
    "cast u16 ·s+s one launch": (6, lambda: capi.call("agpu_fused_cast_chain", h, U16, vp(B), C.cast(c_so, C.c_void_p), n_so, vp(O), n)),
    "(a+s)·t chain": (8, lambda: capi.call("agpu_fused_chain", h, F32, vp(A), C.cast(c_so, C.c_void_p), n_so, vp(O), n)),
    "(x·s).sin() chain": (8, lambda: capi.call("agpu_fused_chain", h, F32, vp(A), C.cast(c_hv, C.c_void_p), n_hv, vp(O), n)),
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
    for rnd in range(3):
        for mode in ("seq", "two"):
            if mode == "two":
                os.environ["AGPU_EXP_TWO"] = "1"
            else:
                os.environ.pop("AGPU_EXP_TWO", None)
            ms = med(f)
            r.setdefault(mode, []).append(round(bpr * n / ms / 1e6 / 8000, 4))
    os.environ.pop("AGPU_EXP_TWO", None)
    rows.append(r)
    print(json.dumps(r), flush=True)
# the plain unary kernel (two streams from 131 072 tiles = 128 MiB on) across sizes: GB/s, not a fraction — small columns live in the caches
for rows_ in (1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 1 << 29):
    r = {"kernel": "f32 neg", "rows": rows_}
    for rnd in range(3):
        for mode, grid in (("seq", 1), ("two", 0)):
            # stream_grid = a huge explicit grid keeps one tile per block but switches the two-stream order off (launch_ew)
            p.set_tuning("stream_grid", (rows_ // 256) if grid else 0)
            ms = med(lambda: capi.call("agpu_unary", h, capi.UN_NEG, F32, vp(A), vp(O), rows_))
            r.setdefault(mode, []).append(round(8 * rows_ / ms / 1e6, 1))
    p.set_tuning("stream_grid", 0)
    rows.append(r)
    print(json.dumps(r), flush=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r06b_two_streams_ab2.json"), "w"), indent=1)
