#!/usr/bin/env python3
"""DEV TOOL (round 6b): the same-width streaming kernels with the block order cut into TWO lock-step streams half a column apart
(AGPU_EXP_TWO_STREAMS read at every launch) against the sequential order, alternating in one process on the same table-placed buffers."""
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
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-1000), C.c_float(1000))
capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-1000), C.c_float(1000))
p.sync()
ops = {
    "f32 add": (12, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(A), vp(B), vp(O), n)),
    "f32 mul": (12, lambda: capi.call("agpu_binary", h, capi.OP_MUL, capi.F32, vp(A), vp(B), vp(O), n)),
    "i32 add": (12, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.I32, vp(A), vp(B), vp(O), n)),
    "u8 add": (3, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.U8, vp(A), vp(B), vp(O), n)),
    "f32 add_scalar": (8, lambda: capi.call("agpu_scalar", h, capi.OP_ADD, capi.F32, vp(A), vp(S), vp(O), n)),
    "f32 neg": (8, lambda: capi.call("agpu_unary", h, capi.UN_NEG, capi.F32, vp(A), vp(O), n)),
    "f32 sqrt": (8, lambda: capi.call("agpu_unary", h, capi.UN_SQRT, capi.F32, vp(A), vp(O), n)),
    "f32 exp": (8, lambda: capi.call("agpu_unary", h, capi.UN_EXP, capi.F32, vp(A), vp(O), n)),
    "f32 sin": (8, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(A), vp(O), n)),
    "f32 neg in place": (8, lambda: capi.call("agpu_unary", h, capi.UN_NEG, capi.F32, vp(A), vp(A), n)),
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
                os.environ["AGPU_EXP_TWO_STREAMS"] = "1"
            else:
                os.environ.pop("AGPU_EXP_TWO_STREAMS", None)
            ms = med(f)
            r.setdefault(mode, []).append(round(bpr * n / ms / 1e6 / 8000, 4))
    os.environ.pop("AGPU_EXP_TWO_STREAMS", None)
    rows.append(r)
    print(json.dumps(r), flush=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r06b_two_streams_ab.json"), "w"), indent=1)
