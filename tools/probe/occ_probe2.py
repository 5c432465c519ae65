#!/usr/bin/env python3
"""DEV TOOL (GPU box): occupancy caps (AGPU_EXP_LDS = bytes of unused dynamic LDS per wave) on the kernels not yet measured: reductions,
sub-word shifts, the sub-word compare.  One process per setting.    AGPU_EXP_LDS=6800 python tools/probe/occ_probe2.py"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "occ2")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
A, B, O = dev.create_table_buffers([4 * n] * 3)
R = dev.create_empty_buffer(64)
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-1.0), C.c_float(1.0))
capi.call("agpu_synth_i32", h, vp(B), n, 2, 0, 16)
p.sync()
F32 = capi.F32
K = {
    "sum_f32_tree": (4.0, lambda: capi.call("agpu_reduce", h, capi.RED_SUM, F32, vp(A), None, n, vp(R))),
    "min_f32": (4.0, lambda: capi.call("agpu_reduce", h, capi.RED_MIN, F32, vp(A), None, n, vp(R))),
    "max_f32": (4.0, lambda: capi.call("agpu_reduce", h, capi.RED_MAX, F32, vp(A), None, n, vp(R))),
    "sum_i32": (4.0, lambda: capi.call("agpu_reduce", h, capi.RED_SUM, capi.I32, vp(A), None, n, vp(R))),
    "sum_f64": (4.0, lambda: capi.call("agpu_reduce_sum_f64", h, vp(A), None, n, vp(R))),
    "u16_shr": (8.0, lambda: capi.call("agpu_binary", h, capi.OP_SHR, capi.U16, vp(A), vp(B), vp(O), n)),
    "u8_shl": (6.0, lambda: capi.call("agpu_binary", h, capi.OP_SHL, capi.U8, vp(A), vp(B), vp(O), n)),
    "u8_eq": (2.125, lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.U8, vp(A), vp(B), vp(O), n)),
    "u16_lt": (4.125, lambda: capi.call("agpu_compare", h, capi.CMP_LT, capi.U16, vp(A), vp(B), vp(O), n)),
}


def med(fn, reps=9):
    for _ in range(4):
        fn()
    p.sync()
    ts = []
    for _ in range(reps):
        q.begin(p); fn(); q.end(p)
        ts.append(q.wait_for_results())
    return sorted(ts)[len(ts) // 2]


out = {"exp_lds": os.environ.get("AGPU_EXP_LDS", "0")}
for rnd in range(2):
    for name, (bpr, fn) in K.items():
        out.setdefault(name, []).append(round(bpr * n / med(fn) / 1e6 / 8000.0, 4))
print(json.dumps(out))
