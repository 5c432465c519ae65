#!/usr/bin/env python3
"""Round-3 sweep of the launch tunings that exist for the narrow kernels, one process, same buffers, 1e9 rows:
u8 eq → bitmap (stream_unroll 1 / 2 / 4), fused sin_u8 and sin_u16 (table_tiles 1 / 2 / 4), u16 shr (stream_unroll)."""
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
p = ArrowComputePipeline(dev, "narrow")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
n = 1_000_000_000
A, B, O = dev.create_table_buffers([4 * n] * 3)
OB, = dev.create_table_buffers([(n + 63) // 64 * 8])
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-1000), C.c_float(1000))
capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-1000), C.c_float(1000))
p.sync()
rows = []


def t(label, bpr, f):
    for _ in range(4):
        f()
    p.sync()
    ts = []
    for _ in range(9):
        q.begin(p)
        f()
        q.end(p)
        ts.append(q.wait_for_results())
    ms = float(np.median(ts))
    r = {"kernel": label, "ms": round(ms, 4), "frac_8TBs": round(bpr * n / ms / 1e6 / 8000, 4)}
    rows.append(r)
    print(r, flush=True)


for rep in range(2):
    for u in (8, 2, 4):  # compare.hip: 8 = one pack per lane (the round-2 shape), 2 = default, 4
        p.set_tuning("stream_unroll", u)
        t(f"u8 eq -> bitmap, stream_unroll {u}", 2.125, lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.U8, vp(A), vp(B), vp(OB), n))
        t(f"u16 lt -> bitmap, stream_unroll {u}", 4.125, lambda: capi.call("agpu_compare", h, capi.CMP_LT, capi.U16, vp(A), vp(B), vp(OB), n))
    p.set_tuning("stream_unroll", 1)
    for k in (1, 2, 4):
        p.set_tuning("table_tiles", k)
        t(f"sin_u8, table_tiles {k}", 5, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(B), vp(O), n))
        t(f"sin_u16, table_tiles {k}", 6, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U16, vp(B), vp(O), n))
    p.set_tuning("table_tiles", 1)
    t("cast u8->f32", 5, lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(B), vp(O), n))
    t("cast u8->f32, input from A", 5, lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(A), vp(O), n))
    t("sin_u8, input from A", 5, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(A), vp(O), n))
    t("u16 shr", 8, lambda: capi.call("agpu_binary", h, capi.OP_SHR, capi.U16, vp(A), vp(B), vp(O), n))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r03_narrow_tunings.json"), "w"), indent=1)
