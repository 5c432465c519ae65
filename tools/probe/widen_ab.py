#!/usr/bin/env python3
"""A/B helper: the widening family (cast u8→f32 / u8→u32 / i16→f32 / u8→u16, fused sin_u8 / cos_i8 / sin_u16) and the
memory-bound references (f32 add, broadcast) at 1e9 rows, 12 launches each, in this process; then a size sweep of f32 add
and u8 add (fixed per-launch cost vs per-byte rate).  Usage: [AGPU_LIB=variant.so] python tools/probe/widen_ab.py [--sizes]"""
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
p = ArrowComputePipeline(dev, "ab")
q = CmpQuery(dev)
h = p._handle
vp = lambda b, off=0: C.c_void_p(b.ptr + off)  # noqa: E731
A, B, O = (dev.create_empty_buffer(4 * n) for _ in range(3))
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-8), C.c_float(8))
capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-8), C.c_float(8))
U8 = dev.create_empty_buffer(2 * n)
capi.call("agpu_synth_u8", h, vp(U8), 2 * n, 3, 0)
p.sync()


def t(f, reps=12):
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts)), float(min(ts))


rows = []


def row(label, bytes_per_row, f, rows_n=n):
    med, mn = t(f)
    r = {"kernel": label, "ms_median": round(med, 4), "ms_min": round(mn, 4), "frac_8TBs": round(bytes_per_row * rows_n / med / 1e6 / 8000, 4)}
    rows.append(r)
    print(json.dumps(r), flush=True)


row("f32 add", 12, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(A), vp(B), vp(O), n))
row("broadcast f32", 4, lambda: capi.call("agpu_broadcast", h, capi.F32, 0x3F800000, vp(O), n))
row("cast u8->f32", 5, lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(U8), vp(O), n))
row("cast u8->u32", 5, lambda: capi.call("agpu_cast", h, capi.U8, capi.U32, vp(U8), vp(O), n))
row("cast i8->i32", 5, lambda: capi.call("agpu_cast", h, capi.I8, capi.I32, vp(U8), vp(O), n))
row("cast i16->f32", 6, lambda: capi.call("agpu_cast", h, capi.I16, capi.F32, vp(U8), vp(O), n))
row("cast u8->u16", 3, lambda: capi.call("agpu_cast", h, capi.U8, capi.U16, vp(U8), vp(O), n))
row("sin_u8", 5, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(U8), vp(O), n))
row("cos_i8", 5, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.I8, vp(U8), vp(O), n))
row("sin_u16", 6, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U16, vp(U8), vp(O), n))
row("cos_i16", 6, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.I16, vp(U8), vp(O), n))
row("u8 add", 3, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.U8, vp(U8), vp(U8, n), vp(O), n))
row("cast f32->u8", 5, lambda: capi.call("agpu_cast", h, capi.F32, capi.U8, vp(A), vp(O), n))
row("f32 add", 12, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(A), vp(B), vp(O), n))
if "--sizes" in sys.argv:
    for m in (1 << 22, 1 << 24, 1 << 26, 1 << 27, 1 << 28, 1 << 29, 1_000_000_000):
        row(f"f32 add n={m}", 12, lambda m=m: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(A), vp(B), vp(O), m), m)
    for m in (1 << 24, 1 << 26, 1 << 28, 1_000_000_000, 2_000_000_000):
        k = min(m, n)  # two u8 operands inside the 2e9-byte buffer
        row(f"u8 add n={k}", 3, lambda k=k: capi.call("agpu_binary", h, capi.OP_ADD, capi.U8, vp(U8), vp(U8, n), vp(O), k), k)
    big = dev.create_empty_buffer(4 * n)
    capi.call("agpu_synth_u8", h, vp(big), 4 * n, 5, 0)
    row("u8 add n=2e9 (2 GB per operand)", 3, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.U8, vp(big), vp(big, 2 * n), vp(A), 2 * n), 2 * n)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tag = os.path.basename(os.environ.get("AGPU_LIB", "product")).replace(".so", "")
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", f"widen_ab_{tag}.json"), "w"), indent=1)
