#!/usr/bin/env python3
"""DEV TOOL (GPU box): launch the config-4 kernels a few times at N rows so that a profiler pass sees them —
cast u8→f32 (cvt_wide_kernel), sin_u8 / cos_u8 (lut8_kernel), sin_u16 (trig16_kernel), sin_f32 / cos_f32 (ew_kernel),
cast→sin in one launch (agpu_fused_cast_chain → lut8_kernel) and cast→scale→offset (cast_chain_kernel).
    python tools/probe/narrow_run.py [rows] [iters]      prints median HIP-event ms per kernel as JSON"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "narrow")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
u8, u16, f, g = dev.create_table_buffers([n, 2 * n, 4 * n, 4 * n])
capi.call("agpu_synth_u8", h, vp(u8), n, 6, 0)
capi.call("agpu_synth_u8", h, vp(u16), 2 * n, 7, 0)
capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(f), n)
sc = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
want = os.environ.get("NARROW_ONLY", "")
f2 = u8b = None
if "add_f32" in want or "cast_f32_u8" in want:  # reference rows for the memory-side counters (tools/probe/pmc_memside.sh)
    f2, u8b = dev.create_table_buffers([4 * n, n])
    capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(f2), n)
p.sync()


class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]


def chain(*items):
    arr = (Step * len(items))()
    for k, (op, kind, operand) in enumerate(items):
        arr[k].op, arr[k].kind, arr[k].operand = op, kind, (operand.ptr if operand is not None else None)
    return arr, len(items)


st_sin, n_sin = chain((capi.UN_SIN, 0, None))
st_so, n_so = chain((capi.OP_MUL, 1, sc), (capi.OP_ADD, 1, sc))
st_heavy, n_heavy = chain((capi.OP_MUL, 1, sc), (capi.UN_SIN, 0, None))
rows = {
    "cast_u8_f32": (5.0, lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(g), n)),
    "cast_u16_f32": (6.0, lambda: capi.call("agpu_cast", h, capi.U16, capi.F32, vp(u16), vp(g), n)),
    "sin_u8": (5.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(u8), vp(g), n)),
    "cos_u8": (5.0, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.U8, vp(u8), vp(g), n)),
    "sin_u16": (6.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U16, vp(u16), vp(g), n)),
    "sin_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(g), n)),
    "cos_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.F32, vp(f), vp(g), n)),
    "sinh_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_SINH, capi.F32, vp(f), vp(g), n)),
    "cast_u8_then_sin_one_launch": (5.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u8), C.cast(st_sin, C.c_void_p), n_sin, vp(g), n)),
    "cast_u8_scale_offset_one_launch": (5.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u8), C.cast(st_so, C.c_void_p), n_so, vp(g), n)),
    "cast_u16_then_sin_one_launch": (6.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U16, vp(u16), C.cast(st_sin, C.c_void_p), n_sin, vp(g), n)),
    "cast_u8_scale_then_sin_one_launch": (5.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u8), C.cast(st_heavy, C.c_void_p), n_heavy, vp(g), n)),
    "add_f32": (12.0, lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(f), vp(f2), vp(g), n)),
    "fill_f32": (4.0, lambda: capi.call("agpu_broadcast", h, capi.F32, 0x3F000000, vp(g), n)),
    "cast_f32_u8": (5.0, lambda: capi.call("agpu_cast", h, capi.F32, capi.U8, vp(f), vp(u8b), n)),
    "u8_eq": (2.125, lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.U8, vp(u8), vp(u16), vp(g), n)),
}
only = os.environ.get("NARROW_ONLY")
out = {}
for name, (bpr, fn) in rows.items():
    if only and name not in only.split(","):
        continue
    for _ in range(3):
        fn()
    p.sync()
    ts = []
    for _ in range(iters):
        q.begin(p)
        fn()
        q.end(p)
        ts.append(q.wait_for_results())
    ms = sorted(ts)[len(ts) // 2]
    out[name] = {"ms": round(ms, 4), "frac_hbm_peak": round(bpr * n / ms / 1e6 / 8000.0, 4)}
print(json.dumps(out))
