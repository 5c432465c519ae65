#!/usr/bin/env python3
"""DEV TOOL (GPU box): the kernels with an occupancy cap (tuning wave_lds, common.hpp wave_lds_for) at 1e9 rows: product default against
no cap (wave_lds = -1), alternating three times in one process; adaptive tiles off and on.   python tools/probe/caps_ab.py"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "caps")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
u8, u16, f, g = dev.create_table_buffers([n, 2 * n, 4 * n, 4 * n])
capi.call("agpu_synth_u8", h, vp(u8), n, 6, 0)
capi.call("agpu_synth_u8", h, vp(u16), 2 * n, 7, 0)
capi.call("agpu_synth_f32", h, vp(f), n, 1, 0, C.c_float(0.001), C.c_float(1000.0))
p.sync()


class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]


st = (Step * 1)()
st[0].op, st[0].kind, st[0].operand = capi.UN_SIN, 0, None
K = {
    "sin_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(g), n)),
    "cos_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.F32, vp(f), vp(g), n)),
    "cast_u8_f32": (5.0, lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(g), n)),
    "cast_i8_i32": (5.0, lambda: capi.call("agpu_cast", h, capi.I8, capi.I32, vp(u8), vp(g), n)),
    "cast_u16_f32": (6.0, lambda: capi.call("agpu_cast", h, capi.U16, capi.F32, vp(u16), vp(g), n)),
    "cast_i16_f32": (6.0, lambda: capi.call("agpu_cast", h, capi.I16, capi.F32, vp(u16), vp(g), n)),
    "sin_u8": (5.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(u8), vp(g), n)),
    "cos_u8": (5.0, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.U8, vp(u8), vp(g), n)),
    "sinh_i8": (5.0, lambda: capi.call("agpu_unary", h, capi.UN_SINH, capi.I8, vp(u8), vp(g), n)),
    "cast_u8_then_sin_one_launch": (5.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u8), C.cast(st, C.c_void_p), 1, vp(g), n)),
}


def med(fn, reps=9):
    for _ in range(9):
        fn()
    p.sync()
    fn()
    ts = []
    for _ in range(reps):
        q.begin(p); fn(); q.end(p)
        ts.append(q.wait_for_results())
    return sorted(ts)[len(ts) // 2]


out = {}
for name, (bpr, fn) in K.items():
    row = {}
    for rnd in range(3):
        for label, lds, auto in (("no_cap", -1, 1), ("default", 0, 1), ("default_adaptive_tiles", 0, 0)):
            p.set_tuning("wave_lds", lds)
            p.set_tuning("tile_auto", auto)
            row.setdefault(label, []).append(round(bpr * n / med(fn) / 1e6 / 8000.0, 4))
    out[name] = row
    print(name, row, file=sys.stderr)
out["tile_auto"] = dev.tile_auto_info()
print(out["tile_auto"], file=sys.stderr)
print(json.dumps(out, indent=1))
