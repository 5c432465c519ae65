#!/usr/bin/env python3
"""DEV TOOL (GPU box): the streaming kernels at 1e9 rows under an occupancy cap (AGPU_DYN_LDS = unused dynamic LDS per wave of the one-wave
blocks: 0 → 32 waves per CU, 5600 → 28, 6800 → 24, 8000 → 20, 10240 → 16).  One process per setting (the switch is read once).
    AGPU_DYN_LDS=6800 python tools/probe/occ_probe.py"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "occ")
p.set_tuning("tile_auto", 1)
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
A, B, O = dev.create_table_buffers([4 * n] * 3)
u8, u16 = dev.create_table_buffers([n, 2 * n])
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(0.001), C.c_float(1000.0))
capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-1000.0), C.c_float(1000.0))
capi.call("agpu_synth_u8", h, vp(u8), n, 6, 0)
capi.call("agpu_synth_u8", h, vp(u16), 2 * n, 7, 0)
import numpy as np  # noqa: E402
S = dev.create_gpu_buffer_with_data(np.array([3.0], np.float32))
p.sync()
F32 = capi.F32
K = {
    "add_f32": (12.0, lambda: capi.call("agpu_binary", h, capi.OP_ADD, F32, vp(A), vp(B), vp(O), n)),
    "mul_f32": (12.0, lambda: capi.call("agpu_binary", h, capi.OP_MUL, F32, vp(A), vp(B), vp(O), n)),
    "add_scalar_f32": (8.0, lambda: capi.call("agpu_scalar", h, capi.OP_ADD, F32, vp(A), vp(S), vp(O), n)),
    "neg_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_NEG, F32, vp(A), vp(O), n)),
    "sqrt_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_SQRT, F32, vp(A), vp(O), n)),
    "exp_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_EXP, F32, vp(B), vp(O), n)),
    "sin_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, F32, vp(A), vp(O), n)),
    "cos_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_COS, F32, vp(A), vp(O), n)),
    "sinh_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_SINH, F32, vp(B), vp(O), n)),
    "cast_u8_f32": (5.0, lambda: capi.call("agpu_cast", h, capi.U8, F32, vp(u8), vp(O), n)),
    "cast_u16_f32": (6.0, lambda: capi.call("agpu_cast", h, capi.U16, F32, vp(u16), vp(O), n)),
    "log_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_LOG, F32, vp(A), vp(O), n)),
    "sin_u8": (5.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(u8), vp(O), n)),
    "sin_u16": (6.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U16, vp(u16), vp(O), n)),
}
IA, IB, VA, VB, OB, OV = dev.create_table_buffers([4 * n] * 2 + [(n + 63) // 64 * 8] * 4)
capi.call("agpu_synth_i32", h, vp(IA), n, 1, 0, 1024)
capi.call("agpu_synth_i32", h, vp(IB), n, 2, 0, 1024)
capi.call("agpu_synth_bits", h, vp(VA), n, 3, 0, C.c_double(0.9))
capi.call("agpu_synth_bits", h, vp(VB), n, 4, 0, C.c_double(0.9))
p.sync()


class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]


def chain(*items):
    arr = (Step * len(items))()
    for k, (op, kind, operand) in enumerate(items):
        arr[k].op, arr[k].kind, arr[k].operand = op, kind, (operand.ptr if operand is not None else None)
    return arr, len(items)


st_am, n_am = chain((capi.OP_ADD, 1, S), (capi.OP_MUL, 1, S))
st_pred, n_pred = chain((capi.OP_MUL, 2, B), (capi.OP_ADD, 2, B))
st_sin, n_sin = chain((capi.UN_SIN, 0, None))
st_so, n_so = chain((capi.OP_MUL, 1, S), (capi.OP_ADD, 1, S))
K.update({
    "pow_f32": (12.0, lambda: capi.call("agpu_binary", h, capi.OP_POW, F32, vp(A), vp(B), vp(O), n)),
    "pow_f32_scalar": (8.0, lambda: capi.call("agpu_scalar", h, capi.OP_POW, F32, vp(A), vp(S), vp(O), n)),
    "cast_f32_u8": (5.0, lambda: capi.call("agpu_cast", h, F32, capi.U8, vp(A), vp(O), n)),
    "fused_add_mul_scalar": (8.0, lambda: capi.call("agpu_fused_chain", h, F32, vp(A), C.cast(st_am, C.c_void_p), n_am, vp(O), n)),
    "fused_cast_u16_sin": (6.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U16, vp(u16), C.cast(st_sin, C.c_void_p), n_sin, vp(O), n)),
    "fused_cast_u16_scale_offset": (6.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U16, vp(u16), C.cast(st_so, C.c_void_p), n_so, vp(O), n)),
    "fused_cast_u8_scale_offset": (5.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u8), C.cast(st_so, C.c_void_p), n_so, vp(O), n)),
    "eq_i32_validity": (8.5, lambda: capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, vp(IA), vp(IB), vp(VA), vp(VB), vp(OB), vp(OV), n)),
    "eq_i32": (8.125, lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(IA), vp(IB), vp(OB), n)),
})
TWO = {"sin_f32": "heavy_tiles", "cos_f32": "heavy_tiles", "sinh_f32": "heavy_tiles", "cast_u8_f32": "cast_tiles", "cast_u16_f32": "cast_tiles",
       "log_f32": "table_tiles", "sin_u8": "table_tiles", "sin_u16": "table_tiles"}


def med(fn, reps=9):
    for _ in range(4):
        fn()
    p.sync()
    ts = []
    for _ in range(reps):
        q.begin(p); fn(); q.end(p)
        ts.append(q.wait_for_results())
    return sorted(ts)[len(ts) // 2]


out = {"dyn_lds": os.environ.get("AGPU_DYN_LDS", "0")}
for rnd in range(2):
    for name, (bpr, fn) in K.items():
        out.setdefault(name, []).append(round(bpr * n / med(fn) / 1e6 / 8000.0, 4))
        if name in TWO:
            p.set_tuning(TWO[name], 2)
            out.setdefault(name + " x2", []).append(round(bpr * n / med(fn) / 1e6 / 8000.0, 4))
            p.set_tuning(TWO[name], 0)
print(json.dumps(out))
