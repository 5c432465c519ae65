#!/usr/bin/env python3
"""DEV TOOL (GPU box): tiles per block with the next tile's loads issued before the current one is evaluated —
tuning heavy_tiles for sin / cos / sinh / log f32 (ew_prefetch_kernel), cast_tiles for the widening casts and the cast-headed
chains (cvt_wide_kernel / cast_chain_kernel), table_tiles for the LDS-table kernels (lut8 / trig16).  1e9 rows, median of 7
HIP-event timings, alternating K so that drift shows.  PREFETCH_KS=0,1,64,256,… sweeps other tile counts — large ones make the
kernels PERSISTENT (grid = tiles / K: one moving front instead of K fronts a grid apart; round 5).     python tools/probe/prefetch_sweep.py [rows] > gpurun_out/r04_prefetch_sweep.json"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "sweep")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
u8, u16, f, g, f2 = dev.create_table_buffers([n, 2 * n, 4 * n, 4 * n, 4 * n])
capi.call("agpu_synth_u8", h, vp(u8), n, 6, 0)
capi.call("agpu_synth_u8", h, vp(u16), 2 * n, 7, 0)
capi.call("agpu_synth_f32", h, vp(f), n, 1, 0, C.c_float(0.001), C.c_float(1000.0))
capi.call("agpu_synth_f32", h, vp(f2), n, 2, 0, C.c_float(-3.0), C.c_float(3.0))
sc = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
p.sync()


class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]


def chain(*items):
    arr = (Step * len(items))()
    for k, (op, kind, operand) in enumerate(items):
        arr[k].op, arr[k].kind, arr[k].operand = op, kind, (operand.ptr if operand is not None else None)
    return arr, len(items)


st_so, n_so = chain((capi.OP_MUL, 1, sc), (capi.OP_ADD, 1, sc))
st_sin, n_sin = chain((capi.UN_SIN, 0, None))
st_hv, n_hv = chain((capi.OP_MUL, 1, sc), (capi.UN_SIN, 0, None))
KERNELS = {
    "heavy_tiles": {
        "sin_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(g), n)),
        "cos_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.F32, vp(f), vp(g), n)),
        "sinh_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_SINH, capi.F32, vp(f), vp(g), n)),
        "log_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_LOG, capi.F32, vp(f), vp(g), n)),
    },
    "cast_tiles": {
        "cast_u8_f32": (5.0, lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(g), n)),
        "cast_u16_f32": (6.0, lambda: capi.call("agpu_cast", h, capi.U16, capi.F32, vp(u16), vp(g), n)),
        "cast_u8_scale_offset": (5.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u8), C.cast(st_so, C.c_void_p), n_so, vp(g), n)),
        "cast_u16_then_sin": (6.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U16, vp(u16), C.cast(st_sin, C.c_void_p), n_sin, vp(g), n)),
        "cast_u8_scale_then_sin": (5.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u8), C.cast(st_hv, C.c_void_p), n_hv, vp(g), n)),
    },
    "table_tiles": {
        "sin_u8": (5.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(u8), vp(g), n)),
        "sin_u16": (6.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U16, vp(u16), vp(g), n)),
        "log_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_LOG, capi.F32, vp(f), vp(g), n)),
        "pow_f32": (12.0, lambda: capi.call("agpu_binary", h, capi.OP_POW, capi.F32, vp(f), vp(f2), vp(g), n)),
        "pow_f32_scalar": (8.0, lambda: capi.call("agpu_scalar", h, capi.OP_POW, capi.F32, vp(f), vp(sc), vp(g), n)),
    },
}
KS = [int(x) for x in os.environ.get("PREFETCH_KS", "0,1,2,3,4,6,8,0,1,2,4").split(",")]


def median_ms(fn, reps=7):
    fn(), fn()
    ts = []
    for _ in range(reps):
        q.begin(p)
        fn()
        q.end(p)
        ts.append(q.wait_for_results())
    return sorted(ts)[len(ts) // 2]


out = {}
for key, kernels in KERNELS.items():
    for name, (bpr, fn) in kernels.items():
        row = []
        for k in KS:
            p.set_tuning(key, k)
            ms = median_ms(fn)
            row.append({"k": k, "ms": round(ms, 4), "frac": round(bpr * n / ms / 1e6 / 8000.0, 4)})
        p.set_tuning(key, 0)
        out[name] = {"tuning": key, "sweep": row}
        print(name, " ".join(f"{r['k']}:{r['frac']:.3f}" for r in row), file=sys.stderr)
print(json.dumps(out))
