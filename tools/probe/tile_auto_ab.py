#!/usr/bin/env python3
"""DEV TOOL (GPU box): what the adaptive tiles-per-block policy (tuning tile_auto) decides in THIS process and what that is worth: every
kernel family it covers at 1e9 rows with tile_auto = 1 (static: one tile), forced two tiles, and tile_auto = 0 (adaptive, after its warm-up),
alternating twice; median of 9 HIP-event timings.    python tools/probe/tile_auto_ab.py > gpurun_out/r05_tile_auto_ab.json"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "ab")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
u8, u16, f, g = dev.create_table_buffers([n, 2 * n, 4 * n, 4 * n])
capi.call("agpu_synth_u8", h, vp(u8), n, 6, 0)
capi.call("agpu_synth_u8", h, vp(u16), 2 * n, 7, 0)
capi.call("agpu_synth_f32", h, vp(f), n, 1, 0, C.c_float(0.001), C.c_float(1000.0))
p.sync()
K = {
    "sin_f32": ("heavy_tiles", 8.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(g), n)),
    "cos_f32": ("heavy_tiles", 8.0, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.F32, vp(f), vp(g), n)),
    "sinh_f32": ("heavy_tiles", 8.0, lambda: capi.call("agpu_unary", h, capi.UN_SINH, capi.F32, vp(f), vp(g), n)),
    "log_f32": ("table_tiles", 8.0, lambda: capi.call("agpu_unary", h, capi.UN_LOG, capi.F32, vp(f), vp(g), n)),
    "cast_u8_f32": ("cast_tiles", 5.0, lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(g), n)),
    "cast_u16_f32": ("cast_tiles", 6.0, lambda: capi.call("agpu_cast", h, capi.U16, capi.F32, vp(u16), vp(g), n)),
    "sin_u8": ("table_tiles", 5.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(u8), vp(g), n)),
    "cos_u8": ("table_tiles", 5.0, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.U8, vp(u8), vp(g), n)),
    "sin_u16": ("table_tiles", 6.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U16, vp(u16), vp(g), n)),
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
for name, (key, bpr, fn) in K.items():
    row = {}
    for rnd in range(2):
        for label, auto, k in (("one_tile", 1, 0), ("two_tiles", 1, 2), ("adaptive", 0, 0)):
            p.set_tuning("tile_auto", auto)
            p.set_tuning(key, k)
            row.setdefault(label, []).append(round(bpr * n / med(fn) / 1e6 / 8000.0, 4))
    p.set_tuning("tile_auto", 0)
    p.set_tuning(key, 0)
    out[name] = row
    print(name, row, file=sys.stderr)
out["decisions"] = dev.tile_auto_info()
print(out["decisions"], file=sys.stderr)
print(json.dumps(out, indent=1))
