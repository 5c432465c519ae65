#!/usr/bin/env python3
"""DEV TOOL (GPU box): tuning wave_lds (bytes of unused dynamic LDS per wave: the occupancy cap of common.hpp wave_lds_for) swept in ONE
process over the kernels that take it, 1e9 rows, median of 7.    python tools/probe/caps_sweep.py > gpurun_out/r05_caps_sweep.json"""
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
p.set_tuning("tile_auto", 1)
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
u8, u16, f, g = dev.create_table_buffers([n, 2 * n, 4 * n, 4 * n])
capi.call("agpu_synth_u8", h, vp(u8), n, 6, 0)
capi.call("agpu_synth_u8", h, vp(u16), 2 * n, 7, 0)
capi.call("agpu_synth_f32", h, vp(f), n, 1, 0, C.c_float(0.001), C.c_float(1000.0))
p.sync()
K = {
    "sin_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(g), n)),
    "cos_f32": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.F32, vp(f), vp(g), n)),
    "cast_u8_f32": (5.0, lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(g), n)),
    "cast_u16_f32": (6.0, lambda: capi.call("agpu_cast", h, capi.U16, capi.F32, vp(u16), vp(g), n)),
    "sin_u8": (5.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(u8), vp(g), n)),
}
CAPS = [-1, 4200, 5000, 5600, 6200, 6800, 7400, 8000, 9000, 10240, 12000, 14000, -1, 6800, 10240]


def med(fn, reps=7):
    for _ in range(3):
        fn()
    p.sync()
    ts = []
    for _ in range(reps):
        q.begin(p); fn(); q.end(p)
        ts.append(q.wait_for_results())
    return sorted(ts)[len(ts) // 2]


out = {}
for name, (bpr, fn) in K.items():
    row = []
    for cap in CAPS:
        p.set_tuning("wave_lds", cap)
        row.append([cap, round(bpr * n / med(fn) / 1e6 / 8000.0, 4)])
    out[name] = row
    print(name, " ".join(f"{c}:{v:.3f}" for c, v in row), file=sys.stderr)
print(json.dumps(out))
