#!/usr/bin/env python3
"""DEV TOOL (GPU box): does two-tiles-per-block win only when the shader clock is up?  cast u8→f32 / sin_u8 at k = 1, 2 — cold (after memory-bound
launches only) and right after a burst of VALU-heavy launches (f32 pow), alternating; also reports how long a kernel keeps the gain."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "clk")
h = p._handle
q = CmpQuery(dev)
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
f, g, g2 = dev.create_table_buffers([4 * n] * 3)
u8, = dev.create_table_buffers([n])
capi.call("agpu_synth_u8", h, vp(u8), n, 6, 0)
capi.call("agpu_synth_f32", h, vp(f), n, 1, 0, C.c_float(0.5), C.c_float(2.0))
capi.call("agpu_synth_f32", h, vp(g2), n, 2, 0, C.c_float(-3.0), C.c_float(3.0))
p.sync()
cast = lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(g), n)  # noqa: E731
sin8 = lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(u8), vp(g), n)  # noqa: E731
heavy = lambda: capi.call("agpu_binary", h, capi.OP_POW, capi.F32, vp(f), vp(g2), vp(g), n)  # noqa: E731
copy = lambda: capi.call("agpu_copy", h, vp(g), vp(f), 4 * n)  # noqa: E731


def series(fn, reps):
    ts = []
    for _ in range(reps):
        q.begin(p)
        fn()
        q.end(p)
        ts.append(round(5.0 * n / q.wait_for_results() / 1e6 / 8000, 3))
    return ts


out = {}
for name, key, fn in (("cast_u8_f32", "cast_tiles", cast), ("sin_u8", "table_tiles", sin8)):
    for k in (1, 2, 1, 2):
        p.set_tuning(key, k)
        for _ in range(30):
            copy()
        p.sync()
        cold = series(fn, 12)
        for _ in range(30):
            heavy()
        hot = series(fn, 12)
        out.setdefault(f"{name}/k{k}", []).append({"after_copies": cold, "after_pow": hot})
        print(name, "k", k, "after copies", cold, "after pow", hot, file=sys.stderr)
    p.set_tuning(key, 0)
print(json.dumps(out))
