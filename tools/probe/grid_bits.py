#!/usr/bin/env python3
"""DEV TOOL (GPU box): two tiles per block, walked G tiles apart — which G are good?  cast u8→f32 (1 KiB in / 4 KiB out per tile) and sin_u8
(lut8: 2 KiB in / 8 KiB out per tile) with the grid forced to G = ceil(tiles / 2) + d (tuning stream_grid), d over the low bits and over single higher
bits, in two buffer layouts.  Prints frac of the HBM roof per d."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "grid")
h = p._handle
q = CmpQuery(dev)
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731


def med(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        q.begin(p)
        fn()
        q.end(p)
        ts.append(q.wait_for_results())
    return sorted(ts)[len(ts) // 2]


DS = list(range(0, 32)) + [1 << k for k in range(5, 20)] + [(1 << k) + 2 for k in range(5, 20)]
out = {}
for layout in ("table", "bench"):
    if layout == "table":
        u8, f, g = dev.create_table_buffers([n, 4 * n, 4 * n])
    else:
        f, g, g2 = dev.create_table_buffers([4 * n] * 3)
        u8, = dev.create_table_buffers([n])
    capi.call("agpu_synth_u8", h, vp(u8), n, 6, 0)
    p.sync()
    for name, tile_rows, fn in (("cast_u8_f32", 1024, lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(g), n)),
                                ("sin_u8", 2048, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(u8), vp(g), n))):
        tiles = n // tile_rows
        x = (tiles + 1) // 2
        p.set_tuning("stream_grid", 0)
        p.set_tuning("cast_tiles", 1), p.set_tuning("table_tiles", 1)
        base = 5.0 * n / med(fn) / 1e6 / 8000
        row = {}
        for d in DS:
            p.set_tuning("stream_grid", x + d)
            row[d] = round(5.0 * n / med(fn) / 1e6 / 8000, 4)
        p.set_tuning("stream_grid", 0)
        p.set_tuning("cast_tiles", 0), p.set_tuning("table_tiles", 0)
        out[f"{layout}/{name}"] = {"one_tile_per_block": round(base, 4), "x": x, "by_d": row}
        print(layout, name, "one tile:", round(base, 4), "x =", x, file=sys.stderr)
        print("  low bits :", " ".join(f"{d}:{row[d]:.3f}" for d in range(32)), file=sys.stderr)
        print("  1<<k     :", " ".join(f"{k}:{row[1 << k]:.3f}" for k in range(5, 20)), file=sys.stderr)
        print("  (1<<k)+2 :", " ".join(f"{k}:{row[(1 << k) + 2]:.3f}" for k in range(5, 20)), file=sys.stderr)
    del u8, f, g
    capi.call("agpu_device_trim", dev._handle)
print(json.dumps(out))
