#!/usr/bin/env python3
"""DEV TOOL (GPU box): cast u8→f32 and sin_u8 with one / two tiles per block as a function of WHERE the input and the output lie relative to
each other inside one allocation (units of 512 MiB + a colour of 0 / 4 / 8 / 12 KiB): what made two tiles per block win in one layout
(0.84 vs 0.80) and lose in the others (0.74 vs 0.80)?"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
UNIT = 512 << 20
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "off")
h = p._handle
q = CmpQuery(dev)
block = dev.create_empty_buffer(40 * UNIT)
base = block.ptr
capi.call("agpu_synth_u8", h, C.c_void_p(base), n, 6, 0)
p.sync()


def med(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        q.begin(p)
        fn()
        q.end(p)
        ts.append(q.wait_for_results())
    return sorted(ts)[len(ts) // 2]


out = []
# input at unit ui, output at unit uo (the sweep layout that liked two tiles per block had the input at unit 28 and the output at unit 12)
PAIRS = [(28, 12), (0, 12), (28, 0), (16, 8), (12, 28), (20, 4), (30, 12), (26, 12), (28, 14), (28, 10), (24, 12), (4, 12), (8, 12), (28, 20), (29, 12), (28, 13)]
for ui, uo in PAIRS:
    src = base + ui * UNIT
    dst = base + uo * UNIT + 4096
    capi.call("agpu_synth_u8", h, C.c_void_p(src), n, 6, 0)
    p.sync()
    row = {"in_unit": ui, "out_unit": uo}
    for name, key, fn in (("cast", "cast_tiles", lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, C.c_void_p(src), C.c_void_p(dst), n)),
                          ("sin_u8", "table_tiles", lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, C.c_void_p(src), C.c_void_p(dst), n))):
        for k in (1, 2):
            p.set_tuning(key, k)
            row[f"{name}_k{k}"] = round(5.0 * n / med(fn) / 1e6 / 8000, 4)
        p.set_tuning(key, 0)
    out.append(row)
    print(row, file=sys.stderr)
print(json.dumps(out))
