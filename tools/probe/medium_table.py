#!/usr/bin/env python3
"""DEV TOOL (round 5): tables whose columns are 64 MiB … 1 GiB: 2 MiB granules + colour (AGPU_TABLE_BIG_COLUMN_MIB unset = 1024) against the
big-column rule (512 MiB multiples + colour) from 64 MiB up (AGPU_TABLE_BIG_COLUMN_MIB=64).  f32 add, i32 eq, sin f32 at 1.6e7 … 2.5e8 rows."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "medt"); q = CmpQuery(dev); h = p._handle
vp = lambda b: C.c_void_p(b.ptr)
def med(fn):
    for _ in range(6): fn()
    p.sync(); ts = []
    for _ in range(15):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return float(np.median(ts))
row = {"big_column_MiB": os.environ.get("AGPU_TABLE_BIG_COLUMN_MIB", "1024")}
for n in (16_000_000, 50_000_000, 100_000_000, 150_000_000, 200_000_000, 250_000_000):
    a, b, c, bm = dev.create_table_buffers([4 * n, 4 * n, 4 * n, n // 8 + 64])
    capi.call("agpu_synth_f32", h, vp(a), n, 1, 0, C.c_float(-3.0), C.c_float(3.0)); capi.call("agpu_synth_f32", h, vp(b), n, 2, 0, C.c_float(-3.0), C.c_float(3.0)); p.sync()
    ms = med(lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(a), vp(b), vp(c), n)); row[f"add {n:.0e}"] = round(12.0 * n / ms / 1e6 / 8000, 3)
    ms = med(lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(a), vp(b), vp(bm), n)); row[f"eq {n:.0e}"] = round(8.125 * n / ms / 1e6 / 8000, 3)
    ms = med(lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(a), vp(c), n)); row[f"sin {n:.0e}"] = round(8.0 * n / ms / 1e6 / 8000, 3)
    del a, b, c, bm
print(json.dumps(row), flush=True)
