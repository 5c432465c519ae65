#!/usr/bin/env python3
"""DEV TOOL (GPU box): why do the prefetching tile loops win in tools/probe/prefetch_sweep.py and lose inside bench.py?  The same three
kernels (cast u8→f32, sin_u8, sin_f32) at tiles-per-block = default and = 1, over {one table | bench.py's allocations} × {a host sync
between launches | launches back to back}."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "ctx")
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731


def ev():
    e = C.c_void_p()
    capi.call("agpu_event_create", dev._handle, C.byref(e))
    return e


def ms_between(s, e):
    ms = C.c_float()
    capi.call("agpu_event_elapsed_ms", s, e, C.byref(ms))
    return ms.value


def timed_b2b(fn, reps=9):
    fn(), fn()
    pairs = [(ev(), ev()) for _ in range(reps)]
    for s_, e_ in pairs:
        capi.call("agpu_event_record", s_, h)
        fn()
        capi.call("agpu_event_record", e_, h)
    ts = sorted(ms_between(s_, e_) for s_, e_ in pairs)
    return ts[len(ts) // 2]


q = CmpQuery(dev)


def timed_sync(fn, reps=9):
    fn(), fn()
    ts = []
    for _ in range(reps):
        q.begin(p)
        fn()
        q.end(p)
        ts.append(q.wait_for_results())
    return sorted(ts)[len(ts) // 2]


out = {}
for layout in ("table", "bench"):
    if layout == "table":
        u8, f, g = dev.create_table_buffers([n, 4 * n, 4 * n])
    else:
        f, g, g2 = dev.create_table_buffers([4 * n] * 3)
        ia, ib, va, vb, ob, ov = dev.create_table_buffers([4 * n] * 2 + [(n + 63) // 64 * 8] * 4)
        u8, = dev.create_table_buffers([n])
    capi.call("agpu_synth_u8", h, vp(u8), n, 6, 0)
    capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(f), n)
    p.sync()
    kernels = {"cast_u8_f32": ("cast_tiles", 5.0, lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(g), n)),
               "sin_u8": ("table_tiles", 5.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(u8), vp(g), n)),
               "sin_f32": ("heavy_tiles", 8.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(g), n))}
    for name, (key, bpr, fn) in kernels.items():
        for timing, tf in (("sync", timed_sync), ("b2b", timed_b2b)):
            row = {}
            for k in (0, 1, 0, 1):
                p.set_tuning(key, k)
                ms = tf(fn)
                row.setdefault("default" if k == 0 else "one", []).append(round(bpr * n / ms / 1e6 / 8000.0, 4))
            p.set_tuning(key, 0)
            out[f"{layout}/{name}/{timing}"] = row
            print(layout, name, timing, row, file=sys.stderr)
    del u8, f, g
    if layout == "bench":
        del g2, ia, ib, va, vb, ob, ov
    capi.call("agpu_device_trim", dev._handle)
print(json.dumps(out))
