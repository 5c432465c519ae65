#!/usr/bin/env python3
"""take / put: direct kernels vs the bucketed form (swizzle.hip), one process, same buffers, uniformly random indices.
Writes gpurun_out/bucket_sweep.json.  Usage: python tools/probe/bucket_sweep.py [--quick] [--only-bucketed]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice  # noqa: E402

dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "sweep")
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731


def ev():
    e = C.c_void_p()
    capi.call("agpu_event_create", dev._handle, C.byref(e))
    return e


def timeit(fn, iters=5):
    fn()
    p.sync()
    s, e = ev(), ev()
    capi.call("agpu_event_record", s, h)
    for _ in range(iters):
        fn()
    capi.call("agpu_event_record", e, h)
    ms = C.c_float()
    capi.call("agpu_event_elapsed_ms", s, e, C.byref(ms))
    return ms.value / iters


quick = "--quick" in sys.argv
for a in sys.argv:
    if a.startswith("--region-bits="):
        p.set_tuning("gather_region_bits", int(a.split("=")[1]))
    if a.startswith("--offsets="):
        p.set_tuning("gather_offsets", int(a.split("=")[1]))
modes = (("bucketed", 2),) if "--only-bucketed" in sys.argv else (("direct", 1), ("bucketed", 2), ("auto", 0))  # auto: size thresholds + the probe
rows = []
shapes = [(1 << 26, 1 << 22), (1 << 26, 1 << 26), (1 << 28, 1 << 24), (1 << 28, 1 << 26), (1 << 28, 1 << 28)]
if quick:
    shapes = [(1 << 26, 1 << 26), (1 << 28, 1 << 28)]
if "--crossover" in sys.argv:
    shapes = [(n, nv) for n in (1 << 18, 1 << 20, 1 << 22, 1 << 24) for nv in (1 << 20, 1 << 24, 1 << 28)]
if "--crossover3" in sys.argv:  # round 3: where does the merge-back take start to win?
    shapes = [(n, nv) for n in (1 << 21, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 26) for nv in (1 << 22, 1 << 24, 1 << 26, 1 << 28)]
for n, nv in shapes:
    values, out = dev.create_empty_buffer(4 * nv), dev.create_empty_buffer(4 * n)
    idx, idx2 = dev.create_empty_buffer(4 * n), dev.create_empty_buffer(4 * n)
    capi.call("agpu_synth_i32", h, vp(values), nv, 1, 0, 0)
    capi.call("agpu_synth_i32", h, vp(idx), n, 2, 0, nv)   # uniform on [0, nv)
    capi.call("agpu_synth_i32", h, vp(idx2), n, 3, 0, n)   # uniform on [0, n)
    p.sync()
    rec = {"n_idx": n, "n_values": nv}
    for name, mode in modes:
        p.set_tuning("gather_bucket", mode)
        ms = timeit(lambda: capi.call("agpu_take", h, 4, vp(values), nv, vp(idx), vp(out), n))
        rec[f"take_{name}_ms"] = round(ms, 3)
        rec[f"take_{name}_Grows_s"] = round(n / ms / 1e6, 1)
        cs = dev.create_empty_buffer(16)
        capi.call("agpu_checksum", h, vp(out), 4 * n, vp(cs))
        rec[f"take_{name}_checksum"] = int(dev.retrive_data(cs, 8, pipeline=p).view(np.uint64)[0])
    if len(modes) >= 2:
        rec["take_same_result"] = rec["take_direct_checksum"] == rec["take_bucketed_checksum"]
        rec["take_speedup"] = round(rec["take_direct_ms"] / rec["take_bucketed_ms"], 2)
    # put: src = values (nv rows, random source index), dst = out (n rows, random destination index; duplicates have no
    # defined winner, so no checksum here — tests/test_gpu_bucketed.py checks results with distinct destinations)
    for name, mode in modes:
        p.set_tuning("gather_bucket", mode)
        ms = timeit(lambda: capi.call("agpu_put_bounded", h, 4, vp(values), nv, vp(idx), vp(out), n, vp(idx2), n))
        rec[f"put_{name}_ms"] = round(ms, 3)
        rec[f"put_{name}_Grows_s"] = round(n / ms / 1e6, 1)
    if len(modes) >= 2:
        rec["put_speedup"] = round(rec["put_direct_ms"] / rec["put_bucketed_ms"], 2)
    print(json.dumps(rec), flush=True)
    rows.append(rec)
    del values, out, idx, idx2
    capi.call("agpu_device_trim", dev._handle)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": "take/put direct vs bucketed, uniformly random u32 indices, 4-byte values", "rows": rows},
          open(os.path.join(ROOT, "gpurun_out", "bucket_sweep.json"), "w"), indent=1)
