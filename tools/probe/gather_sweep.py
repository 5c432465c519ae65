#!/usr/bin/env python3
"""DEV TOOL: sweep random-gather (take) shapes and infer the line-fetch granularity.   python tools/probe/gather_sweep.py"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libgather_probe.so"))
vpt = C.c_void_p
lib.probe_take.argtypes = [vpt, vpt, vpt, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, vpt]
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "gather")
q = CmpQuery(dev)
h = p._handle
m = 1 << 26  # indices per launch
rows = []


def timeit(label, f, extra):
    f(); p.sync()
    ts = []
    for _ in range(7):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    ms = float(np.median(ts))
    r = {"kernel": label, "ms": round(ms, 4), "Grows_per_s": round(m / ms / 1e6, 2), "alg_TBps": round(12 * m / ms / 1e9, 3)}
    r.update(extra)
    rows.append(r)
    print(r, flush=True)


OUT = dev.create_empty_buffer(4 * m)
rng = np.random.default_rng(7)
for log_src in (28, 24, 20):
    nsrc = 1 << log_src
    V = dev.create_empty_buffer(4 * nsrc)
    capi.call("agpu_synth_i32", h, vpt(V.ptr), nsrc, 1, 0, 0)
    patterns = {
        "random": rng.integers(0, nsrc, m, dtype=np.uint32),
    }
    # pairs (i, i ^ 16): same 128-byte line, other 64-byte half; pairs (i, i ^ 8): same 64-byte half, other 32-byte sector
    base = rng.integers(0, nsrc, m // 2, dtype=np.uint32)
    for name, x in (("pair_other_64B_half", 16), ("pair_other_32B_sector", 8), ("pair_same_32B_sector", 1)):
        pr = np.empty(m, np.uint32)
        pr[0::2] = base
        pr[1::2] = base ^ np.uint32(x)
        patterns[name] = pr
    srt = np.sort(patterns["random"])
    patterns["sorted"] = srt
    for pname, idx in patterns.items():
        IDX = dev.create_gpu_buffer_with_data(idx)
        p.sync()
        timeit("PRODUCT take", lambda: capi.call("agpu_take", h, 4, vpt(V.ptr), nsrc, vpt(IDX.ptr), vpt(OUT.ptr), m),
               {"src_log2": log_src, "pattern": pname})
        if pname in ("random",):
            for g in (4, 8, 16):
                for nt in (0, 1):
                    for block in (64, 256):
                        def f(g=g, nt=nt, block=block):
                            rc = lib.probe_take(vpt(V.ptr), vpt(IDX.ptr), vpt(OUT.ptr), m, g, nt, block, 0, vpt(p.stream()))
                            assert rc == 0, rc
                        timeit("probe take", f, {"src_log2": log_src, "pattern": pname, "g": g, "nt": nt, "block": block})
        del IDX
    del V
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/gather_sweep.json", "w"), indent=1)
