#!/usr/bin/env python3
"""DEV TOOL: sweep launch shapes of the f32 add stream on the GPU box.  python tools/probe/sweep.py --rows 1000000000"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--iters", type=int, default=7)
    args = ap.parse_args()
    n = args.rows
    here = os.path.dirname(os.path.abspath(__file__))
    lib = C.CDLL(os.path.join(here, "libstream_probe.so"))
    lib.probe_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "sweep")
    q = CmpQuery(dev)
    A, B, O = (dev.create_empty_buffer(4 * n) for _ in range(3))
    capi.call("agpu_synth_f32", p._handle, C.c_void_p(A.ptr), n, 1, 0, C.c_float(-1000), C.c_float(1000))
    capi.call("agpu_synth_f32", p._handle, C.c_void_p(B.ptr), n, 2, 0, C.c_float(-1000), C.c_float(1000))
    p.sync()
    stream = C.c_void_p(p.stream())
    rows = []

    def run(u, nt, block, grid):
        def f():
            rc = lib.probe_add(C.c_void_p(A.ptr), C.c_void_p(B.ptr), C.c_void_p(O.ptr), n, u, nt, block, grid, stream)
            assert rc == 0, rc
        f()
        p.sync()
        ts = []
        for _ in range(args.iters):
            q.begin(p)
            f()
            q.end(p)
            ts.append(q.wait_for_results())
        ms = float(np.median(ts))
        r = {"u": u, "nt": nt, "block": block, "grid": grid, "ms": round(ms, 4), "TBps": round(12 * n / ms / 1e9, 3),
             "min_ms": round(min(ts), 4)}
        rows.append(r)
        print(r, flush=True)

    for block in (256, 512, 1024):
        for u in (1, 2, 4, 8):
            for nt in (0, 1, 2, 3):
                run(u, nt, block, 0)
    for grid in (2048, 4096, 8192, 16384, 32768):
        for u in (2, 4):
            for nt in (0, 3):
                run(u, nt, 256, grid)
    rows.sort(key=lambda r: r["ms"])
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/sweep_add.json", "w") as f:
        json.dump(rows, f, indent=1)
    print("BEST", rows[:8])


if __name__ == "__main__":
    main()
