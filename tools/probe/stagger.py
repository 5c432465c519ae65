#!/usr/bin/env python3
"""DEV TOOL: does the relative placement of the three streams of `a + b -> out` matter (HBM channel / bank aliasing)?
All three columns are carved out of ONE allocation; b and out are shifted by the given byte offsets from their
4 GB-aligned slots.   python tools/probe/stagger.py [--rows 1000000000]"""
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
    ap.add_argument("--iters", type=int, default=9)
    args = ap.parse_args()
    n = args.rows
    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "stagger")
    q = CmpQuery(dev)
    h = p._handle
    slot = (4 * n + (1 << 30) - 1) // (1 << 30) * (1 << 30) + (1 << 30)  # 1 GiB-aligned slots with 1 GiB of slack
    big = dev.create_empty_buffer(3 * slot)
    base = (big.ptr + (1 << 21) - 1) // (1 << 21) * (1 << 21)
    vp = C.c_void_p
    rows = []
    offsets = [(0, 0), (256, 512), (1024, 2048), (4096, 8192), (4096 + 256, 8192 + 512), (65536, 131072),
               (1 << 20, 1 << 21), ((1 << 20) + 4096, (1 << 21) + 8192), (12345 * 16, 54321 * 16), (1 << 28, 1 << 29), (0, 0)]
    for ob, oo in offsets:
        a, b, o = base, base + slot + ob, base + 2 * slot + oo
        capi.call("agpu_synth_f32", h, vp(a), n, 1, 0, C.c_float(-1000), C.c_float(1000))
        capi.call("agpu_synth_f32", h, vp(b), n, 2, 0, C.c_float(-1000), C.c_float(1000))
        p.sync()

        def f():
            capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(a), vp(b), vp(o), n)

        f()
        p.sync()
        ts = []
        for _ in range(args.iters):
            q.begin(p)
            f()
            q.end(p)
            ts.append(q.wait_for_results())
        ms = float(np.median(ts))
        r = {"off_b": ob, "off_out": oo, "ms": round(ms, 4), "TBps": round(12 * n / ms / 1e9, 3), "min_ms": round(min(ts), 4)}
        rows.append(r)
        print(r, flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/stagger.json", "w"), indent=1)


if __name__ == "__main__":
    main()
