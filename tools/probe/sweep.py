#!/usr/bin/env python3
"""DEV TOOL: sweep launch shapes of the two headline streams on the GPU box, with the product kernels timed in the
same process (interleaved A/B).   python tools/probe/sweep.py --rows 1000000000 [--what add,eq]"""
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
    ap.add_argument("--what", default="add,eq")
    ap.add_argument("--tag", default="r01b")
    args = ap.parse_args()
    n = args.rows
    here = os.path.dirname(os.path.abspath(__file__))
    lib = C.CDLL(os.path.join(here, "libstream_probe.so"))
    vpt = C.c_void_p
    lib.probe_add.argtypes = [vpt, vpt, vpt, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vpt]
    lib.probe_eq.argtypes = [vpt, vpt, vpt, vpt, vpt, vpt, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vpt]
    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "sweep")
    q = CmpQuery(dev)
    h = p._handle
    A, B, O = (dev.create_empty_buffer(4 * n) for _ in range(3))
    nb = (n + 63) // 64 * 8
    VA, VB, OB, OV = (dev.create_empty_buffer(nb) for _ in range(4))
    capi.call("agpu_synth_f32", h, vpt(A.ptr), n, 1, 0, C.c_float(-1000), C.c_float(1000))
    capi.call("agpu_synth_f32", h, vpt(B.ptr), n, 2, 0, C.c_float(-1000), C.c_float(1000))
    capi.call("agpu_synth_bits", h, vpt(VA.ptr), n, 3, 0, C.c_double(0.9))
    capi.call("agpu_synth_bits", h, vpt(VB.ptr), n, 4, 0, C.c_double(0.9))
    p.sync()
    stream = vpt(p.stream())
    rows = []

    def timeit(label, f, alg_bytes, extra=None):
        f()
        p.sync()
        ts = []
        for _ in range(args.iters):
            q.begin(p)
            f()
            q.end(p)
            ts.append(q.wait_for_results())
        ms = float(np.median(ts))
        r = {"kernel": label, "ms": round(ms, 4), "TBps": round(alg_bytes / ms / 1e9, 3), "min_ms": round(min(ts), 4)}
        if extra:
            r.update(extra)
        rows.append(r)
        print(r, flush=True)

    what = args.what.split(",")
    if "add" in what:
        def product_add():
            capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vpt(A.ptr), vpt(B.ptr), vpt(O.ptr), n)

        def probe_add(u, nt, block, grid, xcd):
            def f():
                rc = lib.probe_add(vpt(A.ptr), vpt(B.ptr), vpt(O.ptr), n, u, nt, block, grid, xcd, stream)
                assert rc == 0, rc
            return f

        timeit("PRODUCT add_f32", product_add, 12 * n)
        for block in (64, 128, 256, 512):
            for u in (1, 2):
                for xcd in (0, 1):
                    timeit("probe add", probe_add(u, 3, block, 0, xcd), 12 * n, {"u": u, "nt": 3, "block": block, "grid": 0, "xcd": xcd})
        timeit("PRODUCT add_f32", product_add, 12 * n)
        for grid in (32768, 65536, 131072, 262144):
            for u in (1, 2, 4):
                timeit("probe add", probe_add(u, 3, 256, grid, 0), 12 * n, {"u": u, "nt": 3, "block": 256, "grid": grid, "xcd": 0})
        timeit("PRODUCT add_f32", product_add, 12 * n)

    if "eq" in what:
        def product_eq():
            capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, vpt(A.ptr), vpt(B.ptr), vpt(VA.ptr), vpt(VB.ptr),
                      vpt(OB.ptr), vpt(OV.ptr), n)

        def probe_eq(variant, ru, nt, block, grid):
            def f():
                rc = lib.probe_eq(vpt(A.ptr), vpt(B.ptr), vpt(VA.ptr), vpt(VB.ptr), vpt(OB.ptr), vpt(OV.ptr), n, variant, ru,
                                  nt, block, grid, stream)
                assert rc == 0, rc
            return f

        timeit("PRODUCT eq_i32+validity", product_eq, 8.5 * n)
        for block in (64, 128, 256, 512):
            for r in (1, 2, 4, 8, 16):
                timeit("probe eq ballot", probe_eq(0, r, 1, block, 0), 8.5 * n, {"variant": 0, "R": r, "nt": 1, "block": block, "grid": 0})
        for block in (64, 128, 256, 512):
            for u in (1, 2, 4):
                timeit("probe eq vec", probe_eq(1, u, 1, block, 0), 8.5 * n, {"variant": 1, "U": u, "nt": 1, "block": block, "grid": 0})
        for r in (4, 16):
            timeit("probe eq ballot", probe_eq(0, r, 0, 256, 0), 8.5 * n, {"variant": 0, "R": r, "nt": 0, "block": 256, "grid": 0})
        timeit("PRODUCT eq_i32+validity", product_eq, 8.5 * n)

    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/sweep_{args.tag}.json", "w") as f:
        json.dump(rows, f, indent=1)
    print("BEST", sorted(rows, key=lambda r: -r["TBps"])[:10])


if __name__ == "__main__":
    main()
