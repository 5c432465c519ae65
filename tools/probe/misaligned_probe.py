#!/usr/bin/env python3
"""DEV TOOL: can an XCD-contiguous tile mapping and/or cacheable loads win back what columns that are only 16-byte aligned
lose (tools/probe/stagger.py)?  Uses the stream probe's add kernel (u, nt, block, grid, xcd) on shifted pointers."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libstream_probe.so"))
vpt = C.c_void_p
lib.probe_add.argtypes = [vpt, vpt, vpt, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vpt]
n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "mis")
q = CmpQuery(dev)
h = p._handle
slot = 5 << 30
big = dev.create_empty_buffer(3 * slot)
base = (big.ptr + (1 << 21) - 1) // (1 << 21) * (1 << 21)
for ob, oo in ((0, 0), (12345 * 16, 54321 * 16), (256, 512)):
    a, b, o = base, base + slot + ob, base + 2 * slot + oo
    capi.call("agpu_synth_f32", h, vpt(a), n, 1, 0, C.c_float(-1000), C.c_float(1000))
    capi.call("agpu_synth_f32", h, vpt(b), n, 2, 0, C.c_float(-1000), C.c_float(1000))
    p.sync()
    for xcd in (0, 1):
        for nt in (3, 2, 0):
            def f():
                rc = lib.probe_add(vpt(a), vpt(b), vpt(o), n, 1, nt, 64, 0, xcd, vpt(p.stream()))
                assert rc == 0
            f(); p.sync()
            ts = []
            for _ in range(7):
                q.begin(p); f(); q.end(p)
                ts.append(q.wait_for_results())
            ms = float(np.median(ts))
            print({"off_b": ob, "off_out": oo, "xcd": xcd, "nt": nt, "ms": round(ms, 4), "TBps": round(12 * n / ms / 1e9, 3)}, flush=True)
