#!/usr/bin/env python3
"""DEV TOOL: where does the time of a host-layer `a.add(b)` go beyond the kernel?  (stream creation, allocation,
first touch of fresh memory, buffer release).   python tools/probe/alloc_probe.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import arrow_gpu_amd as ag  # noqa: E402
from arrow_gpu_amd import _capi as capi  # noqa: E402

dev = ag.GPU_DEVICE()


def med(f, k=7):
    ts = []
    for _ in range(k):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 4)


for nbytes in (1 << 20, 1 << 26, 1 << 30, 4_000_000_000):
    bufs = []
    t_m = med(lambda: bufs.append(dev.create_empty_buffer(nbytes)), 5)
    t_f = med(lambda: bufs.pop(), 5)
    print(f"{nbytes:>11} B: malloc {t_m} ms, free {t_f} ms")

pipes = []
print("pipeline create ms", med(lambda: pipes.append(ag.ArrowComputePipeline(dev, "x"))), "destroy ms", med(lambda: pipes.pop()))

n = 100_000_000
a = ag.Float32ArrayGPU.broadcast(1.5, n, dev)
b = ag.Float32ArrayGPU.broadcast(2.5, n, dev)
p = ag.ArrowComputePipeline(dev, "probe")
vp = lambda x: C.c_void_p(x.ptr)  # noqa: E731


def kernel_into(out):
    capi.call("agpu_binary", p._handle, capi.OP_ADD, capi.F32, vp(a.data), vp(b.data), vp(out), n)
    p.sync()


fresh = [dev.create_empty_buffer(4 * n) for _ in range(7)]
it = iter(fresh)
print("kernel + sync into FRESH buffers ms", med(lambda: kernel_into(next(it))))
print("kernel + sync into a TOUCHED buffer ms", med(lambda: kernel_into(fresh[0])))
del fresh, it


def api_add():
    c = a.add(b)
    p.sync()
    return c


print("host-layer a.add(b) (pipeline + alloc + kernel + validity) ms", med(api_add))
keep = []
print("  … keeping the results alive ms", med(lambda: keep.append(api_add())))
