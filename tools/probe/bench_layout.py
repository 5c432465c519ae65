#!/usr/bin/env python3
"""The bench's step (f32 add, then i32 eq + validity, alternating, 20 steps) with its nine buffers (A) as nine pool
blocks, the way bench.py allocates them, and (B) carved out of ONE allocation at chosen distances.  One process."""
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
p = ArrowComputePipeline(dev, "bench")
h = p._handle
G, M, K = 1 << 30, 1 << 20, 1 << 10
n = 1_000_000_000
nb = (n + 63) // 64 * 8
vp = C.c_void_p


def ev():
    e = C.c_void_p()
    capi.call("agpu_event_create", dev._handle, C.byref(e))
    return e


def run(label, fa, fb, fo, ia, ib, va, vb, ob, ov, steps=20):
    capi.call("agpu_synth_f32", h, vp(fa), n, 1, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_synth_f32", h, vp(fb), n, 2, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_synth_i32", h, vp(ia), n, 3, 0, 1024)
    capi.call("agpu_synth_i32", h, vp(ib), n, 4, 0, 1024)
    capi.call("agpu_synth_bits", h, vp(va), n, 5, 0, C.c_double(0.9))
    capi.call("agpu_synth_bits", h, vp(vb), n, 6, 0, C.c_double(0.9))
    p.sync()
    add_ev = [(ev(), ev()) for _ in range(steps)]
    eq_ev = [(ev(), ev()) for _ in range(steps)]

    def step(i=None):
        if i is not None:
            capi.call("agpu_event_record", add_ev[i][0], h)
        capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(fa), vp(fb), vp(fo), n)
        if i is not None:
            capi.call("agpu_event_record", add_ev[i][1], h)
            capi.call("agpu_event_record", eq_ev[i][0], h)
        capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(va), vp(vb), vp(ob), vp(ov), n)
        if i is not None:
            capi.call("agpu_event_record", eq_ev[i][1], h)

    for _ in range(3):
        step()
    p.sync()
    for i in range(steps):
        step(i)
    p.sync()

    def mean_ms(pairs):
        tot = 0.0
        for s, e in pairs:
            ms = C.c_float()
            capi.call("agpu_event_elapsed_ms", s, e, C.byref(ms))
            tot += ms.value
        return tot / len(pairs)

    a, e = mean_ms(add_ev), mean_ms(eq_ev)
    r = {"layout": label, "add_ms": round(a, 4), "add_frac": round(12 * n / a / 1e6 / 8000, 4), "eq_validity_ms": round(e, 4),
         "eq_validity_frac": round(8.5 * n / e / 1e6 / 8000, 4), "step_GBps": round(20.5 * n / (a + e) / 1e6, 1)}
    print(json.dumps(r), flush=True)
    return r


rows = []
for rep in range(2):
    bufs = [dev.create_empty_buffer(4 * n) for _ in range(5)] + [dev.create_empty_buffer(nb) for _ in range(4)]
    rows.append(run("nine pool blocks (bench.py)", *[b.ptr for b in bufs]))
    rows[-1]["offsets_from_first"] = [hex(b.ptr - bufs[0].ptr) for b in bufs]
    print("   ", rows[-1]["offsets_from_first"])
    del bufs
    capi.call("agpu_device_trim", dev._handle)
    big = dev.create_empty_buffer(22 * G)
    base = big.ptr
    B = base + 21 * G
    S = 4001366016
    for label, offs in (("slab: columns 4 GiB apart", (0, 4 * G, 8 * G, 12 * G, 16 * G)),
                        ("slab: columns 4 GiB + 8K/4K/12K/8K", (0, 4 * G + 8 * K, 8 * G + 4 * K, 12 * G + 12 * K, 16 * G + 8 * K)),
                        ("slab: a, b = a + 4G + 2M, out = a + 8G + 8K; ia, ib = ia + 4G + 8K", (0, 4 * G + 2 * M, 8 * G + 8 * K, 12 * G, 16 * G + 8 * K)),
                        ("slab: columns S apart (what consecutive pool blocks would be)", (0, S, 2 * S, 3 * S, 4 * S))):
        rows.append(run(label, base + offs[0], base + offs[1], base + offs[2], base + offs[3], base + offs[4],
                        B, B + 128 * M + 8 * K, B + 256 * M + 4 * K, B + 384 * M + 12 * K))
    del big
    capi.call("agpu_device_trim", dev._handle)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": __doc__, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "bench_layout.json"), "w"), indent=1)
