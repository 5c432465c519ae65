#!/usr/bin/env python3
"""Round 6, VERDICT r5 item 1: what an ordinary caller's allocations give the headline kernels.

One process = one scenario (placement differs by process, so the shell script starts several):
  --arena 0|1     tuning pool_arena
  --churn 0|1     before measuring, do what bench.py's earlier legs do to the pool: tables allocated and freed, 1-column
                  tables (plain 4 GiB blocks) left in the cache
  --reps K        K times: three/five fresh pool blocks (a, b through agpu_malloc; the OUTPUT through the host API's a.add(b), i.e.
                  agpu_malloc_like), timed by the library's per-launch events; everything freed in between (no trim)
Prints one JSON line per repetition with the pointers (hex) and the fractions of the 8 TB/s roof."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import arrow_gpu_amd as ag  # noqa: E402
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--arena", type=int, default=1)
ap.add_argument("--churn", type=int, default=0)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--rows", type=int, default=1_000_000_000)
ap.add_argument("--tune", action="append", default=[])
args = ap.parse_args()
capi.call("agpu_set_tuning", b"pool_arena", args.arena)
for kv in args.tune:
    k, v = kv.split("=", 1)
    capi.call("agpu_set_tuning", k.encode(), int(v))
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "probe")
h = p._handle
n = args.rows
nb = (n + 63) // 64 * 8
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731


def synth(fa, fb, ia, ib, va, vb):
    capi.call("agpu_synth_f32", h, vp(fa), n, 1, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_synth_f32", h, vp(fb), n, 2, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_synth_i32", h, vp(ia), n, 3, 0, 1024)
    capi.call("agpu_synth_i32", h, vp(ib), n, 4, 0, 1024)
    capi.call("agpu_synth_bits", h, vp(va), n, 5, 0, C.c_double(0.9))
    capi.call("agpu_synth_bits", h, vp(vb), n, 6, 0, C.c_double(0.9))
    p.sync()


if args.churn:
    t1 = dev.create_table_buffers([4 * n] * 3)
    t2 = dev.create_table_buffers([4 * n] * 2 + [nb] * 4)
    u8, = dev.create_table_buffers([n])
    fd, = dev.create_table_buffers([4 * n])
    tmp, = dev.create_table_buffers([4 * n])
    q = dev.create_table_buffers([4 * n] * 4 + [nb])
    e1 = [dev.create_empty_buffer(4 * n) for _ in range(2)]
    p.sync()
    del u8, fd, tmp, q, e1
    p.sync()

p.enable_timing(2)


def kernel_ms(op, k=7):
    ts, last = [], None
    for i in range(k + 2):
        r = op()
        ns, _ = p.last_kernel_ns()
        p.sync()
        last = r
        if i >= 2:
            ts.append(ns / 1e6)
        del r
    return float(np.median(ts)), last


for rep in range(args.reps):
    fa, fb, ia, ib = (dev.create_empty_buffer(4 * n) for _ in range(4))
    va, vb = dev.create_empty_buffer(nb), dev.create_empty_buffer(nb)
    synth(fa, fb, ia, ib, va, vb)
    A = ag.Float32ArrayGPU(fa, dev, n, None)
    B = ag.Float32ArrayGPU(fb, dev, n, None)
    IA = ag.Int32ArrayGPU(ia, dev, n, ag.NullBitBufferGpu(va, n, dev))
    IB = ag.Int32ArrayGPU(ib, dev, n, ag.NullBitBufferGpu(vb, n, dev))
    t_add, r_add = kernel_ms(lambda: A.add_op(B, p))
    out_ptr = r_add.data.ptr
    del r_add
    t_eq, r_eq = kernel_ms(lambda: IA.eq_op(IB, p))
    ob_ptr = r_eq.data.ptr
    del r_eq
    print(json.dumps({"arena": args.arena, "churn": args.churn, "rep": rep,
                      "add_frac": round(12.0 * n / t_add / 1e6 / 8000, 4), "eq_frac": round(8.5 * n / t_eq / 1e6 / 8000, 4),
                      "add_ms": round(t_add, 4), "eq_ms": round(t_eq, 4),
                      "ptrs": {"fa": hex(fa.ptr), "fb": hex(fb.ptr), "fo": hex(out_ptr), "ia": hex(ia.ptr), "ib": hex(ib.ptr),
                               "va": hex(va.ptr), "vb": hex(vb.ptr), "ob": hex(ob_ptr)}}), flush=True)
    del A, B, IA, IB, fa, fb, ia, ib, va, vb
    p.sync()
