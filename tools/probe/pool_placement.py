#!/usr/bin/env python3
"""A/B of the pool's placed arenas (runtime.hip "pool placement", tuning "pool_arena") in ONE process on the same box:
the bench step (f32 add; i32 eq + validity AND) at 1e9 rows over
  A  nine ordinary agpu_malloc blocks, pool_arena = 0  (one hipMalloc per block: round 2's behaviour)
  B  nine ordinary agpu_malloc blocks, pool_arena = 1  (≥ 1 GiB blocks carved from an arena at 512 MiB multiples + rotating colour)
  C  two agpu_malloc_table tables                       (the layout bench.py's headline uses)
and the HOST API exactly as a caller uses it — Int32ArrayGPU.eq / Float32ArrayGPU.add on arrays whose buffers came from
plain create_empty_buffer, outputs allocated by the op — under both settings.  Several repetitions with everything freed
and trimmed in between, so that each repetition gets fresh placements.  Prints one JSON object."""
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

dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "bench")
h = p._handle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nb = (n + 63) // 64 * 8
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731


def ev():
    e = C.c_void_p()
    capi.call("agpu_event_create", dev._handle, C.byref(e))
    return e


def ms_of(f, k=9):
    f(), f()
    ts = []
    for _ in range(k):
        s, e = ev(), ev()
        capi.call("agpu_event_record", s, h)
        f()
        capi.call("agpu_event_record", e, h)
        ms = C.c_float()
        capi.call("agpu_event_elapsed_ms", s, e, C.byref(ms))
        ts.append(ms.value)
        capi.lib().agpu_event_destroy(s), capi.lib().agpu_event_destroy(e)
    return float(np.median(ts))


def synth(fa, fb, ia, ib, va, vb):
    capi.call("agpu_synth_f32", h, vp(fa), n, 1, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_synth_f32", h, vp(fb), n, 2, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_synth_i32", h, vp(ia), n, 3, 0, 1024)
    capi.call("agpu_synth_i32", h, vp(ib), n, 4, 0, 1024)
    capi.call("agpu_synth_bits", h, vp(va), n, 5, 0, C.c_double(0.9))
    capi.call("agpu_synth_bits", h, vp(vb), n, 6, 0, C.c_double(0.9))
    p.sync()


def step_times(bufs):
    fa, fb, fo, ia, ib, va, vb, ob, ov = bufs
    synth(fa, fb, ia, ib, va, vb)
    add = ms_of(lambda: capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(fa), vp(fb), vp(fo), n))
    eq = ms_of(lambda: capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(va), vp(vb), vp(ob), vp(ov), n))
    return {"add_ms": round(add, 4), "add_frac": round(12.0 * n / add / 1e6 / 8000, 4), "eq_ms": round(eq, 4),
            "eq_frac": round(8.5 * n / eq / 1e6 / 8000, 4)}


def host_api_times():
    """arrays as a caller builds them (separate buffers), ops through the host API: the op allocates its own outputs"""
    bufs = [dev.create_empty_buffer(4 * n) for _ in range(4)] + [dev.create_empty_buffer(nb) for _ in range(2)]
    fa, fb, ia, ib, va, vb = bufs
    synth(fa, fb, ia, ib, va, vb)
    A = ag.Float32ArrayGPU(fa, dev, n, None)
    B = ag.Float32ArrayGPU(fb, dev, n, None)
    IA = ag.Int32ArrayGPU(ia, dev, n, ag.NullBitBufferGpu(va, n, dev))
    IB = ag.Int32ArrayGPU(ib, dev, n, ag.NullBitBufferGpu(vb, n, dev))
    # per-launch HIP event pairs inside the library (agpu_pipeline_enable_timing): the op allocates on the host between
    # our own event records, which would put host time into them
    p.enable_timing(2)

    def kernel_ms(op, k=9):
        ts = []
        for i in range(k + 2):
            r = op()
            ns, _ = p.last_kernel_ns()
            p.sync()  # drops the pipeline's keep-alives: the output goes back to the pool, the next call reuses it
            del r
            if i >= 2:
                ts.append(ns / 1e6)
        return float(np.median(ts))

    t_add = kernel_ms(lambda: A.add_op(B, p))
    t_eq = kernel_ms(lambda: IA.eq_op(IB, p))
    p.enable_timing(0)
    return {"add_ms": round(t_add, 4), "add_frac": round(12.0 * n / t_add / 1e6 / 8000, 4), "eq_ms": round(t_eq, 4),
            "eq_frac": round(8.5 * n / t_eq / 1e6 / 8000, 4)}


out = {"rows": n, "reps": []}
sizes = [4 * n] * 3 + [4 * n] * 2 + [nb] * 4
for r in range(reps):
    rec = {}
    for label, arena in (("A_pool_blocks_arena_off", 0), ("B_pool_blocks_arena_on", 1)):
        capi.call("agpu_set_tuning", b"pool_arena", arena)
        bufs = [dev.create_empty_buffer(s) for s in sizes]  # plain agpu_malloc, in the order a, b, out, ia, ib, bitmaps
        rec[label] = step_times(bufs)
        del bufs
        rec[label.replace("pool_blocks", "host_api")] = host_api_times()
        p.sync()
        capi.call("agpu_device_trim", dev._handle)
    capi.call("agpu_set_tuning", b"pool_arena", 1)
    t1 = dev.create_table_buffers([4 * n] * 3)
    t2 = dev.create_table_buffers([4 * n] * 2 + [nb] * 4)
    rec["C_tables"] = step_times(t1 + t2)
    del t1, t2
    p.sync()
    capi.call("agpu_device_trim", dev._handle)
    out["reps"].append(rec)
    print(json.dumps(rec), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_pool_placement.json"), "w"), indent=1)
