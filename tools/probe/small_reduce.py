#!/usr/bin/env python3
"""DEV TOOL (GPU box): the reference's own criterion shapes for the reductions (crates/benchmarks/benches/compare_sum.rs:17-40: u32 sum at 1 Mi / 10 Mi
rows) — GPU time between events and the host API call + sync, best and median of 200."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import arrow_gpu_amd as ag
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
if os.environ.get("SYNC_SPIN"): capi.call("agpu_set_tuning", b"sync_spin", int(os.environ["SYNC_SPIN"]))   # < 0: the waits without the mailbox (R5.10)
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "sr"); q = CmpQuery(dev); h = p._handle
for n in (1 << 20, 10 << 20, 64 << 20):
    a = dev.create_gpu_buffer_with_data(np.arange(n, dtype=np.uint32))
    r = dev.create_empty_buffer(16)
    for op, dt, name in ((capi.RED_SUM, capi.U32, "u32 sum"), (capi.RED_MAX, capi.F32, "f32 max")):
        f = lambda: capi.call("agpu_reduce", h, op, dt, C.c_void_p(a.ptr), None, n, C.c_void_p(r.ptr))
        for _ in range(20): f()
        p.sync(); ts = []
        for _ in range(200):
            q.begin(p); f(); q.end(p); ts.append(q.wait_for_results() * 1e3)
        ws = []
        for _ in range(200):
            t0 = time.perf_counter(); f(); p.sync(); ws.append((time.perf_counter() - t0) * 1e6)
        print(f"{name} {n >> 20:3d} Mi rows: events best {min(ts):.1f} median {sorted(ts)[100]:.1f} us; call + sync best {min(ws):.1f} median {sorted(ws)[100]:.1f} us", flush=True)
    arr = ag.UInt32ArrayGPU(a, dev, n, None)
    ws = []
    for _ in range(200):
        t0 = time.perf_counter(); s = arr.sum(); dev.sync(); ws.append((time.perf_counter() - t0) * 1e6)
    print(f"host API UInt32ArrayGPU.sum() {n >> 20} Mi rows: best {min(ws):.1f} median {sorted(ws)[100]:.1f} us", flush=True)
    ws = []
    for _ in range(200):
        t0 = time.perf_counter(); f(); v = dev.retrive_data(r, 4, pipeline=p); ws.append((time.perf_counter() - t0) * 1e6)
    print(f"agpu_reduce + download of the scalar {n >> 20} Mi rows: best {min(ws):.1f} median {sorted(ws)[100]:.1f} us (value {int(v.view(np.uint32)[0]) if False else v.view(np.uint32)[0]})", flush=True)
    ws = []
    for _ in range(200):
        t0 = time.perf_counter(); v = arr.sum().values(); ws.append((time.perf_counter() - t0) * 1e6)
    print(f"host API UInt32ArrayGPU.sum().values() {n >> 20} Mi rows: best {min(ws):.1f} median {sorted(ws)[100]:.1f} us ({v})", flush=True)
