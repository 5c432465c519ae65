#!/usr/bin/env python3
"""DEV TOOL (GPU box): the device-level wait with ONE, TWO and THREE streams outstanding (a reduction of 1 Mi rows on each), with and without a
scalar travelling along — median µs of 300.  SYNC_SPIN=-1: hipDeviceSynchronize (+ hipMemcpy) every time."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
if os.environ.get("SYNC_SPIN"): capi.call("agpu_set_tuning", b"sync_spin", int(os.environ["SYNC_SPIN"]))
dev = GpuDevice(0)
n = 1 << 20
ps = [ArrowComputePipeline(dev, f"s{k}") for k in range(3)]
a = dev.create_gpu_buffer_with_data(np.arange(n, dtype=np.uint32))
outs = [dev.create_empty_buffer(16) for _ in ps]
def med(fn, k=300):
    for _ in range(30): fn()
    ts = []
    for _ in range(k):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e6)
    return sorted(ts)[k // 2]
for active in (1, 2, 3):
    def work():
        for k in range(active):
            capi.call("agpu_reduce", ps[k]._handle, capi.RED_SUM, capi.U32, C.c_void_p(a.ptr), None, n, C.c_void_p(outs[k].ptr))
    def sync_only():
        work(); dev.sync()
    def with_scalar():
        work(); return dev.retrive_data(outs[active - 1], 4)
    print(f"{active} stream(s): reductions + device sync {med(sync_only):.1f} us; + the last result on the host {med(with_scalar):.1f} us", flush=True)
