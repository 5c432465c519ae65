#!/usr/bin/env python3
"""DEV TOOL (GPU box): which path do host ranges of 4–64 MiB take through agpu_upload / agpu_download by ALLOCATOR, and what
does it cost?  Since round 4 such a range goes straight to the runtime only when it is proven to be a mapping of its own (glibc thread
arenas are the hazard, arrow_cdata.hip host_range_is_own_mapping); buffers out of pyarrow's mimalloc / jemalloc pools, numpy views at
an offset and blocks that merged with a neighbour fall onto the page-locked chunk engine.  Route: 0 bounce slot, 1 chunk engine, 2 direct.
    python tools/probe/host_copy_routes.py > gpurun_out/r05_host_copy_routes.json"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import pyarrow as pa

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice  # noqa: E402

lib = capi.lib()
lib.agpu_internal_host_copy_path.restype = C.c_int32
lib.agpu_internal_host_copy_path.argtypes = [C.c_void_p, C.c_size_t]
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "routes")
dbuf = dev.create_empty_buffer(64 << 20)
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]


def rate(ptr, nbytes, up):
    name = "agpu_upload" if up else "agpu_download"
    args = (p._handle, C.c_void_p(dbuf.ptr), C.c_void_p(ptr), nbytes) if up else (p._handle, C.c_void_p(ptr), C.c_void_p(dbuf.ptr), nbytes)
    for _ in range(2):
        capi.call(name, *args)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        capi.call(name, *args)
        ts.append(time.perf_counter() - t0)
    return nbytes / sorted(ts)[len(ts) // 2] / 1e9


out = []
keep = []
for mib in (4.5, 8, 16, 32, 48, 64):
    nbytes = int(mib * (1 << 20))
    cases = {}
    a = np.empty(nbytes, np.uint8); a[:] = 1
    cases["numpy_fresh"] = (a.ctypes.data, a)
    b = np.empty(nbytes + (1 << 20), np.uint8); b[:] = 1
    cases["numpy_view_at_1MiB"] = (b.ctypes.data + (1 << 20), b)
    for pool_name in ("mimalloc", "jemalloc", "system"):
        pool = getattr(pa, pool_name + "_memory_pool")()
        buf = pa.allocate_buffer(nbytes, memory_pool=pool)
        np.frombuffer(buf, np.uint8)[:] = 1
        cases["pyarrow_" + pool_name] = (buf.address, buf)
    arr = pa.array(np.arange(nbytes // 4, dtype=np.int32))  # what an import through the C Data Interface hands over
    cases["pyarrow_array_from_numpy"] = (arr.buffers()[1].address, arr)
    m = libc.malloc(nbytes)
    C.memset(m, 1, nbytes)
    cases["glibc_malloc_main_thread"] = (m, None)
    for name, (ptr, owner) in cases.items():
        keep.append(owner)
        row = {"MiB": mib, "allocator": name, "route": lib.agpu_internal_host_copy_path(ptr, nbytes),
               "upload_GBps": round(rate(ptr, nbytes, True), 1), "download_GBps": round(rate(ptr, nbytes, False), 1)}
        out.append(row)
        print(row, file=sys.stderr)
    libc.free(m)
print(json.dumps(out, indent=1))
