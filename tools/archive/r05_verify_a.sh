#!/bin/bash
# Round 5, GPU box: the GPU suite on the new host-copy routing + the route probe again ("after")
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu_b.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_pytest_gpu_b.log
timeout 600 python tools/probe/host_copy_routes.py > gpurun_out/r05_host_copy_routes_after.json 2> gpurun_out/r05_host_copy_routes_after.txt; echo "routes rc=$?"; cat gpurun_out/r05_host_copy_routes_after.txt
# the staging engine itself on an arena range (a non-main thread's malloc), 8 / 32 MiB
python - <<'PY'
import ctypes as C, threading, time, sys, os
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
lib = capi.lib(); lib.agpu_internal_host_copy_path.restype = C.c_int32; lib.agpu_internal_host_copy_path.argtypes = [C.c_void_p, C.c_size_t]
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "arena"); d = dev.create_empty_buffer(64 << 20)
libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]
libc.mallopt.argtypes = [C.c_int, C.c_int]; libc.mallopt(-3, 1 << 30)  # M_MMAP_THRESHOLD: keep big blocks inside the arena
def worker():
    for mib in (8, 32):
        n = mib << 20; q = libc.malloc(n); C.memset(q, 1, n)
        for up in (True, False):
            name = "agpu_upload" if up else "agpu_download"
            args = (p._handle, C.c_void_p(d.ptr), C.c_void_p(q), n) if up else (p._handle, C.c_void_p(q), C.c_void_p(d.ptr), n)
            for _ in range(2): capi.call(name, *args)
            ts = []
            for _ in range(9):
                t0 = time.perf_counter(); capi.call(name, *args); ts.append(time.perf_counter() - t0)
            print({"MiB": mib, "allocator": "glibc_thread_arena", "route": lib.agpu_internal_host_copy_path(q, n), name: round(n / sorted(ts)[4] / 1e9, 1)})
        libc.free(q)
t = threading.Thread(target=worker); t.start(); t.join()
PY
