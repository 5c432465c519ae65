"""GPU: host↔HBM copies never hand brk-heap pages to the runtime (the GPU memory fault of round 2) — the
routing is checked directly, and tools/probe/heap_copy_stress.py (copies out of and into a heap whose top keeps moving) runs
clean in a fresh process."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from arrow_gpu_amd import _capi as capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_heap_copy_stress_runs_clean():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "heap_copy_stress.py"), "--iters", "200"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "Memory access fault" not in r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ok"] and line["iterations"] == 200
    # the stress really exercised heap ranges above the bounce size, and none of them went to the runtime directly
    assert line["paths"]["chunk_engine_brk"] > 20 and line["paths"]["direct"] == 0, line


def test_routing_of_host_ranges(ag):
    lib = capi.lib()
    lib.agpu_internal_host_copy_path.restype = C.c_int32
    lib.agpu_internal_host_copy_path.argtypes = [C.c_void_p, C.c_size_t]
    big = np.zeros(64 << 20, np.uint8)  # numpy → malloc → far above any mmap threshold: its own mapping
    assert lib.agpu_internal_host_copy_path(big.ctypes.data, big.nbytes) == 2
    assert lib.agpu_internal_host_copy_path(big.ctypes.data, 4 << 20) == 0  # ≤ 4 MiB: the bounce slot, wherever it lives
    # an upload + download of each kind round-trips
    dev = ag.GPU_DEVICE()
    p = ag.ArrowComputePipeline(dev, "routes")
    for nbytes in (1, 4 << 20, (4 << 20) + 1, 48 << 20):
        src = np.random.default_rng(nbytes).integers(0, 256, nbytes, dtype=np.uint8)
        buf = dev.create_empty_buffer(nbytes)
        capi.call("agpu_upload", p._handle, C.c_void_p(buf.ptr), C.c_void_p(src.ctypes.data), nbytes)
        dst = np.zeros(nbytes, np.uint8)
        capi.call("agpu_download", p._handle, C.c_void_p(dst.ctypes.data), C.c_void_p(buf.ptr), nbytes)
        assert np.array_equal(src, dst)
