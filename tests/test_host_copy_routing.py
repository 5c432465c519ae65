"""CPU: which path a host range takes on its way to / from HBM (arrow_cdata.hip agpu_internal_host_copy) — decided from the address alone,
so no GPU is needed.  Pages of malloc arenas (the brk heap AND the mmap'ed arenas glibc gives non-main threads: ADVICE r3) come and go
under a running process and must never be pinned by the runtime; only ranges PROVEN to be mappings of their own go to it directly."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import ctypes as C, json, sys, threading
sys.path.insert(0, {root!r})
from arrow_gpu_amd import _capi as capi
lib = capi.lib()
lib.agpu_internal_host_copy_path.restype = C.c_int32
lib.agpu_internal_host_copy_path.argtypes = [C.c_void_p, C.c_size_t]
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]
out = {{}}
n = 8 << 20
p = libc.malloc(n)                      # main thread, mmap threshold raised by the environment: the brk heap
out["main_thread_malloc"] = lib.agpu_internal_host_copy_path(p, n)
def worker():
    q = libc.malloc(n)                  # a non-main thread: glibc's mmap'ed thread arena
    C.memset(q, 1, n)
    out["thread_arena_malloc"] = lib.agpu_internal_host_copy_path(q, n)
    out["thread_arena_interior"] = lib.agpu_internal_host_copy_path(q + (1 << 20), 5 << 20)
    libc.free(q)
t = threading.Thread(target=worker); t.start(); t.join()
libc.free(p)
print(json.dumps(out))
"""


def _path(lib, ptr, nbytes):
    lib.agpu_internal_host_copy_path.restype = C.c_int32
    lib.agpu_internal_host_copy_path.argtypes = [C.c_void_p, C.c_size_t]
    return lib.agpu_internal_host_copy_path(ptr, nbytes)


def test_own_mappings_go_direct_and_everything_unproven_is_staged(tmp_path):
    import mmap

    from arrow_gpu_amd import _capi as capi

    lib = capi.lib()
    big = np.zeros(128 << 20, np.uint8)          # numpy → malloc → an mmapped chunk: ≥ 64 MiB cannot live in a malloc arena
    assert _path(lib, big.ctypes.data, big.nbytes) == 2
    assert _path(lib, big.ctypes.data, 4 << 20) == 0                         # ≤ 4 MiB: the bounce slot, wherever it lives
    assert _path(lib, big.ctypes.data + (16 << 20), 32 << 20) == 1           # deep inside a mapping, below 64 MiB: unproven → staged
    assert _path(lib, big.ctypes.data + (16 << 20), 100 << 20) == 2
    # 4–64 MiB: direct only when PROVEN to be a mapping of its own — a file mapping never merges with its neighbours
    f = open(tmp_path / "col.bin", "wb+")
    f.truncate(32 << 20)
    m = mmap.mmap(f.fileno(), 32 << 20)
    arr = np.frombuffer(m, np.uint8)
    assert _path(lib, arr.ctypes.data, arr.nbytes) == 2
    assert _path(lib, arr.ctypes.data + (1 << 20), 16 << 20) == 1            # an interior slice of it is not
    mid = np.zeros(32 << 20, np.uint8)           # an anonymous mmapped chunk may have merged with a neighbouring mapping: either is fine
    assert _path(lib, mid.ctypes.data, mid.nbytes) in (1, 2)
    del arr
    m.close()
    f.close()


def test_malloc_arena_ranges_are_staged_brk_heap_and_thread_arenas():
    env = dict(os.environ, MALLOC_MMAP_THRESHOLD_=str(1 << 30), MALLOC_TOP_PAD_="0")
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out == {"main_thread_malloc": 1, "thread_arena_malloc": 1, "thread_arena_interior": 1}, out
