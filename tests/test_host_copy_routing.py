"""CPU: which path a host range takes on its way to / from HBM (arrow_cdata.hip agpu_internal_host_copy) — decided from the address alone,
so no GPU is needed.  Pages of malloc arenas (the brk heap AND the mmap'ed arenas glibc gives non-main threads: ADVICE r3) come and go
under a running process and must never be pinned by the runtime.  Round 5: the test is "is this range inside a glibc thread arena" (the
heap_info header at the 64 MiB boundary below it), not "is it proven to be a mapping of its own" — the round-4 proof failed for fresh numpy
arrays, numpy views and every pyarrow pool, which then moved at a third of the link's rate (profiles/r05_host_copy_routes*.json)."""
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


def test_everything_that_is_not_a_malloc_arena_goes_direct(tmp_path):
    import mmap

    import pyarrow as pa

    from arrow_gpu_amd import _capi as capi

    lib = capi.lib()
    big = np.zeros(128 << 20, np.uint8)          # numpy → malloc → an mmapped chunk: ≥ 64 MiB cannot live in a malloc arena
    assert _path(lib, big.ctypes.data, big.nbytes) == 2
    assert _path(lib, big.ctypes.data, 4 << 20) == 0                         # ≤ 4 MiB: the bounce slot, wherever it lives
    assert _path(lib, big.ctypes.data + (16 << 20), 32 << 20) == 2           # deep inside a mapping: no heap_info below it
    assert _path(lib, big.ctypes.data + (16 << 20), 100 << 20) == 2
    f = open(tmp_path / "col.bin", "wb+")
    f.truncate(32 << 20)
    m = mmap.mmap(f.fileno(), 32 << 20)
    arr = np.frombuffer(m, np.uint8)
    assert _path(lib, arr.ctypes.data, arr.nbytes) == 2
    assert _path(lib, arr.ctypes.data + (1 << 20), 16 << 20) == 2            # an interior slice of a file mapping
    # what real callers hand over (main thread): fresh numpy arrays of every size class, views at an offset, every pyarrow pool,
    # an Arrow array built from numpy — the round-4 rule staged ALL of these
    keep = []
    for mib in (5, 8, 16, 32, 48, 63):
        n = mib << 20
        a = np.ones(n, np.uint8)
        b = np.ones(n + (1 << 20), np.uint8)
        keep += [a, b]
        assert _path(lib, a.ctypes.data, n) == 2, ("numpy", mib)
        assert _path(lib, b.ctypes.data + (1 << 20), n) == 2, ("numpy view", mib)
        for pool in ("mimalloc_memory_pool", "jemalloc_memory_pool", "system_memory_pool"):
            try:
                buf = pa.allocate_buffer(n, memory_pool=getattr(pa, pool)())
            except Exception:  # a pool this pyarrow build does not have
                continue
            keep.append(buf)
            assert _path(lib, buf.address, n) == 2, (pool, mib)
        arr2 = pa.array(np.arange(n // 4, dtype=np.int32))
        keep.append(arr2)
        assert _path(lib, arr2.buffers()[1].address, n) == 2, ("pa.array", mib)
    del arr
    m.close()
    f.close()


def test_a_lookalike_header_is_only_slower_never_wrong():
    """memory that happens to carry a plausible heap_info at a 64 MiB boundary is staged (the safe path) — and memory whose first words are
    anything else is not mistaken for an arena"""
    from arrow_gpu_amd import _capi as capi

    lib = capi.lib()
    raw = np.zeros((192 << 20) // 8, np.uint64)                      # contains at least two 64 MiB-aligned addresses
    base = raw.ctypes.data
    A = (base + (64 << 20) - 1) & ~((64 << 20) - 1)
    w = (A - base) // 8
    assert _path(lib, A + (1 << 20), 8 << 20) == 2                   # zeros at A: not a header
    raw[w:w + 4] = [A + 48, 0, 16 << 20, 16 << 20]                   # first heap of an arena: ar_ptr right behind the header, no prev
    assert _path(lib, A + (1 << 20), 8 << 20) == 1
    assert _path(lib, A + (60 << 20), 8 << 20) == 2                  # crosses the next 64 MiB boundary: no single heap holds it
    raw[w:w + 4] = [A + (128 << 20) + 48, A + (128 << 20), 8 << 20, 8 << 20]   # a later heap: prev and ar_ptr point at another boundary
    assert _path(lib, A + (1 << 20), 8 << 20) == 1
    raw[w:w + 4] = [A + 48, 0, (16 << 20) + 8, 16 << 20]             # size not a page multiple / larger than mprotect_size: not a header
    assert _path(lib, A + (1 << 20), 8 << 20) == 2
    raw[w:w + 4] = [12345, 0, 16 << 20, 16 << 20]                    # ar_ptr not near a 64 MiB boundary
    assert _path(lib, A + (1 << 20), 8 << 20) == 2


def test_malloc_arena_ranges_are_staged_brk_heap_and_thread_arenas():
    env = dict(os.environ, MALLOC_MMAP_THRESHOLD_=str(1 << 30), MALLOC_TOP_PAD_="0")
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out == {"main_thread_malloc": 1, "thread_arena_malloc": 1, "thread_arena_interior": 1}, out
