"""GPU: the runtime's resource pools (include/arrow_gpu.h "Resource pools").  The reference's default API builds a
pipeline and an output buffer per op; on ROCm a stream costs milliseconds to create and hipFree synchronises the
device, so idle streams and freed blocks are recycled — without letting a recycled block be handed out while work
that was queued at free time may still touch it."""
import ctypes as C
import time

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi

pytestmark = pytest.mark.gpu


def pool_info(dev):
    b, k, s = C.c_uint64(), C.c_uint64(), C.c_uint64()
    capi.call("agpu_device_pool_info", dev._handle, C.byref(b), C.byref(k), C.byref(s))
    return b.value, k.value, s.value


def small_pool_info(dev):
    b, f, l = C.c_uint64(), C.c_uint64(), C.c_uint64()
    capi.call("agpu_device_small_pool_info", dev._handle, C.byref(b), C.byref(f), C.byref(l))
    return b.value, f.value, l.value


def test_blocks_are_recycled_and_trim_releases_them(ag):
    import gc

    dev = ag.GPU_DEVICE()
    gc.collect()  # buffers of earlier tests that sit in reference cycles would otherwise land in the pool mid-test
    capi.call("agpu_device_trim", dev._handle)
    assert pool_info(dev)[:2] == (0, 0)
    a = dev.create_empty_buffer(64 << 20)
    ptr = a.ptr
    del a
    cached, blocks, _ = pool_info(dev)
    assert blocks == 1 and cached == 64 << 20
    b = dev.create_empty_buffer((64 << 20) - 4096)  # rounds to the same 2 MiB granule count → same block
    assert b.ptr == ptr and pool_info(dev)[:2] == (0, 0)
    slab0, free0, live0 = small_pool_info(dev)
    small = dev.create_empty_buffer(4096)  # below 1 MiB: a 4 KiB block of the slab pool, not the large-block cache
    assert pool_info(dev)[:2] == (0, 0)
    slab1, free1, live1 = small_pool_info(dev)
    assert live1 == live0 + 1 and slab1 >= max(slab0, 2 << 20)
    del small
    assert small_pool_info(dev)[2] == live0
    again = [dev.create_empty_buffer(4000) for _ in range(3)]  # same size class; no hipMalloc, no hipFree
    assert small_pool_info(dev)[0] == slab1 and len({b.ptr for b in again}) == 3
    del again
    del b
    capi.call("agpu_device_trim", dev._handle)
    assert pool_info(dev)[:2] == (0, 0)
    capi.call("agpu_set_tuning", b"mem_pool", 0)
    try:
        c = dev.create_empty_buffer(8 << 20)
        del c
        assert pool_info(dev)[:2] == (0, 0)
    finally:
        capi.call("agpu_set_tuning", b"mem_pool", 1)


def test_recycled_block_waits_for_work_queued_at_free_time(ag):
    """Free the output of a long chain while the chain is still running, re-allocate (same block comes back) and
    overwrite it from ANOTHER pipeline: the late kernels of the first chain must not clobber the new contents."""
    dev = ag.GPU_DEVICE()
    n = 1 << 26
    p1 = ag.ArrowComputePipeline(dev, "producer")
    p2 = ag.ArrowComputePipeline(dev, "consumer")
    vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
    a = dev.create_empty_buffer(4 * n)
    capi.call("agpu_synth_f32", p1._handle, vp(a), n, 1, 0, C.c_float(-1), C.c_float(1))
    p1.sync()
    import gc

    gc.collect()
    capi.call("agpu_device_trim", dev._handle)
    for _ in range(3):
        out = dev.create_empty_buffer(4 * n)
        ptr = out.ptr
        for _ in range(40):  # ~40 × 0.13 ms of queued work writing `out`
            capi.call("agpu_unary", p1._handle, capi.UN_SIN, capi.F32, vp(a), vp(out), n)
        del out  # freed while p1 is still busy
        again = dev.create_empty_buffer(4 * n)
        assert again.ptr == ptr
        capi.call("agpu_broadcast", p2._handle, capi.F32, int(np.float32(7.0).view(np.uint32)), vp(again), n)
        p2.sync()
        p1.sync()
        got = dev.retrive_data(again, 4 * n, pipeline=p2).view(np.float32)
        assert np.all(got == np.float32(7.0))
        del again


def test_default_api_ops_do_not_pay_for_stream_creation(ag):
    dev = ag.GPU_DEVICE()
    n = 1 << 20
    a = ag.Int32ArrayGPU.from_slice(O.synth_i32(n, 1, 0, 0), dev)
    b = ag.Int32ArrayGPU.from_slice(O.synth_i32(n, 2, 0, 0), dev)
    for _ in range(3):
        a.add(b)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        c = a.add(b)  # new pipeline + new output buffer per call, like the reference's `add`
        ts.append(time.perf_counter() - t0)
        del c
    med_ms = float(np.median(ts)) * 1e3
    print(f"a.add(b), 1 Mi rows, default API: {med_ms:.3f} ms per call, idle streams {pool_info(dev)[2]}")
    assert med_ms < 2.0  # 7 ms of hipStreamCreate/Destroy per call without the pool
    assert np.array_equal(a.add(b).raw_values(), O.binary(O.OP_ADD, O.I32, a.raw_values(), b.raw_values()))


def test_pools_under_concurrent_threads(ag):
    """8 host threads hammer one device: new pipeline + new buffers per op (the reference's default API shape), results
    checked every time.  ctypes releases the GIL inside the C calls, so the pools' locking is really exercised."""
    import threading

    dev = ag.GPU_DEVICE()
    n = 300_000  # 1.2 MB columns: pooled blocks
    errors = []

    def worker(tid):
        try:
            rng = np.random.default_rng(tid)
            a_host = rng.integers(-1000, 1000, n).astype(np.int32)
            b_host = rng.integers(-1000, 1000, n).astype(np.int32)
            a = ag.Int32ArrayGPU.from_slice(a_host, dev)
            b = ag.Int32ArrayGPU.from_slice(b_host, dev)
            for it in range(40):
                c = a.add(b)                      # own pipeline, own output, freed right after
                m = c.gt(a)
                if it % 8 == 0:
                    assert np.array_equal(c.raw_values(), a_host + b_host)
                    assert np.array_equal(m.raw_values(), (a_host + b_host) > a_host)
                del c, m
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_table_buffers_come_from_one_block_with_hash_colours_and_free_independently():
    """agpu_malloc_table: columns ≥ 1 GiB a multiple of 512 MiB apart + 0 / 8 / 4 / 12 KiB by index; usable and freeable
    one by one in any order; the block returns to the pool with the last column"""
    import ctypes as C

    import numpy as np

    from arrow_gpu_amd import _capi as capi
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    dev = GpuDevice(0)
    capi.call("agpu_device_trim", dev._handle)
    cached0, blocks0 = C.c_uint64(), C.c_uint64()
    capi.call("agpu_device_pool_info", dev._handle, C.byref(cached0), C.byref(blocks0), None)
    big, small = (1 << 30) + 12345, 5_000_000
    bufs = dev.create_table_buffers([big, big, small, big, small])
    base = bufs[0].ptr
    offs = [b.ptr - base for b in bufs]
    # big columns (index 0, 1, 3) first, 1.5 GiB strides (1 GiB + 12345 B + colour room → three 512 MiB units), colours 0 / 8 / 4 KiB;
    # the small ones (index 2, 4: < 32 MiB) packed back to back behind them at 256-byte alignment, no colour
    unit = 512 << 20
    assert offs == [0, 3 * unit + 8192, 9 * unit, 6 * unit + 4096, 9 * unit + (small + 255) // 256 * 256]
    p = ArrowComputePipeline(dev, "table")
    rng = np.random.default_rng(0)
    n = 1_000_000
    a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    capi.call("agpu_upload", p._handle, C.c_void_p(bufs[0].ptr), C.c_void_p(a.ctypes.data), a.nbytes)
    capi.call("agpu_upload", p._handle, C.c_void_p(bufs[1].ptr), C.c_void_p(b.ctypes.data), b.nbytes)
    capi.call("agpu_binary", p._handle, capi.OP_ADD, capi.F32, C.c_void_p(bufs[0].ptr), C.c_void_p(bufs[1].ptr), C.c_void_p(bufs[2].ptr), n)
    got = dev.retrive_data(bufs[2], 4 * n, pipeline=p).view(np.float32)
    assert np.array_equal(got, a + b)
    for k in (3, 0, 4, 1):  # any order; the block stays live while one column does
        bufs[k] = None
    cached, blocks = C.c_uint64(), C.c_uint64()
    capi.call("agpu_device_pool_info", dev._handle, C.byref(cached), C.byref(blocks), None)
    assert blocks.value == blocks0.value
    got = dev.retrive_data(bufs[2], 4 * n, pipeline=p).view(np.float32)
    assert np.array_equal(got, a + b)
    bufs[2] = None
    capi.call("agpu_device_pool_info", dev._handle, C.byref(cached), C.byref(blocks), None)
    assert blocks.value == blocks0.value + 1 and cached.value >= 3 * big  # the whole block is back in the pool's cache
    capi.call("agpu_device_trim", dev._handle)
    # medium columns (32 MiB … 1 GiB) keep 2 MiB granules and the colours; a record batch of small columns and bitmaps costs
    # about its own size — not a 2 MiB granule per 8 KiB bitmap
    med, gran = 100_000_000, 2 << 20
    bufs = dev.create_table_buffers([med, 8192, med, 262144, 8192])
    offs = [b.ptr - bufs[0].ptr for b in bufs]
    step = (med + 16384 + gran - 1) // gran * gran
    assert offs == [0, 2 * step, step + 8192, 2 * step + 8192, 2 * step + 8192 + 262144]
    del bufs
    free0 = dev.mem_info()[0]
    tables = [dev.create_table_buffers([262144, 262144, 8192, 8192]) for _ in range(2000)]  # 2 000 batches of 64 Ki rows
    used = free0 - dev.mem_info()[0]
    # 528 KiB per batch → the 1 MiB slab class: 2 GiB for the 2 000 batches (it was 4 × 2 MiB per batch = 16 GiB)
    assert used <= 2000 * (1 << 20) + (64 << 20), used
    ptrs = sorted(b.ptr for t in tables for b in t)
    assert all(q - p_ >= 8192 for p_, q in zip(ptrs, ptrs[1:]))  # no overlap
    del tables
    capi.call("agpu_device_trim", dev._handle)


def test_big_pool_blocks_come_out_of_placed_arenas():
    """pool placement (runtime.hip): agpu_malloc blocks of ≥ 1 GiB are carved out of one arena at multiples of 512 MiB plus a
    rotating colour of 0 / 8 / 4 / 12 KiB — the layout agpu_malloc_table gives one table, for buffers allocated one by one;
    agpu_malloc_like places an op's output against its inputs; blocks never overlap; trim gives the arena back"""
    import ctypes as C

    import numpy as np

    from arrow_gpu_amd import _capi as capi
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "arena")
    capi.call("agpu_device_trim", dev._handle)
    free0 = dev.mem_info()[0]
    unit, big = 512 << 20, (1 << 30) + 4096
    bufs = [dev.create_empty_buffer(big) for _ in range(5)]
    base = min(b.ptr for b in bufs) & ~0x3FFF
    offs = [b.ptr - base for b in bufs]
    assert [o % unit for o in offs] == [0, 8192, 4096, 12288, 0]            # the rotating colour
    assert sorted(o // unit for o in offs) == [0, 3, 6, 9, 12]              # (1 GiB + 4 KiB + colour room) → three units each
    assert free0 - dev.mem_info()[0] >= 15 * unit                           # one arena holds them all
    # every block is usable to its last byte without touching its neighbours
    pat = [np.full(1 << 16, 17 * (k + 1), np.uint8) for k in range(5)]
    for k, b in enumerate(bufs):
        capi.call("agpu_memset", p._handle, C.c_void_p(b.ptr), 17 * (k + 1), big)
    for k, b in enumerate(bufs):
        for off in (0, big - (1 << 16)):
            got = np.empty(1 << 16, np.uint8)
            capi.call("agpu_download", p._handle, C.c_void_p(got.ctypes.data), C.c_void_p(b.ptr + off), 1 << 16)
            assert np.array_equal(got, pat[k]), (k, off)
    # an output placed against its inputs: colours 0 (bufs[0]) and 4 KiB (bufs[2]) → bit 13 must differ from both
    out = dev.create_empty_buffer(big, like=(bufs[0], bufs[2]))
    assert (out.ptr - base) % unit in (8192, 12288)
    out2 = dev.create_empty_buffer(big, like=(bufs[1],))                    # neighbour at 8 KiB → 0 or 4 KiB
    assert (out2.ptr - base) % unit in (0, 4096)
    # a freed block is reused (from the cache) and re-coloured for its new neighbours
    units_out = (out.ptr - base) // unit
    del out
    again = dev.create_empty_buffer(big, like=(bufs[1], bufs[3]))           # neighbours at 8 and 12 KiB → 0 or 4 KiB
    assert (again.ptr - base) // unit == units_out and (again.ptr - base) % unit in (0, 4096)
    del bufs, out2, again, b  # (`b`: the loop variable above still names the last block)
    p.sync()
    capi.call("agpu_device_trim", dev._handle)
    assert dev.mem_info()[0] >= free0 - (64 << 20)                          # the arena went back to the driver
    # switched off: one hipMalloc per block, as before
    capi.call("agpu_set_tuning", b"pool_arena", 0)
    try:
        a, b = dev.create_empty_buffer(big), dev.create_empty_buffer(big)
        assert free0 - dev.mem_info()[0] < 3 * (1 << 30)
        del a, b
    finally:
        capi.call("agpu_set_tuning", b"pool_arena", 1)
        capi.call("agpu_device_trim", dev._handle)


def test_cached_arena_blocks_and_table_blocks_do_not_stand_in_for_each_other():
    """Round 6: blocks of >= 1 GiB come in two kinds.  A table's own block never takes a cached ARENA block (a table
    carved from an arena loses 4-6 points on the compare) and — the case that cost the host API 6-7 points on the add — an ordinary
    placed request never takes a cached PLAIN block (two freed one-column tables as the inputs of an add: 0.78 of the roof).
    And ADVICE r3 still holds where a cached arena block does serve a request without colour room (tuning pool_arena = 0 after arena
    blocks were cached): it is handed out at colour 0 — moved up by 4-12 KiB its zero fill ran into the next arena block."""
    import ctypes as C

    import numpy as np

    from arrow_gpu_amd import _capi as capi
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "arena-table")
    capi.call("agpu_device_trim", dev._handle)
    unit, big = 512 << 20, (1 << 30) + 4096
    try:
        for attempt in range(4):  # the rotating colour: every starting phase of it
            blocks = [dev.create_empty_buffer(big) for _ in range(3)]             # units 0-2, 3-5, 6-8 of one arena
            base = min(b.ptr for b in blocks) & ~0x3FFF
            order = sorted(range(3), key=lambda k: blocks[k].ptr)
            mid, last = blocks[order[1]], blocks[order[2]]
            assert (last.ptr - base) // unit == 6 and (mid.ptr - base) // unit == 3
            capi.call("agpu_memset", p._handle, C.c_void_p(last.ptr), 0xA5, 1 << 20)
            p.sync()
            mid_unit = mid.ptr & ~0x3FFF
            blocks[order[1]] = None
            del mid                                                                # → the size-keyed cache, 3 units
            # (1) a table of exactly that size gets a block of its own, outside the arena
            table = dev.create_table_buffers([3 * unit - 16384], zero_fill=True)
            assert not (base <= table[0].ptr < base + 64 * unit), (attempt, hex(table[0].ptr), hex(base))
            tptr = table[0].ptr
            del table                                                              # → the cache: a PLAIN block of 3 units
            # (2) an ordinary placed request of that size takes the cached ARENA block (first in the cache or not), never the plain one
            again = dev.create_empty_buffer(big)
            assert again.ptr & ~0x3FFF == mid_unit, (attempt, hex(again.ptr), hex(mid_unit), hex(tptr))
            del again                                                              # back into the cache BEHIND the plain block of the same size
            again = dev.create_empty_buffer(big)                                   # … which the lookup has to step over
            assert again.ptr & ~0x3FFF == mid_unit, (attempt, hex(again.ptr), hex(mid_unit), hex(tptr))
            del again
            # (3) pool_arena = 0: kinds no longer matter; whichever block comes back, an arena block comes at colour 0 and keeps to its units
            capi.call("agpu_set_tuning", b"pool_arena", 0)
            got_blocks = [dev.create_empty_buffer(3 * unit, zero_fill=True) for _ in range(2)]   # 3 units exactly, no colour room
            ptrs = sorted(b.ptr for b in got_blocks)
            assert ptrs == sorted([mid_unit, tptr]), (attempt, [hex(x) for x in ptrs], hex(mid_unit), hex(tptr))
            got = np.empty(1 << 20, np.uint8)
            capi.call("agpu_download", p._handle, C.c_void_p(got.ctypes.data), C.c_void_p(last.ptr), 1 << 20)
            assert (got == 0xA5).all(), attempt                                    # the neighbour is intact
            capi.call("agpu_memset", p._handle, C.c_void_p(mid_unit), 0x11, 3 * unit)  # usable to its last byte
            capi.call("agpu_download", p._handle, C.c_void_p(got.ctypes.data), C.c_void_p(last.ptr), 1 << 20)
            assert (got == 0xA5).all(), attempt
            capi.call("agpu_set_tuning", b"pool_arena", 1)
            del got_blocks, blocks, last
            p.sync()
            capi.call("agpu_device_trim", dev._handle)
    finally:
        capi.call("agpu_set_tuning", b"pool_arena", 1)
