"""GPU: runtime semantics of the C ABI that the reference gets for free from its single wgpu queue and wgpu's allocator
(SURVEY §8b): cross-pipeline ordering at `finish`, the small-block pool, per-pipeline error words and tuning, the
per-launch profiling hooks, and the unaligned f32 Sum."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi

pytestmark = pytest.mark.gpu

N_BIG = 100_000_000


def vp(b, off=0):
    return C.c_void_p(b.ptr + off)


def f32_bits(x):
    return int(np.float32(x).view(np.uint32))


def slow_chain(p, buf, one, n, steps):
    """`steps` in-place `buf += 1` passes queued on p: the final value appears only when the LAST kernel has run."""
    for _ in range(steps):
        capi.call("agpu_scalar", p._handle, capi.OP_ADD, capi.F32, vp(buf), vp(one), vp(buf), n)


# ------------------------------------------------------------------ cross-pipeline ordering (the reference: one queue)
def test_clone_after_op_on_another_pipeline_sees_the_result(ag):
    """x = (long chain on p1); p1.finish(); y = x.clone_array() — the clone runs on the thread's default pipeline, a
    different HIP stream; `finish` must order it behind the chain [ref: every submit is ordered on the one wgpu queue]."""
    dev = ag.GPU_DEVICE()
    n, steps = N_BIG, 24
    p1 = ag.ArrowComputePipeline(dev, "producer")
    x = ag.Float32ArrayGPU.broadcast_op(0.0, n, p1)
    one = dev.create_scalar_buffer(1.0, np.float32)
    slow_chain(p1, x.data, one, n, steps)
    p1.finish()
    y = x.clone_array()  # no sync anywhere in between
    got = y.raw_values()
    assert got[0] == steps and got[-1] == steps and np.all(got[:: 9973] == steps)


def test_default_op_after_finished_pipeline_is_ordered(ag):
    """y = a.add_op(b, p1) … p1.finish(); z = y.mul(c) — `mul` gets a fresh pooled stream while p1 is still alive."""
    dev = ag.GPU_DEVICE()
    n, steps = N_BIG, 24
    p1 = ag.ArrowComputePipeline(dev, "producer")
    y = ag.Float32ArrayGPU.broadcast_op(0.0, n, p1)
    one = dev.create_scalar_buffer(1.0, np.float32)
    slow_chain(p1, y.data, one, n, steps)
    p1.finish()
    c = ag.Float32ArrayGPU.broadcast(2.0, n, dev)
    z = y.mul(c)  # own pipeline, own stream
    got = z.raw_values()
    assert np.all(got[:: 9973] == 2.0 * steps) and got[-1] == 2.0 * steps
    del p1


def test_destroy_publishes_like_finish(ag):
    dev = ag.GPU_DEVICE()
    n, steps = N_BIG, 16
    p1 = ag.ArrowComputePipeline(dev, "producer")
    y = ag.Float32ArrayGPU.broadcast_op(0.0, n, p1)
    one = dev.create_scalar_buffer(1.0, np.float32)
    slow_chain(p1, y.data, one, n, steps)
    hold = ag.ArrowComputePipeline(dev, "takes another stream")  # so the next default op cannot reuse p1's stream
    del p1  # destroyed without finish / sync
    z = y.add_scalar(ag.Float32ArrayGPU.from_slice(np.array([1.0], np.float32), dev)) if hasattr(y, "add_scalar") else None
    if z is None:
        pytest.skip("no add_scalar on this host layer")
    got = z.raw_values()
    assert np.all(got[:: 9973] == steps + 1)
    del hold


def test_chains_from_several_threads_stay_ordered(ag):
    """c = a.add(b); m = c.gt(a) with 4 host threads and 5e7-row columns: the idle-stream pool is shared, so the two ops
    of one thread regularly land on different streams — results must still be right."""
    dev = ag.GPU_DEVICE()
    n = 50_000_000
    errors = []

    def worker(tid):
        try:
            a = ag.Float32ArrayGPU.broadcast(float(tid + 1), n, dev)
            b = ag.Float32ArrayGPU.broadcast(0.5, n, dev)
            for _ in range(6):
                c = a.add(b)
                m = c.gt(a)
                d = c.sub(b)
                vals = d.raw_values()
                assert vals[0] == tid + 1 and vals[-1] == tid + 1 and np.all(vals[:: 99991] == tid + 1)
                bits = m.raw_bytes()
                assert bits[0] == 0xFF and bits[(n // 8) - 1] == 0xFF
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_wait_pipeline_orders_without_finish(ag):
    dev = ag.GPU_DEVICE()
    n, steps = N_BIG, 16
    p1, p2 = ag.ArrowComputePipeline(dev, "a"), ag.ArrowComputePipeline(dev, "b")
    x = dev.create_empty_buffer(4 * n)
    out = dev.create_empty_buffer(4 * n)
    one = dev.create_scalar_buffer(1.0, np.float32)
    capi.call("agpu_broadcast", p1._handle, capi.F32, f32_bits(0.0), vp(x), n)
    slow_chain(p1, x, one, n, steps)
    p2.wait_pipeline(p1)
    capi.call("agpu_copy", p2._handle, vp(out), vp(x), 4 * n)
    got = dev.retrive_data(out, 4 * n, pipeline=p2).view(np.float32)
    assert np.all(got[:: 9973] == steps)


# ------------------------------------------------------------------ sticky error words
def test_error_word_is_not_inherited_by_the_streams_next_owner(ag):
    dev = ag.GPU_DEVICE()
    n = N_BIG
    p1 = ag.ArrowComputePipeline(dev, "old owner")
    x = dev.create_empty_buffer(4 * n)
    one = dev.create_scalar_buffer(1.0, np.float32)
    capi.call("agpu_broadcast", p1._handle, capi.F32, f32_bits(0.0), vp(x), n)
    slow_chain(p1, x, one, n, 12)
    idx = dev.create_gpu_buffer_with_data(np.array([0, 5, 1 << 30], np.uint32))  # out of range for 16 values
    vals = dev.create_gpu_buffer_with_data(np.arange(16, dtype=np.float32))
    out = dev.create_empty_buffer(64)
    capi.call("agpu_take", p1._handle, 4, vp(vals), 16, vp(idx), vp(out), 3)  # runs after the chain: sets the flag LATE
    del p1  # never synchronised: the report is dropped with the pipeline
    p2 = ag.ArrowComputePipeline(dev, "new owner")  # pops p1's stream from the idle pool
    capi.call("agpu_broadcast", p2._handle, capi.F32, f32_bits(3.0), vp(out), 4)
    p2.sync()  # must not raise ShapeError for work it never issued
    dev.sync()
    p2.sync()
    # and its own errors are still reported, exactly once
    capi.call("agpu_take", p2._handle, 4, vp(vals), 16, vp(idx), vp(out), 3)
    with pytest.raises(ag.ArrowErrorGPU):
        p2.sync()
    p2.sync()


# ------------------------------------------------------------------ per-pipeline tuning
def test_tuning_is_per_pipeline(ag):
    dev = ag.GPU_DEVICE()
    p1, p2 = ag.ArrowComputePipeline(dev, "a"), ag.ArrowComputePipeline(dev, "b")
    p1.set_tuning("cmp_variant", 1)
    v = C.c_int64(-1)
    capi.call("agpu_pipeline_get_tuning", p2._handle, b"cmp_variant", C.byref(v))
    assert v.value == 0
    capi.call("agpu_pipeline_get_tuning", p1._handle, b"cmp_variant", C.byref(v))
    assert v.value == 1
    assert capi.lib().agpu_pipeline_set_tuning(p1._handle, b"no_such_key", 1) == capi.ERR_ARG
    # a new default reaches pipelines created afterwards only
    capi.call("agpu_set_tuning", b"tiles", 2)
    try:
        p3 = ag.ArrowComputePipeline(dev, "c")
        capi.call("agpu_pipeline_get_tuning", p3._handle, b"tiles", C.byref(v))
        assert v.value == 2
        capi.call("agpu_pipeline_get_tuning", p2._handle, b"tiles", C.byref(v))
        assert v.value == 0  # 0 = each kernel's measured best
    finally:
        capi.call("agpu_set_tuning", b"tiles", 0)


# ------------------------------------------------------------------ profiling hooks [ref: compute_query.rs, gpu_device.rs:132]
def test_per_launch_timing(ag):
    dev = ag.GPU_DEVICE()
    p = ag.ArrowComputePipeline(dev, "timed")
    n = 1 << 26
    a, b, out = (dev.create_empty_buffer(4 * n) for _ in range(3))
    capi.call("agpu_synth_f32", p._handle, vp(a), n, 1, 0, C.c_float(-1), C.c_float(1))
    capi.call("agpu_synth_f32", p._handle, vp(b), n, 2, 0, C.c_float(-1), C.c_float(1))
    p.enable_timing(0)  # whatever AGPU_PROFILE set for new pipelines: start from "off"
    assert capi.lib().agpu_pipeline_last_kernel_ns(p._handle, C.byref(C.c_uint64()), None) == capi.ERR_ARG  # nothing timed yet
    p.enable_timing(3)  # roctx ranges + event pair
    capi.call("agpu_binary", p._handle, capi.OP_ADD, capi.F32, vp(a), vp(b), vp(out), n)
    ns, name = p.last_kernel_ns()
    assert name == "agpu_binary"
    alg = 12.0 * n
    assert 0.5e12 < alg / (ns * 1e-9) < 8.0e12, ns  # between 0.5 and 8 TB/s: it timed the kernel, not the host
    names = [b"arithmetic/f32/array", b"add_f32"]
    ins = (C.c_void_p * 2)(a.ptr, b.ptr)
    capi.call("agpu_launch_by_name", p._handle, names[0], names[1], ins, 2, vp(out), n)
    ns2, name2 = p.last_kernel_ns()
    assert name2 == "arithmetic/f32/array::add_f32" and ns2 > 0
    sizes = (C.c_uint64 * 2)(4 * n, 4 * n)
    capi.call("agpu_launch_by_name_sized", p._handle, b"compare/f32/min_max", b"max_", ins, sizes, 2, vp(out), 4 * n, (n + 255) // 256)
    assert p.last_kernel_ns()[1] == "compare/f32/min_max::max_"
    p.enable_timing(0)
    capi.call("agpu_binary", p._handle, capi.OP_ADD, capi.F32, vp(a), vp(b), vp(out), n)
    p.sync()


# ------------------------------------------------------------------ small arrays: the reference's own test sizes
def test_small_arrays_never_reach_hipmalloc_or_hipfree(ag):
    dev = ag.GPU_DEVICE()
    a = ag.Int32ArrayGPU.from_slice(np.arange(100, dtype=np.int32), dev)
    b = ag.Int32ArrayGPU.from_slice(np.arange(100, dtype=np.int32)[::-1].copy(), dev)
    for _ in range(50):
        a.add(b)

    def info():
        s, f, l = C.c_uint64(), C.c_uint64(), C.c_uint64()
        capi.call("agpu_device_small_pool_info", dev._handle, C.byref(s), C.byref(f), C.byref(l))
        return s.value, f.value, l.value

    slab0 = info()[0]
    ts = []
    for _ in range(300):
        t0 = time.perf_counter()
        c = a.add(b)
        ts.append(time.perf_counter() - t0)
        del c
    assert info()[0] <= slab0 + (4 << 20)  # steady state: blocks come from the slabs already there
    med_us = float(np.median(ts)) * 1e6
    print(f"a.add(b), 100 rows, default API (new pipeline + new output per op): {med_us:.1f} us per call")
    assert med_us < 500.0  # hipMalloc + hipFree (device sync) per tiny array cost ≈ 250–400 us before the slab pool
    assert np.array_equal(a.add(b).raw_values(), np.full(100, 99, np.int32))


# ------------------------------------------------------------------ f32 Sum of a column that is only 4-byte aligned
@pytest.mark.parametrize("n", [1000, 65536 + 5, 3 * 65536 + 77, N_BIG + 3])
def test_unaligned_f32_sum_keeps_the_reference_tree(ag, n):
    dev = ag.GPU_DEVICE()
    p = ag.ArrowComputePipeline(dev, "sum")
    buf = dev.create_empty_buffer(4 * (n + 4))
    capi.call("agpu_synth_f32", p._handle, vp(buf, 0), n + 1, 77, 0, C.c_float(-1), C.c_float(1))
    out = dev.create_empty_buffer(16)
    capi.call("agpu_reduce", p._handle, capi.RED_SUM, capi.F32, vp(buf, 4), None, n, vp(out))  # slice [1, n+1)
    got = dev.retrive_data(out, 4, pipeline=p).view(np.float32)[0]
    host = O.synth_f32(n + 1, 77, 0, -1.0, 1.0)[1:]
    exp = O.reduce(O.RED_SUM, O.F32, host)
    assert np.float32(got).view(np.uint32) == np.float32(exp).view(np.uint32), (got, exp)
    # with a validity bitmap too (bit i belongs to row i of the slice)
    if n <= 3 * 65536 + 77:
        v = O.synth_bits(n, 5, 0, 0.8)
        dv = dev.create_gpu_buffer_with_data(v)
        capi.call("agpu_reduce", p._handle, capi.RED_SUM, capi.F32, vp(buf, 4), vp(dv), n, vp(out))
        got = dev.retrive_data(out, 4, pipeline=p).view(np.float32)[0]
        exp = O.reduce(O.RED_SUM, O.F32, host, v)
        assert np.float32(got).view(np.uint32) == np.float32(exp).view(np.uint32)
