"""Round 5 (R5.10): the pinned-mailbox waits — agpu_pipeline_sync, agpu_download of ≤ 64 bytes, agpu_device_sync / agpu_device_download.
They replace hipStreamSynchronize / hipDeviceSynchronize on the latency path; what they must keep: every byte delivered, every earlier
kernel over, sticky kernel errors surfacing at the sync, any number of threads and pipelines, graph captures left alone."""
import ctypes as C
import threading

import numpy as np
import pytest

from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return GpuDevice(0)


def _vp(b, off=0):
    return C.c_void_p(b.ptr + off)


@pytest.mark.parametrize("spin", [0, -1, 1])
def test_small_copies_deliver_every_byte_at_any_offset(dev, spin):
    """uploads and downloads of 1 … AGPU_MAILBOX_MAX_BYTES (and a few just above: the ordinary path) at odd device offsets, 200 rounds with
    fresh contents: the mailbox payload, its fallback (spin = 1 µs ⇒ mostly the blocking path) and the plain copies (spin < 0) move exactly
    the bytes asked for and nothing beside them"""
    p = ArrowComputePipeline(dev, "mb")
    p.set_tuning("sync_spin", spin)
    rng = np.random.default_rng(5 + spin)
    M = capi.MAILBOX_MAX_BYTES
    buf = dev.create_empty_buffer(16384)
    shadow = rng.integers(0, 256, 16384, dtype=np.uint8)
    capi.call("agpu_upload", p._handle, _vp(buf), C.c_void_p(shadow.ctypes.data), 16384)
    sizes = [1, 2, 3, 4, 15, 16, 17, 63, 64, 65, 255, 256, 1000, 2048, M - 1, M, M + 1, M + 17]
    for r in range(200):
        n = int(sizes[r % len(sizes)]) if r % 3 else int(rng.integers(1, M + 64))
        off = int(rng.integers(0, 16384 - n)) if r % 2 else 16 * int(rng.integers(0, (16384 - n) // 16))
        data = rng.integers(0, 256, n, dtype=np.uint8)
        capi.call("agpu_upload", p._handle, _vp(buf, off), C.c_void_p(data.ctypes.data), n)
        shadow[off:off + n] = data
        m = int(sizes[(r * 7) % len(sizes)])
        o2 = int(rng.integers(0, 16384 - m))
        out = np.zeros(m + 2, np.uint8)
        out[-2:] = (0xA5, 0x5A)
        capi.call("agpu_download", p._handle, C.c_void_p(out.ctypes.data), _vp(buf, o2), m)
        assert np.array_equal(out[:m], shadow[o2:o2 + m]) and out[-2] == 0xA5 and out[-1] == 0x5A, (r, n, off, m, o2)
    full = dev.retrive_data(buf, 16384, pipeline=p)
    assert np.array_equal(full, shadow)


def test_device_download_of_a_small_array(dev):
    """retrive_data of up to AGPU_MAILBOX_MAX_BYTES is one device-level wait that carries the bytes; one byte more takes sync + copy"""
    p = ArrowComputePipeline(dev, "dd")
    M = capi.MAILBOX_MAX_BYTES
    x = np.arange(2048, dtype=np.float32)
    a = dev.create_gpu_buffer_with_data(x)
    b = dev.create_empty_buffer(4 * 2048)
    for n in (5, 100, M // 4, M // 4 + 1, 2048):
        capi.call("agpu_unary", p._handle, capi.UN_NEG, capi.F32, _vp(a), _vp(b), n)
        got = dev.retrive_data(b, 4 * n).view(np.float32)
        assert np.array_equal(got, -x[:n]), n


def test_a_scalar_download_waits_for_the_kernels_before_it(dev):
    """a 1e8-row reduction (≈ 60 µs of GPU time: longer than the launch) followed at once by the 8-byte download of its result: always the sum"""
    p = ArrowComputePipeline(dev, "mb2")
    n = 100_000_000
    a = dev.create_empty_buffer(4 * n)
    out = dev.create_empty_buffer(16)
    for k in range(1, 6):
        capi.call("agpu_synth_i32", p._handle, _vp(a), n, k, 0, 1000 * k)
        capi.call("agpu_reduce", p._handle, capi.RED_MAX, capi.I32, _vp(a), None, n, _vp(out))
        got = np.zeros(1, np.int32)
        capi.call("agpu_download", p._handle, C.c_void_p(got.ctypes.data), _vp(out), 4)
        ref = dev.retrive_data(a, 4 * 65536, pipeline=p).view(np.int32)
        full = np.zeros(1, np.int32)
        capi.call("agpu_reduce", p._handle, capi.RED_MAX, capi.I32, _vp(a), None, n, _vp(out))
        p.sync()
        capi.call("agpu_download", p._handle, C.c_void_p(full.ctypes.data), _vp(out), 4)
        assert got[0] == full[0] and ref.max() <= got[0] < 1000 * k, (k, got, full)


def test_sync_still_surfaces_the_sticky_index_error(dev):
    import arrow_gpu_amd as ag

    p = ArrowComputePipeline(dev, "mb3")
    vals = dev.create_gpu_buffer_with_data(np.arange(16, dtype=np.uint32))
    idx = dev.create_gpu_buffer_with_data(np.array([0, 5, 1 << 30], np.uint32))
    out = dev.create_empty_buffer(64)
    capi.call("agpu_take", p._handle, 4, _vp(vals), 16, _vp(idx), _vp(out), 3)
    with pytest.raises(ag.ArrowErrorGPU):
        p.sync()
    p.sync()  # reported once


def test_device_wait_covers_every_stream(dev):
    """three pipelines with long kernels outstanding (the slow path: hipDeviceSynchronize), then one (the mailbox on that stream), then none:
    agpu_device_download returns the value the LAST kernel wrote"""
    n = 50_000_000
    ps = [ArrowComputePipeline(dev, f"dw{k}") for k in range(3)]
    bufs = [dev.create_empty_buffer(4 * n) for _ in ps]
    outs = [dev.create_empty_buffer(16) for _ in ps]
    for rounds, active in ((3, 3), (3, 1), (2, 0)):
        for r in range(rounds):
            for k in range(active):
                capi.call("agpu_synth_i32", ps[k]._handle, _vp(bufs[k]), n, 100 * r + k, 0, 1 << 20)
                capi.call("agpu_reduce", ps[k]._handle, capi.RED_MAX, capi.I32, _vp(bufs[k]), None, n, _vp(outs[k]))
            got = [dev.retrive_data(outs[k], 4).view(np.int32)[0] for k in range(3)]
            for k in range(active):
                exp = np.zeros(1, np.int32)
                ps[k].sync()
                capi.call("agpu_download", ps[k]._handle, C.c_void_p(exp.ctypes.data), _vp(outs[k]), 4)
                assert got[k] == exp[0], (rounds, active, r, k)


def test_device_sync_leaves_a_capturing_stream_alone(dev):
    """a pipeline in graph capture has 'work outstanding' as far as the counters go: a device-level wait (from another thread — the capturing
    thread itself may not synchronise) must not launch its post kernel into the capture: it would become a node of the graph and the host
    would wait for a kernel that never runs ('the device's mailbox was not posted')"""
    import os

    if int(os.environ.get("AGPU_SYNC_SPIN", "0") or 0) < 0:
        pytest.skip("AGPU_SYNC_SPIN < 0: device waits are hipDeviceSynchronize, which invalidates an open capture (the behaviour before round 5)")
    p = ArrowComputePipeline(dev, "cap")
    n = 1 << 20
    a = dev.create_gpu_buffer_with_data(np.arange(n, dtype=np.float32))
    b = dev.create_empty_buffer(4 * n)
    dev.sync()
    capi.call("agpu_pipeline_begin_capture", p._handle)
    capi.call("agpu_unary", p._handle, capi.UN_NEG, capi.F32, _vp(a), _vp(b), n)
    errs = []

    def waiter():
        try:
            dev.sync()                                   # p's stream is the ONE with calls since its last wait — and it is capturing
            errs.append(dev.retrive_data(a, 8).view(np.float32).tolist())
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    t = threading.Thread(target=waiter)
    t.start()
    t.join()
    g = C.c_void_p()
    capi.call("agpu_pipeline_end_capture", p._handle, C.byref(g))
    assert errs == [[0.0, 1.0]], errs
    for _ in range(3):
        capi.call("agpu_graph_launch", g, p._handle)
    p.sync()
    assert np.array_equal(dev.retrive_data(b, 16, pipeline=p).view(np.float32), -np.arange(4, dtype=np.float32))
    capi.call("agpu_graph_destroy", g)


def test_threads_with_their_own_pipelines_and_device_waits(dev):
    """four threads: each runs reductions on its own pipeline and reads the result through the pipeline's mailbox; two of them also call the
    device-level download (which may land its post kernel on ANOTHER thread's stream)"""
    n = 1 << 22
    errs = []

    def work(k):
        try:
            p = ArrowComputePipeline(dev, f"t{k}")
            a = dev.create_empty_buffer(4 * n)
            out = dev.create_empty_buffer(16)
            for r in range(150):
                capi.call("agpu_synth_i32", p._handle, _vp(a), n, 7 * r + k, 0, 1 << 20)
                capi.call("agpu_reduce", p._handle, capi.RED_SUM, capi.I32, _vp(a), None, n, _vp(out))
                got = np.zeros(1, np.int32)
                capi.call("agpu_download", p._handle, C.c_void_p(got.ctypes.data), _vp(out), 4)
                via_dev = dev.retrive_data(out, 4).view(np.int32)[0] if k % 2 == 0 else got[0]
                if r % 50 == 0:
                    host = dev.retrive_data(a, 4 * n, pipeline=p).view(np.int32)
                    assert got[0] == host.sum(dtype=np.int64).astype(np.int32)
                assert via_dev == got[0], (k, r)
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_pipeline_slots_are_recycled_with_their_sequence_numbers(dev):
    """200 short-lived pipelines, each downloading through its mailbox right away: a recycled pinned slot continues its previous owner's count"""
    a = dev.create_gpu_buffer_with_data(np.arange(64, dtype=np.uint32))
    for k in range(200):
        p = ArrowComputePipeline(dev, "short")
        out = np.zeros(2, np.uint32)
        capi.call("agpu_download", p._handle, C.c_void_p(out.ctypes.data), _vp(a, 4 * (k % 60)), 8)
        assert out.tolist() == [k % 60, k % 60 + 1]
        del p


def test_device_wait_behind_a_long_queue_sleeps_instead_of_spinning(dev):
    """process-wide sync_spin = 1 µs: the device-level wait leaves its spin at once and sleep-polls the sequence word (with the stream's liveness
    check every few milliseconds) behind ≈ 20 ms of queued kernels; the value it delivers is still the last kernel's"""
    p = ArrowComputePipeline(dev, "long")
    n = 200_000_000
    a = dev.create_empty_buffer(4 * n)
    out = dev.create_empty_buffer(16)
    dev.sync()
    capi.call("agpu_set_tuning", b"sync_spin", 1)
    try:
        for r in range(3):
            for k in range(12):
                capi.call("agpu_synth_i32", p._handle, _vp(a), n, 31 * r + k, 0, 1 << 20)
            capi.call("agpu_reduce", p._handle, capi.RED_MAX, capi.I32, _vp(a), None, n, _vp(out))
            got = dev.retrive_data(out, 4).view(np.int32)[0]
            exp = np.zeros(1, np.int32)
            capi.call("agpu_download", p._handle, C.c_void_p(exp.ctypes.data), _vp(out), 4)
            assert got == exp[0] and 0 < got < (1 << 20)
    finally:
        capi.call("agpu_set_tuning", b"sync_spin", 0)


def test_device_waits_from_every_thread_while_pipelines_come_and_go(dev):
    """six threads, each: a short-lived pipeline per round (its stream goes back to the pool with work outstanding), a reduction, the result
    through the DEVICE-level download — which may find one, several or no streams outstanding, and posts on whichever it finds"""
    n = 1 << 21
    errs = []
    srcs = []
    for k in range(6):
        x = (np.arange(n, dtype=np.uint64) * (2 * k + 5) + k).astype(np.uint32)
        srcs.append((dev.create_gpu_buffer_with_data(x), int(x.astype(np.uint64).sum() & 0xFFFFFFFF), int(x.max())))

    def work(k):
        try:
            buf, s_exp, m_exp = srcs[k]
            out = dev.create_empty_buffer(16)
            for r in range(120):
                p = ArrowComputePipeline(dev, f"churn{k}")
                op, exp = (capi.RED_SUM, s_exp) if r % 2 == 0 else (capi.RED_MAX, m_exp)
                capi.call("agpu_reduce", p._handle, op, capi.U32, _vp(buf), None, n, _vp(out))
                capi.call("agpu_pipeline_finish", p._handle)
                del p
                got = int(dev.retrive_data(out, 4).view(np.uint32)[0])
                assert got == exp, (k, r, got, exp)
                if r % 40 == 39:
                    dev.sync()
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_device_wait_covers_foreign_work_on_a_stream_whose_handle_was_handed_out(dev):
    """the device-level waits skip owned streams with no ABI call since their last completed wait.  A stream whose raw handle was
    handed out (agpu_pipeline_stream) may carry work the library never saw: it must be waited for EVERY time.  Here: a pipeline is drained,
    its handle taken, a 2 GiB hipMemsetAsync queued on it behind the library's back — agpu_device_download of the buffer's last bytes must
    see the new value (before the fix the stream counted as empty and the bytes came back stale), and with sync_spin < 0 agpu_device_sync
    calls the runtime's wait whatever the counters say."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    hip.hipMemsetAsync.restype = C.c_int
    n = 2 << 30
    p = ArrowComputePipeline(dev, "exposed")
    buf = dev.create_empty_buffer(n)
    for round_, spin in enumerate((0, -1)):
        capi.call("agpu_set_tuning", b"sync_spin", spin)
        try:
            capi.call("agpu_memset", p._handle, C.c_void_p(buf.ptr), 0x11 + round_, n)
            p.sync()
            dev.sync()  # everything this library queued is through and KNOWN to be: the stream counts as drained
            stream = p.stream()  # … and from here on as carrying foreign work
            assert hip.hipMemsetAsync(C.c_void_p(buf.ptr), 0x77 + round_, n, C.c_void_p(stream)) == 0
            if spin < 0:
                dev.sync()
                got = np.empty(64, np.uint8)
                capi.call("agpu_download", p._handle, C.c_void_p(got.ctypes.data), C.c_void_p(buf.ptr + n - 64), 64)
            else:
                got = np.empty(64, np.uint8)
                capi.call("agpu_device_download", dev._handle, C.c_void_p(got.ctypes.data), C.c_void_p(buf.ptr + n - 64), 64)
            assert (got == 0x77 + round_).all(), (spin, got[:8])
        finally:
            capi.call("agpu_set_tuning", b"sync_spin", 0)
    p.sync()
