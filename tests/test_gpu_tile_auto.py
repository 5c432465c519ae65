"""GPU: the adaptive tiles-per-block policy (tuning tile_auto, runtime.hip agpu_tiles_pick).  For a big launch whose tile tuning is "auto" the
device times the first launches per (kernel family, size class, buffer region) with one and with two tiles per block and keeps the faster form.
What it must guarantee whatever it measures: the result never changes, it only looks at launches that move ≥ 256 MiB, explicit tunings and
tile_auto = 1 switch it off, and it decides after four samples of each form — in a loop with a sync per launch and in a burst without one."""
import ctypes as C
import re

import numpy as np
import pytest

from arrow_gpu_amd import _capi as capi

pytestmark = pytest.mark.gpu

N = 1 << 27  # rows: 640 MiB of traffic for cast u8 → f32, 1 GiB for sin f32


def vp(b):
    return C.c_void_p(b.ptr)


def entries(dev):
    out = {}
    for part in filter(None, (x.strip() for x in dev.tile_auto_info().split(";"))):
        m = re.fullmatch(r"(\w+)\.\d+ lg=(\d+) tiles=(\d) samples=(\d+)/(\d+) ns_per_GB=(\d+)/(\d+)", part)
        assert m, part
        out.setdefault(m.group(1), []).append({"lg": int(m.group(2)), "tiles": int(m.group(3)), "n": (int(m.group(4)), int(m.group(5))),
                                               "ns_per_GB": (int(m.group(6)), int(m.group(7)))})
    return out


@pytest.fixture(scope="module")
def ctx():
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "tile-auto")
    u8 = dev.create_empty_buffer(2 * N)
    f = dev.create_empty_buffer(4 * N)
    out = dev.create_empty_buffer(4 * N)
    chk = dev.create_empty_buffer(8)
    capi.call("agpu_synth_u8", p._handle, vp(u8), 2 * N, 6, 0)
    capi.call("agpu_synth_f32", p._handle, vp(f), N, 1, 0, C.c_float(-50.0), C.c_float(50.0))
    p.sync()
    return dev, p, u8, f, out, chk


def checksum(dev, p, out, chk, nbytes):
    capi.call("agpu_checksum", p._handle, vp(out), nbytes, vp(chk))
    return int(dev.retrive_data(chk, 8, pipeline=p).view(np.uint64)[0])


LAUNCHES = {
    "cast": ("tiles", lambda h, u8, f, out: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(out), N)),
    "heavy": ("tiles", lambda h, u8, f, out: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(out), N)),
    "lut8": ("tiles", lambda h, u8, f, out: capi.call("agpu_unary", h, capi.UN_COS, capi.U8, vp(u8), vp(out), N)),
    "log": ("tiles", lambda h, u8, f, out: capi.call("agpu_unary", h, capi.UN_LOG, capi.F32, vp(f), vp(out), N)),
}


@pytest.mark.parametrize("family", list(LAUNCHES))
def test_decides_after_six_samples_and_never_changes_a_result(ctx, family):
    dev, p, u8, f, out, chk = ctx
    key, launch = LAUNCHES[family]
    h = p._handle
    sums = {}
    for k in (1, 2, 3):  # the forced forms first: what the result must be
        p.set_tuning(key, k)
        launch(h, u8, f, out)
        sums[k] = checksum(dev, p, out, chk, 4 * N)
    assert sums[1] == sums[2] == sums[3]
    assert family not in entries(dev), "explicit tile counts must not be measured"
    p.set_tuning(key, 0)
    p.set_tuning("tile_auto", 1)
    launch(h, u8, f, out)
    p.sync()
    assert family not in entries(dev), "tile_auto = 1 switches the policy off"
    p.set_tuning("tile_auto", 0)
    for i in range(10):  # a sync per launch: every sample is back before the next launch looks
        launch(h, u8, f, out)
        assert checksum(dev, p, out, chk, 4 * N) == sums[1], (family, i)
    e = entries(dev)[family]
    assert len(e) == 1 and e[0]["tiles"] in (1, 2) and e[0]["n"] == (4, 4), e
    assert all(50_000 < v < 2_000_000 for v in e[0]["ns_per_GB"]), e  # 0.5 … 20 TB/s: the samples are kernel times, not queueing
    # the faster form must be the chosen one unless two tiles won by less than 2.5 %
    one, two = e[0]["ns_per_GB"]
    assert e[0]["tiles"] == (2 if two * 1.025 < one else 1), e
    for i in range(3):  # and from now on no launch is timed
        launch(h, u8, f, out)
    p.sync()
    assert entries(dev)[family] == e


def test_small_launches_are_left_alone_and_bursts_converge(ctx):
    dev, p, u8, f, out, chk = ctx
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline

    h = p._handle
    before = entries(dev)
    n_small = 1 << 24  # 80 MiB of traffic
    for _ in range(8):
        capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(out), n_small)
        p.sync()
    assert entries(dev) == before
    # another size class of the cast, launched in bursts with no sync in between: the eight samples of a burst are all in flight at once
    n2 = 3 << 25
    q = ArrowComputePipeline(dev, "burst")
    for burst in range(3):
        for _ in range(10):
            capi.call("agpu_cast", q._handle, capi.U8, capi.F32, vp(u8), vp(out), n2)
        q.sync()
    mine = [x for x in entries(dev)["cast"] if x["lg"] == (5 * n2).bit_length() - 1]
    assert len(mine) == 1 and mine[0]["tiles"] in (1, 2) and mine[0]["n"] == (4, 4), entries(dev)
    ref = checksum(dev, q, out, chk, 4 * n2)
    q.set_tuning("tiles", 1)
    capi.call("agpu_cast", q._handle, capi.U8, capi.F32, vp(u8), vp(out), n2)
    assert checksum(dev, q, out, chk, 4 * n2) == ref


def test_on_a_wrapped_foreign_stream_nothing_is_timed(ctx):
    """a wrapped stream may be inside a capture of its owner's that the library cannot see"""
    dev, p, u8, f, out, chk = ctx
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline

    owner = ArrowComputePipeline(dev, "owner")           # stands in for the foreign owner of the stream
    w = ArrowComputePipeline(dev, "wrapped", hip_stream=owner.stream())
    before = entries(dev)
    n4 = 7 << 24
    for _ in range(10):
        capi.call("agpu_cast", w._handle, capi.U8, capi.F32, vp(u8), vp(out), n4)
        w.sync()
    assert entries(dev) == before


def test_inside_a_captured_graph_nothing_is_timed(ctx):
    dev, p, u8, f, out, chk = ctx
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline

    q = ArrowComputePipeline(dev, "graph")
    n3 = 5 << 24
    capi.call("agpu_cast", q._handle, capi.I8, capi.F32, vp(u8), vp(out), 1 << 20)  # warm anything lazy outside the capture
    q.sync()
    before = entries(dev)
    g = C.c_void_p()
    capi.call("agpu_pipeline_begin_capture", q._handle)
    capi.call("agpu_cast", q._handle, capi.I8, capi.F32, vp(u8), vp(out), n3)
    capi.call("agpu_pipeline_end_capture", q._handle, C.byref(g))
    for _ in range(3):
        capi.call("agpu_graph_launch", g, q._handle)
    q.sync()
    capi.call("agpu_graph_destroy", g)
    assert entries(dev) == before


def test_four_host_threads_sampling_the_same_key(ctx):
    """four pipelines on four host threads launch the same kernel on the same buffers while the policy is still measuring: a sample slot
    belongs to ONE launch from its pick to its harvest (two launches on one event pair would time garbage), results stay bit-identical
    and the entry still converges"""
    import threading

    dev, p, u8, f, out, chk = ctx
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline

    n5 = 9 << 23
    outs = [dev.create_empty_buffer(4 * n5) for _ in range(4)]
    p.set_tuning("tiles", 1)
    capi.call("agpu_cast", p._handle, capi.I8, capi.F32, vp(u8), vp(out), n5)
    ref = checksum(dev, p, out, chk, 4 * n5)
    p.set_tuning("tiles", 0)
    errs = []

    def worker(k):
        try:
            q = ArrowComputePipeline(dev, f"t{k}")
            q.set_tuning("tile_auto", 1 << 20)  # every launch of this size is eligible
            c = dev.create_empty_buffer(8)
            for _ in range(12):
                capi.call("agpu_cast", q._handle, capi.I8, capi.F32, vp(u8), vp(outs[k]), n5)
            q.sync()
            capi.call("agpu_checksum", q._handle, vp(outs[k]), 4 * n5, vp(c))
            got = int(dev.retrive_data(c, 8, pipeline=q).view(np.uint64)[0])
            if got != ref:
                errs.append((k, got, ref))
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for fam_entries in entries(dev).values():
        for e in fam_entries:
            assert e["n"][0] <= 4 and e["n"][1] <= 4, e   # never more samples than slots were handed out for
