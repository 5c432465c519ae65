"""GPU: the bucketed take / put (swizzle.hip "bucketed take / put", tuning key gather_bucket = 2) must give exactly the
results of the direct kernels and of the oracle — all widths, out-of-range indices (robust-access outcome + sticky
flag), ragged sizes around the 16 Ki-row tile, skewed index distributions, distinct destinations for put (duplicate
destinations have no defined winner in the reference either).  [ref: crates/routines/src/take.rs:9-55, put.rs:9-56]"""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi

pytestmark = pytest.mark.gpu
NPW = {1: np.uint8, 2: np.uint16, 4: np.uint32}


@pytest.fixture(scope="module")
def ctx():
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "bucketed")
    p.set_tuning("gather_bucket", 2)
    return dev, p


def vp(b):
    return C.c_void_p(b.ptr)


@pytest.mark.parametrize("width", [4, 2, 1])
@pytest.mark.parametrize("n,n_values,dist", [(16384, 1000, "uniform"), (16385, 1 << 20, "uniform"), (300_001, 3_000_017, "uniform"),
                                             (1 << 20, 1 << 22, "skew"), (2_000_003, 5, "uniform"), (1 << 21, (1 << 22) + 7, "sorted")])
def test_bucketed_take_equals_oracle(ctx, width, n, n_values, dist):
    dev, p = ctx
    rng = np.random.default_rng(n + width)
    values = rng.integers(0, 1 << (8 * width), n_values, dtype=np.uint64).astype(NPW[width])
    if dist == "skew":  # 90 % of the rows hit one 64-element neighbourhood: LDS counters and one bucket take the load
        idx = np.where(rng.random(n) < 0.9, rng.integers(1000, 1064, n), rng.integers(0, n_values, n)).astype(np.uint32)
    elif dist == "sorted":
        idx = np.sort(rng.integers(0, n_values, n)).astype(np.uint32)
    else:
        idx = rng.integers(0, n_values, n).astype(np.uint32)
    dv, di = dev.create_gpu_buffer_with_data(values), dev.create_gpu_buffer_with_data(idx)
    out = dev.create_empty_buffer(n * width + 16)
    capi.call("agpu_take", p._handle, width, vp(dv), n_values, vp(di), vp(out), n)
    p.sync()
    got = dev.retrive_data(out, n * width, pipeline=p).view(NPW[width])
    assert np.array_equal(got, O.take(width, values, idx))


def test_bucketed_take_out_of_range_reads_zero_and_flags(ctx):
    import arrow_gpu_amd as ag

    dev, p = ctx
    n, n_values = 100_000, 50_000
    rng = np.random.default_rng(1)
    values = rng.integers(1, 1 << 32, n_values, dtype=np.uint64).astype(np.uint32)
    idx = rng.integers(0, n_values, n).astype(np.uint32)
    bad = rng.choice(n, 37, replace=False)
    idx[bad] = rng.integers(n_values, 1 << 32, 37, dtype=np.uint64).astype(np.uint32)
    dv, di, out = dev.create_gpu_buffer_with_data(values), dev.create_gpu_buffer_with_data(idx), dev.create_empty_buffer(4 * n)
    capi.call("agpu_take", p._handle, 4, vp(dv), n_values, vp(di), vp(out), n)
    with pytest.raises(ag.ArrowErrorGPU):
        p.sync()
    got = dev.retrive_data(out, 4 * n, pipeline=p).view(np.uint32)
    exp = np.where(idx < n_values, values[np.minimum(idx, n_values - 1)], 0).astype(np.uint32)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("width", [4, 2, 1])
@pytest.mark.parametrize("n,n_src,n_dst", [(16384, 5000, 40000), (100_003, 1 << 20, 1 << 21), (1 << 20, 3_000_001, (1 << 22) + 5)])
def test_bucketed_put_equals_oracle(ctx, width, n, n_src, n_dst):
    dev, p = ctx
    rng = np.random.default_rng(n + 10 * width)
    src = rng.integers(0, 1 << (8 * width), n_src, dtype=np.uint64).astype(NPW[width])
    dst = rng.integers(0, 1 << (8 * width), n_dst, dtype=np.uint64).astype(NPW[width])
    si = rng.integers(0, n_src, n).astype(np.uint32)
    di = rng.permutation(n_dst)[:n].astype(np.uint32)  # distinct destinations: the result is defined
    ds, dd = dev.create_gpu_buffer_with_data(src), dev.create_gpu_buffer_with_data(dst)
    dsi, ddi = dev.create_gpu_buffer_with_data(si), dev.create_gpu_buffer_with_data(di)
    capi.call("agpu_put_bounded", p._handle, width, vp(ds), n_src, vp(dsi), vp(dd), n_dst, vp(ddi), n)
    p.sync()
    got = dev.retrive_data(dd, n_dst * width, pipeline=p).view(NPW[width])
    exp = dst.copy()
    exp[di] = src[si]
    assert np.array_equal(got, exp)


def test_bucketed_put_drops_out_of_range_rows(ctx):
    import arrow_gpu_amd as ag

    dev, p = ctx
    n, n_src, n_dst = 50_000, 20_000, 60_000
    rng = np.random.default_rng(3)
    src = rng.integers(0, 1 << 32, n_src, dtype=np.uint64).astype(np.uint32)
    dst = np.zeros(n_dst, np.uint32)
    si = rng.integers(0, n_src, n).astype(np.uint32)
    di = rng.permutation(n_dst)[:n].astype(np.uint32)
    si[::101] = n_src + 5
    di[::103] = n_dst
    ok = (si < n_src) & (di < n_dst)
    ds, dd = dev.create_gpu_buffer_with_data(src), dev.create_gpu_buffer_with_data(dst)
    dsi, ddi = dev.create_gpu_buffer_with_data(si), dev.create_gpu_buffer_with_data(di)
    capi.call("agpu_put_bounded", p._handle, 4, vp(ds), n_src, vp(dsi), vp(dd), n_dst, vp(ddi), n)
    with pytest.raises(ag.ArrowErrorGPU):
        p.sync()
    got = dev.retrive_data(dd, 4 * n_dst, pipeline=p).view(np.uint32)
    exp = dst.copy()
    exp[di[ok]] = src[si[ok]]
    assert np.array_equal(got, exp)
