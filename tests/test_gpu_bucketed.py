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


# ---- round 3: 4-byte takes go through the merge-back pipeline (tk2_*: 32 Ki-row tiles, ranks recorded, runs merged back);
# gather_bucket = 3 keeps the pair pipeline.  Both against the oracle, around the new tile sizes, with the G2 slow path
# (tiles whose sources span more than two regions), a whole tile of out-of-range rows, and one-region sources.
@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("n,n_values,dist", [(32767, 70_000, "uniform"), (32768, 1 << 20, "uniform"), (32769, 9, "uniform"),
                                             (65_537, 1 << 22, "uniform"), (40_000, 300_000_007, "uniform"),
                                             (1_000_003, 40_000_003, "uniform"), (3 * 32768 + 5, 1 << 24, "oob_tile"),
                                             (500_000, 1 << 23, "skew"), (1 << 20, 1 << 21, "sorted"), (2_500_000, 1 << 20, "dups")])
def test_take_u32_pipelines_equal_the_oracle(ctx, mode, n, n_values, dist):
    dev, p = ctx
    p.set_tuning("gather_bucket", mode)
    try:
        rng = np.random.default_rng(n * 7 + mode)
        values = rng.integers(1, 1 << 32, n_values, dtype=np.uint64).astype(np.uint32)
        idx = rng.integers(0, n_values, n).astype(np.uint32)
        expect_flag = False
        if dist == "skew":
            idx = np.where(rng.random(n) < 0.9, rng.integers(1000, 1064, n), idx).astype(np.uint32)
        elif dist == "sorted":
            idx = np.sort(idx)
        elif dist == "dups":
            idx = (idx // 4096 * 4096).astype(np.uint32)  # 4096 rows per distinct source
        elif dist == "oob_tile":  # the middle tile is out of range from its first to its last row
            idx[32768:65536] = rng.integers(n_values, 1 << 32, 32768, dtype=np.uint64).astype(np.uint32)
            expect_flag = True
        dv, di = dev.create_gpu_buffer_with_data(values), dev.create_gpu_buffer_with_data(idx)
        out = dev.create_empty_buffer(4 * n + 16)
        capi.call("agpu_memset", p._handle, vp(out), 0xEE, 4 * n + 16)
        capi.call("agpu_take", p._handle, 4, vp(dv), n_values, vp(di), vp(out), n)
        if expect_flag:
            import arrow_gpu_amd as ag

            with pytest.raises(ag.ArrowErrorGPU):
                p.sync()
        else:
            p.sync()
        got = dev.retrive_data(out, 4 * n + 16, pipeline=p)
        exp = np.where(idx < n_values, values[np.minimum(idx, n_values - 1)], 0).astype(np.uint32)
        assert np.array_equal(got[: 4 * n].view(np.uint32), exp)
        assert (got[4 * n:] == 0xEE).all()  # nothing past the end
        if not expect_flag:
            assert np.array_equal(exp, O.take(4, values, idx))
    finally:
        p.set_tuning("gather_bucket", 2)


def test_take_u32_at_2_28_rows_merge_back_equals_direct(ctx):
    """full size: 2^28 uniformly random rows over a 1 GiB source — merge-back, pair pipeline and direct kernel agree (checksums
    of the outputs) and a window matches the oracle"""
    dev, p = ctx
    n = 1 << 28
    h = p._handle
    values, idx = dev.create_empty_buffer(4 * n), dev.create_empty_buffer(4 * n)
    capi.call("agpu_synth_i32", h, vp(values), n, 11, 0, 0)
    capi.call("agpu_synth_i32", h, vp(idx), n, 12, 0, n)
    sums = {}
    outs = {}
    for mode in (1, 2, 3):
        p.set_tuning("gather_bucket", mode)
        outs[mode] = dev.create_empty_buffer(4 * n)
        capi.call("agpu_take", h, 4, vp(values), n, vp(idx), vp(outs[mode]), n)
        cs = dev.create_empty_buffer(8)
        capi.call("agpu_checksum", h, vp(outs[mode]), 4 * n, vp(cs))
        sums[mode] = int(dev.retrive_data(cs, 8, pipeline=p).view(np.uint64)[0])
    p.set_tuning("gather_bucket", 2)
    assert sums[1] == sums[2] == sums[3], sums
    w = 1 << 16
    vi = O.synth_i32(n, 11, 0, 0) if False else None  # (the whole column is too slow for the scalar oracle: a window of rows)
    ix = O.synth_i32(w, 12, (n // 2), n).view(np.uint32)
    got = np.empty(w, np.uint32)
    capi.call("agpu_download", h, C.c_void_p(got.ctypes.data), C.c_void_p(outs[2].ptr + 4 * (n // 2)), 4 * w)
    # values[j] = synth_i32 at row j: regenerate just the rows the window's indices name
    exp = np.array([O.synth_i32(1, 11, int(j), 0)[0] for j in ix[:2048]], np.int32).view(np.uint32)
    assert np.array_equal(got[:2048], exp)
    del vi


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


# ---- round 3: take of an array WITH NULLS — agpu_take_validity: the validity bit travels with the value
@pytest.mark.parametrize("mode", [1, 2, 4])
@pytest.mark.parametrize("n,n_values,dist", [(5, 9, "uniform"), (32767, 70_000, "uniform"), (32768, 1 << 20, "uniform"), (65_537, 1 << 22, "uniform"),
                                             (40_000, 300_000_007, "uniform"), (1_000_003, 40_000_003, "uniform"),
                                             (3 * 32768 + 5, 1 << 24, "oob_tile"), (500_000, 1 << 23, "skew"), (2_500_000, 1 << 20, "dups")])
def test_take_validity_equals_take_plus_take_bits(ctx, mode, n, n_values, dist):
    """direct (mode 1: agpu_take + agpu_take_bits) and merge-back (mode 2: one pipeline) against the oracle's take and take_bits"""
    dev, p = ctx
    p.set_tuning("gather_bucket", mode)
    try:
        rng = np.random.default_rng(n * 11 + mode)
        values = rng.integers(1, 1 << 32, n_values, dtype=np.uint64).astype(np.uint32)
        vbits = np.packbits(rng.random((n_values + 63) // 64 * 64) < 0.8, bitorder="little")  # whole words; padding random (never addressed)
        idx = rng.integers(0, n_values, n).astype(np.uint32)
        expect_flag = False
        if dist == "skew":
            idx = np.where(rng.random(n) < 0.9, rng.integers(1000, 1064, n), idx).astype(np.uint32)
        elif dist == "dups":
            idx = (idx // 4096 * 4096).astype(np.uint32)
        elif dist == "oob_tile":
            idx[32768:65536] = rng.integers(n_values, 1 << 32, 32768, dtype=np.uint64).astype(np.uint32)
            expect_flag = True
        dv, dvb, di = dev.create_gpu_buffer_with_data(values), dev.create_gpu_buffer_with_data(vbits), dev.create_gpu_buffer_with_data(idx)
        nb = O.bitmap_bytes(n)
        out, outv = dev.create_empty_buffer(4 * n + 16), dev.create_empty_buffer(nb + 16)
        capi.call("agpu_memset", p._handle, vp(out), 0xEE, 4 * n + 16)
        capi.call("agpu_memset", p._handle, vp(outv), 0xEE, nb + 16)
        capi.call("agpu_take_validity", p._handle, 4, vp(dv), n_values, vp(dvb), vp(di), vp(out), vp(outv), n)
        if expect_flag:
            import arrow_gpu_amd as ag

            with pytest.raises(ag.ArrowErrorGPU):
                p.sync()
        else:
            p.sync()
        got = dev.retrive_data(out, 4 * n + 16, pipeline=p)
        gotv = dev.retrive_data(outv, nb + 16, pipeline=p)
        ok = idx < n_values
        exp = np.where(ok, values[np.minimum(idx, n_values - 1)], 0).astype(np.uint32)
        src_bits = np.unpackbits(vbits, bitorder="little")
        exp_bits = np.where(ok, src_bits[np.minimum(idx, n_values - 1)], 0).astype(np.uint8)
        assert np.array_equal(got[: 4 * n].view(np.uint32), exp)
        assert (got[4 * n:] == 0xEE).all()
        got_bits = np.unpackbits(gotv[:nb], bitorder="little")
        assert np.array_equal(got_bits[:n], exp_bits)
        assert not got_bits[n:].any()            # padding bits 0
        assert (gotv[nb:] == 0xEE).all()         # nothing past the bitmap
        if not expect_flag:
            assert np.array_equal(np.unpackbits(O.take_bits(vbits, n_values, idx), bitorder="little")[:n], exp_bits)
    finally:
        p.set_tuning("gather_bucket", 2)


def test_host_take_of_an_array_with_nulls_uses_the_fused_call(ctx):
    import arrow_gpu_amd as ag

    dev, p = ctx
    rng = np.random.default_rng(5)
    n_values, n = 200_000, 150_000
    vals = [None if rng.random() < 0.3 else float(v) for v in rng.standard_normal(n_values).astype(np.float32)]
    a = ag.Float32ArrayGPU.from_optional_slice(vals, dev)
    idx_np = rng.integers(0, n_values, n).astype(np.uint32)
    idx = ag.UInt32ArrayGPU.from_slice(idx_np, dev)
    for mode in (1, 2):
        p.set_tuning("gather_bucket", mode)
        got = a.take_op(idx, p)
        p.finish()
        p.sync()
        assert got.values() == [vals[i] for i in idx_np]
    p.set_tuning("gather_bucket", 2)


# ---- round 3: Boolean take (agpu_take_bits) through the merge-back pipeline — the bitmap's words are the elements
@pytest.mark.parametrize("mode", [1, 2, 4])
@pytest.mark.parametrize("n,n_bits,dist", [(32768, 9, "uniform"), (32769, 70_001, "uniform"), (65_537, 1 << 22, "uniform"), (40_000, 300_000_007, "uniform"),
                                           (1_000_003, 40_000_003, "uniform"), (3 * 32768 + 5, 1 << 24, "oob_tile"), (500_000, 1 << 23, "skew"),
                                           (2_500_000, 1 << 20, "dups"), (2_000_000, (1 << 29) - 3, "uniform"), (100_000, (1 << 29) + 77, "uniform"),
                                           (1_500_000, 1_000_000_007, "uniform"), (400_000, 3_000_000_011, "uniform"), (300_000, 1_000_000_007, "skew")])
def test_take_bits_pipelines_equal_the_oracle(ctx, mode, n, n_bits, dist):
    """[ref: crates/routines/src/bool.rs:15-46 + bool/take.wgsl] direct (mode 1) and merge-back (mode 2; bitmaps over 2^29 bits — a 1e9-row
    column's — through the WIDE gather with larger regions) against oracle.take_bits; out-of-range rows read 0 and raise the sticky flag"""
    dev, p = ctx
    p.set_tuning("gather_bucket", mode)
    try:
        rng = np.random.default_rng(n * 13 + mode)
        bits = rng.integers(0, 256, (n_bits + 63) // 64 * 8, dtype=np.uint8)  # padding bits random: never addressed
        idx = rng.integers(0, n_bits, n).astype(np.uint32)
        expect_flag = False
        if dist == "skew":
            idx = np.where(rng.random(n) < 0.9, rng.integers(1000, 1064, n), idx).astype(np.uint32)
        elif dist == "dups":
            idx = (idx // 4096 * 4096).astype(np.uint32)
        elif dist == "oob_tile":
            idx[32768:65536] = rng.integers(n_bits, 1 << 32, 32768, dtype=np.uint64).astype(np.uint32)
            expect_flag = True
        db, di = dev.create_gpu_buffer_with_data(bits), dev.create_gpu_buffer_with_data(idx)
        nb = O.bitmap_bytes(n)
        outb = dev.create_empty_buffer(nb + 16)
        capi.call("agpu_memset", p._handle, vp(outb), 0xEE, nb + 16)
        capi.call("agpu_take_bits", p._handle, vp(db), n_bits, vp(di), vp(outb), n)
        if expect_flag:
            import arrow_gpu_amd as ag

            with pytest.raises(ag.ArrowErrorGPU):
                p.sync()
        else:
            p.sync()
        got = dev.retrive_data(outb, nb + 16, pipeline=p)
        ok = idx < n_bits
        src = np.unpackbits(bits, bitorder="little")
        exp = np.where(ok, src[np.minimum(idx, n_bits - 1)], 0).astype(np.uint8)
        got_bits = np.unpackbits(got[:nb], bitorder="little")
        assert np.array_equal(got_bits[:n], exp)
        assert not got_bits[n:].any()
        assert (got[nb:] == 0xEE).all()
        if not expect_flag:
            assert np.array_equal(np.unpackbits(O.take_bits(bits, n_bits, idx), bitorder="little")[:n], exp)
    finally:
        p.set_tuning("gather_bucket", 2)


def test_take_bits_at_2_28_rows_merge_back_equals_direct(ctx):
    """the size the auto policy serves: both forms give the same bitmap (and the same popcount as the host's gather)"""
    dev, p = ctx
    n = n_bits = 1 << 28
    rng = np.random.default_rng(77)
    bits = rng.integers(0, 256, n_bits // 8, dtype=np.uint8)
    idx = rng.integers(0, n_bits, n, dtype=np.uint32)
    db, di = dev.create_gpu_buffer_with_data(bits), dev.create_gpu_buffer_with_data(idx)
    outs = {}
    try:
        for mode in (1, 0):  # 0 = auto: the merge-back form at this size
            p.set_tuning("gather_bucket", mode)
            ob = dev.create_empty_buffer(n // 8)
            capi.call("agpu_take_bits", p._handle, vp(db), n_bits, vp(di), vp(ob), n)
            p.sync()
            outs[mode] = dev.retrive_data(ob, n // 8, pipeline=p)
    finally:
        p.set_tuning("gather_bucket", 2)
    assert np.array_equal(outs[0], outs[1])
    sample = slice(12_345_678, 12_345_678 + 1_000_000)
    src = np.unpackbits(bits, bitorder="little")
    assert np.array_equal(np.unpackbits(outs[0], bitorder="little")[sample], src[idx[sample]])


# ---- the put's pair pipeline with ragged tiles (dropped rows), 4- and 1-byte values, and the bucketed take beside it over the same columns
# (rounds 3-5 ran this over six ways of getting the pipeline's range starts — tuning gather_offsets — and the pair-pipeline take,
# gather_bucket = 3; round 6 removed the variants that never beat the default)
@pytest.mark.parametrize("width", [4, 1])
def test_pair_pipeline_with_dropped_rows_and_the_bucketed_take(ctx, width):
    dev, p = ctx
    try:
        rng = np.random.default_rng(21 + width)
        n, n_src, n_dst = 3 * 16384 + 77, 2_000_003, (1 << 21) + 9
        src = rng.integers(0, 1 << (8 * width), n_src, dtype=np.uint64).astype(NPW[width])
        dst = rng.integers(0, 1 << (8 * width), n_dst, dtype=np.uint64).astype(NPW[width])
        si = rng.integers(0, n_src, n).astype(np.uint32)
        di = rng.permutation(n_dst)[:n].astype(np.uint32)
        si[5::997] = n_src + 1  # dropped rows: the tiles behind P are ragged
        ok = si < n_src
        ds, dd = dev.create_gpu_buffer_with_data(src), dev.create_gpu_buffer_with_data(dst)
        dsi, ddi = dev.create_gpu_buffer_with_data(si), dev.create_gpu_buffer_with_data(di)
        p.set_tuning("gather_bucket", 2)
        capi.call("agpu_put_bounded", p._handle, width, vp(ds), n_src, vp(dsi), vp(dd), n_dst, vp(ddi), n)
        import arrow_gpu_amd as ag

        with pytest.raises(ag.ArrowErrorGPU):
            p.sync()
        got = dev.retrive_data(dd, n_dst * width, pipeline=p).view(NPW[width])
        exp = dst.copy()
        exp[di[ok]] = src[si[ok]]
        assert np.array_equal(got, exp)
        out = dev.create_empty_buffer(n * width + 16)
        capi.call("agpu_take", p._handle, width, vp(ds), n_src, vp(dsi), vp(out), n)
        with pytest.raises(ag.ArrowErrorGPU):
            p.sync()
        got = dev.retrive_data(out, n * width, pipeline=p).view(NPW[width])
        assert np.array_equal(got, np.where(ok, src[np.minimum(si, n_src - 1)], 0).astype(NPW[width]))
    finally:
        p.set_tuning("gather_bucket", 2)


# ---- round 3: Boolean put (agpu_put_bits_bounded) bucketed by destination region — LDS-resident regions, no global atomics
@pytest.mark.parametrize("mode", [1, 2, 4])
@pytest.mark.parametrize("n,n_src,n_dst,dist", [(5, 9, 77, "uniform"), (32768, 70_001, 40_000, "uniform"), (100_003, 1 << 20, (1 << 21) + 5, "uniform"),
                                                (1_000_003, 40_000_003, 3_000_017, "uniform"), (300_000, 1 << 22, (1 << 18) * 5 + 3, "oob"),
                                                (500_000, 1 << 23, 1 << 24, "skew"), (200_000, 1 << 20, 1_200_000_007, "uniform"),
                                                (150_000, 1 << 20, 2_147_000_001, "uniform")])  # 64 KiB and 128 KiB regions in LDS
def test_put_bits_pipelines_equal_the_oracle(ctx, mode, n, n_src, n_dst, dist):
    """[ref: crates/routines/src/bool.rs put + bool/put.wgsl] direct (mode 1: one atomic per row) and bucketed (mode 2) against
    oracle.put_bits with distinct destinations; rows with either index out of range are dropped and raise the sticky flag"""
    import arrow_gpu_amd as ag

    dev, p = ctx
    p.set_tuning("gather_bucket", mode)
    try:
        rng = np.random.default_rng(n * 17 + mode)
        src = rng.integers(0, 256, (n_src + 63) // 64 * 8, dtype=np.uint8)
        dst = rng.integers(0, 256, (n_dst + 63) // 64 * 8, dtype=np.uint8)
        si = rng.integers(0, n_src, n).astype(np.uint32)
        if n_dst > 100_000_000:  # distinct destinations without a 1.2 G-entry permutation
            di = np.unique(rng.integers(0, n_dst, 2 * n).astype(np.uint32))[:n]
            rng.shuffle(di)
            assert len(di) == n
        else:
            di = rng.permutation(n_dst)[:n].astype(np.uint32)
        if dist == "skew":  # most rows land in one destination region
            di = np.where(rng.random(n) < 0.9, rng.permutation(1 << 18)[:n] if n <= (1 << 18) else di, di).astype(np.uint32)
            di = np.unique(di)
            rng.shuffle(di)
            si = si[: len(di)]
        ok = np.ones(len(di), bool)
        if dist == "oob":
            si[::101] = n_src + 5
            di[::103] = n_dst
            ok = (si < n_src) & (di < n_dst)
        k = len(di)
        ds, dd = dev.create_gpu_buffer_with_data(src), dev.create_gpu_buffer_with_data(np.concatenate([dst, np.full(16, 0xEE, np.uint8)]))
        dsi, ddi = dev.create_gpu_buffer_with_data(si), dev.create_gpu_buffer_with_data(di)
        capi.call("agpu_put_bits_bounded", p._handle, vp(ds), n_src, vp(dsi), vp(dd), n_dst, vp(ddi), k)
        if dist == "oob":
            with pytest.raises(ag.ArrowErrorGPU):
                p.sync()
        else:
            p.sync()
        got = dev.retrive_data(dd, len(dst) + 16, pipeline=p)
        exp = O.put_bits(src, si[ok], dst, di[ok])
        assert np.array_equal(got[: len(dst)], exp)
        assert (got[len(dst):] == 0xEE).all()
    finally:
        p.set_tuning("gather_bucket", 2)


def test_put_bits_at_2_27_rows_bucketed_equals_direct(ctx):
    dev, p = ctx
    n, n_src, n_dst = 1 << 27, 1 << 28, 1 << 28
    rng = np.random.default_rng(99)
    src = rng.integers(0, 256, n_src // 8, dtype=np.uint8)
    dst = rng.integers(0, 256, n_dst // 8, dtype=np.uint8)
    si = rng.integers(0, n_src, n, dtype=np.uint32)
    di = rng.permutation(n_dst).astype(np.uint32)[:n]
    ds, dsi, ddi = dev.create_gpu_buffer_with_data(src), dev.create_gpu_buffer_with_data(si), dev.create_gpu_buffer_with_data(di)
    outs = {}
    try:
        for mode in (1, 0):  # 0 = auto: bucketed at this size
            p.set_tuning("gather_bucket", mode)
            dd = dev.create_gpu_buffer_with_data(dst)
            capi.call("agpu_put_bits_bounded", p._handle, vp(ds), n_src, vp(dsi), vp(dd), n_dst, vp(ddi), n)
            p.sync()
            outs[mode] = dev.retrive_data(dd, n_dst // 8, pipeline=p)
    finally:
        p.set_tuning("gather_bucket", 2)
    assert np.array_equal(outs[0], outs[1])
    sb, db = np.unpackbits(src, bitorder="little"), np.unpackbits(dst, bitorder="little")
    db[di] = sb[si]
    assert np.array_equal(outs[0], np.packbits(db, bitorder="little"))


# ---- round 3: 1- and 2-byte takes through the merge-back pipeline too (with and without the source's validity)
@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("width", [1, 2])
@pytest.mark.parametrize("n,n_values,dist", [(32768, 9, "uniform"), (32769, 70_001, "uniform"), (300_001, 3_000_017, "uniform"), (40_000, 300_000_007, "uniform"),
                                             (3 * 32768 + 5, 1 << 24, "oob_tile"), (500_000, 1 << 23, "skew"), (1_000_003, 1 << 20, "dups"),
                                             (700_001, 600_000_007, "uniform"), (300_001, 1_200_000_011, "uniform")])  # > 4095 · 2^17 / 2^18 elements: 2^18- / 2^19-element regions (the 8-slot gather)
def test_narrow_takes_through_both_pipelines(ctx, mode, width, n, n_values, dist):
    import arrow_gpu_amd as ag

    dev, p = ctx
    # the very large 2-byte sources (2^18- / 2^19-element regions, the 8-slot gather) are built ON THE DEVICE — no gigabytes of host data:
    # values[j] = half (j & 1) of the u32 word (j >> 1) · 2654435761, validity from the counter-based generator the oracle reproduces
    on_device = n_values > 500_000_000 and width == 2
    p.set_tuning("gather_bucket", mode)
    try:
        rng = np.random.default_rng(n * 19 + mode + width)
        idx = rng.integers(0, n_values, n).astype(np.uint32)
        expect_flag = False
        if dist == "skew":
            idx = np.where(rng.random(n) < 0.9, rng.integers(1000, 1064, n), idx).astype(np.uint32)
        elif dist == "dups":
            idx = (idx // 4096 * 4096).astype(np.uint32)
        elif dist == "oob_tile":
            idx[32768:65536] = rng.integers(n_values, 1 << 32, 32768, dtype=np.uint64).astype(np.uint32)
            expect_flag = True
        ok = idx < n_values
        if on_device:
            words = (n_values + 1) // 2
            iota = dev.create_empty_buffer(4 * words)
            seed_rows = 1 << 20
            small = dev.create_gpu_buffer_with_data(np.arange(seed_rows, dtype=np.uint32))
            capi.call("agpu_copy", p._handle, vp(iota), vp(small), 4 * seed_rows)
            filled = seed_rows
            while filled < words:
                k = min(filled, words - filled)
                cur = dev.create_gpu_buffer_with_data(np.array([filled], np.uint32))
                capi.call("agpu_scalar", p._handle, capi.OP_ADD, capi.U32, vp(iota), vp(cur), C.c_void_p(iota.ptr + 4 * filled), k)
                p.sync()
                filled += k
            dv = dev.create_empty_buffer(4 * words)
            mul = dev.create_gpu_buffer_with_data(np.array([2654435761], np.uint32))
            capi.call("agpu_scalar", p._handle, capi.OP_MUL, capi.U32, vp(iota), vp(mul), vp(dv), words)
            dvb = dev.create_empty_buffer(O.bitmap_bytes(n_values) + 64)
            capi.call("agpu_synth_bits", p._handle, vp(dvb), n_values, 77, 0, C.c_double(0.7))
            p.sync()
            del iota
            word = (idx >> np.uint32(1)) * np.uint32(2654435761)
            exp = ((word >> ((idx & np.uint32(1)) * np.uint32(16))) & np.uint32(0xFFFF)).astype(np.uint16)
            exp_bits = np.unpackbits(O.take_bits(O.synth_bits(n_values, 77, 0, 0.7), n_values, idx), bitorder="little")[:n]
        else:
            values = rng.integers(0, 1 << (8 * width), n_values, dtype=np.uint64).astype(NPW[width])
            vbits = np.packbits(rng.random((n_values + 63) // 64 * 64) < 0.7, bitorder="little")
            dv, dvb = dev.create_gpu_buffer_with_data(values), dev.create_gpu_buffer_with_data(vbits)
            exp = np.where(ok, values[np.minimum(idx, n_values - 1)], 0).astype(NPW[width])
            exp_bits = np.where(ok, np.unpackbits(vbits, bitorder="little")[np.minimum(idx, n_values - 1)], 0).astype(np.uint8)
        di = dev.create_gpu_buffer_with_data(idx)
        nb = O.bitmap_bytes(n)
        for with_validity in (False, True):
            out, outv = dev.create_empty_buffer(width * n + 16), dev.create_empty_buffer(nb + 16)
            capi.call("agpu_memset", p._handle, vp(out), 0xEE, width * n + 16)
            capi.call("agpu_memset", p._handle, vp(outv), 0xEE, nb + 16)
            if with_validity:
                capi.call("agpu_take_validity", p._handle, width, vp(dv), n_values, vp(dvb), vp(di), vp(out), vp(outv), n)
            else:
                capi.call("agpu_take", p._handle, width, vp(dv), n_values, vp(di), vp(out), n)
            if expect_flag:
                with pytest.raises(ag.ArrowErrorGPU):
                    p.sync()
            else:
                p.sync()
            got = dev.retrive_data(out, width * n + 16, pipeline=p)
            assert np.array_equal(got[: width * n].view(NPW[width]), exp)
            assert (got[width * n:] == 0xEE).all()
            if with_validity:
                gotv = dev.retrive_data(outv, nb + 16, pipeline=p)
                gb = np.unpackbits(gotv[:nb], bitorder="little")
                assert np.array_equal(gb[:n], exp_bits) and not gb[n:].any() and (gotv[nb:] == 0xEE).all()
        if not expect_flag and not on_device:
            assert np.array_equal(O.take(width, values, idx), exp)
    finally:
        p.set_tuning("gather_bucket", 2)


def test_bucketed_pipelines_from_several_threads(ctx):
    """four host threads, each with its own pipeline, run the forced pipelines (take, take with validity, Boolean take / put, put) over
    their own columns at the same time: the temporaries come from one pool and go back to it mid-stream — results stay the oracle's"""
    import threading

    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline

    dev, _ = ctx
    errors = []

    def worker(tid):
        try:
            p = ArrowComputePipeline(dev, f"bucketed-{tid}")
            p.set_tuning("gather_bucket", 2)
            rng = np.random.default_rng(100 + tid)
            n, n_src, n_dst = 200_003 + 4096 * tid, 1 << 21, (1 << 21) + 12345
            values = rng.integers(0, 1 << 32, n_src, dtype=np.uint64).astype(np.uint32)
            vbits = np.packbits(rng.random((n_src + 63) // 64 * 64) < 0.5, bitorder="little")
            dst = rng.integers(0, 1 << 32, n_dst, dtype=np.uint64).astype(np.uint32)
            dbits = np.packbits(rng.random((n_dst + 63) // 64 * 64) < 0.5, bitorder="little")
            si = rng.integers(0, n_src, n).astype(np.uint32)
            di = rng.permutation(n_dst)[:n].astype(np.uint32)
            dv, dvb, dsi, ddi = (dev.create_gpu_buffer_with_data(x) for x in (values, vbits, si, di))
            nb = O.bitmap_bytes(n)
            exp_take, exp_tbits = values[si], O.take_bits(vbits, n_src, si)
            exp_put = dst.copy()
            exp_put[di] = values[si]
            exp_pbits = O.put_bits(vbits, si, dbits, di)
            for _ in range(5):
                out, outv = dev.create_empty_buffer(4 * n + 16), dev.create_empty_buffer(nb + 16)
                capi.call("agpu_take_validity", p._handle, 4, vp(dv), n_src, vp(dvb), vp(dsi), vp(out), vp(outv), n)
                ob = dev.create_empty_buffer(nb + 16)
                capi.call("agpu_take_bits", p._handle, vp(dvb), n_src, vp(dsi), vp(ob), n)
                dd, ddb = dev.create_gpu_buffer_with_data(dst), dev.create_gpu_buffer_with_data(dbits)
                capi.call("agpu_put_bounded", p._handle, 4, vp(dv), n_src, vp(dsi), vp(dd), n_dst, vp(ddi), n)
                capi.call("agpu_put_bits_bounded", p._handle, vp(dvb), n_src, vp(dsi), vp(ddb), n_dst, vp(ddi), n)
                p.sync()
                assert np.array_equal(dev.retrive_data(out, 4 * n, pipeline=p).view(np.uint32), exp_take)
                assert np.array_equal(dev.retrive_data(outv, nb, pipeline=p), exp_tbits)
                assert np.array_equal(dev.retrive_data(ob, nb, pipeline=p), exp_tbits)
                assert np.array_equal(dev.retrive_data(dd, 4 * n_dst, pipeline=p).view(np.uint32), exp_put)
                assert np.array_equal(dev.retrive_data(ddb, len(dbits), pipeline=p), exp_pbits)
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


# ---- round 3: the auto policy decides on the DEVICE (locality probe) between the pipeline and the direct kernel
@pytest.mark.parametrize("dist", ["uniform", "sorted", "hot_window", "half_sorted_half_random"])
def test_auto_policy_is_right_either_way(ctx, dist):
    """2^25 rows under the auto policy: whichever form the probe picks (random indices → the pipeline's kernels run and the direct
    kernel behind them returns at once; sorted / clustered → the other way round), values, validity and Boolean takes equal numpy's"""
    dev, p = ctx
    n, n_src = 1 << 25, 1 << 27
    rng = np.random.default_rng(31)
    values = rng.integers(0, 1 << 32, n_src, dtype=np.uint64).astype(np.uint32)
    vbits = rng.integers(0, 256, n_src // 8, dtype=np.uint8)
    if dist == "uniform":
        idx = rng.integers(0, n_src, n, dtype=np.uint32)
    elif dist == "sorted":
        idx = np.sort(rng.integers(0, n_src, n, dtype=np.uint32))
    elif dist == "hot_window":
        idx = np.where(rng.random(n) < 0.9, rng.integers(5000, 5064, n), rng.integers(0, n_src, n)).astype(np.uint32)
    else:
        idx = np.concatenate([np.sort(rng.integers(0, n_src, n // 2, dtype=np.uint32)), rng.integers(0, n_src, n // 2, dtype=np.uint32)])
    dv, dvb, di = dev.create_gpu_buffer_with_data(values), dev.create_gpu_buffer_with_data(vbits), dev.create_gpu_buffer_with_data(idx)
    src_bits = np.unpackbits(vbits, bitorder="little")
    exp, exp_bits = values[idx], np.packbits(src_bits[idx], bitorder="little")
    try:
        p.set_tuning("gather_bucket", 0)
        out, outv, ob = dev.create_empty_buffer(4 * n), dev.create_empty_buffer(n // 8), dev.create_empty_buffer(n // 8)
        capi.call("agpu_take_validity", p._handle, 4, vp(dv), n_src, vp(dvb), vp(di), vp(out), vp(outv), n)
        capi.call("agpu_take_bits", p._handle, vp(dvb), n_src, vp(di), vp(ob), n)
        out2 = dev.create_empty_buffer(4 * n)
        capi.call("agpu_take", p._handle, 4, vp(dv), n_src, vp(di), vp(out2), n)
        p.sync()
        assert np.array_equal(dev.retrive_data(out, 4 * n, pipeline=p).view(np.uint32), exp)
        assert np.array_equal(dev.retrive_data(out2, 4 * n, pipeline=p).view(np.uint32), exp)
        assert np.array_equal(dev.retrive_data(outv, n // 8, pipeline=p), exp_bits)
        assert np.array_equal(dev.retrive_data(ob, n // 8, pipeline=p), exp_bits)
    finally:
        p.set_tuning("gather_bucket", 2)


@pytest.mark.parametrize("src_dist,dst_dist", [("uniform", "uniform"), ("sorted", "sorted"), ("sequential", "sequential"), ("sorted", "uniform")])
def test_put_auto_policy_is_right_either_way(ctx, src_dist, dst_dist):
    """2^24 rows under the auto policy (put probes both index columns, the Boolean put its source side): whichever kernels the device-side
    decision lets run, values and bits equal numpy's (distinct destinations)"""
    dev, p = ctx
    n, n_src, n_dst = 1 << 24, 1 << 26, 1 << 26
    rng = np.random.default_rng(47)
    src = rng.integers(0, 1 << 32, n_src, dtype=np.uint64).astype(np.uint32)
    dst = rng.integers(0, 1 << 32, n_dst, dtype=np.uint64).astype(np.uint32)
    sbits, dbits = rng.integers(0, 256, n_src // 8, dtype=np.uint8), rng.integers(0, 256, n_dst // 8, dtype=np.uint8)
    si = {"uniform": lambda: rng.integers(0, n_src, n, dtype=np.uint32), "sorted": lambda: np.sort(rng.integers(0, n_src, n, dtype=np.uint32)),
          "sequential": lambda: np.arange(n, dtype=np.uint32)}[src_dist]()
    perm = rng.permutation(n_dst).astype(np.uint32)[:n]
    di = {"uniform": lambda: perm, "sorted": lambda: np.sort(perm), "sequential": lambda: np.arange(n, dtype=np.uint32) + 12345}[dst_dist]()
    ds, dsb, dsi, ddi = (dev.create_gpu_buffer_with_data(x) for x in (src, sbits, si, di))
    dd, ddb = dev.create_gpu_buffer_with_data(dst), dev.create_gpu_buffer_with_data(dbits)
    try:
        p.set_tuning("gather_bucket", 0)
        capi.call("agpu_put_bounded", p._handle, 4, vp(ds), n_src, vp(dsi), vp(dd), n_dst, vp(ddi), n)
        capi.call("agpu_put_bits_bounded", p._handle, vp(dsb), n_src, vp(dsi), vp(ddb), n_dst, vp(ddi), n)
        p.sync()
    finally:
        p.set_tuning("gather_bucket", 2)
    exp = dst.copy()
    exp[di] = src[si]
    assert np.array_equal(dev.retrive_data(dd, 4 * n_dst, pipeline=p).view(np.uint32), exp)
    eb = np.unpackbits(dbits, bitorder="little")
    eb[di] = np.unpackbits(sbits, bitorder="little")[si]
    assert np.array_equal(dev.retrive_data(ddb, n_dst // 8, pipeline=p), np.packbits(eb, bitorder="little"))


@pytest.mark.parametrize("corner", ["source_local", "destination_local"])
def test_put_with_one_local_column_at_2_26_rows(ctx, corner):
    """the two extra forms of a put under the auto policy (from 2^26 rows): source column local → the destination-only pipeline;
    destination column local → the take's merge-back pipeline storing through the destination column.  Distinct destinations."""
    dev, p = ctx
    n, n_src, n_dst = 1 << 26, 1 << 27, 1 << 27
    rng = np.random.default_rng(53)
    src = rng.integers(0, 1 << 32, n_src, dtype=np.uint64).astype(np.uint32)
    dst = rng.integers(0, 1 << 32, n_dst, dtype=np.uint64).astype(np.uint32)
    i = np.arange(n, dtype=np.uint64)
    if corner == "source_local":
        si = (i + 777).astype(np.uint32)
        di = ((i * 0x9E3779B1) % n_dst).astype(np.uint32)  # an odd multiplier: a bijection on 2^27 — distinct, and no two neighbours share a line
    else:
        si = rng.integers(0, n_src, n, dtype=np.uint32)
        di = (i * 2 + (i & 1)).astype(np.uint32)
    si[::1_000_003] = n_src + 9  # a few rows with a source index out of range: dropped, sticky flag
    ok = si < n_src
    ds, dd, dsi, ddi = (dev.create_gpu_buffer_with_data(x) for x in (src, dst, si, di))
    import arrow_gpu_amd as ag

    try:
        p.set_tuning("gather_bucket", 0)
        capi.call("agpu_put_bounded", p._handle, 4, vp(ds), n_src, vp(dsi), vp(dd), n_dst, vp(ddi), n)
        with pytest.raises(ag.ArrowErrorGPU):
            p.sync()
    finally:
        p.set_tuning("gather_bucket", 2)
    exp = dst.copy()
    exp[di[ok]] = src[si[ok]]
    assert np.array_equal(dev.retrive_data(dd, 4 * n_dst, pipeline=p).view(np.uint32), exp)


def test_auto_policy_on_a_busy_stream_falls_back_to_the_gated_forms(ctx):
    """the host waits 150 µs for the probe's answer; with milliseconds of work queued ahead on the same stream it cannot come in time, so every
    form is enqueued and the device-side copy of the decision picks — same results (sorted and random index columns, take and put)"""
    dev, p = ctx
    n, n_src = 1 << 25, 1 << 27
    rng = np.random.default_rng(61)
    values = rng.integers(0, 1 << 32, n_src, dtype=np.uint64).astype(np.uint32)
    dv = dev.create_gpu_buffer_with_data(values)
    ballast = dev.create_empty_buffer(1 << 32)
    try:
        p.set_tuning("gather_bucket", 0)
        for dist in ("sorted", "uniform"):
            idx = rng.integers(0, n_src, n, dtype=np.uint32)
            if dist == "sorted":
                idx = np.sort(idx)
            di = dev.create_gpu_buffer_with_data(idx)
            out = dev.create_empty_buffer(4 * n)
            p.sync()
            for _ in range(6):  # ≈ 4 ms of fills ahead of the take
                capi.call("agpu_memset", p._handle, vp(ballast), 0x5A, 1 << 32)
            capi.call("agpu_take", p._handle, 4, vp(dv), n_src, vp(di), vp(out), n)
            p.sync()
            assert np.array_equal(dev.retrive_data(out, 4 * n, pipeline=p).view(np.uint32), values[idx])
    finally:
        p.set_tuning("gather_bucket", 2)
