"""GPU: Arrow ingest/egress (arrow_gpu_amd.interop) and an INDEPENDENT semantic cross-check of the kernels against
pyarrow.compute (Arrow C++), on random arrays with nulls, slices with bit offsets and chunked arrays.  pyarrow is not
the reference (that is psvri/arrow-gpu), it is the third-party implementation of the same Arrow semantics SURVEY §8c
names as the auxiliary cross-check."""
import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")
pc = pytest.importorskip("pyarrow.compute")

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(20250418)


def rand_pa(n, typ, null_frac=0.2, lo=-100, hi=100):
    mask = RNG.random(n) < null_frac
    if pa.types.is_floating(typ):
        vals = (RNG.standard_normal(n) * 50).astype(np.float32)
    elif pa.types.is_boolean(typ):
        vals = RNG.random(n) < 0.5
    else:
        info = np.iinfo(typ.to_pandas_dtype())
        vals = RNG.integers(max(lo, info.min), min(hi, info.max) + 1, n).astype(typ.to_pandas_dtype())
    return pa.array(vals, type=typ, mask=mask)


def same(gpu_arr, expected: pa.Array, ulp=0):
    got = gpu_arr.to_arrow()
    assert got.type == expected.type, (got.type, expected.type)
    assert len(got) == len(expected)
    assert got.is_valid().equals(expected.is_valid())
    g = got.filter(got.is_valid())
    e = expected.filter(expected.is_valid())
    if ulp and pa.types.is_floating(got.type):
        gi = g.to_numpy(zero_copy_only=False).view(np.int32).astype(np.int64)
        ei = e.to_numpy(zero_copy_only=False).view(np.int32).astype(np.int64)
        gi = np.where(gi < 0, -(2**31) - gi, gi)
        ei = np.where(ei < 0, -(2**31) - ei, ei)
        assert np.abs(gi - ei).max(initial=0) <= ulp
    else:
        assert g.equals(e), (g[:10], e[:10])


@pytest.mark.parametrize("typ", [pa.float32(), pa.int32(), pa.uint32(), pa.int16(), pa.uint16(), pa.int8(), pa.uint8(),
                                 pa.date32(), pa.bool_()])
def test_roundtrip_with_slices_and_nulls(ag, typ):
    dev = ag.GPU_DEVICE()
    for n in (0, 1, 7, 64, 1000, 100_003):
        base = rand_pa(n + 37, typ) if typ != pa.date32() else rand_pa(n + 37, pa.int32()).cast(pa.date32())
        for off in (0, 1, 13, 37):
            arr = base.slice(off, n)                       # Arrow `offset` ≠ 0: values AND bitmap start mid-buffer
            g = ag.from_arrow(arr, dev)
            back = g.to_arrow()
            assert back.equals(arr), (typ, n, off)
            assert pa.array(g).equals(arr)                 # __arrow_c_array__ (C Data Interface capsule)
        no_nulls = rand_pa(n, typ if typ != pa.date32() else pa.int32(), null_frac=0.0)
        if typ == pa.date32():
            no_nulls = no_nulls.cast(pa.date32())
        g = ag.from_arrow(no_nulls, dev)
        assert g.null_buffer is None and g.to_arrow().equals(no_nulls)


def test_chunked_ingest(ag):
    dev = ag.GPU_DEVICE()
    chunks = [rand_pa(n, pa.float32()) for n in (1000, 1, 4097)]
    gs = ag.from_arrow_chunked(pa.chunked_array(chunks), dev)
    assert [g.len for g in gs] == [1000, 1, 4097]
    for g, c in zip(gs, chunks):
        assert g.to_arrow().equals(c)


def test_pinned_staging_async_roundtrip(ag):
    import ctypes as C

    from arrow_gpu_amd import _capi as capi

    dev = ag.GPU_DEVICE()
    n = 1 << 22
    st_in, st_out = ag.PinnedStaging(dev, 4 * n), ag.PinnedStaging(dev, 4 * n)
    src = RNG.standard_normal(n).astype(np.float32)
    st_in.view[:] = src.view(np.uint8)
    p = ag.ArrowComputePipeline(dev, "staging")
    a, out = dev.create_empty_buffer(4 * n), dev.create_empty_buffer(4 * n)
    st_in.upload(p, a, 4 * n)                               # H2D, kernel and D2H all queued, one sync
    capi.call("agpu_unary", p._handle, capi.UN_NEG, capi.F32, C.c_void_p(a.ptr), C.c_void_p(out.ptr), n)
    st_out.download(p, out, 4 * n)
    p.sync()
    assert np.array_equal(st_out.view.view(np.float32), -src)


def test_kernels_agree_with_pyarrow_compute(ag):
    dev = ag.GPU_DEVICE()
    n = 200_003
    a, b = rand_pa(n, pa.float32()), rand_pa(n, pa.float32())
    ga, gb = ag.from_arrow(a, dev), ag.from_arrow(b, dev)
    same(ga.add(gb), pc.add(a, b))
    same(ga.sub(gb), pc.subtract(a, b))
    same(ga.mul(gb), pc.multiply(a, b))
    nz = pc.if_else(pc.equal(b, 0), pa.scalar(1.0, pa.float32()), b)
    same(ga.div(ag.from_arrow(nz, dev)), pc.divide(a, nz))
    same(ga.neg(), pc.negate(a))
    same(ga.abs(), pc.abs(a))
    same(ga.abs().sqrt(), pc.sqrt(pc.abs(a)))
    same(ga.sin(), pc.sin(a), ulp=1)
    same(ga.cos(), pc.cos(a), ulp=1)
    same(ga.max(gb), pc.max_element_wise(a, b, skip_nulls=False))
    same(ga.min(gb), pc.min_element_wise(a, b, skip_nulls=False))
    same(ga.gt(gb), pc.greater(a, b))
    same(ga.lteq(gb), pc.less_equal(a, b))

    ia, ib = rand_pa(n, pa.int32(), lo=-5, hi=5), rand_pa(n, pa.int32(), lo=-5, hi=5)
    gia, gib = ag.from_arrow(ia, dev), ag.from_arrow(ib, dev)
    same(gia.add(gib), pc.add(ia, ib))
    same(gia.eq(gib), pc.equal(ia, ib))
    same(gia.lt(gib), pc.less(ia, ib))
    same(gia.gteq(gib), pc.greater_equal(ia, ib))
    same(gia.bitwise_and(gib), pc.bit_wise_and(ia, ib))
    same(gia.bitwise_xor(gib), pc.bit_wise_xor(ia, ib))
    same(gia.bitwise_not(), pc.bit_wise_not(ia))
    big = pa.array(RNG.integers(-2**31, 2**31, n), pa.int32())
    same(ag.from_arrow(big, dev).add(ag.from_arrow(big, dev)), pc.add(big, big))   # wrapping, like pc.add (unchecked)

    u8 = rand_pa(n, pa.uint8(), lo=0, hi=255)
    gu8 = ag.from_arrow(u8, dev)
    same(gu8.cast(ag.Float32ArrayGPU), pc.cast(u8, pa.float32()))
    same(gu8.cast(ag.UInt32ArrayGPU), pc.cast(u8, pa.uint32()))
    same(gu8.cast(ag.Int16ArrayGPU), pc.cast(u8, pa.int16()))
    i8 = rand_pa(n, pa.int8(), lo=-128, hi=127)
    same(ag.from_arrow(i8, dev).cast(ag.Int32ArrayGPU), pc.cast(i8, pa.int32()))
    same(gu8.sin(), pc.sin(pc.cast(u8, pa.float32())), ulp=1)

    ba, bb = rand_pa(n, pa.bool_()), rand_pa(n, pa.bool_())
    gba, gbb = ag.from_arrow(ba, dev), ag.from_arrow(bb, dev)
    same(gba.bitwise_and(gbb), pc.and_(ba, bb))            # non-Kleene: null if either side is null, like the reference
    same(gba.bitwise_or(gbb), pc.or_(ba, bb))
    same(gba.bitwise_xor(gbb), pc.xor(ba, bb))
    same(gba.bitwise_not(), pc.invert(ba))
    same(gba.cast(ag.Float32ArrayGPU), pc.cast(ba, pa.float32()))

    # merge == if_else (null mask ⇒ null), take == take
    same(ga.merge(gb, gba), pc.if_else(ba, a, b))
    same(gia.merge(gib, gba), pc.if_else(ba, ia, ib))
    idx = pa.array(RNG.integers(0, n, 50_000).astype(np.uint32))
    gidx = ag.from_arrow(idx, dev)
    same(ga.take(gidx), pc.take(a, idx))
    same(gba.take(gidx), pc.take(ba, idx))

    # reductions: min/max exact; f32 sum vs Arrow's f64-accumulated sum within the pairwise-summation error bound
    dense = pa.array((RNG.standard_normal(n) * 10).astype(np.float32))
    gd = ag.from_arrow(dense, dev)
    s = float(gd.sum().raw_values()[0])
    ref = pc.sum(dense).as_py()
    assert abs(s - ref) <= 1e-6 * float(np.sum(np.abs(dense.to_numpy())))
    import ctypes as C

    from arrow_gpu_amd import _capi as capi

    p = ag.ArrowComputePipeline(dev, "minmax")
    out = dev.create_empty_buffer(16)
    mm = pc.min_max(dense)
    for op, want in ((capi.RED_MIN, mm["min"].as_py()), (capi.RED_MAX, mm["max"].as_py())):
        capi.call("agpu_reduce", p._handle, op, capi.F32, C.c_void_p(gd.data.ptr), None, n, C.c_void_p(out.ptr))
        assert float(dev.retrive_data(out, 4, pipeline=p).view(np.float32)[0]) == want
    # null-aware variants against Arrow (nulls skipped)
    capi.call("agpu_reduce", p._handle, capi.RED_MAX, capi.F32, C.c_void_p(ga.data.ptr), C.c_void_p(ga.null_buffer.bit_buffer.ptr),
              n, C.c_void_p(out.ptr))
    assert float(dev.retrive_data(out, 4, pipeline=p).view(np.float32)[0]) == pc.max(a).as_py()
    assert gba.count_set_bits() == pc.sum(pc.cast(pa.Array.from_buffers(pa.bool_(), n, [None, ba.buffers()[1]], offset=ba.offset), pa.int64())).as_py()
