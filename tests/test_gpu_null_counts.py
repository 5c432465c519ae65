"""GPU: null counts / set-bit counts as a BY-PRODUCT of the kernels that produce a bitmap (north_star "wavefront ballot/popc
for null counts") — agpu_compare_validity_count, agpu_bitmap_binary_count,
agpu_bitmap_merge_validity_count — and the stand-alone agpu_bitmap_popcount (one-wave blocks + fold), bit-exact against the
oracle's popcount of the oracle's bitmaps, with dirty padding bits, ragged sizes and every kernel variant.
[ref: the reference counts with a second pass — countob + Sum, crates/logical/src/boolean.rs:120-146]"""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi
from gpu_util import Dev, rand_values

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 63, 64, 65, 1023, 1024, 1025, 4096 * 8 + 5, 262_144, 300_001, 1_048_576, 1_048_576 + 64 * 3 + 7, 5_000_000 + 13]


@pytest.fixture(scope="module")
def D():
    return Dev()


def dirty_bitmap(n, seed, p_set=0.8):
    """a bitmap of n bits in whole 8-byte words whose PADDING bits (past n) are all set: they must never be counted"""
    rng = np.random.default_rng(seed * 1000003 + n)
    nbytes = max(O.bitmap_bytes(n), 8)
    full = np.ones(nbytes * 8, np.uint8)
    full[:n] = rng.random(n) < p_set
    return np.packbits(full, bitorder="little")


def u64(D, ptr):
    return int(D.down(ptr, np.uint64, 1)[0])


@pytest.mark.parametrize("n", SIZES)
def test_popcount_standalone_matches_the_oracle(D, n):
    b = dirty_bitmap(n, 3)
    out = D.empty(8)
    D.call("agpu_bitmap_popcount", D.up(b).vp, n, out.vp)
    assert u64(D, out) == O.bitmap_popcount(b, n)
    # unaligned-by-8 start: the element-granular path
    if n > 200:
        shifted = np.concatenate([np.zeros(8, np.uint8), b])
        D.call("agpu_bitmap_popcount", C.c_void_p(D.up(shifted).vp.value + 8), n, out.vp)
        assert u64(D, out) == O.bitmap_popcount(b, n)
    D.release()


@pytest.mark.parametrize("n", SIZES)
def test_bitmap_binary_count_is_the_popcount_of_the_result(D, n):
    a, b = dirty_bitmap(n, 5, 0.9), dirty_bitmap(n, 6, 0.7)
    for op, oop in ((capi.OP_AND, O.OP_AND), (capi.OP_OR, O.OP_OR), (capi.OP_XOR, O.OP_XOR)):
        out, cnt = D.empty(O.bitmap_bytes(n) + 8), D.empty(8, fill=0xEE)
        D.call("agpu_bitmap_binary_count", op, D.up(a).vp, D.up(b).vp, out.vp, n, cnt.vp)
        exp = O.bitmap_binary(oop, a, b, n)
        assert np.array_equal(D.down(out, np.uint8, (n + 7) // 8), exp[: (n + 7) // 8])
        assert u64(D, cnt) == O.bitmap_popcount(exp, n), (op, n)
    D.release()


@pytest.mark.parametrize("n", [0, 5, 64, 4097, 300_001, 2_000_003])
def test_merge_validity_count(D, n):
    va, vb, m, vm = (dirty_bitmap(n, s, p) for s, p in ((1, 0.9), (2, 0.8), (3, 0.5), (4, 0.95)))
    for use in ((1, 1, 1), (1, 0, 0), (0, 1, 1), (0, 0, 1)):
        out, cnt = D.empty(O.bitmap_bytes(n) + 8), D.empty(8, fill=0xEE)
        D.call("agpu_bitmap_merge_validity_count", D.up(va).vp if use[0] else None, D.up(vb).vp if use[1] else None, D.up(m).vp,
               D.up(vm).vp if use[2] else None, out.vp, n, cnt.vp)
        exp = O.merge_validity(va if use[0] else None, vb if use[1] else None, m, vm if use[2] else None, n)
        assert np.array_equal(D.down(out, np.uint8, (n + 7) // 8), exp[: (n + 7) // 8])
        assert u64(D, cnt) == O.bitmap_popcount(exp, n), (use, n)
    D.release()


@pytest.mark.parametrize("dtype", [capi.I32, capi.F32, capi.U8, capi.I16])
@pytest.mark.parametrize("n", SIZES)
def test_compare_validity_null_count(D, dtype, n):
    """ballot variant (32-bit) and vector variant (sub-word): the validity blocks of the launch count what they store"""
    a, b = rand_values(dtype, n, 1), rand_values(dtype, n, 2)
    va, vb = dirty_bitmap(n, 7, 0.9), dirty_bitmap(n, 8, 0.9)
    nb = O.bitmap_bytes(n) + 8
    for use in ((1, 1), (1, 0), (0, 1), (0, 0)):
        ob, ov, cnt = D.empty(nb), D.empty(nb), D.empty(8, fill=0xEE)
        D.call("agpu_compare_validity_count", capi.CMP_LT, dtype, D.up(a).vp, D.up(b).vp, D.up(va).vp if use[0] else None,
               D.up(vb).vp if use[1] else None, ob.vp, ov.vp, n, cnt.vp)
        full = n // 8
        assert np.array_equal(D.down(ob, np.uint8, (n + 7) // 8)[:full], O.compare(O.CMP_LT, dtype, a, b)[:full])
        if use == (0, 0):
            assert u64(D, cnt) == 0  # (None, None) → None: no nulls
            continue
        ev = O.validity_and(va if use[0] else None, vb if use[1] else None, n)
        assert np.array_equal(D.down(ov, np.uint8, (n + 7) // 8)[:full], ev[:full])
        assert u64(D, cnt) == n - O.bitmap_popcount(ev, n), (dtype, n, use)
    D.release()


def test_compare_count_under_a_capped_grid_and_sweep_tunings(D):
    """grid-stride launches (the stream_grid tuning) visit the same VIRTUAL validity blocks: same count"""
    n = 3_000_000 + 77
    a, b = rand_values(capi.I32, n, 1), rand_values(capi.I32, n, 2)
    va, vb = dirty_bitmap(n, 7, 0.9), dirty_bitmap(n, 8, 0.9)
    want = n - O.bitmap_popcount(O.validity_and(va, vb, n), n)
    nb = O.bitmap_bytes(n) + 8
    da, db, dva, dvb = D.up(a), D.up(b), D.up(va), D.up(vb)
    for key, val in (("stream_grid", 7), ("stream_grid", 1024), ("stream_grid", 512), ("cmp_variant", 1)):
        capi.call("agpu_pipeline_set_tuning", D.p._h, key.encode(), val)
        ob, ov, cnt = D.empty(nb), D.empty(nb), D.empty(8, fill=0xEE)
        D.call("agpu_compare_validity_count", capi.CMP_EQ, capi.I32, da.vp, db.vp, dva.vp, dvb.vp, ob.vp, ov.vp, n, cnt.vp)
        assert u64(D, cnt) == want, (key, val)
        out, c2 = D.empty(nb), D.empty(8, fill=0xEE)
        D.call("agpu_bitmap_binary_count", capi.OP_AND, dva.vp, dvb.vp, out.vp, n, c2.vp)
        assert u64(D, c2) == n - want, (key, val)
        capi.call("agpu_pipeline_set_tuning", D.p._h, key.encode(), 0)
    D.release()


def test_host_api_carries_the_count(ag):
    """a.eq(b) with nulls: the BooleanArrayGPU's validity knows its null count without a pass over the bitmap, to_arrow hands
    it to the consumer; the plain validity AND of an arithmetic op carries it too"""
    dev = ag.GPU_DEVICE()
    rng = np.random.default_rng(0)
    n = 100_003
    av = [None if rng.random() < 0.1 else int(v) for v in rng.integers(0, 50, n)]
    bv = [None if rng.random() < 0.1 else int(v) for v in rng.integers(0, 50, n)]
    a, b = ag.Int32ArrayGPU.from_optional_slice(av, dev), ag.Int32ArrayGPU.from_optional_slice(bv, dev)
    want = sum(1 for x, y in zip(av, bv) if x is None or y is None)
    r = a.eq(b)
    assert r.null_buffer.null_count_known() and r.null_buffer.null_count() == want
    s = a.add(b)
    assert s.null_buffer.null_count_known() and s.null_buffer.null_count() == want
    assert not a.null_buffer.null_count_known() and a.null_buffer.null_count() == sum(1 for x in av if x is None)
    pa = pytest.importorskip("pyarrow")
    arr = ag.interop.to_arrow(r)
    assert arr.null_count == want and isinstance(arr, pa.BooleanArray)


def test_counts_at_1e9_rows_cost_no_pass_over_the_bitmap(D):
    """full size: the null count of i32 eq + validity at 1e9 rows equals the stand-alone popcount of the stored bitmap, and
    inclusion–exclusion ties the by-product counts of AND and OR together"""
    n = 1_000_000_000
    dev, p = D.dev, D.p
    nb = (n + 63) // 64 * 8
    ia, ib, va, vb, ob, ov = dev.create_table_buffers([4 * n] * 2 + [nb] * 4)
    h = p._handle
    vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
    capi.call("agpu_synth_i32", h, vp(ia), n, 3, 0, 1024)
    capi.call("agpu_synth_i32", h, vp(ib), n, 4, 0, 1024)
    capi.call("agpu_synth_bits", h, vp(va), n, 5, 0, C.c_double(0.9))
    capi.call("agpu_synth_bits", h, vp(vb), n, 6, 0, C.c_double(0.9))
    cnt = [dev.create_empty_buffer(8) for _ in range(6)]
    capi.call("agpu_compare_validity_count", h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(va), vp(vb), vp(ob), vp(ov), n, vp(cnt[0]))
    capi.call("agpu_bitmap_popcount", h, vp(ov), n, vp(cnt[1]))
    capi.call("agpu_bitmap_binary_count", h, capi.OP_OR, vp(va), vp(vb), vp(ob), n, vp(cnt[2]))
    capi.call("agpu_bitmap_popcount", h, vp(va), n, vp(cnt[3]))
    capi.call("agpu_bitmap_popcount", h, vp(vb), n, vp(cnt[4]))
    capi.call("agpu_bitmap_binary_count", h, capi.OP_AND, vp(va), vp(vb), vp(ob), n, vp(cnt[5]))
    v = [int(dev.retrive_data(c, 8, pipeline=p).view(np.uint64)[0]) for c in cnt]
    nulls, set_and, set_or, set_a, set_b, set_and2 = v
    assert nulls == n - set_and and set_and == set_and2
    assert set_and + set_or == set_a + set_b            # inclusion–exclusion
    assert abs(nulls / n - 0.19) < 0.001                 # 10 % nulls per side, independent
    # a 2^20-row window of the inputs against the oracle pins the absolute value
    w = 1 << 20
    ea = O.bitmap_popcount(O.synth_bits(w, 5, 0, 0.9), w)
    got = np.empty(w // 8, np.uint8)
    capi.call("agpu_download", h, C.c_void_p(got.ctypes.data), vp(va), w // 8)
    assert O.bitmap_popcount(got, w) == ea
