"""GPU: BASELINE.json's full sizes (1e9 rows per column) — too big for the scalar oracle in a test's time budget, so
parity is asserted through size-independent PROPERTIES of the domain (exact, integer-checkable), plus oracle windows:

  * wrapping-sum linearity      sum(a) + sum(b) == sum(a + b)  (mod 2^32)            [i32 add + Sum]
  * min/max partition           sum(min(a,b)) + sum(max(a,b)) == sum(a) + sum(b)       [MinMax]
  * compare complements         popcount(a < b) + popcount(a >= b) == n;  eq(a, a) is all ones
  * bitmap inclusion–exclusion  |va & vb| + |va | vb| == |va| + |vb|                   [validity AND]
  * cast round trip             u8 → f32 → u8 is the identity (checksum equality)
  * commutativity               checksum(a + b) == checksum(b + a)                     [f32 add]
  * gather/scatter inverses     take(a, perm) scattered back by put(perm) == a; merge(a, a, m) == a
  * windows                     first / middle / last 65 536 rows of the outputs, bit-exact vs the oracle on the same
                                counter-based generator
"""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi

pytestmark = pytest.mark.gpu

N = 1_000_000_000
SEED = 20250418
WINDOW = 1 << 16


@pytest.fixture(scope="module")
def ctx():
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "fullsize")
    return dev, p


def vp(buf, off=0):
    return C.c_void_p(buf.ptr + off)


def scalar_u64(dev, p, buf):
    return int(dev.retrive_data(buf, 8, pipeline=p).view(np.uint64)[0])


def scalar_u32(dev, p, buf):
    return int(dev.retrive_data(buf, 4, pipeline=p).view(np.uint32)[0])


def windows(n):
    return (0, (n // 2) // 64 * 64, (n - WINDOW) // 64 * 64)


def download(dev, p, buf, byte_off, nbytes):
    out = np.empty(nbytes, np.uint8)
    capi.call("agpu_download", p._handle, C.c_void_p(out.ctypes.data), C.c_void_p(buf.ptr + byte_off), nbytes)
    return out


def test_i32_add_sum_linearity_minmax_partition_and_compare(ctx):
    dev, p = ctx
    h = p._handle
    a, b, out = (dev.create_empty_buffer(4 * N) for _ in range(3))
    res = dev.create_empty_buffer(16)
    capi.call("agpu_synth_i32", h, vp(a), N, SEED + 2, 0, 1024)
    capi.call("agpu_synth_i32", h, vp(b), N, SEED + 3, 0, 1024)

    def wsum(buf):
        capi.call("agpu_reduce", h, capi.RED_SUM, capi.I32, vp(buf), None, N, vp(res))
        return scalar_u32(dev, p, res)

    sa, sb = wsum(a), wsum(b)
    capi.call("agpu_binary", h, capi.OP_ADD, capi.I32, vp(a), vp(b), vp(out), N)
    assert wsum(out) == (sa + sb) & 0xFFFFFFFF
    for start in windows(N):
        got = download(dev, p, out, 4 * start, 4 * WINDOW).view(np.int32)
        exp = O.binary(O.OP_ADD, O.I32, O.synth_i32(WINDOW, SEED + 2, start, 1024), O.synth_i32(WINDOW, SEED + 3, start, 1024))
        assert np.array_equal(got, exp)
    capi.call("agpu_binary", h, capi.OP_MIN, capi.I32, vp(a), vp(b), vp(out), N)
    smin = wsum(out)
    capi.call("agpu_binary", h, capi.OP_MAX, capi.I32, vp(a), vp(b), vp(out), N)
    smax = wsum(out)
    assert (smin + smax) & 0xFFFFFFFF == (sa + sb) & 0xFFFFFFFF

    nb = (N + 63) // 64 * 8
    bits, cnt = dev.create_empty_buffer(nb), dev.create_empty_buffer(16)

    def popcount_of(op, x, y):
        capi.call("agpu_compare", h, op, capi.I32, vp(x), vp(y), vp(bits), N)
        capi.call("agpu_bitmap_popcount", h, vp(bits), N, vp(cnt))
        return scalar_u64(dev, p, cnt)

    lt, gteq, eq = popcount_of(capi.CMP_LT, a, b), popcount_of(capi.CMP_GTEQ, a, b), popcount_of(capi.CMP_EQ, a, b)
    assert lt + gteq == N
    assert popcount_of(capi.CMP_EQ, a, a) == N
    assert popcount_of(capi.CMP_GT, a, a) == 0
    # values are uniform on 0..1023 ⇒ P(eq) = 1/1024; a loose sanity band on the count
    assert abs(eq - N / 1024) < 5 * (N / 1024) ** 0.5 + 1000
    for variant in (0, 1):  # both compare kernels agree on the full column
        capi.call("agpu_pipeline_set_tuning", h, b"cmp_variant", variant)
        try:
            assert popcount_of(capi.CMP_LTEQ, a, b) == lt + eq
        finally:
            capi.call("agpu_pipeline_set_tuning", h, b"cmp_variant", 0)


def test_eq_with_validity_fullsize_windows_and_bitmap_identities(ctx):
    dev, p = ctx
    h = p._handle
    nb = (N + 63) // 64 * 8
    a, b = dev.create_empty_buffer(4 * N), dev.create_empty_buffer(4 * N)
    va, vb, ob, ov, tmp = (dev.create_empty_buffer(nb) for _ in range(5))
    cnt = dev.create_empty_buffer(16)
    capi.call("agpu_synth_i32", h, vp(a), N, SEED + 2, 0, 1024)
    capi.call("agpu_synth_i32", h, vp(b), N, SEED + 3, 0, 1024)
    capi.call("agpu_synth_bits", h, vp(va), N, SEED + 4, 0, C.c_double(0.9))
    capi.call("agpu_synth_bits", h, vp(vb), N, SEED + 5, 0, C.c_double(0.9))
    capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, vp(a), vp(b), vp(va), vp(vb), vp(ob), vp(ov), N)

    def pop(buf):
        capi.call("agpu_bitmap_popcount", h, vp(buf), N, vp(cnt))
        return scalar_u64(dev, p, cnt)

    n_va, n_vb, n_and = pop(va), pop(vb), pop(ov)
    capi.call("agpu_bitmap_binary", h, capi.OP_OR, vp(va), vp(vb), vp(tmp), N)
    assert n_and + pop(tmp) == n_va + n_vb                      # inclusion–exclusion
    capi.call("agpu_bitmap_binary", h, capi.OP_AND, vp(va), vp(vb), vp(tmp), N)
    cs1, cs2 = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
    capi.call("agpu_checksum", h, vp(tmp), nb, vp(cs1))
    capi.call("agpu_checksum", h, vp(ov), nb, vp(cs2))
    assert scalar_u64(dev, p, cs1) == scalar_u64(dev, p, cs2)   # fused validity == stand-alone bitmap AND
    assert abs(n_va / N - 0.9) < 1e-3 and abs(n_and / N - 0.81) < 1e-3  # 10 % nulls per side ⇒ ≈19 % null output
    for start in windows(N):
        wb = WINDOW // 8
        eb = O.compare(O.CMP_EQ, O.I32, O.synth_i32(WINDOW, SEED + 2, start, 1024), O.synth_i32(WINDOW, SEED + 3, start, 1024))
        ev = O.bitmap_binary(O.OP_AND, O.synth_bits(WINDOW, SEED + 4, start, 0.9), O.synth_bits(WINDOW, SEED + 5, start, 0.9), WINDOW)
        assert np.array_equal(download(dev, p, ob, start // 8, wb), eb[:wb])
        assert np.array_equal(download(dev, p, ov, start // 8, wb), ev[:wb])


def test_f32_arithmetic_fullsize(ctx):
    dev, p = ctx
    h = p._handle
    a, b, o1, o2 = (dev.create_empty_buffer(4 * N) for _ in range(4))
    cs1, cs2 = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
    capi.call("agpu_synth_f32", h, vp(a), N, SEED, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_synth_f32", h, vp(b), N, SEED + 1, 0, C.c_float(-1000.0), C.c_float(1000.0))
    for op in (capi.OP_ADD, capi.OP_MUL):  # commutative ops: a∘b and b∘a are bitwise equal
        capi.call("agpu_binary", h, op, capi.F32, vp(a), vp(b), vp(o1), N)
        capi.call("agpu_binary", h, op, capi.F32, vp(b), vp(a), vp(o2), N)
        capi.call("agpu_checksum", h, vp(o1), 4 * N, vp(cs1))
        capi.call("agpu_checksum", h, vp(o2), 4 * N, vp(cs2))
        assert scalar_u64(dev, p, cs1) == scalar_u64(dev, p, cs2)
    for op in (capi.OP_ADD, capi.OP_SUB, capi.OP_MUL, capi.OP_DIV):
        capi.call("agpu_binary", h, op, capi.F32, vp(a), vp(b), vp(o1), N)
        for start in windows(N):
            got = download(dev, p, o1, 4 * start, 4 * WINDOW).view(np.uint32)
            exp = O.binary(op, O.F32, O.synth_f32(WINDOW, SEED, start, -1000.0, 1000.0), O.synth_f32(WINDOW, SEED + 1, start, -1000.0, 1000.0))
            assert np.array_equal(got, exp.view(np.uint32)), (op, start)
    # a − a == +0.0 everywhere, a / a == 1.0 wherever a != 0
    capi.call("agpu_binary", h, capi.OP_SUB, capi.F32, vp(a), vp(a), vp(o1), N)
    mx = dev.create_empty_buffer(16)
    capi.call("agpu_unary", h, capi.UN_ABS, capi.F32, vp(o1), vp(o2), N)
    capi.call("agpu_reduce", h, capi.RED_MAX, capi.F32, vp(o2), None, N, vp(mx))
    assert dev.retrive_data(mx, 4, pipeline=p).view(np.float32)[0] == 0.0
    # f32 sum in the reference's order is deterministic: two runs, identical bits
    capi.call("agpu_reduce", h, capi.RED_SUM, capi.F32, vp(a), None, N, vp(cs1))
    s1 = scalar_u32(dev, p, cs1)
    capi.call("agpu_reduce", h, capi.RED_SUM, capi.F32, vp(a), None, N, vp(cs1))
    assert scalar_u32(dev, p, cs1) == s1


def test_cast_roundtrip_and_trig_fullsize(ctx):
    dev, p = ctx
    h = p._handle
    u = dev.create_empty_buffer(N)
    f, g = dev.create_empty_buffer(4 * N), dev.create_empty_buffer(4 * N)
    u2 = dev.create_empty_buffer(N)
    cs1, cs2 = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
    capi.call("agpu_synth_u8", h, vp(u), N, SEED + 6, 0)
    capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u), vp(f), N)
    capi.call("agpu_cast", h, capi.F32, capi.U8, vp(f), vp(u2), N)
    capi.call("agpu_checksum", h, vp(u), N, vp(cs1))
    capi.call("agpu_checksum", h, vp(u2), N, vp(cs2))
    assert scalar_u64(dev, p, cs1) == scalar_u64(dev, p, cs2)           # u8 → f32 → u8 is the identity
    # fused sin_u8 == sin_f32(cast(u8)) bit for bit (same device function behind the table)
    capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(u), vp(g), N)
    capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(f), N)  # in place
    capi.call("agpu_checksum", h, vp(g), 4 * N, vp(cs1))
    capi.call("agpu_checksum", h, vp(f), 4 * N, vp(cs2))
    assert scalar_u64(dev, p, cs1) == scalar_u64(dev, p, cs2)
    for start in windows(N):
        got = download(dev, p, g, 4 * start, 4 * WINDOW).view(np.float32)
        exp = O.unary(O.UN_SIN, O.U8, O.synth_u8(WINDOW, SEED + 6, start))
        ulp = np.abs(got.view(np.int32).astype(np.int64) - exp.view(np.int32).astype(np.int64)).max()
        assert ulp <= 1
    # |sin| ≤ 1 over the whole column
    mx = dev.create_empty_buffer(16)
    capi.call("agpu_reduce", h, capi.RED_MAX, capi.F32, vp(g), None, N, vp(mx))
    assert dev.retrive_data(mx, 4, pipeline=p).view(np.float32)[0] <= 1.0


def test_take_put_merge_inverses_fullsize(ctx):
    """2^28 rows (the index arrays make 1e9 rows a 16 GB test for no extra coverage)."""
    dev, p = ctx
    h = p._handle
    n = 1 << 28
    a, t, back = (dev.create_empty_buffer(4 * n) for _ in range(3))
    idx = dev.create_empty_buffer(4 * n)
    iota = dev.create_empty_buffer(4 * n)
    cs1, cs2 = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
    capi.call("agpu_synth_i32", h, vp(a), n, 1, 0, 0)
    # a permutation of 0..n-1: i → (i * odd + c) mod 2^28, built with the integer kernels themselves
    host_iota = np.arange(1 << 20, dtype=np.uint32)
    small = dev.create_gpu_buffer_with_data(host_iota)
    one = dev.create_gpu_buffer_with_data(np.array([1 << 20], np.uint32))
    capi.call("agpu_copy", h, vp(iota), vp(small), 4 << 20)
    filled = 1 << 20
    while filled < n:  # doubling: iota[filled:2*filled] = iota[:filled] + filled
        cur = dev.create_gpu_buffer_with_data(np.array([filled], np.uint32))
        capi.call("agpu_scalar", h, capi.OP_ADD, capi.U32, vp(iota), vp(cur), vp(iota, 4 * filled), filled)
        p.sync()
        filled *= 2
    mul = dev.create_gpu_buffer_with_data(np.array([2654435761], np.uint32))   # odd ⇒ bijection mod 2^32
    msk = dev.create_gpu_buffer_with_data(np.array([n - 1], np.uint32))
    capi.call("agpu_scalar", h, capi.OP_MUL, capi.U32, vp(iota), vp(mul), vp(idx), n)
    capi.call("agpu_scalar", h, capi.OP_AND, capi.U32, vp(idx), vp(msk), vp(idx), n)  # low 28 bits: still a bijection
    mx = dev.create_empty_buffer(16)
    capi.call("agpu_index_max", h, vp(idx), n, vp(mx))
    assert scalar_u32(dev, p, mx) == n - 1
    capi.call("agpu_take", h, 4, vp(a), n, vp(idx), vp(t), n)                  # t[i] = a[perm[i]]
    capi.call("agpu_memset", h, vp(back), 0, 4 * n)
    capi.call("agpu_put", h, 4, vp(t), vp(iota), vp(back), vp(idx), n)         # back[perm[i]] = t[i]
    capi.call("agpu_checksum", h, vp(a), 4 * n, vp(cs1))
    capi.call("agpu_checksum", h, vp(back), 4 * n, vp(cs2))
    assert scalar_u64(dev, p, cs1) == scalar_u64(dev, p, cs2)
    m = dev.create_empty_buffer((n + 63) // 64 * 8)
    capi.call("agpu_synth_bits", h, vp(m), n, 3, 0, C.c_double(0.5))
    capi.call("agpu_merge", h, 4, vp(a), vp(a), vp(m), vp(t), n)               # merge(a, a, mask) == a
    capi.call("agpu_checksum", h, vp(t), 4 * n, vp(cs2))
    assert scalar_u64(dev, p, cs1) == scalar_u64(dev, p, cs2)
    del one


def test_columns_longer_than_2_pow_32_rows(ctx):
    """u8 columns of 2^32 + 70 001 rows (4.3 GB each): every index is 64-bit — windows past row 2^32 are bit-exact
    against the oracle for add, eq -> bitmap, cast u8 -> f32 (17 GB output) and the popcount of the compare result."""
    dev, p = ctx
    h = p._handle
    n = (1 << 32) + 70_001
    a, b, o = (dev.create_empty_buffer(n) for _ in range(3))
    bits = dev.create_empty_buffer((n + 63) // 64 * 8)
    cnt = dev.create_empty_buffer(16)
    capi.call("agpu_synth_u8", h, vp(a), n, SEED, 0)
    capi.call("agpu_synth_u8", h, vp(b), n, SEED + 1, 0)
    capi.call("agpu_binary", h, capi.OP_ADD, capi.U8, vp(a), vp(b), vp(o), n)
    capi.call("agpu_compare", h, capi.CMP_EQ, capi.U8, vp(a), vp(b), vp(bits), n)
    capi.call("agpu_bitmap_popcount", h, vp(bits), n, vp(cnt))
    n_eq = scalar_u64(dev, p, cnt)
    assert abs(n_eq / n - 1 / 256) < 1e-4  # independent uniform bytes agree 1 time in 256
    wide = dev.create_empty_buffer(4 * n)
    capi.call("agpu_cast", h, capi.U8, capi.F32, vp(a), vp(wide), n)
    for start in ((1 << 32) - 4096, (1 << 32) + 1024, (n - WINDOW) // 64 * 64):
        cntw = min(WINDOW, n - start)
        ea, eb = O.synth_u8(cntw, SEED, start), O.synth_u8(cntw, SEED + 1, start)
        got = download(dev, p, o, start, cntw)
        assert np.array_equal(got, O.binary(O.OP_ADD, O.U8, ea, eb)), start
        gb = download(dev, p, bits, start // 8, cntw // 8)
        assert np.array_equal(gb, O.compare(O.CMP_EQ, O.U8, ea, eb)[: cntw // 8]), start
        gw = download(dev, p, wide, 4 * start, 4 * cntw).view(np.float32)
        assert np.array_equal(gw, ea.astype(np.float32)), start
    # the very last rows (ragged tail of every kernel)
    tail = 70_001 % 4096 + 4096
    ea, eb = O.synth_u8(tail, SEED, n - tail), O.synth_u8(tail, SEED + 1, n - tail)
    assert np.array_equal(download(dev, p, o, n - tail, tail), O.binary(O.OP_ADD, O.U8, ea, eb))
    assert np.array_equal(download(dev, p, wide, 4 * (n - tail), 4 * tail).view(np.float32), ea.astype(np.float32))


def test_lt_gt_with_validity_fullsize_oracle_windows(ctx):
    """BASELINE config 3 names eq / lt / gt: lt and gt → bitmap with the fused validity AND at 1e9 rows, 10 % nulls per
    side, windows bit-exact against the oracle (eq is covered above)."""
    dev, p = ctx
    h = p._handle
    nb = (N + 63) // 64 * 8
    a, b = dev.create_empty_buffer(4 * N), dev.create_empty_buffer(4 * N)
    va, vb, ob, ov = (dev.create_empty_buffer(nb) for _ in range(4))
    cnt = dev.create_empty_buffer(16)
    capi.call("agpu_synth_i32", h, vp(a), N, SEED + 2, 0, 1024)
    capi.call("agpu_synth_i32", h, vp(b), N, SEED + 3, 0, 1024)
    capi.call("agpu_synth_bits", h, vp(va), N, SEED + 4, 0, C.c_double(0.9))
    capi.call("agpu_synth_bits", h, vp(vb), N, SEED + 5, 0, C.c_double(0.9))
    counts = {}
    for op, oop in ((capi.CMP_LT, O.CMP_LT), (capi.CMP_GT, O.CMP_GT)):
        capi.call("agpu_compare_validity", h, op, capi.I32, vp(a), vp(b), vp(va), vp(vb), vp(ob), vp(ov), N)
        capi.call("agpu_bitmap_popcount", h, vp(ob), N, vp(cnt))
        counts[op] = scalar_u64(dev, p, cnt)
        for start in windows(N):
            wb = WINDOW // 8
            ea, eb_ = O.synth_i32(WINDOW, SEED + 2, start, 1024), O.synth_i32(WINDOW, SEED + 3, start, 1024)
            ev = O.bitmap_binary(O.OP_AND, O.synth_bits(WINDOW, SEED + 4, start, 0.9), O.synth_bits(WINDOW, SEED + 5, start, 0.9), WINDOW)
            assert np.array_equal(download(dev, p, ob, start // 8, wb), O.compare(oop, O.I32, ea, eb_)[:wb]), (op, start)
            assert np.array_equal(download(dev, p, ov, start // 8, wb), ev[:wb]), (op, start)
    capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(a), vp(b), vp(ob), N)
    capi.call("agpu_bitmap_popcount", h, vp(ob), N, vp(cnt))
    assert counts[capi.CMP_LT] + counts[capi.CMP_GT] + scalar_u64(dev, p, cnt) == N  # trichotomy over the whole column


def test_sub_mul_div_with_both_validities_fullsize(ctx):
    """f32 sub / mul / div through the HOST layer at 1e9 rows with a validity bitmap on both sides (10 % nulls each):
    value windows bit-exact, output validity = AND of the inputs (windows + whole-column null count)."""
    import arrow_gpu_amd as ag

    dev, p = ctx
    h = p._handle
    nb = (N + 63) // 64 * 8
    da, db = dev.create_empty_buffer(4 * N), dev.create_empty_buffer(4 * N)
    va, vb = dev.create_empty_buffer(nb), dev.create_empty_buffer(nb)
    capi.call("agpu_synth_f32", h, vp(da), N, SEED, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_synth_f32", h, vp(db), N, SEED + 1, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_synth_bits", h, vp(va), N, SEED + 4, 0, C.c_double(0.9))
    capi.call("agpu_synth_bits", h, vp(vb), N, SEED + 5, 0, C.c_double(0.9))
    p.finish()
    a = ag.Float32ArrayGPU(da, dev, N, ag.NullBitBufferGpu(va, N, dev))
    b = ag.Float32ArrayGPU(db, dev, N, ag.NullBitBufferGpu(vb, N, dev))
    cnt = dev.create_empty_buffer(16)
    for name, oop in (("sub", O.OP_SUB), ("mul", O.OP_MUL), ("div", O.OP_DIV)):
        c = getattr(a, name)(b)
        assert c.len == N and c.null_buffer is not None
        for start in windows(N):
            ea = O.synth_f32(WINDOW, SEED, start, -1000.0, 1000.0)
            eb_ = O.synth_f32(WINDOW, SEED + 1, start, -1000.0, 1000.0)
            got = download(dev, p, c.data, 4 * start, 4 * WINDOW).view(np.uint32)
            assert np.array_equal(got, O.binary(oop, O.F32, ea, eb_).view(np.uint32)), (name, start)
            ev = O.bitmap_binary(O.OP_AND, O.synth_bits(WINDOW, SEED + 4, start, 0.9), O.synth_bits(WINDOW, SEED + 5, start, 0.9), WINDOW)
            assert np.array_equal(download(dev, p, c.null_buffer.bit_buffer, start // 8, WINDOW // 8), ev[: WINDOW // 8]), (name, start)
        capi.call("agpu_bitmap_popcount", h, vp(c.null_buffer.bit_buffer), N, vp(cnt))
        assert abs(scalar_u64(dev, p, cnt) / N - 0.81) < 1e-3
        del c


def test_cos_fullsize_u8_f32_and_cast_then_cos(ctx):
    """BASELINE config 4 (cast u8→f32 then sin/cos): the cos half at 1e9 rows — fused cos_u8, cos f32 of the cast result
    (bit-identical to the fused kernel) and f32 cos on a wide-range f32 column, oracle windows ≤ 1 ULP."""
    dev, p = ctx
    h = p._handle
    u = dev.create_empty_buffer(N)
    f, g = dev.create_empty_buffer(4 * N), dev.create_empty_buffer(4 * N)
    cs1, cs2, mx = dev.create_empty_buffer(16), dev.create_empty_buffer(16), dev.create_empty_buffer(16)
    capi.call("agpu_synth_u8", h, vp(u), N, SEED + 6, 0)
    capi.call("agpu_unary", h, capi.UN_COS, capi.U8, vp(u), vp(g), N)          # fused cos_u8
    capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u), vp(f), N)              # cast, then cos
    capi.call("agpu_unary", h, capi.UN_COS, capi.F32, vp(f), vp(f), N)
    capi.call("agpu_checksum", h, vp(g), 4 * N, vp(cs1))
    capi.call("agpu_checksum", h, vp(f), 4 * N, vp(cs2))
    assert scalar_u64(dev, p, cs1) == scalar_u64(dev, p, cs2)
    for start in windows(N):
        got = download(dev, p, g, 4 * start, 4 * WINDOW).view(np.float32)
        exp = O.unary(O.UN_COS, O.U8, O.synth_u8(WINDOW, SEED + 6, start))
        assert np.abs(got.view(np.int32).astype(np.int64) - exp.view(np.int32).astype(np.int64)).max() <= 1
    capi.call("agpu_unary", h, capi.UN_ABS, capi.F32, vp(g), vp(f), N)
    capi.call("agpu_reduce", h, capi.RED_MAX, capi.F32, vp(f), None, N, vp(mx))
    assert dev.retrive_data(mx, 4, pipeline=p).view(np.float32)[0] <= 1.0
    # f32 cos over [-1000, 1000): argument reduction is exercised, not just the table
    capi.call("agpu_synth_f32", h, vp(f), N, SEED, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_unary", h, capi.UN_COS, capi.F32, vp(f), vp(g), N)
    for start in windows(N):
        got = download(dev, p, g, 4 * start, 4 * WINDOW).view(np.float32)
        exp = O.unary(O.UN_COS, O.F32, O.synth_f32(WINDOW, SEED, start, -1000.0, 1000.0))
        assert np.abs(got.view(np.int32).astype(np.int64) - exp.view(np.int32).astype(np.int64)).max() <= 1


def test_pow_log_fullsize_windows_and_slice_consistency(ctx):
    """The two LDS-table kernels at 1e9 rows: f32 pow (array ∘ array) and log against oracle windows (≤ 1 ULP, NaN for NaN),
    log over a column that is half negative (mixed waves: table form + general form in one wave), and the whole column's
    result bit-identical to the same column processed as two slices split at a row that is no tile boundary."""
    dev, p = ctx
    h = p._handle
    x, y, o, o2 = (dev.create_empty_buffer(4 * N) for _ in range(4))
    cs1, cs2 = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
    capi.call("agpu_synth_f32", h, vp(x), N, SEED + 11, 0, C.c_float(0.001), C.c_float(1000.0))
    capi.call("agpu_synth_f32", h, vp(y), N, SEED + 12, 0, C.c_float(-8.0), C.c_float(8.0))

    def ulps(got, exp):
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        return int(np.abs(got.view(np.int32).astype(np.int64)[ok] - exp.view(np.int32).astype(np.int64)[ok]).max())

    capi.call("agpu_binary", h, capi.OP_POW, capi.F32, vp(x), vp(y), vp(o), N)
    for start in windows(N):
        got = download(dev, p, o, 4 * start, 4 * WINDOW).view(np.float32)
        exp = O.binary(O.OP_POW, O.F32, O.synth_f32(WINDOW, SEED + 11, start, 0.001, 1000.0), O.synth_f32(WINDOW, SEED + 12, start, -8.0, 8.0))
        assert ulps(got, exp) <= 1
    split = (N // 3) // 4 * 4 + 4 * 37          # 16-byte aligned, not a multiple of the 1024-row tile
    capi.call("agpu_binary", h, capi.OP_POW, capi.F32, vp(x), vp(y), vp(o2), split)
    capi.call("agpu_binary", h, capi.OP_POW, capi.F32, vp(x, 4 * split), vp(y, 4 * split), vp(o2, 4 * split), N - split)
    capi.call("agpu_checksum", h, vp(o), 4 * N, vp(cs1))
    capi.call("agpu_checksum", h, vp(o2), 4 * N, vp(cs2))
    assert scalar_u64(dev, p, cs1) == scalar_u64(dev, p, cs2)

    capi.call("agpu_unary", h, capi.UN_LOG, capi.F32, vp(x), vp(o), N)
    for start in windows(N):
        got = download(dev, p, o, 4 * start, 4 * WINDOW).view(np.float32)
        assert ulps(got, O.unary(O.UN_LOG, O.F32, O.synth_f32(WINDOW, SEED + 11, start, 0.001, 1000.0))) <= 1
    capi.call("agpu_unary", h, capi.UN_LOG, capi.F32, vp(y), vp(o), N)   # half the rows negative → NaN
    for start in windows(N):
        got = download(dev, p, o, 4 * start, 4 * WINDOW).view(np.float32)
        exp = O.unary(O.UN_LOG, O.F32, O.synth_f32(WINDOW, SEED + 12, start, -8.0, 8.0))
        assert ulps(got, exp) <= 1 and np.isnan(exp).any() and not np.isnan(exp).all()
    capi.call("agpu_unary", h, capi.UN_LOG, capi.F32, vp(y), vp(o2), split)
    capi.call("agpu_unary", h, capi.UN_LOG, capi.F32, vp(y, 4 * split), vp(o2, 4 * split), N - split)
    capi.call("agpu_checksum", h, vp(o), 4 * N, vp(cs1))
    capi.call("agpu_checksum", h, vp(o2), 4 * N, vp(cs2))
    assert scalar_u64(dev, p, cs1) == scalar_u64(dev, p, cs2)


def test_f32_to_small_int_casts_fullsize(ctx):
    """The reference-absent narrowing casts (f32 → i8 / i16 / u16) at 1e9 rows: windows bit-exact vs the oracle and the
    round trip i16 → f32 → i16 is the identity."""
    dev, p = ctx
    h = p._handle
    f = dev.create_empty_buffer(4 * N)
    o = dev.create_empty_buffer(2 * N)
    capi.call("agpu_synth_f32", h, vp(f), N, SEED, 0, C.c_float(-70000.0), C.c_float(70000.0))
    for to, oto, npd in ((capi.I8, O.I8, np.int8), (capi.I16, O.I16, np.int16), (capi.U16, O.U16, np.uint16)):
        capi.call("agpu_cast", h, capi.F32, to, vp(f), vp(o), N)
        w = np.dtype(npd).itemsize
        for start in windows(N):
            got = download(dev, p, o, w * start, w * WINDOW).view(npd)
            exp = O.cast(O.F32, oto, O.synth_f32(WINDOW, SEED, start, -70000.0, 70000.0))
            assert np.array_equal(got, exp), (to, start)
    cs1, cs2 = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
    back = dev.create_empty_buffer(2 * N)
    capi.call("agpu_cast", h, capi.I16, capi.F32, vp(o), vp(f), N)   # o holds u16 bits; reinterpret as i16
    capi.call("agpu_cast", h, capi.F32, capi.I16, vp(f), vp(back), N)
    capi.call("agpu_checksum", h, vp(o), 2 * N, vp(cs1))
    capi.call("agpu_checksum", h, vp(back), 2 * N, vp(cs2))
    assert scalar_u64(dev, p, cs1) == scalar_u64(dev, p, cs2)


def test_f32_tree_sum_fullsize_is_the_reference_tree_of_its_aligned_chunks(ctx):
    """f32 Sum at 1e9 rows in the reference's order [aggregate.wgsl:21-41]: the tree is perfect over aligned 256^k-row
    blocks, so the whole-column result must equal the reference's 256-ary tree over the sums of the aligned
    16 777 216-row chunks (the 4th level: 60 values, zero-padded) — each chunk summed by its own launch, three of them
    checked bit for bit against the oracle's literal tree; the same with a validity bitmap."""
    dev, p = ctx
    h = p._handle
    a = dev.create_empty_buffer(4 * N)
    v = dev.create_empty_buffer((N + 63) // 64 * 8)
    out = dev.create_empty_buffer(16)
    capi.call("agpu_synth_f32", h, vp(a), N, SEED + 7, 0, C.c_float(-1000.0), C.c_float(1000.0))
    capi.call("agpu_synth_bits", h, vp(v), N, SEED + 8, 0, C.c_double(0.9))
    chunk = 1 << 24
    nchunks = (N + chunk - 1) // chunk
    for validity in (None, v):
        capi.call("agpu_reduce", h, capi.RED_SUM, capi.F32, vp(a), vp(validity) if validity is not None else None, N, vp(out))
        whole = scalar_u32(dev, p, out)
        parts = np.zeros(nchunks, np.float32)
        for k in range(nchunks):
            rows = min(chunk, N - k * chunk)
            capi.call("agpu_reduce", h, capi.RED_SUM, capi.F32, vp(a, 4 * k * chunk), vp(validity, k * chunk // 8) if validity is not None else None,
                      rows, vp(out))
            parts[k] = dev.retrive_data(out, 4, pipeline=p).view(np.float32)[0]
        assert np.float32(O.reduce(O.RED_SUM, O.F32, parts)).view(np.uint32) == whole
        for k in (0, nchunks // 2, nchunks - 1):  # includes the ragged last chunk
            rows = min(chunk, N - k * chunk)
            vals = O.synth_f32(rows, SEED + 7, k * chunk, -1000.0, 1000.0)
            if validity is not None:
                bits = O.synth_bits(rows, SEED + 8, k * chunk, 0.9)
                keep = np.unpackbits(bits, bitorder="little")[:rows].astype(bool)
                vals = np.where(keep, vals, np.float32(0.0))
            assert np.float32(O.reduce(O.RED_SUM, O.F32, vals)).view(np.uint32) == parts[k].view(np.uint32), k


def test_cast_headed_chains_fullsize_equal_the_unfused_sequence(ctx):
    """BASELINE config 4 as worded — "cast u8→f32 then sin/cos" — at 1e9 rows in ONE launch (agpu_fused_cast_chain): the whole result column
    checksum-equal to the two-launch pair, for the 8-bit table route (sin, cos, a scale + offset, a heavy chain) and the 16-bit per-row route;
    windows against the oracle's sin_u8 (≤ 1 ULP)."""
    dev, p = ctx
    h = p._handle
    u = dev.create_empty_buffer(2 * N)       # N bytes as u8, 2N bytes as i16
    f, g = dev.create_empty_buffer(4 * N), dev.create_empty_buffer(4 * N)
    sc = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
    cs1, cs2 = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
    capi.call("agpu_synth_u8", h, vp(u), 2 * N, SEED + 6, 0)

    class Step(C.Structure):
        _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]

    def chain(*items):
        arr = (Step * len(items))()
        for k, (op, kind, operand) in enumerate(items):
            arr[k].op, arr[k].kind, arr[k].operand = op, kind, (operand.ptr if operand is not None else None)
        return arr, len(items)

    def unfused(src_dt, steps):
        capi.call("agpu_cast", h, src_dt, capi.F32, vp(u), vp(f), N)
        for op, kind, operand in steps:
            if kind == 0:
                capi.call("agpu_unary", h, op, capi.F32, vp(f), vp(f), N)
            else:
                capi.call("agpu_scalar", h, op, capi.F32, vp(f), vp(operand), vp(f), N)

    cases = [(capi.U8, [(capi.UN_SIN, 0, None)]), (capi.U8, [(capi.UN_COS, 0, None)]), (capi.U8, [(capi.OP_MUL, 1, sc), (capi.OP_ADD, 1, sc)]),
             (capi.U8, [(capi.OP_MUL, 1, sc), (capi.UN_SIN, 0, None), (capi.UN_ABS, 0, None), (capi.UN_SQRT, 0, None)]),
             (capi.I16, [(capi.UN_COS, 0, None)]), (capi.I16, [(capi.OP_MUL, 1, sc), (capi.UN_NEG, 0, None)])]
    for src_dt, steps in cases:
        st, ns = chain(*steps)
        capi.call("agpu_memset", h, vp(g), 0, 1 << 20)
        capi.call("agpu_fused_cast_chain", h, src_dt, vp(u), C.cast(st, C.c_void_p), ns, vp(g), N)
        unfused(src_dt, steps)
        capi.call("agpu_checksum", h, vp(g), 4 * N, vp(cs1))
        capi.call("agpu_checksum", h, vp(f), 4 * N, vp(cs2))
        assert scalar_u64(dev, p, cs1) == scalar_u64(dev, p, cs2), (src_dt, [s_[0] for s_ in steps])
    st, ns = chain((capi.UN_SIN, 0, None))
    capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u), C.cast(st, C.c_void_p), ns, vp(g), N)
    for start in windows(N):
        got = download(dev, p, g, 4 * start, 4 * WINDOW).view(np.float32)
        exp = O.unary(O.UN_SIN, O.U8, O.synth_u8(WINDOW, SEED + 6, start))
        assert np.abs(got.view(np.int32).astype(np.int64) - exp.view(np.int32).astype(np.int64)).max() <= 1, start
