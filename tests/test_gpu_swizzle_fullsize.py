"""GPU: take / put at BASELINE scale with 4- and 2-byte values — the code paths a 1e9-row column takes:
1 MiB regions (3 815 of them at 1e9 source elements) and, past 4095 · 2^18 ≈ 1.07e9 source elements, the 8-slot / 19-bit-offset
gather.  [ref: crates/routines/src/take.rs:9-55, put.rs:9-56, bool.rs:15-128]

No gigabyte host arrays are generated: the columns are built on the device —
    values[i] = i · 2654435761 mod 2^32          (iota by doubling, then a wrapping u32 multiply)
    16-bit values = the same buffer read as u16  (element j = half j & 1 of word j >> 1)
    idx = agpu_synth_i32(seed, modulus = n_values)   (the counter-based generator the oracle reproduces)
and checked in three layers:
  1. the auto policy's output is downloaded in 2^26-row chunks together with the index column and compared WHOLE with numpy
     (out == idx · odd; 16-bit: the matching half) — every row of the 1e9-row column;
  2. every forced form (gather_bucket 1 = direct kernel, 2 = bucketed / merge-back pipelines, 3 = the pair pipeline) must equal that
     verified column element for element on the device (agpu_compare EQ → popcount == n) — and have the same checksum;
  3. validity / Boolean results: windows against the ORACLE (oracle.take_bits over the generator's bitmap) and whole-bitmap equality
     between the forms; puts are pinned through their inverse (take(dst_after, dst_idx) == take(src, src_idx), a bijective dst_idx)
     plus the wrapping sum of the destination (untouched slots keep their pattern).
"""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi

pytestmark = pytest.mark.gpu

ODD = 2654435761
CHUNK = 1 << 26
SEED = 20250418

# (rows, source elements, destination elements of the puts, rows of the puts)
CASES = {
    "1e9_from_1e9": dict(n=1_000_000_000, n_values=1_000_000_000, n_dst=1 << 30, dst_off=0),
    # > 4095 · 2^18 source elements: 2^19-element regions / the 8-slot gather; a destination past the same bound, hit in its upper 2^30
    "2p30_from_1.25e9": dict(n=(1 << 30) + 70_001, n_values=1_250_000_011, n_dst=1_250_000_011, dst_off=1_250_000_011 - (1 << 30)),
}


def vp(buf, off=0):
    return C.c_void_p(buf.ptr + off)


class Cols:
    """The device columns of one case (built once per module parameter)."""

    def __init__(self, dev, p, n, n_values, n_dst, dst_off):
        self.dev, self.p, self.n, self.n_values, self.n_dst, self.dst_off = dev, p, n, n_values, n_dst, dst_off
        h = p._handle
        m = max(n, n_values, 1 << 30)
        self.m = m
        self.iota = dev.create_empty_buffer(4 * m)
        seed_rows = 1 << 20
        small = dev.create_gpu_buffer_with_data(np.arange(seed_rows, dtype=np.uint32))
        capi.call("agpu_copy", h, vp(self.iota), vp(small), 4 * seed_rows)
        filled = seed_rows
        while filled < m:  # doubling: iota[filled : filled + k] = iota[:k] + filled
            k = min(filled, m - filled)
            cur = dev.create_gpu_buffer_with_data(np.array([filled], np.uint32))
            capi.call("agpu_scalar", h, capi.OP_ADD, capi.U32, vp(self.iota), vp(cur), vp(self.iota, 4 * filled), k)
            p.sync()
            filled += k
        self.values = dev.create_empty_buffer(4 * n_values)
        mul = dev.create_gpu_buffer_with_data(np.array([ODD], np.uint32))
        capi.call("agpu_scalar", h, capi.OP_MUL, capi.U32, vp(self.iota), vp(mul), vp(self.values), n_values)
        self.idx = dev.create_empty_buffer(4 * n)
        capi.call("agpu_synth_i32", h, vp(self.idx), n, SEED + 8, 0, n_values)
        # a bijection of [0, 2^30) → rows i < n_put land on distinct destinations: perm[i] = ((i · odd) & (2^30 − 1)) + dst_off
        self.n_put = min(n, 1 << 30)
        self.perm = dev.create_empty_buffer(4 * self.n_put)
        msk = dev.create_gpu_buffer_with_data(np.array([(1 << 30) - 1], np.uint32))
        capi.call("agpu_scalar", h, capi.OP_MUL, capi.U32, vp(self.iota), vp(mul), vp(self.perm), self.n_put)
        capi.call("agpu_scalar", h, capi.OP_AND, capi.U32, vp(self.perm), vp(msk), vp(self.perm), self.n_put)
        if dst_off:
            off = dev.create_gpu_buffer_with_data(np.array([dst_off], np.uint32))
            capi.call("agpu_scalar", h, capi.OP_ADD, capi.U32, vp(self.perm), vp(off), vp(self.perm), self.n_put)
        self.vbits = dev.create_empty_buffer(O.bitmap_bytes(n_values) + 64)
        capi.call("agpu_synth_bits", h, vp(self.vbits), n_values, SEED + 10, 0, C.c_double(0.5))
        self.nb = O.bitmap_bytes(n)
        self.cmp_bits = dev.create_empty_buffer(O.bitmap_bytes(m) + 64)
        self.cnt = dev.create_empty_buffer(16)
        self.cs = dev.create_empty_buffer(16)
        p.sync()
        self.verified = {}  # width → device buffer holding the take result that was compared whole with numpy

    # ---- device-side helpers
    def count_equal(self, dt, a, b, n):
        h = self.p._handle
        capi.call("agpu_compare", h, capi.CMP_EQ, dt, vp(a), vp(b), vp(self.cmp_bits), n)
        capi.call("agpu_bitmap_popcount", h, vp(self.cmp_bits), n, vp(self.cnt))
        return int(self.dev.retrive_data(self.cnt, 8, pipeline=self.p).view(np.uint64)[0])

    def checksum(self, buf, nbytes):
        capi.call("agpu_checksum", self.p._handle, vp(buf), nbytes, vp(self.cs))
        return int(self.dev.retrive_data(self.cs, 8, pipeline=self.p).view(np.uint64)[0])

    def popcount(self, bits, n):
        capi.call("agpu_bitmap_popcount", self.p._handle, vp(bits), n, vp(self.cnt))
        return int(self.dev.retrive_data(self.cnt, 8, pipeline=self.p).view(np.uint64)[0])

    def wsum_u32(self, buf, n):
        capi.call("agpu_reduce", self.p._handle, capi.RED_SUM, capi.U32, vp(buf), None, n, vp(self.cnt))
        return int(self.dev.retrive_data(self.cnt, 4, pipeline=self.p).view(np.uint32)[0])

    def download(self, buf, byte_off, nbytes):
        out = np.empty(nbytes, np.uint8)
        capi.call("agpu_download", self.p._handle, C.c_void_p(out.ctypes.data), C.c_void_p(buf.ptr + byte_off), nbytes)
        return out

    def expected(self, width, idx):
        """values[idx] computed from the definition of the column"""
        if width == 4:
            return idx * np.uint32(ODD)
        word = (idx >> np.uint32(1)) * np.uint32(ODD)
        return ((word >> ((idx & np.uint32(1)) * np.uint32(16))) & np.uint32(0xFFFF)).astype(np.uint16)

    def compare_whole_column_with_numpy(self, width, out, idx_buf, n):
        npw = {4: np.uint32, 2: np.uint16}[width]
        bad = 0
        for r0 in range(0, n, CHUNK):
            k = min(CHUNK, n - r0)
            idx = self.download(idx_buf, 4 * r0, 4 * k).view(np.uint32)
            got = self.download(out, width * r0, width * k).view(npw)
            bad += int(np.count_nonzero(got != self.expected(width, idx)))
        return bad


@pytest.fixture(scope="module", params=list(CASES))
def cols(request):
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "swizzle-fullsize")
    c = Cols(dev, p, **CASES[request.param])
    yield c
    p.set_tuning("gather_bucket", 0)
    c.verified.clear()
    del c
    capi.call("agpu_device_trim", dev._handle)


def test_the_generated_columns_are_what_the_checks_assume(cols):
    """iota, values = iota · odd, the index column (oracle's generator, in range) and the bijective destination column"""
    c = cols
    for r0 in (0, (c.n_values // 2) // 64 * 64, c.n_values - 65536):
        got = c.download(c.values, 4 * r0, 4 * 65536).view(np.uint32)
        assert np.array_equal(got, (np.arange(r0, r0 + 65536, dtype=np.uint64) * ODD & 0xFFFFFFFF).astype(np.uint32)), r0
    for r0 in (0, (c.n // 2) // 64 * 64, c.n - 65536):
        got = c.download(c.idx, 4 * r0, 4 * 65536).view(np.uint32)
        assert np.array_equal(got, O.synth_i32(65536, SEED + 8, r0, c.n_values).view(np.uint32)), r0
    mx = c.dev.create_empty_buffer(16)
    capi.call("agpu_index_max", c.p._handle, vp(c.idx), c.n, vp(mx))
    assert int(c.dev.retrive_data(mx, 4, pipeline=c.p).view(np.uint32)[0]) < c.n_values
    # perm is injective: its wrapping sum over the 2^30 rows of the full bijection is the sum of 0..2^30-1 (+ offset); here only
    # the range is checked on the device, injectivity follows from odd · i mod 2^30
    capi.call("agpu_index_max", c.p._handle, vp(c.perm), c.n_put, vp(mx))
    assert int(c.dev.retrive_data(mx, 4, pipeline=c.p).view(np.uint32)[0]) < c.n_dst
    w = c.download(c.perm, 0, 4 * 65536).view(np.uint32)
    assert np.array_equal(w, ((np.arange(65536, dtype=np.uint64) * ODD) & ((1 << 30) - 1)).astype(np.uint32) + np.uint32(c.dst_off))


@pytest.mark.parametrize("width", [4, 2])
def test_take_whole_column_against_numpy_then_every_form_against_it(cols, width):
    c = cols
    h = c.p._handle
    dt = {4: capi.U32, 2: capi.U16}[width]
    n_values = c.n_values  # 16-bit: the first n_values halves of the same buffer
    out = c.dev.create_empty_buffer(width * c.n + 16)
    c.p.set_tuning("gather_bucket", 0)
    capi.call("agpu_memset", h, vp(out), 0xEE, width * c.n + 16)
    capi.call("agpu_take", h, width, vp(c.values), n_values, vp(c.idx), vp(out), c.n)
    c.p.sync()  # also: no index-range flag
    assert c.compare_whole_column_with_numpy(width, out, c.idx, c.n) == 0
    assert (c.download(out, width * c.n, 16) == 0xEE).all()  # nothing written behind the column
    c.verified[width] = out
    cs_ref = c.checksum(out, width * c.n)
    other = c.dev.create_empty_buffer(width * c.n + 16)
    try:
        for mode in (1, 2, 3):
            c.p.set_tuning("gather_bucket", mode)
            capi.call("agpu_memset", h, vp(other), 0x55, width * c.n + 16)
            capi.call("agpu_take", h, width, vp(c.values), n_values, vp(c.idx), vp(other), c.n)
            c.p.sync()
            assert c.count_equal(dt, other, out, c.n) == c.n, (width, mode)
            assert c.checksum(other, width * c.n) == cs_ref, (width, mode)
            assert (c.download(other, width * c.n, 16) == 0x55).all(), (width, mode)
    finally:
        c.p.set_tuning("gather_bucket", 0)


@pytest.mark.parametrize("width", [4, 2])
def test_take_with_validity_whole_column(cols, width):
    """values as above; the validity bits against the oracle in windows and identical across the forms over the whole bitmap"""
    c = cols
    h = c.p._handle
    dt = {4: capi.U32, 2: capi.U16}[width]
    if width not in c.verified:
        pytest.skip("the plain take of this width did not verify")
    ref_vals = c.verified[width]
    out = c.dev.create_empty_buffer(width * c.n + 16)
    outv = {m: c.dev.create_empty_buffer(c.nb + 16) for m in (0, 1, 2)}
    host_bits = O.synth_bits(c.n_values, SEED + 10, 0, 0.5)  # n_values / 8 bytes on the host: the generator's bitmap
    try:
        for mode in (1, 0, 2):
            c.p.set_tuning("gather_bucket", mode)
            capi.call("agpu_memset", h, vp(outv[mode]), 0xEE, c.nb + 16)
            capi.call("agpu_take_validity", h, width, vp(c.values), c.n_values, vp(c.vbits), vp(c.idx), vp(out), vp(outv[mode]), c.n)
            c.p.sync()
            assert c.count_equal(dt, out, ref_vals, c.n) == c.n, (width, mode)
            assert (c.download(outv[mode], c.nb, 16) == 0xEE).all(), (width, mode)
        # the direct form's bitmap against the oracle: eight 65 536-row windows spread over the column (+ the ragged end)
        starts = [int(x) // 64 * 64 for x in np.linspace(0, c.n - 65536, 8)] + [(c.n - 65536) // 64 * 64]
        for r0 in starts:
            k = min(65536, c.n - r0)
            idx = c.download(c.idx, 4 * r0, 4 * k).view(np.uint32)
            exp = O.take_bits(host_bits, c.n_values, idx)
            got = c.download(outv[1], r0 // 8, (k + 7) // 8)
            assert np.array_equal(np.unpackbits(got, bitorder="little")[:k], np.unpackbits(exp, bitorder="little")[:k]), r0
        cs = {m: c.checksum(outv[m], c.nb) for m in outv}
        assert cs[0] == cs[1] == cs[2]  # padding bits 0 in every form, every bit equal
        # and the stand-alone Boolean take (bits as data): every form against the verified bitmap
        ob = c.dev.create_empty_buffer(c.nb + 16)
        for mode in (1, 0, 2):
            c.p.set_tuning("gather_bucket", mode)
            capi.call("agpu_memset", h, vp(ob), 0xEE, c.nb + 16)
            capi.call("agpu_take_bits", h, vp(c.vbits), c.n_values, vp(c.idx), vp(ob), c.n)
            c.p.sync()
            assert c.checksum(ob, c.nb) == cs[1], ("take_bits", mode)
    finally:
        c.p.set_tuning("gather_bucket", 0)


CORNERS = [("random", "random"), ("random", "sequential"), ("sequential", "random"), ("sequential", "sequential")]


@pytest.mark.parametrize("width", [4, 2])
def test_put_all_four_corners_pinned_through_the_inverse_take(cols, width):
    """dst[dst_idx[i]] = src[src_idx[i]] with a bijective (or sequential) destination column: gathering the destination back through
    dst_idx must give exactly take(src, src_idx) — the column verified against numpy above — and the wrapping sum of the destination
    pins the slots no row touched"""
    c = cols
    h = c.p._handle
    dt = {4: capi.U32, 2: capi.U16}[width]
    if width not in c.verified:
        pytest.skip("the plain take of this width did not verify")
    n = c.n_put
    taken = c.verified[width]  # rows 0..n of take(values, idx)
    dst = c.dev.create_empty_buffer(width * c.n_dst + 16)
    back = c.dev.create_empty_buffer(width * n + 16)
    pattern = 0x5A
    words_total = width * c.n_dst // 4  # the destination as u32 words for the wrapping sum (n_dst even for 16-bit: checked below)
    assert (width * c.n_dst) % 4 == 0 or width == 2
    try:
        for src_kind, dst_kind in CORNERS:
            src_idx = c.idx if src_kind == "random" else c.iota
            dst_idx = c.perm if dst_kind == "random" else c.iota
            # what the inverse gather must return: take(values, src_idx)[0..n)
            if src_kind == "random":
                expect = taken
            else:  # src_idx = iota: the source column itself
                expect = c.values
            for mode in (0, 2, 1):  # auto (the probe decides), the forced pipeline, the direct scatter
                c.p.set_tuning("gather_bucket", mode)
                capi.call("agpu_memset", h, vp(dst), pattern, width * c.n_dst + 16)
                capi.call("agpu_put_bounded", h, width, vp(c.values), c.n_values, vp(src_idx), vp(dst), c.n_dst, vp(dst_idx), n)
                c.p.sync()
                c.p.set_tuning("gather_bucket", 1)  # the inverse through the direct kernel (verified against numpy at this size)
                capi.call("agpu_take", h, width, vp(dst), c.n_dst, vp(dst_idx), vp(back), n)
                c.p.sync()
                tag = (width, src_kind, dst_kind, mode)
                assert c.count_equal(dt, back, expect, n) == n, tag
                assert (c.download(dst, width * c.n_dst, 16) == pattern).all(), tag
                if width == 4:  # untouched slots: sum(dst) == sum(put values) + pattern word · (n_dst − n)  (mod 2^32)
                    s_put = c.wsum_u32(expect, n)
                    s_dst = c.wsum_u32(dst, c.n_dst)
                    assert s_dst == (s_put + 0x5A5A5A5A * (c.n_dst - n)) & 0xFFFFFFFF, tag
                elif dst_kind == "sequential":  # 16-bit: the tail behind the n written rows is untouched
                    tail = c.download(dst, width * n, min(1 << 20, width * (c.n_dst - n)))
                    assert (tail == pattern).all(), tag
    finally:
        c.p.set_tuning("gather_bucket", 0)


def test_put_bits_pinned_through_the_inverse_bit_take(cols):
    """Boolean put with a bijective destination column into a zeroed bitmap: taking the destination's bits back through dst_idx gives
    take_bits(src, src_idx) (verified against the oracle above), and the destination's popcount equals that result's"""
    c = cols
    h = c.p._handle
    n = c.n_put
    nb = O.bitmap_bytes(n)
    nbd = O.bitmap_bytes(c.n_dst)
    host_bits = O.synth_bits(c.n_values, SEED + 10, 0, 0.5)
    expect = c.dev.create_empty_buffer(nb + 16)
    c.p.set_tuning("gather_bucket", 1)
    capi.call("agpu_take_bits", h, vp(c.vbits), c.n_values, vp(c.idx), vp(expect), n)
    c.p.sync()
    for r0 in (0, (n // 2) // 64 * 64, (n - 65536) // 64 * 64):  # anchor: the direct bit take against the oracle
        idx = c.download(c.idx, 4 * r0, 4 * 65536).view(np.uint32)
        assert np.array_equal(c.download(expect, r0 // 8, 8192), O.take_bits(host_bits, c.n_values, idx)[:8192]), r0
    cs_exp, pop_exp = c.checksum(expect, nb), c.popcount(expect, n)
    dst = c.dev.create_empty_buffer(nbd + 16)
    back = c.dev.create_empty_buffer(nb + 16)
    try:
        for mode in (0, 2, 1):
            c.p.set_tuning("gather_bucket", mode)
            capi.call("agpu_memset", h, vp(dst), 0, nbd + 16)
            capi.call("agpu_put_bits_bounded", h, vp(c.vbits), c.n_values, vp(c.idx), vp(dst), c.n_dst, vp(c.perm), n)
            c.p.sync()
            assert c.popcount(dst, c.n_dst) == pop_exp, mode           # nothing set outside the n destinations
            c.p.set_tuning("gather_bucket", 1)
            capi.call("agpu_take_bits", h, vp(dst), c.n_dst, vp(c.perm), vp(back), n)
            c.p.sync()
            assert c.checksum(back, nb) == cs_exp, mode
            # and clearing: a destination of all ones, the same put → the complement pattern survives only where the source bit is 1
            c.p.set_tuning("gather_bucket", mode)
            capi.call("agpu_memset", h, vp(dst), 0xFF, nbd)
            capi.call("agpu_put_bits_bounded", h, vp(c.vbits), c.n_values, vp(c.idx), vp(dst), c.n_dst, vp(c.perm), n)
            c.p.sync()
            c.p.set_tuning("gather_bucket", 1)
            capi.call("agpu_take_bits", h, vp(dst), c.n_dst, vp(c.perm), vp(back), n)
            c.p.sync()
            assert c.checksum(back, nb) == cs_exp, ("ones", mode)
    finally:
        c.p.set_tuning("gather_bucket", 0)
