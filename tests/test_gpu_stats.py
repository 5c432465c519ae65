"""GPU: whole-column f32 statistics in ONE pass (agpu_reduce_stats_f32 / agpu_comm_reduce_stats_f32; north_star config 5 names sum / min / max
of one column).  Every field must be BIT-IDENTICAL to the separate reduction of the same column — agpu_reduce(SUM) in the reference's tree
order [aggregate_kernels.rs:24-51], agpu_reduce(MIN / MAX) under Arrow's NaN rule, agpu_reduce_sum_f64 — and to the oracle, on the fused
path (aligned, no validity, >= 2^20 rows) and on the fallback (small, null-aware, unaligned)."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi
from gpu_util import Dev, nan_aware_bits_equal, rand_values

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def _dev():
    return Dev()


@pytest.fixture()
def D(_dev):
    yield _dev
    _dev.release()


def _record(D, buf):
    raw = D.down(buf, np.uint8, 24)
    f = raw[:12].view(np.float32)
    return {"sum": f[0:1].copy(), "min": f[1:2].copy(), "max": f[2:3].copy(), "sum_f64": raw[16:24].view(np.float64).copy(), "reserved": int(raw[12:16].view(np.uint32)[0])}


def _separate(D, dx, validity, n):
    out = D.empty(16)
    res = {}
    for key, op in (("sum", capi.RED_SUM), ("min", capi.RED_MIN), ("max", capi.RED_MAX)):
        D.call("agpu_reduce", op, capi.F32, dx.vp, validity, n, out.vp)
        res[key] = D.down(out, np.float32, 1)
    D.call("agpu_reduce_sum_f64", dx.vp, validity, n, out.vp)
    res["sum_f64"] = D.down(out, np.float64, 1)
    return res


def _check(D, x, validity_bits=None, offset=0):
    n = len(x)
    dx = D.up(x, offset)
    dv = D.up(validity_bits) if validity_bits is not None else None
    rec = D.empty(32)
    D.call("agpu_reduce_stats_f32", dx.vp, dv.vp if dv else None, n, rec.vp)
    got = _record(D, rec)
    assert got["reserved"] == 0
    sep = _separate(D, dx, dv.vp if dv else None, n)
    for key in ("sum", "min", "max"):
        assert nan_aware_bits_equal(got[key], sep[key]), (key, n, got[key], sep[key])
        exp = np.array([O.reduce({"sum": O.RED_SUM, "min": O.RED_MIN, "max": O.RED_MAX}[key], O.F32, x, validity_bits)], np.float32)
        assert nan_aware_bits_equal(got[key], exp), ("oracle", key, n, got[key], exp)
    assert got["sum_f64"].view(np.uint64)[0] == sep["sum_f64"].view(np.uint64)[0] or (np.isnan(got["sum_f64"][0]) and np.isnan(sep["sum_f64"][0])), (n, got["sum_f64"], sep["sum_f64"])


@pytest.mark.parametrize("n", [0, 1, 1000, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, (1 << 20) + 16383, 3 * (1 << 20) + 5000 + 3,
                               (1 << 24) + 7, (1 << 24) + (1 << 22) + 16384 * 3 + 77, 3 * (1 << 24) + 70001])
def test_stats_equal_the_four_reductions(D, n):
    x = O.synth_f32(n, 31, 0, -1000.0, 1000.0)
    _check(D, x)


def test_stats_specials_on_the_fused_path(D):
    n = (1 << 21) + 4321
    x = rand_values(capi.F32, n, 41)      # ±0, ±inf, NaN, denormals, extremes among ordinary values
    x[np.isinf(x)] = 1.0                  # (keep the sums finite so that the comparison says something)
    x[:16384] = np.nan                    # a whole quarter of NaN
    x[16384 * 5 + 7] = np.nan
    x.view(np.uint32)[16384 * 9 + 3] = 0x7FA00000   # a SIGNALLING NaN among ordinary rows (the kernel quiets before v_min / v_max)
    x.view(np.uint32)[16384 * 11: 16384 * 12] = 0xFFA00001  # … and a whole quarter of them
    _check(D, x)
    _check(D, np.full(n, np.nan, np.float32))          # nothing but NaN: min = max = NaN
    w = np.zeros(n, np.float32); w[16384 * 3 + 1] = -0.0
    _check(D, w)                                      # -0.0 < +0.0
    dn = (np.arange(n, dtype=np.int64) % 2001 - 1000).astype(np.float32) * np.float32(1e-42)   # nothing but denormals and zeros: kept, not flushed
    _check(D, dn)
    y = O.synth_f32(n, 32, 0, -1.0, 1.0); y[-1] = np.float32(3.0e38); y[-2] = np.float32(-3.0e38)
    _check(D, y)                                      # the extremes in the < 1-quarter tail


def test_stats_null_aware_in_one_pass_and_the_unaligned_fallback(D):
    n = (1 << 21) + 999
    x = O.synth_f32(n, 33, 0, -10.0, 10.0)
    _check(D, x, validity_bits=O.synth_bits(n, 34, 0, 0.8))
    # quarters without a valid row, quarters whose valid rows are all NaN, both in one column; no valid row at all; a valid extreme in the tail
    v = np.unpackbits(O.synth_bits(n, 35, 0, 0.6), bitorder="little")[:n].astype(bool)
    v[:16384] = False
    y = x.copy(); y[16384 * 2: 16384 * 3] = np.nan; v[16384 * 4: 16384 * 5] = True; y[16384 * 4: 16384 * 5] = np.nan
    _check(D, y, validity_bits=O.pack_bits(v))
    only_nan = y.copy(); only_nan[v] = np.nan
    _check(D, only_nan, validity_bits=O.pack_bits(v))      # every valid row NaN → min = max = NaN (the null quarters must not turn that into ±inf)
    _check(D, x, validity_bits=O.pack_bits(np.zeros(n, bool)))  # nothing valid: sum 0, min +inf, max −inf
    t = v.copy(); t[-1] = True
    z = x.copy(); z[-1] = np.float32(-3.0e38)
    _check(D, z, validity_bits=O.pack_bits(t))
    _check(D, x[: 1 << 20], offset=4)     # only 4-byte aligned: the four reductions one after the other


def test_stats_through_the_hosts_and_the_communicator(D):
    import arrow_gpu_amd as ag
    from arrow_gpu_amd.sharding import Communicator

    n = (1 << 22) + 12345
    x = O.synth_f32(n, 35, 0, -1.0, 1.0)
    dev = D.dev
    col = ag.Float32ArrayGPU.from_slice(x, dev)
    v = col.stats().values()
    assert np.float32(v["sum"]).tobytes() == np.float32(O.reduce(O.RED_SUM, O.F32, x)).tobytes()
    assert np.float32(v["min"]) == x.min() and np.float32(v["max"]) == x.max()
    assert abs(float(v["sum_f64"]) - float(np.sum(x.astype(np.float64)))) <= 1e-9 * float(np.sum(np.abs(x.astype(np.float64))))
    assert np.float32(v["sum"]).tobytes() == np.float32(col.sum().raw_values()[0]).tobytes()
    # one rank: the collective form leaves the local statistics (the combine of one record is the record)
    comm = Communicator.single(dev)
    try:
        dx, rec = D.up(x), D.empty(32)
        comm.reduce_stats_f32(D.p, dx.buf, None, n, rec.buf)
        comm.sync(D.p)
        got = _record(D, rec)
        sep = _separate(D, dx, None, n)
        for key in ("sum", "min", "max"):
            assert nan_aware_bits_equal(got[key], sep[key]), key
        assert got["sum_f64"].view(np.uint64)[0] == sep["sum_f64"].view(np.uint64)[0]
    finally:
        comm.close()


def test_stats_at_1e9_rows_equal_the_four_reductions_and_read_the_column_once(D):
    """BASELINE config 5's shard size: the one-pass form against the four separate reductions, bit for bit, and its time against theirs."""
    n = 1_000_000_000
    buf = D.dev.create_empty_buffer(4 * n)
    D.call("agpu_synth_f32", C.c_void_p(buf.ptr), n, 77, 0, C.c_float(-1.0), C.c_float(1.0))
    from gpu_util import _Ptr

    dx = _Ptr(buf, 0)
    rec = D.empty(32)
    D.call("agpu_reduce_stats_f32", dx.vp, None, n, rec.vp)
    got = _record(D, rec)
    sep = _separate(D, dx, None, n)
    for key in ("sum", "min", "max"):
        assert nan_aware_bits_equal(got[key], sep[key]), (key, got[key], sep[key])
    assert got["sum_f64"].view(np.uint64)[0] == sep["sum_f64"].view(np.uint64)[0]
    assert -1.0 <= float(got["min"][0]) < -0.999999 and 0.999999 < float(got["max"][0]) < 1.0


def test_stats_two_stream_kernels_and_take_columns_inside_a_captured_graph(D):
    """this session's forms under hipGraph capture: a one-input kernel over a column big enough for the two-stream order, the reductions'
    single finishing launch, the one-pass statistics, agpu_take_columns (direct kernels while capturing) — replayed with new inputs in the
    same buffers"""
    n = (1 << 25) + 256 * 3 + 7
    x = O.synth_f32(n, 51, 0, -5.0, 5.0)
    dx, neg, red, rec = D.up(x), D.empty(4 * n), D.empty(16), D.empty(32)
    idx = np.random.default_rng(9).integers(0, n, 50_000).astype(np.uint32)
    didx, t0, t1 = D.up(idx), D.empty(4 * len(idx)), D.empty(4 * len(idx))
    widths = (C.c_int32 * 2)(4, 4)
    vals = (C.c_void_p * 2)(dx.vp.value, neg.vp.value)
    outs = (C.c_void_p * 2)(t0.vp.value, t1.vp.value)

    def enqueue():
        D.call("agpu_unary", capi.UN_NEG, capi.F32, dx.vp, neg.vp, n)
        D.call("agpu_reduce", capi.RED_MAX, capi.F32, neg.vp, None, n, red.vp)
        D.call("agpu_reduce_stats_f32", dx.vp, None, n, rec.vp)
        D.call("agpu_take_columns", 2, widths, vals, n, didx.vp, outs, len(idx))

    enqueue()  # warm-up: scratch grows here, not during capture
    D.p.sync()
    g = C.c_void_p()
    D.call("agpu_pipeline_begin_capture")
    enqueue()
    D.call("agpu_pipeline_end_capture", C.byref(g))
    for rep in range(3):
        if rep:
            x = O.synth_f32(n, 60 + rep, 0, -5.0 - rep, 5.0)
            capi.call("agpu_upload", D.h, dx.vp, x.ctypes.data_as(C.c_void_p), x.nbytes)
        capi.call("agpu_memset", D.h, rec.vp, 0xCD, 32)
        capi.call("agpu_graph_launch", g, D.h)
        assert nan_aware_bits_equal(D.down(neg, np.float32, n), -x)
        assert D.down(red, np.float32, 1)[0] == np.float32(-x.min())
        got = _record(D, rec)
        assert got["reserved"] == 0
        assert nan_aware_bits_equal(got["sum"], np.array([O.reduce(O.RED_SUM, O.F32, x)], np.float32))
        assert got["min"][0] == x.min() and got["max"][0] == x.max()
        assert nan_aware_bits_equal(D.down(t0, np.float32, len(idx)), x[idx]) and nan_aware_bits_equal(D.down(t1, np.float32, len(idx)), -x[idx])
    capi.call("agpu_graph_destroy", g)
