"""GPU: agpu_take_columns — the columns of one table taken by ONE index column (the merge-back pipeline's index work shared by all columns)
gives exactly what one agpu_take per column gives and what the oracle says, through every form: small (column by column), the forced
pipeline, the forced direct kernel, the auto policy with random and with sorted indices, mixed widths, the Python host with nulls."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi
from gpu_util import Dev, bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def _dev():
    return Dev()


@pytest.fixture()
def D(_dev):
    yield _dev
    _dev.p.set_tuning("gather_bucket", 0)
    _dev.release()


def _take_columns(D, cols, idx):
    k = len(cols)
    dcols = [D.up(c) for c in cols]
    didx = D.up(idx)
    outs = [D.empty(len(idx) * c.dtype.itemsize) for c in cols]
    widths = (C.c_int32 * k)(*[c.dtype.itemsize for c in cols])
    vals = (C.c_void_p * k)(*[d.vp.value for d in dcols])
    outp = (C.c_void_p * k)(*[o.vp.value for o in outs])
    D.call("agpu_take_columns", k, widths, vals, len(cols[0]) if cols else 1, didx.vp, outp, len(idx))
    return [D.down(o, c.dtype, len(idx)) for o, c in zip(outs, cols)]


def _table(rng, n_src, dtypes):
    return [rng.integers(0, np.iinfo(dt).max, n_src, dtype=dt) for dt in dtypes]


@pytest.mark.parametrize("mode", [0, 1, 2, 4])
@pytest.mark.parametrize("n_src,n_idx", [(1000, 777), (1 << 16, 70001), (1 << 20, (1 << 20) + 333), (3_000_017, 2_500_003)])
def test_take_columns_equals_column_by_column(D, mode, n_src, n_idx):
    rng = np.random.default_rng(n_src + mode)
    cols = _table(rng, n_src, [np.uint32, np.uint16, np.uint32, np.uint8, np.uint32])
    idx = rng.integers(0, n_src, n_idx, dtype=np.uint32)
    D.p.set_tuning("gather_bucket", mode)
    got = _take_columns(D, cols, idx)
    for c, g in zip(cols, got):
        assert bits_equal(g, O.take(c.dtype.itemsize, c, idx)), (mode, n_src, c.dtype)


def test_take_columns_auto_policy_random_and_sorted_indices(D):
    rng = np.random.default_rng(5)
    n = (1 << 25) + 4099            # the auto policy's pipeline size for 4-byte sources
    cols = _table(rng, n, [np.uint32, np.uint32, np.uint16])
    for idx in (rng.integers(0, n, n, dtype=np.uint32), np.sort(rng.integers(0, n, n, dtype=np.uint32))):
        got = _take_columns(D, cols, idx)
        for c, g in zip(cols, got):
            assert bits_equal(g, O.take(c.dtype.itemsize, c, idx)), c.dtype
        D.release()


@pytest.mark.parametrize("mode", [0, 1, 2, 4])
def test_take_columns_with_validity_bitmaps(D, mode):
    """columns with and without nulls in one call: values as above, out_validity bit i = validity bit idx[i] (agpu_take_validity's rule)"""
    rng = np.random.default_rng(40 + mode)
    n_src, n_idx = 3_000_017, 2_500_003
    cols = _table(rng, n_src, [np.uint32, np.uint32, np.uint16, np.uint8])
    vbits = [O.synth_bits(n_src, 50, 0, 0.8), None, O.synth_bits(n_src, 51, 0, 0.5), O.synth_bits(n_src, 52, 0, 0.95)]
    idx = rng.integers(0, n_src, n_idx, dtype=np.uint32)
    D.p.set_tuning("gather_bucket", mode)
    k = len(cols)
    dcols, didx = [D.up(c) for c in cols], D.up(idx)
    dv = [D.up(v) if v is not None else None for v in vbits]
    outs = [D.empty(n_idx * c.dtype.itemsize) for c in cols]
    outv = [D.empty(O.bitmap_bytes(n_idx)) if v is not None else None for v in vbits]
    widths = (C.c_int32 * k)(*[c.dtype.itemsize for c in cols])
    vals = (C.c_void_p * k)(*[d.vp.value for d in dcols])
    vb = (C.c_void_p * k)(*[d.vp.value if d is not None else None for d in dv])
    outp = (C.c_void_p * k)(*[o.vp.value for o in outs])
    voutp = (C.c_void_p * k)(*[o.vp.value if o is not None else None for o in outv])
    D.call("agpu_take_columns_validity", k, widths, vals, vb, n_src, didx.vp, outp, voutp, n_idx)
    for c, o, v, ov in zip(cols, outs, vbits, outv):
        assert bits_equal(D.down(o, c.dtype, n_idx), O.take(c.dtype.itemsize, c, idx)), (mode, c.dtype)
        if v is not None:
            assert bits_equal(D.down(ov, np.uint8, O.bitmap_bytes(n_idx)), O.take_bits(v, n_src, idx)), (mode, "validity", c.dtype)


def test_take_columns_edges(D):
    rng = np.random.default_rng(6)
    cols = _table(rng, 5000, [np.uint32])
    idx = rng.integers(0, 5000, 100, dtype=np.uint32)
    assert bits_equal(_take_columns(D, cols, idx)[0], O.take(4, cols[0], idx))          # one column
    assert _take_columns(D, [], idx) == []                                            # no column
    assert len(_take_columns(D, cols, np.zeros(0, np.uint32))[0]) == 0                # no index


def test_take_columns_through_the_python_host_with_nulls(D):
    import arrow_gpu_amd as ag

    dev = D.dev
    rng = np.random.default_rng(7)
    n = 50_000
    a = ag.Int32ArrayGPU.from_optional_slice([None if rng.random() < 0.1 else int(v) for v in rng.integers(-1000, 1000, n)], dev)
    b = ag.Float32ArrayGPU.from_slice(rng.standard_normal(n).astype(np.float32), dev)
    c = ag.UInt16ArrayGPU.from_slice(rng.integers(0, 65535, n).astype(np.uint16), dev)
    idx = ag.UInt32ArrayGPU.from_slice(rng.integers(0, n, 30_000).astype(np.uint32), dev)
    m = ag.BooleanArrayGPU.from_optional_slice([None if rng.random() < 0.05 else bool(v) for v in rng.integers(0, 2, n)], dev)
    ta, tm, tb, tc = ag.take_columns([a, m, b, c], idx)      # a Boolean column among them: taken on its own, in its place
    assert tm.values() == m.take(idx).values()
    assert ta.values() == a.take(idx).values()
    assert tb.values() == b.take(idx).values()
    assert tc.values() == c.take(idx).values()
    assert ta.null_buffer is not None and tb.null_buffer is None
