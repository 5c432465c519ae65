"""GPU: null-aware put (SURVEY §8f-3 — `todo!()` in the reference, crates/routines/src/lib.rs:164-169).
dst.values[dst_idx[i]] = src.values[src_idx[i]] and the validity bit travels with the value; an absent bitmap counts
as all-valid and dst gains one when src has nulls.  Expected results are plain numpy scatter on (values, valid)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(ag, cls, vals, valid, dev):
    if valid is None:
        return cls.from_slice(list(vals), dev)
    return cls.from_optional_slice([v if ok else None for v, ok in zip(vals, valid)], dev)


def expected(src_vals, src_valid, dst_vals, dst_valid, si, di):
    sv = np.ones(len(src_vals), bool) if src_valid is None else np.asarray(src_valid)
    dv = np.ones(len(dst_vals), bool) if dst_valid is None else np.asarray(dst_valid).copy()
    # from_optional_slice stores 0 at null slots (primitive_array_gpu.rs:39-41); the raw value travels as it is
    vals = np.where(dv, np.array(dst_vals), 0.0)
    vals[di] = np.where(sv, np.asarray(src_vals), 0.0)[si]
    dv[di] = sv[si]
    return vals, dv


@pytest.mark.parametrize("src_nulls,dst_nulls", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("n", [7, 1000, 70_001])
def test_put_carries_validity(ag, src_nulls, dst_nulls, n):
    dev = ag.GPU_DEVICE()
    rng = np.random.default_rng(n + 2 * src_nulls + dst_nulls)
    n_src, n_dst, k = n, n + 13, max(1, n // 2)
    src_vals = [float(x) for x in rng.integers(-1000, 1000, n_src)]
    dst_vals = [float(x) for x in rng.integers(-1000, 1000, n_dst)]
    src_valid = (rng.random(n_src) < 0.7) if src_nulls else None
    dst_valid = (rng.random(n_dst) < 0.7) if dst_nulls else None
    si = rng.integers(0, n_src, k).astype(np.uint32)
    di = rng.permutation(n_dst)[:k].astype(np.uint32)  # unique destinations (duplicates are unspecified)
    src = make(ag, ag.Float32ArrayGPU, src_vals, src_valid, dev)
    dst = make(ag, ag.Float32ArrayGPU, dst_vals, dst_valid, dev)
    src.put(ag.UInt32ArrayGPU.from_slice(si, dev), dst, ag.UInt32ArrayGPU.from_slice(di, dev))
    exp_vals, exp_valid = expected(src_vals, src_valid, dst_vals, dst_valid, si, di)
    assert np.array_equal(dst.raw_values(), exp_vals.astype(np.float32))
    got = dst.values()
    import oracle.model as M  # the oracle-side restatement of the host rules gives the same array

    msrc, mdst = make(M, M.Float32ArrayGPU, src_vals, src_valid, None), make(M, M.Float32ArrayGPU, dst_vals, dst_valid, None)
    msrc.put(M.UInt32ArrayGPU.from_slice(si), mdst, M.UInt32ArrayGPU.from_slice(di))
    assert mdst.values() == got
    if src_valid is None and dst_valid is None:
        assert dst.null_buffer is None
    else:
        assert dst.null_buffer is not None and dst.null_buffer.len == n_dst
        assert [g is not None for g in got] == list(exp_valid)


def test_boolean_put_carries_validity(ag):
    dev = ag.GPU_DEVICE()
    rng = np.random.default_rng(5)
    n = 5000
    sv, dv = rng.random(n) < 0.5, rng.random(n) < 0.5
    s_ok, d_ok = rng.random(n) < 0.8, rng.random(n) < 0.8
    src = ag.BooleanArrayGPU.from_optional_slice([bool(v) if ok else None for v, ok in zip(sv, s_ok)], dev)
    dst = ag.BooleanArrayGPU.from_optional_slice([bool(v) if ok else None for v, ok in zip(dv, d_ok)], dev)
    si = rng.integers(0, n, n // 3).astype(np.uint32)
    di = rng.permutation(n)[: n // 3].astype(np.uint32)
    src.put(ag.UInt32ArrayGPU.from_slice(si, dev), dst, ag.UInt32ArrayGPU.from_slice(di, dev))
    exp_v, exp_ok = np.where(d_ok, dv, False), d_ok.copy()  # from_optional_slice stores False at null slots
    exp_v[di] = np.where(s_ok, sv, False)[si]
    exp_ok[di] = s_ok[si]
    got = dst.values()
    assert [g is not None for g in got] == list(exp_ok)
    assert [bool(g) for g, ok in zip(got, exp_ok) if ok] == [bool(v) for v, ok in zip(exp_v, exp_ok) if ok]
