"""GPU: the kernels that walk big columns as TWO lock-step streams half a column apart (common.hpp two_streams: one-input element-wise
kernels, ×2 width changes, array-free chains, device copies — from 128 MiB on) give the bytes of the oracle over the WHOLE column, at sizes
on both sides of the threshold, with an odd tile count (the last tile belongs to neither stream) and a sub-tile tail."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi
from gpu_util import Dev, bits_equal, nan_aware_bits_equal

pytestmark = pytest.mark.gpu

# 2^25 rows of f32 = 128 MiB = the threshold; + 256·3 rows: an odd number of extra tiles; + 77: a tail launch
SIZES = [(1 << 25) - 256, 1 << 25, (1 << 25) + 256 * 3 + 77, 3 * (1 << 24) + 12345]


@pytest.fixture(scope="module")
def _dev():
    return Dev()


@pytest.fixture()
def D(_dev):
    yield _dev
    _dev.release()


def _within_one_ulp(got, exp) -> bool:
    return int(np.max(np.abs(got.view(np.int32).astype(np.int64) - exp.view(np.int32).astype(np.int64)))) <= 1


class _Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]


def _chain(*items):
    arr = (_Step * len(items))()
    for k, (op, kind, operand) in enumerate(items):
        arr[k].op, arr[k].kind, arr[k].operand = op, kind, (operand.vp.value if operand is not None else None)
    return arr, len(items)


@pytest.mark.parametrize("n", SIZES)
def test_unary_and_scalar_f32_whole_column(D, n):
    x = O.synth_f32(n, 11, 0, -50.0, 50.0)
    dx, out = D.up(x), D.empty(4 * n)
    for op in (capi.UN_NEG, capi.UN_SQRT):
        D.call("agpu_unary", op, capi.F32, dx.vp, out.vp, n)
        assert nan_aware_bits_equal(D.down(out, np.float32, n), O.unary(op, O.F32, x)), op
    D.call("agpu_unary", capi.UN_SIN, capi.F32, dx.vp, out.vp, n)
    assert _within_one_ulp(D.down(out, np.float32, n), O.unary(capi.UN_SIN, O.F32, x))
    s = np.array([1.25], np.float32)
    ds = D.up(s)
    for op in (capi.OP_ADD, capi.OP_MUL):
        D.call("agpu_scalar", op, capi.F32, dx.vp, ds.vp, out.vp, n)
        assert bits_equal(D.down(out, np.float32, n), O.scalar(op, O.F32, x, s)), op
    # in place: the same mapping reads and writes the same rows
    D.call("agpu_unary", capi.UN_NEG, capi.F32, dx.vp, dx.vp, n)
    assert bits_equal(D.down(dx, np.float32, n), O.unary(capi.UN_NEG, O.F32, x))


@pytest.mark.parametrize("n", [(1 << 26) + 512 * 3 + 5, 1 << 26, (1 << 26) - 512])
def test_x2_casts_whole_column(D, n):
    rng = np.random.default_rng(5)
    u16 = rng.integers(0, 65536, n, dtype=np.uint16)
    du, out = D.up(u16), D.empty(4 * n)
    D.call("agpu_cast", capi.U16, capi.F32, du.vp, out.vp, n)
    assert bits_equal(D.down(out, np.float32, n), O.cast(O.U16, O.F32, u16))
    D.call("agpu_cast", capi.I16, capi.I32, du.vp, out.vp, n)
    assert bits_equal(D.down(out, np.int32, n), O.cast(O.I16, O.I32, u16.view(np.int16)))
    f = O.synth_f32(n, 12, 0, -40000.0, 70000.0)
    df = D.up(f)
    D.call("agpu_cast", capi.F32, capi.I16, df.vp, out.vp, n)
    assert bits_equal(D.down(out, np.int16, n), O.cast(O.F32, O.I16, f))
    D.call("agpu_cast", capi.F32, capi.U16, df.vp, out.vp, n)
    assert bits_equal(D.down(out, np.uint16, n), O.cast(O.F32, O.U16, f))
    # u8 → u16: 1 KiB in, 2 KiB out per chunk — two streams from 2^26 rows on
    u8 = rng.integers(0, 256, n, dtype=np.uint8)
    d8 = D.up(u8)
    D.call("agpu_cast", capi.U8, capi.U16, d8.vp, out.vp, n)
    assert bits_equal(D.down(out, np.uint16, n), O.cast(O.U8, O.U16, u8))
    # the fused 16-bit sin = cast → sin per row through the same kernel
    D.call("agpu_unary", capi.UN_SIN, capi.U16, du.vp, out.vp, n)
    got = D.down(out, np.float32, n)
    exp = O.unary(capi.UN_SIN, O.F32, O.cast(O.U16, O.F32, u16))
    assert _within_one_ulp(got, exp)


@pytest.mark.parametrize("n", [(1 << 25) + 256 * 5 + 3, (1 << 26) + 512 * 3 + 9])
def test_array_free_chains_equal_the_unfused_sequence(D, n):
    x = O.synth_f32(n, 13, 0, -20.0, 20.0)
    dx, out, ref = D.up(x), D.empty(4 * n), D.empty(4 * n)
    s, t = D.up(np.array([0.37], np.float32)), D.up(np.array([100.0], np.float32))
    st, ns = _chain((capi.OP_ADD, 1, t), (capi.OP_MUL, 1, s))
    D.call("agpu_fused_chain", capi.F32, dx.vp, C.cast(st, C.c_void_p), ns, out.vp, n)
    exp = O.scalar(capi.OP_MUL, O.F32, O.scalar(capi.OP_ADD, O.F32, x, np.array([100.0], np.float32)), np.array([0.37], np.float32))
    assert bits_equal(D.down(out, np.float32, n), exp)
    st2, ns2 = _chain((capi.OP_MUL, 1, s), (capi.UN_SIN, 0, None))
    D.call("agpu_fused_chain", capi.F32, dx.vp, C.cast(st2, C.c_void_p), ns2, out.vp, n)
    D.call("agpu_scalar", capi.OP_MUL, capi.F32, dx.vp, s.vp, ref.vp, n)
    D.call("agpu_unary", capi.UN_SIN, capi.F32, ref.vp, ref.vp, n)
    assert bits_equal(D.down(out, np.float32, n), D.down(ref, np.float32, n))
    # a 16-bit column at the head (cast_chain_kernel): cast(u16)·s + t in one launch against cast, mul, add
    u16 = np.random.default_rng(6).integers(0, 65536, n, dtype=np.uint16)
    du = D.up(u16)
    st3, ns3 = _chain((capi.OP_MUL, 1, s), (capi.OP_ADD, 1, t))
    D.call("agpu_fused_cast_chain", capi.U16, du.vp, C.cast(st3, C.c_void_p), ns3, out.vp, n)
    exp3 = O.scalar(capi.OP_ADD, O.F32, O.scalar(capi.OP_MUL, O.F32, O.cast(O.U16, O.F32, u16), np.array([0.37], np.float32)), np.array([100.0], np.float32))
    assert bits_equal(D.down(out, np.float32, n), exp3)


@pytest.mark.parametrize("nbytes", [(128 << 20) - 1024, 128 << 20, (128 << 20) + 1024 * 3 + 16, (160 << 20) + 16 * 5 + 7])
def test_device_copy_whole_buffer(D, nbytes):
    src = np.random.default_rng(7).integers(0, 256, nbytes, dtype=np.uint8)
    ds, dd = D.up(src), D.empty(nbytes)
    D.call("agpu_copy", dd.vp, ds.vp, nbytes)
    assert bits_equal(D.down(dd, np.uint8, nbytes), src)
