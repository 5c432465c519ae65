"""GPU parity tests proper: every kernel family of the C ABI against the CPU oracle on the same seeded inputs —
bit-exact for integer / compare / cast / logical / bitmap / swizzle / f32 + − × ÷ % min max / f32 SUM (reference
order), ≤ MAX_ULP for the transcendental f32 functions (tolerance stated in golden_runner.MAX_ULP).
Sizes cover empty, ragged tails around every tile boundary (64, 256, 1024, 4096, 65536) and ≈1 M rows; pointers
are also deliberately mis-aligned to force the element-granular fallbacks."""
import ctypes as C

import numpy as np
import pytest

import golden_runner as G
import oracle as O
from arrow_gpu_amd import _capi as capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def _dev():
    from gpu_util import Dev

    return Dev()


@pytest.fixture()
def D(_dev):
    yield _dev
    _dev.release()  # buffers created by a test live until its end (kernels are asynchronous)


from gpu_util import (ALL_DTYPES, INT_DTYPES, NP, SIZES, SMALL_SIZES, bits_equal, max_ulp, nan_aware_bits_equal,  # noqa: E402
                      rand_values)

F32_BIN = [capi.OP_ADD, capi.OP_SUB, capi.OP_MUL, capi.OP_DIV, capi.OP_REM, capi.OP_MIN, capi.OP_MAX]
INT32_BIN = [capi.OP_ADD, capi.OP_SUB, capi.OP_MUL, capi.OP_DIV, capi.OP_REM, capi.OP_MIN, capi.OP_MAX, capi.OP_AND,
             capi.OP_OR, capi.OP_XOR]
SMALL_BIN = [capi.OP_ADD, capi.OP_SUB, capi.OP_MUL, capi.OP_MIN, capi.OP_MAX, capi.OP_AND, capi.OP_OR, capi.OP_XOR]


def ops_for(dtype):
    if dtype == capi.F32:
        return F32_BIN
    return INT32_BIN if NP[dtype]().itemsize == 4 else SMALL_BIN


# ------------------------------------------------------------------ element-wise
@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_binary_and_scalar_all_ops(D, dtype):
    for n in SIZES:
        a, b = rand_values(dtype, n, 100 + n), rand_values(dtype, n, 200 + n)
        if n > 4 and dtype != capi.F32:
            b[:4] = 0  # division by zero / shift edge
            if NP[dtype]().itemsize == 4 and np.issubdtype(NP[dtype], np.signedinteger):
                a[0], b[0] = np.iinfo(np.int32).min, -1
        da, db = D.up(a), D.up(b)
        out = D.empty(max(a.nbytes, 1))
        s = rand_values(dtype, 1, 7, special=False)
        ds = D.up(s)
        for op in ops_for(dtype):
            D.call("agpu_binary", op, dtype, da.vp, db.vp, out.vp, n)
            assert nan_aware_bits_equal(D.down(out, NP[dtype], n), O.binary(op, dtype, a, b)), (op, n)
            D.call("agpu_scalar", op, dtype, da.vp, ds.vp, out.vp, n)
            assert nan_aware_bits_equal(D.down(out, NP[dtype], n), O.scalar(op, dtype, a, s)), ("scalar", op, n)


@pytest.mark.parametrize("dtype", [capi.F32, capi.I32, capi.U8, capi.I16])
def test_binary_unaligned_pointers_and_in_place(D, dtype):
    n = 10007
    w = NP[dtype]().itemsize
    a, b = rand_values(dtype, n, 1), rand_values(dtype, n, 2)
    da, db = D.up(a, offset_bytes=w), D.up(b, offset_bytes=w)  # aligned to the element only
    out = D.empty(a.nbytes, offset_bytes=w)
    D.call("agpu_binary", capi.OP_ADD, dtype, da.vp, db.vp, out.vp, n)
    assert nan_aware_bits_equal(D.down(out, NP[dtype], n), O.binary(capi.OP_ADD, dtype, a, b))
    # in place on an aligned buffer: out aliases a exactly
    da2 = D.up(a)
    D.call("agpu_binary", capi.OP_MUL, dtype, da2.vp, D.up(b).vp, da2.vp, n)
    assert nan_aware_bits_equal(D.down(da2, NP[dtype], n), O.binary(capi.OP_MUL, dtype, a, b))


@pytest.mark.parametrize("dtype", INT_DTYPES)
def test_shifts(D, dtype):
    for n in SMALL_SIZES:
        a = rand_values(dtype, n, 5)
        s = np.random.default_rng(n).integers(0, 40, n).astype(np.uint32)  # includes amounts ≥ width and ≥ 32
        da, dsh, out = D.up(a), D.up(s), D.empty(max(a.nbytes, 1))
        for op in (capi.OP_SHL, capi.OP_SHR):
            D.call("agpu_binary", op, dtype, da.vp, dsh.vp, out.vp, n)
            assert bits_equal(D.down(out, NP[dtype], n), O.binary(op, dtype, a, s)), (op, n)


@pytest.mark.parametrize("dtype", [capi.U8, capi.I8])
def test_every_operand_pair_of_the_8_bit_types(D, dtype):
    """all 65 536 (a, b) pairs of an 8-bit type through every binary op, both shifts (amounts 0 … 255 reach past the width and past 32),
    every compare and the scalar form — the whole input space of those kernels, bit-exact against the oracle"""
    info = np.iinfo(NP[dtype])
    vals = np.arange(info.min, info.max + 1, dtype=np.int64).astype(NP[dtype])
    a, b = np.repeat(vals, 256), np.tile(vals, 256)
    n = len(a)
    da, db, out = D.up(a), D.up(b), D.empty(n)
    for op in SMALL_BIN:
        D.call("agpu_binary", op, dtype, da.vp, db.vp, out.vp, n)
        assert bits_equal(D.down(out, NP[dtype], n), O.binary(op, dtype, a, b)), op
    amounts = np.tile(np.arange(256, dtype=np.uint32), 256)
    dam = D.up(amounts)
    for op in (capi.OP_SHL, capi.OP_SHR):
        D.call("agpu_binary", op, dtype, da.vp, dam.vp, out.vp, n)
        assert bits_equal(D.down(out, NP[dtype], n), O.binary(op, dtype, a, amounts)), op
    outb = D.empty(O.bitmap_bytes(n) + 8)
    for op in range(5):
        D.call("agpu_compare", op, dtype, da.vp, db.vp, outb.vp, n)
        assert bits_equal(D.down(outb, np.uint8, O.bitmap_bytes(n)), O.compare(op, dtype, a, b)), op
    for sv in (vals[0], vals[-1], vals[len(vals) // 2], vals[3]):
        sarr = np.array([sv], NP[dtype])
        for op in SMALL_BIN:
            D.call("agpu_scalar", op, dtype, da.vp, D.up(sarr).vp, out.vp, n)
            assert bits_equal(D.down(out, NP[dtype], n), O.scalar(op, dtype, a, sarr)), (op, sv)


def test_int32_power(D):
    n = 5003
    rng = np.random.default_rng(3)
    a = rng.integers(-6, 7, n).astype(np.int32)
    p = rng.integers(-5, 12, n).astype(np.int32)
    p[:3] = [np.iinfo(np.int32).min, 31, 0]
    out = D.empty(4 * n)
    D.call("agpu_binary", capi.OP_POW, capi.I32, D.up(a).vp, D.up(p).vp, out.vp, n)
    assert bits_equal(D.down(out, np.int32, n), O.binary(O.OP_POW, O.I32, a, p))


def test_f32_power_domain_and_accuracy(D):
    rng = np.random.default_rng(4)
    n = 100003
    a = np.abs(rng.standard_normal(n)).astype(np.float32) * 10
    b = (rng.standard_normal(n) * 3).astype(np.float32)
    a[:8] = [1.0, -1.0, 10.0, -10.0, np.nan, np.inf, -np.inf, 0.0]
    b[:8] = [0.0, 0.0, 0.0, 0.0, np.nan, np.inf, -np.inf, 2.0]
    out = D.empty(4 * n)
    D.call("agpu_binary", capi.OP_POW, capi.F32, D.up(a).vp, D.up(b).vp, out.vp, n)
    got, exp = D.down(out, np.float32, n), O.binary(O.OP_POW, O.F32, a, b)
    assert np.array_equal(np.isnan(got), np.isnan(exp))  # negative / NaN base → NaN, like the reference's GPUs
    assert max_ulp(got, exp) <= G.MAX_ULP


@pytest.mark.parametrize("domain", ["wide", "near1", "near1_huge_y", "denormal_x", "big_exponent", "specials"])
def test_f32_power_hard_domains(D, domain):
    """pow needs log2 x to ~2^-33 relative: exercise the full exponent range, x -> 1 with |y| up to 1e9, denormal
    bases, results across overflow/underflow, and the IEEE special-case grid (array and scalar exponent)."""
    rng = np.random.default_rng(17)
    n = 1 << 20
    f32 = np.float32
    sgn = rng.choice([-1.0, 1.0], n)
    if domain == "wide":
        a, b = (2.0 ** rng.uniform(-126, 127, n)).astype(f32), rng.uniform(-1.2, 1.2, n).astype(f32)
    elif domain == "near1":
        a, b = (1 + rng.uniform(-1e-3, 1e-3, n)).astype(f32), (2.0 ** rng.uniform(0, 16, n) * sgn).astype(f32)
    elif domain == "near1_huge_y":
        a, b = (1 + rng.uniform(-6e-7, 6e-7, n)).astype(f32), (2.0 ** rng.uniform(10, 30, n) * sgn).astype(f32)
    elif domain == "denormal_x":
        a, b = (2.0 ** rng.uniform(-149, -120, n)).astype(f32), rng.uniform(-0.9, 0.9, n).astype(f32)
    elif domain == "big_exponent":
        a, b = (2.0 ** rng.uniform(-3, 3, n)).astype(f32), rng.uniform(-60, 60, n).astype(f32)
    else:
        v = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, np.inf, -np.inf, np.nan, 1e-45, 3.4e38, 0.99999994, 1.0000001,
                      -2.0, 1e-20, 1e20, 3.0, -3.0, 0.33333334], f32)
        a, b = [x.ravel().astype(f32) for x in np.meshgrid(v, v)]
        n = len(a)
    out = D.empty(4 * n)
    D.call("agpu_binary", capi.OP_POW, capi.F32, D.up(a).vp, D.up(b).vp, out.vp, n)
    got, exp = D.down(out, np.float32, n), O.binary(O.OP_POW, O.F32, a, b)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    assert max_ulp(got, exp) <= G.MAX_ULP, domain
    s = np.array([b[n // 2]], f32)  # scalar exponent: same arithmetic
    D.call("agpu_scalar", capi.OP_POW, capi.F32, D.up(a).vp, D.up(s).vp, out.vp, n)
    got, exp = D.down(out, np.float32, n), O.scalar(O.OP_POW, O.F32, a, s)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    assert max_ulp(got, exp) <= G.MAX_ULP, domain


def test_f32_power_exact_for_exponents_one_and_two(D):
    rng = np.random.default_rng(23)
    n = 1 << 20
    a = (2.0 ** rng.uniform(-70, 70, n)).astype(np.float32)
    out = D.empty(4 * n)
    with np.errstate(over="ignore"):
        sq = a * a
    for e, exp in ((1.0, a), (2.0, sq)):
        D.call("agpu_scalar", capi.OP_POW, capi.F32, D.up(a).vp, D.up(np.array([e], np.float32)).vp, out.vp, n)
        assert bits_equal(D.down(out, np.float32, n), exp)  # x^1 = x, x^2 = RN(x·x) (incl. overflow to inf)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_unary_exact_ops(D, dtype):
    ops = [capi.UN_NEG, capi.UN_ABS] + ([capi.UN_SQRT] if dtype == capi.F32 else [capi.UN_NOT, capi.UN_POPCOUNT])
    for n in SMALL_SIZES:
        a = rand_values(dtype, n, 9)
        da, out = D.up(a), D.empty(max(a.nbytes, 1))
        for op in ops:
            D.call("agpu_unary", op, dtype, da.vp, out.vp, n)
            assert nan_aware_bits_equal(D.down(out, NP[dtype], n), O.unary(op, dtype, a)), (op, n)


@pytest.mark.parametrize("op,name,lo,hi", [
    (capi.UN_SIN, "sin", -20.0, 15.0), (capi.UN_COS, "cos", -20.0, 15.0), (capi.UN_EXP, "exp", -20.0, 6.0),
    (capi.UN_EXP2, "exp2", -20.0, 6.0), (capi.UN_LOG, "log", -30.0, 30.0), (capi.UN_LOG2, "log2", -30.0, 30.0),
    (capi.UN_CBRT, "cbrt", -30.0, 30.0), (capi.UN_SINH, "sinh", -20.0, 6.0), (capi.UN_ACOS, "acos", -20.0, 0.0)])
def test_f32_transcendentals_within_ulp(D, op, name, lo, hi):
    """|x| log-uniform in [2^lo, 2^hi] both signs, plus ±0, ±inf, NaN, denormals; oracle = f64 libm rounded to f32."""
    rng = np.random.default_rng(11)
    n = 1 << 24 if name in ("sin", "cos", "sinh") else 1 << 22  # 2^24 points for the trig sweep
    x = (2.0 ** rng.uniform(lo, hi, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    x[:8] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-42, -1e-42, 1.0]
    x[8:16] = [1e6, -1e6, 1e9, 3.0e38, 1.17549435e-38, 89.4, -89.4, 0.99999994]  # slow-path / range-edge arguments
    out = D.empty(4 * n)
    D.call("agpu_unary", op, capi.F32, D.up(x).vp, out.vp, n)
    got, exp = D.down(out, np.float32, n), O.unary(op, O.F32, x)
    assert max_ulp(got, exp) <= G.MAX_ULP, f"{name}: {max_ulp(got, exp)} ULP"
    if name in ("sin", "sinh", "cbrt"):  # odd functions keep the sign of zero (f(−0.0) = −0.0), like libm
        assert np.signbit(got[1]) and not np.signbit(got[0]) and got[1] == 0.0


@pytest.mark.parametrize("op,name", [(capi.UN_SIN, "sin"), (capi.UN_COS, "cos"), (capi.UN_LOG, "log"), (capi.UN_SINH, "sinh"),
                                     (capi.UN_EXP, "exp"), (capi.UN_SQRT, "sqrt"), (capi.UN_CBRT, "cbrt"), (capi.UN_ACOS, "acos"),
                                     (capi.UN_EXP2, "exp2"), (capi.UN_LOG2, "log2")])
def test_f32_functions_exhaustive_over_all_bit_patterns(D, op, name):
    """ALL 2^32 f32 bit patterns (every NaN payload, denormal and huge argument included) through the kernel's own
    arithmetic against the f64 device library rounded once (agpu_selftest_unary_f32): ≤ MAX_ULP everywhere.  The CPU
    oracle (f64 libm) pins the same functions on 2^22–2^24 samples in test_f32_transcendentals_within_ulp."""
    import ctypes as C
    mx, worst = C.c_uint32(0), C.c_uint32(0)
    D.call("agpu_selftest_unary_f32", op, 0, 1 << 32, C.byref(mx), C.byref(worst))
    x = np.array([worst.value], np.uint32).view(np.float32)[0]
    assert mx.value <= G.MAX_ULP, f"{name}: {mx.value} ULP at bits {worst.value:#010x} (x = {x!r})"


def test_f32_trig_exhaustive_against_the_cpu_oracle():
    """ALL 2^32 bit patterns through agpu_unary against the CPU ORACLE (f64 libm rounded once) — not the device's own f64 library:
    tests/tools/exhaustive_vs_oracle.py in a fresh process (its oracle workers are forked before that process touches the GPU).
    sin, cos (BASELINE config 4), log and the reference's f32 → u8 cast (bit-exact) by default, ≈ 40 s; AGPU_EXHAUSTIVE=1: all ten functions,
    pow on 2^30 pairs and the six f32 → integer casts (profiles/r03_exhaustive_vs_oracle.json)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = [] if os.environ.get("AGPU_EXHAUSTIVE") == "1" else ["sin", "cos", "log", "cast_f32_u8"]  # sin / cos / log: the packed-f32 forms of round 6
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "exhaustive_vs_oracle.py")] + names, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.load(open(os.path.join(root, "gpurun_out", "r03_exhaustive_vs_oracle.json")))["functions"]
    for name in names or res:
        assert res[name]["max_ulp"] <= G.MAX_ULP and res[name]["zeros_with_the_other_sign"] == 0, (name, res[name])


@pytest.mark.parametrize("domain", [0, 1, 2])
def test_f32_power_device_selftest_2_pow_32_pairs(D, domain):
    """2^32 pseudo-random operand pairs per domain (any positive x incl. denormals / inf / NaN; x → 1 with |y| up to 2^30;
    results across overflow / underflow) against the f64 device library rounded once: ≤ MAX_ULP."""
    import ctypes as C
    mx, wx, wy = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    D.call("agpu_selftest_pow_f32", 20250418 + domain, 1 << 32, domain, C.byref(mx), C.byref(wx), C.byref(wy))
    xy = np.array([wx.value, wy.value], np.uint32).view(np.float32)
    assert mx.value <= G.MAX_ULP, f"pow domain {domain}: {mx.value} ULP at x = {xy[0]!r}, y = {xy[1]!r}"


@pytest.mark.parametrize("dtype", [capi.U8, capi.I8, capi.U16, capi.I16])
@pytest.mark.parametrize("op", [capi.UN_SIN, capi.UN_COS, capi.UN_SINH])
def test_fused_small_int_trig_exhaustive(D, dtype, op):
    info = np.iinfo(NP[dtype])
    x = np.arange(info.min, info.max + 1, dtype=np.int64).astype(NP[dtype])  # every representable input
    x = np.concatenate([x, x[: 4097 - len(x) % 4096]])
    out = D.empty(4 * len(x))
    D.call("agpu_unary", op, dtype, D.up(x).vp, out.vp, len(x))
    got, exp = D.down(out, np.float32, len(x)), O.unary(op, dtype, x)
    assert max_ulp(got, exp) <= G.MAX_ULP


CASTS = [(capi.I8, t) for t in (capi.U8, capi.U16, capi.U32, capi.I16, capi.I32, capi.F32)] + \
        [(capi.I16, t) for t in (capi.I32, capi.U16, capi.U32, capi.F32)] + \
        [(capi.U8, t) for t in (capi.U16, capi.U32, capi.I8, capi.I16, capi.I32, capi.F32)] + \
        [(capi.U16, t) for t in (capi.U32, capi.I16, capi.I32, capi.F32)] + [(capi.F32, capi.U8)] + \
        [(capi.F32, t) for t in (capi.I8, capi.I16, capi.U16, capi.I32, capi.U32)]  # reference-absent (north_star "<->")


@pytest.mark.parametrize("frm,to", CASTS)
def test_casts_exact(D, frm, to):
    for n in SMALL_SIZES + [1048576 + 5]:
        a = rand_values(frm, n, 13)
        if frm == capi.F32 and n > 8:
            a[:8] = [0, 1, -1, 5713, -5713, 255, 256, 4.3e9]
            if n > 24:  # range edges of every target, both signs, NaN / inf
                a[8:24] = [127.9, 128, -128.9, -129, 32767.5, 32768, -32768.9, -32769, 65535.9, 65536, 2147483520, 2147483648,
                           -2147483648, -2147483904, np.nan, -np.inf]
        out = D.empty(max(n * NP[to]().itemsize, 1))
        D.call("agpu_cast", frm, to, D.up(a).vp, out.vp, n)
        got = D.down(out, NP[to], n)
        if NP[frm]().itemsize == NP[to]().itemsize and frm != capi.F32:
            exp = a.view(NP[to])
        else:
            exp = O.cast(frm, to, a)
        assert bits_equal(got, exp), (frm, to, n)


def test_cast_bool_to_f32_and_unsupported(D):
    for n in SMALL_SIZES:
        bits = O.synth_bits(n, 3, 0, 0.5)
        out = D.empty(max(4 * n, 1))
        D.call("agpu_cast", capi.BOOL, capi.F32, D.up(bits).vp, out.vp, n)
        assert bits_equal(D.down(out, np.float32, n), O.cast(O.BOOL, O.F32, bits, n))
    x = D.up(np.zeros(4, np.float32))
    assert D.status("agpu_cast", capi.U32, capi.F32, x.vp, x.vp, 4) == capi.ERR_UNSUPPORTED  # not in the reference's table
    assert D.status("agpu_binary", capi.OP_DIV, capi.U8, x.vp, x.vp, x.vp, 4) == capi.ERR_UNSUPPORTED
    assert D.status("agpu_unary", capi.UN_SQRT, capi.I32, x.vp, x.vp, 4) == capi.ERR_UNSUPPORTED
    assert b"not supported" in capi.lib().agpu_last_error()


@pytest.mark.parametrize("dtype", ALL_DTYPES + [capi.BOOL])
def test_broadcast(D, dtype):
    for n in SMALL_SIZES:
        if dtype == capi.BOOL:
            for v in (0, 1):
                out = D.empty(O.bitmap_bytes(n) + 8)
                D.call("agpu_broadcast", capi.BOOL, v, out.vp, n)
                assert bits_equal(D.down(out, np.uint8, O.bitmap_bytes(n)), O.broadcast(O.BOOL, v, n))
            continue
        v = rand_values(dtype, 1, 21, special=False)
        raw = np.zeros(4, np.uint8)
        raw[: v.nbytes] = v.view(np.uint8)
        nbytes = n * v.itemsize
        out = D.empty(nbytes + 16)
        D.call("agpu_broadcast", dtype, int(raw.view(np.uint32)[0]), out.vp, n)
        assert bits_equal(D.down(out, NP[dtype], n), np.full(n, v[0], NP[dtype]))
        D.call("agpu_broadcast_from_device", dtype, D.up(v).vp, out.vp, n)
        assert bits_equal(D.down(out, NP[dtype], n), np.full(n, v[0], NP[dtype]))
        # nothing past the last element was touched
        assert (D.down(out, np.uint8, nbytes + 16)[nbytes:] == 0xCD).all()


# ------------------------------------------------------------------ compare → bitmap (+ fused validity)
@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("variant", [0, 1])
def test_compare_all_ops(D, dtype, variant):
    D.call("agpu_pipeline_set_tuning", b"cmp_variant", variant)
    try:
        for n in SIZES:
            a, b = rand_values(dtype, n, 31), rand_values(dtype, n, 32)
            if n:
                b[::3] = a[::3]
            nb = O.bitmap_bytes(n)
            da, db = D.up(a), D.up(b)
            va, vb = O.synth_bits(n, 41, 0, 0.9), O.synth_bits(n, 42, 0, 0.9)
            dva, dvb = D.up(va), D.up(vb)
            out, outv = D.empty(nb + 8), D.empty(nb + 8)
            for op in range(5):
                D.call("agpu_compare", op, dtype, da.vp, db.vp, out.vp, n)
                exp = O.compare(op, dtype, a, b)
                assert bits_equal(D.down(out, np.uint8, nb), exp), (op, n)  # padding bits are zero too
                D.call("agpu_compare_validity", op, dtype, da.vp, db.vp, dva.vp, dvb.vp, out.vp, outv.vp, n)
                assert bits_equal(D.down(out, np.uint8, nb), exp), ("fused", op, n)
                assert bits_equal(D.down(outv, np.uint8, nb), O.bitmap_binary(O.OP_AND, va, vb, n)), ("validity", op, n)
            # one side without validity → copy of the other; both absent → out_validity untouched
            D.call("agpu_compare_validity", capi.CMP_EQ, dtype, da.vp, db.vp, None, dvb.vp, out.vp, outv.vp, n)
            assert bits_equal(D.down(outv, np.uint8, nb), vb[:nb])
    finally:
        D.call("agpu_pipeline_set_tuning", b"cmp_variant", 0)


def test_compare_f32_nan_inf_table(D):
    a = np.array([-1, 3, np.nan, np.inf, -np.inf, -np.inf, np.inf, np.nan, 0.0, -0.0], dtype=np.float32)
    b = np.array([0, 2, np.nan, np.inf, -np.inf, np.inf, -np.inf, 3, -0.0, 0.0], dtype=np.float32)
    out = D.empty(16)
    for op in range(5):
        D.call("agpu_compare", op, capi.F32, D.up(a).vp, D.up(b).vp, out.vp, len(a))
        assert bits_equal(D.down(out, np.uint8, 8), O.compare(op, O.F32, a, b))


def test_compare_unaligned_inputs(D):
    n = 70001
    a, b = rand_values(capi.I16, n, 1), rand_values(capi.I16, n, 2)
    out = D.empty(O.bitmap_bytes(n))
    D.call("agpu_compare", capi.CMP_LT, capi.I16, D.up(a, 2).vp, D.up(b, 2).vp, out.vp, n)
    assert bits_equal(D.down(out, np.uint8, O.bitmap_bytes(n)), O.compare(O.CMP_LT, O.I16, a, b))


# ------------------------------------------------------------------ bitmaps
def test_bitmap_ops_popcount_any(D):
    for n in SIZES:
        a, b, m, vm = (O.synth_bits(n, s, 0, p) for s, p in ((1, 0.5), (2, 0.5), (3, 0.5), (4, 0.9)))
        nb = O.bitmap_bytes(n)
        da, db, dm, dvm = D.up(a), D.up(b), D.up(m), D.up(vm)
        out = D.empty(nb + 8)
        for op in (capi.OP_AND, capi.OP_OR, capi.OP_XOR):
            D.call("agpu_bitmap_binary", op, da.vp, db.vp, out.vp, n)
            assert bits_equal(D.down(out, np.uint8, nb), O.bitmap_binary(op, a, b, n))
        D.call("agpu_bitmap_not", da.vp, out.vp, n)
        assert bits_equal(D.down(out, np.uint8, nb), O.bitmap_not(a, n))
        D.call("agpu_merge_bits", da.vp, db.vp, dm.vp, out.vp, n)
        assert bits_equal(D.down(out, np.uint8, nb), O.merge_bits(a, b, m, n))
        for va, vb, vmm in ((a, b, vm), (None, b, vm), (a, None, None), (None, None, vm), (a, b, None)):
            D.call("agpu_bitmap_merge_validity", da.vp if va is not None else None, db.vp if vb is not None else None,
                   dm.vp, dvm.vp if vmm is not None else None, out.vp, n)
            assert bits_equal(D.down(out, np.uint8, nb), O.merge_validity(va, vb, m, vmm, n))
        cnt = D.empty(16)
        # set the padding bits on purpose: only the first n bits may count
        dirty = a.copy()
        if n % 64:
            dirty[n // 8] |= np.uint8((0xFF << (n % 8)) & 0xFF)
            dirty[n // 8 + 1: nb] = 0xFF
        D.call("agpu_bitmap_popcount", D.up(dirty).vp, n, cnt.vp)
        assert int(D.down(cnt, np.uint64, 1)[0]) == O.bitmap_popcount(a, n)
        D.call("agpu_bitmap_any", da.vp, n, cnt.vp)
        assert bool(D.down(cnt, np.uint32, 1)[0]) == O.bitmap_any(a, n)
    zero = np.zeros(O.bitmap_bytes(100000), np.uint8)
    cnt = D.empty(16)
    D.call("agpu_bitmap_any", D.up(zero).vp, 100000, cnt.vp)
    assert D.down(cnt, np.uint32, 1)[0] == 0
    assert D.status("agpu_bitmap_merge_validity", None, None, D.up(zero).vp, None, cnt.vp, 8) == capi.ERR_ARG


def test_bitmap_copy_bits_realigns_sliced_bitmaps(D):
    src = O.synth_bits(300_000, 17, 0, 0.5)
    dsrc = D.up(src)
    for off, n in ((0, 0), (0, 1), (1, 63), (7, 64), (13, 65), (63, 1000), (64, 4096), (100_001, 199_999), (5, 299_995)):
        out = D.empty(O.bitmap_bytes(n) + 8)
        D.call("agpu_bitmap_copy_bits", dsrc.vp, off, out.vp, n)
        assert bits_equal(D.down(out, np.uint8, O.bitmap_bytes(n)), O.bitmap_copy_bits(src, off, n)), (off, n)


# ------------------------------------------------------------------ reductions
@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 257, 511, 65535, 65536, 65537, 131072, 200001, 16777216 + 7, 3 * 16777216 + 70001])
def test_f32_sum_is_bit_identical_to_the_reference_tree(D, n):
    x = O.synth_f32(n, 77, 0, -1000.0, 1000.0)
    out = D.empty(16)
    D.call("agpu_reduce", capi.RED_SUM, capi.F32, D.up(x).vp, None, n, out.vp)
    got = D.down(out, np.float32, 1)
    exp = np.array([O.reduce(O.RED_SUM, O.F32, x)], np.float32)
    assert bits_equal(got, exp), (n, got, exp)


def test_f32_sum_reference_kats_and_specials(D):
    out = D.empty(16)
    for n in (65536, 4 * 1024 * 1024):  # test_f32_sum / test_f32_sum_large (crates/arithmetic/src/f32.rs:267-289)
        D.call("agpu_reduce", capi.RED_SUM, capi.F32, D.up(np.full(n, 5.0, np.float32)).vp, None, n, out.vp)
        assert D.down(out, np.float32, 1)[0] == np.float32(5.0 * n)
    for special in ([-0.0] * 256, [-0.0] * 257, [np.inf, 1.0, 2.0], [np.inf, -np.inf], [np.nan, 1.0], [3.0e38] * 4):
        x = np.array(special, np.float32)
        D.call("agpu_reduce", capi.RED_SUM, capi.F32, D.up(x).vp, None, len(x), out.vp)
        assert nan_aware_bits_equal(D.down(out, np.float32, 1), np.array([O.reduce(O.RED_SUM, O.F32, x)], np.float32)), special


@pytest.mark.parametrize("dtype", [capi.F32, capi.I32, capi.U32])
def test_reduce_min_max_sum_with_and_without_validity(D, dtype):
    out = D.empty(16)
    for n in SIZES:
        x = rand_values(dtype, n, 55)
        v = O.synth_bits(n, 56, 0, 0.8)
        dx, dv = D.up(x), D.up(v)
        ops = [capi.RED_MIN, capi.RED_MAX] + ([capi.RED_SUM] if dtype != capi.F32 else [])
        for op in ops:
            for val, dval in ((None, None), (v, dv.vp)):
                D.call("agpu_reduce", op, dtype, dx.vp, dval, n, out.vp)
                got = D.down(out, NP[dtype], 1)
                exp = np.array([O.reduce(op, dtype, x, val)], NP[dtype])
                assert nan_aware_bits_equal(got, exp), (op, n, val is not None, got, exp)
        if dtype == capi.F32:  # null-aware tree sum (extension): nulls count as +0.0
            xf = np.where(np.isnan(x), np.float32(1.0), x)  # keep the comparison finite
            D.call("agpu_reduce", capi.RED_SUM, dtype, D.up(xf).vp, dv.vp, n, out.vp)
            got = D.down(out, np.float32, 1)
            assert nan_aware_bits_equal(got, np.array([O.reduce(O.RED_SUM, O.F32, xf, v)], np.float32)), n


def test_reduce_all_nan_and_unaligned(D):
    out = D.empty(16)
    x = np.full(1000, np.nan, np.float32)
    for op in (capi.RED_MIN, capi.RED_MAX):
        D.call("agpu_reduce", op, capi.F32, D.up(x).vp, None, len(x), out.vp)
        assert np.isnan(D.down(out, np.float32, 1)[0])
    y = O.synth_f32(100003, 5, 0, -1, 1)
    for op in (capi.RED_SUM, capi.RED_MIN, capi.RED_MAX):
        D.call("agpu_reduce", op, capi.F32, D.up(y, 4).vp, None, len(y), out.vp)
        assert bits_equal(D.down(out, np.float32, 1), np.array([O.reduce(op, O.F32, y)], np.float32)), op


def test_reduce_wave_path_with_nan_chunks_and_tails(D):
    """The one-wave-per-chunk form (aligned columns of >= 1 Mi rows without validity) + its ONE finishing workgroup: a chunk of NaNs
    leaves a NaN partial that the fold must skip; an all-NaN column is NaN; -0.0 orders below +0.0 across partials; the < 1-chunk tail
    and a tail that holds the extreme; wrapping sums over the same shapes."""
    out = D.empty(16)
    for n in (1 << 20, (1 << 20) + 1, (1 << 20) + 16383, 3 * (1 << 20) + 5000, (1 << 23) + 77):
        x = rand_values(capi.F32, n, 91)
        x[np.isnan(x)] = 1.0
        x[:16384] = np.nan                      # chunk 0: all NaN
        x[16384 * 7: 16384 * 8] = np.nan        # another one
        x[16384 * 9 + 5] = np.nan               # a NaN inside an ordinary chunk
        variants = [x]
        y = x.copy(); y[-1] = np.float32(3.0e38); y[-2] = np.float32(-3.0e38)  # the extremes sit in the tail (or the last chunk)
        variants.append(y)
        z = np.full(n, np.nan, np.float32)      # nothing but NaN
        variants.append(z)
        w = np.full(n, 0.0, np.float32); w[16384 * 3 + 1] = -0.0  # -0.0 < +0.0, one of them in one chunk
        variants.append(w)
        for v in variants:
            dv = D.up(v)
            for op in (capi.RED_MIN, capi.RED_MAX):
                D.call("agpu_reduce", op, capi.F32, dv.vp, None, n, out.vp)
                got = D.down(out, np.float32, 1)
                assert nan_aware_bits_equal(got, np.array([O.reduce(op, O.F32, v)], np.float32)), (op, n, got)
        for dtype in (capi.I32, capi.U32):
            xi = rand_values(dtype, n, 92)
            di = D.up(xi)
            for op in (capi.RED_SUM, capi.RED_MIN, capi.RED_MAX):
                D.call("agpu_reduce", op, dtype, di.vp, None, n, out.vp)
                assert bits_equal(D.down(out, NP[dtype], 1), np.array([O.reduce(op, dtype, xi)], NP[dtype])), (op, dtype, n)


def test_null_aware_reductions_through_the_wave_kernel(D):
    """columns of >= 1 Mi rows WITH a validity bitmap (round 6b: the one-wave-per-chunk kernel reads the bits beside the values): chunks
    without a single valid row, chunks whose valid rows are all NaN, both together (the fold must keep "nothing seen" apart from "only NaN
    seen"), a column without any valid row, the tail under the bitmap"""
    out = D.empty(16)
    for n in (1 << 20, (1 << 20) + 16383 + 5, 3 * (1 << 20) + 777):
        rng = np.random.default_rng(n)
        x = rand_values(capi.F32, n, 93)
        x[np.isnan(x)] = 2.0
        v = np.unpackbits(O.synth_bits(n, 94, 0, 0.7), bitorder="little")[:n].astype(bool)
        v[:16384] = False                              # chunk 0: no valid row
        x[16384 * 2: 16384 * 3] = np.nan               # chunk 2: its valid rows are all NaN
        v[16384 * 4: 16384 * 5] = True; x[16384 * 4: 16384 * 5] = np.nan
        cases = [(x, v)]
        allnan = x.copy(); allnan[v] = np.nan          # every VALID row NaN (some chunks hold no valid row at all): the answer is NaN
        cases.append((allnan, v))
        cases.append((x, np.zeros(n, bool)))           # no valid row anywhere: the identities
        tailv = v.copy(); tailv[-(n % 16384 or 1):] = True
        y = x.copy(); y[-1] = np.float32(-3.0e38)      # the extreme in the tail, valid
        cases.append((y, tailv))
        for xs, vs in cases:
            bits = O.pack_bits(vs)
            dx, dv = D.up(xs), D.up(bits)
            for op in (capi.RED_MIN, capi.RED_MAX, capi.RED_SUM):
                xx = np.where(np.isnan(xs), np.float32(1.0), xs) if op == capi.RED_SUM else xs
                if op == capi.RED_SUM:
                    dx2 = D.up(xx)
                    D.call("agpu_reduce", op, capi.F32, dx2.vp, dv.vp, n, out.vp)
                else:
                    D.call("agpu_reduce", op, capi.F32, dx.vp, dv.vp, n, out.vp)
                got = D.down(out, np.float32, 1)
                assert nan_aware_bits_equal(got, np.array([O.reduce(op, O.F32, xx, bits)], np.float32)), (op, n, got)
            finite = np.where(np.isfinite(xs), xs, np.float32(0.5))   # (the f64 leg: finite rows, so that the comparison says something)
            D.call("agpu_reduce_sum_f64", D.up(finite).vp, dv.vp, n, out.vp)
            xs64 = finite.astype(np.float64)
            assert abs(float(D.down(out, np.float64, 1)[0]) - float(xs64[vs].sum())) <= 1e-9 * max(1.0, float(np.abs(xs64[vs]).sum()))
        for dtype in (capi.I32, capi.U32):
            xi = rand_values(dtype, n, 95)
            bits = O.pack_bits(v)
            di, dv = D.up(xi), D.up(bits)
            for op in (capi.RED_SUM, capi.RED_MIN, capi.RED_MAX):
                D.call("agpu_reduce", op, dtype, di.vp, dv.vp, n, out.vp)
                assert bits_equal(D.down(out, NP[dtype], 1), np.array([O.reduce(op, dtype, xi, bits)], NP[dtype])), (op, dtype, n)
            D.release()


def test_sum_f64_accumulator(D):
    out = D.empty(16)
    for n in (0, 1, 4097, 1 << 22):
        x = O.synth_f32(n, 9, 0, -1, 1)
        D.call("agpu_reduce_sum_f64", D.up(x).vp, None, n, out.vp)
        got = float(D.down(out, np.float64, 1)[0])
        exact = float(np.sum(x.astype(np.float64)))
        assert abs(got - exact) <= 1e-9 * max(1.0, float(np.sum(np.abs(x.astype(np.float64)))))


# ------------------------------------------------------------------ swizzle
@pytest.mark.parametrize("width,npd", [(4, np.uint32), (2, np.uint16), (1, np.uint8)])
def test_take_put_merge_values(D, width, npd):
    rng = np.random.default_rng(width)
    for n_values, n_idx in ((1, 100), (1000, 0), (1000, 1), (5000, 4099), (100000, 300007)):
        values = rng.integers(0, np.iinfo(npd).max, n_values).astype(npd)
        idx = rng.integers(0, n_values, n_idx).astype(np.uint32)
        out = D.empty(max(n_idx * width, 1))
        D.call("agpu_take", width, D.up(values).vp, n_values, D.up(idx).vp, out.vp, n_idx)
        assert bits_equal(D.down(out, npd, n_idx), O.take(width, values, idx))
        # put with unique destinations (duplicates are unspecified in the reference)
        n_dst = n_values + 17
        dst = rng.integers(0, np.iinfo(npd).max, n_dst).astype(npd)
        k = min(n_idx, n_dst)
        di = rng.permutation(n_dst)[:k].astype(np.uint32)
        si = idx[:k]
        ddst = D.up(dst)
        D.call("agpu_put", width, D.up(values).vp, D.up(si).vp, ddst.vp, D.up(di).vp, k)
        assert bits_equal(D.down(ddst, npd, n_dst), O.put(width, values, si, dst, di))
    for n in SIZES:
        a = rng.integers(0, np.iinfo(npd).max, n).astype(npd)
        b = rng.integers(0, np.iinfo(npd).max, n).astype(npd)
        m = O.synth_bits(n, 8, 0, 0.5)
        out = D.empty(max(n * width, 1))
        D.call("agpu_merge", width, D.up(a).vp, D.up(b).vp, D.up(m).vp, out.vp, n)
        assert bits_equal(D.down(out, npd, n), O.merge(width, a, b, m))


def test_take_put_bits_and_index_max(D):
    rng = np.random.default_rng(99)
    for n_bits, n_idx in ((1, 100), (77, 64), (10000, 65), (100000, 200001)):
        bits = O.synth_bits(n_bits, 1, 0, 0.5)
        idx = rng.integers(0, n_bits, n_idx).astype(np.uint32)
        out = D.empty(O.bitmap_bytes(n_idx) + 8)
        D.call("agpu_take_bits", D.up(bits).vp, n_bits, D.up(idx).vp, out.vp, n_idx)
        assert bits_equal(D.down(out, np.uint8, O.bitmap_bytes(n_idx)), O.take_bits(bits, n_bits, idx))
        n_dst = n_bits + 100
        dst = O.synth_bits(n_dst, 2, 0, 0.5)
        k = min(n_idx, n_dst)
        di = rng.permutation(n_dst)[:k].astype(np.uint32)
        ddst = D.up(dst)
        D.call("agpu_put_bits", D.up(bits).vp, D.up(idx[:k]).vp, ddst.vp, D.up(di).vp, k)
        assert bits_equal(D.down(ddst, np.uint8, O.bitmap_bytes(n_dst)), O.put_bits(bits, idx[:k], dst, di))
        mx = D.empty(16)
        D.call("agpu_index_max", D.up(idx).vp, n_idx, mx.vp)
        assert int(D.down(mx, np.uint32, 1)[0]) == O.index_max(idx)


def test_out_of_range_indices_are_handled_in_the_kernels(D):
    """Robust-access outcome (out-of-range read -> 0, write dropped) + one sticky AGPU_ERR_SHAPE at the next sync,
    with no pre-pass over the indices and no blocking call at take/put time."""
    rng = np.random.default_rng(5)
    n_values, n_idx = 5000, 20_003
    values = rng.integers(1, 1000, n_values).astype(np.uint32)
    idx = rng.integers(0, n_values, n_idx).astype(np.uint32)
    bad = rng.choice(n_idx, 37, replace=False)
    idx[bad] = n_values + rng.integers(0, 1 << 20, 37).astype(np.uint32)
    idx[bad[0]] = 0xFFFFFFFF
    out = D.empty(4 * n_idx)
    D.call("agpu_take", 4, D.up(values).vp, n_values, D.up(idx).vp, out.vp, n_idx)
    assert D.status("agpu_pipeline_sync") == capi.ERR_SHAPE and "out of range" in capi.last_error()
    assert D.status("agpu_pipeline_sync") == capi.OK  # reported once
    exp = np.where(idx < n_values, values[np.minimum(idx, n_values - 1)], 0).astype(np.uint32)
    assert bits_equal(D.down(out, np.uint32, n_idx), exp)
    # bits
    bits = O.synth_bits(n_values, 1, 0, 0.5)
    outb = D.empty(O.bitmap_bytes(n_idx) + 8)
    D.call("agpu_take_bits", D.up(bits).vp, n_values, D.up(idx).vp, outb.vp, n_idx)
    assert D.status("agpu_pipeline_sync") == capi.ERR_SHAPE
    ok_idx = np.where(idx < n_values, idx, 0).astype(np.uint32)
    expb = np.unpackbits(O.take_bits(bits, n_values, ok_idx), bitorder="little")[:n_idx] & (idx < n_values)
    gotb = np.unpackbits(D.down(outb, np.uint8, O.bitmap_bytes(n_idx)), bitorder="little")[:n_idx]
    assert np.array_equal(gotb, expb)
    # put: out-of-range source or destination ⇒ that element is skipped
    n_dst, k = 7000, 3000
    dst = rng.integers(1, 1000, n_dst).astype(np.uint32)
    si = rng.integers(0, n_values, k).astype(np.uint32)
    di = rng.permutation(n_dst)[:k].astype(np.uint32)
    si[::97] = n_values + 5
    di[5::101] = n_dst
    ddst = D.up(dst)
    D.call("agpu_put_bounded", 4, D.up(values).vp, n_values, D.up(si).vp, ddst.vp, n_dst, D.up(di).vp, k)
    assert D.status("agpu_pipeline_sync") == capi.ERR_SHAPE
    keep = (si < n_values) & (di < n_dst)
    exp = dst.copy()
    exp[di[keep]] = values[si[keep]]
    assert bits_equal(D.down(ddst, np.uint32, n_dst), exp)
    # in range: no flag
    D.call("agpu_put_bounded", 4, D.up(values).vp, n_values, D.up(si[keep]).vp, ddst.vp, n_dst, D.up(di[keep]).vp, int(keep.sum()))
    assert D.status("agpu_pipeline_sync") == capi.OK


# ------------------------------------------------------------------ by-name shim, synth, checksum, graphs
def test_launch_by_name_matches_typed_entry_points(D):
    n = 4099
    a, b = rand_values(capi.F32, n, 1), rand_values(capi.F32, n, 2)
    ia, ib = rand_values(capi.I32, n, 3), rand_values(capi.I32, n, 4)
    u8 = rand_values(capi.U8, n, 5)
    out = D.empty(4 * n + 64)
    lib = capi.lib()

    def by_name(key, entry, inputs, n_out):
        arr = (C.c_void_p * len(inputs))(*[p.vp for p in inputs])
        capi.check(lib.agpu_launch_by_name(D.h, key, entry, arr, len(inputs), out.vp, n_out), entry.decode())

    by_name(b"arithmetic/f32/array", b"add_f32", [D.up(a), D.up(b)], n)
    assert nan_aware_bits_equal(D.down(out, np.float32, n), O.binary(O.OP_ADD, O.F32, a, b))
    by_name(b"arithmetic/i32/scalar", b"i32_mul", [D.up(ia), D.up(ib[:1])], n)
    assert bits_equal(D.down(out, np.int32, n), O.scalar(O.OP_MUL, O.I32, ia, ib[:1]))
    by_name(b"compare/i32/cmp", b"lteq", [D.up(ia), D.up(ib)], n)
    assert bits_equal(D.down(out, np.uint8, O.bitmap_bytes(n)), O.compare(O.CMP_LTEQ, O.I32, ia, ib))
    by_name(b"cast/u8/cast_f32", b"cast_f32", [D.up(u8)], n)
    assert bits_equal(D.down(out, np.float32, n), O.cast(O.U8, O.F32, u8))
    by_name(b"logical/u32/logical", b"bitwise_xor", [D.up(ia.view(np.uint32)), D.up(ib.view(np.uint32))], n)
    assert bits_equal(D.down(out, np.uint32, n), O.binary(O.OP_XOR, O.U32, ia.view(np.uint32), ib.view(np.uint32)))
    by_name(b"arithmetic/f32/aggregate", b"sum", [D.up(a)], n)
    af = np.where(np.isfinite(a), a, np.float32(0))
    by_name(b"arithmetic/f32/aggregate", b"sum", [D.up(af)], n)
    assert bits_equal(D.down(out, np.float32, 1), np.array([O.reduce(O.RED_SUM, O.F32, af)], np.float32))
    by_name(b"array/f32/broadcast", b"broadcast", [D.up(a[:1] * 0 + np.float32(2.5))], n)
    assert (D.down(out, np.float32, n) == 2.5).all()
    arr = (C.c_void_p * 2)(D.up(a).vp, D.up(b).vp)
    assert lib.agpu_launch_by_name(D.h, b"arithmetic/f32/array", b"nope", arr, 2, out.vp, n) == capi.ERR_UNSUPPORTED
    assert lib.agpu_launch_by_name(D.h, b"arithmetic/f32/array", b"add_f32", arr, 1, out.vp, n) == capi.ERR_ARG


def test_synthetic_generators_and_checksum_match_oracle(D):
    for n, row0 in ((0, 0), (1, 5), (4099, 0), (100003, 1 << 33)):
        out = D.empty(max(4 * n, 8))
        D.call("agpu_synth_f32", out.vp, n, 20250418, row0, C.c_float(-1000.0), C.c_float(1000.0))
        assert bits_equal(D.down(out, np.float32, n), O.synth_f32(n, 20250418, row0, -1000.0, 1000.0))
        D.call("agpu_synth_i32", out.vp, n, 7, row0, 1024)
        assert bits_equal(D.down(out, np.int32, n), O.synth_i32(n, 7, row0, 1024))
        D.call("agpu_synth_u8", out.vp, n, 8, row0)
        assert bits_equal(D.down(out, np.uint8, n), O.synth_u8(n, 8, row0))
        bout = D.empty(O.bitmap_bytes(n) + 8)
        D.call("agpu_synth_bits", bout.vp, n, 9, row0, C.c_double(0.9))
        assert bits_equal(D.down(bout, np.uint8, O.bitmap_bytes(n)), O.synth_bits(n, 9, row0, 0.9))
        x = O.synth_u8(n, 10, 0)
        cs = D.empty(16)
        D.call("agpu_checksum", D.up(x).vp, x.nbytes, cs.vp)
        assert int(D.down(cs, np.uint64, 1)[0]) == O.checksum(x)


def test_graph_capture_replays_an_op_chain(D):
    """examples/simple.rs-style chain ((a + s) * s) captured once into a hipGraph and replayed."""
    n = 100000
    a = O.synth_f32(n, 1, 0, -10, 10)
    s = np.array([20.0], np.float32)
    da, ds = D.up(a), D.up(s)
    t, out = D.empty(4 * n), D.empty(4 * n)
    g = C.c_void_p()
    D.call("agpu_pipeline_begin_capture")
    D.call("agpu_scalar", capi.OP_ADD, capi.F32, da.vp, ds.vp, t.vp, n)
    D.call("agpu_scalar", capi.OP_MUL, capi.F32, t.vp, ds.vp, out.vp, n)
    D.call("agpu_pipeline_end_capture", C.byref(g))
    exp = O.scalar(O.OP_MUL, O.F32, O.scalar(O.OP_ADD, O.F32, a, s), s)
    for _ in range(3):
        capi.call("agpu_memset", D.h, out.vp, 0, 4 * n)
        capi.call("agpu_graph_launch", g, D.h)
        assert bits_equal(D.down(out, np.float32, n), exp)
    capi.call("agpu_graph_destroy", g)


def test_graph_capture_of_cast_headed_chains_rebuilds_the_table_on_replay(D):
    """agpu_fused_cast_chain inside a captured graph: the 8-bit route's 256-entry table is built by a kernel of the SAME capture from the
    scalar operands as they are at replay time (new source bytes and a new scalar between replays → new results); the 16-bit route
    evaluates per row.  Scratch grows on the warm-up call, not during capture."""

    class Step(C.Structure):
        _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]

    n = 3 * 65536 + 0
    x8, x16 = rand_values(capi.U8, n, 1), rand_values(capi.I16, n, 2)
    s = np.array([0.25], np.float32)
    d8, d16, ds = D.up(x8), D.up(x16), D.up(s)
    o8, o16 = D.empty(4 * n), D.empty(4 * n)
    steps = (Step * 2)()
    steps[0].op, steps[0].kind, steps[0].operand = capi.OP_MUL, 1, ds.vp.value
    steps[1].op, steps[1].kind, steps[1].operand = capi.UN_SQRT, 0, None
    D.call("agpu_fused_cast_chain", capi.U8, d8.vp, C.cast(steps, C.c_void_p), 2, o8.vp, n)  # warm-up: scratch for the table
    g = C.c_void_p()
    D.call("agpu_pipeline_begin_capture")
    D.call("agpu_fused_cast_chain", capi.U8, d8.vp, C.cast(steps, C.c_void_p), 2, o8.vp, n)
    D.call("agpu_fused_cast_chain", capi.I16, d16.vp, C.cast(steps, C.c_void_p), 2, o16.vp, n)
    D.call("agpu_pipeline_end_capture", C.byref(g))
    for rep in range(3):
        if rep:  # new inputs in the same device buffers
            x8, x16 = rand_values(capi.U8, n, 10 + rep), rand_values(capi.I16, n, 20 + rep)
            s = np.array([0.25 + rep], np.float32)
            for buf, arr in ((d8, x8), (d16, x16), (ds, s)):
                capi.call("agpu_upload", D.h, buf.vp, arr.ctypes.data_as(C.c_void_p), arr.nbytes)
        capi.call("agpu_memset", D.h, o8.vp, 0, 4 * n)
        capi.call("agpu_memset", D.h, o16.vp, 0, 4 * n)
        capi.call("agpu_graph_launch", g, D.h)
        assert float(s[0]) == 0.25 + rep
        for out, frm, x in ((o8, capi.U8, x8), (o16, capi.I16, x16)):
            exp = O.unary(O.UN_SQRT, O.F32, O.scalar(O.OP_MUL, O.F32, O.cast(frm, O.F32, x), s))
            assert nan_aware_bits_equal(D.down(out, np.float32, n), exp), (rep, frm)
    capi.call("agpu_graph_destroy", g)


def test_graph_capture_of_takes_and_puts_uses_the_direct_kernels(D):
    """the bucketed pipelines allocate temporaries and (under the auto policy) decide on the device: neither belongs in a captured
    graph — while a pipeline is capturing, take / put / their Boolean forms enqueue the direct kernels, whatever the tuning says"""
    n, n_src, n_dst = 100_000, 250_000, 300_000
    rng = np.random.default_rng(8)
    vals, dst = rand_values(capi.U32, n_src, 1), rand_values(capi.U32, n_dst, 2)
    bits = O.synth_bits(n_src, 3, 0, 0.5)
    idx = rng.integers(0, n_src, n).astype(np.uint32)
    di = rng.permutation(n_dst)[:n].astype(np.uint32)
    dv, db, dix, ddi = D.up(vals), D.up(bits), D.up(idx), D.up(di)
    out, outb, ddst = D.empty(4 * n), D.empty(O.bitmap_bytes(n) + 8), D.up(dst)
    for mode in (2, 4, 0):
        D.p.set_tuning("gather_bucket", mode)
        g = C.c_void_p()
        D.call("agpu_pipeline_begin_capture")
        D.call("agpu_take", 4, dv.vp, n_src, dix.vp, out.vp, n)
        D.call("agpu_take_bits", db.vp, n_src, dix.vp, outb.vp, n)
        D.call("agpu_put_bounded", 4, dv.vp, n_src, dix.vp, ddst.vp, n_dst, ddi.vp, n)
        D.call("agpu_pipeline_end_capture", C.byref(g))
        for _ in range(2):
            capi.call("agpu_memset", D.h, out.vp, 0, 4 * n)
            capi.call("agpu_graph_launch", g, D.h)
            assert bits_equal(D.down(out, np.uint32, n), O.take(4, vals, idx))
            assert bits_equal(D.down(outb, np.uint8, O.bitmap_bytes(n)), O.take_bits(bits, n_src, idx))
            assert bits_equal(D.down(ddst, np.uint32, n_dst), O.put(4, vals, idx, dst, di))
        capi.call("agpu_graph_destroy", g)
    D.p.set_tuning("gather_bucket", 0)


def test_two_pipelines_from_two_threads(D):
    import threading

    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline

    errs = []

    def work(seed):
        try:
            p = ArrowComputePipeline(D.dev, f"t{seed}")
            n = 200003
            a, b = O.synth_f32(n, seed, 0, -1, 1), O.synth_f32(n, seed + 100, 0, -1, 1)
            da, db = D.dev.create_gpu_buffer_with_data(a), D.dev.create_gpu_buffer_with_data(b)
            out = D.dev.create_empty_buffer(4 * n)
            for _ in range(20):
                capi.call("agpu_binary", p._handle, capi.OP_ADD, capi.F32, C.c_void_p(da.ptr), C.c_void_p(db.ptr),
                          C.c_void_p(out.ptr), n)
            got = np.empty(n, np.float32)
            capi.call("agpu_download", p._handle, C.c_void_p(got.ctypes.data), C.c_void_p(out.ptr), 4 * n)
            assert bits_equal(got, a + b)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


@pytest.mark.parametrize("frm,to", [(capi.U8, capi.F32), (capi.I8, capi.I32), (capi.U8, capi.U16), (capi.I16, capi.F32), (capi.U16, capi.U32),
                                    (capi.F32, capi.U8), (capi.F32, capi.I8), (capi.F32, capi.I16), (capi.F32, capi.U16)])
def test_width_changing_casts_on_slices_and_chunk_boundaries(D, frm, to):
    """The wave-transposed cast kernels (cvt_wide_kernel / cvt_narrow_kernel: 16-byte accesses on BOTH sides) take only
    16-byte-aligned columns and whole 512/1024-row chunks; slices that start at any element and sizes around the chunk
    boundaries must go through the tail / element-granular paths with the same bits."""
    wi, wo = NP[frm]().itemsize, NP[to]().itemsize
    base = rand_values(frm, 70_000, 5)
    if frm == capi.F32:
        with np.errstate(invalid="ignore"):
            base = (base % np.float32(70000.0)).astype(np.float32)  # both signs, beyond every target's range; NaN / inf stay NaN
    src = D.up(base)
    for first in (0, 1, 3, 4, 16 // wi, 17):
        for n in (0, 1, 511, 512, 513, 1023, 1024, 1025, 2047, 2048 + 7, 65_536 - first):
            for out_shift in (0, wo):  # output pointer 16-byte aligned or shifted by one element
                out = D.empty(n * wo + 64)
                D.call("agpu_cast", frm, to, C.c_void_p(src.buf.ptr + first * wi), C.c_void_p(out.buf.ptr + out_shift), n)
                got = D.down(out, np.uint8, n * wo + 64)[out_shift:out_shift + n * wo].view(NP[to]) if n else np.empty(0, NP[to])
                assert bits_equal(got, O.cast(frm, to, base[first:first + n])), (first, n, out_shift)


@pytest.mark.parametrize("dtype", [capi.U8, capi.I8, capi.U16, capi.I16])
@pytest.mark.parametrize("op", [capi.UN_SIN, capi.UN_COS])
def test_fused_small_int_trig_on_slices(D, dtype, op):
    """lut8_kernel / the 16-bit cast-then-function kernel (cvt_wide_kernel) in the wave-transposed form: 16-byte loads, so slices that start at any element and
    ragged sizes take the tail kernels — identical bits."""
    w = NP[dtype]().itemsize
    base = rand_values(dtype, 40_000, 9)
    src = D.up(base)
    for first in (0, 1, 2, 8 // w, 16 // w, 5):
        for n in (1, 4095, 4096, 4097, 8192 + 3, 30_000):
            out = D.empty(4 * n + 16)
            D.call("agpu_unary", op, dtype, C.c_void_p(src.buf.ptr + first * w), out.vp, n)
            assert max_ulp(D.down(out, np.float32, n), O.unary(op, dtype, base[first:first + n])) <= G.MAX_ULP, (first, n)


def test_f32_min_max_reduction_orders_signed_zeros_and_ignores_nan(D):
    """RedMinMaxF32 uses v_min_f32 / v_max_f32 directly: -0 < +0 in both directions and positions, NaN skipped unless all
    NaN, +-inf ordered — bit-exact against the oracle"""
    cases = [[0.0, -0.0], [-0.0, 0.0], [0.0] * 300 + [-0.0], [-0.0] * 300 + [0.0], [np.nan, -0.0, 0.0, np.nan], [np.nan] * 5,
             [np.inf, -np.inf, np.nan, 0.0], [1e-45, -1e-45, 0.0, -0.0], [-0.0] * 70_000 + [0.0] + [-0.0] * 70_000]
    for vals in cases:
        x = np.array(vals, np.float32)
        for op in (capi.RED_MIN, capi.RED_MAX):
            out = D.empty(16)
            D.call("agpu_reduce", op, capi.F32, D.up(x).vp, None, len(x), out.vp)
            got = D.down(out, np.float32, 1)
            exp = np.float32(O.reduce(op, O.F32, x))
            assert got.view(np.uint32)[0] == exp.view(np.uint32) or (np.isnan(got[0]) and np.isnan(exp)), (vals[:4], op)


def test_config_1_literal_i32_add_1Mi_rows(ag):
    """BASELINE.json configs[0] as SURVEY §8(d) writes it: Int32ArrayGPU of 1 048 576 rows, a[i] = i, b[i] = 1_000_000 − i
    (mod 2^32), no nulls, through `add` AND `add_dyn`, bit-exact against numpy's wrapping int32 add
    [ref: crates/arithmetic/src/i32.rs:103-110 impl_arithmetic_array_op!(Int32ArrayGPU, add_i32);
    arithmetic_kernels.rs:77-120 add_dyn].  (The reference runs it on wgpu's fallback adapter; here there is ONE backend,
    the HIP path — no CPU path exists in the product.)"""
    dev = ag.GPU_DEVICE()
    n = 1_048_576
    i = np.arange(n, dtype=np.int64)
    a_np = i.astype(np.int32)
    b_np = ((1_000_000 - i) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    want = (a_np.astype(np.int64) + b_np.astype(np.int64)).astype(np.uint64).astype(np.uint32).view(np.int32)  # wrap-add
    assert (want == 1_000_000).all()
    a, b = ag.Int32ArrayGPU.from_slice(a_np, dev), ag.Int32ArrayGPU.from_slice(b_np, dev)
    got = a.add(b)
    assert isinstance(got, ag.Int32ArrayGPU) and len(got) == n and got.null_buffer is None
    assert np.array_equal(got.raw_values(), want)
    dyn = ag.add_dyn(a, b)
    assert np.array_equal(ag.Int32ArrayGPU.try_from(dyn).raw_values(), want)
    assert np.array_equal(got.raw_values(), O.binary(O.OP_ADD, O.I32, a_np, b_np))  # and the oracle agrees with numpy
    # the same columns with a wrap in them: i + (2^31 − 1 − i + 5) overflows for every row
    c_np = ((0x7FFFFFFF - i + 5) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    want2 = (a_np.view(np.uint32) + c_np.view(np.uint32)).view(np.int32)
    assert (want2 < 0).all()
    assert np.array_equal(a.add(ag.Int32ArrayGPU.from_slice(c_np, dev)).raw_values(), want2)
