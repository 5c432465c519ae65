"""GPU: differential fuzzing of the `*_dyn` surface.  Random programs (random array types with nulls and special
values, random `*_dyn` calls whose inputs are earlier results) run through BOTH namespaces that expose the reference's
API — `oracle.model` (CPU restatement of the host rules + the C oracle) and `arrow_gpu_amd` (HIP) — and must agree on:
which (op, type, type) combinations are rejected (the reference's dyn tables), result type, length, validity, and
values (bit-exact; single-step transcendental results within 1 ULP and not fed forward).
Runs with eager pipelines and with fusing pipelines (AGPU_FUSE behaviour) — the results must not depend on it."""
import math
import os

import numpy as np
import pytest

import golden_runner as G
import oracle.model as M

pytestmark = pytest.mark.gpu

PRIMS = {"Float32ArrayGPU": np.float32, "Int32ArrayGPU": np.int32, "UInt32ArrayGPU": np.uint32, "Date32ArrayGPU": np.int32,
         "Int16ArrayGPU": np.int16, "UInt16ArrayGPU": np.uint16, "Int8ArrayGPU": np.int8, "UInt8ArrayGPU": np.uint8}
EXACT_BINARY = ["add_dyn", "sub_dyn", "mul_dyn", "div_dyn", "max_dyn", "min_dyn", "gt_dyn", "gteq_dyn", "lt_dyn", "lteq_dyn",
                "eq_dyn", "bitwise_and_dyn", "bitwise_or_dyn", "bitwise_xor_dyn", "bitwise_shl_dyn", "bitwise_shr_dyn",
                "add_scalar_dyn", "sub_scalar_dyn", "mul_scalar_dyn", "div_scalar_dyn", "rem_scalar_dyn"]
EXACT_UNARY = ["neg_dyn", "abs_dyn", "sqrt_dyn", "bitwise_not_dyn"]
ULP_UNARY = ["sin_dyn", "cos_dyn", "exp_dyn", "exp2_dyn", "log_dyn", "log2_dyn", "cbrt_dyn", "sinh_dyn", "acos_dyn"]
ULP_BINARY = ["power_dyn"]
F32_SPECIALS = [0.0, -0.0, 1.0, -1.0, float("inf"), float("-inf"), float("nan"), 1e-40, 3.4e38, 0.5, 255.0, 256.0]


def random_values(rng, name, n):
    if name == "BooleanArrayGPU":
        vals = [bool(x) for x in rng.integers(0, 2, n)]
    elif name == "Float32ArrayGPU":
        x = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6, n)).astype(np.float32)
        for i in range(n):
            if rng.random() < 0.15:
                x[i] = np.float32(F32_SPECIALS[rng.integers(len(F32_SPECIALS))])
        vals = [float(v) for v in x]
    else:
        info = np.iinfo(PRIMS[name])
        x = rng.integers(info.min, int(info.max) + 1, n, dtype=np.int64)
        small = rng.random(n) < 0.5
        x = np.where(small, rng.integers(-40, 41, n) if info.min < 0 else rng.integers(0, 41, n), x)
        vals = [int(v) for v in np.clip(x, info.min, info.max)]
    if rng.random() < 0.5:
        return vals
    return [None if rng.random() < 0.25 else v for v in vals]


def build(ns, name, vals, dev):
    cls = getattr(ns, name)
    if any(v is None for v in vals):
        return cls.from_optional_slice(vals, dev)
    return cls.from_slice(vals, dev)


def same_float(g, e, ulp):
    if math.isnan(e) or math.isnan(g):
        return math.isnan(e) and math.isnan(g)
    if ulp == 0:
        return np.float32(g).view(np.uint32) == np.float32(e).view(np.uint32)
    return G.ulp_diff(g, e) <= ulp


def compare(got, exp, ulp, what):
    assert type(got).__name__ == type(exp).__name__, what
    assert got.len == exp.len, what
    gv, ev = got.values(), exp.values()
    assert [v is None for v in gv] == [v is None for v in ev], f"{what}: validity differs"
    for i, (g, e) in enumerate(zip(gv, ev)):
        if e is None:
            continue
        if isinstance(e, float) or isinstance(g, float):
            assert same_float(float(g), float(e), ulp), f"{what}[{i}]: {g!r} != {e!r}"
        else:
            assert int(g) == int(e), f"{what}[{i}]: {g!r} != {e!r}"


@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("seed", range(int(os.environ.get("AGPU_FUZZ_BASE", "0")), int(os.environ.get("AGPU_FUZZ_BASE", "0")) + int(os.environ.get("AGPU_FUZZ_SEEDS", "30"))))  # soaks: AGPU_FUZZ_SEEDS = how many, AGPU_FUZZ_BASE = the first
def test_random_dyn_programs_agree_with_the_model(ag, seed, fuse, monkeypatch):
    import arrow_gpu_amd.gpu_utils as gu

    monkeypatch.setattr(gu, "_FUSE_DEFAULT", fuse)
    rng = np.random.default_rng(1000 + seed)
    dev = ag.GPU_DEVICE()
    n = int(rng.choice([1, 5, 64, 257, 1000, 4099]))
    names = list(PRIMS) + ["BooleanArrayGPU"]
    pool = []  # (product array, model array)
    for _ in range(8):
        name = names[rng.integers(len(names))]
        length = 1 if rng.random() < 0.2 else n
        vals = random_values(rng, name, length)
        pool.append((build(ag, name, vals, dev), build(M, name, vals, None)))
    accepted = rejected = 0
    type_names = ["BooleanType", "Float32Type", "UInt32Type", "UInt16Type", "UInt8Type", "Int32Type", "Int16Type", "Int8Type",
                  "Date32Type"]
    for step in range(120):
        r = rng.random()
        if r < 0.18:  # the other dyn tables: cast / bitcast (type → type), take (gather), merge (select by mask)
            kind = ("cast_dyn", "bitcast_dyn", "take_dyn", "merge_dyn")[rng.integers(4)]
            src = pool[rng.integers(len(pool))]
            if kind in ("cast_dyn", "bitcast_dyn"):
                tn = type_names[rng.integers(len(type_names))]
                margs, gargs = (src[1], getattr(M.ArrowType, tn)), (src[0], getattr(ag.ArrowType, tn))
            elif kind == "take_dyn":
                idx = rng.integers(0, src[0].len, int(rng.integers(1, 300))).astype(np.uint32)
                margs, gargs = (src[1], M.UInt32ArrayGPU.from_slice(idx)), (src[0], ag.UInt32ArrayGPU.from_slice(idx, dev))
            else:
                same = [q for q in pool if type(q[0]) is type(src[0]) and q[0].len == src[0].len]
                other = same[rng.integers(len(same))]
                mvals = random_values(rng, "BooleanArrayGPU", src[0].len)
                margs = (src[1], other[1], build(M, "BooleanArrayGPU", mvals, None))
                gargs = (src[0], other[0], build(ag, "BooleanArrayGPU", mvals, dev))
            what = f"seed {seed} step {step}: {kind}({type(src[0]).__name__}, …)"
            try:
                exp = getattr(M, kind)(*margs)
            except M.OracleUnsupported:
                with pytest.raises(ag.ArrowErrorGPU):  # OperationNotSupported / CastingNotSupported
                    getattr(ag, kind)(*gargs)
                rejected += 1
                continue
            got = getattr(ag, kind)(*gargs)
            compare(got, exp, 0, what)
            accepted += 1
            if len(pool) < 40 and got.len == n:
                pool.append((got, exp))
            continue
        if r < 0.6:
            fn, ulp, arity = EXACT_BINARY[rng.integers(len(EXACT_BINARY))], 0, 2
        elif r < 0.8:
            fn, ulp, arity = EXACT_UNARY[rng.integers(len(EXACT_UNARY))], 0, 1
        elif r < 0.95:
            fn, ulp, arity = ULP_UNARY[rng.integers(len(ULP_UNARY))], 1, 1
        else:
            fn, ulp, arity = ULP_BINARY[0], 1, 2
        args = [pool[rng.integers(len(pool))] for _ in range(arity)]
        if arity == 2 and rng.random() < 0.75:  # mostly same-typed operands, or most draws die in the dyn tables
            same = [q for q in pool if type(q[0]) is type(args[0][0])]
            args[1] = same[rng.integers(len(same))]
        if arity == 2 and args[0][0].len != args[1][0].len and 1 not in (args[0][0].len, args[1][0].len):
            continue
        what = f"seed {seed} step {step}: {fn}({', '.join(type(a[0]).__name__ for a in args)})"
        try:
            exp = getattr(M, fn)(*[a[1] for a in args])
        except M.OracleUnsupported:
            with pytest.raises(ag.OperationNotSupported):
                getattr(ag, fn)(*[a[0] for a in args])
            rejected += 1
            continue
        except AssertionError:
            continue  # shape the model refuses (len-1 vs len-n array op): not a dispatch question
        got = getattr(ag, fn)(*[a[0] for a in args])
        compare(got, exp, ulp, what)
        accepted += 1
        if ulp == 0 and len(pool) < 40:
            pool.append((got, exp))
    print(f"seed {seed} fuse {fuse}: n={n} accepted {accepted} rejected {rejected}")
    assert accepted + rejected >= 20  # (how many of each depends on the types the seed drew)
