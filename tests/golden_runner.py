"""Runs the reference's own unit-test vectors (tests/golden/*.json) against a namespace that exposes the
reference's API — `oracle.model` on CPU, `arrow_gpu_amd` on the GPU.  Each runner mirrors one Rust test macro
(crates/test_macros/src/lib.rs, crates/cast/src/lib.rs:196-247, crates/routines/src/{take,put,merge}.rs tests):
build arrays with from_slice / from_optional_slice, run the typed method AND the `_dyn` function, read back with
raw_values() / values(), compare.

Comparison rules: integers / booleans / casts / bitmaps exact; floats bit-exact unless the fixture carries "tol"
(the reference's float_eq_in_error 1e-2 rule, test_macros/src/lib.rs:88-109), in which case BOTH the reference's
rule and this repo's own bound (≤ MAX_ULP units in the last place of the f64-rounded expectation) are asserted.
"""
from __future__ import annotations

import json
import math
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")
MAX_ULP = 1  # north_star: f32 math/trig within 1 ULP (kernels evaluate in f64 and round once; oracle = f64 libm rounded)


def load(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)


def dec(v):
    if isinstance(v, list):
        return [dec(x) for x in v]
    if v == "nan":
        return float("nan")
    if v == "inf":
        return float("inf")
    if v == "-inf":
        return float("-inf")
    return v


def float_eq_in_error(left: float, right: float) -> bool:
    """The reference's tolerance rule, verbatim semantics (test_macros/src/lib.rs:88-109)."""
    if math.isnan(left) != math.isnan(right):
        return False
    if math.isnan(left):
        return True
    for inf in (float("-inf"), float("inf")):
        if (left == inf) != (right == inf):
            return False
    if math.isinf(left):
        return True
    return abs(abs(left) - abs(right)) <= 0.01


def ulp_diff(a: float, b: float) -> int:
    x = np.array([a, b], dtype=np.float32).view(np.int32).astype(np.int64)
    x = np.where(x < 0, np.int64(-(2**31)) - x, x)  # monotone map of the float order
    return int(abs(x[0] - x[1]))


def assert_values(got, expected, tol, what):
    got = list(got)
    assert len(got) == len(expected), f"{what}: length {len(got)} != {len(expected)}"
    for i, (g, e) in enumerate(zip(got, expected)):
        if e is None or g is None:
            assert g is None and e is None, f"{what}[{i}]: {g!r} != {e!r}"
            continue
        if isinstance(e, bool) or isinstance(g, (bool, np.bool_)):
            assert bool(g) == bool(e), f"{what}[{i}]: {g!r} != {e!r}"
        elif isinstance(e, float):
            g = float(g)
            if tol:
                assert float_eq_in_error(e, g), f"{what}[{i}]: {g!r} vs {e!r} outside the reference tolerance"
                if not (math.isnan(e) or math.isinf(e)):
                    assert ulp_diff(g, e) <= MAX_ULP, f"{what}[{i}]: {g!r} vs {e!r} = {ulp_diff(g, e)} ULP"
            else:
                if math.isnan(e):
                    assert math.isnan(g), f"{what}[{i}]: {g!r} != nan"
                else:
                    same = np.float32(g).view(np.uint32) == np.float32(e).view(np.uint32)
                    # +0.0 == -0.0 under the reference's assert_eq!
                    assert same or (g == 0.0 and e == 0.0), f"{what}[{i}]: {g!r} != {e!r}"
        else:
            assert int(g) == int(e), f"{what}[{i}]: {g!r} != {e!r}"


def run_vector(ns, r):
    """Run one record of reference_vectors.json against namespace `ns`."""
    dev = ns.GPU_DEVICE()
    kind = r["kind"]
    tol = r.get("tol")
    T = lambda name: getattr(ns, name)  # noqa: E731

    if kind == "array_op":
        t1, t2, to = (T(x) for x in r["types"])
        a = t1.from_optional_slice(dec(r["a"]), dev)
        b = t2.from_optional_slice(dec(r["b"]), dev)
        out = getattr(a, r["op"])(b)
        assert_values(out.values(), dec(r["expected"]), tol, r["name"])
        if r.get("op_dyn"):
            out = to.try_from(getattr(ns, r["op_dyn"])(a.into(), b.into()))
            assert_values(out.values(), dec(r["expected"]), tol, r["name"] + " dyn")
    elif kind == "scalar_op":
        ti, ts, to = (T(x) for x in r["types"])
        a = ti.from_slice(dec(r["a"]), dev)
        s = ts.from_slice([dec(r["scalar"])], dev)
        out = getattr(a, r["op"])(s)
        assert_values(out.raw_values(), dec(r["expected"]), tol, r["name"])
        out = to.try_from(getattr(ns, r["op_dyn"])(a.into(), s.into()))
        assert_values(out.raw_values(), dec(r["expected"]), tol, r["name"] + " dyn")
    elif kind == "unary_op":
        ti, to = (T(x) for x in r["types"])
        a = ti.from_slice(dec(r["a"]), dev)
        out = getattr(a, r["op"])()
        assert_values(out.raw_values(), dec(r["expected"]), tol, r["name"])
        if r.get("op_dyn"):
            out = to.try_from(getattr(ns, r["op_dyn"])(a.into()))
            assert_values(out.raw_values(), dec(r["expected"]), tol, r["name"] + " dyn")
    elif kind == "cast":
        ti, to = (T(x) for x in r["types"])
        a = ti.from_slice(dec(r["a"]), dev)
        out = a.cast(to)
        assert_values(out.raw_values(), dec(r["expected"]), None, r["name"])
        out = to.try_from(ns.cast_dyn(a.into(), getattr(ns.ArrowType, r["cast_type"])))
        assert_values(out.raw_values(), dec(r["expected"]), None, r["name"] + " dyn")
    elif kind == "bitcast":
        ti, to = (T(x) for x in r["types"])
        a = ti.from_slice(dec(r["a"]), dev)
        for out in (a.bitcast(to), to.try_from(ns.bitcast_dyn(a.into(), getattr(ns.ArrowType, r["cast_type"])))):
            bits = np.asarray(out.raw_values(), dtype=np.float32).view(np.uint32)
            assert [int(x) for x in bits] == r["expected_bits"], r["name"]
    elif kind == "sum":
        ty = T(r["types"][0])
        arr = ty.broadcast(r["base"], r["size"], dev)
        got = arr.sum().raw_values()
        assert len(got) == 1
        assert_values(got, [dec(r["expected"])], None, r["name"])
    elif kind == "broadcast":
        ty = T(r["types"][0])
        arr = ty.broadcast(r["value"], r["size"], dev)
        assert_values(arr.raw_values(), [dec(r["value"])] * r["size"], None, r["name"])
    elif kind == "take":
        t1, t2, to = (T(x) for x in r["types"])
        idx = t2.from_slice(r["indexes"], dev)
        if r.get("op_dyn"):
            a = t1.from_optional_slice(dec(r["values"]), dev)
            assert_values(a.take(idx).values(), dec(r["expected"]), None, r["name"])
            out = to.try_from(ns.take_dyn(a.into(), idx))
            assert_values(out.values(), dec(r["expected"]), None, r["name"] + " dyn")
        else:
            a = t1.from_slice(dec(r["values"]), dev)
            assert_values(a.take(idx).raw_values(), dec(r["expected"]), None, r["name"])
    elif kind == "put":
        ty = T(r["types"][0])
        src = ty.from_slice(dec(r["src"]), dev)
        si = ns.UInt32ArrayGPU.from_slice(r["src_indexes"], dev)
        di = ns.UInt32ArrayGPU.from_slice(r["dst_indexes"], dev)
        dst = ty.from_slice(dec(r["dst"]), dev)
        src.put(si, dst, di)
        assert_values(dst.raw_values(), dec(r["expected"]), None, r["name"])
        if r.get("op_dyn"):
            dst2 = ty.from_slice(dec(r["dst"]), dev)
            ns.put_dyn(src.into(), si, dst2, di)
            assert_values(dst2.raw_values(), dec(r["expected"]), None, r["name"] + " dyn")
    elif kind == "merge":
        t1, t2, to = (T(x) for x in r["types"])
        a = t1.from_optional_slice(dec(r["a"]), dev)
        b = t2.from_optional_slice(dec(r["b"]), dev)
        mask = ns.BooleanArrayGPU.from_optional_slice(r["mask"], dev)
        assert_values(a.merge(b, mask).values(), dec(r["expected"]), None, r["name"])
        if r.get("op_dyn"):
            out = to.try_from(ns.merge_dyn(a.into(), b.into(), mask))
            assert_values(out.values(), dec(r["expected"]), None, r["name"] + " dyn")
    else:
        raise AssertionError(f"unknown fixture kind {kind}")


def expand_runs(runs):
    out = []
    for v, n in runs:
        out.extend([v] * n)
    return out


def run_handwritten(ns, r):
    """Hand-written reference tests (tests/golden/reference_handwritten.json)."""
    dev = ns.GPU_DEVICE()
    kind = r["kind"]
    if kind == "large_add_scalar":
        n = r["size"]
        a = ns.Float32ArrayGPU.from_slice(np.arange(n, dtype=np.float32), dev)
        s = ns.Float32ArrayGPU.from_slice([r["scalar"]], dev)
        got = np.asarray(a.add_scalar(s).raw_values(), dtype=np.float32)
        exp = np.arange(n, dtype=np.float32) + np.float32(r["scalar"])
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    elif kind == "from_optional_and_merge":
        a = ns.Float32ArrayGPU.from_optional_slice(r["a"], dev)
        b = ns.Float32ArrayGPU.from_optional_slice(r["b"], dev)
        assert_values(a.raw_values(), r["a_raw"], None, "a raw")
        assert_values(b.raw_values(), r["b_raw"], None, "b raw")
        return a, b  # validity bytes are checked by the caller (namespace-specific accessors)
    elif kind == "boolean_values":
        vals = r["pattern"] * r["repeat"]
        arr = ns.BooleanArrayGPU.from_optional_slice(vals, dev)
        assert_values(arr.raw_values(), r["raw_pattern"] * r["repeat"], None, "raw")
        assert_values(arr.values(), vals, None, "values")
    elif kind in ("any", "all"):
        for case in r["cases"]:
            arr = ns.BooleanArrayGPU.from_slice(np.array(expand_runs(case["runs"]), dtype=bool), dev)
            assert getattr(arr, kind)() == case["expected"], (r["name"], case["runs"])
    elif kind == "example_basic_add":
        vals = [float(i) for i in range(r["size"])]
        a = ns.Float32ArrayGPU.from_slice(vals, dev)
        s = ns.Float32ArrayGPU.from_slice([r["scalar"]], dev)
        assert_values(a.add_scalar(s).values(), [v + r["scalar"] for v in vals], None, "add_scalar")
        out = ns.add_scalar_dyn(a.into(), s.into())
        assert type(out).__name__ == "Float32ArrayGPU"
        assert_values(out.values(), [v + r["scalar"] for v in vals], None, "add_scalar_dyn")
    elif kind == "example_pipeline_ops":
        vals = [float(i) for i in range(r["size"])]
        pipeline = ns.ArrowComputePipeline(dev, "example")
        a = ns.Float32ArrayGPU.from_slice(vals, dev)
        s = ns.Float32ArrayGPU.from_slice([r["scalar"]], dev)
        r1 = ns.add_scalar_op_dyn(a.into(), s.into(), pipeline)
        r2 = ns.mul_scalar_op_dyn(r1, s.into(), pipeline)
        pipeline.finish()
        assert type(r2).__name__ == "Float32ArrayGPU"
        assert_values(r2.values(), [(v + r["scalar"]) * r["scalar"] for v in vals], None, "pipeline ops")
    else:
        raise AssertionError(f"unknown handwritten kind {kind}")
