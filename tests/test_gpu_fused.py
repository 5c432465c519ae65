"""GPU: fused element-wise chains (agpu_fused_chain / FusedChain) are bit-identical to running the same ops one kernel
at a time and obey the same validity rules (that they also move fewer bytes is timed in tests/test_zz_gpu_perf.py)."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).tobytes()


def test_simple_rs_chain_fused_equals_unfused(ag):
    """examples/simple.rs:45-72: ((0..100) + 20) * 20 — the reference's 2-dispatch chain as ONE kernel."""
    dev = ag.GPU_DEVICE()
    vals = [float(i) for i in range(100)]
    a = ag.Float32ArrayGPU.from_slice(vals, dev)
    s = ag.Float32ArrayGPU.from_slice([20.0], dev)
    fused = ag.FusedChain(a).add_scalar(s).mul_scalar(s).finish()
    assert fused.values() == [(v + 20.0) * 20.0 for v in vals]
    assert fused.values() == a.add_scalar(s).mul_scalar(s).values()


@pytest.mark.parametrize("n", [0, 1, 5, 255, 256, 257, 4099, 1_000_003])
def test_f32_chain_bit_identical_and_validity(ag, n):
    dev = ag.GPU_DEVICE()
    rng = np.random.default_rng(n)
    mk = lambda seed, nulls: ag.Float32ArrayGPU.from_optional_slice(  # noqa: E731
        [None if (nulls and rng.random() < 0.2) else float(v) for v in O.synth_f32(n, seed, 0, -50, 50)], dev) if n < 5000 \
        else ag.Float32ArrayGPU.from_slice(O.synth_f32(n, seed, 0, -50, 50), dev)
    a, b, c = mk(1, True), mk(2, True), mk(3, False)
    s = ag.Float32ArrayGPU.from_slice([3.5], dev)
    fused = ag.FusedChain(a).mul(b).add(c).abs().sqrt().add_scalar(s).neg().sin().finish()
    unfused = a.mul(b).add(c).abs().sqrt().add_scalar(s).neg().sin()
    assert bits(fused.raw_values()) == bits(unfused.raw_values())
    assert fused.values() == unfused.values() or (n >= 5000)
    if n < 5000:
        assert (fused.null_buffer is None) == (unfused.null_buffer is None)


def test_int_chain_and_errors(ag):
    dev = ag.GPU_DEVICE()
    n = 70_001
    a = ag.Int32ArrayGPU.from_slice(O.synth_i32(n, 1, 0, 0), dev)
    b = ag.Int32ArrayGPU.from_slice(O.synth_i32(n, 2, 0, 0), dev)
    s = ag.Int32ArrayGPU.from_slice([7], dev)
    fused = ag.FusedChain(a).add(b).mul_scalar(s).bitwise_xor(b).abs().rem_scalar(s).finish()
    exp = O.scalar(O.OP_REM, O.I32, O.unary(O.UN_ABS, O.I32, O.binary(O.OP_XOR, O.I32, O.scalar(
        O.OP_MUL, O.I32, O.binary(O.OP_ADD, O.I32, a.raw_values(), b.raw_values()), [7]), b.raw_values())), [7])
    assert bits(fused.raw_values()) == bits(exp)
    with pytest.raises(ag.OperationNotSupported):
        ag.FusedChain(a).sqrt()
    with pytest.raises(ag.OperationNotSupported):
        ag.FusedChain(ag.BooleanArrayGPU.from_slice([True], dev))
    with pytest.raises(ag.OperationNotSupported):   # behind a cast head the chain is an f32 chain: no bitwise steps, f32 operands only
        ag.FusedChain(ag.UInt8ArrayGPU.from_slice([1], dev)).bitwise_not()
    with pytest.raises(ag.OperationNotSupported):
        ag.FusedChain(ag.UInt8ArrayGPU.from_slice([1], dev)).add(ag.UInt8ArrayGPU.from_slice([1], dev))
    with pytest.raises(ag.OperationNotSupported):
        ag.FusedChain(a).add(ag.Float32ArrayGPU.from_slice([1.0] * n, dev))
    ch = ag.FusedChain(a)
    for _ in range(8):
        ch.abs()
    with pytest.raises(ag.ArrowErrorGPU):
        ch.abs()


# ---------------------------------------------------------------- fusing pipelines: ArrowComputePipeline(fuse=True)
def _rand(ag, dev, n, seed, nulls=False):
    vals = O.synth_f32(n, seed, 0, -50, 50)
    if not nulls:
        return ag.Float32ArrayGPU.from_slice(vals, dev)
    rng = np.random.default_rng(seed)
    return ag.Float32ArrayGPU.from_optional_slice([None if rng.random() < 0.2 else float(v) for v in vals], dev)


def test_fusing_pipeline_runs_simple_rs_chain_as_one_kernel(ag):
    """examples/simple.rs:45-72 written in call-chain style: the intermediate dies at once, finish() issues ONE kernel."""
    dev = ag.GPU_DEVICE()
    a = ag.Float32ArrayGPU.from_slice([float(i) for i in range(100)], dev)
    s = ag.Float32ArrayGPU.from_slice([20.0], dev)
    p = ag.ArrowComputePipeline(dev, "example", fuse=True)
    r = ag.mul_scalar_op_dyn(ag.add_scalar_op_dyn(a, s, p), s, p)
    assert p.stats["recorded"] == 2 and p.stats["kernels"] == 0  # nothing has run yet, like a wgpu encoder
    p.finish()
    assert p.stats == {"recorded": 2, "kernels": 1, "fused_chains": 1, "fused_ops": 2}
    assert r.values() == [(float(i) + 20.0) * 20.0 for i in range(100)]


@pytest.mark.parametrize("n", [1, 257, 70_001])
def test_fusing_pipeline_matches_eager_pipeline(ag, n):
    dev = ag.GPU_DEVICE()
    a, b, c = _rand(ag, dev, n, 1, nulls=n < 5000), _rand(ag, dev, n, 2, nulls=n < 5000), _rand(ag, dev, n, 3)
    s = ag.Float32ArrayGPU.from_slice([3.5], dev)

    def program(p):
        x = a.mul_op(b, p).add_op(c, p).abs_op(p).sqrt_op(p)           # 4 fusable ops, intermediates dropped
        kept = x.add_scalar_op(s, p)                                    # caller keeps `kept` AND uses it twice below
        y = kept.neg_op(p).sin_op(p)
        z = kept.mul_op(kept, p)                                        # operand aliases the input
        m = y.lt_op(z, p)                                               # not element-wise-fusable: forces a flush
        w = z.sub_scalar_op(s, p).exp2_op(p).log2_op(p).cbrt_op(p).cos_op(p).neg_op(p).abs_op(p).sqrt_op(p).neg_op(p).neg_op(p)
        p.finish()
        return kept, y, z, m, w

    pf = ag.ArrowComputePipeline(dev, "fused", fuse=True)
    pe = ag.ArrowComputePipeline(dev, "eager", fuse=False)
    got, exp = program(pf), program(pe)
    for g, e in zip(got, exp):
        assert bits(g.raw_values()) == bits(e.raw_values())
        assert n > 5000 or [v is None for v in g.values()] == [v is None for v in e.values()]
    assert pf.stats["fused_chains"] >= 3 and pf.stats["kernels"] < pe.stats["kernels"] + 100
    assert pe.stats["recorded"] == 0


def test_fusing_pipeline_keeps_live_and_reused_intermediates(ag):
    dev = ag.GPU_DEVICE()
    n = 4099
    a, b = _rand(ag, dev, n, 5), _rand(ag, dev, n, 6)
    p = ag.ArrowComputePipeline(dev, "live", fuse=True)
    r1 = a.add_op(b, p)          # kept alive by the caller → must be materialised
    r2 = r1.mul_op(b, p)
    r3 = r1.sub_op(a, p)         # reads r1 again
    del r1
    p.finish()
    assert p.stats["fused_chains"] == 0 and p.stats["kernels"] == 3
    assert bits(r2.raw_values()) == bits(a.add(b).mul(b).raw_values())
    assert bits(r3.raw_values()) == bits(a.add(b).sub(a).raw_values())
    # int chains, > 8 steps split into 8 + rest
    ia = ag.Int32ArrayGPU.from_slice(O.synth_i32(n, 1, 0, 0), dev)
    s = ag.Int32ArrayGPU.from_slice([3], dev)
    q = ag.ArrowComputePipeline(dev, "long", fuse=True)
    x = ia
    for _ in range(11):
        x = x.add_scalar_op(s, q)
    q.sync()  # sync (like anything that needs the stream) issues what is recorded
    assert q.stats["fused_chains"] == 2 and q.stats["fused_ops"] == 11 and q.stats["kernels"] == 2
    assert np.array_equal(x.raw_values(), (O.synth_i32(n, 1, 0, 0).astype(np.int64) + 33).astype(np.int32))


# ---------------------------------------------------------------- chains ending in a compare (agpu_fused_chain_compare)
@pytest.mark.parametrize("n", [1, 63, 256, 257, 4099, 1_000_003])
def test_chain_compare_equals_unfused_predicate(ag, n):
    dev = ag.GPU_DEVICE()
    nulls = n < 5000
    a, b, c, d = (_rand(ag, dev, n, s, nulls=nulls and s != 3) for s in (1, 2, 3, 4))
    s = ag.Float32ArrayGPU.from_slice([0.25], dev)
    for name in ("gt", "gteq", "lt", "lteq", "eq"):
        fused = getattr(ag.FusedChain(a).mul(b).add(c), name)(d)
        unfused = getattr(a.mul(b).add(c), name)(d)
        assert bits(fused.raw_values()) == bits(unfused.raw_values()), (name, n)
        assert (fused.null_buffer is None) == (unfused.null_buffer is None)
        if fused.null_buffer is not None:
            assert bits(fused.null_buffer.raw_values()) == bits(unfused.null_buffer.raw_values())
    # scalar operand, heavy chain, and the zero-step form (a plain compare)
    f2, u2 = ag.FusedChain(a).abs().sqrt().sin().lt(s), a.abs().sqrt().sin().lt(ag.Float32ArrayGPU.broadcast(0.25, n, dev))
    assert bits(f2.raw_values()) == bits(u2.raw_values())
    assert bits(ag.FusedChain(a).gteq(b).raw_values()) == bits(a.gteq(b).raw_values())
    ia = ag.Int32ArrayGPU.from_slice(O.synth_i32(n, 1, 0, 100), dev)
    ib = ag.Int32ArrayGPU.from_slice(O.synth_i32(n, 2, 0, 100), dev)
    k = ag.Int32ArrayGPU.from_slice([50], dev)
    assert bits(ag.FusedChain(ia).add(ib).rem_scalar(k).eq(ib).raw_values()) == bits(ia.add(ib).rem_scalar(k).eq(ib).raw_values())


def test_chain_compare_limits(ag):
    dev = ag.GPU_DEVICE()
    x = ag.Float32ArrayGPU.from_slice([1.0, 2.0], dev)
    ch = ag.FusedChain(x)
    for _ in range(8):
        ch.abs()
    with pytest.raises(ag.ArrowErrorGPU):
        ch.gt(x)  # at most 7 steps before a compare
    with pytest.raises(ag.OperationNotSupported):
        ag.FusedChain(x).gt(ag.Int32ArrayGPU.from_slice([1, 2], dev))


def test_fusing_pipeline_ends_chains_in_compares(ag):
    dev = ag.GPU_DEVICE()
    n = 70_001
    a, b, c, d = (_rand(ag, dev, n, s) for s in (11, 12, 13, 14))
    p = ag.ArrowComputePipeline(dev, "predicate", fuse=True)
    m = a.mul_op(b, p).add_op(c, p).gt_op(d, p)      # (a * b + c) > d — nothing is stored but the bitmap
    keep = a.sub_op(b, p)
    m2 = keep.lteq_op(c, p)                          # `keep` is alive: materialised, compare runs on its own
    p.finish()
    assert p.stats == {"recorded": 5, "kernels": 3, "fused_chains": 1, "fused_ops": 3}
    assert bits(m.raw_values()) == bits(a.mul(b).add(c).gt(d).raw_values())
    assert bits(m2.raw_values()) == bits(a.sub(b).lteq(c).raw_values())
    assert bits(keep.raw_values()) == bits(a.sub(b).raw_values())


# ---------------------------------------------------------------- round 4: a widening cast at the head of a chain (agpu_fused_cast_chain)
# `cast → sin` is what SURVEY §8f-2 names; the reference fuses exactly cast + trig in its *_u8 kernels
# [crates/trigonometry/src/u8_kernel.rs:34-38] and chains `*_op`s in examples/simple.rs:45-72.
NARROW = {"u8": ("UInt8ArrayGPU", np.uint8), "i8": ("Int8ArrayGPU", np.int8), "u16": ("UInt16ArrayGPU", np.uint16), "i16": ("Int16ArrayGPU", np.int16)}


def _narrow(ag, dev, kind, n, seed, nulls=False):
    cls, npd = getattr(ag, NARROW[kind][0]), NARROW[kind][1]
    info = np.iinfo(npd)
    rng = np.random.default_rng(seed)
    vals = rng.integers(info.min, int(info.max) + 1, n, dtype=np.int64).astype(npd)
    if n >= 4:
        vals[:4] = [info.min, info.max, 0, 1]
    if not nulls:
        return cls.from_slice(vals, dev)
    return cls.from_optional_slice([None if rng.random() < 0.2 else int(v) for v in vals], dev)


@pytest.mark.parametrize("kind", list(NARROW))
@pytest.mark.parametrize("n", [0, 1, 5, 511, 512, 513, 1023, 1024, 1025, 4099, 1_000_003])
def test_cast_headed_chain_bit_identical_to_cast_then_ops(ag, kind, n):
    """FusedChain(narrow).steps…  ==  narrow.cast(f32).steps…  bit for bit, values and validity, every tail shape"""
    dev = ag.GPU_DEVICE()
    a = _narrow(ag, dev, kind, n, n + 7, nulls=n < 5000)
    b, c = _rand(ag, dev, n, 2, nulls=0 < n < 5000), _rand(ag, dev, n, 3)
    s, o = ag.Float32ArrayGPU.from_slice([0.0123], dev), ag.Float32ArrayGPU.from_slice([-3.5], dev)
    F32 = ag.Float32ArrayGPU
    cases = {
        "plain cast": (lambda ch: ch, lambda x: x),
        "sin": (lambda ch: ch.sin(), lambda x: x.sin()),
        "cos": (lambda ch: ch.cos(), lambda x: x.cos()),
        "sinh": (lambda ch: ch.sinh(), lambda x: x.sinh()),
        "exp": (lambda ch: ch.exp(), lambda x: x.exp()),   # (16-bit sources: sin / cos / sinh / exp / log alone run the kernel specialised on the function)
        "log": (lambda ch: ch.log(), lambda x: x.log()),
        "scale+offset": (lambda ch: ch.mul_scalar(s).add_scalar(o), lambda x: x.mul_scalar(s).add_scalar(o)),
        "scalars+heavy": (lambda ch: ch.mul_scalar(s).sin().abs().sqrt().neg(), lambda x: x.mul_scalar(s).sin().abs().sqrt().neg()),
        "arrays": (lambda ch: ch.mul(b).add(c).abs().sqrt(), lambda x: x.mul(b).add(c).abs().sqrt()),
        "arrays+heavy": (lambda ch: ch.mul_scalar(s).add(b).cos().sub(c).max(b), lambda x: x.mul_scalar(s).add(b).cos().sub(c).max(b)),
        "acos": (lambda ch: ch.mul_scalar(s).acos(), lambda x: x.mul_scalar(s).acos()),
    }
    for name, (fused_fn, eager_fn) in cases.items():
        fused = fused_fn(ag.FusedChain(a)).finish()
        eager = eager_fn(a.cast(F32))
        assert type(fused) is F32 and fused.len == n
        assert bits(fused.raw_values()) == bits(eager.raw_values()), (kind, n, name)
        assert (fused.null_buffer is None) == (eager.null_buffer is None), (kind, n, name)
        if fused.null_buffer is not None:
            assert bits(fused.null_buffer.raw_values()) == bits(eager.null_buffer.raw_values()), (kind, n, name)


@pytest.mark.parametrize("kind", list(NARROW))
def test_cast_then_trig_equals_the_fused_narrow_kernels_and_the_oracle(ag, kind):
    """every input value of the narrow type: cast → sin / cos / sinh through the chain == the unfused pair (bit for bit) and within
    1 ULP of the oracle's sin_<kind> (f64 libm rounded once); for 8-bit sources also == the reference-shaped sin_u8 kernel"""
    dev = ag.GPU_DEVICE()
    cls, npd = getattr(ag, NARROW[kind][0]), NARROW[kind][1]
    info = np.iinfo(npd)
    vals = np.arange(info.min, int(info.max) + 1, dtype=np.int64).astype(npd)
    a = cls.from_slice(vals, dev)
    odt = {"u8": O.U8, "i8": O.I8, "u16": O.U16, "i16": O.I16}[kind]
    for name, oun in (("sin", O.UN_SIN), ("cos", O.UN_COS), ("sinh", O.UN_SINH)):
        fused = getattr(ag.FusedChain(a), name)().finish().raw_values()
        pair = getattr(a.cast(ag.Float32ArrayGPU), name)().raw_values()
        assert bits(fused) == bits(pair), (kind, name)
        if kind in ("u8", "i8"):
            assert bits(fused) == bits(getattr(a, name)().raw_values()), (kind, name)
        if name != "sinh" or kind in ("u8", "i8"):  # sinh of a 16-bit value overflows to inf beyond |x| ≈ 89: compare where finite
            exp = O.unary(oun, odt, vals)
            fin = np.isfinite(exp) & np.isfinite(fused)
            assert np.array_equal(np.isfinite(exp), np.isfinite(fused))
            d = np.abs(fused[fin].view(np.int32).astype(np.int64) - exp[fin].view(np.int32).astype(np.int64))
            assert d.max() <= 1, (kind, name, int(d.max()))
    # exp / log alone behind the cast: for 16-bit sources the kernel specialised on the function (round 5) — every input value, bit for bit
    # against the unfused pair (log of the non-positive values: -inf / NaN, compared as bits too)
    for name in ("exp", "log"):
        fused = getattr(ag.FusedChain(a), name)().finish().raw_values()
        pair = getattr(a.cast(ag.Float32ArrayGPU), name)().raw_values()
        assert bits(fused) == bits(pair), (kind, name)


def test_fusing_pipeline_collapses_cast_then_sin_into_one_launch(ag):
    """BASELINE config 4 as it is worded — "cast u8→f32 then sin/cos" — written with the reference's `*_op` API on a fusing pipeline:
    ONE launch at finish(), results identical to the eager pair"""
    dev = ag.GPU_DEVICE()
    n = 300_001
    F32 = ag.Float32ArrayGPU
    for kind in NARROW:
        a = _narrow(ag, dev, kind, n, 5)
        s = F32.from_slice([0.5], dev)
        p = ag.ArrowComputePipeline(dev, "cast-sin", fuse=True)
        r = a.cast_op(F32, p).sin_op(p)
        assert p.stats["recorded"] == 2 and p.stats["kernels"] == 0
        p.finish()
        assert p.stats == {"recorded": 2, "kernels": 1, "fused_chains": 1, "fused_ops": 2}, kind
        assert bits(r.raw_values()) == bits(a.cast(F32).sin().raw_values()), kind
        # dyn entry points, a longer chain, a kept intermediate and a compare behind the chain
        p = ag.ArrowComputePipeline(dev, "cast-chain", fuse=True)
        x = ag.cos_op_dyn(ag.mul_scalar_op_dyn(ag.cast_op_dyn(a, ag.ArrowType.Float32Type, p), s, p), p)
        kept = a.cast_op(F32, p)                       # the caller keeps the f32 column: materialised by a plain cast
        y = kept.add_scalar_op(s, p).neg_op(p)
        m = a.cast_op(F32, p).mul_scalar_op(s, p).gt_op(kept, p)   # cast-headed chains store; the compare runs on the stored column
        p.finish()
        assert p.stats["fused_chains"] >= 3, (kind, p.stats)
        e = a.cast(F32)
        assert bits(x.raw_values()) == bits(e.mul_scalar(s).cos().raw_values()), kind
        assert bits(kept.raw_values()) == bits(e.raw_values()) and bits(y.raw_values()) == bits(e.add_scalar(s).neg().raw_values()), kind
        assert bits(m.raw_values()) == bits(e.mul_scalar(s).gt(e).raw_values()), kind


@pytest.mark.parametrize("n_arrays", [4, 5, 7])
def test_fusing_pipeline_cuts_a_cast_headed_chain_at_four_array_operands(ag, n_arrays):
    """agpu_fused_cast_chain takes at most AGPU_CAST_CHAIN_MAX_ARRAYS array operands; a fusing pipeline that recorded
    `cast → 5..7 array ops` with dropped intermediates must cut the chain there (a plain chain continues), not raise at finish()"""
    dev = ag.GPU_DEVICE()
    n = 100_003
    F32 = ag.Float32ArrayGPU
    rng = np.random.default_rng(n_arrays)
    a = ag.UInt8ArrayGPU.from_slice(rng.integers(0, 256, n, dtype=np.int64).astype(np.uint8), dev)
    cols = [F32.from_slice(rng.uniform(-3, 3, n).astype(np.float32), dev) for _ in range(n_arrays)]
    ops = ["add_op", "mul_op", "sub_op", "max_op", "min_op", "add_op", "mul_op"]
    p = ag.ArrowComputePipeline(dev, "cast-arrays", fuse=True)
    r = a.cast_op(F32, p)
    for k in range(n_arrays):
        r = getattr(r, ops[k])(cols[k], p)
    tail = r.neg_op(p)
    p.finish()
    e = a.cast(F32)
    for k in range(n_arrays):
        e = getattr(e, ops[k][:-3])(cols[k])
    assert bits(tail.raw_values()) == bits(e.neg().raw_values())
    assert bits(r.raw_values()) == bits(e.raw_values())  # the kept column in front of the tail
    # cast + 4 arrays in one launch; whatever follows is a second (plain) chain or a single kernel
    assert p.stats["fused_chains"] >= 1 and p.stats["kernels"] <= 3, p.stats
    # the C entry point itself still refuses five arrays — and a pipeline told to run such a chain falls back to single launches
    class Step(C.Structure):
        _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]
    steps = (Step * 5)()
    for k in range(5):
        steps[k].op, steps[k].kind, steps[k].operand = capi.OP_ADD, 2, cols[k % n_arrays].data.ptr
    out = dev.create_empty_buffer(4 * n)
    q = ag.ArrowComputePipeline(dev, "abi")
    assert capi.lib().agpu_fused_cast_chain(q._handle, capi.U8, C.c_void_p(a.data.ptr), C.cast(steps, C.c_void_p), 5, C.c_void_p(out.ptr), n) == capi.ERR_UNSUPPORTED
    assert capi.lib().agpu_fused_cast_chain(q._handle, capi.U8, C.c_void_p(a.data.ptr), C.cast(steps, C.c_void_p), 4, C.c_void_p(out.ptr), n) == capi.OK
    q.sync()


def test_a_failing_chain_does_not_drop_the_rest_of_the_recording(ag, monkeypatch):
    """when one launch raises inside the flush, the ops recorded behind it still run; the first error is raised at the end"""
    dev = ag.GPU_DEVICE()
    F32 = ag.Float32ArrayGPU
    a = F32.from_slice([1.0, 2.0, 3.0, 4.0], dev)
    s = F32.from_slice([2.0], dev)
    p = ag.ArrowComputePipeline(dev, "partial", fuse=True)
    bad = a.add_scalar_op(s, p)
    good = a.mul_scalar_op(s, p)
    real = p._launch_chain
    calls = []

    def flaky(chain):
        calls.append(chain)
        if len(calls) == 1:
            raise capi.ArrowErrorGPU("Runtime", "injected", capi.ERR_HIP)
        return real(chain)

    monkeypatch.setattr(p, "_launch_chain", flaky)
    with pytest.raises(capi.ArrowErrorGPU, match="injected"):
        p.finish()
    assert len(calls) == 2 and not p._pending
    p.sync()
    assert good.values() == [2.0, 4.0, 6.0, 8.0]
    del bad


def test_fused_cast_chain_abi_edges(ag):
    """the C entry point directly: mis-aligned pointers take the element-granular path, unsupported heads / ops are refused"""
    dev = ag.GPU_DEVICE()
    p = ag.ArrowComputePipeline(dev, "abi")
    n = 70_001
    vals = np.random.default_rng(1).integers(0, 65536, n + 8, dtype=np.int64).astype(np.uint16)
    src = dev.create_gpu_buffer_with_data(vals)
    sc = dev.create_gpu_buffer_with_data(np.array([1.5], np.float32))
    out = dev.create_empty_buffer(4 * n + 64)

    class Step(C.Structure):
        _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]

    steps = (Step * 2)()
    steps[0].op, steps[0].kind, steps[0].operand = capi.OP_MUL, 1, sc.ptr
    steps[1].op, steps[1].kind, steps[1].operand = capi.UN_SIN, 0, None
    for in_off, out_off in ((0, 0), (2, 0), (0, 4), (6, 12)):
        capi.call("agpu_fused_cast_chain", p._handle, capi.U16, C.c_void_p(src.ptr + in_off), C.cast(steps, C.c_void_p), 2,
                  C.c_void_p(out.ptr + out_off), n)
        got = dev.retrive_data(out, 4 * n + 64, pipeline=p)[out_off: out_off + 4 * n].view(np.float32)
        x = ag.UInt16ArrayGPU.from_slice(vals[in_off // 2: in_off // 2 + n], dev)
        exp = x.cast(ag.Float32ArrayGPU).mul_scalar(ag.Float32ArrayGPU.from_slice([1.5], dev)).sin().raw_values()
        assert bits(got) == bits(exp), (in_off, out_off)
    lib = capi.lib()
    assert lib.agpu_fused_cast_chain(p._handle, capi.F32, C.c_void_p(src.ptr), C.cast(steps, C.c_void_p), 2, C.c_void_p(out.ptr), n) == capi.ERR_UNSUPPORTED
    assert lib.agpu_fused_cast_chain(p._handle, capi.U32, C.c_void_p(src.ptr), C.cast(steps, C.c_void_p), 2, C.c_void_p(out.ptr), n) == capi.ERR_UNSUPPORTED
    steps[1].op = capi.UN_NOT
    assert lib.agpu_fused_cast_chain(p._handle, capi.U16, C.c_void_p(src.ptr), C.cast(steps, C.c_void_p), 2, C.c_void_p(out.ptr), n) == capi.ERR_UNSUPPORTED
    steps[1].op, steps[0].op = capi.UN_SIN, capi.OP_AND
    assert lib.agpu_fused_cast_chain(p._handle, capi.U16, C.c_void_p(src.ptr), C.cast(steps, C.c_void_p), 2, C.c_void_p(out.ptr), n) == capi.ERR_UNSUPPORTED
    assert lib.agpu_fused_cast_chain(p._handle, capi.U16, C.c_void_p(src.ptr), C.cast(steps, C.c_void_p), 9, C.c_void_p(out.ptr), n) == capi.ERR_ARG
    # `steps` is read before n is looked at — NULL steps with n_steps > 0 is an argument error for every n, never a crash
    for rows in (0, n):
        for head in (capi.U8, capi.U16):
            assert lib.agpu_fused_cast_chain(p._handle, head, C.c_void_p(src.ptr), None, 1, C.c_void_p(out.ptr), rows) == capi.ERR_ARG
    assert lib.agpu_fused_cast_chain(p._handle, capi.U8, None, C.cast(steps, C.c_void_p), 2, None, 0) == capi.OK
