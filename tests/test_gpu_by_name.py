"""Every live `(shader, entry point)` kernel identity of the reference, launched through agpu_launch_by_name_sized with
exactly the arguments the reference's Rust call sites compute — buffers with their BYTE sizes, `new_buffer_size`, the
dispatch size — and checked against the oracle.

The list is data: tests/golden/reference_entry_points.json, generated from /root/reference by
tools/extract_entry_points.py (153 `@compute` entry points outside comments, 148 in WGSL files some Rust source
includes, 5 dead).  `plan(record)` below restates, per shader family, how the reference's caller sizes the call
(file:line cited at each branch); a CPU test asserts that EVERY live record has a plan, the GPU test runs them all."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import golden_runner as G
import oracle as O
from arrow_gpu_amd import _capi as capi

HERE = os.path.dirname(os.path.abspath(__file__))
ENTRIES = json.load(open(os.path.join(HERE, "golden", "reference_entry_points.json")))["entry_points"]
LIVE = [r for r in ENTRIES if not r["dead"]]
DEAD = [r for r in ENTRIES if r["dead"]]

DT = {"f32": (O.F32, np.float32), "i32": (O.I32, np.int32), "u32": (O.U32, np.uint32), "i16": (O.I16, np.int16),
      "u16": (O.U16, np.uint16), "i8": (O.I8, np.int8), "u8": (O.U8, np.uint8)}
BINOP = {"add": O.OP_ADD, "sub": O.OP_SUB, "mul": O.OP_MUL, "div": O.OP_DIV, "rem": O.OP_REM}
CMP = {"gt": O.CMP_GT, "gteq": O.CMP_GTEQ, "lt": O.CMP_LT, "lteq": O.CMP_LTEQ, "eq": O.CMP_EQ}
UNARY = {"sqrt": O.UN_SQRT, "exp": O.UN_EXP, "exp2": O.UN_EXP2, "log": O.UN_LOG, "log2": O.UN_LOG2, "abs": O.UN_ABS,
         "cbrt": O.UN_CBRT, "sin": O.UN_SIN, "cos": O.UN_COS, "acos": O.UN_ACOS, "sinh": O.UN_SINH}
EXACT_UNARY = {"abs", "sqrt"}


def div_ceil(a, b):
    return -(-a // b)


def padded(arr):
    """A wgpu buffer created from this array: its bytes, zero-padded to a multiple of 4 (COPY_BUFFER_ALIGNMENT)."""
    b = np.ascontiguousarray(arr).view(np.uint8).ravel()
    pad = (-len(b)) % 4
    return np.concatenate([b, np.zeros(pad, np.uint8)]) if pad else b.copy()


def vals(dir_, n, seed, lo=None, hi=None):
    from gpu_util import rand_values

    odt, npd = DT[dir_]
    x = rand_values(odt, n, seed)
    if lo is not None:
        rng = np.random.default_rng(seed)
        x = (rng.random(n) * (hi - lo) + lo).astype(np.float32) if npd is np.float32 else rng.integers(lo, hi, n).astype(npd)
    return x


class Plan:
    """inputs: list of byte arrays (wgpu buffers); out_bytes; dispatch; expect(out_bytes_array) → None or raises."""

    def __init__(self, inputs, out_bytes, dispatch, check, out_init=None):
        self.inputs, self.out_bytes, self.dispatch, self.check, self.out_init = inputs, out_bytes, dispatch, check, out_init


def exact(exp_bytes):
    def chk(out):
        e = np.ascontiguousarray(exp_bytes).view(np.uint8).ravel()
        assert out[: len(e)].tobytes() == e.tobytes()
        assert not out[len(e):].any(), "bytes past the processed lanes must stay zero"
    return chk


def f32_close(exp, ulp):
    from gpu_util import max_ulp, nan_aware_bits_equal

    def chk(out):
        got = out[: 4 * len(exp)].view(np.float32)
        if ulp == 0:
            assert nan_aware_bits_equal(got, exp)
        else:
            assert max_ulp(got, exp) <= ulp
    return chk


def lanes_view(buf_bytes, npd):
    return buf_bytes.view(npd)


def plan(r, n=1003):
    """How the reference's Rust caller launches this entry point.  n = array length (deliberately ragged)."""
    crate, dir_, file = r["shader_key"].split("/")
    ep = r["entry_point"]
    if dir_ in DT:
        odt, npd = DT[dir_]
        item = np.dtype(npd).itemsize
    # ---- arithmetic [crates/arithmetic/src/lib.rs:11-50 (scalar), 54-94 (array); arithmetic_kernels.rs:300-307 (neg)]
    if crate == "arithmetic" and file in ("array", "scalar", "neg"):
        a = padded(vals(dir_, n, 1))
        la = lanes_view(a, npd)
        disp = div_ceil(div_ceil(len(a), item), 256)
        if file == "neg":
            exp = O.unary(O.UN_NEG, odt, la)
            return Plan([a], len(a), disp, f32_close(exp, 0))
        if file == "scalar":
            s = vals(dir_, 1, 2, 1, 50) if ep.endswith(("div", "rem")) else vals(dir_, 1, 2)
            sb = padded(s)  # create_scalar_buffer: one element in a 4-byte buffer
            op = BINOP[ep.split("_", 1)[1]]
            exp = O.scalar(op, odt, la, s)
            return Plan([a, sb], len(a), disp, f32_close(exp, 0) if npd is np.float32 else exact(exp))
        b = padded(vals(dir_, n, 3))
        op = {"bitwise_and": O.OP_AND, "bitwise_or": O.OP_OR}.get(ep) or BINOP[ep.split("_")[0]]
        exp = O.binary(op, odt, la, lanes_view(b, npd))
        return Plan([a, b], len(a), disp, f32_close(exp, 0) if npd is np.float32 else exact(exp))
    # ---- Sum: ONE level per dispatch [aggregate_kernels.rs:26-43: dispatch = new_length, out = new_length * 4]
    if crate == "arithmetic" and file == "aggregate":
        x = vals(dir_, n, 4, -1000, 1000) if npd is np.float32 else vals(dir_, n, 4)
        groups = div_ceil(n, 256)
        exp = np.array([O.reduce(O.RED_SUM, odt, x[g * 256:(g + 1) * 256]) for g in range(groups)], npd)
        return Plan([padded(x)], groups * 4, groups, exact(exp))
    # ---- broadcast [crates/array/src/array/f32_gpu.rs:14-37: dispatch = ceil(len/256), out = len*4]
    if crate == "array" and file == "broadcast":
        s = vals(dir_, 1, 5)
        return Plan([padded(s)], n * 4, div_ceil(n, 256), exact(np.full(n, s[0], npd)))
    # ---- logical [crates/logical/src/lib.rs:88-187; boolean.rs:106-146]
    if crate == "logical":
        if file == "any":  # apply_unary_function(data, 4, ANY_SHADER, "any", ceil(words/256))
            bits = np.zeros(bitmap_words(n) * 4, np.uint8)
            bits[min(37, len(bits) - 1)] = 0x10

            def chk(out):
                assert out.view(np.uint32)[0] > 0
            return Plan([bits], 4, div_ceil(len(bits) // 4, 256), chk)
        if file == "countbitones":  # dispatch = ceil(bytes/1024)
            w = vals("u32", n, 6)
            return Plan([padded(w)], 4 * n, div_ceil(4 * n, 1024), exact(O.unary(O.UN_POPCOUNT, O.U32, w)))
        a = padded(vals(dir_, n, 7))
        la = lanes_view(a, npd)
        disp = div_ceil(div_ceil(len(a), item), 256)
        if file == "not":
            return Plan([a], len(a), disp, exact(O.unary(O.UN_NOT, odt, la)))
        if file == "logical":
            b = padded(vals(dir_, n, 8))
            op = {"bitwise_and": O.OP_AND, "bitwise_or": O.OP_OR, "bitwise_xor": O.OP_XOR}[ep]
            return Plan([a, b], len(a), disp, exact(O.binary(op, odt, la, lanes_view(b, npd))))
        if file == "shift":  # rhs is a UInt32ArrayGPU with one amount per lane
            amt = np.random.default_rng(9).integers(0, 8 * item, len(la)).astype(np.uint32)
            op = O.OP_SHL if ep == "bitwise_shl" else O.OP_SHR
            return Plan([a, padded(amt)], len(a), disp, exact(O.binary(op, odt, la, amt)))
    # ---- compare [crates/compare/src/lib.rs:85-140: out = data.size() (over-allocated), dispatch by ITEM_SIZE]
    if crate == "compare":
        a, b = vals(dir_, n, 10), vals(dir_, n, 11)
        b[::3] = a[::3]
        a, b = padded(a), padded(b)
        la, lb = lanes_view(a, npd), lanes_view(b, npd)
        disp = div_ceil(div_ceil(len(a), item), 256)
        if file == "cmp":
            return Plan([a, b], len(a), disp, exact(O.compare(CMP[ep], odt, la, lb)))
        op = O.OP_MAX if ep == "max_" else O.OP_MIN
        exp = O.binary(op, odt, la, lb)
        return Plan([a, b], len(a), disp, f32_close(exp, 0) if npd is np.float32 else exact(exp))
    # ---- cast [crates/cast/src/lib.rs:40-67 impl_cast!(…, item_size, buffer_size_mul); f32_cast.rs:8-19; boolean_cast.rs:39-60]
    if crate == "cast":
        to = {"f32": (O.F32, np.float32), "u8": (O.U8, np.uint8), "u16": (O.U16, np.uint16), "u32": (O.U32, np.uint32),
              "i16": (O.I16, np.int16), "i32": (O.I32, np.int32)}[file[5:]]
        if dir_ == "boolean":
            bits = O.synth_bits(n, 12, 0, 0.5)[: bitmap_words(n) * 4]
            exp = O.cast(O.BOOL, O.F32, bits, n)
            return Plan([padded(bits)], n * 4, div_ceil(n, 256), exact(exp))
        if dir_ == "f32":
            x = vals("f32", n, 13, -300, 70000)
            kat = [0, 1, -1, 5713, -5713, 255, 256]  # the reference's own vector, cast/src/f32_cast.rs:40-48
            x[: min(7, n)] = kat[: min(7, n)]
            a = padded(x)
            out_bytes = (len(a) // 4 + 3) // 4 * 4
            return Plan([a], out_bytes, div_ceil(div_ceil(len(a), 16), 256), exact(O.cast(O.F32, O.U8, x)))
        a = padded(vals(dir_, n, 14))
        la = lanes_view(a, npd)
        mul = to[1]().itemsize // item
        exp = O.cast(odt, to[0], la)
        return Plan([a], len(a) * mul, div_ceil(div_ceil(len(a), item), 256), exact(exp))
    # ---- math [crates/math/src/lib.rs:138-193]
    if crate == "math":
        name = ep[:-1]
        if file in ("floatunary", "unary"):
            x = vals(dir_, n, 15, 0.01, 50.0) if npd is np.float32 else vals(dir_, n, 15)
            a = padded(x)
            exp = O.unary(UNARY[name], odt, lanes_view(a, npd))
            disp = div_ceil(div_ceil(len(a), item), 256)
            if npd is np.float32:
                return Plan([a], len(a), disp, f32_close(exp, 0 if name in EXACT_UNARY else G.MAX_ULP))
            return Plan([a], len(a), disp, exact(exp))
        if npd is np.float32:
            x, y = vals("f32", n, 16, 0.1, 20.0), vals("f32", n, 17, -3.0, 3.0)
        else:
            x, y = vals("i32", n, 16, -9, 10), vals("i32", n, 17, 0, 9)
        a, b = padded(x), padded(y)
        exp = O.binary(O.OP_POW, odt, x, y)
        disp = div_ceil(div_ceil(len(a), item), 256)
        return Plan([a, b], len(a), disp, f32_close(exp, G.MAX_ULP) if npd is np.float32 else exact(exp))
    # ---- trigonometry [crates/trigonometry/src/lib.rs:85-137: out = size * BUFFER_SIZE_MULTIPLIER]
    if crate == "trigonometry":
        name = ep.rsplit("_", 1)[0]
        x = vals(dir_, n, 18, -1.0, 1.0) if (npd is np.float32 and name == "acos") else (
            vals(dir_, n, 18, -20.0, 20.0) if npd is np.float32 else vals(dir_, n, 18))
        a = padded(x)
        la = lanes_view(a, npd)
        exp = O.unary(UNARY[name], odt, la)
        mul = 4 // item
        return Plan([a], len(a) * mul, div_ceil(div_ceil(len(a), item), 256), f32_close(exp, G.MAX_ULP))
    # ---- routines [crates/routines/src/lib.rs:81-171, take.rs:9-55, put.rs:9-56, bool.rs:15-128, merge.rs:17-86]
    if crate == "routines":
        rng = np.random.default_rng(19)
        if dir_ in ("32bit", "16bit", "8bit") and file == "merge":
            w = {"32bit": 4, "16bit": 2, "8bit": 1}[dir_]
            npd2 = {4: np.uint32, 2: np.uint16, 1: np.uint8}[w]
            a, b = padded(rng.integers(0, 1 << (8 * w), n).astype(npd2)), padded(rng.integers(0, 1 << (8 * w), n).astype(npd2))
            lanes = len(a) // w
            mask = O.synth_bits(lanes, 20, 0, 0.5)[: bitmap_words(lanes) * 4]
            exp = O.merge(w, a.view(npd2), b.view(npd2), mask)
            # dispatch_size = data.size() / ITEM_SIZE, passed as the WORKGROUP count (routines/src/lib.rs:88)
            return Plan([a, b, padded(mask)], len(a), len(a) // w, exact(exp))
        if dir_ == "32bit" and file == "take":
            v = rng.integers(0, 1 << 32, 777).astype(np.uint32)
            idx = rng.integers(0, 777, n).astype(np.uint32)
            return Plan([padded(v), padded(idx)], n * 4, div_ceil(n, 256), exact(v[idx]))
        if dir_ == "32bit" and file == "put":
            src = rng.integers(0, 1 << 32, 500).astype(np.uint32)
            dst = rng.integers(0, 1 << 32, 2000).astype(np.uint32)
            si = rng.integers(0, 500, n).astype(np.uint32)
            di = rng.permutation(2000)[:n].astype(np.uint32)
            exp = dst.copy()
            exp[di] = src[si]
            return Plan([padded(src), padded(si), padded(di)], 4 * 2000, div_ceil(n, 256), exact(exp), out_init=padded(dst))
        if dir_ == "bool" and file == "merge":
            words = bitmap_words(n)
            a, b, m = (O.synth_bits(words * 32, 21 + k, 0, 0.5)[: words * 4] for k in range(3))
            exp = (a & m) | (b & ~m)
            return Plan([a, b, m], words * 4, words, exact(exp))  # dispatch_size = data.size() / 4
        if dir_ == "bool" and file == "take":
            nbits = 3000
            bits = O.synth_bits(nbits, 24, 0, 0.5)[: bitmap_words(nbits) * 4]
            idx = rng.integers(0, nbits, n).astype(np.uint32)
            exp = O.take_bits(bits, nbits, idx)[: bitmap_words(n) * 4]
            return Plan([padded(bits), padded(idx)], div_ceil(n, 32) * 4, div_ceil(n, 256), exact(exp))
        if dir_ == "bool" and file == "put":
            m = 200  # ≤ 256: the reference dispatches ceil(ceil(len/32)/256) workgroups = 256 invocations (bool.rs:115)
            sbits = O.synth_bits(1024, 25, 0, 0.5)[:128]
            dbits = O.synth_bits(2048, 26, 0, 0.5)[:256]
            si = rng.integers(0, 1024, m).astype(np.uint32)
            di = rng.permutation(2048)[:m].astype(np.uint32)
            exp = O.put_bits(sbits, si, dbits.copy(), di)
            return Plan([sbits, padded(si), padded(di)], 256, div_ceil(div_ceil(m, 32), 256), exact(exp), out_init=dbits)
        if dir_ == "u32" and file == "merge_null_buffer":
            words = bitmap_words(n)
            a, b = (O.synth_bits(words * 32, 27 + k, 0, 0.6)[: words * 4] for k in range(2))
            exp = {"merge_selected": a & b, "merge_nulls": a & b, "merge_or": a | b, "merge_not_selected": a & ~b}[ep]
            return Plan([a, b], words * 4, div_ceil(words, 256), exact(exp))
    return None


def bitmap_words(n_bits):
    return (n_bits + 31) // 32


# ---------------------------------------------------------------------------------------------- CPU: the table is complete
def test_fixture_lists_the_reference_entry_points():
    assert len(ENTRIES) == 153 and len(LIVE) == 148 and len(DEAD) == 5
    assert len({(r["shader_key"], r["entry_point"]) for r in ENTRIES}) == len(ENTRIES)
    assert {"logical/u32/countbitones"} <= {r["shader_key"] for r in LIVE}


def test_every_live_entry_point_has_a_plan():
    missing = [(r["shader_key"], r["entry_point"]) for r in LIVE if plan(r) is None]
    assert not missing, missing
    for r in LIVE:  # the plan binds every binding the WGSL declares: k inputs + the one output the caller allocates
        assert len(plan(r).inputs) == len(r["bindings"]) - 1, (r["shader_key"], r["entry_point"])


# ---------------------------------------------------------------------------------------------- the shader argument is the TEXT
SHADERS = json.load(open(os.path.join(HERE, "golden", "reference_shader_hashes.json")))["constants"]


def fnv1a64(data: bytes) -> int:
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def test_fixture_lists_the_reference_shader_constants():
    """76 `const *_SHADER` in the op crates (every include_str! of the reference sits in one), 71 distinct texts; together
    they cover exactly the 72 live WGSL files"""
    assert len(SHADERS) == 76 and len({(c["fnv1a64"], c["bytes"]) for c in SHADERS}) == 71
    assert {c["shader_key"] for c in SHADERS} == {r["shader_key"] for r in LIVE}
    twins = [(c["const"], c["shader_key"], c["resolves_to"]) for c in SHADERS if c["shader_key"] != c["resolves_to"]]
    assert twins == [("U32_MIN_MAX_SHADER", "compare/u32/min_max", "compare/i32/min_max")]  # byte-identical files


def test_every_shader_constant_resolves_by_hash():
    """host only (no GPU): (FNV-1a 64, length) of every constant's text → the path key of its kernel file"""
    lib = capi.lib()
    out = C.create_string_buffer(64)
    for c in SHADERS:
        capi.check(lib.agpu_shader_key_for_hash(int(c["fnv1a64"], 16), c["bytes"], out, 64), c["const"])
        assert out.value.decode() == c["resolves_to"], c
    assert lib.agpu_shader_key_for_hash(0x1234, 10, out, 64) == capi.ERR_UNSUPPORTED
    assert lib.agpu_shader_key_for_source(b"fn main() {}", 12, out, 64) == capi.ERR_UNSUPPORTED
    assert lib.agpu_shader_key_for_hash(int(SHADERS[0]["fnv1a64"], 16), SHADERS[0]["bytes"], out, 4) == capi.ERR_ARG


@pytest.mark.skipif(not os.path.isdir("/root/reference/crates"), reason="the reference's WGSL exists only in the build container")
def test_the_reference_texts_themselves_resolve():
    """build container only: the TEXT of every shader constant, assembled the way rustc evaluates include_str! / concat!,
    handed to agpu_shader_key_for_source — the reference's literal `shader` argument
    [ref: crates/arithmetic/src/f32.rs:10-15, crates/compare/src/u8.rs:3-12, gpu_device.rs:145-168]"""
    lib = capi.lib()
    out = C.create_string_buffer(64)
    for c in SHADERS:
        text = b"".join(open(os.path.join("/root/reference", f), "rb").read() for f in c["includes"])
        assert len(text) == c["bytes"] and f"{fnv1a64(text):016x}" == c["fnv1a64"], c["const"]
        capi.check(lib.agpu_shader_key_for_source(text, len(text), out, 64), c["const"])
        assert out.value.decode() == c["resolves_to"]
        assert b"\0" not in text  # travels as a C string through agpu_launch_by_name*


# ---------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ctx():
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    dev = GpuDevice(0)
    return dev, ArrowComputePipeline(dev, "by_name")


def launch(dev, p, r, pl, shader=None):
    bufs = [dev.create_gpu_buffer_with_data(b) for b in pl.inputs]
    out = dev.create_empty_buffer(max(pl.out_bytes, 4), zero_fill=True)  # the reference's outputs are zero-filled
    if pl.out_init is not None:
        capi.call("agpu_upload", p._handle, C.c_void_p(out.ptr), C.c_void_p(pl.out_init.ctypes.data), len(pl.out_init))
    ptrs = (C.c_void_p * len(bufs))(*[b.ptr for b in bufs])
    sizes = (C.c_uint64 * len(bufs))(*[len(b) for b in pl.inputs])
    st = capi.lib().agpu_launch_by_name_sized(p._handle, (shader or r["shader_key"]).encode(), r["entry_point"].encode(), ptrs, sizes,
                                              len(bufs), C.c_void_p(out.ptr), pl.out_bytes, pl.dispatch)
    capi.check(st, f'{r["shader_key"]}::{r["entry_point"]}')
    return dev.retrive_data(out, pl.out_bytes, pipeline=p)


@pytest.mark.gpu
@pytest.mark.parametrize("r", LIVE, ids=[f'{r["shader_key"]}::{r["entry_point"]}' for r in LIVE])
def test_entry_point_through_the_sized_seam(ctx, r):
    dev, p = ctx
    for n in (1003, 5):  # ragged; 5 is the size of most of the reference's own tests
        pl = plan(r, n)
        pl.check(launch(dev, p, r, pl))


@pytest.mark.gpu
@pytest.mark.parametrize("c", SHADERS, ids=[f'{c["rust"].split(":")[0].replace("crates/", "")}::{c["const"]}' for c in SHADERS])
def test_shader_constant_named_by_its_text_reaches_the_same_kernels(ctx, c):
    """the `shader` argument as the reference passes it — the TEXT, here named by (FNV-1a 64, length) because WGSL never
    travels to the GPU box — for every one of the 76 constants: every live entry point of the file it resolves to runs
    through the sized seam and matches the oracle, exactly as with the path key"""
    dev, p = ctx
    ref = f'#{c["fnv1a64"]}:{c["bytes"]}'
    eps = [r for r in LIVE if r["shader_key"] == c["resolves_to"]]
    assert eps
    for r in eps:
        pl = plan(r, 1003)
        pl.check(launch(dev, p, r, pl, shader=ref))


@pytest.mark.gpu
def test_unknown_shader_text_is_refused(ctx):
    dev, p = ctx
    buf = dev.create_empty_buffer(64, zero_fill=True)
    ptrs, sizes = (C.c_void_p * 2)(buf.ptr, buf.ptr), (C.c_uint64 * 2)(64, 64)
    lib = capi.lib()
    wgsl = b"@group(0) @binding(0) var<storage, read> a: array<f32>;\n@compute @workgroup_size(256) fn add_f32() {}\n"
    for shader in (wgsl, b"#0123456789abcdef:12", b"#zz", b"no/such"):
        assert lib.agpu_launch_by_name_sized(p._handle, shader, b"add_f32", ptrs, sizes, 2, C.c_void_p(buf.ptr), 64, 1) in (
            capi.ERR_UNSUPPORTED, capi.ERR_ARG), shader
    assert lib.agpu_launch_by_name(p._handle, wgsl, b"add_f32", ptrs, 2, C.c_void_p(buf.ptr), 16) == capi.ERR_UNSUPPORTED


@pytest.mark.gpu
def test_dead_entry_points_and_bad_arguments_are_rejected(ctx):
    dev, p = ctx
    buf = dev.create_empty_buffer(64, zero_fill=True)
    ptrs, sizes = (C.c_void_p * 2)(buf.ptr, buf.ptr), (C.c_uint64 * 2)(64, 64)
    lib = capi.lib()
    for r in DEAD:
        assert lib.agpu_launch_by_name_sized(p._handle, r["shader_key"].encode(), r["entry_point"].encode(), ptrs, sizes, 2,
                                             C.c_void_p(buf.ptr), 64, 1) == capi.ERR_UNSUPPORTED
    assert lib.agpu_launch_by_name_sized(p._handle, b"arithmetic/f32/array", b"add_f32", ptrs, sizes, 1, C.c_void_p(buf.ptr), 64, 1) == capi.ERR_ARG
    # a caller-supplied key that names a bit-packed cast target must be refused, not divide by its element size of 0
    for key in (b"cast/u8/cast_bool", b"cast/i16/cast_boolean", b"cast/u16/cast_bool"):
        assert lib.agpu_launch_by_name_sized(p._handle, key, b"cast_bool", ptrs, sizes, 1, C.c_void_p(buf.ptr), 64, 1) == capi.ERR_UNSUPPORTED
    bad = (C.c_uint64 * 2)(63, 64)
    assert lib.agpu_launch_by_name_sized(p._handle, b"arithmetic/f32/array", b"add_f32", ptrs, bad, 2, C.c_void_p(buf.ptr), 64, 1) == capi.ERR_SHAPE
    # the dispatch bounds the work like the shader's invocation count: 1 workgroup = 256 words
    n = 1000
    a = dev.create_gpu_buffer_with_data(np.arange(n, dtype=np.int32))
    out = dev.create_empty_buffer(4 * n, zero_fill=True)
    ptrs2, sizes2 = (C.c_void_p * 2)(a.ptr, a.ptr), (C.c_uint64 * 2)(4 * n, 4 * n)
    capi.check(lib.agpu_launch_by_name_sized(p._handle, b"arithmetic/i32/array", b"add_i32", ptrs2, sizes2, 2, C.c_void_p(out.ptr), 4 * n, 1), "add")
    got = dev.retrive_data(out, 4 * n, pipeline=p).view(np.int32)
    assert np.array_equal(got[:256], 2 * np.arange(256)) and not got[256:].any()


# ---------------------------------------------------------------------------------------------- robust-access fuzz
def _lanes_model(family, W, OW, out_bytes, inv):
    """Independent restatement of the rule in include/arrow_gpu.h: a lane is processed iff it lies inside
    dispatch × 256 invocations AND inside every binding.  W = words per input binding, OW = output words."""
    if family == "ew_f32":      # invocation i ↔ word i of a, b, out
        return min(inv, W[0], W[1], OW)
    if family == "scalar_u16":  # word i of a and out (2 lanes); binding 1 is the operand
        return 2 * min(inv, W[0], OW)
    if family == "cmp_u8":      # invocation ↔ input word (4 lanes); 32 lanes per output word
        return min(4 * min(inv, W[0], W[1]), 32 * OW)
    if family == "cast_u8_f32":  # invocation ↔ input word → 4 f32
        return min(4 * min(inv, W[0]), out_bytes // 4)
    if family == "cast_f32_u8":  # invocation ↔ OUTPUT word ← 4 f32
        return min(4 * min(inv, OW), W[0])
    if family == "cast_bool_f32":
        return min(inv, OW, 32 * W[0])
    raise AssertionError(family)


@pytest.mark.gpu
def test_sized_seam_with_arbitrary_sizes_never_touches_lanes_outside_a_binding(ctx):
    """Random byte sizes and dispatch sizes (mismatched on purpose): the lanes the rule selects equal the oracle, every
    byte of the output past them keeps the zero the reference's allocator put there."""
    dev, p = ctx
    rng = np.random.default_rng(20250418)
    lib = capi.lib()
    for trial in range(120):
        family = ["ew_f32", "scalar_u16", "cmp_u8", "cast_u8_f32", "cast_f32_u8", "cast_bool_f32"][trial % 6]
        W = [int(rng.integers(1, 700)), int(rng.integers(1, 700))]
        OW = int(rng.integers(1, 2800 if family in ("cast_u8_f32",) else 700))
        disp = int(rng.integers(1, 4))
        inv = disp * 256
        if family == "ew_f32":
            key, ep = b"arithmetic/f32/array", b"mul_f32"
            a, b = rng.standard_normal(W[0]).astype(np.float32), rng.standard_normal(W[1]).astype(np.float32)
            ins, n = [a, b], None
            def expect(n): return O.binary(O.OP_MUL, O.F32, a[:n], b[:n]).view(np.uint8)  # noqa: E704
        elif family == "scalar_u16":
            key, ep = b"arithmetic/u16/scalar", b"u16_add"
            a = rng.integers(0, 65536, 2 * W[0]).astype(np.uint16)
            s = np.array([7, 0], np.uint16)
            ins = [a, s]
            def expect(n): return O.scalar(O.OP_ADD, O.U16, a[:n], s[:1]).view(np.uint8)  # noqa: E704
        elif family == "cmp_u8":
            key, ep = b"compare/u8/cmp", b"lteq"
            a, b = rng.integers(0, 4, 4 * W[0]).astype(np.uint8), rng.integers(0, 4, 4 * W[1]).astype(np.uint8)
            ins = [a, b]
            def expect(n): return O.compare(O.CMP_LTEQ, O.U8, a[:n], b[:n])[: (n + 7) // 8]  # noqa: E704
        elif family == "cast_u8_f32":
            key, ep = b"cast/u8/cast_f32", b"cast_f32"
            a = rng.integers(0, 256, 4 * W[0]).astype(np.uint8)
            ins = [a]
            def expect(n): return O.cast(O.U8, O.F32, a[:n]).view(np.uint8)  # noqa: E704
        elif family == "cast_f32_u8":
            key, ep = b"cast/f32/cast_u8", b"cast_u8"
            a = (rng.random(W[0]) * 600 - 100).astype(np.float32)
            ins = [a]
            def expect(n): return O.cast(O.F32, O.U8, a[:n])  # noqa: E704
        else:
            key, ep = b"cast/boolean/cast_f32", b"cast_f32"
            a = rng.integers(0, 256, 4 * W[0]).astype(np.uint8)
            ins = [a]
            def expect(n): return O.cast(O.BOOL, O.F32, a, n).view(np.uint8)  # noqa: E704
        n = _lanes_model(family, W, OW, 4 * OW, inv)
        bufs = [dev.create_gpu_buffer_with_data(x) for x in ins]
        out = dev.create_empty_buffer(4 * OW, zero_fill=True)
        ptrs = (C.c_void_p * len(bufs))(*[b_.ptr for b_ in bufs])
        sizes = (C.c_uint64 * len(bufs))(*[x.nbytes if x is not ins[-1] or family != "scalar_u16" else 4 for x in ins])
        capi.check(lib.agpu_launch_by_name_sized(p._handle, key, ep, ptrs, sizes, len(bufs), C.c_void_p(out.ptr), 4 * OW, disp), ep.decode())
        got = dev.retrive_data(out, 4 * OW, pipeline=p)
        exp = np.ascontiguousarray(expect(n)).view(np.uint8).ravel()
        if family == "cmp_u8":  # bits past n inside the last byte are zero too
            assert got[: len(exp)].tobytes() == exp.tobytes(), (family, W, OW, disp, n)
        else:
            assert got[: len(exp)].tobytes() == exp.tobytes(), (family, W, OW, disp, n)
        assert not got[len(exp):].any(), (family, W, OW, disp, n)
