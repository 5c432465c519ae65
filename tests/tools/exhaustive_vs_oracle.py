#!/usr/bin/env python3
"""EVERY f32 bit pattern through the product's unary kernels (agpu_unary over device buffers, the path a caller takes) against
the CPU ORACLE — f64 libm rounded once to f32 (oracle/agpu_oracle.c orc_unary) — not against the device's own f64 library like
agpu_selftest_unary_f32 does.  16 chunks of 2^28 patterns per function; the oracle side runs in a pool of forked workers over the
host cores.  Reports the largest ULP distance, a pattern that attains it, and how many patterns differ at all.

    python tests/tools/exhaustive_vs_oracle.py [sin cos ...]      → gpurun_out/r03_exhaustive_vs_oracle.json

The reference itself pins these functions to 0.01 absolute on a handful of points (crates/trigonometry/src/f32_kernel.rs:62-132,
crates/math/src/f32.rs:84-271); ≤ 1 ULP is north_star's bar.  Test infrastructure: lives under tests/, uses oracle/ as the checker."""
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from arrow_gpu_amd import _capi as capi  # noqa: E402

OPS = {"sin": capi.UN_SIN, "cos": capi.UN_COS, "sinh": capi.UN_SINH, "acos": capi.UN_ACOS, "exp": capi.UN_EXP, "exp2": capi.UN_EXP2,
       "log": capi.UN_LOG, "log2": capi.UN_LOG2, "sqrt": capi.UN_SQRT, "cbrt": capi.UN_CBRT}
CASTS = {"cast_f32_u8": capi.U8, "cast_f32_i8": capi.I8, "cast_f32_u16": capi.U16, "cast_f32_i16": capi.I16, "cast_f32_u32": capi.U32,
         "cast_f32_i32": capi.I32}  # every f32 bit pattern through agpu_cast against the oracle's cast: bit-exact or not
BIN16 = {f"{tn}_{on}": (t, o) for tn, t in (("u16", capi.U16), ("i16", capi.I16))
         for on, o in (("add", capi.OP_ADD), ("sub", capi.OP_SUB), ("mul", capi.OP_MUL), ("min", capi.OP_MIN), ("max", capi.OP_MAX),
                       ("and", capi.OP_AND), ("or", capi.OP_OR), ("xor", capi.OP_XOR))}  # every operand PAIR of a 16-bit type: 2^32 pairs
CHUNK = 1 << 28
PAIRS = 1 << 26    # pow: pairs per launch
_X = _GOT = _Y = None  # inherited by the forked workers


def _ordered(v):
    b = v.view(np.int32).astype(np.int64)
    return np.where(b < 0, np.int64(-(2 ** 31)) - b, b)


def _check_bin16(args):
    (t, o), lo, hi = args
    ab = _X.view(np.uint16)[2 * lo:2 * hi].reshape(-1, 2)  # pair j of the chunk = (a, b) = (high half, low half) of the pair's number
    dt = np.uint16 if t == capi.U16 else np.int16
    a, b = np.ascontiguousarray(ab[:, 1]).view(dt), np.ascontiguousarray(ab[:, 0]).view(dt)
    exp = O.binary(o, t, a, b)
    got = _GOT.view(np.uint16)[lo:hi].view(dt)
    bad = got != exp
    k = int(bad.argmax()) if bad.any() else 0
    return int(bad.sum()), int(_X.view(np.uint32)[lo + k])


def _check_cast(args):
    to, lo, hi = args
    x = _X[lo:hi]
    exp = O.cast(O.F32, to, x)
    got = _GOT.view(np.uint8)[lo * exp.itemsize:hi * exp.itemsize].view(exp.dtype)
    bad = got != exp
    k = int(bad.argmax()) if bad.any() else 0
    return int(bad.sum()), int(x.view(np.uint32)[k])


def _check(args):
    op, lo, hi = args
    x, got = _X[lo:hi], _GOT[lo:hi]
    exp = O.binary(O.OP_POW, O.F32, x, _Y[lo:hi]) if op == "pow" else O.unary(op, O.F32, x)
    nan_e, nan_g = np.isnan(exp), np.isnan(got)
    if not np.array_equal(nan_e, nan_g):
        k = int(np.flatnonzero(nan_e != nan_g)[0])
        return 1 << 31, int(x.view(np.uint32)[k]), int((nan_e != nan_g).sum())
    d = np.abs(_ordered(np.where(nan_e, np.float32(0), got)) - _ordered(np.where(nan_e, np.float32(0), exp)))
    same_zero = (got == 0) & (exp == 0)  # ±0 compare equal in the ULP metric; their SIGN is checked separately below
    d = np.where(same_zero, 0, d)
    sign_mismatch = int((same_zero & (np.signbit(got) != np.signbit(exp))).sum())
    k = int(d.argmax())
    return int(d[k]), int(x.view(np.uint32)[k]), int((d != 0).sum()), sign_mismatch


def main():
    global _X, _GOT, _Y
    import mmap

    names = sys.argv[1:] or list(OPS) + ["pow"] + list(CASTS)  # (the 16-bit pair sweeps, ≈ 15 s each, only by name: "u16_add" …, or "bin16" for all)
    if "bin16" in names:
        names = [x for x in names if x != "bin16"] + list(BIN16)
    workers = int(os.environ.get("AGPU_ORACLE_WORKERS", "16"))
    # the two chunk buffers are shared memory and the workers are forked BEFORE this process touches the GPU: no child ever
    # carries HIP state
    # (anonymous shared mappings: inherited by fork, nothing under /dev/shm to size or to clean up)
    shm_x, shm_g, shm_y = mmap.mmap(-1, 4 * CHUNK), mmap.mmap(-1, 4 * CHUNK), mmap.mmap(-1, 4 * PAIRS)
    _X = np.frombuffer(shm_x, np.float32)
    _GOT = np.frombuffer(shm_g, np.float32)
    _Y = np.frombuffer(shm_y, np.float32)
    pool = mp.get_context("fork").Pool(workers)
    try:
        return _run(names, workers, pool)
    finally:
        pool.terminate()
        pool.join()
        _X = _GOT = _Y = None


def _pow(p, pool, workers, din, dout, dev, vp):
    """pow(x, y) is binary — no exhaustive sweep: 16 launches of 2^26 pseudo-random pairs (2^30 pairs) over four domains against the
    oracle's f64 pow rounded once."""
    dy = dev.create_empty_buffer(4 * PAIRS)
    rng = np.random.default_rng(20250418)
    t0 = time.time()
    out = {}
    for dom in ("any finite or infinite positive base (denormals included), |y| = 2^[-8, 8]", "base -> 1 (1 +- 2^[-24, -1]), |y| = 2^[0, 30]",
                "results across overflow / underflow (y log2 x in [-160, 140])", "negative bases: integer and non-integer exponents"):
        worst, worst_xy, differing, zero_sign = 0, (0, 0), 0, 0
        for rep in range(4):
            n = PAIRS
            if dom.startswith("any"):
                x = rng.integers(0, 0x7F800001, n, dtype=np.uint32).view(np.float32)
                y = (2.0 ** rng.uniform(-8, 8, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
            elif dom.startswith("base"):
                x = (1.0 + 2.0 ** rng.uniform(-24, -1, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
                y = (2.0 ** rng.uniform(0, 30, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
            elif dom.startswith("results"):
                lx = rng.uniform(-126, 127, n)
                lx = np.where(np.abs(lx) < 1e-3, 1.0, lx)
                x = (2.0 ** lx).astype(np.float32)
                y = (rng.uniform(-160, 140, n) / lx).astype(np.float32)
            else:
                x = (-(2.0 ** rng.uniform(-20, 20, n))).astype(np.float32)
                y = np.where(rng.random(n) < 0.7, rng.integers(-40, 41, n).astype(np.float64), rng.uniform(-4, 4, n)).astype(np.float32)
            _X[:n], _Y[:n] = x, y
            capi.call("agpu_upload", p._handle, vp(din), C.c_void_p(_X.ctypes.data), 4 * n)
            capi.call("agpu_upload", p._handle, vp(dy), C.c_void_p(_Y.ctypes.data), 4 * n)
            capi.call("agpu_binary", p._handle, capi.OP_POW, capi.F32, vp(din), vp(dy), vp(dout), n)
            capi.call("agpu_download", p._handle, C.c_void_p(_GOT.ctypes.data), vp(dout), 4 * n)
            p.sync()
            step = n // (workers * 4)
            for r in pool.imap_unordered(_check, [("pow", lo, lo + step) for lo in range(0, n, step)]):
                differing += r[2]
                zero_sign += r[3] if len(r) > 3 else 0
                if r[0] > worst:
                    worst = r[0]
                    k = int(np.flatnonzero(_X[:n].view(np.uint32) == r[1])[0])
                    worst_xy = (float(_X[k]), float(_Y[k]))
        out[dom] = {"pairs": 4 * PAIRS, "max_ulp": worst if worst < (1 << 31) else "NaN mismatch", "worst_x_y": [repr(worst_xy[0]), repr(worst_xy[1])],
                    "pairs_not_bit_identical": differing, "zeros_with_the_other_sign": zero_sign}
    out["seconds"] = round(time.time() - t0, 1)
    out["max_ulp"] = max((d["max_ulp"] if isinstance(d["max_ulp"], int) else 1 << 31) for d in out.values() if isinstance(d, dict))
    out["zeros_with_the_other_sign"] = sum(d["zeros_with_the_other_sign"] for d in out.values() if isinstance(d, dict))
    return out


def _run(names, workers, pool):
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "exhaustive")
    din, dout = dev.create_empty_buffer(4 * CHUNK), dev.create_empty_buffer(4 * CHUNK)
    vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
    res = {"what": "all 2^32 f32 bit patterns: agpu_unary on the device vs the CPU oracle (f64 libm rounded once to f32)", "functions": {}}
    for name in names:
        if name in CASTS:
            to, t0, mism, first_bad = CASTS[name], time.time(), 0, None
            width = {capi.U8: 1, capi.I8: 1, capi.U16: 2, capi.I16: 2}.get(to, 4)
            for c in range(16):
                _X.view(np.uint32)[:] = np.arange(c * CHUNK, (c + 1) * CHUNK, dtype=np.uint32)
                capi.call("agpu_upload", p._handle, vp(din), C.c_void_p(_X.ctypes.data), 4 * CHUNK)
                capi.call("agpu_cast", p._handle, capi.F32, to, vp(din), vp(dout), CHUNK)
                capi.call("agpu_download", p._handle, C.c_void_p(_GOT.ctypes.data), vp(dout), width * CHUNK)
                p.sync()
                step = CHUNK // (workers * 4)
                for r in pool.imap_unordered(_check_cast, [(to, lo, lo + step) for lo in range(0, CHUNK, step)]):
                    mism += r[0]
                    if r[0] and first_bad is None:
                        first_bad = r[1]
            res["functions"][name] = {"patterns_not_bit_identical": mism, "first_mismatch_bits": None if first_bad is None else f"{first_bad:#010x}",
                                      "max_ulp": 0 if mism == 0 else 1 << 30, "zeros_with_the_other_sign": 0, "seconds": round(time.time() - t0, 1)}
            print(name, res["functions"][name], flush=True)
            continue
        if name in BIN16:
            t, o = BIN16[name]
            t0, mism, first_bad = time.time(), 0, None
            da, db = dev.create_empty_buffer(2 * CHUNK), dev.create_empty_buffer(2 * CHUNK)
            for c in range(16):
                pair = np.arange(c * CHUNK, (c + 1) * CHUNK, dtype=np.uint32)
                _X.view(np.uint32)[:] = pair
                a16, b16 = (pair >> 16).astype(np.uint16), (pair & 0xFFFF).astype(np.uint16)
                capi.call("agpu_upload", p._handle, vp(da), C.c_void_p(a16.ctypes.data), 2 * CHUNK)
                capi.call("agpu_upload", p._handle, vp(db), C.c_void_p(b16.ctypes.data), 2 * CHUNK)
                capi.call("agpu_binary", p._handle, o, t, vp(da), vp(db), vp(dout), CHUNK)
                capi.call("agpu_download", p._handle, C.c_void_p(_GOT.ctypes.data), vp(dout), 2 * CHUNK)
                p.sync()
                step = CHUNK // (workers * 4)
                for r in pool.imap_unordered(_check_bin16, [((t, o), lo, lo + step) for lo in range(0, CHUNK, step)]):
                    mism += r[0]
                    if r[0] and first_bad is None:
                        first_bad = r[1]
            res["functions"][name] = {"pairs": 1 << 32, "pairs_not_bit_identical": mism, "first_mismatch_pair": None if first_bad is None else f"{first_bad:#010x}",
                                      "max_ulp": 0 if mism == 0 else 1 << 30, "zeros_with_the_other_sign": 0, "seconds": round(time.time() - t0, 1)}
            print(name, res["functions"][name], flush=True)
            del da, db
            continue
        if name == "pow":
            res["functions"]["pow"] = _pow(p, pool, workers, din, dout, dev, vp)
            print("pow", res["functions"]["pow"], flush=True)
            continue
        op = OPS[name]
        t0 = time.time()
        worst, worst_bits, differing, zero_sign = 0, 0, 0, 0
        for c in range(16):
            _X.view(np.uint32)[:] = np.arange(c * CHUNK, (c + 1) * CHUNK, dtype=np.uint32)
            capi.call("agpu_upload", p._handle, vp(din), C.c_void_p(_X.ctypes.data), 4 * CHUNK)
            capi.call("agpu_unary", p._handle, op, capi.F32, vp(din), vp(dout), CHUNK)
            capi.call("agpu_download", p._handle, C.c_void_p(_GOT.ctypes.data), vp(dout), 4 * CHUNK)
            p.sync()
            step = CHUNK // (workers * 4)
            for r in pool.imap_unordered(_check, [(op, lo, lo + step) for lo in range(0, CHUNK, step)]):
                if r[0] > worst:
                    worst, worst_bits = r[0], r[1]
                differing += r[2]
                zero_sign += r[3] if len(r) > 3 else 0
        x = np.array([worst_bits], np.uint32).view(np.float32)[0]
        res["functions"][name] = {"max_ulp": worst if worst < (1 << 31) else "NaN mismatch", "worst_bits": f"{worst_bits:#010x}", "worst_x": repr(float(x)),
                                  "patterns_not_bit_identical": differing, "zeros_with_the_other_sign": zero_sign, "seconds": round(time.time() - t0, 1)}
        print(name, res["functions"][name], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r03_exhaustive_vs_oracle.json"), "w"), indent=1)
    bad = [n for n, r in res["functions"].items() if not isinstance(r["max_ulp"], int) or r["max_ulp"] > 1]
    print("over 1 ULP:", bad or "none")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
