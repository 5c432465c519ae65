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
CHUNK = 1 << 28
_X = _GOT = None  # inherited by the forked workers


def _ordered(v):
    b = v.view(np.int32).astype(np.int64)
    return np.where(b < 0, np.int64(-(2 ** 31)) - b, b)


def _check(args):
    op, lo, hi = args
    x, got = _X[lo:hi], _GOT[lo:hi]
    exp = O.unary(op, O.F32, x)
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
    global _X, _GOT
    from multiprocessing import shared_memory

    names = sys.argv[1:] or list(OPS)
    workers = int(os.environ.get("AGPU_ORACLE_WORKERS", "16"))
    # the two chunk buffers are shared memory and the workers are forked BEFORE this process touches the GPU: no child ever
    # carries HIP state
    shm_x, shm_g = shared_memory.SharedMemory(create=True, size=4 * CHUNK), shared_memory.SharedMemory(create=True, size=4 * CHUNK)
    _X = np.ndarray(CHUNK, np.float32, buffer=shm_x.buf)
    _GOT = np.ndarray(CHUNK, np.float32, buffer=shm_g.buf)
    pool = mp.get_context("fork").Pool(workers)
    try:
        return _run(names, workers, pool)
    finally:
        pool.terminate()
        _X = _GOT = None
        for m in (shm_x, shm_g):
            m.close()
            m.unlink()


def _run(names, workers, pool):
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "exhaustive")
    din, dout = dev.create_empty_buffer(4 * CHUNK), dev.create_empty_buffer(4 * CHUNK)
    vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
    res = {"what": "all 2^32 f32 bit patterns: agpu_unary on the device vs the CPU oracle (f64 libm rounded once to f32)", "functions": {}}
    for name in names:
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
