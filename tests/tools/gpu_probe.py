#!/usr/bin/env python3
"""Quick on-GPU probe (dev tool): raw C-ABI parity vs the oracle on a few kernels + a first timing sweep.

Usage on the GPU box:  python tools/gpu_probe.py [--rows N] [--sweep]
Writes gpurun_out/probe.json.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle as O  # noqa: E402
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402


def vp(buf):
    return C.c_void_p(buf.ptr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 28)
    ap.add_argument("--sweep", action="store_true")
    args = ap.parse_args()
    os.makedirs("gpurun_out", exist_ok=True)
    res = {"parity": {}, "timing": []}

    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "probe")
    print(dev, capi.lib().agpu_build_info().decode(), flush=True)

    # ------------------------------------------------------------ parity on small/ragged sizes
    def dl(buf, dtype, n):
        return dev.retrive_data(buf, pipeline=p)[: n * np.dtype(dtype).itemsize].view(dtype)

    ok_all = True
    for n in [0, 1, 5, 255, 256, 257, 4095, 4096, 4097, 65537, 1_000_003]:
        a = O.synth_f32(n, 1, 0, -1000, 1000)
        b = O.synth_f32(n, 2, 0, -1000, 1000)
        da, db = dev.create_gpu_buffer_with_data(a), dev.create_gpu_buffer_with_data(b)
        do = dev.create_empty_buffer(max(4 * n, 16))
        for op in (capi.OP_ADD, capi.OP_SUB, capi.OP_MUL, capi.OP_DIV):
            capi.call("agpu_binary", p._handle, op, capi.F32, vp(da), vp(db), vp(do), n)
            got = dl(do, np.float32, n)
            exp = O.binary(op, O.F32, a, b)
            same = np.array_equal(got.view(np.uint32), exp.view(np.uint32))
            ok_all &= same
            if not same:
                print("MISMATCH binary f32 op", op, "n", n, flush=True)
        # compare + validity
        ia = O.synth_i32(n, 3, 0, 16)
        ib = O.synth_i32(n, 4, 0, 16)
        va, vb = O.synth_bits(n, 5, 0, 0.9), O.synth_bits(n, 6, 0, 0.9)
        dia, dib = dev.create_gpu_buffer_with_data(ia), dev.create_gpu_buffer_with_data(ib)
        dva, dvb = dev.create_gpu_buffer_with_data(va), dev.create_gpu_buffer_with_data(vb)
        nb = O.bitmap_bytes(n)
        dob, dov = dev.create_empty_buffer(max(nb, 16)), dev.create_empty_buffer(max(nb, 16))
        for variant in (0, 1):
            capi.call("agpu_pipeline_set_tuning", p._handle, b"cmp_variant", variant)
            for op in (capi.CMP_EQ, capi.CMP_LT, capi.CMP_GT):
                capi.call("agpu_memset", p._handle, vp(dob), 0xAA, max(nb, 16))
                capi.call("agpu_compare_validity", p._handle, op, capi.I32, vp(dia), vp(dib), vp(dva), vp(dvb), vp(dob), vp(dov), n)
                got = dev.retrive_data(dob, pipeline=p)[:nb]
                gotv = dev.retrive_data(dov, pipeline=p)[:nb]
                exp = O.compare(op, O.I32, ia, ib)
                expv = O.bitmap_binary(O.OP_AND, va, vb, n)
                same = np.array_equal(got, exp) and np.array_equal(gotv, expv)
                ok_all &= same
                if not same:
                    print("MISMATCH compare variant", variant, "op", op, "n", n, flush=True)
        capi.call("agpu_pipeline_set_tuning", p._handle, b"cmp_variant", 0)
        # f32 sum (reference order) + min/max
        dr = dev.create_empty_buffer(16)
        x = O.synth_f32(n, 7, 0, -1, 1)
        dx = dev.create_gpu_buffer_with_data(x)
        for rop in (capi.RED_SUM, capi.RED_MIN, capi.RED_MAX):
            capi.call("agpu_reduce", p._handle, rop, capi.F32, vp(dx), None, n, vp(dr))
            got = dl(dr, np.float32, 1)[0]
            exp = O.reduce(rop, O.F32, x)
            same = np.float32(got).view(np.uint32) == np.float32(exp).view(np.uint32)
            ok_all &= bool(same)
            if not same:
                print("MISMATCH reduce", rop, "n", n, got, exp, flush=True)
        # cast u8→f32 and fused sin
        u = O.synth_u8(n, 8, 0)
        du = dev.create_gpu_buffer_with_data(u)
        capi.call("agpu_cast", p._handle, capi.U8, capi.F32, vp(du), vp(do), n)
        same = np.array_equal(dl(do, np.float32, n), O.cast(O.U8, O.F32, u))
        ok_all &= same
        if not same:
            print("MISMATCH cast n", n, flush=True)
        capi.call("agpu_unary", p._handle, capi.UN_SIN, capi.U8, vp(du), vp(do), n)
        got = dl(do, np.float32, n)
        exp = O.unary(O.UN_SIN, O.U8, u)
        if n:
            ulp = np.abs(got.view(np.int32).astype(np.int64) - exp.view(np.int32).astype(np.int64)).max()
            res["parity"][f"sin_u8_max_ulp_n{n}"] = int(ulp)
    res["parity"]["all_bit_exact"] = bool(ok_all)
    print("parity all bit exact:", ok_all, flush=True)

    # sin/cos f32 ULP sweep
    xs = np.concatenate([
        np.linspace(-np.pi, np.pi, 1 << 20, dtype=np.float32),
        (np.random.default_rng(1).uniform(-1, 1, 1 << 20) * 2.0 ** np.random.default_rng(2).integers(-20, 16, 1 << 20)).astype(np.float32),
    ])
    dx = dev.create_gpu_buffer_with_data(xs)
    do = dev.create_empty_buffer(xs.nbytes)
    for name, uop, oop in (("sin", capi.UN_SIN, O.UN_SIN), ("cos", capi.UN_COS, O.UN_COS), ("exp", capi.UN_EXP, O.UN_EXP),
                           ("log", capi.UN_LOG, O.UN_LOG), ("sqrt", capi.UN_SQRT, O.UN_SQRT), ("sinh", capi.UN_SINH, O.UN_SINH)):
        capi.call("agpu_unary", p._handle, uop, capi.F32, vp(dx), vp(do), len(xs))
        got = dl(do, np.float32, len(xs))
        exp = O.unary(oop, O.F32, xs)
        fin = np.isfinite(exp) & np.isfinite(got)
        ulp = np.abs(got[fin].view(np.int32).astype(np.int64) - exp[fin].view(np.int32).astype(np.int64))
        res["parity"][f"{name}_f32_max_ulp"] = int(ulp.max())
        res["parity"][f"{name}_f32_nonfinite_mismatch"] = int((np.isnan(exp) != np.isnan(got)).sum())
        print(name, "max ulp", int(ulp.max()), "mean", float(ulp.mean()), flush=True)

    # ------------------------------------------------------------ timing
    n = args.rows
    q = CmpQuery(dev)
    A = dev.create_empty_buffer(4 * n)
    B = dev.create_empty_buffer(4 * n)
    Cc = dev.create_empty_buffer(4 * n)
    capi.call("agpu_synth_f32", p._handle, vp(A), n, 20250418, 0, C.c_float(-1000), C.c_float(1000))
    capi.call("agpu_synth_f32", p._handle, vp(B), n, 20250419, 0, C.c_float(-1000), C.c_float(1000))
    VA = dev.create_empty_buffer(O.bitmap_bytes(n))
    VB = dev.create_empty_buffer(O.bitmap_bytes(n))
    OB = dev.create_empty_buffer(O.bitmap_bytes(n))
    OV = dev.create_empty_buffer(O.bitmap_bytes(n))
    capi.call("agpu_synth_bits", p._handle, vp(VA), n, 11, 0, C.c_double(0.9))
    capi.call("agpu_synth_bits", p._handle, vp(VB), n, 12, 0, C.c_double(0.9))
    R = dev.create_empty_buffer(16)
    p.sync()

    def timeit(label, fn, alg_bytes, iters=10):
        fn()
        p.sync()
        ts = []
        for _ in range(iters):
            q.begin(p)
            fn()
            q.end(p)
            ts.append(q.wait_for_results())
        ms = float(np.median(ts))
        tbs = alg_bytes / ms / 1e9
        row = {"kernel": label, "ms": ms, "TB/s": tbs, "frac_of_8TBs": tbs / 8.0}
        res["timing"].append(row)
        print(row, flush=True)
        return ms

    def add():
        capi.call("agpu_binary", p._handle, capi.OP_ADD, capi.F32, vp(A), vp(B), vp(Cc), n)

    def eqv():
        capi.call("agpu_compare_validity", p._handle, capi.CMP_EQ, capi.I32, vp(A), vp(B), vp(VA), vp(VB), vp(OB), vp(OV), n)

    def eq():
        capi.call("agpu_compare", p._handle, capi.CMP_EQ, capi.I32, vp(A), vp(B), vp(OB), n)

    grids = [0]
    if args.sweep:
        grids = [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536]
    for g in grids:
        capi.call("agpu_pipeline_set_tuning", p._handle, b"stream_grid", g)
        timeit(f"add_f32 grid={g}", add, 12 * n)
        for variant in (0, 1):
            capi.call("agpu_pipeline_set_tuning", p._handle, b"cmp_variant", variant)
            timeit(f"eq_i32+validity v{variant} grid={g}", eqv, 8.5 * n)
        capi.call("agpu_pipeline_set_tuning", p._handle, b"cmp_variant", 0)
    capi.call("agpu_pipeline_set_tuning", p._handle, b"stream_grid", 0)
    timeit("eq_i32 (no validity)", eq, 8.125 * n)
    timeit("add_scalar_f32", lambda: capi.call("agpu_scalar", p._handle, capi.OP_ADD, capi.F32, vp(A), vp(B), vp(Cc), n), 8 * n)
    timeit("sum_f32 (tree order)", lambda: capi.call("agpu_reduce", p._handle, capi.RED_SUM, capi.F32, vp(A), None, n, vp(R)), 4 * n)
    timeit("min_f32", lambda: capi.call("agpu_reduce", p._handle, capi.RED_MIN, capi.F32, vp(A), None, n, vp(R)), 4 * n)
    timeit("sum_f64acc", lambda: capi.call("agpu_reduce_sum_f64", p._handle, vp(A), None, n, vp(R)), 4 * n)
    timeit("sin_f32", lambda: capi.call("agpu_unary", p._handle, capi.UN_SIN, capi.F32, vp(A), vp(Cc), n), 8 * n)
    timeit("cast_u8_f32", lambda: capi.call("agpu_cast", p._handle, capi.U8, capi.F32, vp(B), vp(Cc), n), 5 * n)
    timeit("sin_u8", lambda: capi.call("agpu_unary", p._handle, capi.UN_SIN, capi.U8, vp(B), vp(Cc), n), 5 * n)
    timeit("bitmap_and", lambda: capi.call("agpu_bitmap_binary", p._handle, capi.OP_AND, vp(VA), vp(VB), vp(OV), n), 0.375 * n)
    timeit("d2d copy 4n", lambda: capi.call("agpu_copy", p._handle, vp(Cc), vp(A), 4 * n), 8 * n)

    with open("gpurun_out/probe.json", "w") as f:
        json.dump(res, f, indent=1)
    print("done", flush=True)


if __name__ == "__main__":
    main()
