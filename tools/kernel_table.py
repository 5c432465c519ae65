#!/usr/bin/env python3
"""DEV TOOL: time every kernel family at BASELINE size on the GPU box and print the DESIGN.md §4 table
(algorithmic bytes ÷ median HIP-event time).   python tools/kernel_table.py [--rows 1000000000] [--tag r01]
Writes gpurun_out/kernel_table_<tag>.json (copy into profiles/)."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--iters", type=int, default=9)
    ap.add_argument("--tag", default="r01")
    args = ap.parse_args()
    n = args.rows
    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "table")
    q = CmpQuery(dev)
    h = p._handle
    vp = lambda b, off=0: C.c_void_p(b.ptr + off)  # noqa: E731
    nb = (n + 63) // 64 * 8
    # one table: the buffers come out of one block, placed for the HBM channel hash (agpu_malloc_table)
    # (two tables, like bench.py: the value columns of the arithmetic rows, and the compare rows' columns with their bitmaps)
    A, B, O, M = dev.create_table_buffers([4 * n] * 3 + [nb])
    IA, IB, VA, VB, OB, OV = dev.create_table_buffers([4 * n] * 2 + [nb] * 4)
    R = dev.create_empty_buffer(64)
    S = dev.create_gpu_buffer_with_data(np.array([3.0], np.float32))
    SI = dev.create_gpu_buffer_with_data(np.array([3], np.int32))
    capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-1000), C.c_float(1000))
    capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-1000), C.c_float(1000))
    capi.call("agpu_unary", h, capi.UN_ABS, capi.F32, vp(A), vp(IA), n)  # |a|: the compare rows do not care, log wants a positive column
    capi.call("agpu_copy", h, vp(IB), vp(B), 4 * n)
    for buf, seed in ((VA, 3), (VB, 4), (M, 5)):
        capi.call("agpu_synth_bits", h, vp(buf), n, seed, 0, C.c_double(0.9 if buf is not M else 0.5))
    p.sync()
    rows = []

    def t(label, bytes_per_row, f, note=""):
        # 6 untimed launches first: the VALU-heavy kernels (pow above all) run their first 4–5 launches after a
        # memory-bound or idle phase at 1.3–1.5 GHz before the shader clock has ramped (tools/probe/pow_clock.py)
        # (10 warm-up launches, a sync before the last)
        for _ in range(9):
            f()
        p.sync()
        f()
        p.sync()
        ts = []
        for _ in range(args.iters):
            q.begin(p)
            f()
            q.end(p)
            ts.append(q.wait_for_results())
        ms = float(np.median(ts))
        r = {"kernel": label, "alg_B_per_row": bytes_per_row, "ms": round(ms, 4), "GBps": round(bytes_per_row * n / ms / 1e6, 1),
             "frac_8TBs": round(bytes_per_row * n / ms / 1e6 / 8000, 4), "note": note}
        rows.append(r)
        print(r, flush=True)

    F32, I32, U8 = capi.F32, capi.I32, capi.U8
    for name, op in (("add", capi.OP_ADD), ("sub", capi.OP_SUB), ("mul", capi.OP_MUL), ("div", capi.OP_DIV), ("min", capi.OP_MIN), ("max", capi.OP_MAX)):
        t(f"f32 {name} (array∘array)", 12, lambda op=op: capi.call("agpu_binary", h, op, F32, vp(A), vp(B), vp(O), n))
    t("i32 add (wrapping)", 12, lambda: capi.call("agpu_binary", h, capi.OP_ADD, I32, vp(A), vp(B), vp(O), n))
    t("u32 and", 12, lambda: capi.call("agpu_binary", h, capi.OP_AND, capi.U32, vp(A), vp(B), vp(O), n))
    t("i32 shl (array of u32 amounts)", 12, lambda: capi.call("agpu_binary", h, capi.OP_SHL, I32, vp(A), vp(B), vp(O), n))
    t("u16 shr (array of u32 amounts)", 8, lambda: capi.call("agpu_binary", h, capi.OP_SHR, capi.U16, vp(A), vp(B), vp(O), n))
    t("u8 add (array∘array)", 3, lambda: capi.call("agpu_binary", h, capi.OP_ADD, U8, vp(A), vp(B), vp(O), n))
    t("bool → f32 cast", 4.125, lambda: capi.call("agpu_cast", h, capi.BOOL, F32, vp(VA), vp(O), n))
    t("f32 add_scalar", 8, lambda: capi.call("agpu_scalar", h, capi.OP_ADD, F32, vp(A), vp(S), vp(O), n))
    t("i32 mul_scalar", 8, lambda: capi.call("agpu_scalar", h, capi.OP_MUL, I32, vp(A), vp(SI), vp(O), n))
    t("f32 rem_scalar", 8, lambda: capi.call("agpu_scalar", h, capi.OP_REM, F32, vp(A), vp(S), vp(O), n))
    for name, op in (("neg", capi.UN_NEG), ("abs", capi.UN_ABS), ("sqrt", capi.UN_SQRT), ("sin", capi.UN_SIN), ("cos", capi.UN_COS),
                     ("exp", capi.UN_EXP), ("log", capi.UN_LOG), ("sinh", capi.UN_SINH), ("cbrt", capi.UN_CBRT), ("acos", capi.UN_ACOS)):
        t(f"f32 {name}", 8, lambda op=op: capi.call("agpu_unary", h, op, F32, vp(IA if name == "log" else A), vp(O), n),
          note="positive column" if name == "log" else "")
    t("f32 log, half the column negative (NaN rows)", 8, lambda: capi.call("agpu_unary", h, capi.UN_LOG, F32, vp(A), vp(O), n),
      note="0 / negatives / NaN / inf: three selects in line; only positive denormals take the general form")
    t("f32 power", 12, lambda: capi.call("agpu_binary", h, capi.OP_POW, F32, vp(A), vp(B), vp(O), n))
    for name, op in (("eq", capi.CMP_EQ), ("lt", capi.CMP_LT), ("gt", capi.CMP_GT)):
        t(f"i32 {name} → bitmap + validity AND (fused)", 8.5,
          lambda op=op: capi.call("agpu_compare_validity", h, op, I32, vp(IA), vp(IB), vp(VA), vp(VB), vp(OB), vp(OV), n))
    t("i32 eq → bitmap (no validity)", 8.125, lambda: capi.call("agpu_compare", h, capi.CMP_EQ, I32, vp(IA), vp(IB), vp(OB), n))
    t("f32 lt → bitmap", 8.125, lambda: capi.call("agpu_compare", h, capi.CMP_LT, F32, vp(IA), vp(IB), vp(OB), n))
    t("u8 eq → bitmap (vector variant)", 2.125, lambda: capi.call("agpu_compare", h, capi.CMP_EQ, U8, vp(A), vp(B), vp(OB), n))
    LAT = "launch-latency-bound: 125 MB per bitmap, the whole launch lasts 20–60 us"
    t("validity AND (bitmap)", 0.375, lambda: capi.call("agpu_bitmap_binary", h, capi.OP_AND, vp(VA), vp(VB), vp(OV), n), note=LAT)
    t("bitmap not", 0.25, lambda: capi.call("agpu_bitmap_not", h, vp(VA), vp(OV), n), note=LAT)
    t("popcount (null count)", 0.125, lambda: capi.call("agpu_bitmap_popcount", h, vp(VA), n, vp(R)), note=LAT)
    R2 = dev.create_empty_buffer(64)
    t("validity AND + set-bit count by-product", 0.375,
      lambda: capi.call("agpu_bitmap_binary_count", h, capi.OP_AND, vp(VA), vp(VB), vp(OV), n, vp(R2)), note=LAT + "; count = per-wave v_bcnt + one folding block")
    t("i32 eq → bitmap + validity AND + null count by-product", 8.5,
      lambda: capi.call("agpu_compare_validity_count", h, capi.CMP_EQ, I32, vp(IA), vp(IB), vp(VA), vp(VB), vp(OB), vp(OV), n, vp(R2)),
      note="the validity blocks of the launch count what they store; + one folding block")
    t("merge validity (fused 4-input)", 0.625, lambda: capi.call("agpu_bitmap_merge_validity", h, vp(VA), vp(VB), vp(M), vp(VA), vp(OV), n), note=LAT)
    t("u32 popcount per element (countob)", 8, lambda: capi.call("agpu_unary", h, capi.UN_POPCOUNT, capi.U32, vp(A), vp(O), n))
    t("cast u8→f32", 5, lambda: capi.call("agpu_cast", h, U8, F32, vp(B), vp(O), n))
    t("cast i16→f32", 6, lambda: capi.call("agpu_cast", h, capi.I16, F32, vp(B), vp(O), n))
    t("cast f32→u8", 5, lambda: capi.call("agpu_cast", h, F32, U8, vp(A), vp(O), n))
    t("cast u8→u32", 5, lambda: capi.call("agpu_cast", h, U8, capi.U32, vp(B), vp(O), n))
    t("cast f32→i8 (reference-absent)", 5, lambda: capi.call("agpu_cast", h, F32, capi.I8, vp(A), vp(O), n))
    t("cast f32→i16 (reference-absent)", 6, lambda: capi.call("agpu_cast", h, F32, capi.I16, vp(A), vp(O), n))
    t("cast f32→u16 (reference-absent)", 6, lambda: capi.call("agpu_cast", h, F32, capi.U16, vp(A), vp(O), n))
    t("fused sin_u8 (LDS table)", 5, lambda: capi.call("agpu_unary", h, capi.UN_SIN, U8, vp(B), vp(O), n))
    t("fused cos_i8 (LDS table)", 5, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.I8, vp(B), vp(O), n))

    # round 4: chains with a widening cast at the head (agpu_fused_cast_chain) — what a fusing pipeline issues at finish() for cast_op → …
    class _Step(C.Structure):
        _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]

    def _chain(*items):
        arr = (_Step * len(items))()
        for k_, (op_, kind_, operand_) in enumerate(items):
            arr[k_].op, arr[k_].kind, arr[k_].operand = op_, kind_, (operand_.ptr if operand_ is not None else None)
        return arr, len(items)

    c_sin, n_sin = _chain((capi.UN_SIN, 0, None))
    c_so, n_so = _chain((capi.OP_MUL, 1, S), (capi.OP_ADD, 1, S))
    c_hv, n_hv = _chain((capi.OP_MUL, 1, S), (capi.UN_SIN, 0, None))
    t("cast u8→f32 then sin, ONE launch (config 4 as worded)", 5, lambda: capi.call("agpu_fused_cast_chain", h, U8, vp(B), C.cast(c_sin, C.c_void_p), n_sin, vp(O), n),
      note="the unfused pair moves 13 B/row in two launches")
    t("cast u8→f32 · s + s, ONE launch", 5, lambda: capi.call("agpu_fused_cast_chain", h, U8, vp(B), C.cast(c_so, C.c_void_p), n_so, vp(O), n),
      note="8-bit source: the chain is evaluated once per byte value into a 256-entry table")
    t("cast u8→f32 · s then sin, ONE launch", 5, lambda: capi.call("agpu_fused_cast_chain", h, U8, vp(B), C.cast(c_hv, C.c_void_p), n_hv, vp(O), n),
      note="same table route: a transcendental step costs nothing per row")
    t("cast i16→f32 then sin, ONE launch", 6, lambda: capi.call("agpu_fused_cast_chain", h, capi.I16, vp(B), C.cast(c_sin, C.c_void_p), n_sin, vp(O), n),
      note="16-bit source: sin evaluated per row (VALU-bound); bit-identical to the pair, which the table kernel sin_i16 is not")
    t("cast u16→f32 · s then sin, ONE launch", 6, lambda: capi.call("agpu_fused_cast_chain", h, capi.U16, vp(B), C.cast(c_hv, C.c_void_p), n_hv, vp(O), n),
      note="the generic cast-headed chain kernel (round 6: the chain interpreted once per chunk, sin in packed f32)")
    t("(x · s).sin() f32, ONE launch", 8, lambda: capi.call("agpu_fused_chain", h, F32, vp(A), C.cast(c_hv, C.c_void_p), n_hv, vp(O), n),
      note="agpu_fused_chain with a transcendental step")
    t("fused sin_u16", 6, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U16, vp(B), vp(O), n))
    t("broadcast f32", 4, lambda: capi.call("agpu_broadcast", h, F32, 0x40400000, vp(O), n))
    t("f32 sum (reference tree order)", 4, lambda: capi.call("agpu_reduce", h, capi.RED_SUM, F32, vp(A), None, n, vp(R)))
    t("f32 sum, null-aware", 4.125, lambda: capi.call("agpu_reduce", h, capi.RED_SUM, F32, vp(A), vp(VA), n, vp(R)))
    t("f32 sum f64-accumulated", 4, lambda: capi.call("agpu_reduce_sum_f64", h, vp(A), None, n, vp(R)))
    t("f32 min", 4, lambda: capi.call("agpu_reduce", h, capi.RED_MIN, F32, vp(A), None, n, vp(R)))
    t("f32 max", 4, lambda: capi.call("agpu_reduce", h, capi.RED_MAX, F32, vp(A), None, n, vp(R)))
    t("i32 sum (wrapping)", 4, lambda: capi.call("agpu_reduce", h, capi.RED_SUM, I32, vp(A), None, n, vp(R)))
    t("f32 sum + min + max + f64 sum in ONE pass (agpu_reduce_stats_f32)", 4, lambda: capi.call("agpu_reduce_stats_f32", h, vp(A), None, n, vp(R)),
      note="config 5's four statistics from one read of the column; every field bit-identical to the separate reduction (four launches: 4 x the time of one)")
    t("… null-aware (validity bitmap read beside the values)", 4.125, lambda: capi.call("agpu_reduce_stats_f32", h, vp(A), vp(VA), n, vp(R)))
    t("f32 min, null-aware", 4.125, lambda: capi.call("agpu_reduce", h, capi.RED_MIN, F32, vp(A), vp(VA), n, vp(R)),
      note="round 6b: the one-wave-per-chunk kernel reads the bits beside the values (rounds 1-6a: the grid-stride form)")
    t("merge f32 by mask", 12.125, lambda: capi.call("agpu_merge", h, 4, vp(A), vp(B), vp(M), vp(O), n))
    # gather / scatter: sequential indices (best case) and a pseudo-random permutation (worst case)
    m = min(n, 1 << 28)
    IDX = dev.create_empty_buffer(4 * m)
    capi.call("agpu_synth_i32", h, vp(IDX), m, 9, 0, m)  # uniform random indices in [0, m)
    p.set_tuning("gather_bucket", 1)
    t("take f32, random idx, direct (2^28 rows)", 12 * m / n, lambda: capi.call("agpu_take", h, 4, vp(A), m, vp(IDX), vp(O), m),
      note="tuning gather_bucket=1; bytes/row quoted per 1e9-row column; rate = 12 B × 2^28 rows ÷ time")
    IDX2 = dev.create_empty_buffer(4 * m)
    capi.call("agpu_synth_i32", h, vp(IDX2), m, 10, 0, m)

    def fix(bytes_per_row):
        rows[-1]["GBps"] = round(bytes_per_row * m / rows[-1]["ms"] / 1e6, 1)
        rows[-1]["frac_8TBs"] = round(rows[-1]["GBps"] / 8000, 4)
        rows[-1]["G_rows_per_s"] = round(m / rows[-1]["ms"] / 1e6, 1)

    fix(12)
    p.set_tuning("gather_bucket", 0)
    t("take f32, random idx, default = merge-back pipeline (2^28 rows)", 12 * m / n, lambda: capi.call("agpu_take", h, 4, vp(A), m, vp(IDX), vp(O), m),
      note="auto policy picks the bucketed form at this size; round 3: ranks recorded, runs merged back (tk2_* kernels)")
    fix(12)
    OV = dev.create_empty_buffer(m // 8 + 64)
    p.set_tuning("gather_bucket", 0)
    t("take f32 WITH validity, random idx, default = one merge-back pipeline (2^28 rows)", 12.25 * m / n,
      lambda: capi.call("agpu_take_validity", h, 4, vp(A), m, vp(VA), vp(IDX), vp(O), vp(OV), m),
      note="agpu_take_validity: the validity bit travels with the value (the reference: take + bool take, two dispatches)")
    fix(12.25)
    p.set_tuning("gather_bucket", 1)
    t("take f32 WITH validity, direct = take + take_bits (2^28 rows)", 12.25 * m / n,
      lambda: capi.call("agpu_take_validity", h, 4, vp(A), m, vp(VA), vp(IDX), vp(O), vp(OV), m), note="tuning gather_bucket=1")
    fix(12.25)
    p.set_tuning("gather_bucket", 1)
    t("put f32, random src and dst idx, direct (2^28 rows)", 16 * m / n,
      lambda: capi.call("agpu_put_bounded", h, 4, vp(A), m, vp(IDX), vp(O), m, vp(IDX2), m), note="tuning gather_bucket=1")
    fix(16)
    p.set_tuning("gather_bucket", 0)
    t("put f32, random src and dst idx, default = bucketed (2^28 rows)", 16 * m / n,
      lambda: capi.call("agpu_put_bounded", h, 4, vp(A), m, vp(IDX), vp(O), m, vp(IDX2), m), note="auto policy picks the bucketed form at this size")
    fix(16)
    t("take_bits (Boolean take), random idx, default = merge-back pipeline (2^28 rows)", 4.25 * m / n,
      lambda: capi.call("agpu_take_bits", h, vp(VA), m, vp(IDX), vp(OB), m), note="round 3: the bitmap's words are the pipeline's elements")
    fix(4.25)
    p.set_tuning("gather_bucket", 1)
    t("take_bits (Boolean take), random idx, direct (2^28 rows)", 4.25 * m / n,
      lambda: capi.call("agpu_take_bits", h, vp(VA), m, vp(IDX), vp(OB), m), note="tuning gather_bucket=1")
    fix(4.25)
    p.set_tuning("gather_bucket", 0)
    t("put_bits (Boolean put), random src and dst idx, default = bucketed by destination region (2^28 rows)", 8.25 * m / n,
      lambda: capi.call("agpu_put_bits_bounded", h, vp(VA), m, vp(IDX), vp(OB), m, vp(IDX2), m),
      note="round 3: Boolean take emitting entries + partition by 32 KiB bitmap regions + LDS apply; no global atomics")
    fix(8.25)
    p.set_tuning("gather_bucket", 1)
    t("put_bits (Boolean put), random src and dst idx, direct (2^28 rows)", 8.25 * m / n,
      lambda: capi.call("agpu_put_bits_bounded", h, vp(VA), m, vp(IDX), vp(OB), m, vp(IDX2), m), note="tuning gather_bucket=1: one device-scope atomic per row")
    fix(8.25)
    p.set_tuning("gather_bucket", 0)
    t("clone_buffer (agpu_copy, 4 GB)", 8, lambda: capi.call("agpu_copy", h, vp(O), vp(A), 4 * n))

    # the reference's own criterion shapes, kernel time only (its benches: crates/benchmarks/benches/compare_gpu_arrow.rs
    # :18-43 f32 column + scalar at 10 Mi rows; compare_sum.rs:17-40 u32 sum at 1 Mi / 10 Mi rows).  The host-API and
    # CPU-port numbers for the same shapes are in bench.py's JSON line (extra.reference_bench_workloads_gpu,
    # cpu_baseline.reference_bench_workloads).
    ref_rows = []
    for label, rows_, bpr, f in (
            ("ref bench: f32 add_scalar, 10 Mi rows", 10 << 20, 8, lambda: capi.call("agpu_scalar", h, capi.OP_ADD, F32, vp(A), vp(S), vp(O), 10 << 20)),
            ("ref bench: u32 sum, 1 Mi rows", 1 << 20, 4, lambda: capi.call("agpu_reduce", h, capi.RED_SUM, capi.U32, vp(A), None, 1 << 20, vp(R))),
            ("ref bench: u32 sum, 10 Mi rows", 10 << 20, 4, lambda: capi.call("agpu_reduce", h, capi.RED_SUM, capi.U32, vp(A), None, 10 << 20, vp(R)))):
        # (10 warm-up launches, a sync before the last)
        for _ in range(9):
            f()
        p.sync()
        f()
        p.sync()
        ts = []
        for _ in range(15):
            q.begin(p)
            f()
            q.end(p)
            ts.append(q.wait_for_results())
        ms = float(np.median(ts))
        ref_rows.append({"kernel": label, "rows": rows_, "us": round(ms * 1e3, 2), "GBps": round(bpr * rows_ / ms / 1e6, 1),
                         "note": "launch-latency-bound (4–84 MB of traffic)"})
        print(ref_rows[-1], flush=True)

    # host link (what the boundary costs when it is handed HOST buffers): pinned staging, 1 GiB each way
    from arrow_gpu_amd.interop import PinnedStaging

    gib = 1 << 30
    st = PinnedStaging(dev, gib)
    st.view[:] = 1
    pcie = {}
    for label, f in (("H2D pinned 1 GiB", lambda: st.upload(p, O, gib)), ("D2H pinned 1 GiB", lambda: st.download(p, O, gib))):
        f()
        p.sync()
        ts = []
        for _ in range(5):
            q.begin(p)
            f()
            q.end(p)
            ts.append(q.wait_for_results())
        ms = float(np.median(ts))
        pcie[label] = {"ms": round(ms, 3), "GBps": round(gib / ms / 1e6, 1)}
        print(label, pcie[label], flush=True)

    # config 1 (1 048 576-row i32 columns, 12 MiB of traffic per op): launch-bound regime.  200 eager launches back to
    # back, and the same 200 ops replayed from one hipGraph (agpu_pipeline_begin_capture / agpu_graph_launch).
    small = 1 << 20
    reps = 200

    def burst():
        for _ in range(reps):
            capi.call("agpu_binary", h, capi.OP_ADD, I32, vp(A), vp(B), vp(O), small)

    def timed(f, k=9):
        f()
        p.sync()
        ts = []
        for _ in range(k):
            q.begin(p)
            f()
            q.end(p)
            ts.append(q.wait_for_results())
        return float(np.median(ts))

    import time as _time

    eager_ms = timed(burst)
    t0 = _time.perf_counter()
    burst()
    host_ms = (_time.perf_counter() - t0) * 1e3  # host-side cost of issuing the launches through ctypes
    p.sync()
    capi.call("agpu_pipeline_begin_capture", h)
    burst()
    g = C.c_void_p()
    capi.call("agpu_pipeline_end_capture", h, C.byref(g))
    graph_ms = timed(lambda: capi.call("agpu_graph_launch", g, h))
    capi.call("agpu_graph_destroy", g)
    small_n = {"rows": small, "ops": reps, "eager_us_per_op_gpu": round(eager_ms / reps * 1e3, 3),
               "eager_us_per_op_host_ctypes": round(host_ms / reps * 1e3, 3), "graph_us_per_op_gpu": round(graph_ms / reps * 1e3, 3),
               "hbm_floor_us_per_op": round(12 * small / 8e12 * 1e6, 3)}
    print("config-1 size (launch-bound):", small_n, flush=True)

    outdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, f"kernel_table_{args.tag}.json"), "w") as f:
        json.dump({"rows": n, "device": dev.name, "kernels": rows, "reference_bench_shapes": ref_rows, "host_link": pcie,
                   "small_n": small_n}, f, indent=1)
    print("\n| kernel | alg. B/row | ms @1e9 | GB/s | frac of 8 TB/s |\n|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['kernel']} | {r['alg_B_per_row']:.4g} | {r['ms']} | {r['GBps']} | {r['frac_8TBs']} |")


if __name__ == "__main__":
    main()
