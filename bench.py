#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic 1-billion-row columns resident in HBM:
    1. f32 add      : Float32ArrayGPU + Float32ArrayGPU, 1e9 rows, no nulls              → 12.0 GB algorithmic
    2. i32 eq       : Int32ArrayGPU == Int32ArrayGPU → packed bitmap, fused validity AND,
                      1e9 rows, 10 % nulls on each side                                   →  8.5 GB algorithmic
(BASELINE.json configs[1] and configs[2]; both are what `metric` is quoted on.)  value = algorithmic GB moved by all
ranks per second.  Inputs are generated on the device (counter-based hash, include/arrow_gpu.h agpu_synth_*), outputs
are pre-allocated; nothing but the two kernel launches per step is inside the timed region.

N > 1 (launched by torch.distributed.run, one process per GPU): the workers do NOT import torch — rank / world /
local rank come from the launcher's environment, the RCCL id travels through a file rendezvous with a deadline
(arrow_gpu_amd/sharding.py), barriers and the timing reduction go through the C ABI's communicator (agpu_comm_*), so
the whole process runs on ONE HIP + RCCL runtime (the one libarrow_gpu_hip.so links, printed in extra.runtime).
  --scaling weak   (default) every rank owns its own 1e9-row shard of an N×1e9-row column (row0 = rank·1e9)
  --scaling strong the 1e9-row column is cut into N contiguous shards (sharding.shard_rows: 125 M rows per GPU at N = 8)
Chunk-sharded, no data-path collective.  A weak run also times a short strong-scaling leg on the resident shards
(extra.strong_scaling), so that one launch per N yields both curves.  RCCL is used only for the barrier / timing
reduction and, after the timed region, for the final reduce of the per-shard sum/min/max (config 5), under "extra".
A rank that fails (or never arrives) makes every rank exit non-zero within the rendezvous deadline instead of hanging.

Also printed in the same JSON line:
  roofline     — the dominant kernel (f32 add), algorithmic bytes per launch ÷ its mean duration measured with HIP
                 events on the launch stream over the timed region, against 8 TB/s HBM3E; `traffic` = HBM bytes per launch
                 MEASURED in this run (N = 1): two child `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE) over a
                 short run of the same workload after the timed region (≈ 6 s; falls back to profiles/hbm_traffic.json
                 and says so when rocprofv3 is not usable).
                 frac_per_rank = the slowest and the fastest rank's launch of the same kernel.
  cpu_baseline — the CPU port (oracle/cpu_baseline.c, arrow-rs-style single pass) timed on this box's host cores on
                 a bounded sample (rank 0, at every world size), a pyarrow (Arrow C++) sanity line, and `gpu_parity`.
  gpu_parity   — EVERY rank downloads three 65 536-row windows of its shard of the benchmarked outputs and checks them bit-exact
                 against the oracle at its own row0; the verdicts are summed over the ranks through the communicator.  Likewise
                 extra.reduce_sum_min_max.verified: what the collectives left on every rank against the oracle's rank-ordered
                 combine of the gathered per-rank local statistics.  A line of any world size proves its own numbers.
  config.host_api — the fractions of the roof of the same two kernels as an ordinary caller of the host API gets them (inputs allocated one
                 by one, the op allocates its output): extra.layout_pool.
The oracle is imported only by these checks and the CPU leg, AFTER the timed region; the measured path never touches it.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Multi-process GPU work on hosts whose driver only offers dmabuf IPC needs this BEFORE the HSA runtime starts (RCCL's
# hipIpcGetMemHandle fails with "invalid argument" otherwise); the launcher normally exports it — only a default here.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROWS = 1_000_000_000
SEED = 20250418
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ADD_BYTES_PER_ROW = 12.0
EQ_BYTES_PER_ROW = 8.5  # 8 data + 0.125 result bits + 0.375 validity in/out


def _ulp_distance(x, y):
    """ULPs between two f32 arrays (±0 equal, NaN only against NaN)."""
    import numpy as np

    def key(v):
        b = v.view(np.int32).astype(np.int64)
        return np.where(b < 0, -(b & 0x7FFFFFFF), b)

    d = np.abs(key(x) - key(y))
    nan = np.isnan(x) | np.isnan(y)
    d[nan] = np.where(np.isnan(x[nan]) & np.isnan(y[nan]), 0, 1 << 40)
    return int(d.max()) if d.size else 0


def check_config_windows(cfg_windows):
    """Oracle check of one 65 536-row window per extra.configs kernel (same synthetic columns, re-generated on the CPU).
    Returns {name: "bit-exact" | "<= 1 ULP" | "MISMATCH…"}.  Only called from the cpu_baseline leg."""
    import numpy as np

    import oracle as O

    out = {}
    for name, w in cfg_windows.items():
        if name.startswith("swz_"):  # take-shaped results: row j of the window holds the value / bit at idx[j]
            m, w0, rows = w["m"], w["w0"], w["rows"]
            idx = O.synth_i32(rows, SEED + 8, w0, m).view(np.uint32)
            ok = True
            if "got" in w:
                exp = np.array([O.synth_i32(1, w.get("values_seed", SEED + 7), int(i), 0)[0] for i in idx], np.int32).view(np.uint32)
                ok &= bool(np.array_equal(np.asarray(w["got"]).view(np.uint32), exp))
            if "bits" in w:
                eb = np.array([O.synth_bits(1, SEED + 10, int(i), 0.5)[0] & 1 for i in idx], np.uint8)
                ok &= bool(np.array_equal(np.unpackbits(np.asarray(w["bits"]), bitorder="little")[:rows], eb))
            out[name] = "bit-exact" if ok else "MISMATCH"
            continue
        cnt, r0, got = w["rows"], w["row"], w["got"]
        fa = lambda: O.synth_f32(cnt, SEED, r0, -1000.0, 1000.0)  # noqa: E731
        fb = lambda: O.synth_f32(cnt, SEED + 1, r0, -1000.0, 1000.0)  # noqa: E731
        ia = lambda: O.synth_i32(cnt, SEED + 2, r0, 1024)  # noqa: E731
        ib = lambda: O.synth_i32(cnt, SEED + 3, r0, 1024)  # noqa: E731
        u8 = lambda: O.synth_u8(cnt, SEED + 6, r0)  # noqa: E731
        ulp = None
        if name in ("sub_f32", "mul_f32", "div_f32"):
            exp = O.binary({"sub_f32": O.OP_SUB, "mul_f32": O.OP_MUL, "div_f32": O.OP_DIV}[name], O.F32, fa(), fb())
        elif name == "add_scalar_f32":
            exp = O.scalar(O.OP_ADD, O.F32, fa(), np.array([100.0], np.float32))
        elif name in ("lt_i32_validity", "gt_i32_validity"):
            bits = O.compare(O.CMP_LT if name.startswith("lt") else O.CMP_GT, O.I32, ia(), ib())
            vd = O.bitmap_binary(O.OP_AND, O.synth_bits(cnt, SEED + 4, r0, 0.9), O.synth_bits(cnt, SEED + 5, r0, 0.9), cnt)
            exp = np.concatenate([bits[: cnt // 8], vd[: cnt // 8]])
        elif name == "eq_i32_unfused":
            exp = O.compare(O.CMP_EQ, O.I32, ia(), ib())[: cnt // 8]
        elif name == "validity_and":
            exp = O.bitmap_binary(O.OP_AND, O.synth_bits(cnt, SEED + 4, r0, 0.9), O.synth_bits(cnt, SEED + 5, r0, 0.9), cnt)[: cnt // 8]
        elif name == "cast_u8_f32":
            exp = O.cast(O.U8, O.F32, u8())
        elif name in ("sin_f32", "cos_f32"):
            exp = O.unary(O.UN_SIN if name == "sin_f32" else O.UN_COS, O.F32, O.cast(O.U8, O.F32, u8()))
            ulp = 1
        elif name in ("sin_u8", "cos_u8", "cast_u8_f32_then_sin_one_launch", "cast_u8_f32_then_cos_one_launch"):
            exp = O.unary(O.UN_COS if "cos" in name else O.UN_SIN, O.U8, u8())
            ulp = 1
        elif name == "fused_add_scalar_then_mul_scalar":
            exp = O.scalar(O.OP_MUL, O.F32, O.scalar(O.OP_ADD, O.F32, fa(), np.array([100.0], np.float32)), np.array([0.37], np.float32))
        else:
            out[name] = "unchecked"
            continue
        if ulp is None:
            ok = np.array_equal(np.asarray(got).view(np.uint8), np.asarray(exp).view(np.uint8)[: np.asarray(got).nbytes])
            out[name] = "bit-exact" if ok else "MISMATCH"
        else:
            d = _ulp_distance(np.asarray(got, np.float32), np.asarray(exp, np.float32))
            out[name] = f"<= {ulp} ULP vs f64 libm rounded once (max {d})" if d <= ulp else f"MISMATCH (max {d} ULP)"
    return out


def check_headline_windows(gpu_windows) -> bool:
    """Oracle check of the windows a rank downloaded from ITS shard of the benchmarked outputs (first / middle / last 65 536 rows, at the
    shard's own row0): f32 add bit-exact, eq bitmap and merged validity bit-exact.  Every rank runs this on its own windows, after the
    timed region (the oracle is only ever the checker)."""
    import numpy as np

    import oracle as O

    ok = True
    for win in gpu_windows:
        cnt, r0 = win["rows"], win["row"]
        exp = O.binary(O.OP_ADD, O.F32, O.synth_f32(cnt, SEED, r0, -1000.0, 1000.0), O.synth_f32(cnt, SEED + 1, r0, -1000.0, 1000.0))
        ok &= bool(np.array_equal(win["add"].view(np.uint32), exp.view(np.uint32)))
        eb = O.compare(O.CMP_EQ, O.I32, O.synth_i32(cnt, SEED + 2, r0, 1024), O.synth_i32(cnt, SEED + 3, r0, 1024))
        evd = O.bitmap_binary(O.OP_AND, O.synth_bits(cnt, SEED + 4, r0, 0.9), O.synth_bits(cnt, SEED + 5, r0, 0.9), cnt)
        full = cnt // 8
        ok &= bool(np.array_equal(win["eq_bits"][:full], eb[:full]) and np.array_equal(win["eq_validity"][:full], evd[:full]))
    return ok


def slot_vector(rank: int, world: int, values) -> list:
    """This rank's contribution to a gather-by-SUM of `len(values)` numbers per rank: a (len(values) × world) vector that is zero except
    for this rank's slots.  Summed over the ranks (any all-reduce SUM of f64) it is the gathered table; integers below 2^53 survive exactly."""
    out = [0.0] * (len(values) * world)
    for k, v in enumerate(values):
        out[k * world + rank] = float(v)
    return out


def reduce_records_from_gathered(gathered, world: int):
    """Inverse of slot_vector for the six numbers a rank contributes to the final-reduce check: the bit patterns of its local f32 sum /
    min / max, the two halves of its local f64 sum, its row count.  → per statistic a list of (value, n_local) in rank order."""
    import numpy as np

    def col(k):
        return [int(round(gathered[k * world + r])) for r in range(world)]

    s, mn, mx, lo, hi, n = (col(k) for k in range(6))
    f32 = lambda b: np.array([b], np.uint32).view(np.float32)[0]  # noqa: E731
    f64 = lambda l, h: np.array([(h << 32) | l], np.uint64).view(np.float64)[0]  # noqa: E731
    return {"sum": [(f32(s[r]), n[r]) for r in range(world)], "min": [(f32(mn[r]), n[r]) for r in range(world)],
            "max": [(f32(mx[r]), n[r]) for r in range(world)], "sum_f64": [(f64(lo[r], hi[r]), n[r]) for r in range(world)]}


def verify_final_reduce(records, got) -> dict:
    """The collective's results (`got`: sum / min / max as f32, sum_f64) against the ORACLE's rank-ordered combine of the per-rank local
    records (oracle.combine_records: the reference tree's next level for the f32 Sum, Arrow's NaN rule for min / max, empty shards
    skipped; the f64 sum added up in rank order).  Bit-exact or not."""
    import numpy as np

    import oracle as O

    exp = {"sum": O.combine_records(O.RED_SUM, O.F32, records["sum"]), "min": O.combine_records(O.RED_MIN, O.F32, records["min"]),
           "max": O.combine_records(O.RED_MAX, O.F32, records["max"])}
    acc = 0.0
    for v, n_r in records["sum_f64"]:
        if n_r:
            acc = acc + float(v)
    ok = {k: bool(np.float32(exp[k]).view(np.uint32) == np.float32(got[k]).view(np.uint32)) for k in ("sum", "min", "max")}
    ok["sum_f64"] = bool(np.float64(acc).view(np.uint64) == np.float64(got["sum_f64"]).view(np.uint64))
    return ok


def cpu_baseline(sample_rows: int, gpu_parity=None, cfg_windows=None):
    """The CPU leg: time the CPU port on a bounded sample of the same workload; `gpu_parity` = the verdict of the oracle check of every
    rank's output windows (check_headline_windows, AND over ranks), carried into the record.  Test/bench infrastructure only (oracle/)."""
    import numpy as np

    import oracle as O

    parity = gpu_parity

    build_dir = os.path.join(ROOT, "oracle", "_build")
    os.makedirs(build_dir, exist_ok=True)
    so = os.path.join(build_dir, "libcpu_baseline_native.so")
    src = os.path.join(ROOT, "oracle", "cpu_baseline.c")
    try:  # build for THIS host's ISA (the prebuilt copy was compiled on another machine)
        subprocess.run(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off", "-fopenmp", "-shared",
                        "-o", so, src, "-lm"], check=True, capture_output=True, timeout=120)
    except Exception:
        so = os.path.join(build_dir, "libcpu_baseline.so")
    lib = C.CDLL(so)
    n = sample_rows
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2)
    phys = max(1, usable // 2)  # SMT siblings share a core's load/store units: one thread per core
    cores = min(phys, 64)  # measured on the pool's boxes: beyond 64 threads the container's CPU quota throttles the team
    p = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731

    def placed(src):
        """Copy of `src` whose pages were first touched by the all-core partition (NUMA-local slices on a 2-socket host)."""
        dst = np.empty_like(src)
        per64 = 64 * src.dtype.itemsize if src.dtype != np.uint8 else 8
        lib.base_first_touch(p(dst), C.c_uint64(dst.nbytes), C.c_uint64(per64), cores)
        dst[...] = src
        return dst

    # two copies of the sample: the plain one (first-touched by this thread: what a single-threaded arrow-rs kernel would
    # read) for the 1-thread line, the placed one for the all-core line
    plain = [O.synth_f32(n, SEED, 0, -1000.0, 1000.0), O.synth_f32(n, SEED + 1, 0, -1000.0, 1000.0), np.zeros(n, np.float32),
             O.synth_i32(n, SEED + 2, 0, 1024), O.synth_i32(n, SEED + 3, 0, 1024), O.synth_bits(n, SEED + 4, 0, 0.9),
             O.synth_bits(n, SEED + 5, 0, 0.9), np.zeros(O.bitmap_bytes(n), np.uint8), np.zeros(O.bitmap_bytes(n), np.uint8)]
    spread = [placed(x) for x in plain]
    a, b, out, ia, ib, va, vb, ob, ov = plain

    def one_pass(threads):
        a_, b_, out_, ia_, ib_, va_, vb_, ob_, ov_ = plain if threads == 1 else spread
        lib.base_add_f32(p(a_), p(b_), p(out_), None, None, None, C.c_uint64(n), threads)
        lib.base_eq_i32(p(ia_), p(ib_), p(ob_), p(va_), p(vb_), p(ov_), C.c_uint64(n), threads)

    def rate(threads, budget_s):
        one_pass(threads)
        t0 = time.perf_counter()
        k = 0
        while True:
            one_pass(threads)
            k += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or k >= 50:
                break
        return (ADD_BYTES_PER_ROW + EQ_BYTES_PER_ROW) * n * k / dt / 1e9

    # spot-check the port against the oracle before timing it
    m = min(n, 1 << 16)
    one_pass(1)
    assert np.array_equal(out[:m], O.binary(O.OP_ADD, O.F32, a[:m], b[:m]))
    assert np.array_equal(ob[: m // 8], O.compare(O.CMP_EQ, O.I32, ia[:m], ib[:m])[: m // 8])
    v1 = rate(1, 8.0)
    vall = rate(cores, 5.0)
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    res = {"value": round(v1, 3), "unit": "GB/s", "cores": 1, "kind": "port",
           "sample": f"{n} rows of the same synthetic columns (f32 add + i32 eq with validity), repeated passes, "
                     f"single thread like arrow-rs's kernels",
           "all_cores": {"value": round(vall, 3), "cores": cores,
                         "what": "OpenMP static row partition over copies of the columns that were first-touched by the same "
                                 "partition (each thread's slice on its own NUMA node); capped at 64 threads — the container's "
                                 "CPU quota throttles larger teams; DRAM-bound, a reported context line"},
           "cpu_model": model, "nproc": os.cpu_count(), "cgroup_cpu_max": _cgroup_cpu_max(),
           "gpu_parity": parity}
    if cfg_windows:
        try:
            res["configs_parity"] = check_config_windows(cfg_windows)
        except Exception as e:  # noqa: BLE001 — reported, the headline stands
            res["configs_parity"] = {"error": repr(e)}
    # the reference's own criterion workloads, CPU side (same port library): f32 column + scalar at 10 Mi rows, u32 sum
    # at 1 Mi / 10 Mi rows [crates/benchmarks/benches/compare_gpu_arrow.rs:18-43, compare_sum.rs:17-40]
    try:
        lib.base_sum_u32.restype = C.c_uint32
        cnt = 10 * 1024 * 1024
        col = np.arange(cnt, dtype=np.float32)
        dst = np.empty(cnt, np.float32)
        u = np.full(cnt, 2, np.uint32)

        def best_ms(fn, k=7):
            fn()
            ts = []
            for _ in range(k):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            return round(min(ts) * 1e3, 4)

        res["reference_bench_workloads"] = {
            "f32_add_scalar_10Mi_ms": best_ms(lambda: lib.base_add_scalar_f32(p(col), C.c_float(100.0), p(dst), C.c_uint64(cnt))),
            "u32_sum_1Mi_ms": best_ms(lambda: lib.base_sum_u32(p(u), C.c_uint64(1024 * 1024))),
            "u32_sum_10Mi_ms": best_ms(lambda: lib.base_sum_u32(p(u), C.c_uint64(cnt))),
            "what": "CPU port, 1 thread, pre-allocated output, best of 7 (arrow-rs itself allocates the result each call)"}
        assert dst[12345] == np.float32(12345.0) + np.float32(100.0) and lib.base_sum_u32(p(u), C.c_uint64(cnt)) == 2 * cnt
    except Exception as e:  # noqa: BLE001 — context only
        res["reference_bench_workloads"] = {"error": repr(e)}
    try:  # third-party sanity line: Arrow C++ through pyarrow on the same columns, same byte accounting
        import pyarrow as pa
        import pyarrow.compute as pc

        pa.set_cpu_count(1)
        fa, fb = pa.array(a), pa.array(b)
        xa = pa.Array.from_buffers(pa.int32(), n, [pa.py_buffer(va), pa.py_buffer(ia)])
        xb = pa.Array.from_buffers(pa.int32(), n, [pa.py_buffer(vb), pa.py_buffer(ib)])
        pc.add(fa, fb), pc.equal(xa, xb)
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 3.0 and k < 20:
            pc.add(fa, fb)
            pc.equal(xa, xb)
            k += 1
        res["pyarrow"] = {"value": round((ADD_BYTES_PER_ROW + EQ_BYTES_PER_ROW) * n * k / (time.perf_counter() - t0) / 1e9, 3),
                          "version": pa.__version__, "cores": 1, "what": "pc.add(f32) + pc.equal(i32 with nulls)"}
    except Exception as e:  # noqa: BLE001 — context only
        res["pyarrow"] = {"value": None, "what": f"unavailable: {type(e).__name__}"}
    return res


def measure_traffic_live(timeout_s: float = 90.0):
    """HBM bytes per launch of the dominant kernel (f32 add), MEASURED for this run: two child processes — rocprofv3 --pmc
    FETCH_SIZE, then --pmc WRITE_SIZE (separate passes, never combined with tracing) — over a short bench.py of the same
    workload, collected and corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes (KiB units; the gfx950 read side
    counts 128-byte requests at 64 bytes: × 2).  Children, not exec: this process has initialised the GPU.  Returns
    (bytes_per_launch | None, note)."""
    import csv
    import glob
    import shutil
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    per = {}
    t0 = time.time()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="agpu_pmc_", dir="/tmp")
        try:
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-configs", "--no-traffic"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited {r.returncode}: {r.stderr[-200:]}"
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    k = row.get("Kernel_Name", "")
                    if row.get("Counter_Name") == counter and "ew_kernel<float, OpAdd, 1" in k:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"no {counter} rows for the add kernel"
            per[counter] = sorted(vals)[len(vals) // 2]
        except Exception as e:  # noqa: BLE001 — the line falls back to the stored figure
            return None, f"{type(e).__name__}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    total = per["FETCH_SIZE"] * 1024 * 2 + per["WRITE_SIZE"] * 1024
    return total, (f"measured in this run: median per launch over two child rocprofv3 passes of the same workload (--pmc FETCH_SIZE, then --pmc "
                   f"WRITE_SIZE; KiB x 1024, read side x 2 per the guide's gfx950 correction), {time.time() - t0:.0f} s")


def _cgroup_cpu_max():
    """The container's CPU quota ("<quota_us> <period_us>" = quota/period CPUs' worth of time, or "max"): it, not nproc,
    bounds what a thread team can use."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            return f.read().strip()
    except OSError:
        return None


def _claim_stdout():
    """RCCL prints a version banner to the C-level stdout when a communicator comes up; the contract is ONE JSON line
    on stdout.  Point fd 1 at stderr for the whole run and hand back the real stdout for the final line."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def _self_launch(n: int, real_stdout) -> int:
    """`python bench.py --gpus N` without torchrun: N children, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set like torchrun sets
    them (the rendezvous path is derived from MASTER_PORT + this pid, sharding.rendezvous_path_from_env)."""
    import socket

    with socket.socket() as sk:  # a free port only names the rendezvous: nothing listens on it
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TORCHELASTIC_RUN_ID=f"bench_self_{os.getpid()}")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = procs[0].communicate()[0]
    rcs = [procs[0].returncode] + [q.wait() for q in procs[1:]]
    if any(rcs):
        print(f"bench.py: worker exit codes {rcs}", file=sys.stderr)
        return 1
    real_stdout.write(out.decode())
    real_stdout.flush()
    return 0


def main():
    real_stdout = _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=ROWS, help="rows per GPU shard (weak) / rows in total (strong); default 1e9 = the BASELINE config")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank owns --rows rows; strong: --rows rows in total, cut into one contiguous shard per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip extra.configs / extra.layout_pool / extra.strong_scaling")
    ap.add_argument("--cpu-sample-rows", type=int, default=1 << 26)
    ap.add_argument("--rendezvous-timeout", type=float, default=120.0)
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic with child rocprofv3 --pmc passes (use profiles/hbm_traffic.json)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="agpu_set_tuning before anything is allocated (A/B runs, e.g. --tune pool_arena=0); recorded in config.tuning")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "OMPI_COMM_WORLD_RANK" not in os.environ and "SLURM_PROCID" not in os.environ:
        # started plainly with --gpus N (no launcher): be our own launcher.  This process has not touched the GPU and never will:
        # it starts one worker per GPU as CHILD processes (never an exec) with the launcher's environment and relays rank 0's line.
        sys.exit(_self_launch(args.gpus, real_stdout))

    # NOTHING of torch is imported by a worker: the process runs on the HIP + RCCL that libarrow_gpu_hip.so links
    # (/opt/rocm), not on torch's bundled copies — one runtime.
    from arrow_gpu_amd import _capi as capi
    from arrow_gpu_amd import sharding
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    for kv in args.tune:
        k, v = kv.split("=", 1)
        capi.call("agpu_set_tuning", k.encode(), int(v))
    rank, world, local_rank = sharding.ranks_from_env()
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): the world check below will refuse the run "
              f"(launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)",
              file=sys.stderr)
    if os.environ.get("AGPU_BENCH_DEVICE_OVERRIDE") is not None:
        # rehearsal on a 1-GPU box only (tools/archive/r03_bootstrap_rehearsal.sh): every rank names the same GPU, so RCCL's bootstrap between
        # the processes runs for real and its init then refuses the duplicate device — a clean failure, never a measurement
        local_rank = int(os.environ["AGPU_BENCH_DEVICE_OVERRIDE"])
    dev = GpuDevice(local_rank)  # ArrowErrorGPU(NoDevice) without an MI355X: the HIP path has no CPU fallback
    p = ArrowComputePipeline(dev, "bench", fuse=False)  # (AGPU_FUSE=1 in the environment must not turn the timed ops into recordings)
    h = p._handle
    # the same communicator code path at every world size (world 1 = a one-rank RCCL communicator): init, barrier,
    # all-reduce are exercised on a single-GPU box too
    # this process is the bench's own: the loopback bootstrap may stay in its environment for good (the library itself only sets it for the
    # duration of a rendezvous — sharding.loopback_bootstrap — so that it never leaks into another RCCL user of a host process)
    sharding.prefer_loopback_bootstrap(world)
    comm = sharding.Communicator.from_env(dev, timeout_s=args.rendezvous_timeout)
    runtime = sharding.Communicator.runtime_info()
    # ---- what the line may claim as n_gpus is what RCCL says, proven by one identity record per rank gathered THROUGH the
    # communicator (rank, ncclCommCount, PCI address, uuid): the launcher's WORLD_SIZE is only the expectation it is checked
    # against.  A world that is not N distinct devices in one communicator ends here, on every rank alike (the verdict is a
    # pure function of the gathered records), with ONE JSON error line from rank 0 and a non-zero exit — never a number.
    peers = comm.peers(p)
    proof = sharding.world_proof(peers, args.gpus)
    if comm.size()[0] != proof["rccl_ranks"]:
        proof["ok"] = False
        proof["errors"].append(f"ncclCommCount {comm.size()[0]} on rank {rank} vs {proof['rccl_ranks']} gathered records")
    if rank == 0:
        print(f"bench.py: world {world} (RCCL reports {proof['rccl_ranks']} ranks on {proof['distinct_devices']} distinct devices), {runtime}", file=sys.stderr)
    if not proof["ok"] and not os.environ.get("AGPU_BENCH_ALLOW_WORLD_MISMATCH"):
        if rank == 0:
            real_stdout.write(json.dumps({"error": "the communicator is not the world --gpus names", "n_gpus_requested": args.gpus,
                                          "launcher_world": world, "rccl_ranks": proof["rccl_ranks"], "distinct_devices": proof["distinct_devices"],
                                          "devices": proof["devices"], "details": proof["errors"]}) + "\n")
            real_stdout.flush()
        comm.barrier(p)
        comm.close()
        sys.exit(3)
    world = proof["rccl_ranks"]  # from here on the world IS what RCCL reports

    if args.scaling == "weak":  # the column has world × rows rows, this rank owns one contiguous chunk of it
        shard = sharding.shard_rows(args.rows * world, world, rank)
    else:  # the column has `rows` rows in total
        shard = sharding.shard_rows(args.rows, world, rank)
    n, row0 = shard.rows, shard.row0
    total_rows = args.rows * world if args.scaling == "weak" else args.rows
    vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
    nb = (n + 63) // 64 * 8

    # two tables (the f32 columns of the add; the i32 columns and the bitmaps of the compare), each out of one block placed for
    # the HBM channel hash (agpu_malloc_table): element i of the columns a kernel reads together falls into
    # different hash classes — adjacent columns of a table differ in the strongest hash bit
    fa, fb, fo = dev.create_table_buffers([4 * n] * 3)
    ia, ib, va, vb, ob, ov = dev.create_table_buffers([4 * n] * 2 + [nb] * 4)

    def synth_inputs(fa, fb, ia, ib, va, vb):
        capi.call("agpu_synth_f32", h, vp(fa), n, SEED, row0, C.c_float(-1000.0), C.c_float(1000.0))
        capi.call("agpu_synth_f32", h, vp(fb), n, SEED + 1, row0, C.c_float(-1000.0), C.c_float(1000.0))
        capi.call("agpu_synth_i32", h, vp(ia), n, SEED + 2, row0, 1024)
        capi.call("agpu_synth_i32", h, vp(ib), n, SEED + 3, row0, 1024)
        capi.call("agpu_synth_bits", h, vp(va), n, SEED + 4, row0, C.c_double(0.9))
        capi.call("agpu_synth_bits", h, vp(vb), n, SEED + 5, row0, C.c_double(0.9))

    synth_inputs(fa, fb, ia, ib, va, vb)
    p.sync()

    def ev():
        e = C.c_void_p()
        capi.call("agpu_event_create", dev._handle, C.byref(e))
        return e

    def ms_between(s, e):
        ms = C.c_float()
        capi.call("agpu_event_elapsed_ms", s, e, C.byref(ms))
        return ms.value

    def mean_ms(pairs):
        return sum(ms_between(s, e) for s, e in pairs) / max(len(pairs), 1)

    def make_step(bufs, rows):
        fa_, fb_, fo_, ia_, ib_, va_, vb_, ob_, ov_ = bufs

        def step(evs=None):
            if evs is not None:
                capi.call("agpu_event_record", evs[0][0], h)
            capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(fa_), vp(fb_), vp(fo_), rows)
            if evs is not None:
                capi.call("agpu_event_record", evs[0][1], h)
                capi.call("agpu_event_record", evs[1][0], h)
            capi.call("agpu_compare_validity", h, capi.CMP_EQ, capi.I32, vp(ia_), vp(ib_), vp(va_), vp(vb_), vp(ob_), vp(ov_), rows)
            if evs is not None:
                capi.call("agpu_event_record", evs[1][1], h)
        return step

    def barrier():
        comm.barrier(p)  # RCCL all-reduce of one word + host wait on the stream (with a deadline)
        dev.sync()

    stat_buf = dev.create_empty_buffer(max(64, 8 * 6 * world))  # across_ranks' f64 staging: six numbers per rank at most (the final-reduce check)

    def across_ranks(values, op):
        """MIN / MAX / SUM of a few f64 over the ranks through the C ABI's communicator (identity at world 1)."""
        import numpy as np

        arr = np.array(values, np.float64)
        capi.call("agpu_upload", h, vp(stat_buf), C.c_void_p(arr.ctypes.data), arr.nbytes)
        comm.all_reduce(p, op, capi.COMM_F64, stat_buf, len(values))
        comm.sync(p)  # the wait with the collective deadline: a download would block for ever behind a collective a dead peer never joins
        out = np.empty_like(arr)
        capi.call("agpu_download", h, C.c_void_p(out.ctypes.data), vp(stat_buf), arr.nbytes)
        return [float(x) for x in out]

    def all_ranks_ok(ok: bool) -> bool:
        """A leg that contains collectives runs inside try/except on every rank; if ONE rank failed, the ranks would issue different
        collective sequences from here on (wrong numbers, or a hang).  So after such a leg the ranks agree on a failure word (MAX):
        at world 1 this is the local flag; at world > 1 any failure is fatal for every rank — exit together, no line."""
        bad = across_ranks([0.0 if ok else 1.0], capi.RED_MAX)[0] > 0.0
        if bad and world > 1:
            print(f"bench.py: rank {rank}: a rank failed inside a collective leg; all ranks exit", file=sys.stderr)
            sys.stderr.flush()
            os._exit(4)
        return not bad

    def timed_run(step, steps, warmup):
        """The contract's timed region: warm-up, barrier + device sync, EXACTLY `steps` steps, barrier + device sync; MAX over
        ranks.  Also the per-kernel HIP-event means of this rank and their min / max over ranks (a straggler shows up here)."""
        add_ev = [(ev(), ev()) for _ in range(steps)]
        eq_ev = [(ev(), ev()) for _ in range(steps)]
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step((add_ev[i], eq_ev[i]))
        barrier()
        mine = time.perf_counter() - t0
        add_ms, eq_ms = mean_ms(add_ev), mean_ms(eq_ev)
        for pair in add_ev + eq_ev:
            for e in pair:
                capi.lib().agpu_event_destroy(e)
        lo = across_ranks([mine, add_ms, eq_ms], capi.RED_MIN)
        hi = across_ranks([mine, add_ms, eq_ms], capi.RED_MAX)
        per_rank = {"elapsed_s": {"min": round(lo[0], 6), "max": round(hi[0], 6)},
                    "add_ms": {"min": round(lo[1], 4), "max": round(hi[1], 4)},
                    "eq_ms": {"min": round(lo[2], 4), "max": round(hi[2], 4)}}
        return hi[0], add_ms, eq_ms, per_rank

    main_bufs = (fa, fb, fo, ia, ib, va, vb, ob, ov)
    elapsed, add_ms, eq_ms, per_rank = timed_run(make_step(main_bufs, n), args.steps, args.warmup)
    step_bytes_per_row = ADD_BYTES_PER_ROW + EQ_BYTES_PER_ROW
    value = step_bytes_per_row * total_rows * args.steps / elapsed / 1e9
    add_gbps = ADD_BYTES_PER_ROW * n / add_ms / 1e6
    eq_gbps = EQ_BYTES_PER_ROW * n / eq_ms / 1e6

    # ---- windows of the benchmarked outputs, EVERY rank its own (first / middle / last 65 536 rows of its shard, at its own row0), checked
    # against the oracle on that rank and AND-ed over the ranks through the communicator: a line of ANY world carries the parity of all
    # its shards.  The oracle is only ever the checker, after the timed region; the product path never sees it.
    want_cpu = rank == 0 and not args.no_cpu_baseline  # the CPU leg: rank 0 at every world size
    gpu_parity = None
    if not args.no_cpu_baseline:
        import numpy as np

        ok_local, why = False, ""
        try:
            windows = []
            w = 1 << 16
            for start in (0, (n // 2) // 64 * 64, max(0, (n - w) // 64 * 64)):
                cnt = min(w, n - start)
                nbytes = (cnt + 7) // 8
                got, gb, gv = np.empty(cnt, np.float32), np.empty(nbytes, np.uint8), np.empty(nbytes, np.uint8)
                capi.call("agpu_download", h, C.c_void_p(got.ctypes.data), C.c_void_p(fo.ptr + 4 * start), 4 * cnt)
                capi.call("agpu_download", h, C.c_void_p(gb.ctypes.data), C.c_void_p(ob.ptr + start // 8), nbytes)
                capi.call("agpu_download", h, C.c_void_p(gv.ctypes.data), C.c_void_p(ov.ptr + start // 8), nbytes)
                windows.append({"row": row0 + start, "rows": cnt, "add": got, "eq_bits": gb, "eq_validity": gv})
            ok_local = check_headline_windows(windows)
        except Exception as e:  # noqa: BLE001 — a rank that cannot check says so; the collective sequence stays the same on every rank
            why = f" ({type(e).__name__}: {e})"
            print(f"bench.py: rank {rank}: window check failed{why}", file=sys.stderr)
        bad_ranks = int(round(across_ranks([0.0 if ok_local else 1.0], capi.RED_SUM)[0]))
        gpu_parity = (f"GPU outputs bit-exact vs oracle on 3 windows of 65536 rows (first/middle/last) of every rank's shard: all {world} rank(s), "
                      f"each at its own row0, AND over ranks through the communicator" if bad_ranks == 0
                      else f"MISMATCH on {bad_ranks} of {world} rank(s){why}")

    extra = {"runtime": runtime, "per_rank": per_rank, "rccl_ranks": proof["rccl_ranks"], "distinct_devices": proof["distinct_devices"],
             # True only for a ONE-rank run whose RCCL bootstrap did not come up within the deadline: the communicator is then local (device
             # copies, no RCCL object) and "rccl_ranks" is the local communicator's 1 — include/arrow_gpu.h agpu_comm_is_local
             "rccl_local_fallback": bool(comm.is_local),
             "devices": peers, "launcher_world": int(os.environ.get("WORLD_SIZE", "1")),
             "world_proof": "one identity record per rank all-gathered through the RCCL communicator (agpu_comm_peers): rank / ncclCommCount as "
                            "RCCL reports them on that rank, PCI address, uuid, pid; n_gpus = ncclCommCount, checked against --gpus"}

    # ---- strong-scaling leg of a weak run: the same step over this rank's share of a `rows`-row column (shard_rows cuts
    # it: 125 M rows per GPU at world 8), on the leading rows of the resident shards; same barrier-bracketed timing.
    # One launch per N then yields both curves.  At world 1 strong == weak, nothing to add.
    if (world > 1 or os.environ.get("AGPU_BENCH_STRONG_LEG")) and args.scaling == "weak" and not args.no_extra_configs:  # (env: world-1 rehearsal of this leg)
        try:
            ns = sharding.shard_rows(args.rows, world, rank).rows
            el_s, add_s, eq_s, pr_s = timed_run(make_step(main_bufs, ns), args.steps, args.warmup)
            extra["strong_scaling"] = {
                "scaling": "strong", "rows_total": args.rows, "rows_per_gpu": ns, "steps": args.steps,
                "value_GBps": round(step_bytes_per_row * args.rows * args.steps / el_s / 1e9, 2),
                "ms_per_step": round(el_s / args.steps * 1e3, 4), "rank0_add_ms": round(add_s, 4), "rank0_eq_ms": round(eq_s, 4),
                "per_rank": pr_s,
                "what": "the 1e9-row column cut into `world` contiguous shards (sharding.shard_rows); same step, same "
                        "barrier-bracketed region, MAX over ranks; run on the leading rows of the weak run's resident shards"}
            leg_ok = True
        except Exception as e:  # noqa: BLE001
            extra["strong_scaling"] = {"error": f"{type(e).__name__}: {e}"}
            leg_ok = False
            if world > 1:  # the other ranks are inside (or past) collectives this rank never issued: nothing to agree on any more
                import traceback

                traceback.print_exc()
                os._exit(4)
        all_ranks_ok(leg_ok)

    # ---- config 5: per-shard sum/min/max + final reduce over RCCL (outside the timed region).  Everything below goes
    # through the C ABI's communicator (include/arrow_gpu.h "multi-GPU"): agpu_comm_reduce = the shard-local kernel
    # (for SUM the reference-order tree, sum_tree_quarter_kernel + sum_tree_combine_kernel) + an all-gather of one 16-byte
    # record per rank + the rank-ordered combine.
    try:  # the headline line must come out even if this leg cannot run (it is reported, not part of `value`) — at world 1;
        # at world > 1 a rank that fails in here has left the collective sequence: fatal for all (below)
        stat_out = {k: dev.create_empty_buffer(16) for k in ("sum", "min", "max", "sum_f64")}
        p.sync()
        stats = {}
        for name, launch in (
                ("sum", lambda: capi.call("agpu_reduce", h, capi.RED_SUM, capi.F32, vp(fa), None, n, vp(stat_out["sum"]))),
                ("min", lambda: capi.call("agpu_reduce", h, capi.RED_MIN, capi.F32, vp(fa), None, n, vp(stat_out["min"]))),
                ("max", lambda: capi.call("agpu_reduce", h, capi.RED_MAX, capi.F32, vp(fa), None, n, vp(stat_out["max"]))),
                ("sum_f64", lambda: capi.call("agpu_reduce_sum_f64", h, vp(fa), None, n, vp(stat_out["sum_f64"])))):
            launch()  # warm (scratch allocation)
            reps = 5
            pairs = [(ev(), ev()) for _ in range(reps)]
            for s_, e_ in pairs:  # the shard-local kernel alone, one event pair per launch
                capi.call("agpu_event_record", s_, h)
                launch()
                capi.call("agpu_event_record", e_, h)
            ms_k = mean_ms(pairs)
            stats[name] = {"local_ms": round(ms_k, 4), "local_GBps": round(4.0 * n / ms_k / 1e6, 1),
                           "frac_hbm_peak": round(4.0 * n / ms_k / 1e6 / HBM_PEAK_GBPS, 4)}
        import numpy as _np

        def scalar(buf, dt):
            return dev.retrive_data(buf, _np.dtype(dt).itemsize, pipeline=p).view(dt)[0]

        # this rank's LOCAL statistics (what the launches above left in the buffers), before the collectives overwrite them: their bit
        # patterns + the row count are gathered over the ranks (a SUM all-reduce of one slot per rank) for the check below
        loc = {"sum": scalar(stat_out["sum"], _np.float32), "min": scalar(stat_out["min"], _np.float32),
               "max": scalar(stat_out["max"], _np.float32), "sum_f64": scalar(stat_out["sum_f64"], _np.float64)}
        b64 = int(_np.array([loc["sum_f64"]], _np.float64).view(_np.uint64)[0])
        mine = [int(_np.array([loc[k]], _np.float32).view(_np.uint32)[0]) for k in ("sum", "min", "max")] + [b64 & 0xFFFFFFFF, b64 >> 32, n]
        gathered = across_ranks(slot_vector(rank, world, mine), capi.RED_SUM)
        # the collective form: local kernel + RCCL all-gather + combine, timed end to end on the stream
        cs, ce = ev(), ev()
        comm.reduce(p, capi.RED_SUM, capi.F32, fa, None, n, stat_out["sum"])  # warm RCCL's first-call setup
        barrier()
        capi.call("agpu_event_record", cs, h)
        comm.reduce(p, capi.RED_SUM, capi.F32, fa, None, n, stat_out["sum"])
        comm.reduce(p, capi.RED_MIN, capi.F32, fa, None, n, stat_out["min"])
        comm.reduce(p, capi.RED_MAX, capi.F32, fa, None, n, stat_out["max"])
        comm.reduce_sum_f64(p, fa, None, n, stat_out["sum_f64"])
        capi.call("agpu_event_record", ce, h)
        comm.sync(p)
        four_ms = ms_between(cs, ce)
        got = {"sum": scalar(stat_out["sum"], _np.float32), "min": scalar(stat_out["min"], _np.float32),
               "max": scalar(stat_out["max"], _np.float32), "sum_f64": scalar(stat_out["sum_f64"], _np.float64)}
        # … and the collectives' results against the ORACLE's rank-ordered combine of the gathered local records, on every rank (each
        # holds the result), AND-ed over the ranks: the final reduce proves itself at any world size
        verified, detail = None, None
        if not args.no_cpu_baseline:
            try:
                detail = verify_final_reduce(reduce_records_from_gathered(gathered, world), got)
                mine_ok = all(detail.values())
            except Exception as e:  # noqa: BLE001
                detail, mine_ok = {"error": f"{type(e).__name__}: {e}"}, False
            verified = int(round(across_ranks([0.0 if mine_ok else 1.0], capi.RED_SUM)[0])) == 0
        # ---- the same four statistics in ONE pass over the shard (agpu_reduce_stats_f32 / agpu_comm_reduce_stats_f32: the column is read once,
        # every field bit-identical to the separate reduction) — local kernel timed like the others, then the collective form end to end
        one = {}
        rec_loc, rec_all = dev.create_empty_buffer(32), dev.create_empty_buffer(32)
        one_launch = lambda: capi.call("agpu_reduce_stats_f32", h, vp(fa), None, n, vp(rec_loc))  # noqa: E731
        one_launch()
        pairs = [(ev(), ev()) for _ in range(5)]
        for s_, e_ in pairs:
            capi.call("agpu_event_record", s_, h)
            one_launch()
            capi.call("agpu_event_record", e_, h)
        one_ms = mean_ms(pairs)

        def record_of(buf):
            raw = dev.retrive_data(buf, 24, pipeline=p)
            f_ = raw[:12].view(_np.float32)
            return {"sum": f_[0], "min": f_[1], "max": f_[2], "sum_f64": raw[16:24].view(_np.float64)[0]}

        def same_bits(a_, b_):
            return all(_np.array([a_[k]]).tobytes() == _np.array([b_[k]], _np.array([a_[k]]).dtype).tobytes() for k in ("sum", "min", "max", "sum_f64"))

        one_local_same = same_bits(record_of(rec_loc), loc)
        comm.reduce_stats_f32(p, fa, None, n, rec_all)  # warm
        barrier()
        os_, oe_ = ev(), ev()
        capi.call("agpu_event_record", os_, h)
        comm.reduce_stats_f32(p, fa, None, n, rec_all)
        capi.call("agpu_event_record", oe_, h)
        comm.sync(p)
        one_all_ms = ms_between(os_, oe_)
        one_all_same = same_bits(record_of(rec_all), got)
        one_all_max = across_ranks([one_all_ms], capi.RED_MAX)[0]
        one_ok = int(round(across_ranks([0.0 if (one_local_same and one_all_same) else 1.0], capi.RED_SUM)[0])) == 0
        one = {"local_ms": round(one_ms, 4), "local_GBps": round(4.0 * n / one_ms / 1e6, 1), "frac_hbm_peak": round(4.0 * n / one_ms / 1e6 / HBM_PEAK_GBPS, 4),
               "with_final_reduce_ms": round(one_all_ms, 4), "with_final_reduce_ms_max_over_ranks": round(one_all_max, 4),
               "bit_identical_to_the_four_reductions_all_ranks": one_ok,
               "what": "agpu_reduce_stats_f32 / agpu_comm_reduce_stats_f32: sum (reference tree), min, max and the f64 sum from ONE read of the shard "
                       "(4 B/row) + the same finishing launches and final reduces; every field compared bit for bit with the separate reductions above"}
        local_sum_ms = sum(stats[k]["local_ms"] for k in ("sum", "min", "max", "sum_f64"))
        one["speedup_vs_four_local_launches"] = round(local_sum_ms / one_ms, 3)
        four_max = across_ranks([four_ms], capi.RED_MAX)[0]
        extra["reduce_sum_min_max"] = {
            "rows_total": total_rows, "sum_f32_reference_tree": float(got["sum"]),
            "min": float(got["min"]), "max": float(got["max"]),
            "sum_f64": float(got["sum_f64"]), "per_statistic": stats,
            "verified": verified, "verified_rank0": detail,
            "verified_what": "sum / min / max / sum_f64 as the collectives left them on EVERY rank, bit for bit against the oracle's rank-ordered "
                             "combine (oracle.combine_records) of the per-rank local statistics gathered through the communicator; AND over ranks",
            "local_rank0": {k: float(v) for k, v in loc.items()},
            "four_statistics_with_final_reduce_ms": round(four_ms, 4),
            "four_statistics_with_final_reduce_ms_max_over_ranks": round(four_max, 4),
            "aggregate_GBps": round(4 * 4.0 * total_rows / four_max / 1e6, 1),
            "final_reduce_overhead_ms": round(four_ms - local_sum_ms, 4),
            "final_reduce": f"C ABI agpu_comm_reduce: RCCL all-gather of one 16-byte record per rank (world {world}) + rank-ordered combine",
            "one_pass": one,
        }
    except Exception as e:  # noqa: BLE001
        extra["reduce_sum_min_max"] = {"error": f"{type(e).__name__}: {e}"}
        if world > 1:
            import traceback

            traceback.print_exc()
            os._exit(4)
    # which class of allocation the compare's table drew (DESIGN.md §3 "lucky / unlucky blocks": the same binary and layout runs the
    # almost read-only compare at 0.81–0.85 or at 0.87–0.89 of the roof by what the driver backed the block with; the add does not care)
    extra["layout"] = {"eq_frac_hbm_peak": round(eq_gbps / HBM_PEAK_GBPS, 4), "add_frac_hbm_peak": round(add_gbps / HBM_PEAK_GBPS, 4),
                       "allocation_class": "lucky" if eq_gbps / HBM_PEAK_GBPS >= 0.865 else "unlucky",
                       "what": "allocation lottery of the compare's table: >= 0.865 of the roof on eq + validity = the lucky class (DESIGN.md §3); "
                               "extra.layout_pool has the same step over ordinary pool blocks"}
    extra["kernels"] = {"add_f32": {"ms": round(add_ms, 4), "GBps": round(add_gbps, 1), "frac_hbm_peak": round(add_gbps / HBM_PEAK_GBPS, 4)},
                        "eq_i32_validity": {"ms": round(eq_ms, 4), "GBps": round(eq_gbps, 1), "frac_hbm_peak": round(eq_gbps / HBM_PEAK_GBPS, 4)}}

    # ---- every other kernel BASELINE.json's configs 2–4 name, at the same 1e9 rows: median of 9 HIP-event timings after
    # 10 untimed launches, algorithmic bytes per row, fraction of the 8 TB/s roof; one 65 536-row window each goes to the
    # cpu_baseline leg for the oracle.  Reported beside the headline; not part of `value`.
    cfg_windows = {}
    if rank == 0 and world == 1 and not args.no_extra_configs:
        try:
            import numpy as np

            u8, = dev.create_table_buffers([n])
            sc = dev.create_gpu_buffer_with_data(np.array([100.0], np.float32))
            capi.call("agpu_synth_u8", h, vp(u8), n, SEED + 6, row0)
            ov2 = dev.create_empty_buffer(nb)
            wrow = (n // 2) // 64 * 64
            wcnt = min(1 << 16, n - wrow)

            def grab(buf, itemsize_bits, dtype, count):
                nbytes = count * itemsize_bits // 8
                got = np.empty(nbytes, np.uint8)
                capi.call("agpu_download", h, C.c_void_p(got.ctypes.data), C.c_void_p(buf.ptr + wrow * itemsize_bits // 8), nbytes)
                return got.view(dtype)

            def timed(launch, reps=9):
                # 10 untimed launches with a sync before the last, then the timed ones
                for _ in range(9):
                    launch()
                p.sync()
                launch()
                pairs = [(ev(), ev()) for _ in range(reps)]
                for s_, e_ in pairs:
                    capi.call("agpu_event_record", s_, h)
                    launch()
                    capi.call("agpu_event_record", e_, h)
                ts = sorted(ms_between(s_, e_) for s_, e_ in pairs)
                for pr in pairs:
                    for e in pr:
                        capi.lib().agpu_event_destroy(e)
                return ts[len(ts) // 2]

            cfgs = {}

            def record(name, config, bpr, launch, window):
                ms = timed(launch)
                gbps = bpr * n / ms / 1e6
                cfgs[name] = {"config": config, "alg_bytes_per_row": bpr, "ms": round(ms, 4), "GBps": round(gbps, 1),
                              "frac_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4)}
                if want_cpu:
                    cfg_windows[name] = {"row": row0 + wrow, "rows": wcnt, "got": window()}

            for nm, op in (("sub_f32", capi.OP_SUB), ("mul_f32", capi.OP_MUL), ("div_f32", capi.OP_DIV)):
                record(nm, 2, 12.0, lambda op=op: capi.call("agpu_binary", h, op, capi.F32, vp(fa), vp(fb), vp(fo), n),
                       lambda: grab(fo, 32, np.float32, wcnt))
            record("add_scalar_f32", 2, 8.0, lambda: capi.call("agpu_scalar", h, capi.OP_ADD, capi.F32, vp(fa), vp(sc), vp(fo), n),
                   lambda: grab(fo, 32, np.float32, wcnt))
            for nm, op in (("lt_i32_validity", capi.CMP_LT), ("gt_i32_validity", capi.CMP_GT)):
                record(nm, 3, 8.5, lambda op=op: capi.call("agpu_compare_validity", h, op, capi.I32, vp(ia), vp(ib), vp(va), vp(vb), vp(ob), vp(ov), n),
                       lambda: np.concatenate([grab(ob, 1, np.uint8, wcnt), grab(ov, 1, np.uint8, wcnt)]))
            record("eq_i32_unfused", 3, 8.125, lambda: capi.call("agpu_compare", h, capi.CMP_EQ, capi.I32, vp(ia), vp(ib), vp(ob), n),
                   lambda: grab(ob, 1, np.uint8, wcnt))
            record("validity_and", 3, 0.375, lambda: capi.call("agpu_bitmap_binary", h, capi.OP_AND, vp(va), vp(vb), vp(ov2), n),
                   lambda: grab(ov2, 1, np.uint8, wcnt))
            record("cast_u8_f32", 4, 5.0, lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(fo), n),
                   lambda: grab(fo, 32, np.float32, wcnt))
            # fo now holds f32(u8 column): the input of the stand-alone sin / cos; results go to the add's output table slot
            # of the second input (fb is re-generated below)
            for nm, op in (("sin_f32", capi.UN_SIN), ("cos_f32", capi.UN_COS)):
                record(nm, 4, 8.0, lambda op=op: capi.call("agpu_unary", h, op, capi.F32, vp(fo), vp(fb), n),
                       lambda: grab(fb, 32, np.float32, wcnt))
            for nm, op in (("sin_u8", capi.UN_SIN), ("cos_u8", capi.UN_COS)):
                record(nm, 4, 5.0, lambda op=op: capi.call("agpu_unary", h, op, capi.U8, vp(u8), vp(fb), n),
                       lambda: grab(fb, 32, np.float32, wcnt))
            # config 4 AS WORDED — "cast u8→f32 then sin/cos" — in ONE launch: what a fusing pipeline issues at finish() for
            # `col.cast_op(F32, p).sin_op(p)` (agpu_fused_cast_chain; 5 B/row instead of the pair's 5 + 8)
            class _Step(C.Structure):
                _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]

            def chain(*items):
                arr = (_Step * max(len(items), 1))()
                for k_, (op_, kind_, operand_) in enumerate(items):
                    arr[k_].op, arr[k_].kind, arr[k_].operand = op_, kind_, (operand_.ptr if operand_ is not None else None)
                return arr, len(items)

            for nm, op in (("cast_u8_f32_then_sin_one_launch", capi.UN_SIN), ("cast_u8_f32_then_cos_one_launch", capi.UN_COS)):
                st, ns_ = chain((op, 0, None))
                record(nm, 4, 5.0, lambda st=st, ns_=ns_: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u8), C.cast(st, C.c_void_p), ns_, vp(fb), n),
                       lambda: grab(fb, 32, np.float32, wcnt))
            pair_ms = cfgs["cast_u8_f32"]["ms"] + cfgs["sin_f32"]["ms"]
            cfgs["cast_u8_f32_then_sin_one_launch"]["unfused_pair_ms"] = round(pair_ms, 4)
            cfgs["cast_u8_f32_then_sin_one_launch"]["what"] = ("agpu_fused_cast_chain(U8, [sin]): the launch a fusing pipeline issues at finish() for "
                                                               "cast_op → sin_op; bit-identical to the two-launch pair (13 B/row)")
            capi.call("agpu_synth_f32", h, vp(fb), n, SEED + 1, row0, C.c_float(-1000.0), C.c_float(1000.0))

            # ---- extra.fused: element-wise chains as ONE kernel against the same chain launch by launch, 1e9 rows
            fused = {}
            fd, = dev.create_table_buffers([4 * n])
            sc2 = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
            cs_a, cs_b = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
            capi.call("agpu_synth_f32", h, vp(fd), n, SEED + 11, row0, C.c_float(-1000.0), C.c_float(1000.0))

            def checksum_of(buf, nbytes, cs):
                capi.call("agpu_checksum", h, vp(buf), nbytes, vp(cs))
                return int(dev.retrive_data(cs, 8, pipeline=p).view(np.uint64)[0])

            def fused_row(name, bpr_fused, bpr_unfused, run_fused, run_unfused, result, result_bytes, what):
                ms_f, ms_u = timed(run_fused), timed(run_unfused)
                run_unfused()
                ref = checksum_of(result, result_bytes, cs_a)
                capi.call("agpu_memset", h, vp(result), 0, min(result_bytes, 1 << 20))
                run_fused()
                same = checksum_of(result, result_bytes, cs_b) == ref
                g = bpr_fused * n / ms_f / 1e6
                fused[name] = {"fused_ms": round(ms_f, 4), "unfused_ms": round(ms_u, 4), "alg_bytes_per_row_fused": bpr_fused,
                               "alg_bytes_per_row_unfused": bpr_unfused, "GBps": round(g, 1), "frac_hbm_peak": round(g / HBM_PEAK_GBPS, 4),
                               "speedup": round(ms_u / ms_f, 3), "bit_identical_to_unfused": bool(same), "what": what}

            st_am, n_am = chain((capi.OP_ADD, 1, sc), (capi.OP_MUL, 1, sc2))

            def unf_am():
                capi.call("agpu_scalar", h, capi.OP_ADD, capi.F32, vp(fa), vp(sc), vp(fo), n)
                capi.call("agpu_scalar", h, capi.OP_MUL, capi.F32, vp(fo), vp(sc2), vp(fo), n)

            fused_row("add_scalar_then_mul_scalar", 8.0, 16.0,
                      lambda: capi.call("agpu_fused_chain", h, capi.F32, vp(fa), C.cast(st_am, C.c_void_p), n_am, vp(fo), n), unf_am, fo, 4 * n,
                      "(a + s) * t [crates/arrow/examples/simple.rs:45-72]: agpu_fused_chain vs agpu_scalar twice")
            if want_cpu:
                cfg_windows["fused_add_scalar_then_mul_scalar"] = {"row": row0 + wrow, "rows": wcnt, "got": grab(fo, 32, np.float32, wcnt)}
            st_p, n_p = chain((capi.OP_MUL, 2, fb), (capi.OP_ADD, 2, fd))
            tmp, = dev.create_table_buffers([4 * n])

            def unf_pred():
                capi.call("agpu_binary", h, capi.OP_MUL, capi.F32, vp(fa), vp(fb), vp(tmp), n)
                capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(tmp), vp(fd), vp(tmp), n)
                capi.call("agpu_compare", h, capi.CMP_GT, capi.F32, vp(tmp), vp(fo), vp(ob), n)

            fused_row("mul_add_gt_predicate", 16.125, 32.125,
                      lambda: capi.call("agpu_fused_chain_compare", h, capi.F32, vp(fa), C.cast(st_p, C.c_void_p), n_p, capi.CMP_GT, 2, vp(fo), vp(ob), n),
                      unf_pred, ob, nb, "(a * b + c) > d → bitmap: agpu_fused_chain_compare vs mul, add, compare (three of the four columns in the add's "
                                        "table, the fourth a block of its own)")
            del tmp
            # the same predicate with its four columns allocated as ONE table (agpu_malloc_table: the placement DESIGN.md §3 describes) — what the
            # four read streams run at when nothing about their relative placement is left to chance
            try:
                qa, qb, qc, qd, qo = dev.create_table_buffers([4 * n] * 4 + [nb])
                for k_, col in enumerate((qa, qb, qc, qd)):
                    capi.call("agpu_synth_f32", h, vp(col), n, SEED + 20 + k_, row0, C.c_float(-1000.0), C.c_float(1000.0))
                st_q, n_q = chain((capi.OP_MUL, 2, qb), (capi.OP_ADD, 2, qc))
                ms_q = timed(lambda: capi.call("agpu_fused_chain_compare", h, capi.F32, vp(qa), C.cast(st_q, C.c_void_p), n_q, capi.CMP_GT, 2, vp(qd), vp(qo), n))
                fused["mul_add_gt_predicate"]["one_table_ms"] = round(ms_q, 4)
                fused["mul_add_gt_predicate"]["one_table_frac_hbm_peak"] = round(16.125 * n / ms_q / 1e6 / HBM_PEAK_GBPS, 4)
                del qa, qb, qc, qd, qo
            except Exception as e:  # noqa: BLE001 — an extra figure: never the reason a bench line is lost
                fused["mul_add_gt_predicate"]["one_table_error"] = f"{type(e).__name__}: {e}"
            st_c, n_c = chain((capi.OP_MUL, 1, sc2), (capi.OP_ADD, 1, sc))

            def unf_cast():
                capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(fo), n)
                capi.call("agpu_scalar", h, capi.OP_MUL, capi.F32, vp(fo), vp(sc2), vp(fo), n)
                capi.call("agpu_scalar", h, capi.OP_ADD, capi.F32, vp(fo), vp(sc), vp(fo), n)

            fused_row("cast_u8_then_scale_offset", 5.0, 21.0,
                      lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u8), C.cast(st_c, C.c_void_p), n_c, vp(fo), n), unf_cast, fo, 4 * n,
                      "f32(u8) * t + s: agpu_fused_cast_chain vs cast, mul_scalar, add_scalar")
            st_s, n_s = chain((capi.UN_SIN, 0, None))

            def unf_sin():
                capi.call("agpu_cast", h, capi.U8, capi.F32, vp(u8), vp(fo), n)
                capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(fo), vp(fo), n)

            fused_row("cast_u8_then_sin", 5.0, 13.0,
                      lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(u8), C.cast(st_s, C.c_void_p), n_s, vp(fo), n), unf_sin, fo, 4 * n,
                      "BASELINE config 4 as worded: cast u8→f32 then sin, one launch vs two")
            fused["what"] = ("median of 9 HIP-event timings after 10 untimed launches, 1e9 rows; bit_identical_to_unfused = order-independent 64-bit "
                             "checksums of the whole result column equal; parity windows of the fused results in cpu_baseline.configs_parity")
            extra["fused"] = fused
            del fd, sc2, cs_a, cs_b
            cfgs["what"] = ("BASELINE.json configs 2-4 beyond the headline pair, same rows, table-placed buffers: median of 9 HIP-event "
                            "timings after 10 untimed launches; parity of one 65536-row window each in cpu_baseline.configs_parity")
            extra["configs"] = cfgs
            del u8, ov2, sc
        except Exception as e:  # noqa: BLE001
            extra["configs"] = {"error": f"{type(e).__name__}: {e}"}
        # ---- the headline step the way a caller of the HOST API runs it: six ordinary buffers as `from_slice` allocates
        # them (agpu_malloc, one by one), the outputs allocated BY THE OPS — `a.add_op(b)` / `ia.eq_op(ib)` of arrow_gpu_amd, i.e.
        # agpu_malloc_like with the inputs as neighbours — after everything the legs above did to the pool.  Kernel times from the
        # library's own per-launch event pairs (agpu_pipeline_enable_timing): the op allocates on the host between our records.
        try:
            import numpy as np

            import arrow_gpu_amd as ag

            pool = [dev.create_empty_buffer(sz) for sz in [4 * n] * 4 + [nb] * 2]
            pfa, pfb, pia, pib, pva, pvb = pool
            synth_inputs(pfa, pfb, pia, pib, pva, pvb)
            p.sync()
            A, B = ag.Float32ArrayGPU(pfa, dev, n, None), ag.Float32ArrayGPU(pfb, dev, n, None)
            IA = ag.Int32ArrayGPU(pia, dev, n, ag.NullBitBufferGpu(pva, n, dev))
            IB = ag.Int32ArrayGPU(pib, dev, n, ag.NullBitBufferGpu(pvb, n, dev))
            p.enable_timing(2)
            k_api = min(args.steps, 10)

            def api_ms(op):
                ts = []
                for i in range(k_api + 2):
                    r = op()
                    ns, _ = p.last_kernel_ns()
                    p.sync()  # drops the pipeline's keep-alives: the output goes back to the pool, the next call places it afresh
                    del r
                    if i >= 2:
                        ts.append(ns / 1e6)
                return float(np.mean(ts)), [round(t, 4) for t in ts]

            add_p, add_ts = api_ms(lambda: A.add_op(B, p))
            eq_p, eq_ts = api_ms(lambda: IA.eq_op(IB, p))
            p.enable_timing(0)
            r1, r2 = A.add(B), IA.eq(IB)  # warm: the outputs' blocks exist
            del r1, r2
            dev.sync()
            t0 = time.perf_counter()
            for _ in range(k_api):  # the whole step as the reference's caller writes it: a.add(b), ia.eq(ib) — a new pipeline, a new output and
                r1 = A.add(B)       # a finish per call; the results dropped as the next ones are asked for (host time of all that included)
                r2 = IA.eq(IB)
                del r1, r2
            dev.sync()
            el_p = time.perf_counter() - t0
            extra["layout_pool"] = {
                "value_GBps": round(step_bytes_per_row * n * k_api / el_p / 1e9, 2),
                "add_ms": round(add_p, 4), "add_frac_hbm_peak": round(ADD_BYTES_PER_ROW * n / add_p / 1e6 / HBM_PEAK_GBPS, 4),
                "eq_ms": round(eq_p, 4), "eq_frac_hbm_peak": round(EQ_BYTES_PER_ROW * n / eq_p / 1e6 / HBM_PEAK_GBPS, 4),
                "add_ms_per_launch": add_ts, "eq_ms_per_launch": eq_ts,
                "what": "the same step through the HOST API: inputs in six ordinary agpu_malloc blocks (what from_slice gives), "
                        "Float32ArrayGPU.add_op / Int32ArrayGPU.eq_op allocate their outputs (agpu_malloc_like) per call; kernel times "
                        "from the library's per-launch events, value_GBps from the wall clock of the host loop"}
            del A, B, IA, IB, pool, pfa, pfb, pia, pib, pva, pvb
        except Exception as e:  # noqa: BLE001
            extra["layout_pool"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- north_star's "take / put … scatter-gather in crates/routines": 2^28 rows, uniformly random indices unless said otherwise, through
    # the auto policy (a locality probe over the index columns picks the form); G rows/s, median of 5 HIP-event timings.  One 4096-row
    # window of the well-defined results goes to the cpu_baseline leg for the oracle.  Reported beside the headline; not part of `value`.
    if rank == 0 and world == 1 and not args.no_extra_configs and n >= (1 << 28):
        try:
            import numpy as np

            m = 1 << 28
            V, I1, I2, SEQ, OUT = (dev.create_empty_buffer(4 * m) for _ in range(5))
            VB, OB = dev.create_empty_buffer(m // 8 + 64), dev.create_empty_buffer(m // 8 + 64)
            capi.call("agpu_synth_i32", h, vp(V), m, SEED + 7, 0, 0)
            capi.call("agpu_synth_i32", h, vp(I1), m, SEED + 8, 0, m)
            capi.call("agpu_synth_i32", h, vp(I2), m, SEED + 9, 0, m)
            capi.call("agpu_synth_bits", h, vp(VB), m, SEED + 10, 0, C.c_double(0.5))
            seq = np.arange(m, dtype=np.uint32)
            capi.call("agpu_upload", h, vp(SEQ), C.c_void_p(seq.ctypes.data), 4 * m)
            p.sync()
            del seq
            w0, wn = (m // 3) // 64 * 64, 4096

            def med5(launch):
                launch()
                pairs = [(ev(), ev()) for _ in range(5)]
                for s_, e_ in pairs:
                    capi.call("agpu_event_record", s_, h)
                    launch()
                    capi.call("agpu_event_record", e_, h)
                ts = sorted(ms_between(s_, e_) for s_, e_ in pairs)
                for pr in pairs:
                    for e in pr:
                        capi.lib().agpu_event_destroy(e)
                return ts[2]

            def win(buf, nbytes, off):
                got = np.empty(nbytes, np.uint8)
                capi.call("agpu_download", h, C.c_void_p(got.ctypes.data), C.c_void_p(buf.ptr + off), nbytes)
                return got

            swz = {}

            def rec(name, launch, what, window=None):
                ms = med5(launch)
                swz[name] = {"ms": round(ms, 4), "G_rows_per_s": round(m / ms / 1e6, 1), "what": what}
                if want_cpu and window is not None:
                    cfg_windows["swz_" + name] = dict(window(), w0=w0, rows=wn, m=m)

            rec("take_f32_random", lambda: capi.call("agpu_take", h, 4, vp(V), m, vp(I1), vp(OUT), m), "out[i] = values[idx[i]], idx uniformly random",
                lambda: {"kind": "take", "got": win(OUT, 4 * wn, 4 * w0).view(np.uint32)})
            rec("take_f32_random_with_validity", lambda: capi.call("agpu_take_validity", h, 4, vp(V), m, vp(VB), vp(I1), vp(OUT), vp(OB), m),
                "values and validity bits in one pipeline", lambda: {"kind": "take_validity", "got": win(OUT, 4 * wn, 4 * w0).view(np.uint32), "bits": win(OB, wn // 8, w0 // 8)})
            rec("take_bits_random", lambda: capi.call("agpu_take_bits", h, vp(VB), m, vp(I1), vp(OB), m), "Boolean take", lambda: {"kind": "take_bits", "bits": win(OB, wn // 8, w0 // 8)})
            rec("take_f32_sequential", lambda: capi.call("agpu_take", h, 4, vp(V), m, vp(SEQ), vp(OUT), m), "local indices: the probe keeps the streaming direct kernel")
            rec("put_f32_random_to_sequential", lambda: capi.call("agpu_put_bounded", h, 4, vp(V), m, vp(I1), vp(OUT), m, vp(SEQ), m),
                "dst[dst_idx[i]] = src[src_idx[i]]: random source, local destination (the gather into a contiguous selection)",
                lambda: {"kind": "take", "got": win(OUT, 4 * wn, 4 * w0).view(np.uint32)})
            rec("put_f32_sequential_to_random", lambda: capi.call("agpu_put_bounded", h, 4, vp(V), m, vp(SEQ), vp(OUT), m, vp(I2), m),
                "local source, random destination (the scatter of a contiguous selection; duplicate destinations: unspecified winner)")
            rec("put_f32_random_to_random", lambda: capi.call("agpu_put_bounded", h, 4, vp(V), m, vp(I1), vp(OUT), m, vp(I2), m), "both index columns random")
            rec("put_bits_random_to_random", lambda: capi.call("agpu_put_bits_bounded", h, vp(VB), m, vp(I1), vp(OB), m, vp(I2), m), "Boolean put, both index columns random")
            # the columns of one table by one index column (agpu_take_columns): the pipeline's index work — a third of a take — runs once
            try:
                V4 = [V] + [dev.create_empty_buffer(4 * m) for _ in range(3)]
                O4 = [OUT] + [dev.create_empty_buffer(4 * m) for _ in range(3)]
                for k_, v_ in enumerate(V4[1:]):
                    capi.call("agpu_synth_i32", h, vp(v_), m, SEED + 30 + k_, 0, 0)
                p.sync()
                w4 = (C.c_int32 * 4)(4, 4, 4, 4)
                v4 = (C.c_void_p * 4)(*[b_.ptr for b_ in V4])
                o4 = (C.c_void_p * 4)(*[b_.ptr for b_ in O4])
                ms4 = med5(lambda: capi.call("agpu_take_columns", h, 4, w4, v4, m, vp(I1), o4, m))
                swz["take_4_columns_f32_random"] = {"ms": round(ms4, 4), "G_rows_per_s_per_column": round(4 * m / ms4 / 1e6, 1),
                                                    "vs_four_takes": round(4 * swz["take_f32_random"]["ms"] / ms4, 3),
                                                    "what": "agpu_take_columns: four 4-byte columns of one table by one random index column — histogram, "
                                                            "scans and partition of the index once, gather + merge-back per column"}
                if want_cpu:
                    cfg_windows["swz_take_4_columns_f32_random"] = {"kind": "take", "got": win(O4[3], 4 * wn, 4 * w0).view(np.uint32), "w0": w0, "rows": wn, "m": m,
                                                                    "values_seed": SEED + 32}
                del V4, O4
            except Exception as e:  # noqa: BLE001 — an extra figure
                swz["take_4_columns_f32_random"] = {"error": f"{type(e).__name__}: {e}"}
            p.set_tuning("gather_bucket", 1)
            rec("take_f32_random_direct_kernel", lambda: capi.call("agpu_take", h, 4, vp(V), m, vp(I1), vp(OUT), m), "tuning gather_bucket = 1: one 128-byte line fetched per row")
            p.set_tuning("gather_bucket", 0)
            swz["rows"] = m
            extra["swizzle"] = swz
            del V, I1, I2, SEQ, OUT, VB, OB
            # … and the two random forms at BASELINE's own row count (1e9 rows over 1e9-element arrays: 1 MiB regions, the take's 8-slot gather);
            # parity at this size: tests/test_gpu_swizzle_fullsize.py
            m = n
            V, I1, I2, OUT = (dev.create_empty_buffer(4 * m) for _ in range(4))
            capi.call("agpu_synth_i32", h, vp(V), m, SEED + 7, 0, 0)
            capi.call("agpu_synth_i32", h, vp(I1), m, SEED + 8, 0, m)
            capi.call("agpu_synth_i32", h, vp(I2), m, SEED + 9, 0, m)
            p.sync()
            big = {}
            for name, launch in (("take_f32_random", lambda: capi.call("agpu_take", h, 4, vp(V), m, vp(I1), vp(OUT), m)),
                                 ("put_f32_random_to_random", lambda: capi.call("agpu_put_bounded", h, 4, vp(V), m, vp(I1), vp(OUT), m, vp(I2), m))):
                ms = med5(launch)
                big[name] = {"ms": round(ms, 4), "G_rows_per_s": round(m / ms / 1e6, 1)}
            big["rows"] = m
            extra["swizzle_full_rows"] = big
            del V, I1, I2, OUT
        except Exception as e:  # noqa: BLE001
            extra["swizzle"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- the reference's own criterion workloads on the GPU, through the HOST API exactly as its benches call it
    # (add_dyn(column, 1-element column) at 10 Mi rows; UInt32ArrayGPU::broadcast(2, n).sum() at 1 Mi / 10 Mi rows)
    # [crates/benchmarks/benches/compare_gpu_arrow.rs:18-43, compare_sum.rs:17-40].  Unlike criterion's loop — which
    # returns after queue.submit and so times submission — every iteration here ends with a pipeline sync.
    if rank == 0 and world == 1:
        import numpy as _np2

        import arrow_gpu_amd as ag

        cnt = 10 * 1024 * 1024
        col = ag.Float32ArrayGPU.from_slice(_np2.arange(cnt, dtype=_np2.float32), dev)
        val = ag.Float32ArrayGPU.from_slice(_np2.array([100.0], _np2.float32), dev)

        def best_ms(fn, k=15):
            fn()
            dev.sync()
            ts = []
            for _ in range(k):
                t0 = time.perf_counter()
                fn()
                dev.sync()
                ts.append(time.perf_counter() - t0)
            return round(min(ts) * 1e3, 4)

        w = {"f32_add_scalar_10Mi_ms": best_ms(lambda: ag.add_dyn(col, val))}
        for label, rows_ in (("u32_sum_1Mi_ms", 1024 * 1024), ("u32_sum_10Mi_ms", cnt)):
            u = ag.UInt32ArrayGPU.broadcast(2, rows_, dev)
            w[label] = best_ms(lambda u=u: u.sum())
            assert int(u.sum().raw_values()[0]) == 2 * rows_
            # … and with the VALUE on the host (what the CPU kernel returns): sum() + raw_values(), one device-level wait that carries the
            # scalar back through the device's pinned mailbox (round 5, R5.10; it was a device sync + a copy + a stream sync)
            w[label.replace("_ms", "_value_on_host_ms")] = best_ms(lambda u=u: u.sum().raw_values())
        w["what"] = "host API call + device sync per iteration (new pipeline + new output buffer per call, like the reference), best of 15"
        extra["reference_bench_workloads_gpu"] = w
    if rank == 0:
        traffic, traffic_source = None, None
        if world == 1 and n == ROWS and not args.no_traffic and not args.no_extra_configs:
            p.sync()
            traffic, traffic_source = measure_traffic_live()
            if traffic is None:
                print(f"bench.py: live traffic measurement unavailable ({traffic_source}); using profiles/hbm_traffic.json", file=sys.stderr)
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")  # written by tools/pmc_traffic.py from rocprofv3 --pmc passes
        if traffic is None and os.path.exists(tpath) and n == ROWS:
            try:
                traffic = json.load(open(tpath)).get("add_f32_bytes_per_launch")
                traffic_source = ("profiles/hbm_traffic.json: HBM bytes per launch of this kernel from separate rocprofv3 --pmc FETCH_SIZE / "
                                  "WRITE_SIZE passes over this same command (tools/profile_bench.sh), read side x2 per the guide's gfx950 "
                                  "correction; NOT measured inside this run")
            except Exception:
                traffic = None
        shard_txt = (f"chunk-sharded x{world}: every rank owns its own {args.rows}-row shard of a {total_rows}-row column"
                     if args.scaling == "weak" else
                     f"chunk-sharded x{world}: one {total_rows}-row column cut into {world} contiguous shards (sharding.shard_rows)")
        line = {
            "metric": "GB/s on 1B-row f32 add + i32 eq (algorithmic bytes, inputs resident in HBM)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32+i32", "data": "synthetic",
            "config": {"workload": "f32 add (1e9 rows, no nulls) + i32 eq -> bitmap with fused validity AND (1e9 rows, 10% nulls/side)",
                       "rows_per_gpu": n, "rows_total": total_rows, "sharding": shard_txt + ", no data-path collective",
                       "layout": "columns allocated as two tables placed for the HBM channel hash (agpu_malloc_table, DESIGN.md §3)",
                       "frac_hbm_peak_per_gpu": round(value / world / HBM_PEAK_GBPS, 4),
                       # the same two kernels over what a caller of the host API gets (extra.layout_pool): ordinary blocks, outputs placed by the ops
                       **({"host_api": {"add_frac_hbm_peak": extra["layout_pool"]["add_frac_hbm_peak"],
                                        "eq_frac_hbm_peak": extra["layout_pool"]["eq_frac_hbm_peak"]}}
                          if "add_frac_hbm_peak" in extra.get("layout_pool", {}) else {}),
                       **({"tuning": args.tune} if args.tune else {})},
            "roofline": {"bound": "hbm", "kernel": "ew_kernel<float, OpAdd> (agpu_binary ADD f32)",
                         "achieved": round(add_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(add_gbps / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": traffic_source if traffic is not None else None,
                         "algorithmic_bytes_per_launch": ADD_BYTES_PER_ROW * n, "launch_ms": round(add_ms, 4),
                         # rank 0's launch above; the slowest and the fastest rank's mean launch of the same kernel beside it
                         "frac_per_rank": {"min": round(ADD_BYTES_PER_ROW * n / per_rank["add_ms"]["max"] / 1e6 / HBM_PEAK_GBPS, 4),
                                           "max": round(ADD_BYTES_PER_ROW * n / per_rank["add_ms"]["min"] / 1e6 / HBM_PEAK_GBPS, 4)}},
            "extra": extra,
        }
        if want_cpu:
            try:
                line["cpu_baseline"] = cpu_baseline(args.cpu_sample_rows, gpu_parity, cfg_windows)
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        else:
            line["cpu_baseline"] = None
        line["gpu_parity"] = gpu_parity  # also inside cpu_baseline; here so that a --no-… variant of the line still says what was checked
        try:
            C.CDLL(None).fflush(None)  # whatever C libraries buffered for "stdout" goes to stderr now, not after the line
        except Exception:  # noqa: BLE001
            pass
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    barrier()  # nobody tears its communicator down while rank 0 still measures the CPU leg (≈ 40 s, inside the collective deadline) / prints
    comm.close()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:  # noqa: BLE001 — a failed rank must END (a pending RCCL rendezvous thread would keep the process alive)
        import traceback

        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
