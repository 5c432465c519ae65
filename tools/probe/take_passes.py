#!/usr/bin/env python3
"""take / put at 2^28 uniformly random rows: direct, pair pipeline (gather_bucket = 3) and — for take — the merge-back pipeline
(gather_bucket = 2), HIP-event medians in one process on the same buffers.  Run under `rocprofv3 --kernel-trace --stats` for the
per-pass times (tools/profile_gather.sh)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "take")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
values, out, idx, idx2 = (dev.create_empty_buffer(4 * n) for _ in range(4))
capi.call("agpu_synth_i32", h, vp(values), n, 1, 0, 0)
capi.call("agpu_synth_i32", h, vp(idx), n, 2, 0, n)
capi.call("agpu_synth_i32", h, vp(idx2), n, 3, 0, n)
p.sync()


def med(f):
    f(), f()
    p.sync()
    ts = []
    for _ in range(iters):
        q.begin(p)
        f()
        q.end(p)
        ts.append(q.wait_for_results())
    return float(np.median(ts))


if os.environ.get("TAKE_REGION_BITS"):  # A/B of the region size (tuning gather_region_bits)
    p.set_tuning("gather_region_bits", int(os.environ["TAKE_REGION_BITS"]))
res = {"rows": n}
PUT_ONLY = bool(os.environ.get("PUT_ONLY"))  # tools/probe/put_variants.sh: only the pair-pipeline rows
for label, mode in ((("take_pairs", 3),) if PUT_ONLY else (("take_direct", 1), ("take_pairs", 3), ("take_mergeback", 2))):
    p.set_tuning("gather_bucket", mode)
    ms = med(lambda: capi.call("agpu_take", h, 4, vp(values), n, vp(idx), vp(out), n))
    res[label] = {"ms": round(ms, 4), "G_rows_per_s": round(n / ms / 1e6, 1)}
    print(label, res[label], flush=True)
for w in (() if PUT_ONLY else (1, 2)):  # narrow values: the same pipelines over the first n bytes / halfwords of the buffers
    for label, mode in ((f"take_u{8 * w}_direct", 1), (f"take_u{8 * w}_pairs", 3), (f"take_u{8 * w}_mergeback", 2)):
        p.set_tuning("gather_bucket", mode)
        ms = med(lambda: capi.call("agpu_take", h, w, vp(values), n, vp(idx), vp(out), n))
        res[label] = {"ms": round(ms, 4), "G_rows_per_s": round(n / ms / 1e6, 1)}
        print(label, res[label], flush=True)
vb, ov = dev.create_empty_buffer(n // 8 + 64), dev.create_empty_buffer(n // 8 + 64)
capi.call("agpu_synth_bits", h, vp(vb), n, 7, 0, C.c_double(0.9))
for label, mode in (() if PUT_ONLY else (("take_with_validity_direct (agpu_take + agpu_take_bits)", 1), ("take_with_validity_mergeback (one pipeline)", 2))):
    p.set_tuning("gather_bucket", mode)
    ms = med(lambda: capi.call("agpu_take_validity", h, 4, vp(values), n, vp(vb), vp(idx), vp(out), vp(ov), n))
    res[label] = {"ms": round(ms, 4), "G_rows_per_s": round(n / ms / 1e6, 1)}
    print(label, res[label], flush=True)
for label, mode in (() if PUT_ONLY else (("take_bits_alone", 1), ("take_bits_mergeback", 2))):  # Boolean take: the bitmap's words are the elements
    p.set_tuning("gather_bucket", mode)
    ms = med(lambda: capi.call("agpu_take_bits", h, vp(vb), n, vp(idx), vp(ov), n))
    res[label] = {"ms": round(ms, 4), "G_rows_per_s": round(n / ms / 1e6, 1)}
    print(label, res[label], flush=True)
if os.environ.get("TAKE_BITS_SWEEP"):  # where the merge-back form starts to win for bits: rows x bitmap size
    sweep = {}
    for lg_n in (22, 24, 25, 26, 27, 28):
        for lg_b in (24, 26, 27, 28):
            if lg_b > 28 or lg_n > 28:
                continue
            capi.call("agpu_synth_i32", h, vp(idx2), 1 << lg_n, 5, 0, 1 << lg_b)
            row = {}
            for label, mode in (("direct", 1), ("mergeback", 2)):
                p.set_tuning("gather_bucket", mode)
                row[label] = round(med(lambda: capi.call("agpu_take_bits", h, vp(vb), 1 << lg_b, vp(idx2), vp(ov), 1 << lg_n)), 4)
            sweep[f"rows 2^{lg_n}, bits 2^{lg_b}"] = row
            print(lg_n, lg_b, row, flush=True)
    res["take_bits_sweep_ms"] = sweep
    capi.call("agpu_synth_i32", h, vp(idx2), n, 3, 0, n)
    p.sync()
for label, mode in ((("put_pairs", 2),) if PUT_ONLY else (("put_direct", 1), ("put_pairs", 2))):
    p.set_tuning("gather_bucket", mode)
    ms = med(lambda: capi.call("agpu_put_bounded", h, 4, vp(values), n, vp(idx), vp(out), n, vp(idx2), n))
    res[label] = {"ms": round(ms, 4), "G_rows_per_s": round(n / ms / 1e6, 1)}
    print(label, res[label], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
if not PUT_ONLY:
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r03_take_passes.json"), "w"), indent=1)
