#!/usr/bin/env python3
"""DEV TOOL: variants of the reference-order f32 tree sum's first level (sum_probe.hip) against the product, 1e9 rows."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000 // 65536 * 65536
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsum_probe.so"))
lib.probe_sum.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "s")
q = CmpQuery(dev)
A = dev.create_empty_buffer(4 * n)
P = dev.create_empty_buffer(4 * (n // 16384) + 64)
R = dev.create_empty_buffer(64)
capi.call("agpu_synth_f32", p._handle, C.c_void_p(A.ptr), n, 1, 0, C.c_float(-1), C.c_float(1))
p.sync()
rows = []


def t(label, f):
    for _ in range(3):
        f()
    p.sync()
    ts = []
    for _ in range(10):
        q.begin(p); f(); q.end(p)
        ts.append(q.wait_for_results())
    ms = float(np.median(ts))
    r = {"kernel": label, "ms": round(ms, 4), "frac_8TBs": round(4 * n / ms / 1e6 / 8000, 4)}
    rows.append(r)
    print(json.dumps(r), flush=True)


t("PRODUCT f32 sum (tree order, all levels)", lambda: capi.call("agpu_reduce", p._handle, capi.RED_SUM, capi.F32, C.c_void_p(A.ptr), None, n, C.c_void_p(R.ptr)))
t("PRODUCT f32 sum f64", lambda: capi.call("agpu_reduce_sum_f64", p._handle, C.c_void_p(A.ptr), None, n, C.c_void_p(R.ptr)))
t("PRODUCT f32 min", lambda: capi.call("agpu_reduce", p._handle, capi.RED_MIN, capi.F32, C.c_void_p(A.ptr), None, n, C.c_void_p(R.ptr)))
for variant, cap in ((0, 0), (1, 0), (2, 0), (3, 0), (2, 65536), (2, 16384), (2, 8192), (3, 16384), (0, 4096), (1, 4096)):
    def f(variant=variant, cap=cap):
        rc = lib.probe_sum(C.c_void_p(A.ptr), n, C.c_void_p(P.ptr), variant, cap, C.c_void_p(p.stream()))
        assert rc == 0, rc
    t(f"probe variant {variant} grid cap {cap}", f)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "sum_probe.json"), "w"), indent=1)
