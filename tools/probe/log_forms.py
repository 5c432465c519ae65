#!/usr/bin/env python3
"""DEV TOOL (GPU box): f32 log — the LDS-table tile kernel (agpu_unary) against the one-wave form that reads the 2 KiB table through L1
(what a 1-step fused chain [log] runs: chain_kernel<float, heavy> → UnLog::ap), uncapped and under occupancy caps."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "log"); q = CmpQuery(dev); h = p._handle
p.set_tuning("tile_auto", 1)
f, g = dev.create_table_buffers([4 * n, 4 * n])
capi.call("agpu_synth_f32", h, C.c_void_p(f.ptr), n, 1, 0, C.c_float(0.001), C.c_float(1000.0)); p.sync()
class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]
def one(op):
    st = (Step * 1)(); st[0].op, st[0].kind, st[0].operand = op, 0, None
    return st
vp = lambda b: C.c_void_p(b.ptr)
def med(fn):
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return 8.0 * n / sorted(ts)[4] / 1e6 / 8000
for name, op in (("log", capi.UN_LOG), ("log2", capi.UN_LOG2), ("exp", capi.UN_EXP), ("cbrt", capi.UN_CBRT), ("acos", capi.UN_ACOS)):
    st = one(op)
    row = [f"unary {med(lambda: capi.call('agpu_unary', h, op, capi.F32, vp(f), vp(g), n)):.3f}"]
    for cap in (-1, 5600, 6800, 8000, 10240):
        p.set_tuning("wave_lds", cap)
        row.append(f"chain@{cap} {med(lambda: capi.call('agpu_fused_chain', h, capi.F32, vp(f), C.cast(st, C.c_void_p), 1, vp(g), n)):.3f}")
    p.set_tuning("wave_lds", 0)
    print(name.ljust(5), "  ".join(row), flush=True)
