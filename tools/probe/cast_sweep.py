#!/usr/bin/env python3
"""DEV TOOL: time the cast-u8→f32 shapes of cast_probe.hip against the product kernel (1e9 rows)."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000 // 4096 * 4096
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcast_probe.so"))
lib.probe_cast.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "c"); q = CmpQuery(dev)
A = dev.create_empty_buffer(n); O = dev.create_empty_buffer(4 * n); O2 = dev.create_empty_buffer(4 * n)
capi.call("agpu_synth_u8", p._handle, C.c_void_p(A.ptr), n, 1, 0)
rows = []
def t(label, f):
    f(); p.sync(); ts = []
    for _ in range(7):
        q.begin(p); f(); q.end(p); ts.append(q.wait_for_results())
    ms = float(np.median(ts)); r = {"kernel": label, "ms": round(ms, 4), "TBps": round(5 * n / ms / 1e9, 3)}
    rows.append(r); print(r, flush=True)
prod = lambda: capi.call("agpu_cast", p._handle, capi.U8, capi.F32, C.c_void_p(A.ptr), C.c_void_p(O.ptr), n)
t("PRODUCT cast u8->f32", prod)
t("PRODUCT cast u8->f32 into the probe's output buffer", lambda: capi.call("agpu_cast", p._handle, capi.U8, capi.F32, C.c_void_p(A.ptr), C.c_void_p(O2.ptr), n))
for v, b, u in ((5, 0, 1), (1, 64, 1), (5, 0, 1), (1, 64, 1), (4, 0, 1)):  # v2/v3: `block` = grid cap (0 = one block per chunk group)
    def f(v=v, b=b, u=u):
        rc = lib.probe_cast(C.c_void_p(A.ptr), C.c_void_p(O2.ptr), n, v, b, u, C.c_void_p(p.stream())); assert rc == 0, rc
    t(f"probe v{v} grid{b} u{u}", f)
# round 3: the load-wave / store-wave split through LDS (v6: block = store waves, u = chunks per round) and the LDS-transposed
# 256-thread tile (v7), alternated with the product kernel on the same buffers
for v, b, u in ((8, 256, 1), (8, 128, 1), (1, 64, 1), (8, 256, 1), (8, 128, 1), (6, 4, 4), (6, 2, 4), (6, 1, 4), (6, 4, 8), (6, 2, 8), (6, 4, 16), (7, 0, 1), (1, 64, 1), (6, 4, 8), (7, 0, 1)):
    def f(v=v, b=b, u=u):
        rc = lib.probe_cast(C.c_void_p(A.ptr), C.c_void_p(O2.ptr), n, v, b, u, C.c_void_p(p.stream())); assert rc == 0, rc
    t(f"probe v{v} {'store-waves' if v == 6 else 'block'}{b} u{u}", f)
    if v in (6, 7, 8):
        cs1, cs2 = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
        prod()
        capi.call("agpu_checksum", p._handle, C.c_void_p(O.ptr), 4 * n, C.c_void_p(cs1.ptr)); capi.call("agpu_checksum", p._handle, C.c_void_p(O2.ptr), 4 * n, C.c_void_p(cs2.ptr))
        assert dev.retrive_data(cs1, 8, pipeline=p).view(np.uint64)[0] == dev.retrive_data(cs2, 8, pipeline=p).view(np.uint64)[0], (v, b, u)
t("PRODUCT cast u8->f32", prod)
for v, b, u in ((0, 64, 4), (0, 256, 2), (1, 128, 1)):
    def f(v=v, b=b, u=u):
        rc = lib.probe_cast(C.c_void_p(A.ptr), C.c_void_p(O2.ptr), n, v, b, u, C.c_void_p(p.stream())); assert rc == 0, rc
    t(f"probe v{v} block{b} u{u}", f)
t("PRODUCT cast u8->f32", prod)
# correctness of the permuted variant
cs1, cs2 = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
prod(); lib.probe_cast(C.c_void_p(A.ptr), C.c_void_p(O2.ptr), n, 4, 0, 1, C.c_void_p(p.stream()))
capi.call("agpu_checksum", p._handle, C.c_void_p(O.ptr), 4 * n, C.c_void_p(cs1.ptr)); capi.call("agpu_checksum", p._handle, C.c_void_p(O2.ptr), 4 * n, C.c_void_p(cs2.ptr))
a = dev.retrive_data(cs1, 8, pipeline=p).view(np.uint64)[0]; b = dev.retrive_data(cs2, 8, pipeline=p).view(np.uint64)[0]
print("v4 output identical to product:", a == b)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/r03_sweep_cast.json", "w"), indent=1)
