#!/usr/bin/env python3
"""DEV TOOL: time the cast-f32→u8 shapes of narrow_probe.hip against the product kernel (1e9 rows, one process)."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000 // 4096 * 4096
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnarrow_probe.so"))
lib.probe_narrow.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "c"); q = CmpQuery(dev)
A, O, O2 = dev.create_table_buffers([4 * n, n, n])
capi.call("agpu_synth_f32", p._handle, C.c_void_p(A.ptr), n, 1, 0, C.c_float(-50), C.c_float(300))
rows = []
def t(label, f):
    f(); p.sync(); ts = []
    for _ in range(9):
        q.begin(p); f(); q.end(p); ts.append(q.wait_for_results())
    ms = float(np.median(ts)); r = {"kernel": label, "ms": round(ms, 4), "frac_8TBs": round(5 * n / ms / 1e6 / 8000, 4)}
    rows.append(r); print(r, flush=True)
prod = lambda: capi.call("agpu_cast", p._handle, capi.F32, capi.U8, C.c_void_p(A.ptr), C.c_void_p(O.ptr), n)
t("PRODUCT cast f32->u8", prod)
for v, b, u in ((0, 256, 4), (0, 256, 2), (0, 256, 1), (0, 64, 4), (0, 64, 2), (0, 64, 1), (1, 64, 1)):
    def f(v=v, b=b, u=u):
        rc = lib.probe_narrow(C.c_void_p(A.ptr), C.c_void_p(O2.ptr), n, v, b, u, C.c_void_p(p.stream())); assert rc == 0, rc
    t(f"probe v{v} block{b} u{u}", f)
t("PRODUCT cast f32->u8", prod)
cs1, cs2 = dev.create_empty_buffer(16), dev.create_empty_buffer(16)
prod(); lib.probe_narrow(C.c_void_p(A.ptr), C.c_void_p(O2.ptr), n, 1, 64, 1, C.c_void_p(p.stream()))
capi.call("agpu_checksum", p._handle, C.c_void_p(O.ptr), n, C.c_void_p(cs1.ptr)); capi.call("agpu_checksum", p._handle, C.c_void_p(O2.ptr), n, C.c_void_p(cs2.ptr))
a = dev.retrive_data(cs1, 8, pipeline=p).view(np.uint64)[0]; b = dev.retrive_data(cs2, 8, pipeline=p).view(np.uint64)[0]
print("v1 output identical to product:", a == b)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "sweep_narrow.json"), "w"), indent=1)
