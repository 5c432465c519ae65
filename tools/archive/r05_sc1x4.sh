#!/bin/bash
# DEV TOOL (round 5): store policy of the ×4 kernels under the occupancy cap.  Product (cvt ×4 = sc1 nt when capped) against a variant whose
# 8-bit table kernel also stores sc1 nt; wave_lds −1 (no cap; cvt falls back to plain nt), 0 (default), three fresh processes each.
for i in 1 2 3; do for lib in "" tools/probe/variants/libagpu_lut8sc1.so; do echo -n "${lib:-product}: "; AGPU_LIB=${lib:+$PWD/$lib} python - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "x4"); q = CmpQuery(dev); h = p._handle
p.set_tuning("tile_auto", 1)
u8, g = dev.create_table_buffers([n, 4 * n])
capi.call("agpu_synth_u8", h, C.c_void_p(u8.ptr), n, 6, 0); p.sync()
def med(fn):
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return 5.0 * n / sorted(ts)[4] / 1e6 / 8000
row = []
for cap in (-1, 0, -1, 0):
    p.set_tuning("wave_lds", cap)
    row.append(f"{cap}: cast {med(lambda: capi.call('agpu_cast', h, capi.U8, capi.F32, C.c_void_p(u8.ptr), C.c_void_p(g.ptr), n)):.3f} "
               f"sin_u8 {med(lambda: capi.call('agpu_unary', h, capi.UN_SIN, capi.U8, C.c_void_p(u8.ptr), C.c_void_p(g.ptr), n)):.3f} "
               f"cos_i8 {med(lambda: capi.call('agpu_unary', h, capi.UN_COS, capi.I8, C.c_void_p(u8.ptr), C.c_void_p(g.ptr), n)):.3f} |")
print(" ".join(row), flush=True)
PY
done; done
