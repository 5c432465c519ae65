#!/usr/bin/env python3
"""DEV TOOL (GPU box, under rocprofv3 --pmc): sin / cos f32, cast u16 → f32 and sin_u8 at 1e9 rows, three launches WITHOUT the occupancy cap
(tuning wave_lds = -1) then three WITH the product default — the dispatches differ in their LDS_Block_Size column."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "pmc"); h = p._handle
p.set_tuning("tile_auto", 1)
u8, u16, f, g = dev.create_table_buffers([n, 2 * n, 4 * n, 4 * n])
capi.call("agpu_synth_u8", h, C.c_void_p(u8.ptr), n, 6, 0)
capi.call("agpu_synth_u8", h, C.c_void_p(u16.ptr), 2 * n, 7, 0)
capi.call("agpu_synth_f32", h, C.c_void_p(f.ptr), n, 1, 0, C.c_float(0.001), C.c_float(1000.0)); p.sync()
vp = lambda b: C.c_void_p(b.ptr)
K = [lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(g), n),
     lambda: capi.call("agpu_cast", h, capi.U16, capi.F32, vp(u16), vp(g), n),
     lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(u8), vp(g), n)]
for cap in (-1, 0):
    p.set_tuning("wave_lds", cap)
    for fn in K:
        for _ in range(3):
            fn()
        p.sync()
