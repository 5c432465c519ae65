#!/usr/bin/env python3
"""DEV TOOL (GPU box), round 6: the packed-f32 sin / cos under the launch tunings that were chosen for the f64 form — occupancy cap (wave_lds),
tiles per block (tuning "tiles") — for the stand-alone kernels and the chains VERDICT r5 names:
    f32 sin, f32 cos, (x·s).sin(), cast i16 → f32 → sin (one launch), cast(u16)·s → sin (one launch)
One process = one library (AGPU_LIB picks an A/B build of AGPU_SINCOS_U); prints one line per kernel: fraction of the 8 TB/s roof per setting."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
capi.call("agpu_set_tuning", b"tile_auto", 1)  # static tiles: the sweep sets them itself
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "sc"); q = CmpQuery(dev); h = p._handle
u16, f, g = dev.create_table_buffers([2 * n, 4 * n, 4 * n])
capi.call("agpu_synth_u8", h, C.c_void_p(u16.ptr), 2 * n, 7, 0)
capi.call("agpu_synth_f32", h, C.c_void_p(f.ptr), n, 1, 0, C.c_float(-1000.0), C.c_float(1000.0)); p.sync()
S = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]
def chain(*items):
    arr = (Step * len(items))()
    for k, (op, kind, operand) in enumerate(items):
        arr[k].op, arr[k].kind, arr[k].operand = op, kind, (operand.ptr if operand is not None else None)
    return arr, len(items)
vp = lambda b: C.c_void_p(b.ptr)
c_sin, n_sin = chain((capi.UN_SIN, 0, None))
c_ms, n_ms = chain((capi.OP_MUL, 1, S), (capi.UN_SIN, 0, None))
capi.call("agpu_synth_f32", h, C.c_void_p(g.ptr), n, 3, 0, C.c_float(0.001), C.c_float(1000.0)); p.sync()  # positive column for log
fpos = dev.create_empty_buffer(4 * n)
capi.call("agpu_synth_f32", h, C.c_void_p(fpos.ptr), n, 3, 0, C.c_float(0.001), C.c_float(1000.0)); p.sync()
c_log, n_log = chain((capi.UN_LOG, 0, None))
K = {"f32 log": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_LOG, capi.F32, vp(fpos), vp(g), n)),
     "cast u16 -> log": (6.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U16, vp(u16), C.cast(c_log, C.c_void_p), n_log, vp(g), n)),
     "f32 sin": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(g), n)),
     "f32 cos": (8.0, lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.F32, vp(f), vp(g), n)),
     "(x*s).sin()": (8.0, lambda: capi.call("agpu_fused_chain", h, capi.F32, vp(f), C.cast(c_ms, C.c_void_p), n_ms, vp(g), n)),
     "cast i16 -> sin": (6.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.I16, vp(u16), C.cast(c_sin, C.c_void_p), n_sin, vp(g), n)),
     "cast(u16)*s -> sin": (6.0, lambda: capi.call("agpu_fused_cast_chain", h, capi.U16, vp(u16), C.cast(c_ms, C.c_void_p), n_ms, vp(g), n))}
def med(fn, bpr):
    for _ in range(3): fn()
    p.sync(); ts = []
    for _ in range(7):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return bpr * n / sorted(ts)[3] / 1e6 / 8000
print("lib:", os.environ.get("AGPU_LIB", "default"), flush=True)
caps = [int(x) for x in os.environ.get("CAPS", "-1,0,4200,6800,10240").split(",")]
tiles = [int(x) for x in os.environ.get("TILES", "1,2,4").split(",")]
for name, (bpr, fn) in K.items():
    row = []
    for t in tiles:
        p.set_tuning("tiles", t)
        for cap in caps:
            p.set_tuning("wave_lds", cap)
            row.append(f"t{t}/c{cap}:{med(fn, bpr):.3f}")
    print(name.ljust(20), " ".join(row), flush=True)
