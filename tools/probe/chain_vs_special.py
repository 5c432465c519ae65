#!/usr/bin/env python3
"""DEV TOOL (GPU box): what does the GENERIC chain kernel cost against a kernel specialised on its one transcendental step?
sinh of an i16 column: agpu_unary(SINH, I16) = cvt_wide_kernel<short, float, CvtThenF32<short, UnSinh>> against
agpu_fused_cast_chain(I16, [sinh]) = cast_chain_kernel<short, heavy>; sin of an f32 column: agpu_unary against agpu_fused_chain([sin])."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "cs"); q = CmpQuery(dev); h = p._handle
u16, f, g = dev.create_table_buffers([2 * n, 4 * n, 4 * n])
capi.call("agpu_synth_u8", h, C.c_void_p(u16.ptr), 2 * n, 7, 0)
capi.call("agpu_synth_f32", h, C.c_void_p(f.ptr), n, 1, 0, C.c_float(-30.0), C.c_float(30.0)); p.sync()
class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]
def one(op):
    st = (Step * 1)(); st[0].op, st[0].kind, st[0].operand = op, 0, None
    return st
def med(fn, bpr):
    for _ in range(10): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    ms = sorted(ts)[4]
    return f"{ms:.3f} ms {bpr * n / ms / 1e6 / 8000:.3f}"
vp = lambda b: C.c_void_p(b.ptr)
st_sinh, st_sin = one(capi.UN_SINH), one(capi.UN_SIN)
for rnd in range(2):
    print("sinh_i16 specialised :", med(lambda: capi.call("agpu_unary", h, capi.UN_SINH, capi.I16, vp(u16), vp(g), n), 6.0))
    print("cast i16 -> sinh chain:", med(lambda: capi.call("agpu_fused_cast_chain", h, capi.I16, vp(u16), C.cast(st_sinh, C.c_void_p), 1, vp(g), n), 6.0))
    print("cast i16 -> sin chain :", med(lambda: capi.call("agpu_fused_cast_chain", h, capi.I16, vp(u16), C.cast(st_sin, C.c_void_p), 1, vp(g), n), 6.0))
    print("sin f32 standalone    :", med(lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(f), vp(g), n), 8.0))
    print("sin f32 1-step chain  :", med(lambda: capi.call("agpu_fused_chain", h, capi.F32, vp(f), C.cast(st_sin, C.c_void_p), 1, vp(g), n), 8.0))
    print("sinh f32 standalone   :", med(lambda: capi.call("agpu_unary", h, capi.UN_SINH, capi.F32, vp(f), vp(g), n), 8.0))
    print("sinh f32 1-step chain :", med(lambda: capi.call("agpu_fused_chain", h, capi.F32, vp(f), C.cast(st_sinh, C.c_void_p), 1, vp(g), n), 8.0), flush=True)
