import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "ct"); q = CmpQuery(dev); h = p._handle
u16, y, g = dev.create_table_buffers([2 * n, 4 * n, 4 * n])
capi.call("agpu_synth_u8", h, C.c_void_p(u16.ptr), 2 * n, 7, 0)
capi.call("agpu_synth_f32", h, C.c_void_p(y.ptr), n, 2, 0, C.c_float(-3.0), C.c_float(3.0)); p.sync()
S = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
class Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]
def chain(*items):
    arr = (Step * len(items))()
    for k, (op, kind, operand) in enumerate(items):
        arr[k].op, arr[k].kind, arr[k].operand = op, kind, (operand.ptr if operand is not None else None)
    return arr, len(items)
vp = lambda b: C.c_void_p(b.ptr)
c1, n1 = chain((capi.OP_MUL, 1, S), (capi.UN_SIN, 0, None))
c2, n2 = chain((capi.OP_MUL, 1, S), (capi.OP_ADD, 2, y), (capi.UN_COS, 0, None))
c3, n3 = chain((capi.OP_MUL, 1, S), (capi.UN_EXP, 0, None))
K = {"cast(u16)*s->sin": (c1, n1, 6.0), "cast(u16)*s+y->cos": (c2, n2, 10.0), "cast(u16)*s->exp": (c3, n3, 6.0)}
def med(fn, bpr):
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return bpr * n / sorted(ts)[4] / 1e6 / 8000
for name, (c, nc, bpr) in K.items():
    row = []
    for k in (0, 1, 2, 4, 8, 16, 0, 2):
        p.set_tuning("cast_tiles", k)
        row.append(f"{k}:{med(lambda: capi.call('agpu_fused_cast_chain', h, capi.U16, vp(u16), C.cast(c, C.c_void_p), nc, vp(g), n), bpr):.3f}")
    print(name.ljust(20), " ".join(row), flush=True)
