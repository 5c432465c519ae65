import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "s", fuse=False); h = p._handle; q = CmpQuery(dev)
vp = lambda b: C.c_void_p(b.ptr)
A, = dev.create_table_buffers([4 * n]); V = dev.create_empty_buffer((n + 63) // 64 * 8); R = dev.create_empty_buffer(64)
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(-1), C.c_float(1)); capi.call("agpu_synth_bits", h, vp(V), n, 2, 0, C.c_double(0.9)); p.sync()
def med(f, reps=7):
    f(); f(); p.sync(); ts = []
    for _ in range(reps):
        q.begin(p); f(); q.end(p); ts.append(q.wait_for_results())
    return round(float(np.median(ts)), 4)
print(json.dumps({"one pass, null-aware ms": med(lambda: capi.call("agpu_reduce_stats_f32", h, vp(A), vp(V), n, vp(R))),
  "sum ms": med(lambda: capi.call("agpu_reduce", h, capi.RED_SUM, capi.F32, vp(A), vp(V), n, vp(R))),
  "min ms": med(lambda: capi.call("agpu_reduce", h, capi.RED_MIN, capi.F32, vp(A), vp(V), n, vp(R))),
  "max ms": med(lambda: capi.call("agpu_reduce", h, capi.RED_MAX, capi.F32, vp(A), vp(V), n, vp(R))),
  "f64 ms": med(lambda: capi.call("agpu_reduce_sum_f64", h, vp(A), vp(V), n, vp(R))),
  "one pass, no validity ms": med(lambda: capi.call("agpu_reduce_stats_f32", h, vp(A), None, n, vp(R)))}))
