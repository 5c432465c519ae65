import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "st")
for name in ("SQRT", "CBRT", "EXP", "EXP2", "LOG", "LOG2", "SIN", "COS", "ACOS", "SINH"):
    mx, w = C.c_uint32(0), C.c_uint32(0)
    t = time.time()
    capi.call("agpu_selftest_unary_f32", p._handle, getattr(capi, "UN_" + name), 0, 1 << 32, C.byref(mx), C.byref(w))
    x = np.array([w.value], np.uint32).view(np.float32)[0]
    print(f"{name:5s} max {mx.value} ULP at {w.value:#010x} = {x!r}   {time.time() - t:.2f} s")
