import numpy as np, sys, ctypes as C
sys.path.insert(0, ".")
import oracle as O
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "d"); h = p._handle
rng = np.random.default_rng(1)
for n in (1003, 512, 1024, 5000):
    x = rng.integers(-32768, 32767, n).astype(np.int16)
    xb = dev.create_gpu_buffer_with_data(x); ob = dev.create_empty_buffer(4 * n + 64)
    for name, call, exp in (("cast", lambda: capi.call("agpu_cast", h, capi.I16, capi.F32, C.c_void_p(xb.ptr), C.c_void_p(ob.ptr), n), x.astype(np.float32)),
                            ("sinh", lambda: capi.call("agpu_unary", h, capi.UN_SINH, capi.I16, C.c_void_p(xb.ptr), C.c_void_p(ob.ptr), n), O.unary(O.UN_SINH, O.I16, x))):
        capi.call("agpu_memset", h, C.c_void_p(ob.ptr), 0xAB, 4 * n)
        call()
        got = dev.retrive_data(ob, 4 * n, pipeline=p).view(np.float32)
        bad = np.nonzero(got.view(np.uint32) != exp.view(np.uint32))[0]
        print(n, name, "mismatches", len(bad), bad[:10], got[bad[:4]], exp[bad[:4]])
