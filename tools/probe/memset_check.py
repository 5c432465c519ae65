import ctypes as C, numpy as np, sys
sys.path.insert(0, '/root/repo')
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "c"); q = CmpQuery(dev); h = p._handle
n = 1_000_000_000
O = dev.create_empty_buffer(4 * n)
for label, f in (("agpu_memset 4 GB", lambda: capi.call("agpu_memset", h, C.c_void_p(O.ptr), 0, 4 * n)),
                 ("agpu_broadcast f32 4 GB", lambda: capi.call("agpu_broadcast", h, capi.F32, 0, C.c_void_p(O.ptr), n)),
                 ("agpu_memset 125 MB", lambda: capi.call("agpu_memset", h, C.c_void_p(O.ptr), 0, 125_000_000)),
                 ("agpu_memset 4 KiB", lambda: capi.call("agpu_memset", h, C.c_void_p(O.ptr), 0, 4096))):
    f(); p.sync(); ts = []
    for _ in range(7):
        q.begin(p); f(); q.end(p); ts.append(q.wait_for_results())
    print(label, round(float(np.median(ts)), 4), "ms")
