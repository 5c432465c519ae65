#!/usr/bin/env python3
"""DEV TOOL: ULP error of the f32 device-library functions vs the oracle (f64 libm rounded), on the GPU box."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle as O
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from gpu_util import max_ulp
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools", "probe", "libmath_probe.so"))
lib.probe_math.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "m")
rng = np.random.default_rng(5); n = 1 << 22
def rnd(lo, hi, signed=True):
    x = 2.0 ** rng.uniform(lo, hi, n)
    if signed: x *= rng.choice([-1.0, 1.0], n)
    return x.astype(np.float32)
cases = [("sinhf", 0, rnd(-20, 6), O.UN_SINH), ("acosf", 1, rnd(-20, 0), O.UN_ACOS), ("cbrtf", 3, rnd(-30, 30), O.UN_CBRT),
         ("exp2f", 4, rnd(-20, 6), O.UN_EXP2), ("log2f", 5, rnd(-30, 30, False), O.UN_LOG2), ("logf", 6, rnd(-30, 30, False), O.UN_LOG),
         ("expf", 7, rnd(-20, 6), O.UN_EXP)]
for name, fn, x, oop in cases:
    dx = dev.create_gpu_buffer_with_data(x); do = dev.create_empty_buffer(4 * n)
    lib.probe_math(C.c_void_p(dx.ptr), C.c_void_p(dx.ptr), C.c_void_p(do.ptr), n, fn, C.c_void_p(p.stream()))
    got = dev.retrive_data(do, pipeline=p)[: 4 * n].view(np.float32)
    print(name, "max ulp", max_ulp(got, O.unary(oop, O.F32, x)), flush=True)
a = np.abs(rnd(-6, 6)); b = rnd(-3, 3)
da, db, do = dev.create_gpu_buffer_with_data(a), dev.create_gpu_buffer_with_data(b), dev.create_empty_buffer(4 * n)
lib.probe_math(C.c_void_p(da.ptr), C.c_void_p(db.ptr), C.c_void_p(do.ptr), n, 2, C.c_void_p(p.stream()))
got = dev.retrive_data(do, pipeline=p)[: 4 * n].view(np.float32)
print("powf max ulp", max_ulp(got, O.binary(O.OP_POW, O.F32, a, b)))
