"""Helpers for the -m gpu parity tests: thin numpy <-> HBM plumbing around the raw C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

import oracle as O
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

SIZES = [0, 1, 3, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 65535, 65536, 65537, 262147, 1048576 + 3]
SMALL_SIZES = [0, 1, 5, 64, 257, 4097, 65537, 300001]

NP = O.NP_DTYPE
INT_DTYPES = [capi.I32, capi.U32, capi.I16, capi.U16, capi.I8, capi.U8]
ALL_DTYPES = [capi.F32] + INT_DTYPES


class Dev:
    def __init__(self):
        self.dev = GpuDevice(0)
        self.p = ArrowComputePipeline(self.dev, "tests")
        self._keep = []  # buffers stay alive until release(): `D.up(x).vp` temporaries must not be freed before the launch

    @property
    def h(self):
        return self.p._handle

    def up(self, arr: np.ndarray, offset_bytes: int = 0):
        """Upload; with offset_bytes the returned pointer is deliberately mis-aligned by that much."""
        arr = np.ascontiguousarray(arr)
        buf = self.dev.create_empty_buffer(arr.nbytes + offset_bytes + 16)
        if arr.nbytes:
            capi.call("agpu_upload", self.h, C.c_void_p(buf.ptr + offset_bytes), C.c_void_p(arr.ctypes.data), arr.nbytes)
        return self._hold(_Ptr(buf, offset_bytes))

    def _hold(self, ptr):
        self._keep.append(ptr)
        return ptr

    def release(self):
        self.p.sync()
        self._keep.clear()

    def empty(self, nbytes: int, offset_bytes: int = 0, fill: int | None = 0xCD):
        buf = self.dev.create_empty_buffer(nbytes + offset_bytes + 16)
        if fill is not None:
            capi.call("agpu_memset", self.h, C.c_void_p(buf.ptr), fill, nbytes + offset_bytes + 16)
        return self._hold(_Ptr(buf, offset_bytes))

    def down(self, ptr, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        capi.call("agpu_download", self.h, C.c_void_p(out.ctypes.data), ptr.vp, out.nbytes)
        return out

    def call(self, name, *args):
        capi.call(name, self.h, *args)

    def status(self, name, *args) -> int:
        return getattr(capi.lib(), name)(self.h, *args)


class _Ptr:
    def __init__(self, buf, off):
        self.buf = buf
        self.off = off

    @property
    def vp(self):
        return C.c_void_p(self.buf.ptr + self.off)


def rand_values(dtype: int, n: int, seed: int, special: bool = True) -> np.ndarray:
    rng = np.random.default_rng(seed)
    npd = NP[dtype]
    if dtype == capi.F32:
        x = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)).astype(np.float32)
        if special and n >= 16:
            sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 3.4e38, -3.4e38, 1e-40, -1e-40, 1.17549435e-38,
                           255.0, 256.0, 0.5, -0.5], dtype=np.float32)
            pos = rng.choice(n, size=len(sp), replace=False)
            x[pos] = sp
        return x
    info = np.iinfo(npd)
    x = rng.integers(info.min, int(info.max) + 1, n, dtype=np.int64).astype(npd)
    if special and n >= 8:
        sp = np.array([0, 1, info.max, info.min, info.max - 1, 0, 2, 3]).astype(npd)
        pos = rng.choice(n, size=len(sp), replace=False)
        x[pos] = sp
    # make values collide often so eq / min / max see ties
    if n:
        x[rng.integers(0, n, n // 4)] = x[rng.integers(0, n, n // 4)]
    return x


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def nan_aware_bits_equal(got: np.ndarray, exp: np.ndarray) -> bool:
    """f32 results: identical bits, except that any NaN payload matches any NaN."""
    if got.dtype != np.float32:
        return bits_equal(got, exp)
    gn, en = np.isnan(got), np.isnan(exp)
    if not np.array_equal(gn, en):
        return False
    return bits_equal(got[~gn], exp[~en])


def max_ulp(got: np.ndarray, exp: np.ndarray) -> int:
    fin = np.isfinite(exp) & np.isfinite(got)
    if not np.array_equal(np.isnan(exp), np.isnan(got)):
        return 1 << 30
    inf = np.isinf(exp)
    if not np.array_equal(got[inf], exp[inf]):
        return 1 << 30
    if not fin.any():
        return 0
    g = got[fin].view(np.int32).astype(np.int64)
    e = exp[fin].view(np.int32).astype(np.int64)
    g = np.where(g < 0, np.int64(-(2**31)) - g, g)
    e = np.where(e < 0, np.int64(-(2**31)) - e, e)
    return int(np.abs(g - e).max())
