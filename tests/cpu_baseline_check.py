"""Checks the timed CPU baseline loops (oracle/cpu_baseline.c) against the oracle at ragged sizes.  Run directly by
tests/test_cpu_baseline.py and, with CPU_BASELINE_LIB / ORACLE_LIB pointing at the sanitizer builds, by
tests/test_sanitizers.py under LD_PRELOAD=libasan."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402


def main():
    so = os.environ.get("CPU_BASELINE_LIB")
    if not so:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_build/libcpu_baseline.so"], check=True, capture_output=True)
        so = os.path.join(ROOT, "oracle", "_build", "libcpu_baseline.so")
    lib = C.CDLL(so)
    lib.base_sum_f32.restype = C.c_float
    p = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731
    for n in (0, 1, 63, 64, 65, 1000, 4099, 100_003):
        for threads in (1, 3):
            a, b = O.synth_f32(n, 1, 0, -1000.0, 1000.0), O.synth_f32(n, 2, 0, -1000.0, 1000.0)
            ia, ib = O.synth_i32(n, 3, 0, 16), O.synth_i32(n, 4, 0, 16)
            va, vb = O.synth_bits(n, 5, 0, 0.9), O.synth_bits(n, 6, 0, 0.9)
            nb = O.bitmap_bytes(n)
            out, ov, ob = np.empty(n, np.float32), np.zeros(max(nb, 8), np.uint8), np.zeros(max(nb, 8), np.uint8)
            lib.base_add_f32(p(a), p(b), p(out), p(va), p(vb), p(ov), C.c_uint64(n), threads)
            assert np.array_equal(out.view(np.uint32), O.binary(O.OP_ADD, O.F32, a, b).view(np.uint32)), n
            assert np.array_equal(ov[:nb], O.bitmap_binary(O.OP_AND, va, vb, n)), n
            ov2 = np.zeros(max(nb, 8), np.uint8)
            lib.base_eq_i32(p(ia), p(ib), p(ob), p(va), p(vb), p(ov2), C.c_uint64(n), threads)
            full = n // 8  # bits past n in the last byte are the baseline's own business
            assert np.array_equal(ob[:full], O.compare(O.CMP_EQ, O.I32, ia, ib)[:full]), n
            assert np.array_equal(ov2[:full], O.bitmap_binary(O.OP_AND, va, vb, n)[:full]), n
            s = lib.base_sum_f32(p(a), C.c_uint64(n), threads)
            ref = float(np.sum(a.astype(np.float64)))
            assert abs(float(s) - ref) <= 1e-3 * max(1.0, float(np.sum(np.abs(a.astype(np.float64))))), (n, s, ref)
    print("cpu_baseline OK")


if __name__ == "__main__":
    main()
