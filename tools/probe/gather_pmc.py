#!/usr/bin/env python3
"""One variant of take / put at 2^28 uniformly random rows, a few launches, for rocprofv3 --pmc passes
(tools/profile_gather.sh).  Usage: gather_pmc.py {take,put,takebits,putbits}_{direct,bucketed} | take_pairs [log2_rows]
take_bucketed = the merge-back pipeline (round 3), take_pairs = the pair pipeline (tuning gather_bucket = 3)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice  # noqa: E402

variant = sys.argv[1]
n = (int(float(sys.argv[2])) if len(sys.argv) > 2 and float(sys.argv[2]) > 64 else 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 28))  # log2 or a row count
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "pmc")
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
values, out, idx, idx2 = (dev.create_empty_buffer(4 * n) for _ in range(4))
capi.call("agpu_synth_i32", h, vp(values), n, 1, 0, 0)
capi.call("agpu_synth_i32", h, vp(idx), n, 2, 0, n)
capi.call("agpu_synth_i32", h, vp(idx2), n, 3, 0, n)
p.set_tuning("gather_bucket", 3 if variant.endswith("pairs") else 2 if variant.endswith("bucketed") else 1)
bits_a, bits_b = dev.create_empty_buffer(n // 8 + 64), dev.create_empty_buffer(n // 8 + 64)
if "bits" in variant:
    capi.call("agpu_synth_bits", h, vp(bits_a), n, 7, 0, C.c_double(0.5))
    capi.call("agpu_synth_bits", h, vp(bits_b), n, 8, 0, C.c_double(0.5))
for _ in range(3):
    if variant.startswith("takebits"):
        capi.call("agpu_take_bits", h, vp(bits_a), n, vp(idx), vp(bits_b), n)
    elif variant.startswith("putbits"):
        capi.call("agpu_put_bits_bounded", h, vp(bits_a), n, vp(idx), vp(bits_b), n, vp(idx2), n)
    elif variant.startswith("take"):
        capi.call("agpu_take", h, 4, vp(values), n, vp(idx), vp(out), n)
    else:
        capi.call("agpu_put_bounded", h, 4, vp(values), n, vp(idx), vp(out), n, vp(idx2), n)
p.sync()
print(variant, "done", flush=True)
