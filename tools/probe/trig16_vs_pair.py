#!/usr/bin/env python3
"""DEV TOOL (GPU box): is the fused 16-bit trig kernel (trig16_kernel: f64 table form) bit-identical to cast → sin_f32 on every
16-bit input?  Decides whether agpu_fused_cast_chain may route `cast(u16) → sin` to it.  Prints the number of differing inputs."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import arrow_gpu_amd as ag  # noqa: E402

dev = ag.GPU_DEVICE()
out = {}
for name, cls, npd in (("u16", ag.UInt16ArrayGPU, np.uint16), ("i16", ag.Int16ArrayGPU, np.int16)):
    info = np.iinfo(npd)
    a = cls.from_slice(np.arange(info.min, int(info.max) + 1, dtype=np.int64).astype(npd), dev)
    f = a.cast(ag.Float32ArrayGPU)
    for fn in ("sin", "cos"):
        fused = getattr(a, fn)().raw_values()
        pair = getattr(f, fn)().raw_values()
        d = np.abs(fused.view(np.int32).astype(np.int64) - pair.view(np.int32).astype(np.int64))
        out[f"{fn}_{name}"] = {"differing_inputs": int(np.count_nonzero(d)), "max_ulp": int(d.max())}
print(json.dumps(out))
