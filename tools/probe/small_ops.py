#!/usr/bin/env python3
"""Host-side cost of `a.add(b)` (default API: new pipeline + new output + finish per call) through the PYTHON host at
100 rows and 1 Mi rows, resource pools on and off; runs tools/probe/small_ops (C++ host) beside it when it was built.
Writes gpurun_out/small_ops.json."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
res = {"what": "host-side microseconds per a.add(b) call; result dropped at once; GPU runs behind the host"}
exe = os.path.join(ROOT, "tools", "probe", "small_ops_cpp")
if os.path.exists(exe):  # BEFORE this process touches the GPU (a child of a GPU-initialised process may not exec)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    try:
        res["cpp_host"] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        res["cpp_host"] = {"error": (r.stdout + r.stderr)[-300:]}
import arrow_gpu_amd as ag  # noqa: E402
from arrow_gpu_amd import _capi as capi  # noqa: E402

dev = ag.GPU_DEVICE()
py = {}
for n in (100, 1 << 20):
    a = ag.Int32ArrayGPU.from_slice(np.arange(n, dtype=np.int32), dev)
    b = ag.Int32ArrayGPU.from_slice(np.arange(n, dtype=np.int32), dev)
    for pool in (1, 0):
        capi.call("agpu_set_tuning", b"mem_pool", pool)
        for _ in range(20):
            a.add(b)
        dev.sync()
        reps = 2000 if pool else 200
        t0 = time.perf_counter()
        for _ in range(reps):
            a.add(b)
        dt = time.perf_counter() - t0
        dev.sync()
        py[f"n{n}_pool{pool}_us"] = round(dt / reps * 1e6, 2)
    capi.call("agpu_set_tuning", b"mem_pool", 1)
    assert a.add(b).raw_values()[-1] == 2 * (n - 1)
res["python_host"] = py
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "small_ops.json"), "w"), indent=1)
