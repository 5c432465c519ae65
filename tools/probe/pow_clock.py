#!/usr/bin/env python3
"""Why did f32 pow swing between 0.69 and 0.79 of the HBM roof in round 1?  pow is the one VALU-heavy kernel of the table:
time 3 × 20 launches of pow (back to back / after 2 s of idle / interleaved with the memory-bound add) with HIP events,
sample sclk + socket power around every group (rocm-smi), and — when run under `rocprofv3 --pmc GRBM_GUI_ACTIVE` —
let tools/probe/pow_clock_pmc.py turn busy cycles ÷ dispatch duration into the effective shader clock of every launch.
Writes gpurun_out/pow_clock.json."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice  # noqa: E402

n = 1_000_000_000
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "pow")
q = CmpQuery(dev)
h = p._handle
vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
A, B, O = (dev.create_empty_buffer(4 * n) for _ in range(3))
capi.call("agpu_synth_f32", h, vp(A), n, 1, 0, C.c_float(0.001), C.c_float(1000))
capi.call("agpu_synth_f32", h, vp(B), n, 2, 0, C.c_float(-8), C.c_float(8))
p.sync()


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        card = next(iter(d.values()))
        return {k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:120]}


def pow_():
    capi.call("agpu_binary", h, capi.OP_POW, capi.F32, vp(A), vp(B), vp(O), n)


def add_():
    capi.call("agpu_binary", h, capi.OP_ADD, capi.F32, vp(A), vp(B), vp(O), n)


def timed(f):
    q.begin(p)
    f()
    q.end(p)
    return round(q.wait_for_results(), 4)


res = {"rows": n, "groups": []}
pow_(); add_(); p.sync()
for label, seq, idle in (("pow x20 back to back", [pow_] * 20, 0.0), ("pow x20 after 2 s idle", [pow_] * 20, 2.0),
                         ("add,pow interleaved x10", [add_, pow_] * 10, 0.0), ("add x20 back to back", [add_] * 20, 0.0)):
    if idle:
        time.sleep(idle)
    before = smi()
    ts = [timed(f) for f in seq]
    after = smi()
    g = {"group": label, "ms": ts, "smi_before": before, "smi_after": after}
    if "interleaved" in label:
        g["add_ms"], g["pow_ms"] = ts[0::2], ts[1::2]
    res["groups"].append(g)
    print(label, "min/median/max ms:", min(ts), float(np.median(ts)), max(ts), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
if "--no-json" not in sys.argv:
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "pow_clock.json"), "w"), indent=1)
