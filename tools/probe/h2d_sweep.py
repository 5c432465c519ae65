#!/usr/bin/env python3
"""Host↔HBM staging: GB/s of agpu_staged_copy in its three modes (pageable hipMemcpy / threaded page-locked chunks /
hipHostRegister in place) against the pinned-to-device link rate, and the end-to-end rate of an overlapped f32 add fed
from and returned to pageable host memory (interop.map_chunks).  Writes gpurun_out/h2d_sweep.json."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import arrow_gpu_amd as ag  # noqa: E402
from arrow_gpu_amd import _capi as capi  # noqa: E402

dev = ag.GpuDevice(0)
p = ag.ArrowComputePipeline(dev, "h2d")
h = p._handle
res = {"staged_copy": [], "map_chunks": []}
nbytes = 1 << 30
host = np.random.default_rng(0).integers(0, 256, nbytes, dtype=np.uint8)
back = np.empty(nbytes, np.uint8)
buf = dev.create_empty_buffer(nbytes)

# reference point: page-locked memory → device, one async copy (the link itself)
pin = C.c_void_p()
capi.call("agpu_host_alloc", dev._handle, nbytes, C.byref(pin))
C.memmove(pin, host.ctypes.data, nbytes)
for d in ("h2d", "d2h"):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        capi.call("agpu_upload_async" if d == "h2d" else "agpu_download_async", h, C.c_void_p(buf.ptr) if d == "h2d" else pin,
                  pin if d == "h2d" else C.c_void_p(buf.ptr), nbytes)
        p.sync()
        ts.append(time.perf_counter() - t0)
    res[f"pinned_{d}_GBps"] = round(nbytes / min(ts) / 1e9, 1)
capi.call("agpu_host_free", dev._handle, pin)

for mode, name in ((1, "pageable hipMemcpy"), (2, "threaded pinned staging"), (3, "hipHostRegister in place")):
    for threads in ((0,) if mode != 2 else (1, 2, 4, 8, 16, 32)):
        p.set_tuning("h2d_mode", mode)
        p.set_tuning("h2d_threads", threads)
        rec = {"mode": mode, "name": name, "threads": threads}
        for d, to_dev, hp in (("h2d", 1, host), ("d2h", 0, back)):
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                capi.call("agpu_staged_copy", h, C.c_void_p(buf.ptr), C.c_void_p(hp.ctypes.data), nbytes, to_dev)
                p.sync()
                ts.append(time.perf_counter() - t0)
            rec[f"{d}_GBps"] = round(nbytes / min(ts) / 1e9, 1)
        rec["round_trip_ok"] = bool(np.array_equal(host[:: 4097], back[:: 4097]))
        print(rec, flush=True)
        res["staged_copy"].append(rec)
p.set_tuning("h2d_mode", 0)
p.set_tuning("h2d_threads", 0)

n = 1 << 28
a = np.random.default_rng(1).standard_normal(n).astype(np.float32)
b = np.random.default_rng(2).standard_normal(n).astype(np.float32)
out = np.empty(n, np.float32)


def launch(pp, ins, o, rows):
    capi.call("agpu_binary", pp._handle, capi.OP_ADD, capi.F32, C.c_void_p(ins[0].ptr), C.c_void_p(ins[1].ptr), C.c_void_p(o.ptr), rows)


for chunk in (1 << 22, 1 << 24, 1 << 26):
    info = ag.interop.map_chunks(dev, [a, b], out, chunk, launch)
    rec = {"rows": n, "chunk_rows": chunk, "seconds": round(info["seconds"], 4), "GBps_host_bytes": round(info["GBps_host_bytes"], 1),
           "ok": bool(np.array_equal(out[:: 9973], (a + b)[:: 9973]))}
    print(rec, flush=True)
    res["map_chunks"].append(rec)
# Arrow C Data Interface ingest / egress of a 1 GiB f32 column with 10 % nulls
import pyarrow as pa  # noqa: E402

mask = np.random.default_rng(3).random(n) < 0.1
arr = pa.array(a, mask=mask)
for _ in range(2):
    t0 = time.perf_counter()
    g = ag.from_arrow(arr, dev)
    t1 = time.perf_counter()
    back_arr = ag.to_arrow(g)
    t2 = time.perf_counter()
res["arrow_cdata_1GiB_f32_10pct_nulls"] = {"from_arrow_GBps": round((4 * n + n / 8) / (t1 - t0) / 1e9, 1), "to_arrow_GBps": round((4 * n + n / 8) / (t2 - t1) / 1e9, 1),
                                           "ok": bool(back_arr.equals(arr))}
print(res["arrow_cdata_1GiB_f32_10pct_nulls"], flush=True)
del g, back_arr
# the serial way (what the reference does: upload everything, compute, read back)
t0 = time.perf_counter()
ga = ag.Float32ArrayGPU.from_slice(a, dev)
gb = ag.Float32ArrayGPU.from_slice(b, dev)
r = ga.add(gb).raw_values()
res["serial_from_slice_add_raw_values"] = {"seconds": round(time.perf_counter() - t0, 4), "GBps_host_bytes": round(12.0 * n / (time.perf_counter() - t0) / 1e9, 1),
                                           "ok": bool(np.array_equal(r[:: 9973], (a + b)[:: 9973]))}
print(res["serial_from_slice_add_raw_values"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "h2d_sweep.json"), "w"), indent=1)
