#!/usr/bin/env python3
"""DEV TOOL: put of 2^28 random rows (pair pipeline) in fresh processes.  Round 4 used it with a probe build whose spacing of G's reservation
cursors came from AGPU_PROBE_CUR_STRIDE (1 = packed: 5.01-5.05 ms; 8 / 16 / 32 words: 4.92-4.95 ms; the product now uses 8 beyond 1024 regions)."""
import ctypes as C, os, sys, subprocess, json
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    import numpy as np
    from arrow_gpu_amd import _capi as capi
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
    dev = GpuDevice(0); p = ArrowComputePipeline(dev, "put"); q = CmpQuery(dev); h = p._handle
    vp = lambda b: C.c_void_p(b.ptr)
    n = 1 << 28
    values, out, idx, idx2 = (dev.create_empty_buffer(4 * n) for _ in range(4))
    capi.call("agpu_synth_i32", h, vp(values), n, 1, 0, 0); capi.call("agpu_synth_i32", h, vp(idx), n, 2, 0, n); capi.call("agpu_synth_i32", h, vp(idx2), n, 3, 0, n)
    p.sync(); p.set_tuning("gather_bucket", 2)
    f = lambda: capi.call("agpu_put_bounded", h, 4, vp(values), n, vp(idx), vp(out), n, vp(idx2), n)
    f(); f(); p.sync(); ts = []
    for _ in range(7):
        q.begin(p); f(); q.end(p); ts.append(q.wait_for_results())
    print(f"stride {os.environ.get('AGPU_PROBE_CUR_STRIDE','default')}: put {float(np.median(ts)):.4f} ms")
else:
    for s in ("", "1", "2", "4", "8", "16", "32", "", "8", "32"):
        env = dict(os.environ); 
        if s: env["AGPU_PROBE_CUR_STRIDE"] = s
        else: env.pop("AGPU_PROBE_CUR_STRIDE", None)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-300:], flush=True)
