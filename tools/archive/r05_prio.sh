#!/bin/bash
# Round 5, GPU box: wave priority around the load phase of sin / cos / sinh / log (build variants -DAGPU_EW_PRIO=1: loads at priority 3,
# compute at 0; =2: the reverse), with and without the occupancy cap
set -u
mkdir -p gpurun_out
for lib in "" tools/probe/variants/libagpu_prio1.so tools/probe/variants/libagpu_prio2.so "" tools/probe/variants/libagpu_prio1.so; do
  echo "== AGPU_LIB=$lib"
  AGPU_LIB=${lib:+$PWD/$lib} python - <<'PY' 2>&1 | tee -a gpurun_out/r05_prio.txt
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "prio"); q = CmpQuery(dev); h = p._handle
p.set_tuning("tile_auto", 1)
f, g = dev.create_table_buffers([4 * n, 4 * n])
capi.call("agpu_synth_f32", h, C.c_void_p(f.ptr), n, 1, 0, C.c_float(0.001), C.c_float(1000.0)); p.sync()
def med(op):
    fn = lambda: capi.call("agpu_unary", h, op, capi.F32, C.c_void_p(f.ptr), C.c_void_p(g.ptr), n)
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return 8.0 * n / sorted(ts)[4] / 1e6 / 8000
for name, op in (("sin", capi.UN_SIN), ("cos", capi.UN_COS), ("sinh", capi.UN_SINH)):
    row = []
    for cap in (-1, 0, -1, 0):
        p.set_tuning("wave_lds", cap); row.append(f"{'capped' if cap == 0 else 'uncapped'} {med(op):.3f}")
    print(name, "  ".join(row), flush=True)
PY
done
