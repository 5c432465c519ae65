#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "shift or shl or shr or logical" 2>&1 | grep -E "passed|failed" 
python - <<'PY' 2>&1 | tee gpurun_out/r05_shift_ab.txt
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "shift"); q = CmpQuery(dev); h = p._handle
A, B, O = dev.create_table_buffers([4 * n] * 3)
capi.call("agpu_synth_i32", h, C.c_void_p(A.ptr), n, 1, 0, 1 << 30)
capi.call("agpu_synth_i32", h, C.c_void_p(B.ptr), n, 2, 0, 16); p.sync()
def med(dt, op, bpr):
    fn = lambda: capi.call("agpu_binary", h, op, dt, C.c_void_p(A.ptr), C.c_void_p(B.ptr), C.c_void_p(O.ptr), n)
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return bpr * n / sorted(ts)[4] / 1e6 / 8000
for name, dt, op, bpr in (("u8 shl", capi.U8, capi.OP_SHL, 6.0), ("i8 shr", capi.I8, capi.OP_SHR, 6.0), ("u16 shr", capi.U16, capi.OP_SHR, 8.0), ("i16 shl", capi.I16, capi.OP_SHL, 8.0)):
    row = []
    for blk in (256, 1, 256, 1):
        p.set_tuning("stream_unroll", blk); row.append(f"{'256-thread' if blk == 256 else 'one-wave'} {med(dt, op, bpr):.3f}")
    print(name, "  ".join(row), flush=True)
PY
bash tools/archive/r05_sincos_u.sh 2>&1 | grep -E '^(==|sin|cos)'
