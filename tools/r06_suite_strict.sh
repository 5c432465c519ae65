#!/bin/bash
# round 6: the strict suite + the kernel table under rocprofv3 once more (the evidence call's table process drew a slow block: 0.80 on every three-array kernel)
cd "$(dirname "$0")/.."
export AGPU_PERF_STRICT=1
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=10 > gpurun_out/r06_gpu_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a gpurun_out/r06_gpu_suite.log
unset AGPU_PERF_STRICT
grep -E "passed|failed|FAILED|^E  " gpurun_out/r06_gpu_suite.log | head
grep "^\[perf\]" gpurun_out/r06_gpu_suite.log | cut -c1-300
cp gpurun_out/kernel_table_r06.json gpurun_out/kernel_table_r06_first_draw.json 2>/dev/null
timeout 2400 bash tools/profile_table.sh r06 > gpurun_out/r06_profile_table.log 2>&1
echo "profile_table rc=$?"
python - <<'P'
import json
d=json.load(open("gpurun_out/kernel_table_r06.json"))
for x in d["kernels"][:3]+[y for y in d["kernels"] if y["kernel"] in ("f32 sin","f32 log","fused sin_u16","i32 eq → bitmap + validity AND (fused)")]: print(x["kernel"], x["frac_8TBs"])
P
