#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1800 python -m pytest tests/ -x -q -m gpu --durations=5 > $O/sixth_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/sixth_suite.log
grep -E "passed|failed|FAILED|^E  " $O/sixth_suite.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > $O/sixth_bench.json 2> $O/sixth_bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06/sixth_bench.json").read().strip().splitlines()[-1]); k=d["extra"]["kernels"]
print(d["value"], k["add_f32"]["frac_hbm_peak"], k["eq_i32_validity"]["frac_hbm_peak"], d["config"].get("host_api"), d.get("gpu_parity","")[:30], d["extra"]["reduce_sum_min_max"].get("verified"))
print({k2:v for k2,v in d["extra"]["configs"].items() if isinstance(v,dict) and "frac_hbm_peak" in v and ("sin" in k2 or "cos" in k2 or "cast" in k2)})
P
timeout 900 python tools/kernel_table.py --tag r06c 2>/dev/null | grep "^|" | grep -E "f32 sin|f32 cos|f32 log|cast u8→f32 \||cast i16|fused sin|cast u16|sinh|lut|\(x"
