#!/bin/bash
# round 6b: one-pass statistics — parity tests, the bench line's config-5 leg
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_stats.py -x -q -m gpu > gpurun_out/r06b_t5.log 2>&1; echo "stats tests rc=$?"; grep -E "passed|failed|^E " gpurun_out/r06b_t5.log | head -20
timeout 900 python bench.py --no-traffic > gpurun_out/r06b_bench_stats.json 2> gpurun_out/r06b_bench_stats.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06b_bench_stats.json").read().strip().splitlines()[-1])
r=d["extra"]["reduce_sum_min_max"]
print(d["value"], {k:v for k,v in r.items() if k in ("verified","four_statistics_with_final_reduce_ms","final_reduce_overhead_ms","error")})
print(r.get("one_pass")); print({k:v["local_ms"] for k,v in r.get("per_statistic",{}).items()})
P
tail -3 gpurun_out/r06b_bench_stats.err
