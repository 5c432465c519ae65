#!/bin/bash
# round 6: suite + exhaustive sin / cos + kernel table after the bit-cast fix; five fresh bench processes (host-API leg)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=10 > $O/second_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/second_suite.log
tail -5 $O/second_suite.log
timeout 1500 python tests/tools/exhaustive_vs_oracle.py sin cos > $O/second_exhaustive.log 2>&1
echo "exhaustive rc=$?"; tail -4 $O/second_exhaustive.log
cp gpurun_out/r03_exhaustive_vs_oracle.json $O/exhaustive_sincos.json 2>/dev/null
timeout 1200 python tools/kernel_table.py --tag r06a > $O/second_table.log 2>&1
echo "table rc=$?"; grep -i "sin\|cos" $O/second_table.log | head -30
for i in 1 2 3 4 5; do
  AGPU_ALLOC_TRACE=1 timeout 600 python bench.py --no-traffic > $O/second_bench_$i.json 2> $O/second_bench_$i.err
  python - $O/second_bench_$i.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["extra"]["kernels"]; lp=d["extra"]["layout_pool"]
print(sys.argv[1], d["value"], k["add_f32"]["frac_hbm_peak"], k["eq_i32_validity"]["frac_hbm_peak"], "host_api:", d["config"].get("host_api"), lp.get("value_GBps"), lp.get("error"), lp.get("add_ms_per_launch"), lp.get("eq_ms_per_launch"))
print("   parity:", d.get("gpu_parity"), "| reduce verified:", d["extra"]["reduce_sum_min_max"].get("verified"), d["extra"]["reduce_sum_min_max"].get("verified_rank0"), d["extra"]["reduce_sum_min_max"].get("error"))
P
done
