#!/bin/bash
# round 6, first contact of the packed-f32 sin / cos and of the segregated pool: suite, every f32 pattern of sin / cos against the oracle,
# the kernel table, three fresh bench processes (host-API leg = extra.layout_pool)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=10 > $O/first_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/first_suite.log
tail -5 $O/first_suite.log
timeout 1500 python tests/tools/exhaustive_vs_oracle.py sin cos > $O/first_exhaustive.log 2>&1
echo "exhaustive rc=$?"; tail -4 $O/first_exhaustive.log
cp gpurun_out/r03_exhaustive_vs_oracle.json $O/exhaustive_sincos.json 2>/dev/null
timeout 1200 python tools/kernel_table.py --tag r06a > $O/first_table.log 2>&1
echo "table rc=$?"; tail -3 $O/first_table.log
for i in 1 2 3; do
  AGPU_ALLOC_TRACE=1 timeout 600 python bench.py --no-traffic > $O/first_bench_$i.json 2> $O/first_bench_$i.err
  python - $O/first_bench_$i.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["extra"]["kernels"]
print(sys.argv[1], d["value"], k["add_f32"]["frac_hbm_peak"], k["eq_i32_validity"]["frac_hbm_peak"], "host_api:", d["config"].get("host_api"), d["extra"]["layout_pool"].get("value_GBps"), d["extra"]["layout_pool"].get("error"))
P
done
