#!/bin/bash
# round 6: ONE box, one call — the whole GPU suite (timing expectations asserted), smoke, the plain bench line, five more fresh bench
# processes (config.host_api), the bench and the kernel table under rocprofv3 (trace + the two PMC passes each), every f32 pattern of sin / cos
# against the oracle.  Outputs under gpurun_out/; the summaries are copied into profiles/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/probe/box_fingerprint.sh > gpurun_out/r06_box_evidence.txt 2>&1; tail -1 gpurun_out/r06_box_evidence.txt
export AGPU_PERF_STRICT=1
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=10 > gpurun_out/r06_gpu_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a gpurun_out/r06_gpu_suite.log
unset AGPU_PERF_STRICT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/r06_smoke.log
timeout 900 python bench.py > gpurun_out/r06_bench_plain.json 2> gpurun_out/r06_bench_plain.err
echo "bench rc=$?"
for i in 1 2 3 4 5; do timeout 600 python bench.py --no-traffic > gpurun_out/r06_bench_repeat_$i.json 2> gpurun_out/r06_bench_repeat_$i.err; done
timeout 1500 python tests/tools/exhaustive_vs_oracle.py log sin cos > gpurun_out/r06_exhaustive.log 2>&1
echo "exhaustive rc=$?"; cp gpurun_out/r03_exhaustive_vs_oracle.json gpurun_out/r06_exhaustive_sincos.json
timeout 1500 bash tools/profile_bench.sh r06 5 > gpurun_out/r06_profile_bench.log 2>&1
echo "profile_bench rc=$?"
timeout 2400 bash tools/profile_table.sh r06 > gpurun_out/r06_profile_table.log 2>&1
echo "profile_table rc=$?"
grep -E "passed|failed" gpurun_out/r06_gpu_suite.log | tail -2
python - <<'P'
import json,glob
for f in ["gpurun_out/r06_bench_plain.json"]+sorted(glob.glob("gpurun_out/r06_bench_repeat_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["extra"]["kernels"]
        print(f, d["value"], k["add_f32"]["frac_hbm_peak"], k["eq_i32_validity"]["frac_hbm_peak"], d["config"].get("host_api"), d.get("gpu_parity","")[:40], d["extra"]["reduce_sum_min_max"].get("verified"), d["roofline"].get("traffic"))
    except Exception as e: print(f,"ERR",e)
P
