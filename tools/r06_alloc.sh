#!/bin/bash
# Round 6, VERDICT r5 item 1: where do an ordinary caller's blocks come from, and what do the headline kernels make of them?
# bench.py (layout_pool leg) with the allocation trace, arenas on / off, fresh processes; then the host-API probe.
mkdir -p gpurun_out/r06_alloc
O=gpurun_out/r06_alloc
export AGPU_ALLOC_TRACE=1
for i in 1 2 3; do
  python bench.py --no-traffic > $O/bench_arena1_$i.json 2> $O/bench_arena1_$i.err
  python bench.py --no-traffic --tune pool_arena=0 > $O/bench_arena0_$i.json 2> $O/bench_arena0_$i.err
done
for i in 1 2; do
  for arena in 1 0; do
    for churn in 0 1; do
      timeout 300 python tools/probe/r06_api_alloc.py --arena $arena --churn $churn --reps 4 > $O/api_a${arena}_c${churn}_$i.jsonl 2> $O/api_a${arena}_c${churn}_$i.err
    done
  done
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_alloc/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); lp=d["extra"]["layout_pool"]; k=d["extra"]["kernels"]
        print(f, d["value"], k["add_f32"]["frac_hbm_peak"], k["eq_i32_validity"]["frac_hbm_peak"], "pool:", lp.get("add_frac_hbm_peak"), lp.get("eq_frac_hbm_peak"))
    except Exception as e: print(f, "ERR", e)
for f in sorted(glob.glob("gpurun_out/r06_alloc/api_*.jsonl")):
    for l in open(f):
        try:
            d=json.loads(l); print(f, d["rep"], d["add_frac"], d["eq_frac"])
        except Exception: pass
P
