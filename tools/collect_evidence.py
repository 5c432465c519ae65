#!/usr/bin/env python3
"""After tools/r06_evidence.sh (run through gpurun; its outputs land in gpurun_out/): copy the summaries the docs cite into profiles/.
   python tools/collect_evidence.py r06"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def line_of(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


runs = []
for f in [f"{rnd}_bench_plain.json"] + sorted(os.path.basename(x) for x in glob.glob(os.path.join(G, f"{rnd}_bench_repeat_*.json"))):
    d = line_of(os.path.join(G, f))
    k = d["extra"]["kernels"]
    runs.append({"file": f, "value_GBps": d["value"], "add_frac": k["add_f32"]["frac_hbm_peak"], "eq_frac": k["eq_i32_validity"]["frac_hbm_peak"],
                 "host_api": d["config"].get("host_api"), "host_api_step_GBps": d["extra"].get("layout_pool", {}).get("value_GBps"),
                 "gpu_parity": d.get("gpu_parity"), "reduce_verified": d["extra"]["reduce_sum_min_max"].get("verified"),
                 "reduce_fracs": {s: v["frac_hbm_peak"] for s, v in d["extra"]["reduce_sum_min_max"].get("per_statistic", {}).items()},
                 "traffic": d["roofline"].get("traffic"), "allocation_class": d["extra"]["layout"]["allocation_class"]})
json.dump({"what": f"tools/{rnd}_evidence.sh: the plain bench line and five more fresh processes on the same box (final evidence call of the round)",
           "runs": runs}, open(os.path.join(P, f"{rnd}_bench_repeats.json"), "w"), indent=1)
shutil.copy(os.path.join(G, f"{rnd}_bench_plain.json"), os.path.join(P, f"{rnd}_bench_plain.json"))
shutil.copy(os.path.join(G, f"{rnd}_box_evidence.txt"), os.path.join(P, f"{rnd}_box_evidence.txt"))
suite = [ln.strip() for ln in open(os.path.join(G, f"{rnd}_gpu_suite.log")) if " passed" in ln or " failed" in ln]
open(os.path.join(P, f"{rnd}_gpu_suite_result.txt"), "w").write("\n".join(suite[-1:]) + "\n")
ex = os.path.join(G, f"{rnd}_exhaustive_sincos.json")
if os.path.exists(ex):
    shutil.copy(ex, os.path.join(P, f"{rnd}_exhaustive_log_sin_cos.json"))
for r in runs:
    print(r["file"], r["value_GBps"], r["add_frac"], r["eq_frac"], r["host_api"], r["reduce_fracs"], r["allocation_class"])
print(open(os.path.join(P, f"{rnd}_gpu_suite_result.txt")).read())
