import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = b["extra"]["kernels"]
print(b["value"], "add", k["add_f32"]["frac_hbm_peak"], "eq+v", k["eq_i32_validity"]["frac_hbm_peak"])
