# A/B of bench.py's headline step under allocator settings (one box, alternating fresh processes)
run() { python bench.py --no-cpu-baseline --no-traffic --steps 10 $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['extra']['kernels']; lp=d['extra'].get('layout_pool',{}); print('$1 $2', d['value'], 'table add/eq', e['add_f32']['frac_hbm_peak'], e['eq_i32_validity']['frac_hbm_peak'], '| pool blocks add/eq', lp.get('add_frac_hbm_peak'), lp.get('eq_frac_hbm_peak'))"; }
for i in 1 2 3; do
  run "default" ""
  run "noarena" "--tune pool_arena=0"
done
