set -u
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r03_pytest_full.log
echo "pytest rc=$?"; tail -6 gpurun_out/r03_pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r03_bench_plain.json 2> gpurun_out/r03_bench_plain.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_plain.json'))
e=d['extra']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], e['kernels'])
print('layout_pool', e.get('layout_pool'))
print({k:(v['frac_hbm_peak'] if isinstance(v,dict) else v) for k,v in e['configs'].items() if k!='what'})
print(d['cpu_baseline'].get('gpu_parity'), d['cpu_baseline'].get('configs_parity'))
PY
