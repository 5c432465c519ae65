#!/bin/bash
# DEV TOOL (GPU box): bench.py's config rows with the prefetching tile loops at their defaults vs switched off (tiles per block = 1), alternating
cd "$(dirname "$0")/../.."
for i in 1 2; do
  for mode in default off; do
    if [ $mode = off ]; then T="--tune cast_tiles=1 --tune heavy_tiles=1 --tune table_tiles=1"; else T=""; fi
    python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['extra']['configs']
print('$mode', {k:v['frac_hbm_peak'] for k,v in e.items() if isinstance(v,dict) and k in ('cast_u8_f32','sin_f32','cos_f32','sin_u8','cos_u8','cast_u8_f32_then_sin_one_launch')}, {k:v['frac_hbm_peak'] for k,v in d['extra']['fused'].items() if isinstance(v,dict)})"
  done
done
