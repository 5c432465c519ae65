#!/bin/bash
# Round 5, GPU box: waves per CU capped with unused dynamic LDS (AGPU_DYN_LDS bytes per wave: 0 → 32 waves, 6800 → 24, 10240 → 16, 20480 → 8)
# × tiles per block — is it the doubled number of loads in flight that costs the two-tile form its 8 % on most boxes?
set -u
mkdir -p gpurun_out
bash tools/probe/box_fingerprint.sh > gpurun_out/r05_box_occ.txt 2>&1; tail -1 gpurun_out/r05_box_occ.txt
for L in 0 6800 10240 20480 0 10240; do
  echo "== AGPU_DYN_LDS=$L"
  AGPU_DYN_LDS=$L PREFETCH_KS=1,2,3,4,1,2 timeout 600 python tools/probe/prefetch_sweep.py 2>&1 >> gpurun_out/r05_occ_$L.jsonl | grep -E '^(sin_f32|cos_f32|sinh_f32|cast_u8_f32|cast_u16_f32) ' | tee -a gpurun_out/r05_occ_$L.txt
done
