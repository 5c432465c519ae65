set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bucketed.py -x -q -m gpu 2>&1 | tail -5
for w in 8 4; do
  echo "== G2 waves/EU $w"
  AGPU_TK2_G_WPE=$w python tools/probe/take_passes.py 268435456 7 2>&1 | grep "take_mergeback\|take_pairs"
done
