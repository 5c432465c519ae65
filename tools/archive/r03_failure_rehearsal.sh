#!/bin/bash
# On the 1-GPU box: what happens when a rank cannot come up?  torchrun starts TWO workers; local rank 1 has no device, so it
# fails at once; rank 0 must notice within the rendezvous deadline and exit non-zero — no hang inside RCCL.  Also: the weak
# run's strong-scaling leg (a world > 1 code path) forced at world 1, and the bitmap-kernel block-size A/B.
set -u
mkdir -p gpurun_out
t0=$(date +%s)
timeout 180 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --rendezvous-timeout 10 > gpurun_out/r03_failed_rank_stdout.txt 2> gpurun_out/r03_failed_rank_stderr.txt
rc=$?
t1=$(date +%s)
echo "{\"what\": \"torchrun --nproc-per-node 2 on a 1-GPU box: rank 1 has no device\", \"exit_code\": $rc, \"seconds\": $((t1 - t0)), \"rendezvous_timeout_s\": 10}" > gpurun_out/r03_failed_rank.json
cat gpurun_out/r03_failed_rank.json
grep -h "NoDevice\|TimeoutError\|never arrived" gpurun_out/r03_failed_rank_stderr.txt | head -5
AGPU_BENCH_STRONG_LEG=1 python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('strong leg at world 1:', d['extra'].get('strong_scaling'))"
for b in 256 64 256 64; do AGPU_BITMAP_BLOCK=$b python tools/kernel_table.py --tag bm$b 2>&1 | grep "^{'kernel': 'validity AND (bitmap)'\|^{'kernel': 'bitmap not'\|^{'kernel': 'merge validity" | cut -c1-120 | sed "s/^/block $b: /"; done
