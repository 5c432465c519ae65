#!/bin/bash
# round 4 rehearsals on a 1-GPU box: the torchrun launch at world 1 (weak + the strong leg), the self-launcher refusing a world that is not
# `--gpus` distinct devices (two workers naming GPU 0), and the whole GPU suite with every pipeline fusing (AGPU_FUSE=1)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
AGPU_BENCH_STRONG_LEG=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_torchrun_world1.json 2> gpurun_out/r04_bench_torchrun_world1.err
echo "torchrun world1 rc=$?"
# --gpus 2 asked of a launcher that starts ONE rank: the world check must refuse with one JSON error line and a non-zero exit
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > gpurun_out/r04_world_mismatch.json 2> gpurun_out/r04_world_mismatch.err
echo "mismatch rc=$? (expected non-zero: 3 from the worker)"; cat gpurun_out/r04_world_mismatch.json
AGPU_BENCH_DEVICE_OVERRIDE=0 AGPU_COMM_TIMEOUT_MS=20000 timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --rendezvous-timeout 20 > gpurun_out/r04_two_workers_one_gpu.json 2> gpurun_out/r04_two_workers_one_gpu.err
echo "two workers on one GPU rc=$? (expected non-zero)"; tail -3 gpurun_out/r04_two_workers_one_gpu.err
AGPU_FUSE=1 AGPU_PERF_STRICT=0 timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04_gpu_suite_fuse1.log 2>&1
echo "suite with AGPU_FUSE=1 rc=$?"; grep -E "passed|failed" gpurun_out/r04_gpu_suite_fuse1.log | tail -2
