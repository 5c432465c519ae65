#!/bin/bash
# round 6 rehearsals on a 1-GPU box: the torchrun launch at world 1 WITH the CPU leg and the per-rank checks (the N > 1 code paths that can run here: the
# strong leg, gpu_parity through the communicator, the final-reduce verification), the world-mismatch refusal, two workers naming one GPU (rendezvous +
# RCCL bootstrap between processes, then a clean refusal), and the whole GPU suite with every pipeline fusing (AGPU_FUSE=1)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
AGPU_BENCH_STRONG_LEG=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --no-extra-configs > gpurun_out/r06_bench_torchrun_world1.json 2> gpurun_out/r06_bench_torchrun_world1.err
echo "torchrun world1 rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06_bench_torchrun_world1.json").read().strip().splitlines()[-1])
print(d["value"], d["n_gpus"], "| parity:", d.get("gpu_parity"), "| cpu:", d["cpu_baseline"] and d["cpu_baseline"]["value"], "| verified:", d["extra"]["reduce_sum_min_max"].get("verified"), d["extra"]["reduce_sum_min_max"].get("verified_rank0"), "| strong:", d["extra"].get("strong_scaling",{}).get("value_GBps"), "| frac_per_rank:", d["roofline"].get("frac_per_rank"))
P
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > gpurun_out/r06_world_mismatch.json 2> gpurun_out/r06_world_mismatch.err
echo "mismatch rc=$? (expected non-zero: 3 from the worker)"; cat gpurun_out/r06_world_mismatch.json | cut -c1-300
AGPU_BENCH_DEVICE_OVERRIDE=0 AGPU_COMM_TIMEOUT_MS=20000 timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --rendezvous-timeout 20 > gpurun_out/r06_two_workers_one_gpu.json 2> gpurun_out/r06_two_workers_one_gpu.err
echo "two workers on one GPU rc=$? (expected non-zero)"; tail -3 gpurun_out/r06_two_workers_one_gpu.err | cut -c1-300
AGPU_FUSE=1 AGPU_PERF_STRICT=0 timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_gpu_suite_fuse1.log 2>&1
echo "suite with AGPU_FUSE=1 rc=$?"; grep -E "passed|failed" gpurun_out/r06_gpu_suite_fuse1.log | tail -2
