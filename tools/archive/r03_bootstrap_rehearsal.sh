#!/bin/bash
# GPU box (one GPU): two bench workers that both name GPU 0.  What this shows: the file rendezvous between the processes, RCCL's
# socket bootstrap between two ranks on this pool's network configuration, and that ncclCommInitRank's refusal ("Duplicate GPU
# detected") comes back as a non-zero exit within seconds — not a hang.  It can not show a working 2-rank communicator.
set -u
mkdir -p gpurun_out
export NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,BOOTSTRAP,NET AGPU_BENCH_DEVICE_OVERRIDE=0 AGPU_COMM_TIMEOUT_MS=30000
t0=$(date +%s)
timeout 180 python bench.py --gpus 2 --steps 2 --warmup 1 --rendezvous-timeout 30 --no-cpu-baseline --no-extra-configs > gpurun_out/r03_bootstrap_rehearsal.out 2> gpurun_out/r03_bootstrap_rehearsal.err
rc=$?
t1=$(date +%s)
echo "exit code $rc after $((t1 - t0)) s"
grep -E "Bootstrap|bootstrap|NET/|Duplicate|duplicate|ncclCommInitRank|worker exit|Error|error|AGPU|Timeout" gpurun_out/r03_bootstrap_rehearsal.err | cut -c1-220 | head -40
