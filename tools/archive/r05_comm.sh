#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_comm.py tests/test_cpp_host.py tests/test_gpu_tile_auto.py -x -q 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
AGPU_COMM_TEST_STALL_INIT_MS=200000 timeout 600 python bench.py --steps 5 --warmup 2 --rendezvous-timeout 5 > gpurun_out/r05_bench_local_fallback.json 2> gpurun_out/r05_bench_local_fallback.err; echo "bench (stalled bootstrap) rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r05_bench_local_fallback.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['extra']['rccl_local_fallback'], d['extra']['rccl_ranks'])"
