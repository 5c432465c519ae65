#!/bin/bash
# round 6, second session, call 1: the stream-split probe (tools/probe/stream_split.hip), log with specials in line (parity + the table rows)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 tools/probe/stream_split.bin 9 > gpurun_out/r06b_stream_split.jsonl 2> gpurun_out/r06b_stream_split.err
echo "stream_split rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unary or log or exhaustive_over_all or golden" 2>&1 | tail -5
timeout 1200 python tests/tools/exhaustive_vs_oracle.py log 2>&1 | tail -5
cp gpurun_out/r03_exhaustive_vs_oracle.json gpurun_out/r06b_exhaustive_log.json 2>/dev/null
timeout 900 python tools/kernel_table.py --tag r06b > gpurun_out/r06b_kernel_table.log 2>&1
echo "kernel_table rc=$?"
grep -i "log" gpurun_out/r06b_kernel_table.log | head
cat gpurun_out/r06b_stream_split.jsonl
