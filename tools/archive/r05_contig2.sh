#!/bin/bash
# Round 5, GPU box: the channel-hash sweep (tools/probe/hash_bits.py) over a PHYSICALLY CONTIGUOUS 22 GiB block and, as the same-box
# control, over a plain hipMalloc one; host-copy routes by allocator.
set -u
mkdir -p gpurun_out
for flags in 4 0; do
  echo "== hash_bits AGPU_DEVICE_MALLOC_FLAGS=$flags"
  AGPU_DEVICE_MALLOC_FLAGS=$flags timeout 900 python tools/probe/hash_bits.py > gpurun_out/r05_hash_bits_flags$flags.jsonl 2>&1
  cp gpurun_out/hash_bits.json gpurun_out/r05_hash_bits_flags$flags.json 2>/dev/null
  python3 - gpurun_out/r05_hash_bits_flags$flags.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        r = json.loads(l); print(r['series'], r['bit'], r['frac'])
    else: print(l.strip())
PY
done
timeout 600 python tools/probe/host_copy_routes.py > gpurun_out/r05_host_copy_routes.json 2> gpurun_out/r05_host_copy_routes.txt; echo "routes rc=$?"; cat gpurun_out/r05_host_copy_routes.txt
