#!/bin/bash
# round 6, second session: two lock-step streams in the same-width streaming kernels, A/B in one process
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python tools/probe/two_streams_cmp.py 2>&1 | tail -24
