#!/bin/bash
# round 6b: one dev probe under gpurun — bash tools/r06b_probe.sh <script.py> [args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python "$@" 2>&1 | tail -30
