#!/bin/bash
# DEV TOOL: build-time variants of the pair pipeline (tile size, destination region size) as separate libraries under
# tools/probe/variants/ (git-ignored, travels with gpurun); run tools/probe/take_passes.py with AGPU_LIB=<variant> for each.
#   tools/probe/put_variants.sh build      (here: hipcc cross-compiles)
#   tools/probe/put_variants.sh run        (on the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/arrow_gpu_amd/csrc
OUT=$ROOT/tools/probe/variants
VARIANTS=("e16_rd0:" "e8_rd0:-DBKT_E=8" "e16_rd1:-DBKT_RD_EXTRA=1" "e16_rd2:-DBKT_RD_EXTRA=2" "e8_rd1:-DBKT_E=8 -DBKT_RD_EXTRA=1" ${PUT_EXTRA_VARIANTS})
if [ "$1" = build ]; then
  mkdir -p "$OUT"
  (cd "$CS" && make -j8 all >/dev/null)
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c "$CS/swizzle.hip" -o "$OUT/swizzle_$name.o" &
  done
  wait
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    objs=$(ls "$CS"/build/*.o | grep -v "swizzle.o\|_nosc1.o")
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libput_$name.so" $objs "$OUT/swizzle_$name.o" -L/opt/rocm/lib -lrccl -lrocprofiler-sdk-roctx
    rm -f "$OUT/swizzle_$name.o"
  done
  ls -la "$OUT"
else
  mkdir -p "$ROOT/gpurun_out"
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    echo "== $name"
    AGPU_LIB="$OUT/libput_$name.so" PUT_ONLY=1 python "$ROOT/tools/probe/take_passes.py" 2>&1 | grep -E "put_|take_pairs"
  done | tee "$ROOT/gpurun_out/r03_put_variants.txt"
fi
