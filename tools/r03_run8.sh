set -u
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_abi.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -4
for b in 256 512 1024; do echo "== trig16 block $b"; AGPU_TRIG16_BLOCK=$b python tools/probe/narrow_tunings.py 2>&1 | grep "^{" | grep "sin_u16, table_tiles 1\|sin_u8, table_tiles 1\|u8 eq -> bitmap, stream_unroll 2"; done
