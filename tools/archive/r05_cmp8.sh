#!/bin/bash
# DEV TOOL (round 5): is u8 eq's 0.76 a property of the kernel, of the 2 GB problem, or of where 1e9-byte columns land?
# (a) the same kernel at 0.25e9 … 4e9 rows; (b) 1e9 / 2e9 rows run on the FRONT of columns allocated for 4e9 rows
for n in 250000000 1000000000 2000000000 4000000000; do echo -n "rows $n: "; N=$n python tools/probe/cmp8_ab.py; done
for n in 250000000 1000000000 2000000000; do echo -n "rows $n of columns for 4e9: "; N=$n NALLOC=4000000000 python tools/probe/cmp8_ab.py; done
