#!/bin/bash
# DEV TOOL (round 5): is u8 eq's 0.76 a property of the kernel or of the 2 GB problem?  Same kernel at 1e9, 2e9, 4e9 and 0.25e9 rows.
for n in 250000000 1000000000 2000000000 4000000000; do echo -n "rows $n: "; N=$n python tools/probe/cmp8_ab.py; done
