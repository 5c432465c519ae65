#!/bin/bash
# DEV TOOL (round 5): tools/probe/medium_table.py with the big-column threshold as a build-time experiment (AGPU_TABLE_BIG_COLUMN_MIB was a dev switch of that experiment, not in the product)
for i in 1 2 3; do for mib in "" 64; do AGPU_TABLE_BIG_COLUMN_MIB=$mib python tools/probe/medium_table.py; done; done
