#!/bin/bash
# DEV TOOL (round 5): tools/probe/sc1_more.py, product against the variant, three fresh processes each
for i in 1 2 3; do for lib in "" tools/probe/variants/libagpu_sc1more.so; do echo -n "${lib:-product}: "; AGPU_LIB=${lib:+$PWD/$lib} python tools/probe/sc1_more.py; done; done
