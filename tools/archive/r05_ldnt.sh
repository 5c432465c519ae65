#!/bin/bash
# DEV TOOL (round 5): tools/probe/ldnt_ab.py, product (nontemporal loads) against plain loads, three fresh processes each
for i in 1 2 3; do for lib in "" tools/probe/variants/libagpu_ldplain.so; do echo -n "${lib:-product}: "; AGPU_LIB=${lib:+$PWD/$lib} python tools/probe/ldnt_ab.py; done; done
