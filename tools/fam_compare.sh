#!/bin/bash
# throughput of the fused LM kernel per landmark family (tools/diag_family.py), one line each
for k in rb2d cart2d mono stereo cart3d rb3d; do timeout 600 python tools/diag_family.py $k 2>&1 | tail -1 | cut -c1-260; done
