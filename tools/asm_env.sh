#!/bin/bash
# usage (on the GPU box): tools/asm_env.sh "VAR=val VAR2=val" ...  -- tools/diag_assemble.py per environment setting (knobs of srba_assemble.hip)
python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary > /dev/null 2>&1   # fills the capsule cache
for v in "$@"; do echo "== $v"; env $v timeout 200 python tools/diag_assemble.py 30000 20 2>&1 | tail -1; env $v SRBA_HIP_PHASE_TIMING=1 timeout 200 python tools/diag_assemble.py 30000 20 2>&1 | grep "per capsule\|in flight"; done
