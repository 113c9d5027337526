#!/bin/bash
# usage (on the GPU box): tools/asm_sweep.sh "<wpw>,<bin_kb>" ...   -- tools/diag_assemble.py (srba_hip_linearize on the benchmark batch) per launch geometry of the fused
# normal-equations kernel (SRBA_HIP_ASM_WPW wavefronts = capsules per bin, SRBA_HIP_ASM_BIN_KB of LDS per bin); the capsules are harvested once (bench.py's cache)
python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary > /dev/null 2>&1   # fills the capsule cache
for v in "$@"; do w=${v%,*}; kb=${v#*,}; echo "== wavefronts per bin $w, bin $kb KB"
  SRBA_HIP_ASM_WPW=$w SRBA_HIP_ASM_BIN_KB=$kb timeout 200 python tools/diag_assemble.py 30000 20 2>&1 | tail -1
  SRBA_HIP_ASM_WPW=$w SRBA_HIP_ASM_BIN_KB=$kb SRBA_HIP_PHASE_TIMING=1 timeout 200 python tools/diag_assemble.py 30000 20 2>&1 | grep "per capsule\|in flight\|starts"; done
