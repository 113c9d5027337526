#!/bin/bash
# usage (on the GPU box): tools/asm_variants.sh <variant> ...  -- tools/diag_assemble.py once per variant library (srba_amd/lib/variants/libsrba_hip_<variant>.so; "base" = the built one)
cp srba_amd/lib/libsrba_hip.so /tmp/libsrba_hip_orig.so
python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary > /dev/null 2>&1   # fills the capsule cache
for v in "$@"; do echo "== $v"; if [ "$v" != base ]; then cp srba_amd/lib/variants/libsrba_hip_$v.so srba_amd/lib/libsrba_hip.so; touch srba_amd/lib/libsrba_hip.so srba_amd/lib/libsrba_engine.so; else cp /tmp/libsrba_hip_orig.so srba_amd/lib/libsrba_hip.so; fi
  timeout 200 python tools/diag_assemble.py 2>&1 | tail -1; SRBA_HIP_PHASE_TIMING=1 timeout 200 python tools/diag_assemble.py 2>&1 | grep "per capsule\|in flight"; done
cp /tmp/libsrba_hip_orig.so srba_amd/lib/libsrba_hip.so
