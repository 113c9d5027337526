#!/bin/bash
# the fused LM kernel of the relative-pose family with a register budget for three wavefronts per SIMD (variant library occ3) against the built one, at 8 / 10 / 12 wavefronts per CU in the launch plan
cp srba_amd/lib/libsrba_hip.so /tmp/orig.so
run() { timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-secondary 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; }
run base
SRBA_HIP_WAVES_PER_CU=12 run base_w12
cp srba_amd/lib/variants/libsrba_hip_occ3.so srba_amd/lib/libsrba_hip.so; touch srba_amd/lib/libsrba_hip.so srba_amd/lib/libsrba_engine.so
run occ3_w8
SRBA_HIP_WAVES_PER_CU=10 run occ3_w10
SRBA_HIP_WAVES_PER_CU=12 run occ3_w12
cp /tmp/orig.so srba_amd/lib/libsrba_hip.so
