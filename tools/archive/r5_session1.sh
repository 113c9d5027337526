#!/bin/bash
# round 5, session 1: parity tests of the block LDL^t solver + spec fallback, A/B of the fused launch (LDL^t vs LL^t), family baselines
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/s1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 600 bash tools/r4_variants.sh "lib=llt" "lib=ldl" "lib=llt" "lib=ldl" > $O/ab_ldl.txt 2>&1; cat $O/ab_ldl.txt
timeout 600 bash tools/fam_compare.sh > $O/families.log 2>&1; cat $O/families.log
