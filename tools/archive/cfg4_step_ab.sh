#!/bin/bash
# cfg4 workload: panel step + previous trailing update in one launch per 32 columns (1, k_chol_step) against two launches (0, k_chol_panel + k_chol_update)
for e in 1 0 1; do SRBA_HIP_BIG_FUSED_STEP=$e timeout 500 python bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('fused step $e: %.1f it/s, %.2f ms/step, %d factorisations in %d sequences, %.3f ms per sequence, %.4f ms per factorisation, frac %.5f, share of step %.2f' % (d['value'], d['ms_per_step'], r['factorisations'], r['launch_sequences'], r['kernel_ms'], r['ms_per_factorisation'], r['frac'], r['lane_time_over_step_time']))"; done
