#!/bin/bash
# cfg4 workload with the large windows in lock-step on one stream (gang, 1) and with one host thread + stream per window (0)
for e in 1 0 1; do SRBA_HIP_BIG_GANG=$e timeout 500 python bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('gang $e: %.1f it/s, %.2f ms/step, %d factorisations in %d sequences, %.3f ms per sequence, %.4f ms per factorisation, frac %.5f, share of step %.2f' % (d['value'], d['ms_per_step'], r['factorisations'], r['launch_sequences'], r['kernel_ms'], r['ms_per_factorisation'], r['frac'], r['lane_time_over_step_time']))"; done
