#!/bin/bash
# cfg3 workload: landmark windows with SRBA_HIP_GANG_FROM_NB block rows or more run on the lock-step multi-workgroup path instead of one wavefront each (0 = off)
for e in ${NBS:-0 100 80 60}; do SRBA_HIP_GANG_FROM_NB=$e timeout 600 python bench.py --workload cfg3 --steps 3 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('gang from nb $e: %.1f it/s, %.2f ms/step, trials per step %s' % (d['value'], d['ms_per_step'], d['config'].get('lm_trials_per_step')))"; done
