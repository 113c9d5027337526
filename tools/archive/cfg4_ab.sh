#!/bin/bash
# cfg4 workload, 16 and 32 windows per step, three runs each: one line per run (A/B of two builds = two calls)
for w in 16 32; do for e in 1 2 3; do timeout 500 python bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-seconds 0 --cfg4-windows $w "$@" 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('windows $w: %.1f it/s, %.2f ms/step, %.3f ms per sequence, sequential %.1f ms/kf' % (d['value'], d['ms_per_step'], r['kernel_ms'], d['config']['sequential_ms_per_kf']))"; done; done
