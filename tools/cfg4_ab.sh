#!/bin/bash
# cfg4 workload with the persistent Cholesky kernel (1) and with one launch per panel step / update (0)
for e in 1 0 1 0; do SRBA_HIP_BIG_PERSISTENT=$e timeout 400 python bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-seconds 0 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('persistent $e', round(d['value'],1), round(d['ms_per_step'],2), round(r['kernel_ms'],4), r['frac'], r.get('lane_time_over_step_time'))"; done
