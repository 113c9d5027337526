#!/bin/bash
# launch-plan knobs of the fused LM kernel on the benchmark batch: one short bench per setting
run() { env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-secondary 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; }
for s in "$@"; do run $s; done
