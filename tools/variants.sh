#!/bin/bash
# usage (on the GPU box): tools/variants.sh "ENV=.. ENV2=.." ...  -- runs the short bench once per environment setting
for v in "$@"; do echo "== $v"; env $v python bench.py --steps 5 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; done
