#!/bin/bash
# usage (on the GPU box): tools/variants.sh "lib=<variant> ENV=.. ENV2=.." ...  -- runs the short bench once per setting
cp srba_amd/lib/libsrba_hip.so /tmp/libsrba_hip_orig.so
for v in "$@"; do echo "== $v"; lib=$(echo "$v" | tr ' ' '\n' | grep '^lib=' | cut -d= -f2); envs=$(echo "$v" | tr ' ' '\n' | grep -v '^lib=' | tr '\n' ' ')
  if [ -n "$lib" ]; then cp srba_amd/lib/variants/libsrba_hip_$lib.so srba_amd/lib/libsrba_hip.so; else cp /tmp/libsrba_hip_orig.so srba_amd/lib/libsrba_hip.so; fi
  env $envs python bench.py --steps 5 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; done
cp /tmp/libsrba_hip_orig.so srba_amd/lib/libsrba_hip.so
