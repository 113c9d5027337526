#!/bin/bash
# usage (GPU box): tools/r4_variants.sh "lib=<variant> ENV=.." ...   -- 24 fused launches per setting (tools/diag_launch_order.py) with the variant library in place of the product one
cp srba_amd/lib/libsrba_hip.so /tmp/orig.so
python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-secondary > /dev/null 2>&1   # fills the capsule cache
for v in "$@"; do lib=$(echo "$v" | tr ' ' '\n' | grep '^lib=' | cut -d= -f2); envs=$(echo "$v" | tr ' ' '\n' | grep -v '^lib=' | tr '\n' ' ')
  if [ -n "$lib" ]; then cp srba_amd/lib/variants/libsrba_hip_$lib.so srba_amd/lib/libsrba_hip.so; else cp /tmp/orig.so srba_amd/lib/libsrba_hip.so; fi
  echo -n "[$v] "; env $envs python tools/diag_launch_order.py 2>/dev/null | tail -1 | cut -c1-200; done
cp /tmp/orig.so srba_amd/lib/libsrba_hip.so
