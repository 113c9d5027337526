#!/bin/bash
# usage (GPU box): tools/r5_big_variants.sh "lib=<variant> ENV=.." ... -- `bench.py --workload cfg4` per setting with srba_amd/lib/variants/libsrba_hip_<variant>.so (tools/quick_big.sh)
# in place of the product library (no lib=: the product). The objects are touched first so that bench.py's build() check does not recompile on the box.
out=gpurun_out/cfg4_ab; mkdir -p $out; cp srba_amd/lib/libsrba_hip.so /tmp/product.so
touch srba_amd/lib/*.o; sleep 1; touch srba_amd/lib/libsrba_hip.so; sleep 1; touch srba_amd/lib/libsrba_engine.so srba_amd/bin/srba-slam oracle/_build/*.so
for v in "$@"; do lib=$(echo "$v" | tr ' ' '\n' | grep '^lib=' | cut -d= -f2); envs=$(echo "$v" | tr ' ' '\n' | grep -v '^lib=' | tr '\n' ' ')
  if [ -n "$lib" ]; then cp srba_amd/lib/variants/libsrba_hip_$lib.so srba_amd/lib/libsrba_hip.so; else cp /tmp/product.so srba_amd/lib/libsrba_hip.so; fi
  sleep 1; touch srba_amd/lib/libsrba_engine.so srba_amd/bin/srba-slam oracle/_build/*.so
  echo -n "[$v] "; env $envs python bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-seconds ${CPU:-0} 2>$out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = d['roofline']; c = d.get('cpu_baseline') or {}
print('%.0f it/s  %.2f ms/step  seq %.3f ms  chi2 vs oracle %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], c.get('chi2_final_rel_diff_vs_gpu')))"; grep -c "hipcc" $out/err.txt; done
cp /tmp/product.so srba_amd/lib/libsrba_hip.so
