#!/bin/bash
# (GPU box) cfg4 A/B on `bench.py --workload cfg4`: settings given as arguments ("SRBA_HIP_SCHUR_XCD=0 SRBA_HIP_BIG_GANGS=1" ...; "" = defaults), then the kernel timeline of the
# defaults and the big-path parity tests
out=gpurun_out/cfg4_ab; mkdir -p $out
run() { echo -n "[$*] "; env $* python bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-seconds ${CPU:-0} 2>$out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = d['roofline']; c = d.get('cpu_baseline') or {}
print('%.0f it/s  %.2f ms/step  trials %d  seq %.3f ms  sequential %.1f ms/kf  chi2 vs oracle %s' % (d['value'], d['ms_per_step'],
      d['config']['lm_trials_per_step'], r['kernel_ms'], d['config']['sequential_ms_per_kf'], c.get('chi2_final_rel_diff_vs_gpu')))"; }
for s in "$@"; do run $s; done > $out/ab.txt 2>&1
cat $out/ab.txt
bash tools/diag_cfg4_timeline.sh; cp gpurun_out/cfg4_timeline.txt $out/timeline_default.txt; head -8 $out/timeline_default.txt
timeout 1200 python -m pytest tests -x -q -m gpu -k "big or deep or gang or cfg4 or large or schur" > $out/pytest_big.log 2>&1; tail -3 $out/pytest_big.log
