# (GPU box) cfg4: number of gangs side by side (after s_setprio on the panel chains)
out=gpurun_out/cfg4_ab; mkdir -p $out
run() { echo -n "[$*] "; env "$@" python bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-seconds 0 2>$out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = d['roofline']
print('%.0f it/s  %.2f ms/step  seq %.3f ms' % (d['value'], d['ms_per_step'], r['kernel_ms']))"; }
run SRBA_HIP_BIG_GANGS=1
run SRBA_HIP_BIG_GANGS=2
run SRBA_HIP_BIG_GANGS=3
run SRBA_HIP_BIG_GANGS=4
run SRBA_HIP_BIG_GANGS=2
