# (GPU box) cfg4 with the mono-only variant library (srba_amd/lib/variants/libsrba_hip_mono.so) in place of the product one: gangs side by side with stream priorities
out=gpurun_out/cfg4_ab; mkdir -p $out
touch srba_amd/lib/*.o; sleep 1; cp srba_amd/lib/variants/libsrba_hip_mono.so srba_amd/lib/libsrba_hip.so; sleep 1; touch srba_amd/lib/libsrba_engine.so srba_amd/bin/srba-slam oracle/_build/*.so
run() { echo -n "[$*] "; env "$@" python bench.py --workload cfg4 --steps 5 --warmup 1 --cpu-seconds 0 2>$out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = d['roofline']
print('%.0f it/s  %.2f ms/step  seq %.3f ms' % (d['value'], d['ms_per_step'], r['kernel_ms']))"; }
run SRBA_HIP_BIG_GANGS=4 SRBA_HIP_BIG_FRESH=1
run SRBA_HIP_BIG_GANGS=5 SRBA_HIP_BIG_FRESH=1
run SRBA_HIP_BIG_GANGS=6 SRBA_HIP_BIG_FRESH=1
run SRBA_HIP_BIG_GANGS=8 SRBA_HIP_BIG_FRESH=1
run SRBA_HIP_BIG_GANGS=8 SRBA_HIP_BIG_FRESH=1 GPU_MAX_HW_QUEUES=8
run SRBA_HIP_BIG_GANGS=4 SRBA_HIP_BIG_FRESH=1
grep -c "\[build\] hipcc" $out/err.txt
