#!/bin/bash
# usage: gpu_variants.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- short bench per environment setting (capsule cache shared), one line each
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for v in "$@"; do
  env $v python bench.py --steps 10 --warmup 2 --cpu-seconds 0 2> gpurun_out/var.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-50s value %.3f M it/s  ms_per_step %.2f  kernel_ms %.2f' % ('$v', d['value'] / 1e6, d['ms_per_step'], d['roofline']['kernel_ms']))
" || tail -5 gpurun_out/var.err
done
