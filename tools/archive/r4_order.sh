#!/bin/bash
# launch-order experiments (GPU box): the variant library in place of the product one, one short run per setting
cp srba_amd/lib/libsrba_hip.so /tmp/orig.so; cp srba_amd/lib/variants/libsrba_hip_x.so srba_amd/lib/libsrba_hip.so
python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-secondary > /dev/null 2>&1   # fills the capsule cache
SRBA_HIP_PLAN_DEBUG=1 python tools/diag_launch_order.py 2>&1 | grep -E "plan|mean" 
for s in "$@"; do env $s python tools/diag_launch_order.py 2>/dev/null | tail -1; done
cp /tmp/orig.so srba_amd/lib/libsrba_hip.so
