#!/bin/bash
# round 5, session 12: workgroup kernels compiled for three wavefronts per SIMD (168 registers, 768-thread top class) against two (256 registers, 512 threads); stereo family only
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/s12; mkdir -p $O
cp srba_amd/lib/libsrba_hip.so /tmp/orig.so
for v in w2 w3 w2 w3; do cp srba_amd/lib/variants/libsrba_hip_$v.so srba_amd/lib/libsrba_hip.so; echo "== $v"; timeout 600 python tools/r5_wg_check.py stereo 64 "WG_HS=1" 2>&1 | tail -1; done > $O/ab_waves.txt 2>&1
cp /tmp/orig.so srba_amd/lib/libsrba_hip.so; cat $O/ab_waves.txt
timeout 900 python tools/r5_wg_check.py rb3d 64 "WG_HS=1" > $O/wg_rb3d.txt 2>&1; tail -1 $O/wg_rb3d.txt
