#!/bin/bash
# round 5, session 10: batch record by pointer + laundered references + global-address-space pointer fields, against the record as a kernel argument
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/s10; mkdir -p $O
timeout 900 bash tools/r4_variants.sh "lib=byval2" "lib=gp2" "lib=byval2" "lib=gp2" > $O/ab_gp.txt 2>&1; cat $O/ab_gp.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 900 python tools/r5_wg_check.py stereo 64 "WG_HS=1" "WG_HS=1,PHASES=1" > $O/wg_stereo.txt 2>&1; tail -6 $O/wg_stereo.txt
timeout 900 python tools/r5_wg_check.py rb3d 64 "WG_HS=1" "WG=0" > $O/wg_rb3d.txt 2>&1; tail -2 $O/wg_rb3d.txt
timeout 600 python bench.py --workload cfg3 --steps 3 --warmup 1 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cut -c1-330 $O/bench_cfg3.json
