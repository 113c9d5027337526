#!/bin/bash
# round 5, session 3: workgroup kernel with chunked K6, Y per U_Apf block, two tile rows per pass, block-row assembly
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/s3; mkdir -p $O
timeout 900 python tools/r5_wg_check.py stereo 64 "WG=1" "WG=1,PHASES=1" "WG=1,WG256_FROM=24" "WG=1,WG256_FROM=1000" > $O/wg_stereo.txt 2>&1; tail -12 $O/wg_stereo.txt
timeout 900 python tools/r5_wg_check.py mono 64 "WG=1" > $O/wg_mono.txt 2>&1; tail -2 $O/wg_mono.txt
timeout 900 python tools/r5_wg_check.py rb3d 64 "WG=0" "WG=1" > $O/wg_rb3d.txt 2>&1; tail -2 $O/wg_rb3d.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
timeout 600 python bench.py --workload cfg3 --steps 3 --warmup 1 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cut -c1-400 $O/bench_cfg3.json
