#!/bin/bash
# round 5, session 6: factor look-ahead + fused Cholesky / inverse, two terms in flight in the LDS K6 / Schur loops
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/s6; mkdir -p $O
timeout 900 python tools/r5_wg_check.py stereo 64 "WG_HS=1" "WG_HS=1,PHASES=1" > $O/wg_stereo.txt 2>&1; tail -6 $O/wg_stereo.txt
timeout 900 python tools/r5_wg_check.py mono 64 "WG_HS=1" > $O/wg_mono.txt 2>&1; tail -1 $O/wg_mono.txt
timeout 900 python tools/r5_wg_check.py rb3d 64 "WG_HS=1" > $O/wg_rb3d.txt 2>&1; tail -1 $O/wg_rb3d.txt
timeout 900 python tools/r5_wg_check.py cart3d 64 "WG_HS=1" > $O/wg_cart3d.txt 2>&1; tail -1 $O/wg_cart3d.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
timeout 600 python bench.py --workload cfg3 --steps 3 --warmup 1 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cut -c1-330 $O/bench_cfg3.json
timeout 600 python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-seconds 0 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-330 $O/bench_cfg4.json
