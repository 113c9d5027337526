#!/bin/bash
# round 5, session 7: padded LDS block stride (bank conflicts of the ds_add_f64 loops), two-pass K10
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/s7; mkdir -p $O
timeout 900 python tools/r5_wg_check.py stereo 64 "WG_HS=1" "WG_HS=1,PHASES=1" > $O/wg_stereo.txt 2>&1; tail -6 $O/wg_stereo.txt
timeout 900 python tools/r5_wg_check.py mono 64 "WG_HS=1" > $O/wg_mono.txt 2>&1; tail -1 $O/wg_mono.txt
timeout 900 python tools/r5_wg_check.py rb3d 64 "WG_HS=1" > $O/wg_rb3d.txt 2>&1; tail -1 $O/wg_rb3d.txt
timeout 900 python tools/r5_wg_check.py cart3d 64 "WG_HS=1" > $O/wg_cart3d.txt 2>&1; tail -1 $O/wg_cart3d.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --workload cfg3 --steps 3 --warmup 1 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cut -c1-330 $O/bench_cfg3.json
