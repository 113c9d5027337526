#!/bin/bash
# round 5, session 2: first run of the workgroup kernel of the landmark families
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/s2; mkdir -p $O
timeout 600 python tools/r5_wg_check.py stereo 8 "WG=0" "WG=1" "WG=1,WG256_FROM=1000" "WG=1,WG256_FROM=24" > $O/wg_stereo_small.txt 2>&1; tail -8 $O/wg_stereo_small.txt
timeout 900 python tools/r5_wg_check.py stereo 64 "WG=0" "WG=1" "WG=1,PHASES=1" "WG=1,WG256_FROM=48" "WG=1,WG_FROM=60" > $O/wg_stereo.txt 2>&1; tail -14 $O/wg_stereo.txt
timeout 900 python tools/r5_wg_check.py mono 64 "WG=0" "WG=1" > $O/wg_mono.txt 2>&1; tail -4 $O/wg_mono.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "landmark or stereo or mono or schur or families or cfg3 or room" 2>&1 | tail -15 > $O/pytest_lm.log; tail -6 $O/pytest_lm.log
