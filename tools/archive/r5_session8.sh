#!/bin/bash
# round 5, session 8: the batch record read from device memory through per-phase laundered references (SGPR spills 220-280 -> 93-100) against the record as a kernel argument
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/s8; mkdir -p $O
timeout 900 bash tools/r4_variants.sh "lib=byval" "lib=lnd" "lib=byval" "lib=lnd" > $O/ab_lnd.txt 2>&1; cat $O/ab_lnd.txt
timeout 900 python tools/r5_wg_check.py stereo 64 "WG_HS=1" > $O/wg_stereo.txt 2>&1; tail -1 $O/wg_stereo.txt
timeout 900 python tools/r5_wg_check.py mono 64 "WG_HS=1" > $O/wg_mono.txt 2>&1; tail -1 $O/wg_mono.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-secondary > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/s8/bench.json") if l.startswith("{")][-1]); print("seq ms/kf", d["config"].get("sequential_ms_per_kf"), "launch_order", d["config"].get("launch_order"))
PY
