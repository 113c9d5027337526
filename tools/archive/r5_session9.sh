#!/bin/bash
# round 5, session 9: full tests, all landmark families, cfg3 bench + its PMC passes, headline bench with secondary workloads
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/s9; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 900 bash tools/fam_compare.sh > $O/families.log 2>&1; cat $O/families.log
timeout 900 bash tools/pmc_cfg3.sh > $O/pmc_cfg3.log 2>&1; tail -12 $O/pmc_cfg3.log; cp gpurun_out/pmc_traffic_cfg3.json gpurun_out/sq_summary_cfg3.json $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
