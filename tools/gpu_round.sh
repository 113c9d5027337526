#!/bin/bash
# One GPU session for the round's evidence: parity tests, smoke(), the default benchmark command (CPU leg and the cfg3 / cfg4 secondary workloads in its line), a rocprofv3 kernel trace of the same command, the
# per-kernel PMC passes (SQ mix, FETCH_SIZE, WRITE_SIZE: tools/pmc_bench.sh), and the cfg4 workload with its kernel statistics.
# Every command runs under its own timeout: a crashed process under rocprofv3 must not hold the box until gpurun's limit.
# Everything lands in gpurun_out/round/ ; tools/collect_profiles.py copies the summaries into profiles/rNN_*.
export GPU_MAX_HW_QUEUES=16
R=$PWD; O=$R/gpurun_out/round; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o lm -- python $R/bench.py > $O/bench_prof.json 2> $O/bench_prof.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -o c4 -- python $R/bench.py --workload cfg4 --cfg4-kf 200 --steps 3 --warmup 1 --cpu-seconds 0 > $O/bench_cfg4_prof.json 2> $O/bench_cfg4_prof.err
cd $R
timeout 1500 bash tools/pmc_bench.sh > $O/pmc_bench.log 2>&1
cp gpurun_out/sq_summary.json $O/sq_summary.json
timeout 600 bash tools/fam_compare.sh > $O/families.log 2>&1   # fused-kernel throughput per landmark family
find $O -name "*kernel_trace*" -delete
tail -3 $O/pytest_gpu.log; cat $O/smoke.log | tail -2; cut -c1-600 $O/bench.json
if [ "${SOAK:-1}" != "0" ]; then timeout 1200 python tools/soak_parity.py 10 2>&1 | grep -v "^\[build" | tail -5 > $O/soak.log; tail -2 $O/soak.log; fi
