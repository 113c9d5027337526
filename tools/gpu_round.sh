#!/bin/bash
# One GPU session for the round's evidence: parity tests, smoke(), the default benchmark command (CPU leg and the cfg3 / cfg4 secondary workloads in its line), a rocprofv3 kernel trace of the same command, the
# per-kernel PMC passes (SQ mix, FETCH_SIZE, WRITE_SIZE: tools/pmc_bench.sh), and the cfg4 workload with its kernel statistics.
# Every command runs under its own timeout: a crashed process under rocprofv3 must not hold the box until gpurun's limit.
# Everything lands in gpurun_out/round/ ; tools/collect_profiles.py copies the summaries into profiles/rNN_*.
export GPU_MAX_HW_QUEUES=16
R=$PWD; O=$R/gpurun_out/round; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 1800 python bench.py > $O/bench.json 2> $O/bench.err   # (with the BASELINE-size cfg4 leg: about eight minutes more)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o lm -- python $R/bench.py --cfg4-full-budget-s 0 > $O/bench_prof.json 2> $O/bench_prof.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -o c4 -- python $R/bench.py --workload cfg4 --cfg4-kf 200 --steps 3 --warmup 1 --cpu-seconds 0 > $O/bench_cfg4_prof.json 2> $O/bench_cfg4_prof.err
cd $R
timeout 1500 bash tools/pmc_bench.sh > $O/pmc_bench.log 2>&1
cp gpurun_out/sq_summary.json $O/sq_summary.json
timeout 900 bash tools/pmc_cfg3.sh > $O/pmc_cfg3.log 2>&1; cp gpurun_out/pmc_traffic_cfg3.json gpurun_out/sq_summary_cfg3.json $O/ 2>/dev/null   # the workgroup landmark kernels on the cfg3 workload
# the driver's N = 8 line with the FULL 30 000-key-frame map per rank, eight ranks on this one GPU over gloo: do the eight harvests serialise on the shared host?
( time env SRBA_BENCH_DEVICE=0 SRBA_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
	--master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 3 --warmup 1 --cpu-seconds 0 --cache-dir "" ) > $O/eight_ranks_full_map.log 2>&1
timeout 300 python tools/diag_launch_stamps.py > $O/launch_order.txt 2>&1
# round 6: the fused normal equations (per-CU timeline of the launch, counters) and the one-map sweep
SRBA_HIP_PHASE_TIMING=1 timeout 300 python tools/diag_assemble.py 30000 20 > $O/assemble_timeline.txt 2>&1; timeout 300 python tools/diag_assemble.py 30000 20 2>&1 | tail -1 >> $O/assemble_timeline.txt
timeout 900 bash tools/asm_pmc.sh > $O/assemble_pmc.log 2>&1; cp gpurun_out/asm_pmc/summary.txt $O/assemble_pmc.txt
timeout 400 python tools/diag_sweep.py 6000 1 > $O/sweep.txt 2>&1
timeout 400 bash tools/diag_cfg4_timeline.sh > /dev/null 2>&1; cp gpurun_out/cfg4_timeline.txt $O/cfg4_timeline.txt   # where a cfg4 step goes, kernel by kernel
timeout 600 bash tools/fam_compare.sh > $O/families.log 2>&1   # fused-kernel throughput per landmark family
find $O -name "*kernel_trace*" -delete
tail -3 $O/pytest_gpu.log; cat $O/smoke.log | tail -2; cut -c1-600 $O/bench.json
if [ "${SOAK:-1}" != "0" ]; then timeout 1200 python tools/soak_parity.py 10 2>&1 | grep -v "^\[build" | tail -5 > $O/soak.log; tail -2 $O/soak.log; fi
