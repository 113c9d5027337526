#!/bin/bash
# One GPU session: parity tests, the full benchmark, a rocprofv3 kernel trace of the same command, and two PMC passes (HBM bytes).
# Summaries land in gpurun_out/ ; copy the ones to keep into profiles/.
set -x
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 3000 gpurun_out/bench.json
ROOTDIR=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof -o lm -- python $ROOTDIR/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $ROOTDIR/gpurun_out/bench_prof.json 2> $ROOTDIR/gpurun_out/bench_prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/pmc_fetch -o f -- python $ROOTDIR/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $ROOTDIR/gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/pmc_write -o w -- python $ROOTDIR/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $ROOTDIR/gpurun_out/pmc_write.err
cd $ROOTDIR
python tools/summarize_profiles.py gpurun_out > gpurun_out/profile_summary.json
cat gpurun_out/profile_summary.json
# keep only small summaries
find gpurun_out -name "*kernel_trace*" -size +2M -delete
find gpurun_out -name "*counter_collection*" -size +4M -delete
