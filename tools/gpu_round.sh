#!/bin/bash
# One GPU session: parity tests, the full benchmark, and a rocprofv3 kernel trace of the same command (summaries -> gpurun_out/).
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 3000 gpurun_out/bench.json
ROOTDIR=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof -o lm -- python $ROOTDIR/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $ROOTDIR/gpurun_out/bench_prof.json 2> $ROOTDIR/gpurun_out/bench_prof.err
cd $ROOTDIR
find gpurun_out/prof -name "*stats*" | head; find gpurun_out/prof -name "*kernel_stats*" -exec head -20 {} \;
# keep only small summaries
find gpurun_out/prof -name "*kernel_trace*" -size +2M -delete
