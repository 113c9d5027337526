#!/bin/bash
# (GPU box) second measurement session of round 5: product = -disable-machine-licm -sink-insts-to-avoid-spills + opaque lane row in wg_diag
out=gpurun_out/s2; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
bash tools/r4_variants.sh "" "lib=nl" "" "lib=nl" > $out/headline.txt 2>&1
for k in stereo mono cart3d rb3d; do timeout 300 python tools/r5_wg_check.py $k 64 WG=1 2>&1 | tail -1 >> $out/families.txt; done
bash tools/diag_cfg4_timeline.sh
timeout 900 python bench.py --cfg4-full-budget-s 0 > $out/bench.json 2> $out/bench.err
