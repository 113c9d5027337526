#!/bin/bash
# SQ instruction-mix counters of the fused LM kernel on the replicated-capsule microbenchmark (two PMC passes)
R=$PWD; mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp OCC_PADS=0
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc -o p1 -- python $R/tools/diag_occupancy.py "$@" 2>&1 | tail -2
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $R/gpurun_out/pmc -o p2 -- python $R/tools/diag_occupancy.py "$@" 2>&1 | tail -2
cd $R; ls gpurun_out/pmc
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/*counter_collection.csv')):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'k_lm_run' in r['Kernel_Name']: acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for k in acc: print(f.split('/')[-1], k, acc[k], 'over', n[k], 'dispatches')
PY
