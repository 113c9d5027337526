#!/bin/bash
# Instruction-mix / stall / L1 counters of the fused LM kernel on the replicated-capsule microbenchmark (separate PMC passes)
R=$PWD; mkdir -p gpurun_out/pmc; rm -rf gpurun_out/pmc/*; cd /tmp; export TMPDIR=/tmp OCC_PADS=0
run() { rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc -o p$N -- python $R/tools/diag_occupancy.py $ARGS > /dev/null 2>&1; N=$((N+1)); }
N=1; ARGS="$*"
run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH
# (hung on this pool) run TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
# (hung) run TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
# (hung) run TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/*counter_collection.csv')):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_lm_run' in r['Kernel_Name'] and int(r['Grid_Size']) > 64 * 64: acc[r['Counter_Name']] += float(r['Counter_Value'])
    for k in acc: print(f.split('/')[-1][:3], k, "%.4g" % acc[k])
PY
