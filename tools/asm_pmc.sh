#!/bin/bash
# HBM byte counters of the fused normal-equations kernel on the benchmark batch (separate --pmc passes, kernel trace only). Output: gpurun_out/asm_pmc/summary.txt
R=$PWD; O=$R/gpurun_out/asm_pmc; rm -rf $O; mkdir -p $O
python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary > /dev/null 2>&1   # fills the capsule cache
cd /tmp; export TMPDIR=/tmp; N=1
run() { timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O -o p$N -- python $R/tools/diag_assemble.py 30000 2 > $O/p$N.log 2>&1; N=$((N+1)); }
run FETCH_SIZE
run WRITE_SIZE
# (passes that hang rocprofv3 on this box and cost their whole timeout: TCC_EA0_RDREQ_sum / TCC_HIT_sum ..., TA_*_STALLED_*_sum, TCP_*_STALL_CYCLES_sum, TCP_*_LATENCY_sum -- removed)
run TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU
run SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
if [ "$ASM_PMC_MORE" != 0 ]; then   # where the vector-memory path stalls (round 6)
run SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM
run TD_TD_BUSY_sum TD_TC_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum
fi
cd $R
python - <<'PY' > $O/summary.txt
import csv, glob, collections
acc = collections.defaultdict(float); nd = collections.defaultdict(set)
for f in sorted(glob.glob('gpurun_out/asm_pmc/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'k_assemble' in r['Kernel_Name']: acc[r['Counter_Name']] += float(r['Counter_Value']); nd[r['Counter_Name']].add(r['Dispatch_Id'])
for k in sorted(acc): print("%-34s %18.1f per call (%d dispatches, 3 calls)" % (k, acc[k] / 3.0, len(nd[k])))
PY
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
