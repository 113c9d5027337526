#!/bin/bash
# round 5, session 5: what bounds the workgroup landmark kernel -- address translation? cache? ; the launch order stamps
export GPU_MAX_HW_QUEUES=16
R=$PWD; O=$R/gpurun_out/s5; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $O/counters_list.txt 2>&1; grep -i -c "counter" $O/counters_list.txt; grep -i -o "TCP_UTCL1[A-Z_0-9]*\|TCP_TCP_LATENCY[A-Z_0-9]*\|TCC_[A-Z0-9_]*HIT[A-Z_0-9]*\|TCC_[A-Z0-9_]*MISS[A-Z_0-9]*\|TCP_TA_TCP_STATE_READ[A-Z_0-9]*\|TCP_TCC_READ_REQ[A-Z_0-9]*\|TCP_TOTAL_CACHE_ACCESSES[A-Z_0-9]*\|GRBM_GUI_ACTIVE\|TCP_[A-Z_]*UTCL2[A-Z_0-9]*" $O/counters_list.txt | sort -u | head -60
n=0
for c in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "FETCH_SIZE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  n=$((n+1)); timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc -o p$n -- python $R/tools/r5_wg_check.py stereo 64 "WG_HS=1" > $O/p$n.log 2>&1; tail -1 $O/p$n.log | cut -c1-120
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); disp = collections.defaultdict(int)
for f in sorted(glob.glob("gpurun_out/s5/pmc/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_lm_wg" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]] += 1
for k in sorted(acc): print("%-40s %.4g  (%d dispatches)" % (k, acc[k], disp[k]))
PY
find $O -name "*.csv" -size +1M -delete
timeout 300 python tools/diag_launch_stamps.py > $O/launch_stamps.txt 2>&1; tail -30 $O/launch_stamps.txt
