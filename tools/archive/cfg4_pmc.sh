#!/bin/bash
# SQ / memory counters of the big path's two heaviest kernels (k_chol_step, kb_schur_reduce) on the cfg4 workload: separate --pmc passes with --kernel-trace only; summary -> gpurun_out/cfg4_pmc.txt
R=$PWD; O=$R/gpurun_out/pmc4; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
N=1
run() { timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O -o p$N -- python $R/bench.py --workload cfg4 --steps 2 --warmup 0 --cpu-seconds 0 --cfg4-kf 200 > /dev/null 2> $O/p$N.err; N=$((N+1)); }
run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
run FETCH_SIZE
run WRITE_SIZE
run TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
cd $R
python - <<'PY' > gpurun_out/cfg4_pmc.txt
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob('gpurun_out/pmc4/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = 'k_chol_step' if 'k_chol_step' in n else ('kb_schur_reduce' if 'kb_schur_reduce' in n else ('k_chol_bsub' if 'k_chol_bsub' in n else None))
        if not k: continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]): print('   %-34s per launch %14.1f   (launches %d)' % (c, tot[k][c] / max(1, cnt[k][c]), cnt[k][c]))
    t = tot[k]; n = lambda c: t[c] / max(1, cnt[k][c])
    if n('SQ_WAVE_CYCLES'): print('   -> waiting %.0f %% of the wave cycles, VALU issuing %.0f %%; lanes live per VALU instruction %.1f' % (100 * n('SQ_WAIT_ANY') / n('SQ_WAVE_CYCLES'), 100 * n('SQ_ACTIVE_INST_VALU') / n('SQ_WAVE_CYCLES') if n('SQ_ACTIVE_INST_VALU') else -1, n('SQ_THREAD_CYCLES_VALU') / max(1.0, n('SQ_ACTIVE_INST_VALU')) if n('SQ_THREAD_CYCLES_VALU') else -1))
PY
cat gpurun_out/cfg4_pmc.txt
rm -rf $O   # the raw counter files exceed what gpurun copies back
