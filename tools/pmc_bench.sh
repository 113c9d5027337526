#!/bin/bash
# SQ instruction-mix / stall counters and HBM byte counters of the fused LM kernel and of the streaming kernels on the BENCHMARK batch (separate --pmc passes, kernel trace only, as the guide prescribes).
# Output: gpurun_out/sq_summary.json (per-launch sums over the k_lm_run dispatches) -> copy to profiles/rNN_sq_summary.json
R=$PWD; mkdir -p gpurun_out/pmcb; rm -rf gpurun_out/pmcb/*
export GPU_MAX_HW_QUEUES=16
python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary > gpurun_out/pmcb_bench.json 2> gpurun_out/pmcb_bench.err   # fills the capsule cache
cd /tmp; export TMPDIR=/tmp
N=1
run() { timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmcb -o p$N -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-secondary > /dev/null 2> $R/gpurun_out/pmcb/p$N.err; N=$((N+1)); }
run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH
run SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_ADDR_CONFLICT
run FETCH_SIZE
run WRITE_SIZE
run TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum
run TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
cd $R
python - <<'PY'
import csv, glob, collections, json
d = json.loads([l for l in open('gpurun_out/pmcb_bench.json') if l.startswith('{')][-1])
# per launch: the bench (--steps 1 --warmup 0) makes 2 fused launches (functional run + timed step) and 11 launches of each streaming kernel
KERNELS = {'k_lm_run': 2, 'k_linearize': 11, 'k_assemble_se2rel': 11, 'kf_spantree': 11, 'k_residuals': 11}
res = {}
for kname, n_launch in KERNELS.items():
    out = {}
    for f in sorted(glob.glob('gpurun_out/pmcb/*counter_collection.csv')):
        # (k_lm_run = the FUSED LAUNCH: since round 4 its size classes run on three instantiations of the same loop -- k_lm_run, k_lm_run_lean, k_lm_run2 -- summed here)
        rows = [r for r in csv.DictReader(open(f)) if any((kn + '<' in r['Kernel_Name'] or kn + 'ILi' in r['Kernel_Name']) for kn in (('k_lm_run', 'k_lm_run_lean', 'k_lm_run2') if kname == 'k_lm_run' else (kname,)))]
        rows = [r for r in rows if kname == 'k_lm_run' and int(r.get('Grid_Size', 0)) >= 192 or int(r.get('Grid_Size', 1 << 20)) > 4096]   # not the single-capsule launches of the sequential harvest (when the cache is cold)
        if not rows: continue
        disp = sorted(set(int(r['Dispatch_Id']) for r in rows))
        acc = collections.defaultdict(float)
        for r in rows: acc[r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in acc.items(): out[k] = v / n_launch
        out.setdefault('_dispatches_per_launch', len(disp) / n_launch)
    if not out: continue
    tot = sum(out.get(k, 0) for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SMEM', 'SQ_INSTS_BRANCH'))
    out['_derived'] = {'wave_instructions_per_launch': tot,
                       'wait_any_over_wave_cycles': out.get('SQ_WAIT_ANY', 0) / max(1, out.get('SQ_WAVE_CYCLES', 1)),
                       'active_inst_any_over_wave_cycles': out.get('SQ_ACTIVE_INST_ANY', 0) / max(1, out.get('SQ_WAVE_CYCLES', 1)),
                       'mean_active_lanes_per_valu_inst': out.get('SQ_THREAD_CYCLES_VALU', 0) / max(1, 4 * out.get('SQ_ACTIVE_INST_VALU', 1)) if out.get('SQ_THREAD_CYCLES_VALU') else None,
                       'hbm_bytes_per_launch': {'FETCH_SIZE_KB_raw': out.get('FETCH_SIZE'), 'WRITE_SIZE_KB_raw': out.get('WRITE_SIZE'),
                                                'read_bytes_by_request_size': (128 * out.get('TCC_EA0_RDREQ_128B_sum', 0) + 64 * out.get('TCC_EA0_RDREQ_64B_sum', 0)
                                                        + 32 * out.get('TCC_EA0_RDREQ_32B_sum', 0)) if 'TCC_EA0_RDREQ_128B_sum' in out else None,
                                                'write_bytes_by_request_size': (64 * out.get('TCC_EA0_WRREQ_64B_sum', 0)
                                                        + 32 * (out.get('TCC_EA0_WRREQ_sum', 0) - out.get('TCC_EA0_WRREQ_64B_sum', 0))) if 'TCC_EA0_WRREQ_sum' in out else None,
                                                'note': 'FETCH_SIZE / WRITE_SIZE: rocprofv3 derived counters in KB, as reported. Calibrated in round 4 (profiles/r04_counter_calibration.md): '
                                                        'FETCH_SIZE = TCC_EA0_RDREQ x 64 B whatever the request size, so it halves 128-B requests (coalesced streaming reads) and is exact for the 64-B '
                                                        'requests of scattered record gathers; WRITE_SIZE is exact. read_bytes_by_request_size = 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B '
                                                        'needs no factor'}}
    res[kname] = out
res['k_lm_run']['_lm_trials_per_launch'] = d['config']['lm_trials_per_step_per_gpu']; res['k_lm_run']['_kernel_ms'] = d['roofline']['kernel_ms']
res['k_lm_run']['_derived']['wave_instructions_per_trial'] = res['k_lm_run']['_derived']['wave_instructions_per_launch'] / max(1, d['config']['lm_trials_per_step_per_gpu'])
res['_bench'] = {'value': d['value'], 'ms_per_step': d['ms_per_step'], 'capsules': d['config']['capsules_per_gpu'], 'streaming_kernels': d['streaming_kernels'], 'algorithmic_bytes_per_launch': d['roofline']['algorithmic_bytes_per_launch']}
json.dump(res, open('gpurun_out/sq_summary.json', 'w'), indent=1, sort_keys=True)
print(json.dumps({k: (v.get('_derived') if isinstance(v, dict) else v) for k, v in res.items()}, indent=1, sort_keys=True))
PY
find gpurun_out/pmcb -name "*.csv" -size +3M -delete
