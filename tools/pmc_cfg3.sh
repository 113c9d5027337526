#!/bin/bash
# L2-miss bytes per fused launch of the cfg3 workload (bench.py --workload cfg3): two --pmc passes (FETCH_SIZE, WRITE_SIZE) with the kernel trace only; bytes = 2 x FETCH_SIZE + WRITE_SIZE
# (profiles/r04_counter_calibration.md). -> gpurun_out/pmc_traffic_cfg3.json (copy to profiles/rNN_pmc_traffic_cfg3.json: bench.py --workload cfg3 prints it as roofline.traffic)
export GPU_MAX_HW_QUEUES=16
R=$PWD; O=$R/gpurun_out/pmc3; mkdir -p $O; rm -rf $O/*
python bench.py --workload cfg3 --steps 1 --warmup 0 --cpu-seconds 0 > $O/plain.json 2> $O/plain.err
cd /tmp; export TMPDIR=/tmp
n=0
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
	"SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS" \
	"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
	n=$((n+1)); timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O -o p$n -- python $R/bench.py --workload cfg3 --steps 1 --warmup 0 --cpu-seconds 0 > $O/p$n.log 2> $O/p$n.err; done
cd $R
python - <<'PY'
import csv, glob, collections, json
d = json.loads([l for l in open("gpurun_out/pmc3/plain.json") if l.startswith("{")][-1]); acc = collections.defaultdict(float); launches = 2   # the functional run + one timed step
for f in sorted(glob.glob("gpurun_out/pmc3/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if ("k_lm_run" in r["Kernel_Name"] or "k_lm_wg" in r["Kernel_Name"]) and int(r.get("Grid_Size", 0)) > 64 * 64: acc[r["Counter_Name"]] += float(r["Counter_Value"]) / launches   # (not the single-capsule launches of the map build)
rd, wr = 2 * 1024.0 * acc.get("FETCH_SIZE", 0), 1024.0 * acc.get("WRITE_SIZE", 0)
out = {"workload": {"name": "cfg3", "n_kf": d["config"]["workload"].split(":")[1].split("key-frames")[0].strip(), "local_areas": d["config"]["local_areas"], "replicas": d["config"]["replicas"], "extensions": d["config"]["extensions"]},
       "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --workload cfg3 --steps 1 --warmup 0 --cpu-seconds 0 (tools/pmc_cfg3.sh)",
       "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": d["roofline"]["algorithmic_bytes_per_launch"], "kernel_ms": d["roofline"]["kernel_ms"],
       "convention": "read bytes = 2 x FETCH_SIZE (every L2 read request is a 128-byte line on gfx950), WRITE_SIZE as reported: profiles/r04_counter_calibration.md"}
json.dump(out, open("gpurun_out/pmc_traffic_cfg3.json", "w"), indent=1); print(json.dumps(out, indent=1))
# SQ / L2 counters of the same launches (the landmark kernels: k_lm_wg + what is left on k_lm_run), per launch
trials = d["config"]["lm_trials_per_step"]
sq = {"workload": out["workload"], "kernel_ms": out["kernel_ms"], "lm_trials_per_launch": trials, "per_launch": {k: v for k, v in acc.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")}}
wc = acc.get("SQ_WAVE_CYCLES", 0)
sq["derived"] = {"wait_any_over_wave_cycles": acc.get("SQ_WAIT_ANY", 0) / max(wc, 1), "issue_over_wave_cycles": acc.get("SQ_ACTIVE_INST_ANY", 0) / max(wc, 1),
                 "mean_active_lanes_per_valu_inst": acc.get("SQ_THREAD_CYCLES_VALU", 0) / max(1, 4 * acc.get("SQ_ACTIVE_INST_VALU", 1)) if acc.get("SQ_THREAD_CYCLES_VALU") else None,
                 "wave_instructions_per_trial": (acc.get("SQ_INSTS_VALU", 0) + acc.get("SQ_INSTS_SALU", 0) + acc.get("SQ_INSTS_LDS", 0) + acc.get("SQ_INSTS_VMEM_RD", 0) + acc.get("SQ_INSTS_VMEM_WR", 0)) / max(trials, 1),
                 "l2_hit_rate": acc.get("TCC_HIT_sum", 0) / max(1, acc.get("TCC_HIT_sum", 0) + acc.get("TCC_MISS_sum", 0)), "lds_bank_conflict_over_lds_active": acc.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, acc.get("SQ_ACTIVE_INST_LDS", 1))}
json.dump(sq, open("gpurun_out/sq_summary_cfg3.json", "w"), indent=1, sort_keys=True); print(json.dumps(sq["derived"], indent=1))
PY
find $O -name "*.csv" -size +1M -delete
