#!/bin/bash
# HBM byte counters and SQ instruction mix / stall counters of the fused LM kernel on a landmark-family batch (tools/diag_family.py: 59 windows x 64 replicas),
# separate --pmc passes with the kernel trace only. usage: tools/pmc_family.sh [kind]   -> gpurun_out/pmc_family_<kind>.json (copy to profiles/rNN_pmc_<kind>.json)
kind=${1:-stereo}
export GPU_MAX_HW_QUEUES=16
R=$PWD; O=$R/gpurun_out/pmcf; mkdir -p $O; rm -rf $O/*
cd /tmp; export TMPDIR=/tmp
n=0
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS"; do
  n=$((n+1)); timeout 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O -o p$n -- python $R/tools/diag_family.py $kind 64 > $O/p$n.log 2>&1
done
cd $R
python - "$kind" <<'PY'
import csv, glob, collections, json, re, sys
kind = sys.argv[1]; acc = collections.defaultdict(float); launches = 2   # diag_family.py runs the batch twice; the single-capsule launches of the map build (grid of one workgroup) are left out
for f in sorted(glob.glob("gpurun_out/pmcf/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_lm_run" in r["Kernel_Name"] and int(r.get("Grid_Size", 0)) > 64: acc[r["Counter_Name"]] += float(r["Counter_Value"]) / launches
line = [l for l in open("gpurun_out/pmcf/p1.log") if "LM iterations/s" in l][-1]
ms = float(re.search(r"GPU ([0-9.]+) ms", line).group(1))
out = {"batch": line.strip()[:200], "kernel_ms": ms, "per_launch": dict(acc)}
f, w = acc.get("FETCH_SIZE", 0) * 1024, acc.get("WRITE_SIZE", 0) * 1024
out["derived"] = {"fetch_GB": f / 1e9, "write_GB": w / 1e9, "hbm_TB_per_s": (f + w) / 1e12 / (ms * 1e-3),
                  "wait_any_over_wave_cycles": acc.get("SQ_WAIT_ANY", 0) / max(1, acc.get("SQ_WAVE_CYCLES", 1)), "issue_over_wave_cycles": acc.get("SQ_ACTIVE_INST_ANY", 0) / max(1, acc.get("SQ_WAVE_CYCLES", 1)),
                  "mean_active_lanes_per_valu_inst": acc.get("SQ_THREAD_CYCLES_VALU", 0) / max(1, 4 * acc.get("SQ_ACTIVE_INST_VALU", 1)) if acc.get("SQ_THREAD_CYCLES_VALU") else None,
                  "vmem_read_instructions_per_CU": acc.get("SQ_INSTS_VMEM_RD", 0) / 256,
                  "note": "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (MI355X_MICROARCH.md: wide coalesced reads are tallied at half on gfx950; these kernels read 16 bytes per lane, uncalibrated)"}
json.dump(out, open("gpurun_out/pmc_family_%s.json" % kind, "w"), indent=1, sort_keys=True)
print(json.dumps(out["derived"], indent=1))
PY
find $O -name "*.csv" -size +1M -delete
