#!/bin/bash
# L2-miss bytes per fused launch of the cfg3 workload (bench.py --workload cfg3): two --pmc passes (FETCH_SIZE, WRITE_SIZE) with the kernel trace only; bytes = 2 x FETCH_SIZE + WRITE_SIZE
# (profiles/r04_counter_calibration.md). -> gpurun_out/pmc_traffic_cfg3.json (copy to profiles/rNN_pmc_traffic_cfg3.json: bench.py --workload cfg3 prints it as roofline.traffic)
export GPU_MAX_HW_QUEUES=16
R=$PWD; O=$R/gpurun_out/pmc3; mkdir -p $O; rm -rf $O/*
python bench.py --workload cfg3 --steps 1 --warmup 0 --cpu-seconds 0 > $O/plain.json 2> $O/plain.err
cd /tmp; export TMPDIR=/tmp
n=0
for c in FETCH_SIZE WRITE_SIZE; do n=$((n+1)); timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O -o p$n -- python $R/bench.py --workload cfg3 --steps 1 --warmup 0 --cpu-seconds 0 > $O/p$n.log 2> $O/p$n.err; done
cd $R
python - <<'PY'
import csv, glob, collections, json
d = json.loads([l for l in open("gpurun_out/pmc3/plain.json") if l.startswith("{")][-1]); acc = collections.defaultdict(float); launches = 2   # the functional run + one timed step
for f in sorted(glob.glob("gpurun_out/pmc3/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_lm_run" in r["Kernel_Name"] and int(r.get("Grid_Size", 0)) > 64 * 64: acc[r["Counter_Name"]] += float(r["Counter_Value"]) / launches   # (not the single-capsule launches of the map build)
rd, wr = 2 * 1024.0 * acc.get("FETCH_SIZE", 0), 1024.0 * acc.get("WRITE_SIZE", 0)
out = {"workload": {"name": "cfg3", "n_kf": d["config"]["workload"].split(":")[1].split("key-frames")[0].strip(), "local_areas": d["config"]["local_areas"], "replicas": d["config"]["replicas"], "extensions": d["config"]["extensions"]},
       "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --workload cfg3 --steps 1 --warmup 0 --cpu-seconds 0 (tools/pmc_cfg3.sh)",
       "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": d["roofline"]["algorithmic_bytes_per_launch"], "kernel_ms": d["roofline"]["kernel_ms"],
       "convention": "read bytes = 2 x FETCH_SIZE (every L2 read request is a 128-byte line on gfx950), WRITE_SIZE as reported: profiles/r04_counter_calibration.md"}
json.dump(out, open("gpurun_out/pmc_traffic_cfg3.json", "w"), indent=1); print(json.dumps(out, indent=1))
PY
find $O -name "*.csv" -size +1M -delete
