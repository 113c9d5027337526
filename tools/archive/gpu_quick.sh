#!/bin/bash
# Quick GPU iteration: parity tests, short bench (no CPU leg), per-phase breakdown. Output under gpurun_out/quick_*.
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/quick_pytest.log
python bench.py --steps 10 --warmup 2 --cpu-seconds 0 > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err
python tools/diag_bench.py > gpurun_out/quick_diag.log 2>&1
tail -3 gpurun_out/quick_pytest.log; python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/quick_bench.json") if l.startswith("{")][-1])
    print("value %.3f M it/s  ms_per_step %.2f  kernel_ms %.2f  frac %.4f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
    for k in d["streaming_kernels"]: print("  %-60s %.2f ms  %.1f%% of HBM peak" % (k["kernel"], k["ms"], 100 * k["frac_of_hbm_peak"]))
    print("  setup", d["config"]["setup_s"], "seq ms/kf", d["config"].get("sequential_ms_per_kf"))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/quick_bench.err").read()[-2000:])
PY
tail -6 gpurun_out/quick_diag.log
