"""Copies the summaries of the last tools/gpu_round.sh session (gpurun_out/round/) into profiles/rNN_*. usage: collect_profiles.py NN"""
import json, os, shutil, sys
rnd = int(sys.argv[1]); src = os.path.join("gpurun_out", "round"); dst = "profiles"; tag = "r%02d_" % rnd
def last_json(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])
bench = last_json(os.path.join(src, "bench.json")); cfg4 = bench["secondary_workloads"]["cfg4"]; sq = json.load(open(os.path.join(src, "sq_summary.json")))
json.dump(bench, open(os.path.join(dst, tag + "bench.json"), "w"), indent=1)
json.dump(sq, open(os.path.join(dst, tag + "sq_summary.json"), "w"), indent=1, sort_keys=True)
for a, b in (("prof/lm_kernel_stats.csv", "bench_kernel_stats.csv"), ("prof_cfg4/c4_kernel_stats.csv", "cfg4_kernel_stats.csv"), ("prof/lm_domain_stats.csv", "bench_domain_stats.csv"),
        ("pytest_gpu.log", "pytest_gpu.log"), ("smoke.log", "smoke.log"), ("families.log", "families.log"), ("soak.log", "soak_parity.log"),
        ("pmc_traffic_cfg3.json", "pmc_traffic_cfg3.json"), ("sq_summary_cfg3.json", "sq_summary_cfg3.json"), ("eight_ranks_full_map.log", "eight_ranks_full_map.log"), ("launch_order.txt",
                "launch_order.txt"), ("cfg4_timeline.txt", "cfg4_timeline.txt"), ("assemble_timeline.txt", "assemble_timeline.txt"), ("assemble_pmc.txt", "assemble_pmc.txt"), ("sweep.txt", "sweep.txt")):
    if os.path.exists(os.path.join(src, a)): shutil.copy(os.path.join(src, a), os.path.join(dst, tag + b))
lm = sq["k_lm_run"]; fetch_kb, write_kb = lm["FETCH_SIZE"], lm["WRITE_SIZE"]
traffic = {"round": rnd, "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 (tools/pmc_bench.sh)",
           "workload": {"n_kf": bench["config"]["keyframes_per_gpu"], "capsules": bench["config"]["capsules_per_gpu"]},
           "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb,
           "read_bytes_by_request_size": lm["_derived"]["hbm_bytes_per_launch"].get("read_bytes_by_request_size"),
                   "write_bytes_by_request_size": lm["_derived"]["hbm_bytes_per_launch"].get("write_bytes_by_request_size"),
           "correction": "round 4 calibration (profiles/r04_counter_calibration.md): on gfx950 EVERY read request of the L2 to the fabric is a 128-byte line (TCC_EA0_RDREQ_128B "
                   "= TCC_EA0_RDREQ for streaming reads and for 8 .. 72-byte gathers alike) while FETCH_SIZE prices each at 64 B: read bytes = 2 x FETCH_SIZE = 128 x RDREQ_128B "
                   "+ 64 x RDREQ_64B + 32 x RDREQ_32B; WRITE_SIZE = 64 x WRREQ_64B + 32 x (WRREQ - WRREQ_64B) is exact. traffic_bytes_per_launch uses the request-size counters "
                   "(rounds 1-3 restated with the same convention in profiles/README.md)",
           "traffic_bytes_per_launch": float(lm["_derived"]["hbm_bytes_per_launch"].get("read_bytes_by_request_size") or 2 * 1024.0 * fetch_kb)
           + float(lm["_derived"]["hbm_bytes_per_launch"].get("write_bytes_by_request_size") or 1024.0 * write_kb),
           "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
           "streaming_kernels_kb_per_launch": {k: {"FETCH_SIZE": sq[k].get("FETCH_SIZE"), "WRITE_SIZE": sq[k].get("WRITE_SIZE")} for k in ("k_linearize", "kf_spantree", "k_residuals") if k in sq}}
json.dump(traffic, open(os.path.join(dst, tag + "pmc_traffic.json"), "w"), indent=1)
print("value %.3f M it/s, %.2f ms/step, kernel %.2f ms, frac %.4f | cfg4 %.0f it/s | traffic %.1f GB (fetch %.1f + write %.1f)" % (bench["value"] / 1e6, bench["ms_per_step"],
        bench["roofline"]["kernel_ms"], bench["roofline"]["frac"], cfg4["value"], traffic["traffic_bytes_per_launch"] / 1e9, fetch_kb * 1024 / 1e9, write_kb * 1024 / 1e9))
