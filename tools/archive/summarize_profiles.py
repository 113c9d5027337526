"""Condense the rocprofv3 CSVs of tools/gpu_round.sh: per-launch fork..join duration of the fused LM kernel from the kernel trace, the
per-kernel stats, and the HBM byte counters (FETCH_SIZE / WRITE_SIZE, separate passes) summed over the dispatches of ONE fused launch."""
import csv, glob, json, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = {}

def rows(pattern):
    f = sorted(glob.glob(os.path.join(root, pattern), recursive=True))
    return list(csv.DictReader(open(f[0]))) if f else []

def launches(tr):
    """group k_lm_run dispatches into fused launches: a new launch starts when a dispatch starts after every earlier one has ended (+ gap)"""
    ks = sorted([r for r in tr if "k_lm_run" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
    groups, cur, cur_end = [], [], -1
    for r in ks:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if cur and s > cur_end + 200000: groups.append(cur); cur = []   # > 0.2 ms idle gap = next launch (reset_state copy in between)
        cur.append(r); cur_end = max(cur_end, e) if cur[:-1] else e
    if cur: groups.append(cur)
    return groups

tr = rows("prof/**/*kernel_trace.csv")
if tr:
    g = [x for x in launches(tr) if len(x) > 4]          # the single-capsule launches of the harvest are 1 dispatch each
    d = [(max(int(r["End_Timestamp"]) for r in x) - min(int(r["Start_Timestamp"]) for r in x)) / 1e6 for x in g]
    out["kernel_trace"] = {"fused_launches": len(g), "dispatches_per_launch": len(g[-1]) if g else 0, "fork_join_ms": d,
                           "sum_of_dispatch_ms_last_launch": sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in g[-1]) if g else None}
st = rows("prof/**/*kernel_stats.csv")
if st: out["kernel_stats_top"] = st[:6]
for name, pat in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv"), ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv")):
    cc = rows(pat)
    if not cc: continue
    by_disp = {}
    for r in cc:
        if "k_lm_run" in r["Kernel_Name"] and r["Counter_Name"] == name: by_disp[int(r["Dispatch_Id"])] = (float(r["Counter_Value"]), int(r["Grid_Size"]) if "Grid_Size" in r else 0)
    ids = sorted(by_disp)
    # the last fused launch = the trailing run of dispatches with more than one workgroup
    tail = []
    for i in reversed(ids):
        tail.append(by_disp[i][0])
        if len(tail) >= out.get("kernel_trace", {}).get("dispatches_per_launch", 1 << 30): break
    out[name] = {"unit": "KB (rocprofv3 derived counter)", "dispatches": len(tail), "sum_last_launch": sum(tail)}
print(json.dumps(out, indent=1))
