"""Per-launch view of a rocprofv3 --kernel-trace csv of tools/diag_assemble.py: grid size, LDS, duration, start offset inside its call. usage: assemble_trace.py trace.csv"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_assemble" in r["Kernel_Name"] or "k_linearize" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups = collections.defaultdict(list)
for r in rows: groups[(r["Kernel_Name"][:40], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["LDS_Block_Size"]))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print("%-42s %8s %8s %6s %10s" % ("kernel", "capsules", "LDS", "n", "mean us"))
for k, v in sorted(groups.items(), key=lambda kv: -kv[0][2]): print("%-42s %8d %8d %6d %10.1f" % (k[0], k[1], k[2], len(v), sum(e - s for s, e in v) / len(v) / 1e3))
# the last call: span of its launches
n = len(groups); last = rows[-n:]; t0 = min(int(r["Start_Timestamp"]) for r in last); t1 = max(int(r["End_Timestamp"]) for r in last)
print("last call: %.1f us from first start to last end" % ((t1 - t0) / 1e3))
for r in last: print("   start +%7.1f us  dur %7.1f us  capsules %6d  LDS %6s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["LDS_Block_Size"]))
