#!/usr/bin/env python3
"""Timeline of the multi-workgroup path from a rocprofv3 kernel trace (CSV): for the last `--ms` milliseconds of the trace, the busy time of the device, the idle time between
consecutive kernels, and per kernel name: launches, total and average duration, and the idle time that FOLLOWS its launches. usage: cfg4_timeline.py <kernel_trace.csv> [--ms 300]"""
import csv, sys, collections
path = sys.argv[1]; ms = float(sys.argv[sys.argv.index("--ms") + 1]) if "--ms" in sys.argv else 300.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_end = rows[-1][1]; t0 = t_end - int(ms * 1e6)
rows = [r for r in rows if r[0] >= t0]
busy = 0; idle = 0; per = collections.defaultdict(lambda: [0, 0, 0]); gaps = []
cur_end = rows[0][0]
for i, (s, e, n) in enumerate(rows):
    n = n.split("(")[0].replace("void ", "").replace("srbadev::", "")
    per[n][0] += 1; per[n][1] += e - s
    if s > cur_end:
        g = s - cur_end; idle += g; gaps.append(g)
        if i > 0: pn = rows[i - 1][2].split("(")[0].replace("void ", "").replace("srbadev::", ""); per[pn][2] += g
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
span = rows[-1][1] - rows[0][0]
print("window %.1f ms: %d kernels, device busy %.1f ms (%.0f %%), idle between kernels %.1f ms" % (span / 1e6, len(rows), busy / 1e6, 100.0 * busy / span, idle / 1e6))
gaps.sort()
if gaps: print("gaps: median %.1f us, p90 %.1f us, max %.1f us, over 20 us: %d (%.1f ms)" % (gaps[len(gaps) // 2] / 1e3, gaps[int(0.9 * len(gaps))] / 1e3, gaps[-1] / 1e3, sum(g > 20000 for g in gaps),
        sum(g for g in gaps if g > 20000) / 1e6))
print("%-28s %8s %10s %9s %12s" % ("kernel", "calls", "total ms", "avg us", "idle after ms"))
for n, (c, d, g) in sorted(per.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print("%-28s %8d %10.2f %9.1f %12.2f" % (n[:28], c, d / 1e6, d / 1e3 / c, g / 1e6))
