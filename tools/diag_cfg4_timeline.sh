#!/bin/bash
# (GPU box) where a cfg4 step's wall time goes: kernel trace of `bench.py --workload cfg4`, then the last two timed steps of the trace split into kernel-busy time per kernel, idle gaps
# between consecutive kernels (dependent launches / host round trips) and their histogram. Output: gpurun_out/cfg4_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/cfg4_tl; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o tl -- python bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-seconds 0 > $out/bench.json 2> $out/bench.err
python - <<'PY' > gpurun_out/cfg4_timeline.txt
import csv, json, glob, collections
line = json.loads(open('gpurun_out/cfg4_tl/bench.json').read().strip().split('\n')[-1]); ms = line['ms_per_step']
f = glob.glob('gpurun_out/cfg4_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]) for r in csv.DictReader(open(f))]
rows.sort(); t_end = max(r[1] for r in rows); t0 = t_end - int(2 * ms * 1e6)
sel = [r for r in rows if r[0] >= t0]
busy = collections.Counter(); cnt = collections.Counter(); gaps = []
cur_end = sel[0][0]; union = 0
for s, e, n in sel:
    busy[n] += e - s; cnt[n] += 1
    if s > cur_end: gaps.append((s - cur_end, n)); union += e - s; cur_end = e
    elif e > cur_end: union += e - cur_end; cur_end = e
wall = t_end - sel[0][0]
print("cfg4: ms_per_step %.2f ; window = last two steps = %.2f ms ; kernels %d ; GPU busy (union) %.2f ms = %.1f %% ; idle %.2f ms in %d gaps" % (ms, wall / 1e6, len(sel), union / 1e6, 100 * union / wall, (wall - union) / 1e6, len(gaps)))
print("value %.0f it/s, trials per step %s, launch sequences %s" % (line['value'], line['config'].get('lm_trials_per_step'), line['roofline'].get('launch_sequences')))
for n, t in busy.most_common(25): print("  %-42s calls %6d  %8.3f ms  avg %7.1f us  %5.1f %% of wall" % (n, cnt[n], t / 1e6, t / cnt[n] / 1e3, 100 * t / wall))
h = collections.Counter()
for g, n in gaps: h[min(int(g / 1000) // 5 * 5, 200)] += g
print("idle time by gap length (us bucket: total ms):", ", ".join("%d+: %.2f" % (k, v / 1e6) for k, v in sorted(h.items())))
big = collections.Counter()
for g, n in gaps:
    if g > 20000: big[n] += g
print("gaps > 20 us, by the kernel that follows (ms):", ", ".join("%s %.2f" % (k, v / 1e6) for k, v in big.most_common(12)))
PY
