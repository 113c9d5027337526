"""Why does the fused LM launch last 37.8 ms after an idle period and 41.5-43 ms back to back? Clock traces while the launches run:
  * shader clock from INSIDE the device: a one-wavefront probe kernel (tools/calib/clock_probe.hip) samples the constant 100 MHz counter and the shader-cycle counter every ~25 us on its own
    stream while the launches run; the ratio of the increments is the clock the chip actually ran at (10 us resolution, no driver in the loop);
  * what the driver reports, sampled by a host thread as fast as sysfs answers: sclk / mclk / fclk / socclk (pp_dpm_*), power.
usage (GPU box): python tools/diag_clocks.py [n_kf] > gpurun_out/clock_trace.txt"""
import ctypes as C, glob, os, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16"); os.environ.setdefault("SRBA_HIP_CLASS_STREAMS", "10")   # context stream + 10 class streams + the probe stream: every stream gets its own hardware queue (a probe that shares a queue with a class launch would hold that launch back until it ends)
from srba_amd import capi, datasets, runner
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
so = "/tmp/libclock_probe.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "tools", "calib", "clock_probe.hip"), "-o", so], check=True)
probe = C.CDLL(so); probe.probe_start.argtypes = [C.c_int, C.c_int]; probe.probe_stop.argtypes = [C.POINTER(C.c_ulonglong)]
cache = sorted(glob.glob("/tmp/srba_bench_cache/caps_se2_tour_%d_seed1_*.bin" % n_kf))
b = runner.CapsuleBatch.load(cache[-1]) if cache else runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=n_kf, seed=1, path="tour"), backend="hip", submap=10, depth=3)
ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib; hist = (C.c_double * 4)()
def one():
    lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx); lib.srba_hip_kernel_ms_history(ctx.ctx, hist, 1); return hist[0]
one(); one()
# ---- host-side sysfs sampler
dev = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
devdir = os.path.dirname(dev[0]) if dev else None
def cur(path):
    try:
        for l in open(path):
            if l.strip().endswith("*"): return l.split(":")[1].strip().rstrip("*").strip()
    except Exception: return None
hw = sorted(glob.glob(os.path.join(devdir, "hwmon", "hwmon*"))) if devdir else []
def rd(path):
    try: return open(path).read().strip()
    except Exception: return None
samples = []; run = [True]
def sampler():
    while run[0]:
        t = time.perf_counter(); row = [t]
        for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"): row.append(cur(os.path.join(devdir, f)) if devdir else None)
        row.append(rd(os.path.join(hw[0], "power1_average")) if hw else None); row.append(rd(os.path.join(hw[0], "freq1_input")) if hw else None)
        samples.append(row)
th = threading.Thread(target=sampler, daemon=True); th.start()
# ---- schedule: idle 0.3 s | 12 launches back to back | 5 launches with 0.3 s pauses | 6 back to back
N = 120000; assert probe.probe_start(N, 6) == 0   # ~25 us per sample -> 3 s of trace
marks = []; t0 = time.perf_counter()
def mark(name, ms=None): marks.append((time.perf_counter() - t0, name, ms))
time.sleep(0.3); mark("start back-to-back")
for i in range(12): ms = one(); mark("b2b %d" % i, ms)
for i in range(5): time.sleep(0.3); mark("pause end"); ms = one(); mark("paused %d" % i, ms)
for i in range(6): ms = one(); mark("b2b2 %d" % i, ms)
time.sleep(0.1)
buf = (C.c_ulonglong * (2 * N + 1))(); assert probe.probe_stop(buf) == 0; run[0] = False; th.join()
a = np.frombuffer(buf, dtype=np.uint64)[:2 * N].reshape(N, 2).astype(np.int64); k = int(buf[2 * N]); a = a[:k]
tw = (a[:, 0] - a[0, 0]) * 1e-8; dw = np.diff(a[:, 0]); ds = np.diff(a[:, 1]); mhz = ds / np.maximum(dw, 1) * 100.0
print("probe samples %d over %.3f s (every %.1f us); shader clock from the device counters: median %.0f MHz, min %.0f, max %.0f" % (k, tw[-1], 1e6 * tw[-1] / k, np.median(mhz), mhz.min(), mhz.max()))
print("\nlaunch by launch (kernel ms from the library's events; shader clock = mean over the probe samples that fall inside the launch; the probe starts 0.3 s before the first launch):")
prev = 0.3
for t, name, ms in marks:
    if ms is None: prev = t; continue
    sel = (tw[1:] >= t - ms * 1e-3) & (tw[1:] <= t)
    idle = (tw[1:] >= prev) & (tw[1:] < t - ms * 1e-3)
    print("  %-12s ends at %.3f s  kernel %.2f ms  shader clock during %.0f MHz (min %.0f)  | before it: %.0f MHz over %.1f ms" % (name, t, ms, mhz[sel].mean() if sel.any() else float("nan"), mhz[sel].min() if sel.any() else float("nan"), mhz[idle].mean() if idle.any() else float("nan"), 1e3 * (t - ms * 1e-3 - prev)))
    prev = t
print("\nshader clock, 5 ms bins over the whole trace:")
edges = np.arange(0, tw[-1], 0.005)
print("  " + " ".join("%.0f" % mhz[(tw[1:] >= e) & (tw[1:] < e + 0.005)].mean() if ((tw[1:] >= e) & (tw[1:] < e + 0.005)).any() else "nan" for e in edges))
print("\ndriver-reported clocks (sysfs %s), %d samples (every %.2f ms): distinct (sclk, mclk, fclk, socclk) tuples with their share, power range" % (devdir, len(samples), 1e3 * (samples[-1][0] - samples[0][0]) / max(1, len(samples))))
import collections
cnt = collections.Counter(tuple(r[1:5]) for r in samples)
for kk, v in cnt.most_common(12): print("  %s : %.1f %%" % (kk, 100.0 * v / len(samples)))
pw = [float(r[5]) * 1e-6 for r in samples if r[5]]; fq = [float(r[6]) * 1e-6 for r in samples if r[6]]
if pw: print("  power1_average %.0f .. %.0f W" % (min(pw), max(pw)))
if fq: print("  hwmon freq1_input %.0f .. %.0f MHz" % (min(fq), max(fq)))
# clock tuples by phase
def phase(tlo, thi): return collections.Counter(tuple(r[1:5]) for r in samples if tlo <= r[0] - t0 < thi).most_common(3)
b0 = marks[0][0]; b1 = [m for m in marks if m[1] == "b2b 11"][0][0]
print("  during the 12 back-to-back launches:", phase(b0, b1)); print("  during the paused launches:", phase(b1, [m for m in marks if m[1] == "paused 4"][0][0]))
