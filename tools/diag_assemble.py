"""srba_hip_linearize on the benchmark batch (30 000-key-frame tour, capsules from bench.py's cache when there is one): time per call; run it under
rocprofv3 --kernel-trace to see the launches of the LDS size classes one by one (tools/assemble_trace.py prints them). usage: diag_assemble.py [n_kf] [reps]"""
import glob, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # the repo root
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 30000; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cache = sorted(glob.glob("/tmp/srba_bench_cache/caps_se2_tour_%d_seed*_*.bin" % n_kf))
if cache: b = runner.CapsuleBatch.load(cache[-1])
else: b = runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=n_kf, seed=1, path="tour"), backend="hip", submap=10, depth=3)
ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib; st = ctx.stats(); P, L, O, PD = capi.DIMS[b.family]
lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_update_spantree(ctx.ctx, 0); lib.srba_hip_eval_residuals(ctx.ctx, None); lib.srba_hip_linearize(ctx.ctx); lib.srba_hip_sync(ctx.ctx)
t = time.perf_counter()
for _ in range(reps): lib.srba_hip_linearize(ctx.ctx)
lib.srba_hip_sync(ctx.ctx); t = (time.perf_counter() - t) / reps
by = st["n_bp"] * (3 * 24 + 16) + st["n_hap"] * P * P * 8 + st["n_unk_edges"] * P * 8
print("linearize %.4f ms ; fused bytes %.1f MB -> %.1f GB/s (%.1f %% of 8 TB/s); %d capsules, %d blocks, %d H blocks, %d terms" % (1e3 * t, by / 1e6, by / t / 1e9, 100 * by / t / 8e12, b.n, st["n_bp"], st["n_hap"], st["n_hap_terms"]))
if os.environ.get("SRBA_HIP_PHASE_TIMING") == "1":   # per-capsule phase ticks of the last call (100 MHz): [start, end of A, end of B, end]
    tk = ctx.debug(10).reshape(b.n, 16)[:, :4]; nbp = np.array([b[i].n_bp for i in range(b.n)])
    ok = tk[:, 3] > 0; a, bb, cc = (tk[:, 1] - tk[:, 0]) / 100.0, (tk[:, 2] - tk[:, 1]) / 100.0, (tk[:, 3] - tk[:, 2]) / 100.0
    print("per capsule (us): A %.1f  B %.1f  C %.1f  total %.1f ; whole call from first start to last end %.1f us" % (a[ok].mean(), bb[ok].mean(), cc[ok].mean(), (a + bb + cc)[ok].mean(), (tk[ok, 3].max() - tk[ok, 0].min()) / 100.0))
    for lo, hi in ((0, 150), (150, 200), (200, 300), (300, 400), (400, 600), (600, 1000), (1000, 100000)):
        m = ok & (nbp >= lo) & (nbp < hi)
        if m.any(): print("  blocks %4d..%-6d %6d capsules: A %.1f  B %.1f  C %.1f us" % (lo, hi, m.sum(), a[m].mean(), bb[m].mean(), cc[m].mean()))
    tk8 = ctx.debug(10).reshape(b.n, 16)[:, :8]
    if (tk8[:, 7] > 0).any():   # library built with -DSRBA_ASM_TICKS: phase A in pieces
        m = ok & (tk8[:, 7] > 0); d = lambda i, j: ((tk8[m, i] - tk8[m, j]) / 100.0).mean()
        print("phase A in pieces (us): start -> records + edge poses in %.1f | -> first gathers in %.1f | -> first group computed %.1f | -> all groups %.1f | -> scan + emits %.1f" % (d(4, 0), d(5, 4), d(6, 5), d(7, 6), d(1, 7)))
    hw = tk8[:, 4].astype(np.int64)
    if (hw != 0).any():   # where every capsule ran: per-CU residency over the call
        cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xff); t0, t1 = tk[:, 0].astype(np.int64), tk[:, 3].astype(np.int64); T0, T1 = t0[ok].min(), t1[ok].max()
        print("CUs seen: %d ; waves seen per SIMD slot ids: %s" % (len(np.unique(cu[ok])), np.unique(hw[ok] & 0xf)))
        ev = np.concatenate([np.stack([t0[ok], np.ones(ok.sum(), np.int64)], 1), np.stack([t1[ok], -np.ones(ok.sum(), np.int64)], 1)]); ev = ev[np.argsort(ev[:, 0], kind="stable")]
        infl = np.cumsum(ev[:, 1]); dt = np.diff(ev[:, 0]); avg = (infl[:-1] * dt).sum() / max(1, (T1 - T0))
        print("capsules in flight: time-average %.0f (%.2f per CU), peak %d ; by tenth of the call: %s" % (avg, avg / 256, infl.max(), " ".join("%.0f" % ((infl[:-1] * dt)[(ev[:-1, 0] >= T0 + (T1 - T0) * q / 10) & (ev[:-1, 0] < T0 + (T1 - T0) * (q + 1) / 10)].sum() / ((T1 - T0) / 10.0)) for q in range(10))))
        starts = np.sort(t0[ok] - T0) / 100.0; print("capsule starts (us after the first): 10%% %.0f  50%% %.0f  90%% %.0f  last %.0f ; ends: last %.0f" % (starts[len(starts) // 10], starts[len(starts) // 2], starts[9 * len(starts) // 10], starts[-1], (T1 - T0) / 100.0))
