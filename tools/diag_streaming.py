"""The streaming kernels of the stepwise API on the benchmark batch (capsules from bench.py's cache when there is one): ms per call and share of 8 TB/s by algorithmic bytes."""
import os, sys, time, glob
sys.path.insert(0, os.getcwd())
from srba_amd import capi, datasets, runner
cache = sorted(glob.glob("/tmp/srba_bench_cache/caps_se2_tour_30000_seed*_*.bin"))
b = runner.CapsuleBatch.load(cache[-1]) if cache else runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=30000, seed=1, path="tour"), backend="hip", submap=10, depth=3)
ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib; st = ctx.stats(); P, L, O, PD = capi.DIMS[b.family]
lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_update_spantree(ctx.ctx, 0); lib.srba_hip_eval_residuals(ctx.ctx, None); lib.srba_hip_sync(ctx.ctx)
for name, fn, by in (("k_spantree", lambda: lib.srba_hip_update_spantree(ctx.ctx, 0), st["n_path"] * (8 * PD + 4) + st["n_pairs"] * 2 * 8 * PD), ("k_residuals", lambda: lib.srba_hip_eval_residuals(ctx.ctx, None), st["n_obs"] * (8 * PD + O * 8 + 12 + O * 8))):
    fn(); lib.srba_hip_sync(ctx.ctx); t = time.perf_counter()
    for _ in range(20): fn()
    lib.srba_hip_sync(ctx.ctx); t = (time.perf_counter() - t) / 20
    print("%s %.4f ms -> %.1f %% of 8 TB/s" % (name, 1e3 * t, 100 * by / t / 8e12))
