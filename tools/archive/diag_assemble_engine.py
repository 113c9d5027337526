import os, sys, glob, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
mode = sys.argv[1]
eng = None
if mode in ("engine", "engine_closed"):
    b0 = runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=300, seed=1, path="tour"), backend="hip", submap=10, depth=3); eng = b0.engine
    if mode == "engine_closed": eng.close(); del b0
b = runner.CapsuleBatch.load(sorted(glob.glob("/tmp/srba_bench_cache/caps_se2_tour_30000_seed*_*.bin"))[-1])
ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib
lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_update_spantree(ctx.ctx, 0); lib.srba_hip_eval_residuals(ctx.ctx, None); lib.srba_hip_linearize(ctx.ctx); lib.srba_hip_sync(ctx.ctx)
t = time.perf_counter()
for _ in range(10): lib.srba_hip_linearize(ctx.ctx)
lib.srba_hip_sync(ctx.ctx); print(mode, "linearize %.4f ms" % (1e3 * (time.perf_counter() - t) / 10))
