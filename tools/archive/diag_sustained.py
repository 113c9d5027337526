"""Duration of the fused LM launch when launches follow one another without a pause, and after pauses: is the sustained rate a clock / power effect? usage: diag_sustained.py"""
import glob, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
cache = sorted(glob.glob("/tmp/srba_bench_cache/caps_se2_tour_30000_seed*_*.bin"))
b = runner.CapsuleBatch.load(cache[-1]) if cache else runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=30000, seed=1, path="tour"), backend="hip", submap=10, depth=3)
ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib; hist = (C.c_double * 4)()
def one():
    lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx); lib.srba_hip_kernel_ms_history(ctx.ctx, hist, 1); return hist[0]
one()
print("back to back :", " ".join("%.2f" % one() for _ in range(12)))
for pause in (0.05, 0.2, 1.0):
    v = []
    for _ in range(6): time.sleep(pause); v.append(one())
    print("pause %.2f s  :" % pause, " ".join("%.2f" % x for x in v))
print("back to back :", " ".join("%.2f" % one() for _ in range(12)))
# what is it about the launch before? the same launches with a streaming kernel of the stepwise API (K1 over all pairs: 2.6 GB read + written) or the fused linearisation (0.8 GB) in between
def one_after(fn):
    lib.srba_hip_reset_state(ctx.ctx); fn(); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx); lib.srba_hip_kernel_ms_history(ctx.ctx, hist, 1); return hist[0]
print("after K1     :", " ".join("%.2f" % one_after(lambda: lib.srba_hip_update_spantree(ctx.ctx, 0)) for _ in range(8)))
print("after K4     :", " ".join("%.2f" % one_after(lambda: lib.srba_hip_eval_residuals(ctx.ctx, None)) for _ in range(8)))
print("back to back :", " ".join("%.2f" % one() for _ in range(8)))
