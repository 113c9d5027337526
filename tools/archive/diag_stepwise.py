"""Batch-wide stepwise kernels (one launch per phase over all capsules): time and algorithmic HBM bytes -> GB/s per kernel.
These are the streaming kernels of the C ABI (srba_hip_update_spantree / eval_residuals / linearize / apply_update); the fused LM kernel keeps
the same data in a capsule's working set instead. usage: diag_stepwise.py [n_kf]"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
ds = datasets.graph_slam_se2(n_kf=n_kf, seed=1, path="tour")
b = runner.harvest_graph_slam(ds, backend="hip", submap=10, depth=3)
ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib
st = ctx.stats(); P, L, O, PD = capi.DIMS[b.family]; pb = 8 * PD
def timed(fn, reps=10):
    fn(); lib.srba_hip_sync(ctx.ctx)
    t = time.perf_counter()
    for _ in range(reps): fn()
    lib.srba_hip_sync(ctx.ctx)
    return (time.perf_counter() - t) / reps
lib.srba_hip_reset_state(ctx.ctx)
rows = []
t = timed(lambda: lib.srba_hip_update_spantree(ctx.ctx, 0)); by = st["n_path"] * (pb + 4) + st["n_pairs"] * 2 * pb; rows.append(("K1 spanning tree (all pairs)", t, by))
t = timed(lambda: lib.srba_hip_eval_residuals(ctx.ctx, None)); by = st["n_obs"] * (pb + O * 8 + 12 + O * 8); rows.append(("K4 residuals", t, by))
t = timed(lambda: lib.srba_hip_linearize(ctx.ctx)); by = st["n_bp"] * (2 * pb + pb + 16 + O * P * 8) + st["n_hap"] * P * P * 8 + st["n_hap_terms"] * 2 * O * P * 8 + st["n_unk_edges"] * P * 8; rows.append(("K2+K6+K5 linearize (J, H, g)", t, by))
for name, t, by in rows:
    print("%-32s %8.3f ms  %8.1f MB algorithmic  -> %7.1f GB/s (%.1f%% of 8 TB/s)" % (name, 1e3 * t, by / 1e6, by / t / 1e9, 100 * by / t / 8e12))
print("batch: %d capsules, %d observations, %d Jacobian blocks, %d H blocks, %d H terms" % (b.n, st["n_obs"], st["n_bp"], st["n_hap"], st["n_hap_terms"]))
