"""Device time stamps of the class launches of the fused LM kernel (srba_hip_launch_order) on the benchmark batch, several launches in a row: per plan job its delay, its grid and when its first capsule
was taken, relative to the earliest job of the launch. usage: diag_launch_stamps.py [n_kf]"""
import ctypes as C, glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
cache = sorted(glob.glob("/tmp/srba_bench_cache/caps_se2_tour_%d_seed1_*.bin" % n_kf))
b = runner.CapsuleBatch.load(cache[-1]) if cache else runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=n_kf, seed=1, path="tour"), backend="hip", submap=10, depth=3)
ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib
stamp = (C.c_int64 * 64)(); wgs = (C.c_int32 * 64)(); dly = (C.c_int32 * 64)(); held = 0
for it in range(8):
    lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx)
    n = lib.srba_hip_launch_order(ctx.ctx, stamp, wgs, dly, 64); t = np.array([stamp[j] for j in range(n)], dtype=np.int64); t0 = t[t > 0].min()
    ok = bool((t > 0).all() and (np.diff(t) >= 0).all()); held += ok
    print("launch %d: order held %s | " % (it, ok) + " ".join("j%d d%d g%d +%.0fus" % (j, dly[j], wgs[j], (t[j] - t0) / 100.0) for j in range(n)))
print("held %d of 8" % held)
