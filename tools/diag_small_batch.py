"""Fixed cost of a small batch through the C ABI (what a round of a map sweep pays): upload / lm_run / download_state of the first n capsules of a harvested map, per call.
usage: diag_small_batch.py [n ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srba_amd import capi, datasets, runner
b = runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=1200, seed=1, path="tour"), backend="hip", submap=10, depth=3)
ctx = runner.HipContext(b.params); lib = ctx.lib
for n in [int(x) for x in sys.argv[1:]] or [1, 30, 150, 600]:
    s = b.sub(200, n).clone(); res = (capi.LmResult * n)(); t = np.zeros(3)
    for rep in range(12):
        t0 = time.perf_counter(); ctx.upload(s); t1 = time.perf_counter(); ctx._chk(lib.srba_hip_lm_run(ctx.ctx, res), "lm_run"); t2 = time.perf_counter()
        ctx._chk(lib.srba_hip_download_state(ctx.ctx, s.ptr, n), "download"); t3 = time.perf_counter()
        if rep >= 2: t += (t1 - t0, t2 - t1, t3 - t2)
    t *= 1e3 / 10
    print("n = %4d: upload %.3f ms, lm_run %.3f ms (kernel %.3f), download_state %.3f ms -> %.3f ms per capsule" % (n, t[0], t[1], lib.srba_hip_last_kernel_ms(ctx.ctx), t[2], t.sum() / n))
