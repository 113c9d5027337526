"""Per-class cost breakdown of the fused LM kernel on (a prefix of) the benchmark batch. usage: diag_bench.py [n_kf]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SRBA_HIP_PHASE_TIMING"] = "1"
from srba_amd import capi, datasets, runner
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
ds = datasets.graph_slam_se2(n_kf=n_kf, seed=1, path="tour")
t = time.time(); b = runner.harvest_graph_slam(ds, backend="hip", submap=10, depth=3); print("harvest(hip) %.1fs capsules %d" % (time.time() - t, b.n))
ctx = runner.HipContext(b.params); ctx.upload(b)
gpu = ctx.lm_run(); kms = ctx.lib.srba_hip_last_kernel_ms(ctx.ctx)
pc = ctx.debug(10).reshape(b.n, 16); shp = ctx.debug(11).reshape(b.n, 4)
tot_cyc = pc[:, :9].sum(axis=1)  # 100 MHz ticks
print("kernel %.1f ms (with phase timers), trials %d" % (kms, gpu["num_trials"].sum()))
lds = shp[:, 0]; t_ms = tot_cyc * 1e-5
print("sum(wave time) %.1f s ; mean concurrency %.0f waves ; sum(LDS*time)/(256 CU * 160 KB) = %.1f ms" % (t_ms.sum() * 1e-3, t_ms.sum() / kms, (lds * t_ms).sum() / (256 * 160 * 1024)))
names = ["K1all", "jac", "hess", "resid", "grad", "solve", "apply", "K1need", "restore", "schur", "assemble", "factor", "bsub", "feat"]
for lo, hi in ((0, 16), (16, 24), (24, 32), (32, 48), (48, 64), (64, 96), (96, 160), (160, 1e9)):
    m = (lds > lo * 1024) & (lds <= hi * 1024) if lo else (lds <= hi * 1024)
    if hi > 1e6: m = lds == 0
    if not m.any(): continue
    tr = gpu["num_trials"][m].sum(); tt = pc[m].sum(axis=0)
    print("LDS (%3d,%3d] KB: %5d caps, nb %.0f nnzoff %.0f items %.0f, %6d trials (%.1f/cap); us/trial: total %.0f |" % (lo, min(hi, 999), m.sum(), shp[m, 1].mean(), shp[m, 2].mean(), shp[m,
            3].mean(), tr, tr / m.sum(), tt[:9].sum() * 1e-2 / tr),
          " ".join("%s %.1f" % (names[k], tt[k] * 1e-2 / tr) for k in (11, 12, 10, 7, 3, 2, 4, 6, 1,
                  0)), "| share of LDS*time %.0f%%" % (100 * (lds[m] * t_ms[m]).sum() / (lds * t_ms).sum()), "share of wave time %.0f%%" % (100 * t_ms[m].sum() / t_ms.sum()))
