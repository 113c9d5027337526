"""Smallest check of the rounds path with host-side tracing (SRBA_HIP_ROUNDS_DEBUG=3: a synchronisation after every round). usage: diag_rounds_small.py [n_kf]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from srba_amd import capi, datasets, runner
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ds = datasets.graph_slam_se2(n_kf=n_kf, seed=3, path="tour"); b = runner.harvest_graph_slam(ds, backend="hip", submap=10, depth=3)
print("capsules", b.n, flush=True)
os.environ["SRBA_HIP_ROUNDS"] = "0"; ctx = runner.HipContext(b.params); ctx.upload(b); a = ctx.lm_run(); ctx.close(); print("fused done, trials", a["num_trials"].sum(), "max", a["num_trials"].max(), flush=True)
os.environ["SRBA_HIP_ROUNDS"] = "1"; os.environ["SRBA_HIP_ROUNDS_DEBUG"] = sys.argv[2] if len(sys.argv) > 2 else "3"; os.environ["SRBA_HIP_ROUNDS_FIRST"] = "8"
ctx = runner.HipContext(b.params); ctx.upload(b); print("uploaded", flush=True)
c = ctx.lm_run(); print("rounds done, trials", c["num_trials"].sum(), flush=True)
for k in ("num_trials", "chi2_final", "trace_chi2"):
    print(k, "identical" if np.array_equal(a[k], c[k], equal_nan=True) else "DIFFERENT", flush=True)
