"""Two capsules per wavefront (k_lm_pair, SRBA_HIP_PAIR=1) against one (SRBA_HIP_PAIR=0) on the benchmark batch: results and timing. usage: diag_pair.py [n_kf]"""
import ctypes as C, glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
cache = sorted(glob.glob("/tmp/srba_bench_cache/caps_se2_tour_%d_seed1_*.bin" % n_kf))
b = runner.CapsuleBatch.load(cache[-1]) if cache else runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=n_kf, seed=1, path="tour"), backend="hip", submap=10, depth=3)
out = {}
for mode in ("0", "1"):
    os.environ["SRBA_HIP_PAIR"] = mode
    ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib; hist = (C.c_double * 4)()
    r = ctx.lm_run()
    def one():
        lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx); lib.srba_hip_kernel_ms_history(ctx.ctx, hist, 1); return hist[0]
    one(); v = np.array([one() for _ in range(12)]); lib.srba_hip_reset_state(ctx.ctx); r2 = ctx.lm_run()
    assert np.array_equal(r["num_trials"], r2["num_trials"]) and np.array_equal(r["chi2_final"], r2["chi2_final"]), "not reproducible run to run"
    out[mode] = r; print("pair=%s: kernel ms mean %.2f min %.2f max %.2f ; trials %d ; status!=0: %d" % (mode, v.mean(), v.min(), v.max(), r["num_trials"].sum(), (r["status"] != 0).sum()), flush=True)
    ctx.close()
a, c = out["0"], out["1"]
rel = np.abs(a["chi2_final"] - c["chi2_final"]) / np.maximum(np.abs(a["chi2_final"]), 1e-300); rel[np.abs(a["chi2_final"] - c["chi2_final"]) < 1e-20] = 0
ri = np.abs(a["chi2_init"] - c["chi2_init"]) / np.maximum(np.abs(a["chi2_init"]), 1e-300)
print("chi2_init rel diff max %.3e ; chi2_final rel diff max %.3e (> 1e-6: %d, > 1e-9: %d) ; same trial count: %d of %d ; lambda_init rel %.3e" % (ri.max(), rel.max(), (rel > 1e-6).sum(), (rel > 1e-9).sum(), (a["num_trials"] == c["num_trials"]).sum(), len(rel),
      (np.abs(a["lambda_init"] - c["lambda_init"]) / a["lambda_init"]).max()))
