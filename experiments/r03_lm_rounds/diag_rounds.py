"""The LM loop as rounds over the batch (srba_rounds.hpp) against the fused kernel on the same capsules: results must be bit-identical. usage: diag_rounds.py [n_kf] [family: se2|stereo|rb2d|mono]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from srba_amd import capi, datasets, runner
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
fam = sys.argv[2] if len(sys.argv) > 2 else "se2"
if fam == "se2":
    ds = datasets.graph_slam_se2(n_kf=n_kf, seed=3, path="tour"); b = runner.harvest_graph_slam(ds, backend="hip", submap=10, depth=3)
else:
    if fam == "rb2d": ds, _ = datasets.landmarks_dataset_se2(fam, n_kf=n_kf, n_lm=30 * n_kf, seed=7, noise=1e-3)
    else: ds, _ = datasets.landmarks_dataset_se3(fam, n_kf=n_kf, n_lm=10 * n_kf, seed=5, noise=0.1, init_from_gt_noise=(0.05 if fam == "mono" else None), known_first=(1000 if fam == "mono" else 0))
    eng = runner.landmark_engine(fam, backend="hip"); eng.run(ds); b = eng.harvest(); b.engine = eng
print("capsules", b.n)
out = {}
for mode in ("0", "1"):
    os.environ["SRBA_HIP_ROUNDS"] = mode
    ctx = runner.HipContext(b.params); ctx.upload(b)
    t = time.perf_counter(); r = ctx.lm_run(); t1 = time.perf_counter() - t
    ts = []
    for _ in range(5):
        ctx.lib.srba_hip_reset_state(ctx.ctx); t = time.perf_counter(); ctx.lib.srba_hip_lm_run_async(ctx.ctx); ctx.lib.srba_hip_sync(ctx.ctx); ts.append(time.perf_counter() - t)
    work = b.clone(); ctx._chk(ctx.lib.srba_hip_download_state(ctx.ctx, work.ptr, work.n), "download"); r["state"] = work
    out[mode] = r; print("rounds=%s first run %.1f ms, then %s ms; trials %d" % (mode, 1e3 * t1, ["%.1f" % (1e3 * x) for x in ts], r["num_trials"].sum()))
    ctx.close()
a, c = out["0"], out["1"]
def same(x, y): return np.array_equal(x, y, equal_nan=True)
for k in ("status", "num_iters", "num_trials", "num_not_pd", "num_accepted", "num_relinearized", "num_invalid_jacobs", "stop_reason", "chi2_init", "chi2_final", "obs_rmse", "lambda_init", "lambda_final", "trace_chi2", "trace_lambda", "trace_rho"):
    ok = same(a[k], c[k]); print("%-20s %s" % (k, "identical" if ok else "DIFFERENT"))
    if not ok:
        bad = np.flatnonzero([not same(a[k][i], c[k][i]) for i in range(b.n)]); print("   first differing capsules", bad[:8], "of", len(bad))
        i = bad[0]; print("   fused ", a[k][i] if np.ndim(a[k][i]) == 0 else a[k][i][:12]); print("   rounds", c[k][i] if np.ndim(c[k][i]) == 0 else c[k][i][:12])
P, L, O, PD = capi.DIMS[b.family]; bad_e = bad_p = bad_l = 0
for i in range(b.n):
    nk = b[i].n_unk_edges
    bad_e += not same(a["state"].array(i, "edge_pose", np.float64, nk * PD), c["state"].array(i, "edge_pose", np.float64, nk * PD))
    bad_p += not same(a["state"].array(i, "pose", np.float64, 2 * b[i].n_pairs * PD), c["state"].array(i, "pose", np.float64, 2 * b[i].n_pairs * PD))
    bad_l += not same(a["state"].array(i, "ulm_pos", np.float64, b[i].n_unk_lms * L), c["state"].array(i, "ulm_pos", np.float64, b[i].n_unk_lms * L))
print("final state: capsules with different edges %d, spanning-tree poses %d, landmarks %d (of %d)" % (bad_e, bad_p, bad_l, b.n))
