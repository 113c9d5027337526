"""GPU diagnostics (not a test): parity statistics + per-phase device timing of the fused LM kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SRBA_HIP_PHASE_TIMING"] = "1"
from srba_amd import capi, datasets, runner
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _oracle  # tests/_oracle.py: the CPU checker (test infrastructure)

n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 16
t = time.time(); ds = datasets.graph_slam_se2(n_kf=n_kf, seed=1, grid=grid); print("dataset %.1fs obs/kf %.2f" % (time.time() - t, np.mean([len(k["feat_ids"]) for k in ds])))
t = time.time(); b = runner.harvest_graph_slam(ds, backend=_oracle.BACKEND, submap=10, depth=3); print("harvest(oracle) %.1fs  capsules %d" % (time.time() - t, b.n))
for f in ("n_unk_edges", "n_obs", "n_bp", "n_pairs", "n_path", "n_hap", "n_edges"):
    v = np.array([getattr(b[i], f) for i in range(b.n)]); print("  %-12s mean %8.1f  p50 %6d  p95 %6d  max %6d" % (f, v.mean(), np.median(v), np.percentile(v, 95), v.max()))
need = np.array([np.ctypeslib.as_array(b[i].pair_needed, shape=(b[i].n_pairs,)).sum() for i in range(b.n)]); print("  pairs_needed mean %.1f" % need.mean())
ctx = runner.HipContext(b.params); ctx.upload(b)
t = time.time(); gpu = ctx.lm_run(); t_gpu = time.time() - t
kms = ctx.lib.srba_hip_last_kernel_ms(ctx.ctx)
t = time.time(); cpu = _oracle.run_batch(b); t_cpu = time.time() - t
print("GPU kernel %.2f ms, %d trials -> %.0f trials/s | oracle %.2f s -> %.0f trials/s" % (kms, gpu["num_trials"].sum(), gpu["num_trials"].sum() / (kms * 1e-3), t_cpu, cpu["num_trials"].sum() / t_cpu))
same = gpu["num_trials"] == cpu["num_trials"]
print("trial-count identical: %d / %d" % (same.sum(), b.n))
rel = np.abs(gpu["chi2_final"] - cpu["chi2_final"]) / np.maximum(1e-300, np.abs(cpu["chi2_final"]))
print("chi2_final rel diff: max %.3e  p99 %.3e  median %.3e ; #>1e-6: %d" % (rel.max(), np.percentile(rel, 99), np.median(rel), (rel > 1e-6).sum()))
rel0 = np.abs(gpu["chi2_init"] - cpu["chi2_init"]) / np.maximum(1e-300, np.abs(cpu["chi2_init"])); print("chi2_init rel diff max %.3e" % rel0.max())
# first diverging trial
first = []
for i in range(b.n):
    g, c = gpu["trace_chi2"][i], cpu["trace_chi2"][i]
    m = min(gpu["num_trials"][i], cpu["num_trials"][i], capi.TRACE_LEN)
    d = np.abs(g[:m] - c[:m]) / np.maximum(1e-300, np.abs(c[:m])); bad = np.where(~(d < 1e-6))[0]
    first.append(bad[0] if len(bad) else -1)
first = np.array(first); print("capsules whose per-trial chi2 traces agree to 1e-6 on the common prefix: %d / %d" % ((first < 0).sum(), b.n))
worst = np.argsort(-rel)[:3]
for i in worst:
    print("capsule %d nK %d: trials gpu %d cpu %d chi2_final gpu %.9e cpu %.9e first-div %d" % (i, b[i].n_unk_edges, gpu["num_trials"][i], cpu["num_trials"][i], gpu["chi2_final"][i], cpu["chi2_final"][i], first[i]))
    k = max(0, first[i] - 2)
    print("   gpu chi2", gpu["trace_chi2"][i][k:k + 6], "\n   cpu chi2", cpu["trace_chi2"][i][k:k + 6], "\n   gpu rho", gpu["trace_rho"][i][k:k + 6], "\n   cpu rho", cpu["trace_rho"][i][k:k + 6])
pc = ctx.debug(10).reshape(b.n, 16)
names = ["K1 spantree(all)", "K2/K3 jacobians", "K6 hessian", "K4 residuals", "K5 gradient", "solve(total)", "K11 apply", "K1 spantree(needed)", "K12 restore", " schur_reduce", " assemble", " chol_factor", " tri-solve", " schur_features"]
tot = pc.sum(axis=0); allc = tot[[0, 1, 2, 3, 4, 5, 6, 7, 8]].sum()
print("device phase time (sum over workgroups, 100 MHz ticks -> ms):")
for k, nm in enumerate(names):
    print("  %-22s %10.2f ms  %5.1f%%" % (nm, tot[k] * 1e-5, 100.0 * tot[k] / allc))
print("  per-trial mean: %.1f us ; solve per call %.1f us" % (allc * 1e-2 / gpu["num_trials"].sum(), tot[5] * 1e-2 / gpu["num_trials"].sum()))
nk = np.array([b[i].n_unk_edges for i in range(b.n)])
for lo, hi in ((0, 32), (33, 48), (49, 64), (65, 10000)):
    m = (nk >= lo) & (nk <= hi)
    if not m.any():
        continue
    tr = gpu["num_trials"][m].sum(); t = pc[m].sum(axis=0)
    print("class nK in [%d,%d]: %d capsules, %d trials; per trial: total %.1f us | factor %.1f | trisolve %.1f | assemble %.1f | spantree(needed) %.1f | resid %.1f | hess %.1f | grad %.1f | apply %.1f ; max WG time %.2f ms" % (
        lo, hi, m.sum(), tr, t[:9].sum() * 1e-2 / tr, t[11] * 1e-2 / tr, t[12] * 1e-2 / tr, t[10] * 1e-2 / tr, t[7] * 1e-2 / tr, t[3] * 1e-2 / tr, t[2] * 1e-2 / tr, t[4] * 1e-2 / tr, t[6] * 1e-2 / tr, pc[m][:, :9].sum(axis=1).max() * 1e-5))
