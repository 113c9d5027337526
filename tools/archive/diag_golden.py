import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srba_amd import capi, runner
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _oracle  # tests/_oracle.py: the CPU checker (test infrastructure)
name = sys.argv[1]
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
b = runner.CapsuleBatch.load(os.path.join(G, name + ".caps")); g = np.load(os.path.join(G, name + ".npz"))
r = runner.run_batch_hip(b, download=True)
np.set_printoptions(precision=10, linewidth=200)
print("status", r["status"], "invalid", r["num_invalid_jacobs"])
print("chi2_init gpu", r["chi2_init"], "gold", g["chi2_init"])
print("lambda_init gpu", r["lambda_init"], "gold", g["lambda_init"])
print("chi2_final gpu", r["chi2_final"], "gold", g["chi2_final"])
print("trials gpu", r["num_trials"], "gold", g["num_trials"])
for i in range(b.n):
    m = int(min(r["num_trials"][i], g["num_trials"][i], 12))
    print(i, "gpu chi2", r["trace_chi2"][i][:m]); print(i, "gld chi2", g["trace_chi2"][i][:m]); print(i, "gpu rho", r["trace_rho"][i][:m]); print(i, "gld rho", g["trace_rho"][i][:m])
P, L, O, PD = capi.DIMS[b.family]
edges = np.concatenate([r["state"].array(i, "edge_pose", np.float64, b[i].n_unk_edges * PD) for i in range(b.n)]); print("edges max abs diff", np.abs(edges - g["edges"]).max())
lms = np.concatenate([r["state"].array(i, "ulm_pos", np.float64, b[i].n_unk_lms * L) for i in range(b.n)] + [np.zeros(0)])
if len(g["lms"]): dd = np.abs(lms - g["lms"]); print("lms max abs diff", dd.max(), "at", dd.argmax(), "value", g["lms"][dd.argmax()], "p99", np.percentile(dd, 99))
# stage comparison
ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib
lib.srba_hip_update_spantree(ctx.ctx, 0); chi2 = np.zeros(b.n); lib.srba_hip_eval_residuals(ctx.ctx, chi2.ctypes.data_as(capi.PF64)); lib.srba_hip_linearize(ctx.ctx)
res, Jp, Jf, HAp, Hf, HApf, grad = [ctx.debug(k) for k in range(7)]
P, L, O, PD = capi.DIMS[b.family]; o = [0] * 7
for i in range(b.n):
    c = b[i]; ref = _oracle.stage(b, i); n = P * c.n_unk_edges + L * c.n_unk_lms
    for k, (arr, key, cnt) in enumerate(((res, "resid", c.n_obs * O), (Jp, "Jp", c.n_bp * O * P), (Jf, "Jf", c.n_bf * O * L), (HAp, "HAp", c.n_hap * P * P), (Hf, "Hf", c.n_hf * L * L), (HApf, "HApf", c.n_hapf * P * L), (grad, "grad", n))):
        a = arr[o[k]:o[k] + cnt]; d = np.abs(a - ref[key]); sc = max(1e-300, np.abs(ref[key]).max())
        print("capsule %d %-5s max abs diff %.3e (scale %.3e) at %d" % (i, key, d.max() if cnt else 0, sc, int(d.argmax()) if cnt else -1))
        o[k] += cnt
