"""The lambda-ladder speculation of a batch of ONE capsule (k_lm_spec) against the plain two-wavefront run of the same capsule: every result field, trace entry and downloaded
number must be bit-identical; prints the time per capsule of both. usage: diag_spec.py [n_kf] [W ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from srba_amd import capi, datasets, runner
import _oracle

def run_all(b, spec):
    os.environ["SRBA_HIP_SPEC"] = str(spec)
    ctx = runner.HipContext(b.params); P, L, O, PD = capi.DIMS[b.family]; out = []; t_run = 0.0; ms = 0.0; ph = np.zeros(16)
    for rep in range(2):  # second pass timed (first pass sizes the arenas)
        out = []; t_run = 0.0; ms = 0.0; ph[:] = 0
        for i in range(b.n):
            s = b.sub(i, 1); ctx.upload(s)
            t0 = time.perf_counter(); r = ctx.lm_run(); t_run += time.perf_counter() - t0; ms += ctx.lib.srba_hip_last_kernel_ms(ctx.ctx)
            if PHASES: ph += ctx.debug(10)[:16]
            w = s.clone(); ctx._chk(ctx.lib.srba_hip_download_state(ctx.ctx, w.ptr, 1), "download_state")
            r["edge"] = w.array(0, "edge_pose", np.float64, s[0].n_unk_edges * PD); r["pose"] = w.array(0, "pose", np.float64, 2 * s[0].n_pairs * PD); out.append(r)
    ctx.close()
    if PHASES:  # 100 MHz ticks per stage of replica 0 (SRBA_HIP_PHASE_TIMING=1: barriers around every stage, the totals are longer than the untimed run)
        names = {0: "spantree(all)", 1: "jacobians", 2: "hessian", 3: "residuals", 4: "gradient", 5: "solve", 6: "apply/adopt", 7: "spantree(trial)", 10: "  assemble", 11: "  factor+bsub", 14: "exchange"}
        print("   us per capsule: " + "  ".join("%s %.1f" % (names[k], ph[k] / b.n / 100.0) for k in sorted(names) if ph[k] > 0))
    return out, 1e3 * t_run / b.n, ms / b.n

PHASES = os.environ.get("SRBA_HIP_PHASE_TIMING") == "1"
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 600
Ws = [int(x) for x in sys.argv[2:]] or [8]
b = runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=n_kf, seed=1, path="tour"), backend=_oracle.BACKEND, submap=10, depth=3)
ref, t_ref, k_ref = run_all(b, 0)
trials = np.array([int(r["num_trials"][0]) for r in ref])
print("capsules %d, trials per capsule %.1f | plain: %.3f ms per lm_run call (kernel %.3f ms)" % (b.n, trials.mean(), t_ref, k_ref))
for W in Ws:
    got, t_w, k_w = run_all(b, W); bad = 0
    for i in range(b.n):
        for k in ref[i]:
            if k == "kernel_ms": continue
            if not np.array_equal(np.asarray(ref[i][k]), np.asarray(got[i][k]), equal_nan=True):
                bad += 1
                if bad <= 10: print("   capsule %d field %s differs: %s | %s" % (i, k, np.asarray(ref[i][k]).ravel()[:6], np.asarray(got[i][k]).ravel()[:6]))
    print("W = %2d: %.3f ms per lm_run call (kernel %.3f ms) | fields that differ: %d" % (W, t_w, k_w, bad))
