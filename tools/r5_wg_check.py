"""Round 5: the workgroup kernel of the SE3 landmark families (k_lm_wg, srba_wg.hpp) against the one-wavefront kernel and the oracle on the batch of tools/diag_family.py.
usage: r5_wg_check.py [stereo|mono|cart3d|rb3d] [copies] -- prints, per setting of SRBA_HIP_WG / _WG_FROM / _WG256_FROM given as further arguments "WG=0" "WG256_FROM=48" ...:
kernel ms of a run from the pristine state, LM iterations/s, max relative chi2 difference to the oracle over the converged windows, and (PHASES=1) per-phase us per trial by window size."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _oracle  # the CPU checker (test infrastructure)
kind = sys.argv[1] if len(sys.argv) > 1 else "stereo"; copies = int(sys.argv[2]) if len(sys.argv) > 2 else 64
settings = sys.argv[3:] or ["WG=0", "WG=1"]
ds, _ = datasets.landmarks_dataset_se3(kind, n_kf=60, n_lm=600, seed=5, noise=(0.1 if kind in ("stereo", "mono") else 1e-3), init_from_gt_noise=(0.2 if kind == "mono" else None))
os.environ["SRBA_HIP_WG"] = "0"; eng = runner.landmark_engine(kind, backend="hip"); eng.run(ds); b = eng.harvest(); n0 = b.n; del os.environ["SRBA_HIP_WG"]
ro = _oracle.run_batch(b, threads=8)
arr = (capi.Capsule * (n0 * copies))()
for r in range(copies):
    for i in range(n0): arr[r * n0 + i] = b.ptr[i]
class Fake: pass
fb = Fake(); fb.ptr = C.cast(arr, capi.PCAP); fb.n = n0 * copies; fb.params = b.params; fb.family = b.family
nk = np.array([b.ptr[i % n0].n_unk_edges for i in range(fb.n)])
names = ["K1all", "jac", "hess", "resid", "grad", "solve", "apply", "K1need", "restore", "schur", "assemble", "factor", "bsub", "feat", "sch_inv", "sch_terms"]
for st in settings:
    keys = []
    for kv in st.split(","):
        k, v = kv.split("="); os.environ["SRBA_HIP_" + k if k != "PHASES" else "SRBA_HIP_PHASE_TIMING"] = v; keys.append("SRBA_HIP_" + k if k != "PHASES" else "SRBA_HIP_PHASE_TIMING")
    ctx = runner.HipContext(b.params); ctx.upload(fb); r = ctx.lm_run(); lib = ctx.lib
    ms = []
    for _ in range(3):
        lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx); h = (C.c_double * 1)(); lib.srba_hip_kernel_ms_history(ctx.ctx, h, 1); ms.append(h[0])
    g = r["chi2_final"][:n0]; sane = ro["obs_rmse"] < 3.0
    rel = np.abs(g - ro["chi2_final"]) / np.maximum(ro["chi2_final"], 1e-300)
    same = all(np.array_equal(r["chi2_final"][q * n0:(q + 1) * n0], g, equal_nan=True) for q in range(copies))
    print("[%s %s] kernel %.2f ms (%s) -> %.3f M LM it/s ; trials %d (oracle %d x %d) ; max rel chi2 diff vs oracle %.2e on %d converged windows (all %d: %.2e) ; status!=0: %d ; copies identical: %s" % (
        kind, st, min(ms), " ".join("%.1f" % x for x in ms), r["num_trials"].sum() / min(ms) / 1e3, r["num_trials"].sum(), ro["num_trials"].sum(), copies,
                rel[sane].max() if sane.any() else float("nan"), sane.sum(), n0, rel.max(), int((r["status"] != 0).sum()), same), flush=True)
    if os.environ.get("SRBA_HIP_PHASE_TIMING") == "1":
        pc = ctx.debug(10).reshape(fb.n, 16); tr = r["num_trials"]
        for lo, hi in ((0, 20), (20, 32), (32, 40), (40, 70)):
            m = (nk >= lo) & (nk < hi)
            if not m.any(): continue
            t = pc[m].sum(axis=0) * 1e-2 / max(tr[m].sum(), 1)
            print("   edges [%d,%d): %d caps, %.1f trials/cap, per trial us: total %.0f | " % (lo, hi, m.sum(), tr[m].mean(), t[:9].sum()) + " ".join("%s %.0f" % (names[k], t[k]) for k in (9, 14, 15,
                    10, 11, 12, 13, 7, 3, 2, 4, 6, 1, 0)), flush=True)
    ctx.close()
    for k in keys: os.environ.pop(k, None)
