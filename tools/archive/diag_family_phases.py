"""Per-phase time of the fused LM kernel on a landmark-family batch (as tools/diag_family.py), by size of the reduced system. usage: diag_family_phases.py [kind] [copies]"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SRBA_HIP_PHASE_TIMING"] = "1"
from srba_amd import capi, datasets, runner
kind = sys.argv[1] if len(sys.argv) > 1 else "stereo"; copies = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if kind in ("rb2d", "cart2d"): ds, _ = datasets.landmarks_dataset_se2(kind, n_kf=60, n_lm=900, seed=4, noise=1e-3)
else: ds, _ = datasets.landmarks_dataset_se3(kind, n_kf=60, n_lm=600, seed=5, noise=(0.1 if kind in ("stereo", "mono") else 1e-3), init_from_gt_noise=(0.2 if kind == "mono" else None))
eng = runner.landmark_engine(kind, backend="hip"); eng.run(ds); b = eng.harvest(); n0 = b.n
arr = (capi.Capsule * (n0 * copies))()
for r in range(copies):
    for i in range(n0): arr[r * n0 + i] = b.ptr[i]
class Fake: pass
fb = Fake(); fb.ptr = C.cast(arr, capi.PCAP); fb.n = n0 * copies; fb.params = b.params; fb.family = b.family
ctx = runner.HipContext(b.params); ctx.upload(fb); ctx.lm_run(); r = ctx.lm_run(); kms = ctx.lib.srba_hip_last_kernel_ms(ctx.ctx)
pc = ctx.debug(10).reshape(fb.n, 16); shp = ctx.debug(11).reshape(fb.n, 4)
nk = np.array([b.ptr[i % n0].n_unk_edges for i in range(fb.n)]); tr = r["num_trials"]
names = ["K1all", "jac", "hess", "resid", "grad", "solve", "apply", "K1need", "restore", "schur", "assemble", "factor", "bsub", "feat", "sch_inv", "sch_terms"]
print("%s: kernel %.1f ms, %d capsules" % (kind, kms, fb.n))
for lo, hi in ((0, 20), (20, 32), (32, 40), (40, 70)):
    m = (nk >= lo) & (nk < hi)
    if not m.any(): continue
    t = pc[m].sum(axis=0) * 1e-2 / max(tr[m].sum(), 1)   # us per trial
    print("edges [%d,%d): %d caps, LDS %.0f KB, %.1f trials/cap, per trial us: total %.0f | " % (lo, hi, m.sum(), shp[m, 0].mean() / 1024, tr[m].mean(), t[:9].sum()) + " ".join("%s %.0f" % (names[k], t[k]) for k in (9, 14, 15, 10, 11, 12, 13, 7, 3, 2, 4, 6, 1, 0)) + " | per capsule ms %.1f" % (pc[m][:, :9].sum(axis=1).mean() * 1e-5))
