"""Randomised parity soak (not part of the test suite): families x solvers x seeds, GPU vs oracle on harvested capsules; prints the worst relative chi2 difference."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srba_amd import capi, datasets, runner
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _oracle  # tests/_oracle.py: the CPU checker (test infrastructure)
worst = 0.0; n_caps = 0; fails = []
for seed in range(11, 11 + int(sys.argv[1]) if len(sys.argv) > 1 else 14):
    cases = [("graph", capi.SOLVER_NO_SCHUR_SPARSE)] + [(k, s) for k in ("rb2d", "cart2d", "cart3d", "rb3d", "stereo", "mono") for s in (capi.SOLVER_SCHUR_DENSE, capi.SOLVER_NO_SCHUR_SPARSE)]
    for kind, solver in cases:
        if kind == "graph":
            ds = datasets.graph_slam_se2(n_kf=150 + 10 * seed, seed=seed, grid=2, block=25.0 + seed)
            eng = runner.graph_slam_engine(backend=_oracle.BACKEND, submap=5 + seed % 7, depth=2 + seed % 3)
        elif kind in ("rb2d", "cart2d"):
            ds, _ = datasets.landmarks_dataset_se2(kind, n_kf=30, n_lm=800, seed=seed, noise=1e-3); eng = runner.landmark_engine(kind, backend=_oracle.BACKEND, solver=solver, depth=2 + seed % 3)
        else:
            ds, _ = datasets.landmarks_dataset_se3(kind, n_kf=14, n_lm=320, seed=seed, noise=(0.1 if kind in ("stereo", "mono") else 1e-3), init_from_gt_noise=(0.2 if kind == "mono" else None))
            eng = runner.landmark_engine(kind, backend=_oracle.BACKEND, solver=solver, robust=seed % 2)
        try:
            eng.run(ds)
        except RuntimeError as e:
            print("skip", kind, seed, str(e)[:80]); continue
        b = eng.harvest(); b.engine = eng
        sub = b.sub(max(0, b.n - 40), min(40, b.n))
        ref = _oracle.run_batch(sub); gpu = runner.run_batch_hip(sub)
        # a diverged map (edges of 1e12 m: the reference LM can do that on a bad window) only amplifies rounding: keep well-scaled capsules
        P_, L_, O_, PD_ = capi.DIMS[sub.family]
        sane = np.array([np.abs(sub.array(i, "edge_pose", np.float64, sub.ptr[i].n_edges * PD_).reshape(-1, PD_)[:, :2 if PD_ == 3 else 3]).max() < 1e4 for i in range(sub.n)])
        rel = np.where(sane, np.abs(gpu["chi2_final"] - ref["chi2_final"]) / np.maximum(np.abs(ref["chi2_final"]), 1e-18), 0.0)
        ok = (gpu["status"] == ref["status"]).all() and np.isfinite(gpu["chi2_final"]).all()
        worst = max(worst, float(rel.max())); n_caps += int(sane.sum())
        if not ok or rel.max() > 1e-6: fails.append((kind, solver, seed, float(rel.max())))
print("capsules %d, worst relative chi2_final difference %.3e, failures: %s" % (n_caps, worst, fails))
