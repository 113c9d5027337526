"""Decision replay on the datasets of the two formerly statistical parity tests (flat valleys: range-bearing 2D seed 32; the lost monocular map) and on the SE2 batch. usage: diag_replay.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from srba_amd import capi, datasets, runner
import _oracle
def show(name, b):
    gpu = runner.run_batch_hip(b); ref = _oracle.run_batch(b); rep = _oracle.run_batch_replay(b, gpu, threads=8); R = _oracle.replay_report(gpu, rep)
    own = np.abs(gpu["chi2_final"] - ref["chi2_final"]) / np.maximum(np.abs(ref["chi2_final"]), 1e-300)
    print("%-12s windows %d complete %d | own-run final diff max %.2e (>1e-6: %d) | replay: trace max %.2e (bad %d) floor max %.2e (bad %d) final max %.2e (>1e-6: %d, >1e-9: %d) disagreements %d diverged %d forced-notpd %d" % (
        name, b.n, R["complete"].sum(), own.max(), (own > 1e-6).sum(), R["worst_trace"].max(), (~R["trace_ok"]).sum(), R["worst_floor"].max(), (~R["floor_ok"]).sum(),
        np.where(R["complete"], R["final_rel"],
                0).max(), (R["complete"] & (R["final_rel"] > 1e-6)).sum(), (R["complete"] & (R["final_rel"] > 1e-9)).sum(), R["n_disagree"].sum(), (R["diverged_at"] >= 0).sum(), R["forced_notpd"].sum()))
    if name == "mono-lost":
        sens, sens_trial = _oracle.rounding_sensitivity(b, gpu)
        print("   per window: GPU-vs-oracle worst accepted-trial distance / rounding sensitivity (1 ulp on the observations, 3 seeds):")
        print("   dist ", np.array2string(R["worst_trace"], precision=1, max_line_width=250)); print("   sens ", np.array2string(sens, precision=1, max_line_width=250)); print("   floor",
                np.array2string(R["worst_floor"], precision=1, max_line_width=250))
        print("   ratio max (trace, final)", (np.maximum(R["worst_trace"], R["final_rel"]) / np.maximum(sens, 1e-10)).max(), " disputed decisions: move / that trial's sensitivity, max",
                (R["floor_move"] / np.maximum(sens_trial, 1e-10)).max(), "largest disputed move", R["floor_move"].max())
    for i in np.flatnonzero(~R["trace_ok"] | ~R["floor_ok"] | (R["complete"] & (R["final_rel"] > 1e-6)))[:12]:
        k = int(rep["replayed"][i]); acc = rep["decisions"][i][:k] == 2
        with np.errstate(all="ignore"): rel = np.abs(gpu["trace_chi2"][i][:k] - rep["own_chi2"][i][:k]) / np.abs(rep["own_chi2"][i][:k])
        print("   window %d: trials %d, rmse %.3g, chi2 %.4g, trace rel on accepted %s ; disagreements at %s" % (i, gpu["num_trials"][i], gpu["obs_rmse"][i], gpu["chi2_final"][i],
                np.array2string(rel[acc], precision=1), np.flatnonzero(rep["flags"][i][:k] & 2)))
ds, _ = datasets.landmarks_dataset_se2("rb2d", n_kf=30, n_lm=800, seed=32, noise=1e-3)
eng = runner.landmark_engine("rb2d", backend=_oracle.BACKEND, solver=capi.SOLVER_SCHUR_DENSE, depth=2 + 32 % 3); eng.run(ds); b = eng.harvest(); b.engine = eng
show("flat-valley", b.sub(max(0, b.n - 40), min(40, b.n)))
for gf in (True, False):
    ds = datasets.landmarks_dataset_se3("mono", n_kf=60, n_lm=600, seed=5, noise=0.1, init_from_gt_noise=0.05 if gf else 0.2, known_first=1000 if gf else 0)[0]
    eng = runner.landmark_engine("mono", backend=_oracle.BACKEND); eng.run(ds); b = eng.harvest(); b.engine = eng
    show("mono-fixed" if gf else "mono-lost", b)
b = runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=240, seed=5, grid=2, block=30.0), backend=_oracle.BACKEND, submap=10, depth=3)
show("se2-240", b)
b = runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=2000, seed=1, path="tour"), backend=_oracle.BACKEND, submap=10, depth=3)
show("se2-tour-2k", b)
