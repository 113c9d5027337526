"""Randomised parity soak (not part of the test suite): families x solvers x seeds, GPU vs oracle on harvested capsules.
Round 4: every window is compared over its WHOLE trial sequence by decision replay (tests/_oracle.py run_batch_replay: the oracle takes the GPU's accept / reject / not-PD
decisions and reports its own rho and chi2 per trial): (i) chi2 of every accepted trial, (ii) every decision the oracle would have taken differently is a step that moves chi2 by
less than 1e-9 of its value in both runs, (iii) final chi2. usage: soak_parity.py [n_seeds] [first_seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srba_amd import capi, datasets, runner
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _oracle  # tests/_oracle.py: the CPU checker (test infrastructure)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3; first = int(sys.argv[2]) if len(sys.argv) > 2 else 11
tot = dict(caps=0, complete=0, trace_bad=0, floor_bad=0, final_bad=0, disagree=0, diverged=0, forced=0, own_final_bad=0); worst = dict(trace=0.0, floor=0.0, final=0.0, own_final=0.0); fails = []
for seed in range(first, first + n_seeds):
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
        rep = _oracle.run_batch_replay(sub, gpu, threads=8); R = _oracle.replay_report(gpu, rep)
        # a diverged map (edges of 1e12 m: the reference LM can do that on a bad window) only amplifies rounding: keep well-scaled capsules
        P_, L_, O_, PD_ = capi.DIMS[sub.family]
        sane = np.array([np.abs(sub.array(i, "edge_pose", np.float64, sub.ptr[i].n_edges * PD_).reshape(-1, PD_)[:, :2 if PD_ == 3 else 3]).max() < 1e4 for i in range(sub.n)])
        own_rel = np.where(sane, np.abs(gpu["chi2_final"] - ref["chi2_final"]) / np.maximum(np.abs(ref["chi2_final"]), 1e-18), 0.0)
        ok = (gpu["status"] == ref["status"]).all() and np.isfinite(gpu["chi2_final"]).all()
        cm = sane & R["complete"]
        tot["caps"] += int(sane.sum()); tot["complete"] += int(cm.sum()); tot["trace_bad"] += int((sane & ~R["trace_ok"]).sum()); tot["floor_bad"] += int((sane & ~R["floor_ok"]).sum())
        tot["final_bad"] += int((cm & (R["final_rel"] > 1e-6)).sum()); tot["disagree"] += int(R["n_disagree"][sane].sum()); tot["diverged"] += int((sane & (R["diverged_at"] >= 0)).sum()); tot["forced"] += int(R["forced_notpd"][sane].sum())
        tot["own_final_bad"] += int((own_rel > 1e-6).sum())
        worst["trace"] = max(worst["trace"], float(np.where(sane, R["worst_trace"], 0).max())); worst["floor"] = max(worst["floor"], float(np.where(sane, R["worst_floor"], 0).max()))
        worst["final"] = max(worst["final"], float(np.where(cm, R["final_rel"], 0).max())); worst["own_final"] = max(worst["own_final"], float(own_rel.max()))
        bad = sane & (~R["trace_ok"] | ~R["floor_ok"] | (R["complete"] & (R["final_rel"] > 1e-6)))
        if bad.any():   # beyond 1e-9: is it the window's own rounding amplification (observations moved by one ulp, same decisions)?
            sens, _st = _oracle.rounding_sensitivity(sub, gpu); tolw = np.minimum(1e-4, np.maximum(1e-9,
                    50.0 * sens))   # capped: a window whose own arithmetic moves by more than 2e-6 per ulp is "not comparable", never a pass
            chaotic = bad & (sens > 2e-6); tot["not_comparable"] = tot.get("not_comparable", 0) + int(chaotic.sum())
            for i in np.flatnonzero(bad): print("   %s solver %d seed %d window %d: trace %.2e floor %.2e final %.2e ; rounding sensitivity %.2e -> %s" % (kind, solver, seed, i, R["worst_trace"][i],
                    R["worst_floor"][i], R["final_rel"][i], sens[i], "within 50x" if max(R["worst_trace"][i], R["worst_floor"][i]) <= tolw[i] else "BEYOND"))
            bad = bad & ((R["worst_trace"] > tolw) | (R["worst_floor"] > tolw) | (R["complete"] & (R["final_rel"] > np.maximum(tolw, 1e-6))))
        if not ok: fails.append((kind, solver, seed, -1, "status / finite mismatch between the GPU and the oracle"))
        if bad.any():
            for i in np.flatnonzero(bad): fails.append((kind, solver, seed, int(i), "trace %.2e floor %.2e final %.2e complete %d trials %d cond-proxy rmse %.3g" % (R["worst_trace"][i],
                    R["worst_floor"][i], R["final_rel"][i], R["complete"][i], gpu["num_trials"][i], gpu["obs_rmse"][i])))
print("totals", tot); print("worst", worst)
print("failures (%d):" % len(fails))
for f in fails: print("  ", f)
