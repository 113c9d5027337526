"""Runs the fused LM kernel on a harvested batch and dumps every result (counters, traces, final unknowns, spanning-tree poses) to an .npz: two library builds are compared
bit for bit with `dump_results.py a.npz ...; (swap libsrba_hip.so); dump_results.py b.npz ...; dump_results.py --cmp a.npz b.npz`. usage: dump_results.py out.npz [se2|stereo|mono|rb2d] [n_kf]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3]); bad = 0
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True); bad += not same
        if not same:
            d = np.abs(a[k].astype(float) - b[k].astype(float)); print("%-16s DIFFERENT: max abs %.3e, %d of %d entries" % (k, np.nanmax(d), int((d > 0).sum()), d.size))
    print("compared %d arrays: %s" % (len(a.files), "all identical" if not bad else "%d differ" % bad)); sys.exit(1 if bad else 0)
from srba_amd import capi, datasets, runner
import _oracle
fam = sys.argv[2] if len(sys.argv) > 2 else "se2"; n_kf = int(sys.argv[3]) if len(sys.argv) > 3 else (1500 if fam == "se2" else 40)
if fam == "se2":
    ds = datasets.graph_slam_se2(n_kf=n_kf, seed=3, path="tour"); b = runner.harvest_graph_slam(ds, backend=_oracle.BACKEND, submap=10, depth=3)
else:
    if fam == "rb2d": ds, _ = datasets.landmarks_dataset_se2(fam, n_kf=n_kf, n_lm=30 * n_kf, seed=7, noise=1e-3)
    else: ds, _ = datasets.landmarks_dataset_se3(fam, n_kf=n_kf, n_lm=10 * n_kf, seed=5, noise=0.1, init_from_gt_noise=(0.05 if fam == "mono" else None), known_first=(1000 if fam == "mono" else 0))
    eng = runner.landmark_engine(fam, backend=_oracle.BACKEND); eng.run(ds); b = eng.harvest(); b.engine = eng
r = runner.run_batch_hip(b, download=True)
P, L, O, PD = capi.DIMS[b.family]
out = {k: r[k] for k in ("status", "num_iters", "num_trials", "num_not_pd", "num_accepted", "num_relinearized", "stop_reason", "chi2_init", "chi2_final", "obs_rmse", "lambda_final", "trace_chi2",
        "trace_lambda", "trace_rho")}
out["edges"] = np.concatenate([r["state"].array(i, "edge_pose", np.float64, b[i].n_unk_edges * PD) for i in range(b.n)])
out["poses"] = np.concatenate([r["state"].array(i, "pose", np.float64, 2 * b[i].n_pairs * PD) for i in range(b.n)])
out["lms"] = np.concatenate([r["state"].array(i, "ulm_pos", np.float64, b[i].n_unk_lms * L) for i in range(b.n)] + [np.zeros(0)])
np.savez(sys.argv[1], **out); print(fam, "capsules", b.n, "trials", int(r["num_trials"].sum()), "kernel ms %.2f" % r["kernel_ms"])
