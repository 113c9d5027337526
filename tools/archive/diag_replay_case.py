"""One window of the replay soak trial by trial: GPU trace against the oracle following the GPU's decisions. usage: diag_replay_case.py kind solver seed window"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from srba_amd import capi, datasets, runner
import _oracle
kind, solver, seed, win = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if kind in ("rb2d", "cart2d"):
    ds, _ = datasets.landmarks_dataset_se2(kind, n_kf=30, n_lm=800, seed=seed, noise=1e-3); eng = runner.landmark_engine(kind, backend=_oracle.BACKEND, solver=solver, depth=2 + seed % 3)
else:
    ds, _ = datasets.landmarks_dataset_se3(kind, n_kf=14, n_lm=320, seed=seed, noise=(0.1 if kind in ("stereo", "mono") else 1e-3), init_from_gt_noise=(0.2 if kind == "mono" else None))
    eng = runner.landmark_engine(kind, backend=_oracle.BACKEND, solver=solver, robust=seed % 2)
eng.run(ds); b = eng.harvest(); b.engine = eng
sub = b.sub(max(0, b.n - 40), min(40, b.n)); one = sub.sub(win, 1)
gpu = runner.run_batch_hip(one); ref = _oracle.run_batch(one); rep = _oracle.run_batch_replay(one, gpu)
c = one[0]; print("window: %d unknown edges, %d unknown landmarks, %d obs; invalid Jacobian rows gpu %d / oracle %d / replay %d; not-PD gpu %d oracle %d; status %d %d" % (c.n_unk_edges, c.n_unk_lms, c.n_obs, gpu["num_invalid_jacobs"][0], ref["num_invalid_jacobs"][0], rep["num_invalid_jacobs"][0], gpu["num_not_pd"][0], ref["num_not_pd"][0], gpu["status"][0], ref["status"][0]))
print("chi2_init gpu %.10e oracle %.10e ; lambda_init %.6e %.6e" % (gpu["chi2_init"][0], ref["chi2_init"][0], gpu["lambda_init"][0], ref["lambda_init"][0]))
k = int(rep["replayed"][0])
for t in range(k):
    print("trial %2d dec %d  lambda %.3e | chi2 gpu %.10e  oracle(replay) %.10e  rel %.2e | rho gpu %+.3e oracle %+.3e | flags %d" % (t, rep["decisions"][0][t], gpu["trace_lambda"][0][t], gpu["trace_chi2"][0][t], rep["own_chi2"][0][t],
          abs(gpu["trace_chi2"][0][t] - rep["own_chi2"][0][t]) / abs(rep["own_chi2"][0][t]) if rep["own_chi2"][0][t] == rep["own_chi2"][0][t] else float("nan"), gpu["trace_rho"][0][t], rep["own_rho"][0][t], rep["flags"][0][t]))
