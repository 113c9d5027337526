"""The gauge-free monocular windows (tests/test_gpu_parity.py::test_ill_conditioned_mono_windows_match_wherever_the_reference_pins_them): per window, the prefix of the LM trace on which
the reference's two Schur solvers agree, and how the GPU compares with the oracle on it. usage: diag_mono_pinned.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from srba_amd import capi, datasets, runner
import _oracle
ds = datasets.landmarks_dataset_se3("mono", n_kf=60, n_lm=600, seed=5, noise=0.1, init_from_gt_noise=0.2)[0]
eng = runner.landmark_engine("mono", backend=_oracle.BACKEND); eng.run(ds); b = eng.harvest(); b.engine = eng
r0 = _oracle.run_batch(b); b.params.solver = capi.SOLVER_SCHUR_SPARSE; r1 = _oracle.run_batch(b); b.params.solver = capi.SOLVER_SCHUR_DENSE
gpu = runner.run_batch_hip(b)
print("chi2_init max rel", np.max(np.abs(gpu["chi2_init"] - r0["chi2_init"]) / r0["chi2_init"]), "lambda_init max rel", np.max(np.abs(gpu["lambda_init"] - r0["lambda_init"]) / r0["lambda_init"]))
for i in range(b.n):
    m = int(min(gpu["num_trials"][i], r0["num_trials"][i], r1["num_trials"][i], capi.TRACE_LEN))
    c0, c1, g = r0["trace_chi2"][i][:m], r1["trace_chi2"][i][:m], gpu["trace_chi2"][i][:m]
    spread = np.abs(c1 - c0) / np.maximum(np.abs(c0), 1e-300)
    ref_agree = (np.sign(r0["trace_rho"][i][:m]) == np.sign(r1["trace_rho"][i][:m])) & (np.isnan(c0) == np.isnan(c1)) & ~(spread > 1e-7)
    k = m if ref_agree.all() else int(np.argmin(ref_agree))
    dec = np.sign(gpu["trace_rho"][i][:k]) == np.sign(r0["trace_rho"][i][:k]); kd = k if dec.all() else int(np.argmin(dec))
    acc = r0["trace_rho"][i][:k] > 0; dev = np.abs(g[:k] - c0[:k]) / np.abs(c0[:k])
    ratio = dev[acc] / np.maximum(1e-6, 100 * spread[:k][acc]) if acc.any() else np.zeros(1)
    print("win %2d trials g/o0/o1 %2d %2d %2d pinned %2d same-decisions %2d | max dev on accepted %.2e (x tol %.2f) | spread at k-1 %.1e | final rel g-o0 %.1e o1-o0 %.1e" % (
        i, gpu["num_trials"][i], r0["num_trials"][i], r1["num_trials"][i], k, kd, dev[acc].max() if acc.any() else 0, ratio.max(), spread[k - 1] if k else 0,
        abs(gpu["chi2_final"][i] - r0["chi2_final"][i]) / r0["chi2_final"][i], abs(r1["chi2_final"][i] - r0["chi2_final"][i]) / r0["chi2_final"][i]))
