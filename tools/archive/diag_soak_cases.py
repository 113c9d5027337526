"""The windows of tools/soak_parity.py whose chi2_final differs by more than 1e-6 from the oracle's: are they rounding-floor decisions (tests/test_gpu_parity.py::_compare_lm)? usage: diag_soak_cases.py kind:solver:seed ..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from srba_amd import capi, datasets, runner
import _oracle
from test_gpu_parity import _compare_lm
for case in sys.argv[1:]:
    kind, solver, seed = case.split(":"); solver = int(solver); seed = int(seed)
    if kind in ("rb2d", "cart2d"):
        ds, _ = datasets.landmarks_dataset_se2(kind, n_kf=30, n_lm=800, seed=seed, noise=1e-3); eng = runner.landmark_engine(kind, backend=_oracle.BACKEND, solver=solver, depth=2 + seed % 3)
    else:
        ds, _ = datasets.landmarks_dataset_se3(kind, n_kf=14, n_lm=320, seed=seed, noise=(0.1 if kind in ("stereo", "mono") else 1e-3), init_from_gt_noise=(0.2 if kind == "mono" else None))
        eng = runner.landmark_engine(kind, backend=_oracle.BACKEND, solver=solver, robust=seed % 2)
    eng.run(ds); b = eng.harvest(); b.engine = eng
    sub = b.sub(max(0, b.n - 40), min(40, b.n)); ref = _oracle.run_batch(sub); gpu = runner.run_batch_hip(sub)
    rel = np.abs(gpu["chi2_final"] - ref["chi2_final"]) / np.maximum(np.abs(ref["chi2_final"]), 1e-18); bad = np.flatnonzero(rel > 1e-6)
    print(case, "windows over 1e-6:", bad.tolist(), "rel", rel[bad], "rmse cpu", ref["obs_rmse"][bad], "trials cpu / gpu", ref["num_trials"][bad], gpu["num_trials"][bad])
    for i in bad:
        m = int(min(gpu["num_trials"][i], ref["num_trials"][i], capi.TRACE_LEN)); g, c = gpu["trace_chi2"][i][:m], ref["trace_chi2"][i][:m]
        same = (np.sign(gpu["trace_rho"][i][:m]) == np.sign(ref["trace_rho"][i][:m])) & (np.isnan(g) == np.isnan(c)); k = m if same.all() else int(np.argmin(same))
        acc = np.flatnonzero(ref["trace_rho"][i][:k] > 0); e_prev = c[acc[-1]] if len(acc) else ref["chi2_init"][i]
        print("   window", i, ": decisions agree on the first", k, "of", m, "trials; accepted chi2 on that prefix differ by at most %.2e;" % (np.abs(g[:k][ref["trace_rho"][i][:k] > 0] - c[:k][ref["trace_rho"][i][:k] > 0]) / np.maximum(c[:k][ref["trace_rho"][i][:k] > 0], 1e-300)).max() if len(acc) else "   (no accepted trial)",
              "at the split chi2 changes by %.2e (cpu) / %.2e (gpu) of its value" % ((abs(c[k] - e_prev) / e_prev, abs(g[k] - e_prev) / e_prev) if k < m else (0, 0)))
    try:
        _compare_lm(sub, gpu, ref); print("   _compare_lm: passes")
    except AssertionError as e:
        print("   _compare_lm: FAILS", str(e)[:200])
