"""Which synthetic maps are parity datasets (CPU, oracle only).

chi2 parity at 1e-6 (BASELINE.json north_star) is a statement about problems whose solution the reference's arithmetic defines to that precision. The reference ships
three solvers for the same normal equations (lev-marq_solvers.h:28-209, 214-405, 410-591); Schur + dense LL^t and Schur + sparse Cholesky perform the same algebra in two
elimination orders, so the distance between their results on one window measures how well that window is conditioned -- no third implementation (the GPU's block-sparse
LL^t is one) can be asked to sit closer to either of them than they sit to each other.

* gauge-fixed monocular map (the landmarks of key-frame 0 known, like tutorial-srba-monocular-se3.cpp): converges to the pixel noise, the two solvers agree to 1e-9 on every
  window and take the same number of trials -> parity dataset, used at 100 % by tests/test_gpu_parity.py::test_large_batch_moves_wide_lds_images_to_hbm[mono];
* gauge-free monocular map with 20 cm depth noise (round 2's dataset of that test): the map is lost (RMSE ~ 100 px). Its LM traces are chaotic: on the build container's
  CPU the two solvers drift apart geometrically (5e-11 at the first trial of window 31, a different accept / reject decision 13 trials later, chi2_final 9.4e-6 apart, 19
  against 39 trials); on the GPU box's host (another libm code path) they happen to stay together. Either way only the prefix of a trace up to the first decision two runs take
  differently is comparable (test_lost_monocular_map_agrees_up_to_a_rounding_floor_decision).
"""
import numpy as np

from srba_amd import capi, datasets, runner
import _oracle


def _two_schur_solvers(ds):
    eng = runner.landmark_engine("mono", backend=_oracle.BACKEND); eng.run(ds); b = eng.harvest(); b.engine = eng
    r0 = _oracle.run_batch(b, threads=4); b.params.solver = capi.SOLVER_SCHUR_SPARSE; r1 = _oracle.run_batch(b, threads=4); b.params.solver = capi.SOLVER_SCHUR_DENSE
    return b, r0, r1


def test_gauge_fixed_monocular_map_is_a_parity_dataset():
    ds, _ = datasets.landmarks_dataset_se3("mono", n_kf=60, n_lm=600, seed=5, noise=0.1, init_from_gt_noise=0.05, known_first=1000)
    b, r0, r1 = _two_schur_solvers(ds)
    rel = np.abs(r1["chi2_final"] - r0["chi2_final"]) / r0["chi2_final"]
    assert rel.max() < 1e-9 and np.array_equal(r0["num_trials"], r1["num_trials"])
    assert np.median(r0["obs_rmse"]) < 0.2          # pixel noise 0.1: the map converges


def test_gauge_free_monocular_map_is_lost_and_its_solvers_split_only_at_the_floor():
    ds, _ = datasets.landmarks_dataset_se3("mono", n_kf=60, n_lm=600, seed=5, noise=0.1, init_from_gt_noise=0.2)
    b, r0, r1 = _two_schur_solvers(ds)
    assert np.median(r0["obs_rmse"]) > 50.0          # a lost map (pixel noise 0.1)
    for i in range(b.n):                             # wherever the reference's two Schur solvers part ways, they do so at a rounding-floor decision
        m = int(min(r0["num_trials"][i], r1["num_trials"][i], capi.TRACE_LEN))
        c0, c1 = r0["trace_chi2"][i][:m], r1["trace_chi2"][i][:m]
        same = (np.sign(r0["trace_rho"][i][:m]) == np.sign(r1["trace_rho"][i][:m])) & (np.isnan(c0) == np.isnan(c1))
        k = m if same.all() else int(np.argmin(same))
        acc = r0["trace_rho"][i][:k] > 0
        assert np.allclose(c0[:k][acc], c1[:k][acc], rtol=1e-6), i
        if k < m:
            e_prev = c0[np.flatnonzero(acc)[-1]] if acc.any() else r0["chi2_init"][i]
            assert all(np.isnan(e) or abs(e - e_prev) <= 1e-6 * e_prev for e in (c0[k], c1[k])), (i, k)
