"""End-to-end restatements of the reference's MiniProblems tests, run through the product front-end (include/srba.h) with the
CPU oracle plugged in as numeric back-end (these run without a GPU; tests/test_gpu_parity.py repeats them on the device)."""
import numpy as np

from srba_amd import capi, datasets, runner

INV = 2 ** 64 - 1


def test_submaps_edges_init_values():
    """tests/submaps_edge_init_values.cpp:81-173: at the loop closure (KF11 sees KF1) two edges are created, the loop-closure helper
    fields are set, num_observations > 1 and obs_rmse < 1e-6 (:162-169)."""
    for seed in (1, 2, 3):
        ds = datasets.graph_slam_from_entries(datasets.C1_SUBMAPS, 1e-3, np.radians(0.05), seed=seed)
        eng = runner.graph_slam_engine(backend="oracle", submap=5, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.05, solver=capi.SOLVER_SCHUR_DENSE, max_error_per_obs_to_stop=1e-6)
        n_lc = 0
        for k in ds:
            info = eng.add_keyframe(k["feat_ids"], k["z"], k["flags"])
            if info.n_new_edges == 2:
                n_lc += 1
                assert info.lc_base[0] != INV or info.lc_base[1] != INV
                assert info.lc_observer[0] != INV or info.lc_observer[1] != INV
                assert info.num_observations > 1
                assert info.obs_rmse < 1e-6
        assert n_lc == 1
        fr, to, pose = eng.edges()
        assert (int(fr[11]), int(to[11])) == (0, 10)           # the loop-closure edge links the two area centres
        assert abs(pose[11][0] + 10.05) < 0.05 and abs(pose[11][1]) < 0.05
        eng.close()


def test_tutorial_relative_graph_slam_se2_recovers_ground_truth():
    """examples/cpp/tutorial-srba-relative-graph-slam-se2.cpp: noise-free data => every optimisation ends at ~zero error and the
    kf2kf edges reproduce the dataset's relative poses."""
    ds = datasets.graph_slam_from_entries(datasets.C2_TUTORIAL_SE2)
    eng = runner.graph_slam_engine(backend="oracle", submap=5, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.05, solver=capi.SOLVER_SCHUR_DENSE, max_error_per_obs_to_stop=1e-6)
    infos = eng.run(ds)
    assert all(i.obs_rmse < 1e-5 for i in infos[1:])
    fr, to, pose = eng.edges()
    # ground-truth global poses by chaining the consecutive-keyframe observations (k sees k-1)
    cons = {e[0]: e[2:] for e in datasets.C2_TUTORIAL_SE2 if e[1] == e[0] - 1}
    gt = [(0.0, 0.0, 0.0)]
    for k in range(1, 17):
        z = cons[k]                                   # pose of k-1 as seen from k
        c, s_ = np.cos(z[2]), np.sin(z[2])
        inv = (-z[0] * c - z[1] * s_, z[0] * s_ - z[1] * c, -z[2])  # pose of k as seen from k-1
        gt.append(datasets._compose2(gt[-1], inv))
    for f, t, p in zip(fr, to, pose):
        want = datasets._inv_compose2(gt[int(f)], gt[int(t)])  # pose of `from` as seen from `to`
        assert np.allclose(p[:2], want[:2], atol=1e-4) and abs((p[2] - want[2] + np.pi) % (2 * np.pi) - np.pi) < 1e-4, (int(f), int(t), p, want)
    eng.close()


def test_both_graph_slam_solvers_agree():
    """srba-slam's instance uses the no-Schur sparse solver, the test/tutorial the default Schur+dense one: same systems, same answers."""
    ds = datasets.graph_slam_se2(n_kf=80, seed=4, grid=2, block=30.0)
    out = []
    for solver in (capi.SOLVER_NO_SCHUR_SPARSE, capi.SOLVER_SCHUR_DENSE, capi.SOLVER_SCHUR_SPARSE):
        eng = runner.graph_slam_engine(backend="oracle", solver=solver)
        infos = eng.run(ds)
        out.append(np.array([i.chi2_final for i in infos]))
        eng.close()
    assert np.allclose(out[0], out[1], rtol=1e-6, atol=1e-12) and np.allclose(out[0], out[2], rtol=1e-6, atol=1e-12)
