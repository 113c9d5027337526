"""End-to-end restatements of the reference's MiniProblems tests, run through the product front-end (include/srba.h) with the
CPU oracle plugged in as numeric back-end (these run without a GPU; tests/test_gpu_parity.py repeats them on the device)."""
import numpy as np

from srba_amd import capi, datasets, runner
import _oracle  # tests/_oracle.py: the CPU checker

INV = 2 ** 64 - 1


def test_submaps_edges_init_values():
    """tests/submaps_edge_init_values.cpp:81-173: at the loop closure (KF11 sees KF1) two edges are created, the loop-closure helper
    fields are set, num_observations > 1 and obs_rmse < 1e-6 (:162-169)."""
    for seed in (1, 2, 3):
        ds = datasets.graph_slam_from_entries(datasets.C1_SUBMAPS, 1e-3, np.radians(0.05), seed=seed)
        eng = runner.graph_slam_engine(backend=_oracle.BACKEND, submap=5, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.05, solver=capi.SOLVER_SCHUR_DENSE, max_error_per_obs_to_stop=1e-6)
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
    eng = runner.graph_slam_engine(backend=_oracle.BACKEND, submap=5, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.05, solver=capi.SOLVER_SCHUR_DENSE, max_error_per_obs_to_stop=1e-6)
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
        eng = runner.graph_slam_engine(backend=_oracle.BACKEND, solver=solver)
        infos = eng.run(ds)
        out.append(np.array([i.chi2_final for i in infos]))
        eng.close()
    assert np.allclose(out[0], out[1], rtol=1e-6, atol=1e-12) and np.allclose(out[0], out[2], rtol=1e-6, atol=1e-12)


def test_eval_overall_squared_error_graph_slam():
    """RbaEngine<>::eval_overall_squared_error (impl/eval_overall_error.h:15-137) through the front-end with the oracle plugged in:
    ~0 on the noise-free tutorial map after optimisation, and equal to a direct numpy evaluation over the final edges."""
    import numpy as np
    from srba_amd import datasets, runner
    ds = datasets.graph_slam_from_entries(datasets.C2_TUTORIAL_SE2)
    eng = runner.graph_slam_engine(backend=_oracle.BACKEND, submap=5, depth=3, sigma_xy=0.1, sigma_yaw_deg=4.0, harvest=0)
    eng.run(ds)
    e = eng.eval_overall_squared_error()
    assert 0 <= e < 1e-6
    # a noisy random-walk map: compare with an independent evaluation (poses chained over the same breadth-first paths)
    ds = datasets.graph_slam_se2(n_kf=60, seed=3, path="tour", sigma_xy=0.02, sigma_yaw_deg=0.5)
    eng = runner.graph_slam_engine(backend=_oracle.BACKEND, submap=10, depth=3, sigma_xy=0.02, sigma_yaw_deg=0.5, harvest=0)
    eng.run(ds)
    e = eng.eval_overall_squared_error()
    fr, to, pose = eng.edges()
    adj = {}
    for i in range(len(fr)):
        adj.setdefault(int(fr[i]), []).append((int(to[i]), i, False)); adj.setdefault(int(to[i]), []).append((int(fr[i]), i, True))

    def comp(a, b):
        c, s = np.cos(a[2]), np.sin(a[2]); return np.array([a[0] + b[0] * c - b[1] * s, a[1] + b[0] * s + b[1] * c, a[2] + b[2]])

    def inv(a):
        c, s = np.cos(a[2]), np.sin(a[2]); return np.array([-a[0] * c - a[1] * s, a[0] * s - a[1] * c, -a[2]])

    def rel(root, target):  # pose of target seen from root: breadth-first, adjacency order (spantree_create_complete.h)
        prev = {root: None}; q = [root]
        while q and target not in prev:
            cur = q.pop(0)
            for nk, ei, to_is_cur in adj[cur]:
                if nk not in prev: prev[nk] = (cur, ei, to_is_cur); q.append(nk)
        chain = []; k = target
        while prev[k] is not None: chain.append(prev[k]); k = prev[k][0]
        acc = np.zeros(3)
        for cur, ei, to_is_cur in reversed(chain):   # edge seen from `cur`: to_is_cur -> the child is edge.from -> pose = inv_pose ; else child is edge.to -> (-)inv_pose
            acc = comp(acc, pose[ei] if to_is_cur else inv(pose[ei]))
        return acc
    tot = 0.0
    for kf, frame in enumerate(ds):
        for fid, z, fl in zip(frame["feat_ids"], np.asarray(frame["z"]).reshape(-1, 3), frame["flags"]):
            if int(fid) == kf: continue   # the fixed self-landmark: zero residual by construction only if z == 0 (it is)
            a, b = kf, int(fid)
            p = rel(a, b) if a < b else inv(rel(b, a))
            c, s = np.cos(p[2]), np.sin(p[2]); dx, dy = z[0] - p[0], z[1] - p[1]
            r = np.array([dx * c + dy * s, -dx * s + dy * c, (z[2] - p[2] + np.pi) % (2 * np.pi) - np.pi])
            tot += float(r @ r)
    assert abs(e - tot) <= 1e-9 * max(1.0, tot), (e, tot)


def test_classic_linear_rba_edge_creation_policy():
    """ecps::classic_linear_rba (ecps/classic_linear_rba.h:50-118): always an edge (n-1)->n with a null initial pose, plus a loop-closure
    edge new_kf <- base_kf whenever a re-observed landmark's base is farther than max_tree_depth; the map still converges."""
    import numpy as np
    from srba_amd import capi, datasets, runner
    ds = datasets.graph_slam_se2(n_kf=80, seed=6, path="tour", sigma_xy=1e-3, sigma_yaw_deg=0.05, max_range=5.0)
    eng = runner.graph_slam_engine(backend=_oracle.BACKEND, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.05, harvest=0, ecp=1)
    infos = eng.run(ds)
    fr, to, pose = eng.edges()
    edges = list(zip(fr.astype(int).tolist(), to.astype(int).tolist()))
    for n in range(1, 80):
        assert (n - 1, n) in edges                       # the linear backbone
    extra = [e for e in edges if e[1] - e[0] != 1]
    assert all(e[0] < e[1] for e in extra)               # loop closures are (base -> new key-frame)
    # every loop closure links key-frames that were more than max_tree_depth apart along the backbone when it was created
    assert all(e[1] - e[0] > 3 for e in extra)
    # each key-frame reports its backbone edge first, with an approximate initial value
    assert all(i.n_new_edges >= 1 and i.edge_has_init[0] == 1 for i in infos[1:])
    assert max(i.obs_rmse for i in infos[5:]) < 0.05
    assert eng.eval_overall_squared_error() < 1.0


def test_get_global_graphslam_problem_export():
    """RbaEngine<>::get_global_graphslam_problem (impl/get_global_graphslam_problem.h:17-48): node poses = complete breadth-first spanning tree from the root
    (impl/spantree_create_complete.h), one constraint (to -> from, inv_pose) per kf2kf edge. Checked on a noise-free map: every exported node pose equals the
    ground-truth pose relative to the root, and every constraint is consistent with the two node poses it links."""
    ds = datasets.graph_slam_se2(n_kf=60, seed=3, sigma_xy=0.0, sigma_yaw_deg=0.0, path="tour")
    eng = runner.graph_slam_engine(backend=_oracle.BACKEND, submap=5, depth=3, harvest=0); eng.run(ds)
    ids, poses, ft, ep = eng.global_graphslam_problem(root=0)
    fr, to, inv_pose = eng.edges()
    assert list(ids) == list(range(60)) and len(ft) == len(fr)
    assert np.array_equal(ft[:, 0], to) and np.array_equal(ft[:, 1], fr) and np.allclose(ep, inv_pose)   # stored as "normal" poses to -> from (:41-46)
    def comp(a, b):
        c, s = np.cos(a[2]), np.sin(a[2]); return np.array([a[0] + b[0] * c - b[1] * s, a[1] + b[0] * s + b[1] * c, a[2] + b[2]])
    for k in range(len(ft)):   # pose(from) = pose(to) (+) inv_pose  (inv_pose = pose of `from` as seen from `to`)
        d = comp(poses[int(ft[k, 0])], ep[k]) - poses[int(ft[k, 1])]; d[2] = (d[2] + np.pi) % (2 * np.pi) - np.pi
        assert np.abs(d).max() < 1e-6, k
    root_alone = eng.global_graphslam_problem(root=7)
    assert np.allclose(root_alone[1][7], 0) and len(root_alone[0]) == 60
