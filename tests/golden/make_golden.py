"""Regenerates the golden fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

Inputs: literal datasets quoted from the reference's own tests / tutorials (srba_amd/datasets.py cites file:line) and small seeded synthetic
problems of every model family.  The reference cannot be built or imported in this image (MRPT/Eigen/CSparse absent), so the expected
outputs are produced by the CPU oracle (oracle/srba_oracle.cpp); they pin the oracle against regressions and give the GPU tests a
device-independent target.  Each fixture = a capsule file (srba_engine_harvest_save format, a few KB each) + an .npz with the oracle's results."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from srba_amd import capi, datasets, runner  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); import _oracle  # tests/_oracle.py: the CPU checker (test infrastructure)

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, eng, first, count):
    b = eng.harvest()
    first = max(0, min(first, b.n - count))
    path = os.path.join(OUT, name + ".caps")
    assert eng.lib.srba_engine_harvest_save(eng.h, path.encode(), first, count) == 0
    sub = runner.CapsuleBatch.load(path)
    r = _oracle.run_batch(sub, keep_state=True)
    P, L, O, PD = capi.DIMS[sub.family]
    edges = [r["state"].array(i, "edge_pose", np.float64, sub[i].n_unk_edges * PD) for i in range(sub.n)]
    lms = [r["state"].array(i, "ulm_pos", np.float64, sub[i].n_unk_lms * L) for i in range(sub.n)]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), chi2_init=r["chi2_init"], chi2_final=r["chi2_final"], lambda_init=r["lambda_init"], num_trials=r["num_trials"],
                        num_observations=r["num_observations"], num_jacobians=r["num_jacobians"], trace_chi2=r["trace_chi2"], trace_rho=r["trace_rho"],
                        edges=np.concatenate(edges) if edges else np.zeros(0), lms=np.concatenate(lms) if lms else np.zeros(0))
    print("%-28s %d capsules, %d bytes" % (name, sub.n, os.path.getsize(path)))


def new_families():
    """the two families added in round 2 (fixtures added in round 3; the older fixtures are left as they are)"""
    ds, _ = datasets.graph_slam_se3(n_kf=40, seed=6)
    eng = runner.graph_slam_engine_se3(backend=_oracle.BACKEND); eng.run(ds); save("lm_relpose3d", eng, 30, 4)
    ds, _ = datasets.landmarks_dataset_se2_stereo(n_kf=16, n_lm=90, seed=7, noise=0.1)
    eng = runner.landmark_engine("stereo_se2", backend=_oracle.BACKEND); eng.run(ds); save("lm_stereo_se2", eng, 12, 3)


def main():
    if "--new-families" in sys.argv:
        return new_families()
    # C-1: tests/submaps_edge_init_values.cpp (loop closure at KF 11)
    eng = runner.graph_slam_engine(backend=_oracle.BACKEND, submap=5, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.05, solver=capi.SOLVER_SCHUR_DENSE, max_error_per_obs_to_stop=1e-6)
    eng.run(datasets.graph_slam_from_entries(datasets.C1_SUBMAPS, 1e-3, np.radians(0.05), seed=1)); save("c1_submaps_se2", eng, 8, 6)
    # C-2: tutorial-srba-relative-graph-slam-se2.cpp with its noise level, srba-slam's solver
    eng = runner.graph_slam_engine(backend=_oracle.BACKEND, submap=5, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.05)
    eng.run(datasets.graph_slam_from_entries(datasets.C2_TUTORIAL_SE2, 1e-3, np.radians(0.05), seed=2)); save("c2_tutorial_se2", eng, 10, 6)
    # cfg2-like synthetic windows (submap 10, depth 3)
    eng = runner.graph_slam_engine(backend=_oracle.BACKEND)
    eng.run(datasets.graph_slam_se2(n_kf=70, seed=3, path="tour")); save("cfg2_tour_se2", eng, 60, 4)
    # landmark families
    for kind in ("rb2d", "cart2d"):
        ds, _ = datasets.landmarks_dataset_se2(kind, n_kf=24, n_lm=900, seed=4, noise=1e-3)
        eng = runner.landmark_engine(kind, backend=_oracle.BACKEND); eng.run(ds); save("lm_" + kind, eng, 20, 3)
    for kind, noise in (("cart3d", 1e-3), ("rb3d", 1e-3), ("stereo", 0.1), ("mono", 0.1)):
        ds, _ = datasets.landmarks_dataset_se3(kind, n_kf=12, n_lm=300, seed=5, noise=noise, init_from_gt_noise=(0.2 if kind == "mono" else None))
        eng = runner.landmark_engine(kind, backend=_oracle.BACKEND, robust=(1 if kind == "stereo" else 0)); eng.run(ds); save("lm_" + kind, eng, 9, 3)
    new_families()


if __name__ == "__main__":
    main()
