"""Integer state of the host graph layer of a finished run, in a comparable form (test infrastructure).

For the maps listed in MAPS (reference literal datasets C-1/C-2, a 2 000-keyframe cfg2 tour with loop closures, one map per landmark
family, one classic_linear_rba map) this module runs RbaEngine<>::define_new_keyframe() key-frame by key-frame (CPU oracle as numeric
back-end) and collects everything that `north_star` asks to be bit-exact:
  * the symbolic spanning trees: next_edge rows [src trg next dist] and all_edges rows [from to len e0 e1 ...] (srba_engine_st_dump),
  * the kf2kf edge list (from, to) in creation order,
  * every integer array of every problem capsule (stage-1 and local-area optimisations, harvest=3), as one 64-bit digest per capsule
    plus its sizes.
tests/golden/make_graph_golden.py stores that as tests/golden/graph_<map>.npz (written once from the round-1 front-end, BEFORE the
flat-container rewrite of the host layer); tests/test_graph_golden.py recomputes it with the current headers and compares bit by bit.
"""
import hashlib

import numpy as np

from srba_amd import capi, datasets, runner
import _oracle

INT_FIELDS = (("pair_path_off", "n_pairs", 1), ("path_edge", "n_path", 0), ("pair_needed", "n_pairs", 0), ("pose_required", "2n_pairs", 0),
              ("obs_pose", "n_obs", 0), ("obs_lm", "n_obs", 0), ("obs_valid", "n_obs", 0),
              ("bp_col", "n_bp", 0), ("bp_res", "n_bp", 0), ("bp_A", "n_bp", 0), ("bp_D", "n_bp", 0), ("bp_lm", "n_bp", 0), ("bp_normal", "n_bp", 0), ("colp_off", "n_unk_edges", 1),
              ("bf_col", "n_bf", 0), ("bf_res", "n_bf", 0), ("bf_pose", "n_bf", 0), ("colf_off", "n_unk_lms", 1),
              ("hap_i", "n_hap", 0), ("hap_j", "n_hap", 0), ("hap_term_off", "n_hap", 1), ("hap_t1", "n_hap_terms", 0), ("hap_t2", "n_hap_terms", 0),
              ("hf_i", "n_hf", 0), ("hf_j", "n_hf", 0), ("hf_term_off", "n_hf", 1), ("hf_t1", "n_hf_terms", 0), ("hf_t2", "n_hf_terms", 0),
              ("hapf_i", "n_hapf", 0), ("hapf_j", "n_hapf", 0), ("hapf_term_off", "n_hapf", 1), ("hapf_t1", "n_hapf_terms", 0), ("hapf_t2", "n_hapf_terms", 0),
              ("hap_diag", "n_unk_edges", 0), ("hf_diag", "n_unk_lms", 0),
              ("sch_term_off", "n_hap", 1), ("sch_b1", "n_sch_terms", 0), ("sch_b2", "n_sch_terms", 0), ("sch_lm", "n_sch_terms", 0),
              ("lm_hapf_off", "n_unk_lms", 1), ("lm_hapf_idx", "n_hapf", 0))
SIZE_FIELDS = ("n_edges", "n_unk_edges", "n_unk_lms", "n_known_lms", "n_pairs", "n_path", "n_obs", "n_valid", "n_bp", "n_bf", "n_hap", "n_hap_terms",
               "n_hf", "n_hf_terms", "n_hapf", "n_hapf_terms", "n_sch_terms")


def capsule_int_arrays(c):
    """name -> integer numpy array, for one srba_problem_capsule (ctypes struct)."""
    out = {}
    for name, cnt, extra in INT_FIELDS:
        n = 2 * c.n_pairs if cnt == "2n_pairs" else getattr(c, cnt)
        p = getattr(c, name)
        if name.startswith("sch_") and c.n_sch_terms == 0:
            out[name] = np.zeros(0, np.int64); continue
        if not p or n + extra <= 0:
            out[name] = np.zeros(0, np.int64); continue
        if name == "colf_off" and c.n_unk_lms == 0:
            out[name] = np.zeros(0, np.int64); continue
        if name == "lm_hapf_off" and c.n_unk_lms == 0:
            out[name] = np.zeros(0, np.int64); continue
        if name in ("hf_term_off", "hapf_term_off") and n == 0:
            out[name] = np.zeros(0, np.int64); continue
        out[name] = np.ctypeslib.as_array(p, shape=(n + extra,)).astype(np.int64)
    return out


def capsule_digest(c):
    h = hashlib.sha256()
    h.update(np.array([getattr(c, f) for f in SIZE_FIELDS], np.int64).tobytes())
    for name, arr in capsule_int_arrays(c).items():
        h.update(name.encode()); h.update(np.int64(arr.size).tobytes()); h.update(arr.tobytes())
    return np.frombuffer(h.digest()[:8], np.uint64)[0]


def _rows_digest(rows, key_col, n_keys):
    """one 64-bit digest per source key-frame over its rows (rows: 2-D int64, already in dump order)."""
    out = np.zeros(n_keys, np.uint64)
    if rows.size:
        keys = rows[:, key_col]
        bounds = np.flatnonzero(np.diff(keys)) + 1
        for blk in np.split(rows, bounds):
            out[int(blk[0, key_col])] = np.frombuffer(hashlib.sha256(blk.tobytes()).digest()[:8], np.uint64)[0]
    return out


def _all_edges_rows(flat):
    """all_edges dump [from to len e0..] -> list of rows padded to a fixed width (from, to, len, e0..e7)."""
    rows = []; i = 0
    while i < flat.size:
        ln = int(flat[i + 2]); r = np.full(3 + 8, -1, np.int64); r[:3 + ln] = flat[i:i + 3 + ln]; rows.append(r); i += 3 + ln
    return np.array(rows, np.int64).reshape(-1, 11)


def collect(eng, n_full_kfs=300):
    """Everything bit-exact about a finished run of `eng` (harvest=3)."""
    ne = eng.st_dump(0).reshape(-1, 4); ae = _all_edges_rows(eng.st_dump(1))
    fr, to, _ = eng.edges()
    n_kf = int(max(fr.max(), to.max())) + 1 if len(fr) else 1
    b = eng.harvest()
    out = dict(edges=np.stack([fr, to], 1).astype(np.int64), next_edge_digest=_rows_digest(ne, 0, n_kf), all_edges_digest=_rows_digest(ae, 0, n_kf),
               next_edge_head=ne[ne[:, 0] < n_full_kfs], all_edges_head=ae[ae[:, 0] < n_full_kfs],
               n_next_edge=np.int64(len(ne)), n_all_edges=np.int64(len(ae)),
               capsule_kf=np.array([eng.lib.srba_engine_harvest_kf(eng.h, i) for i in range(b.n)], np.int64),
               capsule_sizes=np.array([[getattr(b[i], f) for f in SIZE_FIELDS] for i in range(b.n)], np.int64).reshape(b.n, len(SIZE_FIELDS)),
               capsule_digest=np.array([capsule_digest(b[i]) for i in range(b.n)], np.uint64))
    return out


def _lm(kind, **kw):
    return lambda: runner.landmark_engine(kind, backend=_oracle.BACKEND, harvest=3, **kw)


def build_map(name):
    """(engine, dataset) of a named map."""
    if name == "c1_submaps":
        return (runner.graph_slam_engine(backend=_oracle.BACKEND, submap=5, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.05, solver=capi.SOLVER_SCHUR_DENSE, max_error_per_obs_to_stop=1e-6, harvest=3),
                datasets.graph_slam_from_entries(datasets.C1_SUBMAPS, 1e-3, np.radians(0.05), seed=1))
    if name == "c2_tutorial":
        return (runner.graph_slam_engine(backend=_oracle.BACKEND, submap=5, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.05, harvest=3),
                datasets.graph_slam_from_entries(datasets.C2_TUTORIAL_SE2, 1e-3, np.radians(0.05), seed=2))
    if name == "cfg2_tour2000":
        return runner.graph_slam_engine(backend=_oracle.BACKEND, harvest=3), datasets.graph_slam_se2(n_kf=2000, seed=1, path="tour")
    if name == "cfg2_random1500":
        return runner.graph_slam_engine(backend=_oracle.BACKEND, harvest=3), datasets.graph_slam_se2(n_kf=1500, seed=7, grid=4, block=30.0)
    if name == "linear_rba_se2":
        return (runner.graph_slam_engine(backend=_oracle.BACKEND, depth=3, sigma_xy=1e-3, sigma_yaw_deg=0.05, harvest=3, ecp=1),
                datasets.graph_slam_from_entries(datasets.C2_TUTORIAL_SE2, 1e-3, np.radians(0.05), seed=2))
    if name in ("rb2d", "cart2d"):
        ds, _ = datasets.landmarks_dataset_se2(name, n_kf=60, n_lm=900, seed=4, noise=1e-3)
        return runner.landmark_engine(name, backend=_oracle.BACKEND, harvest=3), ds
    if name in ("cart3d", "rb3d", "stereo", "mono"):
        noise = 1e-3 if name in ("cart3d", "rb3d") else 0.1
        ds, _ = datasets.landmarks_dataset_se3(name, n_kf=40, n_lm=600, seed=5, noise=noise, init_from_gt_noise=(0.2 if name == "mono" else None))
        return runner.landmark_engine(name, backend=_oracle.BACKEND, harvest=3, robust=(1 if name == "stereo" else 0)), ds
    if name == "stereo_sparse_solver_deep":
        ds, _ = datasets.landmarks_dataset_se3("stereo", n_kf=50, n_lm=500, seed=9, noise=0.1)
        return runner.landmark_engine("stereo", backend=_oracle.BACKEND, harvest=3, depth=4, submap=6, solver=capi.SOLVER_SCHUR_SPARSE, min_obs_to_loop_closure=3), ds
    raise KeyError(name)


MAPS = ("c1_submaps", "c2_tutorial", "cfg2_tour2000", "cfg2_random1500", "linear_rba_se2", "rb2d", "cart2d", "cart3d", "rb3d", "stereo", "mono", "stereo_sparse_solver_deep")


def run_map(name):
    eng, ds = build_map(name)
    eng.run(ds)
    return collect(eng)
