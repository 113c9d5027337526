"""Export / configuration side of the header-only front-end (no numeric back-end involved: key-frames are added without optimisation): graphviz export
(impl/export_dot.h:15-126), scene geometry of build_opengl_representation (impl/export_opengl.h), INI parameter files (impl/rba_problem_common.h:60-92,
ecps/local_areas_fixed_size.h:36-47), camera calibration sections of the reference's dataset .cfg files, and the MRPT stand-ins the tutorials use
(CPose3DQuat difference, Eigen-spelled noise matrices, Gaussian draws). The C++ side is tests/cpp/frontend_exports.cpp."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_dot_scene_config_and_mrpt_stand_ins(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    exe = str(tmp_path / "frontend_exports")
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "mrpt_shims"), os.path.join(ROOT, "tests", "cpp", "frontend_exports.cpp"),
            "-o", exe,
                    "-L" + os.path.join(ROOT, "srba_amd", "lib"), "-lsrba_hip", "-Wl,-rpath," + os.path.join(ROOT, "srba_amd", "lib")], check=True, timeout=600)
    p = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    v = {l.split()[0]: l.split()[1:] for l in p.stdout.splitlines() if l.strip()}
    # 5 key-frames in sub-maps of 2: edges 0->1 (member), 0->2 (centre to centre), 2->3, 2->4
    assert v["keyframes"] == ["5"] and v["edges"] == ["4"] and v["known"] == ["1"] and v["unknown"] == ["6"] and v["observations"] == ["15"]
    dot = open(tmp_path / "graph.dot").read(); dot_lm = open(tmp_path / "graph_lm.dot").read(); top = open(tmp_path / "top.dot").read()
    assert dot.startswith("digraph G {") and "0; 1; 2; 3; 4; " in dot and all(e in dot for e in ("0->1;", "0->2;", "2->3;", "2->4;")) and "LANDMARKS" not in dot
    assert "0 -> L0; \n" in dot_lm and "0 -> L1; 0 -> L2; 1 -> L3; 2 -> L4; 3 -> L5; 4 -> L6; " in dot_lm and dot_lm.count(" -> L") == 1 + 6 + 15
    assert top.startswith("graph G {") and "0--2;" in top and top.count("--") == 1
    assert "0 [pos=" in top and "2 [pos=" in top and "3 [pos=" not in top        # only the key-frames with two or more kf2kf edges
    assert v["dot_bad_path"] == ["0"]
    # scene: a corner + a label per key-frame, a line per kf2kf edge, one point per landmark, a label per unknown landmark; schematic tree: one line per non-root key-frame
    assert v["scene_corners"] == ["5"] and v["scene_lines"] == ["4"] and v["scene_points"] == ["7"] and v["scene_texts"] == ["11"] and v["tree_lines"] == ["4"]
    assert v["scene_corners_depth1"] == ["2"]                                     # root 4 and its only neighbour 2
    assert v["cfg_max_tree_depth"] == ["7"] and float(v["cfg_max_lambda"][0]) == 1e9 and v["cfg_robust"] == ["1"] and v["cfg_cov"] == ["0"] and v["cfg_max_iters"] == ["33"]
    assert v["cfg_submap"] == ["2"] and v["cfg_min_obs"] == ["1"]
    ini = open(tmp_path / "params.ini").read()
    assert "[srba]" in ini and "[ecp]" in ini and "cov_recovery = crpNone" in ini
    assert v["cam_left"] == ["200", "150", "512", "384", "1024"] and v["cam_right"] == ["201", "151", "511", "383"] and v["cam_baseline"] == ["0.2"]
    assert float(v["pose_roundtrip"][0]) < 1e-12 and v["eigen_alias"] == ["4", "9"]
    assert abs(float(v["gauss_mean"][0]) - 2.0) < 0.02 and abs(float(v["gauss_std"][0]) - 0.5) < 0.02


import pytest


@pytest.mark.gpu
def test_hessian_condition_number_and_detailed_profiler_sections(tmp_path):
    """What the reference hands back next to the unknowns, through the front-end with the GPU back-end (tests/cpp/extra_results_gpu.cpp): extra_results.hessian = the
    system matrix of the last LM trial (symmetric, positive definite, 3 x unknown edges), HAp_condition_number, and the reference's per-stage opt.* profiler sections
    (SRBA_DETAILED_TIME_PROFILING) fed from the stage counters of the fused kernel."""
    import __graft_entry__ as ge
    ge.build()
    exe = str(tmp_path / "extra_results_gpu")
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "mrpt_shims"), os.path.join(ROOT, "tests", "cpp", "extra_results_gpu.cpp"),
            "-o", exe,
                    "-L" + os.path.join(ROOT, "srba_amd", "lib"), "-lsrba_hip", "-Wl,-rpath," + os.path.join(ROOT, "srba_amd", "lib")], check=True, timeout=600)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    v = {}; sec = {}
    for l in p.stdout.splitlines():
        w = l.split()
        if w and w[0] == "section": sec[w[1]] = float(w[2])
        elif w: v[w[0]] = w[1]
    n = 3 * int(v["edges"])
    assert int(v["edges"]) == 2 and v["hessian_valid"] == "1" and int(v["hessian_size"]) == n * n
    assert float(v["hessian_asym"]) == 0.0 and v["hessian_pd"] == "1" and float(v["hessian_min_diag"]) > 0
    assert 1.0 <= float(v["condition_number"]) < 1e12 and float(v["rmse"]) < 1.0
    for name in ("opt", "opt.update_spanning_tree_num", "opt.recompute_all_Jacobians", "opt.sparse_hessian_update_numeric", "opt.reprojection_residuals", "opt.compute_minus_gradient",
                 "opt.schur_build_reduced", "opt.DenseFill", "opt.DenseChol", "opt.backsub", "opt.schur_features", "opt.add_se3_deltas_to_frames"):
        assert sec[name] > 0, name
    assert sec["opt"] > sec["opt.DenseChol"]
