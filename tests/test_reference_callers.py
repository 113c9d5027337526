"""The boundary against the reference's own callers, without copying them.

Build container only (skipped where /root/reference is absent): every tutorial of examples/cpp and the whole srba-slam front-end of apps/srba-slam are compiled
IN PLACE against this repo's headers -- `g++ -std=c++17 -fsyntax-only -I include -I include/mrpt_shims` -- so every RbaEngine<> member, parameter struct, model
type and MRPT value type those programs touch must exist here with a compatible signature (the tutorials instantiate 10 engine types, the front-end 5).
`__graft_entry__.build()` additionally LINKS the reference's front-end against libsrba_hip.so into oracle/_ref/; on the GPU that binary must produce the same map
as this repo's own front-end on the same dataset (both are thin layers over the same engine: equal edges => equal overall squared error)."""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
FRONTEND = os.path.join(ROOT, "oracle", "_ref", "srba-slam-reference-frontend")
OURS = os.path.join(ROOT, "srba_amd", "bin", "srba-slam")
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples", "cpp")), reason="build-container test: /root/reference is not present")


def syntax_check(src):
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "mrpt_shims"), src], capture_output=True, text=True,
            timeout=600)
    return src, p.returncode, "\n".join(l for l in p.stderr.splitlines() if "error" in l)[:2000]


@needs_reference
def test_reference_tutorials_and_frontend_compile_against_these_headers():
    srcs = sorted(glob.glob(os.path.join(REF, "examples", "cpp", "tutorial-srba-*.cpp"))) + sorted(glob.glob(os.path.join(REF, "apps", "srba-slam", "*.cpp")))
    assert len([s for s in srcs if "tutorial" in s]) >= 11 and len([s for s in srcs if "instance_" in s]) == 5
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(syntax_check, srcs))
    bad = [(os.path.basename(s), err) for s, rc, err in res if rc != 0]
    assert not bad, bad


@needs_reference
def test_reference_frontend_links_and_lists_the_five_problems():
    import __graft_entry__ as ge
    ge.build()
    assert os.path.exists(FRONTEND)
    p = subprocess.run([FRONTEND, "--list-problems"], capture_output=True, text=True, timeout=60)   # (the reference's main returns 1 after printing the list)
    for line in ("--se2 --graph-slam", "--se2 --lm-2d --obs RangeBearing_2D", "--se3 --lm-3d --obs Cartesian_3D", "--se3 --lm-3d --obs MonocularCamera", "--se3 --lm-3d --obs StereoCamera"):
        assert line in p.stdout


def _run_frontend(binary, args, dot):
    p = subprocess.run([binary] + args + ["--save-final-graph-landmarks", dot], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return [l for l in p.stdout.splitlines() if l.startswith("[OPT] Final RMSE=") or l.startswith("[define_new_keyframe] Done.")], open(dot).read()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["graph-slam", "rb2d"])
def test_reference_frontend_on_the_gpu_builds_the_same_map(tmp_path, kind):
    """Both front-ends print the engine's per-optimisation summary ("[OPT] Final RMSE=... #iters=...", verbose 1) and save the final graph with landmarks as a
    graphviz file: the two runs must agree line by line (the reference's app has its numeric dumps compiled out, srba-run-generic-impl.h:600-618,722-774)."""
    if not os.path.exists(FRONTEND):
        pytest.skip("oracle/_ref/srba-slam-reference-frontend was not built (needs /root/reference at build time)")
    from srba_amd import datasets
    if kind == "graph-slam":
        ds = datasets.graph_slam_se2(n_kf=200, seed=4, path="tour"); f = str(tmp_path / "gs.txt"); datasets.write_text_dataset(ds, f, "graph-slam")
        sel = ["--se2", "--graph-slam", "--noise", "0.001", "--noise-ang", "0.2", "--submap-size", "10"]
    else:
        ds, _ = datasets.landmarks_dataset_se2("rb2d", n_kf=40, n_lm=600, seed=3, noise=0.01); f = str(tmp_path / "rb.txt"); datasets.write_text_dataset(ds, f, "rb2d")
        sel = ["--se2", "--lm-2d", "--obs", "RangeBearing_2D", "--noise", "0.05", "--noise-ang", "2.8", "--submap-size", "15"]
    common = sel + ["-d", f, "--max-spanning-tree-depth", "3", "--max-optimize-depth", "3", "--no-gui", "--verbose", "1"]
    theirs, their_dot = _run_frontend(FRONTEND, common, str(tmp_path / "theirs.dot"))
    ours, our_dot = _run_frontend(OURS, common, str(tmp_path / "ours.dot"))
    assert len(theirs) > 2 * (len(ds) - 2) and theirs == ours
    assert their_dot == our_dot and "KEYFRAME->KEYFRAME edges" in our_dot
