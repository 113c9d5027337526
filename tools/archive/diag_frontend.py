"""Where a sequential define_new_keyframe() run with the GPU back-end spends its time (host graph layer vs per-call GPU path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srba_amd import datasets, runner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ds = datasets.graph_slam_se2(n_kf=n, seed=1, path="tour")
eng = runner.graph_slam_engine(backend="hip", submap=10, depth=3, harvest=0, enable_profiler=1)
t = time.time(); eng.run(ds); dt = time.time() - t
print("n=%d sequential run with the GPU back-end: %.2f s = %.3f ms/KF" % (n, dt, 1e3 * dt / n))
for name in ("define_new_keyframe", "define_new_keyframe.determine_edges", "define_new_keyframe.st.update_symbolic", "define_new_keyframe.optimize", "opt", "opt.sparse_hessian_build_symbolic",
             "opt.backend", "opt.backend.optimize_capsule", "opt.backend.lm_run.kernel"):
    print("   %-45s mean %.4f ms" % (name, 1e3 * eng.lib.srba_engine_profiler_mean(eng.h, name.encode())))
