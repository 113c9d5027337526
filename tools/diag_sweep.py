"""Map sweep (RbaEngine<>::optimize_local_areas_batch through srba_amd.multi.sweep_map) against the same local areas re-optimised one optimize_local_area() call at a time, GPU back-end,
one process: time, LM trials, whole-map squared error before / after. usage: diag_sweep.py [n_kf] [stride]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srba_amd import datasets, multi, runner
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 6000; stride = int(sys.argv[2]) if len(sys.argv) > 2 else 1
def build():
    e = runner.graph_slam_engine(backend="hip", submap=10, depth=3, harvest=0, enable_profiler=1); t = time.perf_counter(); e.run(datasets.graph_slam_se2(n_kf=n_kf, seed=1, path="tour")); return e, time.perf_counter() - t
a, tb = build(); b, _ = build(); roots = np.arange(1, n_kf, stride, dtype=np.uint64)
print("map of %d key-frames built in %.1f s (%.3f ms per key-frame); overall squared error %.6e" % (n_kf, tb, 1e3 * tb / n_kf, a.eval_overall_squared_error()))
t = time.perf_counter(); round_of, off, touch, n_rounds = a.plan_sweep(roots, 3); tp = time.perf_counter() - t
t = time.perf_counter(); st = multi.sweep_map(a, roots, 3); ts = time.perf_counter() - t
trials = sum(st["info"][int(r)].lm.num_trials for r in roots)
print("sweep: %d windows in %d rounds (%.0f per round); plan %.2f s; run %.2f s = %.3f ms per window, %d LM trials -> %.0f trials/s; overall squared error %.6e" % (len(roots), n_rounds,
      len(roots) / max(1, n_rounds), tp, ts, 1e3 * ts / len(roots), trials, trials / ts, a.eval_overall_squared_error()))
for name in ("optimize_local_areas_batch", "opt.capsule", "opt.backend", "opt.backend.optimize_batch", "opt.backend.lm_run.kernel"):
    print("   %-34s mean %.3f ms" % (name, 1e3 * a.lib.srba_engine_profiler_mean(a.h, name.encode())))
t = time.perf_counter(); tr2 = 0
for c in range(n_rounds):
    for i in np.nonzero(round_of == c)[0]: tr2 += b.optimize_local_area(roots[i], 3).lm.num_trials
t2 = time.perf_counter() - t
print("the same schedule, one optimize_local_area() call at a time: %.2f s = %.3f ms per window, %d LM trials; overall squared error %.6e ; maps equal: %s (max |diff| %.2e)" % (t2, 1e3 * t2 / len(roots),
      tr2, b.eval_overall_squared_error(), np.array_equal(a.edges()[2], b.edges()[2]), np.abs(a.edges()[2] - b.edges()[2]).max()))
