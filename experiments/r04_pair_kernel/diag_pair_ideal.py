"""k_lm_pair against k_lm_run on (a) ONE typical capsule replicated (the two halves of every wavefront in perfect lock-step) and (b) the capsules of the small size classes of the benchmark
batch only. usage: diag_pair_ideal.py [copies]"""
import ctypes as C, glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cache = sorted(glob.glob("/tmp/srba_bench_cache/caps_se2_tour_30000_seed1_*.bin"))
b = runner.CapsuleBatch.load(cache[-1]) if cache else runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=3000, seed=1, path="tour"), backend="hip", submap=10, depth=3)
nk = np.array([b.ptr[i].n_unk_edges for i in range(b.n)])
def batch_of(idx):
    arr = (capi.Capsule * len(idx))()
    for k, i in enumerate(idx): arr[k] = b.ptr[int(i)]
    class Fake: pass
    fb = Fake(); fb.ptr = C.cast(arr, capi.PCAP); fb.n = len(idx); fb.params = b.params; fb.family = b.family; fb._keep = arr
    return fb
def run(fb, label):
    for mode in ("0", "1"):
        os.environ["SRBA_HIP_PAIR"] = mode
        ctx = runner.HipContext(b.params); ctx.upload(fb); lib = ctx.lib; hist = (C.c_double * 4)()
        r = ctx.lm_run()
        def one():
            lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx); lib.srba_hip_kernel_ms_history(ctx.ctx, hist, 1); return hist[0]
        one(); v = np.array([one() for _ in range(6)])
        print("%-40s pair=%s: %.2f ms (min %.2f) ; %d trials -> %.2f M trials/s" % (label, mode, v.mean(), v.min(), r["num_trials"].sum(), r["num_trials"].sum() / v.mean() / 1e3), flush=True)
        ctx.close()
i0 = int(np.flatnonzero(nk == 27)[len(np.flatnonzero(nk == 27)) // 2])
run(batch_of([i0] * copies), "one capsule (27 edges) x %d" % copies)
small = np.flatnonzero(nk <= 31)
run(batch_of(small), "the %d capsules with <= 31 unknown edges" % len(small))
