"""The small windows of the benchmark batch (<= 31 unknown edges: the wave-slot-bound size classes) alone, and the big ones alone, at 2 .. 8 resident wavefronts per CU (the grid of the persistent
launches): where does the throughput of each phase of the fused launch saturate? usage: diag_small_classes.py"""
import ctypes as C, glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
cache = sorted(glob.glob("/tmp/srba_bench_cache/caps_se2_tour_30000_seed1_*.bin"))
b = runner.CapsuleBatch.load(cache[-1]) if cache else runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=3000, seed=1, path="tour"), backend="hip", submap=10, depth=3)
nk = np.array([b.ptr[i].n_unk_edges for i in range(b.n)])
def batch_of(idx):
    arr = (capi.Capsule * len(idx))()
    for k, i in enumerate(idx): arr[k] = b.ptr[int(i)]
    class Fake: pass
    fb = Fake(); fb.ptr = C.cast(arr, capi.PCAP); fb.n = len(idx); fb.params = b.params; fb.family = b.family; fb._keep = arr
    return fb
for label, idx in (("small (<= 31 edges)", np.flatnonzero(nk <= 31)), ("big (> 31 edges)", np.flatnonzero(nk > 31))):
    fb = batch_of(idx)
    for w in [int(x) for x in os.environ.get("DIAG_WAVES", "2,3,4,5,6,8").split(",")]:
        os.environ["SRBA_HIP_WAVES_PER_CU"] = str(w)
        ctx = runner.HipContext(b.params); ctx.upload(fb); lib = ctx.lib; hist = (C.c_double * 4)(); r = ctx.lm_run()
        def one():
            lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx); lib.srba_hip_kernel_ms_history(ctx.ctx, hist, 1); return hist[0]
        one(); v = np.array([one() for _ in range(5)])
        print("%-20s %5d capsules, at most %d wavefronts per CU: %.2f ms -> %.2f M trials/s" % (label, fb.n, w, v.mean(), r["num_trials"].sum() / v.mean() / 1e3), flush=True); ctx.close()
