"""Is a phase of the fused launch bound by how many capsules are resident? The big windows (> 31 unknown edges) and the small ones of the benchmark batch alone, with SRBA_HIP_LDS_PAD adding
dead LDS to every workgroup (fewer resident wavefronts per CU, same kernel, same work). usage: diag_residency.py"""
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
for label, idx, pads in (("big (> 31 edges)", np.flatnonzero(nk > 31), (0, 8, 16, 32, 64)), ("small (<= 31 edges)", np.flatnonzero(nk <= 31), (0, 4, 8, 16, 32))):
    fb = batch_of(idx)
    for pad in pads:
        os.environ["SRBA_HIP_LDS_PAD"] = str(pad * 1024)
        ctx = runner.HipContext(b.params); ctx.upload(fb); lib = ctx.lib; hist = (C.c_double * 4)(); r = ctx.lm_run()
        def one():
            lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx); lib.srba_hip_kernel_ms_history(ctx.ctx, hist, 1); return hist[0]
        one(); v = np.array([one() for _ in range(4)])
        print("%-20s %5d capsules, +%2d KB of dead LDS per workgroup: %.2f ms -> %.2f M trials/s" % (label, fb.n, pad, v.mean(), r["num_trials"].sum() / v.mean() / 1e3), flush=True); ctx.close()
