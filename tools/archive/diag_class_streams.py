"""Fused LM launch, sixteen launches back to back, for the number of streams the class launches are dealt to (SRBA_HIP_CLASS_STREAMS). usage: diag_class_streams.py N"""
import glob, os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
cache = sorted(glob.glob("/tmp/srba_bench_cache/caps_se2_tour_30000_seed*_*.bin"))
b = runner.CapsuleBatch.load(cache[-1])
ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib; hist = (C.c_double * 4)()
def one():
    lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx); lib.srba_hip_kernel_ms_history(ctx.ctx, hist, 1); return hist[0]
one(); v = np.array([one() for _ in range(16)])
print("class streams %s: mean %.2f  min %.2f  max %.2f  fast launches (< 39 ms): %d of 16 |" % (os.environ.get("SRBA_HIP_CLASS_STREAMS", "all"), v.mean(), v.min(), v.max(), int((v < 39).sum())), " ".join("%.1f" % x for x in v))
if len(sys.argv) > 1 and sys.argv[1] == "flush":   # the same with a read-only sweep over 1.2 GB between the launches (clean lines displace the dirty lines the previous launch left in L2 / Infinity Cache)
    import torch
    a = torch.ones(150_000_000, dtype=torch.float64, device="cuda")
    def flushed():
        s = a.sum(); torch.cuda.synchronize(); return one()
    v = np.array([flushed() for _ in range(16)])
    print("with a read-only sweep between launches: mean %.2f  min %.2f  max %.2f  fast: %d of 16 |" % (v.mean(), v.min(), v.max(), int((v < 39).sum())), " ".join("%.1f" % x for x in v))
    def dirtied():
        a.mul_(1.0); torch.cuda.synchronize(); return one()
    v = np.array([dirtied() for _ in range(16)])
    print("with a read-write sweep between launches: mean %.2f  min %.2f  max %.2f  fast: %d of 16 |" % (v.mean(), v.min(), v.max(), int((v < 39).sum())), " ".join("%.1f" % x for x in v))
