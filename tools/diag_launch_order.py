"""Duration of the fused LM launch, 24 launches back to back, plus the size / LDS of every class launch of the plan (which class is which plan job). usage: diag_launch_order.py [n_kf]
Environment knobs under test: SRBA_HIP_DELAY_US=a,b,c,... (microseconds each plan job's stream is held back), SRBA_HIP_CLASS_PRIO=0|1|2, SRBA_HIP_CLASS_STREAMS."""
import ctypes as C, glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
cache = sorted(glob.glob("/tmp/srba_bench_cache/caps_se2_tour_%d_seed1_*.bin" % n_kf))
b = runner.CapsuleBatch.load(cache[-1]) if cache else runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=n_kf, seed=1, path="tour"), backend="hip", submap=10, depth=3)
ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib; hist = (C.c_double * 4)()
def one():
    lib.srba_hip_reset_state(ctx.ctx); lib.srba_hip_lm_run_async(ctx.ctx); lib.srba_hip_sync(ctx.ctx); lib.srba_hip_kernel_ms_history(ctx.ctx, hist, 1); return hist[0]
one(); one()
import hashlib; lib.srba_hip_reset_state(ctx.ctx); r = ctx.lm_run(); sig = hashlib.sha1(r["chi2_final"].tobytes() + r["num_trials"].tobytes() + r["trace_chi2"].tobytes()).hexdigest()[:12]
v = np.array([one() for _ in range(24)])
print("%-60s mean %.2f  median %.2f  min %.2f  max %.2f | fast (<40 ms): %d of %d | %s" % (" ".join("%s=%s" % (k, os.environ[k]) for k in ("SRBA_HIP_STAGGER_NS", "SRBA_HIP_DELAY_US",
        "SRBA_HIP_CLASS_PRIO", "SRBA_HIP_CLASS_STREAMS", "SRBA_HIP_SCHED", "SRBA_HIP_WAVES_PER_CU", "SRBA_HIP_LDS_PER_CU_KB", "SRBA_HIP_LEAN", "SRBA_HIP_TWO",
        "SRBA_HIP_TWO_FROM_KB") if k in os.environ) or "(defaults)",
      v.mean(), np.median(v), v.min(), v.max(), (v < 40).sum(), len(v), " ".join("%.1f" % x for x in v[:6]) + " | results " + sig + " trials %d" % r["num_trials"].sum()))
