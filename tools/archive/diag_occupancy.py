"""Throughput of the fused LM kernel vs residency: one typical capsule replicated N times, LDS padded to force 1..k waves per CU.
usage: diag_occupancy.py [nb_target] [copies]"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srba_amd import capi, datasets, runner
nb_t = int(sys.argv[1]) if len(sys.argv) > 1 else 27
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
ds = datasets.graph_slam_se2(n_kf=1500, seed=1, path="tour")
b = runner.harvest_graph_slam(ds, backend="hip", submap=10, depth=3)
nk = np.array([b.ptr[i].n_unk_edges for i in range(b.n)])
i0 = int(np.argmin(np.abs(nk - nb_t) + (np.arange(b.n) < 200) * 1000))
print("capsule %d with %d unknown edges, %d obs" % (i0, nk[i0], b.ptr[i0].n_obs))
arr = (capi.Capsule * copies)()
for i in range(copies): arr[i] = b.ptr[i0]
class Fake: pass
fb = Fake(); fb.ptr = C.cast(arr, capi.PCAP); fb.n = copies; fb.params = b.params; fb.family = b.family
for pad_kb in [int(x) for x in os.environ.get('OCC_PADS', '0,4,8,12,16,24,32,48,72,140').split(',')]:
    os.environ["SRBA_HIP_LDS_PAD"] = str(pad_kb * 1024)
    ctx = runner.HipContext(b.params); ctx.upload(fb)
    ctx.lm_run(); r = ctx.lm_run(); kms = ctx.lib.srba_hip_last_kernel_ms(ctx.ctx)
    shp = ctx.debug(11).reshape(copies, 4)
    lds = shp[0, 0] + pad_kb * 1024
    print("LDS/WG %6.1f KB -> %2d WG/CU by LDS: kernel %.2f ms, %d trials, %.2f M trials/s, per-CU-resident-wave rate %.1f trials/ms" % (lds / 1024, int(160 * 1024 // lds), kms, r["num_trials"].sum(), r["num_trials"].sum() / kms / 1e3, r["num_trials"].sum() / kms / 256 / max(1, int(160 * 1024 // lds))))
    if os.environ.get("SRBA_HIP_PHASE_TIMING") == "1":
        pc = ctx.debug(10).reshape(copies, 16).sum(axis=0); tr = r["num_trials"].sum() * 2  # two runs accumulated
        names = ["K1all", "jac", "hess", "resid", "grad", "solve", "apply", "K1need", "restore", "schur", "assemble", "factor", "bsub", "feat"]
        print("   us/trial:", " ".join("%s %.1f" % (names[k], pc[k] * 1e-2 / tr) for k in range(14)), "| sum %.1f" % (pc[:9].sum() * 1e-2 / tr))
    del ctx
