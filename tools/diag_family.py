"""Throughput of the fused LM kernel on a landmark family (BASELINE configs[2]-like): capsules harvested from a synthetic map, replicated to fill the
chip, GPU vs the oracle on the host. usage: diag_family.py [stereo|mono|cart3d|rb3d|rb2d|cart2d] [copies]"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from srba_amd import capi, datasets, runner
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _oracle  # tests/_oracle.py: the CPU checker (test infrastructure)
kind = sys.argv[1] if len(sys.argv) > 1 else "stereo"
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if kind in ("rb2d", "cart2d"):
    ds, _ = datasets.landmarks_dataset_se2(kind, n_kf=60, n_lm=900, seed=4, noise=1e-3)
else:
    ds, _ = datasets.landmarks_dataset_se3(kind, n_kf=60, n_lm=600, seed=5, noise=(0.1 if kind in ("stereo", "mono") else 1e-3), init_from_gt_noise=(0.2 if kind == "mono" else None))
eng = runner.landmark_engine(kind, backend="hip"); eng.run(ds)
b = eng.harvest(); n0 = b.n
arr = (capi.Capsule * (n0 * copies))()
for r in range(copies):
    for i in range(n0): arr[r * n0 + i] = b.ptr[i]
class Fake: pass
fb = Fake(); fb.ptr = C.cast(arr, capi.PCAP); fb.n = n0 * copies; fb.params = b.params; fb.family = b.family
ctx = runner.HipContext(b.params); ctx.upload(fb)
ctx.lm_run(); r = ctx.lm_run(); kms = ctx.lib.srba_hip_last_kernel_ms(ctx.ctx)
t = time.perf_counter(); ro = _oracle.run_batch(b, threads=1); dt = time.perf_counter() - t
nk = np.array([b.ptr[i].n_unk_edges for i in range(n0)]); nf = np.array([b.ptr[i].n_unk_lms for i in range(n0)]); no = np.array([b.ptr[i].n_obs for i in range(n0)])
print("%s: %d capsules x %d copies; mean unknowns %.1f edges + %.1f landmarks, %.0f observations; GPU %.2f ms -> %.3f M LM iterations/s ; oracle 1 thread %.1f k it/s ; ratio %.0f" % (
    kind, n0, copies, nk.mean(), nf.mean(), no.mean(), kms, r["num_trials"].sum() / kms / 1e3, ro["num_trials"].sum() / dt / 1e3, (r["num_trials"].sum() / kms * 1e3) / (ro["num_trials"].sum() / dt)))
