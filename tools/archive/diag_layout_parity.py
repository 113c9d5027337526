"""chi2 parity of a landmark-family batch against the oracle under the layout knobs (SRBA_HIP_HBM_FROM_KB, SRBA_HIP_DENSE_LEFT): which windows differ, and by how much. usage: diag_layout_parity.py"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from srba_amd import capi, datasets, runner
import _oracle
kind = "mono"
ds, _ = datasets.landmarks_dataset_se3(kind, n_kf=60, n_lm=600, seed=5, noise=0.1, init_from_gt_noise=0.2)
eng = runner.landmark_engine(kind, backend=_oracle.BACKEND); eng.run(ds); b = eng.harvest(); b.engine = eng
ref = _oracle.run_batch(b); n0 = b.n
for copies in (1, 18):
    arr = (capi.Capsule * (n0 * copies))()
    for r in range(copies):
        for i in range(n0): arr[r * n0 + i] = b.ptr[i]
    class Rep: pass
    fb = Rep(); fb.ptr = C.cast(arr, capi.PCAP); fb.n = n0 * copies; fb.params = b.params; fb.family = b.family
    ctx = runner.HipContext(b.params); ctx.upload(fb); gpu = ctx.lm_run(); ctx.close()
    rel = np.array([abs(gpu["chi2_final"][q] - ref["chi2_final"][q % n0]) / ref["chi2_final"][q % n0] for q in range(fb.n)])
    nk = np.array([b[i].n_unk_edges for i in range(n0)])
    bad = np.flatnonzero(rel[:n0] > 1e-6)
    print(os.environ.get("SRBA_HIP_HBM_FROM_KB"), os.environ.get("SRBA_HIP_DENSE_LEFT"), "copies", copies, "conv", int((rel <= 1e-6).sum()), "of", fb.n, "max rel", rel.max(), "bad windows (edges):", [(int(i), int(nk[i]), float("%.2g" % rel[i])) for i in bad[:12]], "trials gpu/ref", int(gpu["num_trials"].sum()), int(ref["num_trials"].sum()) * copies)
