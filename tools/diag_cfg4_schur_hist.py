"""(GPU box) cfg4: how the Schur terms of a deep window spread over its U_Ap blocks -- histogram of terms per block, share of the terms in blocks above a size, W blocks per landmark."""
import sys, os, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srba_amd import datasets, runner, multi
n_kf = 300
ds, _ = datasets.mono_deep_window(n_kf=n_kf, n_lm=40 * n_kf, seed=multi.replica_seed(0))
eng = runner.landmark_engine("mono", backend="hip", depth=8, submap=20, sigma=0.5, robust=0, harvest=1, cam=(200., 200., 400., 320.), refresh_all_read_poses=2)
eng.run(ds); b = eng.harvest(); c = b.ptr[b.n - 1]
off = np.ctypeslib.as_array(c.sch_term_off, shape=(c.n_hap + 1,)).copy(); ln = np.diff(off)
hi = np.ctypeslib.as_array(c.hap_i, shape=(c.n_hap,)); hj = np.ctypeslib.as_array(c.hap_j, shape=(c.n_hap,)); diag = hi == hj
print("window: %d unknown edges, %d landmarks, %d obs, n_hap %d, Schur terms %d, W blocks (n_hapf) %d = %.1f per landmark" % (c.n_unk_edges, c.n_unk_lms, c.n_obs, c.n_hap, c.n_sch_terms, c.n_hapf,
        c.n_hapf / max(c.n_unk_lms, 1)))
print("blocks with terms: %d ; diagonal blocks %d holding %.1f %% of the terms (mean %.0f, max %d)" % ((ln > 0).sum(), diag.sum(), 100 * ln[diag].sum() / ln.sum(), ln[diag].mean(), ln[diag].max()))
for lo, hi_ in ((1, 16), (17, 64), (65, 128), (129, 256), (257, 512), (513, 1024), (1025, 4096), (4097, 10**9)):
    m = (ln >= lo) & (ln <= hi_); print("  %5d .. %-10d terms: %6d blocks, %5.1f %% of the terms, lane utilisation of a 256-thread pass %.0f %%" % (lo, hi_, m.sum(), 100 * ln[m].sum() / ln.sum(),
            100 * ln[m].sum() / max(1, (np.ceil(ln[m] / 256) * 256).sum())))
lm = np.ctypeslib.as_array(c.sch_lm, shape=(c.n_sch_terms,)); per_lm = np.bincount(lm, minlength=c.n_unk_lms)
print("terms per landmark: mean %.0f median %.0f max %d" % (per_lm.mean(), np.median(per_lm), per_lm.max()))
