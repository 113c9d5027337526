"""The reference's SchurTests on its EXACT grid (tests/schur_unittest.cpp:285-360): for 2 key-frames + k2k unknown edges and k2f unknown landmarks,
each landmark seen from key-frame 0 (dh_df block only) and from key-frame e>0 (dh_dAp block of edge e-1 + dh_df block) with the test's visibility
probabilities / masks, random Gaussian Jacobian blocks of the <SE3, Euclidean3D, Cartesian_3D> sizes (3x6, 3x3), minus_grad = 1, lambda = 1e3:
the block-sparse path -- Hessians over the symbolic plan (sparse_hessian_build_symbolic + sparse_hessian_update_numeric), then
SchurComplement::numeric_build_reduced_system -- must equal the dense formulas
      HAp - HApf (Hf + lambda I)^-1 HApf^t      and      g_Ap - HApf (Hf + lambda I)^-1 g_f          to 1e-10 of the largest entry (:258,:267).
The symbolic plan comes from the product's CapsuleData::build_plan; the dense matrices are formed here in numpy from the blocks alone, so a block or a
term missing from the plan shows up as a mismatch. CPU: the oracle's numeric path; GPU: K6 + K7/K8 on the device through the stepwise C ABI.
(The reference draws from mrpt's random generator; the draws here come from numpy with the same seeds -- the property does not depend on the values.)"""
import ctypes as C

import numpy as np
import pytest

from srba_amd import capi, runner
import _oracle  # tests/_oracle.py: the CPU checker

P, L, O = 6, 3, 3
MASKS = [(2, 5, [1, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1]),
         (2, 6, [0, 1, 0, 1, 0, 1, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 1, 1])]
GRID = [(1, 6, 1.0, range(1, 5)), (1, 7, 1.0, range(1, 15)), (1, 7, 0.95, range(1, 15)), (1, 30, 1.0, range(1, 15)), (2, 20, 0.9, range(1, 5)), (5, 30, 0.9, range(1, 5))]


def cases():
    for nK, nF, prob, seeds in GRID:
        for seed in seeds:
            rng = np.random.RandomState(seed)
            vis = rng.uniform(size=(nK + 1) * nF) <= prob
            yield "k%d_f%d_p%g_s%d" % (nK, nF, prob, seed), nK, nF, vis, seed
    for nK, nF, m in MASKS:
        yield "mask_k%d_f%d" % (nK, nF), nK, nF, np.array(m, bool), 1


CASES = list(cases())


def build(nK, nF, vis, seed):
    """capsule + random blocks + the dense reference result"""
    rows = [(kf, lm) for kf in range(nK + 1) for lm in range(nF) if vis[kf * nF + lm]]   # observation order of the reference loop (:104-137)
    n = len(rows)
    bp_col = np.array([kf - 1 for kf, _ in rows], np.int32); bf_col = np.array([lm for _, lm in rows], np.int32)
    lib = capi.engine_lib()
    h = lib.srba_capsule_from_blocks(capi.SE3_CART3D, nK, nF, n, bp_col.ctypes.data_as(capi.PI32), bf_col.ctypes.data_as(capi.PI32), 1)
    assert h
    prm = capi.HipParams(); capi.hip_lib().srba_hip_params_default(C.byref(prm), capi.SE3_CART3D)   # identity noise, sigma = 1, Schur + dense Cholesky: the test's my_srba_t defaults
    b = runner.CapsuleBatch(handle=h, params=prm)
    c = b[0]
    rng = np.random.RandomState(1000 + seed)
    Jp = rng.standard_normal((c.n_bp, O, P)); Jf = rng.standard_normal((c.n_bf, O, L))
    # dense J from the block tables
    A = np.zeros((n * O, nK * P)); F = np.zeros((n * O, nF * L))
    col = np.ctypeslib.as_array(c.bp_col, shape=(c.n_bp,)) if c.n_bp else []; res = np.ctypeslib.as_array(c.bp_res, shape=(c.n_bp,)) if c.n_bp else []
    for k in range(c.n_bp):
        A[res[k] * O:(res[k] + 1) * O, col[k] * P:(col[k] + 1) * P] = Jp[k]
    colf = np.ctypeslib.as_array(c.bf_col, shape=(c.n_bf,)); resf = np.ctypeslib.as_array(c.bf_res, shape=(c.n_bf,))
    for k in range(c.n_bf):
        F[resf[k] * O:(resf[k] + 1) * O, colf[k] * L:(colf[k] + 1) * L] = Jf[k]
    lam = 1e3; g = np.ones(nK * P + nF * L)
    HAp, Hf, HApf = A.T @ A, F.T @ F, A.T @ F
    Hfi = np.linalg.inv(Hf + lam * np.eye(nF * L))
    return b, Jp, Jf, g, lam, HAp - HApf @ Hfi @ HApf.T, g[:nK * P] - HApf @ Hfi @ g[nK * P:]


def dense_from_blocks(c, blocks, nK):
    """symmetric dense matrix from the stored upper blocks (i <= j)"""
    M = np.zeros((nK * P, nK * P)); hi = np.ctypeslib.as_array(c.hap_i, shape=(c.n_hap,)); hj = np.ctypeslib.as_array(c.hap_j, shape=(c.n_hap,))
    for k in range(c.n_hap):
        B = blocks[k * P * P:(k + 1) * P * P].reshape(P, P)
        M[hi[k] * P:(hi[k] + 1) * P, hj[k] * P:(hj[k] + 1) * P] = B
        if hi[k] != hj[k]:
            M[hj[k] * P:(hj[k] + 1) * P, hi[k] * P:(hi[k] + 1) * P] = B.T
    # the stored diagonal blocks are full (symmetric); getAsDense(force_symmetry) of the reference mirrors the upper half
    return M


def check(c, nK, HAp_blocks, grad, ref_H, ref_g):
    M = dense_from_blocks(c, HAp_blocks, nK)
    M = np.triu(M) + np.triu(M, 1).T
    assert np.abs(M - ref_H).max() / np.abs(ref_H).max() < 1e-10
    assert np.abs(grad[:nK * P] - ref_g).max() / np.abs(ref_g).max() < 1e-10


@pytest.mark.parametrize("name,nK,nF,vis,seed", CASES, ids=[c[0] for c in CASES])
def test_schur_dense_vs_sparse_oracle(name, nK, nF, vis, seed):
    b, Jp, Jf, g, lam, ref_H, ref_g = build(nK, nF, vis, seed)
    out = _oracle.schur_from_jacobians(b, 0, Jp, Jf, g, lam)
    check(b[0], nK, out["HAp"], out["grad"], ref_H, ref_g)


@pytest.mark.gpu
def test_schur_dense_vs_sparse_on_device_reference_grid():
    for name, nK, nF, vis, seed in CASES:
        b, Jp, Jf, g, lam, ref_H, ref_g = build(nK, nF, vis, seed)
        ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib
        w = lambda what, a: lib.srba_hip_debug_write(ctx.ctx, what, np.ascontiguousarray(a, np.float64).ctypes.data_as(capi.PF64), a.size)
        assert w(1, Jp) == 0 and w(2, Jf) == 0
        assert lib.srba_hip_hessian_from_jacobians(ctx.ctx) == 0
        assert w(6, g) == 0
        lams = np.array([lam]); notpd = np.zeros(1, np.int32)
        assert lib.srba_hip_solve(ctx.ctx, lams.ctypes.data_as(capi.PF64), notpd.ctypes.data_as(capi.PI32)) == 0
        check(b[0], nK, ctx.debug(3), ctx.debug(6), ref_H, ref_g)
        ctx.close()
