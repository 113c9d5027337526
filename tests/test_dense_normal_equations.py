"""The symbolic Hessian / gradient plan (CapsuleData::build_plan = sparse_hessian_build_symbolic.h:22-237 + compute_minus_gradient.h:20-91) checked against the
DENSE normal equations for every model family: from the Jacobian blocks and the block tables alone (bp_col/bp_res, bf_col/bf_res) numpy builds the dense
J (rows = residual rows of the capsule), forms  J^t W J  and  J^t W r  (W = Lambda for the constant-matrix policy; identity and the reference's 1/sigma
scaling of H and g for the identity policy, srba_options_noise.h:52-56,67-71) and compares every stored block of HAp / Hf / HApf and the gradient.
A block that the plan dropped, or a term missing from a block's list, is visible here: dense entries outside the planned blocks must be zero.
(The plan is shared by the oracle and the device, so GPU-vs-oracle parity alone could not detect such an error.)
CPU: the oracle's blocks; GPU: the device's blocks read back through srba_hip_debug_read."""
import glob
import os

import numpy as np
import pytest

from srba_amd import capi, runner
import _oracle  # tests/_oracle.py: the CPU checker

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.caps")))


def arr(c, name, n):
    p = getattr(c, name)
    return np.ctypeslib.as_array(p, shape=(n,)).copy() if (p and n) else np.zeros(0, np.int32)


def dense_check(b, i, a):
    """a: dict(Jp, Jf, resid, HAp, Hf, HApf, grad) of capsule i"""
    c = b[i]; P, L, O, PD = capi.DIMS[b.family]; nK, nF = c.n_unk_edges, c.n_unk_lms
    Jp = a["Jp"].reshape(-1, O, P); Jf = a["Jf"].reshape(-1, O, L); r = a["resid"].reshape(-1)
    A = np.zeros((c.n_obs * O, nK * P)); F = np.zeros((c.n_obs * O, nF * L))
    bp_col, bp_res, bf_col, bf_res = arr(c, "bp_col", c.n_bp), arr(c, "bp_res", c.n_bp), arr(c, "bf_col", c.n_bf), arr(c, "bf_res", c.n_bf)
    for k in range(c.n_bp):
        A[bp_res[k] * O:(bp_res[k] + 1) * O, bp_col[k] * P:(bp_col[k] + 1) * P] += Jp[k]
    for k in range(c.n_bf):
        F[bf_res[k] * O:(bf_res[k] + 1) * O, bf_col[k] * L:(bf_col[k] + 1) * L] += Jf[k]
    prm = b.params
    if prm.noise == capi.NOISE_MATRIX:
        W = np.kron(np.eye(c.n_obs), np.array(list(prm.lambda_)[:O * O]).reshape(O, O)); sc = 1.0
    else:
        W = np.eye(c.n_obs * O); sc = 1.0 / prm.std_noise_observations
    HAp, Hf, HApf = sc * A.T @ W @ A, sc * F.T @ W @ F, sc * A.T @ W @ F
    g = sc * np.concatenate([A.T @ W @ r, F.T @ W @ r])
    def blocks_vs_dense(M, blocks, bi, bj, R, C, upper):
        seen = np.zeros(M.shape, bool); scale = max(np.abs(M).max(), 1e-300)
        for k in range(len(bi)):
            sl = (slice(bi[k] * R, (bi[k] + 1) * R), slice(bj[k] * C, (bj[k] + 1) * C))
            assert np.abs(blocks[k * R * C:(k + 1) * R * C].reshape(R, C) - M[sl]).max() <= 1e-9 * scale, (i, k)
            seen[sl] = True
            if upper:
                seen[slice(bj[k] * C, (bj[k] + 1) * C), slice(bi[k] * R, (bi[k] + 1) * R)] = True
        assert np.abs(M[~seen]).max(initial=0.0) <= 1e-12 * scale, "a non-zero block of the dense matrix is missing from the plan (capsule %d)" % i
    blocks_vs_dense(HAp, a["HAp"], arr(c, "hap_i", c.n_hap), arr(c, "hap_j", c.n_hap), P, P, True)
    if nF:
        blocks_vs_dense(Hf, a["Hf"], arr(c, "hf_i", c.n_hf), arr(c, "hf_j", c.n_hf), L, L, True)
        blocks_vs_dense(HApf, a["HApf"], arr(c, "hapf_i", c.n_hapf), arr(c, "hapf_j", c.n_hapf), P, L, False)
    assert np.abs(a["grad"] - g).max() <= 1e-9 * max(np.abs(g).max(), 1e-300) + 1e-18, i


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-5] for p in GOLD])
def test_plan_equals_dense_normal_equations_oracle(path):
    b = runner.CapsuleBatch.load(path)
    for i in range(b.n):
        dense_check(b, i, _oracle.stage(b, i))


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-5] for p in GOLD])
def test_plan_equals_dense_normal_equations_device(path):
    b = runner.CapsuleBatch.load(path)
    ctx = runner.HipContext(b.params); ctx.upload(b); lib = ctx.lib
    assert lib.srba_hip_update_spantree(ctx.ctx, 0) == 0 and lib.srba_hip_eval_residuals(ctx.ctx, None) == 0 and lib.srba_hip_linearize(ctx.ctx) == 0
    res, Jp, Jf, HAp, Hf, HApf, grad = (ctx.debug(k) for k in (0, 1, 2, 3, 4, 5, 6))
    P, L, O, PD = capi.DIMS[b.family]; o = dict(res=0, Jp=0, Jf=0, HAp=0, Hf=0, HApf=0, g=0)
    for i in range(b.n):
        c = b[i]; n = P * c.n_unk_edges + L * c.n_unk_lms
        cut = lambda v, key, cnt: v[o[key]:o[key] + cnt]
        dense_check(b, i, dict(resid=cut(res, "res", c.n_obs * O), Jp=cut(Jp, "Jp", c.n_bp * O * P), Jf=cut(Jf, "Jf", c.n_bf * O * L), HAp=cut(HAp, "HAp", c.n_hap * P * P),
                               Hf=cut(Hf, "Hf", c.n_hf * L * L), HApf=cut(HApf, "HApf", c.n_hapf * P * L), grad=cut(grad, "g", n)))
        o["res"] += c.n_obs * O; o["Jp"] += c.n_bp * O * P; o["Jf"] += c.n_bf * O * L; o["HAp"] += c.n_hap * P * P; o["Hf"] += c.n_hf * L * L; o["HApf"] += c.n_hapf * P * L; o["g"] += n
    ctx.close()
