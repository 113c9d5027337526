"""Host side of the fused normal-equations kernel of <SE2, RelativePoses2D> (srba_amd/csrc/srba_assemble.hpp), checked WITHOUT a GPU: the packed row records that
srba_hip_upload_problems builds for srba_hip_linearize (one 16-byte record per observation row with Jacobian blocks; srba_hip_debug_assemble_records) against the capsule's own tables:
every Jacobian block appears exactly once, in the row of its residual, with its pose, unknown and direction; every off-diagonal Hessian term of the plan
(sparse_hessian_build_symbolic.h:22-237) is the cross term of exactly one row and points at its block; rows are grouped by their number of blocks and dealt to 16-lane groups so
that rows on the same first unknown meet in a group only when there are more of them than groups. The arithmetic of the kernel is compared with the oracle on the GPU
(tests/test_gpu_parity.py: stepwise / fused linearisation tests)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srba_amd import capi, datasets, runner  # noqa: E402
import _oracle  # noqa: E402


def _records(b, i):
    lib = capi.hip_lib(); cap = 4096
    while True:
        w = np.zeros(4 * cap, np.uint32)
        n = lib.srba_hip_debug_assemble_records(C.cast(C.addressof(b.ptr.contents) + i * C.sizeof(capi.Capsule), capi.PCAP), w.ctypes.data_as(C.POINTER(C.c_uint32)), cap)
        if n < 0:
            cap = -1 - n; continue
        return w[:4 * n].reshape(n, 4).astype(np.int64)


def _decode(w):
    m = (w[:, 0] >> 28) & 3
    D = np.stack([w[:, 0] & 0x3fff, (w[:, 0] >> 14) & 0x3fff, w[:, 1] & 0x3fff], 1) - 1
    row = (w[:, 1] >> 14) & 0x7ff
    col = np.stack([w[:, 1] >> 25, w[:, 2] & 0x7f, (w[:, 2] >> 7) & 0x7f], 1)
    xb = np.stack([(w[:, 2] >> 14) & 0x7ff, w[:, 3] & 0x7ff, (w[:, 3] >> 11) & 0x7ff], 1)
    inv = np.stack([(w[:, 2] >> 25) & 1, (w[:, 2] >> 26) & 1, (w[:, 2] >> 27) & 1], 1)
    return m, D, row, col, xb, inv


def test_records_cover_the_jacobian_and_hessian_plan_of_every_capsule():
    b = runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=260, seed=1, path="tour"), backend=_oracle.BACKEND, submap=10, depth=3)
    assert b.n >= 200
    checked = 0
    for i in range(0, b.n, 7):
        k = b[i]; w = _records(b, i)
        assert len(w) > 0 and len(w) % 16 == 0, i
        m, D, row, col, xb, inv = _decode(w)
        bp_res = b.array(i, "bp_res", np.int32, k.n_bp); bp_col = b.array(i, "bp_col", np.int32, k.n_bp); bp_D = b.array(i, "bp_D", np.int32, k.n_bp)
        bp_normal = b.array(i, "bp_normal", np.uint8, k.n_bp)
        # blocks: a row's record lists the blocks of that residual row, ascending unknown
        seen = 0
        for r in np.nonzero(m)[0]:
            blocks = np.nonzero(bp_res == row[r])[0]
            assert len(blocks) == m[r] and np.all(np.diff(bp_col[blocks]) > 0), (i, r)
            assert np.array_equal(col[r, :m[r]], bp_col[blocks]) and np.array_equal(D[r, :m[r]], bp_D[blocks]) and np.array_equal(inv[r, :m[r]], 1 - (bp_normal[blocks] != 0)), (i, r)
            seen += m[r]
        assert seen == k.n_bp and len(np.unique(row[m > 0])) == int((m > 0).sum())   # every block once, every row once
        # cross terms: slot 0 = blocks (0, 1), 1 = (0, 2), 2 = (1, 2); 0x7ff = no such term in the plan
        hap_i = b.array(i, "hap_i", np.int32, k.n_hap); hap_j = b.array(i, "hap_j", np.int32, k.n_hap); off = b.array(i, "hap_term_off", np.int32, k.n_hap + 1)
        t1 = b.array(i, "hap_t1", np.int32, k.n_hap_terms); t2 = b.array(i, "hap_t2", np.int32, k.n_hap_terms)
        want = {}
        for h in range(k.n_hap):
            if hap_i[h] != hap_j[h]:
                for t in range(off[h], off[h + 1]):
                    assert bp_res[t1[t]] == bp_res[t2[t]]
                    want[(int(bp_res[t1[t]]), int(bp_col[t1[t]]), int(bp_col[t2[t]]))] = h
        got = {}
        for r in np.nonzero(m >= 2)[0]:
            for s, (a, c) in enumerate(((0, 1), (0, 2), (1, 2))):
                if c < m[r] and xb[r, s] != 0x7ff:
                    got[(int(row[r]), int(col[r, a]), int(col[r, c]))] = int(xb[r, s])
        assert got == want, i
        # layout: rows with three blocks first, then two, then one, each class in 16-lane groups of its own; empty records only as padding of a class
        cls = m.reshape(-1, 16)
        assert all(len(set(g[g > 0])) <= 1 for g in cls) and np.all(np.diff([g.max() for g in cls]) <= 0), i
        # conflict avoidance: rows of one first unknown are spread over the groups of their class (no group holds more than ceil(count / groups) + 1 of them)
        for mc in (3, 2, 1):
            groups = [g for g in range(len(cls)) if cls[g].max() == mc]
            if not groups: continue
            c0 = col[:, 0].reshape(-1, 16)
            for u in np.unique(col[m == mc, 0]):
                per_group = [int(((c0[g] == u) & (cls[g] == mc)).sum()) for g in groups]
                assert max(per_group) <= -(-sum(per_group) // len(groups)) + 1, (i, mc, u, per_group)
        checked += 1
    assert checked >= 25


def test_a_capsule_beyond_the_record_fields_is_left_to_the_unfused_kernel():
    b = runner.harvest_graph_slam(datasets.graph_slam_se2(n_kf=60, seed=1, path="tour"), backend=_oracle.BACKEND, submap=10, depth=3)
    c = b.clone(b.n - 1, 1); lib = capi.hip_lib(); w = np.zeros(4 * 8192, np.uint32)
    assert lib.srba_hip_debug_assemble_records(c.ptr, w.ctypes.data_as(C.POINTER(C.c_uint32)), 8192) > 0
    c.ptr[0].n_obs = 5000   # more residual rows than the 11-bit field holds (the arrays are not touched: the size check comes first)
    assert lib.srba_hip_debug_assemble_records(c.ptr, w.ctypes.data_as(C.POINTER(C.c_uint32)), 8192) in (0, -1 - (16 * ((min(5000, c.ptr[0].n_bp) + 15) // 16) + 32))
