"""ONE map over several processes / GPUs (srba_amd/multi.py sweep_map; RbaEngine<>::plan_local_area_sweep, optimize_local_areas_batch): north_star's "sub-maps shard across the GPUs with
RCCL only for shared-edge reduction". The reference has no such mode: parity is defined against the SAME schedule run as sequential optimize_local_area() calls --
rounds in order, the windows of a round in any order (they commute: none writes what another touches).

CPU tier (oracle as numeric back-end, gloo): the plan's rounds are independent; a one-process sweep equals the sequential schedule bit for bit; two ranks over gloo end with the
map of the one-process sweep, bit for bit, on BOTH ranks, and the per-round exchange moves only shared edges. GPU tier: the same sweep with the HIP back-end (one batch per
round) against the oracle's sequential schedule at 1e-6; the exchange over a single-rank nccl (= RCCL) group on device tensors."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srba_amd import datasets, multi, runner  # noqa: E402
import _oracle  # noqa: E402

N_KF, WIN = 400, 3


def _map(backend, n_kf=N_KF):
    eng = runner.graph_slam_engine(backend=backend, submap=10, depth=3, harvest=0); eng.run(datasets.graph_slam_se2(n_kf=n_kf, seed=1, path="tour")); return eng


def _roots(n_kf=N_KF):
    return np.arange(3, n_kf, 2, dtype=np.uint64)


def _sequential(eng, roots, round_of, n_rounds):
    for c in range(n_rounds):
        for i in np.nonzero(round_of == c)[0]:
            eng.optimize_local_area(roots[i], WIN)


def test_rounds_of_a_plan_are_independent():
    eng = _map(_oracle.BACKEND); roots = _roots()
    round_of, off, touch, n_rounds = eng.plan_sweep(roots, WIN)
    assert n_rounds >= 2 and (round_of >= 0).all() and round_of.max() == n_rounds - 1
    edge = (touch & 0x7fffffff).astype(np.int64); wr = (touch >> 31).astype(bool)
    for c in range(n_rounds):
        state = {}   # edge -> 1 read / 2 written by a member of the round
        for i in np.nonzero(round_of == c)[0]:
            e, w = edge[off[i]:off[i + 1]], wr[off[i]:off[i + 1]]
            assert w.any() and len(np.unique(e)) == len(e)
            for ee, ww in zip(e, w):
                assert not (ww and ee in state) and not ((not ww) and state.get(ee) == 2), (c, i, ee)
            for ee, ww in zip(e, w):
                state[ee] = max(state.get(ee, 0), 2 if ww else 1)
    # first fit: a window sits in the first round it does not clash with -- so it clashes with some member of every earlier round
    i = int(np.nonzero(round_of == n_rounds - 1)[0][0]); e, w = edge[off[i]:off[i + 1]], wr[off[i]:off[i + 1]]
    for c in range(n_rounds - 1):
        clash = False
        for j in np.nonzero(round_of == c)[0]:
            ej, wj = edge[off[j]:off[j + 1]], wr[off[j]:off[j + 1]]
            clash |= bool(np.intersect1d(e[w], ej).size or np.intersect1d(e, ej[wj]).size)
        assert clash, c
    eng.close()


def test_batch_of_dependent_windows_is_refused():
    eng = _map(_oracle.BACKEND, 120)
    with pytest.raises(RuntimeError, match="not independent"):
        eng.optimize_batch(np.array([50, 51], np.uint64), WIN)
    eng.close()


def test_one_process_sweep_equals_the_sequential_schedule():
    e1 = _map(_oracle.BACKEND); e2 = _map(_oracle.BACKEND); roots = _roots()
    round_of, _, _, n_rounds = e1.plan_sweep(roots, WIN)
    before = e1.edges()[2].copy()
    st = multi.sweep_map(e1, roots, WIN)
    _sequential(e2, roots, round_of, n_rounds)
    p1, p2 = e1.edges()[2], e2.edges()[2]
    assert st["windows"] == len(roots) and st["rounds"] == n_rounds and len(st["info"]) == len(roots)
    assert np.array_equal(p1, p2) and not np.array_equal(p1, before)
    assert e1.eval_overall_squared_error() == e2.eval_overall_squared_error()
    e1.close(); e2.close()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch  # noqa: F401
    dist = multi.init_process_group("gloo")
    eng = _map(_oracle.BACKEND); roots = _roots()
    st = multi.sweep_map(eng, roots, WIN, dist=dist, device="cpu")
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), poses=eng.edges()[2], windows=st["windows"], shared=st["shared_edges"], bytes=np.array(st["exchange_bytes_per_round"]),
             chi2=eng.eval_overall_squared_error())
    dist.barrier(); dist.destroy_process_group()


def test_two_ranks_over_gloo_end_with_the_map_of_one_process(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in ps: p.start()
    for p in ps:
        p.join(timeout=600); assert p.exitcode == 0
    ref = _map(_oracle.BACKEND); roots = _roots(); st = multi.sweep_map(ref, roots, WIN); want = ref.edges()[2]
    r0, r1 = np.load(os.path.join(str(tmp_path), "rank0.npz")), np.load(os.path.join(str(tmp_path), "rank1.npz"))
    assert np.array_equal(r0["poses"], want) and np.array_equal(r1["poses"], want)            # bit for bit, on both ranks
    assert int(r0["windows"]) + int(r1["windows"]) == len(roots) and min(int(r0["windows"]), int(r1["windows"])) >= len(roots) // 2 - 1
    assert float(r0["chi2"]) == float(r1["chi2"]) == ref.eval_overall_squared_error()
    # only the boundary between the two shards is exchanged: the edges windows of BOTH ranks touch (a window spans ~80 key-frames of this map: a fifth of its edges), and per round
    # at most what the one or two windows next to the boundary wrote -- a quarter of all the values the sweep writes
    assert 0 < int(r0["shared"]) == int(r1["shared"]) <= 0.25 * len(want)
    written = sum(int(st["info"][int(r)].num_k2k) for r in roots) * want.shape[1] * 8
    assert np.array_equal(r0["bytes"], r1["bytes"]) and 0 < r0["bytes"].sum() < 0.25 * written and r0["bytes"].max() <= 2 * 64 * want.shape[1] * 8
    ref.close()


def _lm_map():
    from srba_amd import capi
    ds, _ = datasets.landmarks_dataset_se2("rb2d", n_kf=36, n_lm=1080, seed=7, noise=1e-3)
    eng = runner.landmark_engine("rb2d", backend=_oracle.BACKEND, solver=capi.SOLVER_SCHUR_DENSE, harvest=0); eng.run(ds); return eng


def _lm_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch  # noqa: F401
    dist = multi.init_process_group("gloo")
    eng = _lm_map(); st = multi.sweep_map(eng, np.arange(2, 36, dtype=np.uint64), WIN, dist=dist, device="cpu")
    np.savez(os.path.join(out_dir, "lm_rank%d.npz" % rank), poses=eng.edges()[2], lms=eng.unknown_lms()[2], shared=st["shared_landmarks"], nbytes=np.array(st["exchange_bytes_per_round"]))
    dist.barrier(); dist.destroy_process_group()


def test_landmark_map_over_two_ranks(tmp_path):
    """windows with unknown landmarks: the exchange carries landmark positions beside the edge poses (range-bearing 2D, Schur solver); both ranks end with the one-process map"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); port = _free_port()
    ps = [ctx.Process(target=_lm_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in ps: p.start()
    for p in ps:
        p.join(timeout=900); assert p.exitcode == 0
    ref = _lm_map(); roots = np.arange(2, 36, dtype=np.uint64)
    round_of, _, _, n_rounds = ref.plan_sweep(roots, WIN); seq = _lm_map(); _sequential(seq, roots, round_of, n_rounds)
    st = multi.sweep_map(ref, roots, WIN)
    assert np.array_equal(ref.edges()[2], seq.edges()[2]) and np.array_equal(ref.unknown_lms()[2], seq.unknown_lms()[2])
    assert sum(int(i.lm.num_trials) for i in st["info"].values()) > 0 and max(int(i.num_k2f) for i in st["info"].values()) > 0     # windows with unknown landmarks did run
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), "lm_rank%d.npz" % r))
        assert np.array_equal(d["poses"], ref.edges()[2]) and np.array_equal(d["lms"], ref.unknown_lms()[2]), r
        assert int(d["shared"]) > 0 and d["nbytes"].sum() > 0
    assert st["shared_landmarks"] == 0
    ref.close(); seq.close()


@pytest.mark.gpu
def test_gpu_sweep_matches_the_oracle_schedule():
    """the HIP back-end runs every round as ONE batch (upload, fused launch per size class, read-back); the oracle runs the same rounds window by window"""
    g = _map("hip"); o = _map(_oracle.BACKEND); roots = _roots()
    assert np.allclose(g.edges()[2], o.edges()[2], rtol=1e-6, atol=1e-9)
    round_of, _, _, n_rounds = g.plan_sweep(roots, WIN)
    st = multi.sweep_map(g, roots, WIN)
    _sequential(o, roots, round_of, n_rounds)
    assert st["windows"] == len(roots)
    assert np.allclose(g.edges()[2], o.edges()[2], rtol=1e-6, atol=1e-8)
    cg, co = g.eval_overall_squared_error(), o.eval_overall_squared_error()
    assert abs(cg - co) <= 1e-6 * max(cg, co) + 1e-18
    chi_g = np.array([st["info"][int(r)].chi2_final for r in roots]); assert np.isfinite(chi_g).all()
    g.close(); o.close()


@pytest.mark.gpu
def test_exchange_over_a_single_rank_rccl_group():
    """the per-round all-reduce of sweep_map on DEVICE tensors over nccl (= RCCL): with one rank every edge is its own, the sweep must leave the map of the plain sweep"""
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
from srba_amd import datasets, multi, runner
torch.cuda.set_device(0)
dist = multi.init_process_group("nccl", force=True)
def build():
    e = runner.graph_slam_engine(backend="hip", submap=10, depth=3, harvest=0); e.run(datasets.graph_slam_se2(n_kf=200, seed=1, path="tour")); return e
a, b = build(), build(); roots = np.arange(3, 200, 2, dtype=np.uint64)
multi.sweep_map(a, roots, 3, dist=dist, device="cuda"); multi.sweep_map(b, roots, 3)
assert np.array_equal(a.edges()[2], b.edges()[2])
ids = np.arange(10, dtype=np.int64); before = a.get_edge_poses(ids).copy()
class K: width = a.PD; get = staticmethod(a.get_edge_poses); put = staticmethod(a.set_edge_poses)
multi._exchange(dist, "cuda", [(K, ids, np.ones(10, bool))]); assert np.array_equal(a.get_edge_poses(ids), before)
dist.barrier(); dist.destroy_process_group(); print("sweep-rccl-ok")
''' % (ROOT, ROOT)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "sweep-rccl-ok" in p.stdout, p.stderr[-3000:]
