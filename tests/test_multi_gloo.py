"""N>1 path of the benchmark on CPU: two processes over gloo, each owning an independent map (replicas, no collective on the
data path). Checks the replica seeds, the barrier-bracketed timing and the sum/max aggregation of srba_amd/multi.py, with the capsules of
each rank solved by the CPU oracle (test infrastructure) in place of the GPU."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    import torch  # noqa: F401
    from srba_amd import datasets, multi, runner
    import _oracle  # tests/_oracle.py: the CPU checker
    dist = multi.init_process_group("gloo")
    assert dist is not None and dist.get_world_size() == world and multi.rank_info() == (rank, world, rank)
    ds = datasets.graph_slam_se2(n_kf=120, seed=multi.replica_seed(rank), path="tour")
    batch = runner.harvest_graph_slam(ds, backend=_oracle.BACKEND, submap=10, depth=3)
    res = _oracle.run_batch(batch)
    trials = int(res["num_trials"].sum()); obs = int((res["num_trials"] * res["num_observations"]).sum())
    calls = []

    def step():
        calls.append(1); time.sleep(0.02 * (1 + rank))   # rank 1 is the slow one

    elapsed = multi.timed_region(dist, lambda: None, step, 3)
    tot, tot_obs, mx = multi.aggregate(dist, "cpu", trials, obs, elapsed)
    q.put((rank, trials, obs, elapsed, tot, tot_obs, mx, len(calls), float(res["total_sqr_error_final"].sum()), batch.n))
    dist.barrier(); dist.destroy_process_group()


def test_two_replicas_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    out = sorted(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(timeout=60); assert p.exitcode == 0
    (r0, t0, o0, e0, tot0, tobs0, mx0, c0, chi0, n0), (r1, t1, o1, e1, tot1, tobs1, mx1, c1, chi1, n1) = out
    assert (r0, r1) == (0, 1) and c0 == c1 == 3                 # exactly K steps on every rank
    assert tot0 == tot1 == t0 + t1 and tobs0 == tobs1 == o0 + o1  # whole-job units = sum over ranks (every rank sees the same total)
    assert mx0 == mx1 and abs(mx0 - max(e0, e1)) < 1e-9          # elapsed = max over ranks
    assert e1 >= 0.11 and e0 >= 0.11 - 0.02                      # the closing barrier makes the fast rank wait for the slow one
    assert n0 == n1 and chi0 != chi1                             # same workload size, different maps (independent replicas)


def test_single_process_passthrough():
    from srba_amd import multi
    assert multi.aggregate(None, "cpu", 10, 20, 0.5) == (10, 20, 0.5)
    assert multi.replica_seed(0) == 1 and multi.replica_seed(3) == 4
