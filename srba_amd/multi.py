"""Multi-GPU driver logic.
1. The benchmark: the hot path shards by independent maps -- rank r owns the map generated with seed 1+r and optimises its own capsules; there is NO data-path collective.
   torch.distributed (RCCL on the GPU box, gloo in the CPU tests) is used only for the barrier that brackets the timed region and for the sum / max that turn per-rank counts
   into the job total.
2. ONE map over several GPUs (sweep_map; north_star: "sub-maps shard across the GPUs of one node with RCCL only for shared-edge reduction"; no counterpart in the reference): every
   rank holds the whole map (the host graph layer is integer work), the local areas to re-optimise are dealt to rounds of mutually independent windows by the engine
   (RbaEngine<>::plan_local_area_sweep), the roots of a round are dealt to the ranks by contiguous ranges of key-frames (a rank's windows are neighbours: sub-maps), every rank
   runs its share as ONE batch on its GPU, and the kf2kf edges a round wrote that another rank's later window touches are exchanged with ONE all-reduce per round."""
import os
import time


def rank_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend, force=False):
    """one process per GPU; rendezvous on 127.0.0.1 unless the launcher says otherwise. force: create the group for a single rank as well (tests/test_rccl_single_rank.py
    runs the nccl = RCCL branch -- init, barrier, the two all-reduces of aggregate() on device tensors -- on the one GPU of the test box)"""
    rank, world, _ = rank_info()
    if world <= 1 and not force:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def replica_seed(rank):
    """every rank optimises a different map of the same size (weak scaling)"""
    return 1 + rank


def timed_region(dist, device_sync, step, steps):
    """barrier + device sync on both sides of exactly `steps` calls of step(); returns this rank's elapsed seconds"""
    device_sync()
    if dist is not None:
        dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    if dist is not None:
        dist.barrier()
    device_sync()
    return time.perf_counter() - t0


def aggregate(dist, device, units_per_step, obs_units_per_step, elapsed):
    """whole-job totals: units summed over ranks, elapsed = max over ranks"""
    if dist is None:
        return int(units_per_step), int(obs_units_per_step), float(elapsed)
    import torch
    t = torch.tensor([float(units_per_step), float(obs_units_per_step)], device=device, dtype=torch.float64); dist.all_reduce(t)
    m = torch.tensor([float(elapsed)], device=device, dtype=torch.float64); dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return int(t[0].item()), int(t[1].item()), float(m[0].item())


def max_over_ranks(dist, device, values):
    """element-wise maximum over the ranks of a short list of floats (set-up times in the N > 1 line: the ranks of one node share the host cores); every rank must call it"""
    if dist is None:
        return [float(v) for v in values]
    import torch
    t = torch.tensor([float(v) for v in values], device=device, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.tolist()]


def sweep_map(eng, roots, win, dist=None, device="cpu", final_sync=True):
    """Re-optimise the local areas of `roots` (key-frame ids, any order) of ONE map, sharded over the ranks of `dist` (None: a single process).

    The schedule -- rounds in order, the windows of a round in any order -- is a sequential schedule of optimize_local_area() calls: windows of a round commute (none writes what
    another touches), so every rank's map after the sweep equals, bit for bit, the map of ONE process running the same rounds (and, to the 1e-6 of the back-ends, the CPU engine's).
    Exchange: an edge is SHARED when windows of more than one rank touch it. After round c every rank contributes the values of the shared edges ITS windows wrote in c (zeros
    elsewhere) to one all-reduce(sum): x + 0 + ... + 0 = x exactly -- no rank adds to another's entry, so this is a gather, not an arithmetic reduction, and it moves
    n_shared_written(c) x P doubles (KB-scale). Edges no other rank touches stay local until the final all-reduce (final_sync) that leaves the whole map on every rank.
    Returns a dict: rounds, windows run by this rank, shared edges, bytes exchanged per round, the KfInfo records of this rank's windows {root: info}."""
    import numpy as np
    rank, world = (0, 1) if dist is None else (dist.get_rank(), dist.get_world_size())
    roots = np.asarray(roots, np.uint64); n = len(roots)
    round_of, off, touch, n_rounds = eng.plan_sweep(roots, win)             # identical on every rank: integer work on identical maps
    order = np.argsort(roots, kind="stable"); owner = np.zeros(n, np.int64); owner[order] = (np.arange(n) * world) // max(n, 1)   # contiguous ranges of key-frames per rank
    edge = (touch & 0x7fffffff).astype(np.int64); written = (touch >> 31).astype(bool); win_of = np.repeat(np.arange(n), np.diff(off))
    n_edges = int(edge.max()) + 1 if len(edge) else 0
    # shared edges: touched by windows of more than one rank
    first = np.full(n_edges, -1, np.int64); shared = np.zeros(n_edges, bool)
    for r in range(world):
        e = np.unique(edge[owner[win_of] == r]); shared[e[first[e] >= 0]] = True; first[e[first[e] < 0]] = r
    # per round: the shared edges written in it (by whom: known to everybody) ; last writer of every written edge (final sync)
    stats = {"rounds": n_rounds, "windows": 0, "shared_edges": int(shared.sum()), "exchange_bytes_per_round": [], "info": {}}
    last_writer = np.full(n_edges, -1, np.int64)
    t_round = round_of[win_of]; by_round = np.argsort(t_round, kind="stable"); r_lo = np.searchsorted(t_round[by_round], np.arange(n_rounds)); r_hi = np.searchsorted(t_round[by_round],
            np.arange(n_rounds), side="right")                                       # the touch entries of every round, grouped once
    for c in range(n_rounds):
        mine = np.nonzero((round_of == c) & (owner == rank))[0]
        infos = eng.optimize_batch(roots[mine], win)
        for k, i in enumerate(mine):
            stats["info"][int(roots[i])] = infos[k]
        stats["windows"] += len(mine)
        sel = by_round[r_lo[c]:r_hi[c]]; sel = sel[written[sel]]; w_edges = np.unique(edge[sel]); w_owner = np.zeros(n_edges, np.int64)
        w_owner[edge[sel]] = owner[win_of[sel]]                                         # (one writer per edge and round: the windows of a round are independent)
        last_writer[w_edges] = w_owner[w_edges]
        ex = w_edges[shared[w_edges]]                                                     # exchanged now; the same list on every rank
        stats["exchange_bytes_per_round"].append(int(len(ex) * eng.PD * 8))
        if dist is not None and len(ex):
            _exchange(eng, dist, device, ex, w_owner[ex] == rank)
    if dist is not None and final_sync:
        w = np.nonzero(last_writer >= 0)[0]; w = w[~shared[w]]                           # shared edges are current everywhere already
        if len(w):
            _exchange(eng, dist, device, w, last_writer[w] == rank)
    return stats


def _exchange(eng, dist, device, ids, mine):
    """all-reduce(sum) of the values of the edges `ids`, each contributed by exactly one rank (mine: this rank's), zeros by the others; the result is set on every rank"""
    import numpy as np
    import torch
    buf = np.zeros((len(ids), eng.PD)); buf[mine] = eng.get_edge_poses(ids[mine])
    t = torch.from_numpy(buf).to(device); dist.all_reduce(t); out = t.cpu().numpy()
    eng.set_edge_poses(ids[~mine], out[~mine])
