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


class _Shared(object):
    """One kind of unknown of a sweep (kf2kf edge poses, landmark positions): who touches / writes what in which round, what is shared between ranks"""
    def __init__(self, off, touch, round_of, owner, world, n_rounds, width, get, put):
        import numpy as np
        self.width, self.get, self.put = width, get, put
        n = len(round_of); self.ids = (touch & 0x7fffffff).astype(np.int64); self.written = (touch >> 31).astype(bool); self.win_of = np.repeat(np.arange(n), np.diff(off))
        self.n_ids = int(self.ids.max()) + 1 if len(self.ids) else 0; self.owner = owner
        first = np.full(self.n_ids, -1, np.int64); self.shared = np.zeros(self.n_ids, bool)   # shared: touched by windows of more than one rank
        for r in range(world):
            e = np.unique(self.ids[owner[self.win_of] == r]); self.shared[e[first[e] >= 0]] = True; first[e[first[e] < 0]] = r
        t_round = round_of[self.win_of]; self.by_round = np.argsort(t_round, kind="stable")   # the touch entries of every round, grouped once
        self.lo = np.searchsorted(t_round[self.by_round], np.arange(n_rounds)); self.hi = np.searchsorted(t_round[self.by_round], np.arange(n_rounds), side="right")
        self.last_writer = np.full(self.n_ids, -1, np.int64)

    def written_in(self, c):
        """(ids written in round c, their writer's rank) -- one writer per id and round: the windows of a round are independent"""
        import numpy as np
        sel = self.by_round[self.lo[c]:self.hi[c]]; sel = sel[self.written[sel]]; w = np.unique(self.ids[sel]); who = np.zeros(self.n_ids, np.int64); who[self.ids[sel]] = self.owner[self.win_of[sel]]
        self.last_writer[w] = who[w]
        return w, who[w]


def sweep_map(eng, roots, win, dist=None, device="cpu", final_sync=True):
    """Re-optimise the local areas of `roots` (key-frame ids, any order) of ONE map, sharded over the ranks of `dist` (None: a single process).

    The schedule -- rounds in order, the windows of a round in any order -- is a sequential schedule of optimize_local_area() calls: windows of a round commute (none writes what
    another touches), so every rank's map after the sweep equals, bit for bit, the map of ONE process running the same rounds (and, to the 1e-6 of the back-ends, the CPU engine's).
    Exchange: an unknown (kf2kf edge pose, landmark position) is SHARED when windows of more than one rank touch it. After round c every rank contributes the values of the shared
    unknowns ITS windows wrote in c (zeros elsewhere) to one all-reduce(sum) of 64-bit patterns: x + 0 + ... + 0 = x exactly -- no rank adds to another's entry, so this is a gather, not an arithmetic
    reduction, and it moves n_shared_written(c) x (3 .. 12) doubles (KB-scale). Unknowns no other rank touches stay local until the final all-reduce (final_sync) that leaves the
    whole map on every rank.
    What is NOT exchanged: the numeric spanning-tree table (TSpanningTree::num). Every optimisation recomputes the poses it reads from the edges (K1) and writes them back; after a
    sharded sweep a rank's table holds the poses ITS windows refreshed and older values elsewhere -- as the reference's own table does for pairs no recent optimisation touched
    (SURVEY App. B-11/12). define_new_keyframe() reads the table only for the INITIAL value of a new edge (RbaEngine.h: determine_kf2kf_edges_to_create), so a map can be continued on
    any rank; the continuations of two ranks agree to the optimiser's tolerance, not bit for bit.
    Returns a dict: rounds, windows run by this rank, shared edges / landmarks, bytes exchanged per round, the KfInfo records of this rank's windows {root: info}."""
    import numpy as np
    rank, world = (0, 1) if dist is None else (dist.get_rank(), dist.get_world_size())
    roots = np.asarray(roots, np.uint64); n = len(roots)
    round_of, off, touch, n_rounds = eng.plan_sweep(roots, win)             # identical on every rank: integer work on identical maps
    lm_off, lm_touch = eng.plan_sweep_lms(n)
    order = np.argsort(roots, kind="stable"); owner = np.zeros(n, np.int64); owner[order] = (np.arange(n) * world) // max(n, 1)   # contiguous ranges of key-frames per rank
    kinds = [_Shared(off, touch, round_of, owner, world, n_rounds, eng.PD, eng.get_edge_poses, eng.set_edge_poses)]
    if len(lm_touch):
        kinds.append(_Shared(lm_off, lm_touch, round_of, owner, world, n_rounds, eng.L, eng.get_lm_positions, eng.set_lm_positions))
    stats = {"rounds": n_rounds, "windows": 0, "shared_edges": int(kinds[0].shared.sum()), "shared_landmarks": int(kinds[1].shared.sum()) if len(kinds) > 1 else 0,
             "exchange_bytes_per_round": [], "info": {}}
    for c in range(n_rounds):
        mine = np.nonzero((round_of == c) & (owner == rank))[0]
        infos = eng.optimize_batch(roots[mine], win)
        for k, i in enumerate(mine):
            stats["info"][int(roots[i])] = infos[k]
        stats["windows"] += len(mine)
        parts = []
        for K in kinds:
            w, who = K.written_in(c); ex = K.shared[w]; parts.append((K, w[ex], who[ex] == rank))      # exchanged now; the same lists on every rank
        stats["exchange_bytes_per_round"].append(int(sum(len(w) * K.width * 8 for K, w, _ in parts)))
        if dist is not None and stats["exchange_bytes_per_round"][-1]:
            _exchange(dist, device, parts)
    if dist is not None and final_sync:
        parts = []
        for K in kinds:
            w = np.nonzero(K.last_writer >= 0)[0]; w = w[~K.shared[w]]; parts.append((K, w, K.last_writer[w] == rank))     # shared unknowns are current everywhere already
        if sum(len(w) for _, w, _ in parts):
            _exchange(dist, device, parts)
    return stats


def _exchange(dist, device, parts):
    """ONE all-reduce(sum) of the values of the unknowns of `parts` [(kind, ids, mine)], each value contributed by exactly one rank (mine: this rank's), zeros by the others; the
    result is set on every rank"""
    import numpy as np
    import torch
    bufs = []
    for K, ids, mine in parts:
        b = np.zeros((len(ids), K.width))
        if mine.any(): b[mine] = K.get(ids[mine])
        bufs.append(b.reshape(-1))
    # the doubles travel as their 64-bit patterns: an integer sum with zeros returns every pattern untouched (a floating-point sum would turn -0.0 into +0.0)
    t = torch.from_numpy(np.concatenate(bufs).view(np.int64)).to(device); dist.all_reduce(t); out = t.cpu().numpy().view(np.float64); pos = 0
    for K, ids, mine in parts:
        v = out[pos:pos + len(ids) * K.width].reshape(len(ids), K.width); pos += len(ids) * K.width
        if (~mine).any(): K.put(ids[~mine], v[~mine])
