"""Multi-GPU driver logic of the benchmark: the hot path shards by independent maps -- rank r owns the map generated with
seed 1+r and optimises its own capsules; there is NO data-path collective.  torch.distributed (RCCL on the GPU box, gloo in the CPU
tests) is used only for the barrier that brackets the timed region and for the sum / max that turn per-rank counts into the job total."""
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
