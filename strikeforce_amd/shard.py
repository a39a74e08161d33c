"""Multi-GPU sharding of the arena batch: contiguous arena ranges per rank, no data-path
collective; the only exchange is the all-gather of the end-of-episode result records."""
import ctypes as C


def shard_seeds(workload, rank, base_tb=1_700_000_000, serial=123_456_789):
    """Seeds of rank `rank`'s arenas: global arena id g = rank * arenas + i gets tb = base + g."""
    return workload.seeds(base_tb=base_tb, serial=serial, first_arena=rank * workload.cfg.arenas)


def command_seed(workload, rank, seed0=12345):
    """First LCG seed of the random-action agent for this rank's (arena, agent) pairs."""
    return seed0 + rank * workload.cfg.arenas * workload.cfg.n_agents


def gather_results(local, world):
    """All-gather of the [arenas][agents][8] int32 result records (torch tensor on the rank's device).
    Returns a [world * arenas][agents][8] tensor; RCCL over xGMI under backend 'nccl', gloo on CPU."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    flat = local.contiguous().reshape(-1)
    out = torch.empty(world * flat.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, flat)
    return out.reshape((world * local.shape[0],) + tuple(local.shape[1:]))
