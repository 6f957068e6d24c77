"""Multi-GPU host logic: independent SDPs (beta sweep of experiments/scale.jl:28, the hyperplane
directions of NnSdp.findReach2Dpoly src/NnSdp.jl:73-95, ACAS sub-queries experiments/acas.jl:96-114)
are sharded over ranks with no data-path collective; only the timing/aggregation uses
torch.distributed (RCCL on the GPU box, gloo in the CPU tests)."""
from __future__ import annotations

from typing import List, Sequence


def shard_units(n_units: int, world: int, rank: int) -> List[int]:
    """Contiguous, balanced partition of `n_units` independent SDPs: rank r gets
    units [start, stop) with sizes differing by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(n_units, world)
    start = rank * base + min(rank, rem)
    stop = start + base + (1 if rank < rem else 0)
    return list(range(start, stop))


def bin_pack_by_cost(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time bin packing (cost ~ sum n_k^3 per SDP) for mixed problem sizes."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    bins = [[] for _ in range(world)]
    load = [0.0] * world
    for i in order:
        b = min(range(world), key=lambda j: load[j])
        bins[b].append(i)
        load[b] += costs[i]
    return [sorted(b) for b in bins]


def aggregate_rate(units_done: int, seconds: float, dist=None, device=None) -> float:
    """whole-job throughput: sum of units over ranks / max of seconds over ranks."""
    if dist is None or not dist.is_initialized():
        return units_done / seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_done)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()) / float(t.item())
