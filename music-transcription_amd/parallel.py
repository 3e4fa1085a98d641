"""Multi-GPU sharding of the hot path: one process per GPU, chunks / recordings are
independent (main.py:258-266 keeps no cross-chunk state), so inference shards with NO data-path
collective; only per-recording results (a few floats) are gathered at the end."""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced split of range(n_items): the first n % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def lpt_assign(durations: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of recordings to ranks so that ranks finish together.
    Deterministic (ties by index), identical on every rank."""
    order = sorted(range(len(durations)), key=lambda i: (-float(durations[i]), i))
    loads = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += float(durations[i])
    return out


def gather_values(local_idx: Sequence[int], local_val: Sequence[float], n_total: int, device=None) -> List[float]:
    """Every rank contributes (index, value) pairs for its shard; returns the full list on every rank.
    One small all_reduce (SUM of a zero-initialised vector): works with RCCL ("nccl") and gloo."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = [0.0] * n_total
        for i, v in zip(local_idx, local_val):
            out[i] = float(v)
        return out
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    buf = torch.zeros(n_total, dtype=torch.float64, device=device)
    if len(local_idx):
        buf[torch.as_tensor(list(local_idx), device=device)] = torch.as_tensor(list(local_val), dtype=torch.float64, device=device)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf.cpu().tolist()
