"""Multi-GPU layout of the hot path: images are independent units (no BatchNorm, one mask
quantile per image and slice — reference layers/channel_mask.py:142), so a job of G images
shards across ranks with NO data-path collective.  The only communication is the timing /
scalar aggregation below (RCCL when the backend is "nccl", gloo in the CPU tests)."""
from __future__ import annotations

from typing import List, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of rank's images; the first n_items % world ranks get one more."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(n_items, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def max_over_ranks(value: float, device="cpu") -> float:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values: List[float], device="cpu") -> List[float]:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def whole_job_megapixels_per_s(images_per_rank: int, height: int, width: int, steps: int, world: int,
                               max_seconds: float) -> float:
    """value = pixels processed by ALL ranks / slowest rank's time (weak scaling)."""
    return world * images_per_rank * height * width * steps / 1e6 / max_seconds
