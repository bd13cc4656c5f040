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


def all_reduce_gradients(params) -> int:
    """Average the gradients of ``params`` over the ranks with ONE collective on a flat bucket (the fine-tune
    step's only exchange: BASELINE north_star "RCCL all-reduce over xGMI on the gradients only").  Returns the
    number of bytes reduced (0 when not distributed)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
    return flat.numel() * flat.element_size()
