"""Multi-GPU layout of the hot path: images are independent units (no BatchNorm, one mask
quantile per image and slice — reference layers/channel_mask.py:142), so a job of G images
shards across ranks with NO data-path collective.  The only communication is the timing /
scalar aggregation below (RCCL when the backend is "nccl", gloo in the CPU tests)."""
from __future__ import annotations

import os
from typing import List, Tuple

import torch


def collectives_active() -> bool:
    """True when the exchange steps must issue their collectives: a process group exists and either the job has more
    than one rank or ``VAMPIC_FORCE_COLLECTIVES=1`` asks for them at world size 1 (a 1-rank ``nccl`` group on a single
    GPU runs communicator init, the comm-stream / event ordering against the backward's graph segments, ``work.wait()``
    and the division exactly as an N-rank job does; sums over one rank are the identity, so results do not change)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("VAMPIC_FORCE_COLLECTIVES", "0") == "1"


def init_single_rank_group(device=None, backend: str = "nccl") -> bool:
    """``VAMPIC_FORCE_COLLECTIVES=1`` without a launcher: make this process a 1-rank process group (RCCL on ``device``)
    so that the exchange steps run their collectives.  Returns True when a group was created."""
    import torch.distributed as dist
    if os.environ.get("VAMPIC_FORCE_COLLECTIVES", "0") != "1" or not dist.is_available() or dist.is_initialized():
        return False
    import socket
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = torch.device(device)
    dist.init_process_group(backend, rank=0, world_size=1, **kw)
    return True


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of rank's images; the first n_items % world ranks get one more."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(n_items, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def max_over_ranks(value: float, device="cpu") -> float:
    import torch.distributed as dist
    if not collectives_active():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values: List[float], device="cpu") -> List[float]:
    import torch.distributed as dist
    if not collectives_active():
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
    step's only exchange: BASELINE north_star "RCCL all-reduce over xGMI on the gradients only").

    The bucket layout is RANK-INVARIANT: it spans every parameter handed in (the caller passes the
    ``requires_grad`` set, which is the same on all ranks), with zeros where this rank produced no gradient, plus
    one presence word per parameter.  So every rank always enters the collective, with the same element count,
    whatever subset of the parameters its step touched — a rank whose step used another REM (or none) can neither
    shift another rank's gradients onto the wrong parameters nor leave the others waiting in the all-reduce.
    A parameter that received a gradient on no rank keeps ``grad = None`` (the optimiser then skips it, as in a
    single-process run).  Returns the number of bytes reduced (0 when not distributed)."""
    import torch.distributed as dist
    if not collectives_active():
        return 0
    params = list(params)
    if not params:
        return 0
    dev, dt = params[0].device, params[0].dtype
    n_el = sum(p.numel() for p in params)
    # (round 4, first run on a device: the per-parameter copy_ / scalar stores of the first version cost 12.9 ms for the 420
    # REM tensors — 840 tiny launches; now one multi-tensor copy in, one out, and the presence words travel as one vector)
    flat = torch.zeros(n_el + len(params), dtype=dt, device=dev)
    views, off = [], 0
    for p in params:
        views.append(flat[off:off + p.numel()].view(p.shape))
        off += p.numel()
    have = [i for i, p in enumerate(params) if p.grad is not None]
    if have:
        torch._foreach_copy_([views[i] for i in have], [params[i].grad for i in have])
        flat[n_el:].copy_(torch.tensor([1.0 if p.grad is not None else 0.0 for p in params], dtype=dt), non_blocking=False)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    # which parameters received a gradient on SOME rank: known without asking the device when this rank has them all (the
    # usual case — the job draws one quality per step, so every rank trains the same REM); reading the presence words back
    # is a device-to-host synchronisation that stops the host from running ahead of the GPU (measured: +7 ms per step)
    present = [1.0] * len(params) if len(have) == len(params) else flat[n_el:].tolist()
    flat[:n_el] /= dist.get_world_size()
    dst, src = [], []
    for i, p in enumerate(params):
        if present[i] > 0:
            if p.grad is None:
                p.grad = torch.empty_like(p)
            dst.append(p.grad)
            src.append(views[i])
    if dst:
        torch._foreach_copy_(dst, src)
    return flat.numel() * flat.element_size()


def broadcast_choice(n_choices: int, rng, device="cpu") -> int:
    """An index in [0, n_choices) drawn by rank 0's ``rng`` and shared with every rank: per-step random choices
    that select WHICH parameters train (the sampled quality picks the REM, training/step.py:62-64) must agree
    across ranks, or the gradient bucket would mix different modules' gradients."""
    import torch.distributed as dist
    idx = rng.randint(0, n_choices - 1)
    if not collectives_active():
        return idx
    t = torch.tensor([idx], dtype=torch.int64, device=device)
    dist.broadcast(t, src=0)
    return int(t.item())


def world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def bucket_partition(offsets, numels, done_steps, total: int, n_steps: int, bucket_bytes: int):
    """Cut a flat fp32 gradient buffer (parameter k at ``offsets[k]``, 16-byte aligned, laid out in the order the backward
    FINISHES the parameters) into buckets of at least ``bucket_bytes``: returns (bounds [(first element, one past the
    last)], ready [index of the backward-plan step after which the bucket is final]).  ``ready`` is monotone — a bucket is
    sent after its predecessors — and the last bucket closes with the plan.  Pure arithmetic on the plan's layout, hence
    identical on every rank."""
    bounds, ready = [], []
    lo, r = 0, 0
    for o, n, dn in zip(offsets, numels, done_steps):
        end = o + (n + 3) // 4 * 4
        r = max(r, dn)
        if (end - lo) * 4 >= bucket_bytes:
            bounds.append((lo, end))
            ready.append(r)
            lo = end
    if lo < total:
        bounds.append((lo, total))
        ready.append(n_steps)
    for i in range(1, len(ready)):
        ready[i] = max(ready[i], ready[i - 1])
    ready[-1] = n_steps
    return bounds, ready


class BucketReducer:
    """Gradient exchange of first-stage training (BASELINE configs[3]: "RCCL grad all-reduce over xGMI"): the
    backward plan calls this object with (bucket index, slice of its flat gradient buffer, stream the backward runs on)
    each time a bucket is final; the all-reduce of that bucket is issued on a communication stream behind an event of
    the backward's stream, so it overlaps the rest of the backward.  ``finish`` joins the collectives into the backward's
    stream and divides by the world size (mean over ranks = the single-process gradient of the global batch, since the
    loss is normalised by the LOCAL pixel count, training/loss.py:21,45).  Buckets are few and large (~25 MB): on the
    xGMI mesh a ring is bound by one link, so large messages beat many small ones.  The layout is rank-invariant by
    construction: every rank builds the same plan over the same parameter order, and a training forward uses every
    parameter of the path.  ``log`` keeps (bucket index, elements) in issue order (tests)."""

    def __init__(self, group=None):
        self.group = group
        self.comm = None
        self.pending: list = []
        self.log: list = []

    def __call__(self, idx: int, flat_slice: torch.Tensor, stream=None):
        import torch.distributed as dist
        self.log.append((idx, flat_slice.numel()))
        if not collectives_active():
            return
        if flat_slice.is_cuda:
            if self.comm is None:
                self.comm = torch.cuda.Stream(device=flat_slice.device)
            ev = torch.cuda.Event()
            ev.record(stream if stream is not None else torch.cuda.current_stream(flat_slice.device))
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                work = dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append((work, flat_slice))

    def finish(self, stream=None):
        """Order the caller's stream after every pending collective and turn the sums into means."""
        w = world_size()
        for work, sl in self.pending:
            work.wait()                      # NCCL: the CURRENT stream waits for the collective; gloo: the host does
            sl.div_(w)
        self.pending = []
