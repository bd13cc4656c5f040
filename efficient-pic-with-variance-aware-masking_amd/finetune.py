"""REM fine-tune driver (BASELINE configs[4]; reference train.py:167-184,223-226 and
training/step.py:14-95 with ``--training_type rems``; loss training/loss.py:189-229).

Only the Rate-Enhancement blocks train; every step is
    checkpoint latent at the check level (no grad)  ->  training-mode forward at a sampled quality
    ->  RateLoss  ->  backward (HIP kernels, csrc/train.hip)  ->  gradient all-reduce (RCCL)  ->  clip  ->  Adam.
Images shard across ranks; the only collective is the all-reduce of the REM gradients
(15.2 MB per REM in fp32), issued as ONE flat bucket: xGMI rings are per-link bound, one
large message beats 420 small ones.
"""
from __future__ import annotations

import math
import random
from typing import Iterable, List, Optional, Sequence

import torch
import torch.nn as nn

from . import sharding


class RateLoss(nn.Module):
    """training/loss.py:189-229: loss = bpp(y) + n_recon * bpp(z); the MSE is reported only."""

    def __init__(self, weight=255 ** 2, device="cuda"):
        super().__init__()
        self.weight, self.device = weight, device

    def forward(self, output, target):
        n_img, _, H, W = target.size()
        n_rec = output["x_hat"].shape[0]
        tgt = target.unsqueeze(0)
        if n_rec != 1 and n_rec != n_img:
            tgt = tgt.unsqueeze(0).repeat(n_rec, 1, 1, 1, 1)
        out = {"mse_loss": ((tgt - output["x_hat"].detach()) ** 2).mean(dim=(1, 2, 3, 4))}
        den = -math.log(2) * n_img * H * W
        lik = output["likelihoods"]
        out["bpp_hype"] = torch.log(lik["z"]).sum() / den
        out["bpp_base"] = torch.log(lik["y"].squeeze(0)).sum() / den
        out["bpp_scalable"] = out["bpp_base"]
        out["bpp_loss"] = out["bpp_base"] + n_rec * out["bpp_hype"]
        out["loss"] = out["bpp_loss"]
        return out


def extract_quality_ref(quality: float, check_levels: Sequence[float]) -> Optional[float]:
    """Which checkpoint level feeds the REM at ``quality`` (training/step.py:14-32)."""
    cl = list(check_levels)
    if quality <= cl[0]:
        return None
    if len(cl) in (2, 3) and cl[0] < quality <= cl[1]:
        return cl[0]
    if len(cl) == 2 and quality > cl[1]:
        return cl[1]
    if len(cl) == 3 and cl[1] < quality <= cl[2]:
        return cl[1]
    return cl[-1]


def rems_quality_list(check_levels: Sequence[float], check_levels_np: Sequence[int]) -> List[float]:
    """Qualities sampled during REM training (train.py:167-181)."""
    import numpy as np
    levels = list(check_levels) + [10]
    qs: List[float] = []
    for i in range(len(levels) - 1):
        cur, nxt = levels[i], levels[i + 1]
        start = cur + 0.01 if i == 0 else cur
        qs.extend(np.arange(start, nxt, (nxt - cur) / check_levels_np[i]).tolist())
    qs = [round(q, 4) for q in qs]
    if 10 not in qs:
        qs.append(10)
    return qs


def _dist_backend() -> Optional[str]:
    import torch.distributed as dist
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None


def finetune_step(model, criterion, batch: torch.Tensor, optimizer, quality: float, check_levels: Sequence[float],
                  clip_max_norm: float = 1.0, noise=None, fused: bool = True) -> dict:
    """One optimisation step on this rank's shard of the batch (training/step.py:56-95).  ``fused``: the
    checkpoint latent comes out of the training forward's own front end (``forward_finetune``, same bits) instead
    of a separate ``ExtractChekpointRepr`` pass."""
    optimizer.zero_grad()
    if fused:
        out = model.forward_finetune(batch, quality, noise=noise)
    else:
        q_ref = extract_quality_ref(quality, check_levels)
        with torch.no_grad():
            ck = None if q_ref is None else model.ExtractChekpointRepr(batch, quality=q_ref, rc=False)
        out = model.forward_single_quality(batch, quality=quality, training=True, checkpoint_ref=ck, noise=noise)
    crit = criterion(out, batch)
    crit["loss"].backward()
    sharding.all_reduce_gradients([p for p in model.parameters() if p.requires_grad])   # rank-invariant bucket
    if clip_max_norm > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_max_norm)
    optimizer.step()
    return crit


def train_one_epoch(model, criterion, train_dataloader: Iterable[torch.Tensor], optimizer, epoch: int, counter: int,
                    list_quality: Sequence[float], rems: Sequence[float], clip_max_norm: float = 1.0, rng=random):
    """``train_one_epoch(..., sampling_training=True, rems=check_levels)`` of the reference: one sampled quality
    per batch.  Returns (counter, mean loss, mean bpp, mean mse, mean scalable bpp)."""
    model.train()
    device = next(model.parameters()).device
    tot = {"loss": 0.0, "bpp_loss": 0.0, "mse_loss": 0.0, "bpp_scalable": 0.0}
    n = 0
    for d in train_dataloader:
        d = d.to(device)
        # one quality per step for the WHOLE job: rank 0 draws, every rank uses it (the quality selects which REM
        # trains; ranks drawing independently would average gradients of different REMs)
        red_dev = device if (_dist_backend() == "nccl") else "cpu"
        q = list_quality[sharding.broadcast_choice(len(list_quality), rng, red_dev)]
        crit = finetune_step(model, criterion, d, optimizer, q, rems, clip_max_norm)
        for k in tot:
            tot[k] += float(crit[k].detach().mean())
        n += 1
        counter += 1
    n = max(n, 1)
    return counter, tot["loss"] / n, tot["bpp_loss"] / n, tot["mse_loss"] / n, tot["bpp_scalable"] / n
