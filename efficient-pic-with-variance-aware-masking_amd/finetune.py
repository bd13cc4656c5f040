"""Fine-tune drivers: REM fine-tune (BASELINE configs[4]; reference train.py:167-184,223-226 and
training/step.py:14-95 with ``--training_type rems``; loss training/loss.py:189-229) and decoder refinement
(``--training_type refine_gs`` [``--lrp``]: train.py:150-157,216-218, training/step.py:56-99, loss training/loss.py:126-187;
at the end of this file).

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


class _PlanObjectsParked:
    """The epoch loops' garbage-collector policy.  A training plan holds ~1e6 long-lived Python objects (problem structs,
    views, closures); a generation-2 pass of the cyclic collector over them takes ~150 ms and falls into the step loop every
    few steps (measured r03 / r04: 9-14 ms per REM fine-tune step on average).  After the first step of an epoch has built
    the plans, everything alive is moved to the permanent generation (``gc.freeze``) for the rest of the epoch and
    released again at its end (``gc.unfreeze``): reference counting still frees dropped objects at once, only the
    cycle scans skip the parked ones.  The benches report the step time with and without it."""

    def __init__(self):
        self.parked = False

    def after_step(self):
        if not self.parked:
            import gc
            gc.collect()
            gc.freeze()
            self.parked = True

    def release(self):
        if self.parked:
            import gc
            gc.unfreeze()
            self.parked = False


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
    park = _PlanObjectsParked()
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
        park.after_step()
    park.release()
    n = max(n, 1)
    return counter, tot["loss"] / n, tot["bpp_loss"] / n, tot["mse_loss"] / n, tot["bpp_scalable"] / n


# ============================================================================= decoder refinement (--training_type refine_gs)
class DistortionLoss(nn.Module):
    """training/loss.py:126-187: loss = weight * lmbda * mse(x, x_hat); the rates are reported only (and, like the
    reference's, computed from whatever likelihoods the forward returned)."""

    def __init__(self, weight=255 ** 2, device="cuda"):
        super().__init__()
        self.weight, self.device = weight, device

    def forward(self, output, target, lmbda=1e-2):
        n_img, _, H, W = target.size()
        n_rec = output["x_hat"].shape[0]
        tgt = target
        if n_rec != 1 and n_rec != n_img:
            tgt = target.unsqueeze(0).repeat(n_rec, 1, 1, 1, 1)
        out = {"mse_loss": torch.nn.functional.mse_loss(tgt, output["x_hat"])}
        den = -math.log(2) * n_img * H * W
        lik = output["likelihoods"]
        out["bpp_hype"] = torch.log(lik["z"].detach()).sum() / den
        out["bpp_base"] = torch.log(lik["y"].detach().squeeze(0)).sum() / den
        out["bpp_scalable"] = out["bpp_base"] * 0.0
        out["bpp_loss"] = out["bpp_scalable"] + out["bpp_base"] + n_rec * out["bpp_hype"]
        out["loss"] = self.weight * (lmbda * out["mse_loss"]).mean()
        return out


def refine_gs_quality_list() -> List[float]:
    """The qualities a ``refine_gs`` epoch samples from (train.py:150-155; the same two ``np.arange`` expressions: 203 levels
    up to 1.5, 51 from 1.6 to 10)."""
    import numpy as np
    a = list(np.arange(0.015, 1.5, (1.5 - 0.025) / 200)) + [1.5]
    b = list(np.arange(1.6, 10, (10 - 1.6) / 50)) + [10]
    return [float(q) for q in a + b]


def refine_gs_setup(model, lrp: bool = False):
    """train.py:216-218: ``freeze_all(); unfreeze_decoder(lrp=args.lrp)``; returns the trainable parameters."""
    model.freeze_all()
    model.unfreeze_decoder(lrp=lrp)
    return [p for p in model.parameters() if p.requires_grad]


def refine_gs_step(model, criterion, batch: torch.Tensor, optimizer, quality: float, clip_max_norm: float = 1.0,
                   lmbda: float = 1e-2, noise=None) -> dict:
    """One optimisation step of the decoder-refinement schedule on this rank's shard (training/step.py:56-99 with
    ``sampling_training=True, rems=None``): training-mode forward at ``quality``, DistortionLoss, backward (HIP kernels of
    gs_train.py), gradient all-reduce in one rank-invariant bucket, clip, optimizer step."""
    optimizer.zero_grad()
    out = model.forward_single_quality(batch, quality=quality, training=True, noise=noise)
    crit = criterion(out, batch, lmbda=lmbda)
    crit["loss"].backward()
    sharding.all_reduce_gradients([p for p in model.parameters() if p.requires_grad])
    if clip_max_norm > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_max_norm)
    optimizer.step()
    return crit


def train_one_epoch_refine_gs(model, criterion, train_dataloader: Iterable[torch.Tensor], optimizer, epoch: int, counter: int,
                              list_quality: Optional[Sequence[float]] = None, clip_max_norm: float = 1.0, rng=random):
    """``train_one_epoch(..., sampling_training=True, list_quality=..., lmbda_list=None, rems=None)`` of the reference:
    one sampled quality per batch (drawn on rank 0 for the whole job, so that the ranks' gradients belong to the same
    objective).  Returns (counter, mean loss, mean bpp, mean mse, mean scalable bpp)."""
    model.train()
    device = next(model.parameters()).device
    list_quality = refine_gs_quality_list() if list_quality is None else list_quality
    tot = {"loss": 0.0, "bpp_loss": 0.0, "mse_loss": 0.0, "bpp_scalable": 0.0}
    n = 0
    park = _PlanObjectsParked()
    for d in train_dataloader:
        d = d.to(device)
        red_dev = device if (_dist_backend() == "nccl") else "cpu"
        q = list_quality[sharding.broadcast_choice(len(list_quality), rng, red_dev)]
        crit = refine_gs_step(model, criterion, d, optimizer, q, clip_max_norm)
        for k in tot:
            tot[k] += float(crit[k].detach().mean())
        n += 1
        counter += 1
        park.after_step()
    park.release()
    n = max(n, 1)
    return counter, tot["loss"] / n, tot["bpp_loss"] / n, tot["mse_loss"] / n, tot["bpp_scalable"] / n


# ============================================================================= first-stage training (--training_type first_train)
class ScalableRateDistortionLoss(nn.Module):
    """training/loss.py:6-66: loss = bpp_scalable + bpp_base + n_rec * bpp_hype + weight * mean(lmbda * mse per level).
    Kept with the reference's own accounting: "y_prog" holds the base slices' likelihoods again, so the base rate enters
    twice; ``n_rec`` = x_hat.shape[0] — the number of levels for the stacked output of ``forward``, the
    number of images for a single-quality output."""

    def __init__(self, weight=255 ** 2, lmbda_list=(0.005, 0.05), device="cuda"):
        super().__init__()
        self.scalable_levels = len(lmbda_list)
        self.lmbda = torch.tensor(list(lmbda_list), dtype=torch.float32).to(device)
        self.weight, self.device = weight, device

    def forward(self, output, target, lmbda=None):
        n_img, _, H, W = target.size()
        n_rec = output["x_hat"].shape[0]
        tgt = target.unsqueeze(0)
        if n_rec != 1 and n_rec != n_img:
            tgt = tgt.repeat(n_rec, 1, 1, 1, 1)
        lm = self.lmbda if lmbda is None else torch.tensor([lmbda], dtype=torch.float32).to(self.device)
        out = {"mse_loss": ((tgt - output["x_hat"]) ** 2).mean(dim=(1, 2, 3, 4))}       # one value per level
        den = -math.log(2) * n_img * H * W
        lik = output["likelihoods"]
        out["bpp_hype"] = torch.log(lik["z"]).sum() / den
        if "y_prog" in lik:
            out["bpp_base"] = torch.log(lik["y"]).sum() / den
            out["bpp_scalable"] = torch.log(lik["y_prog"]).sum() / den
        else:
            out["bpp_base"] = torch.log(lik["y"].squeeze(0)).sum() / den
            out["bpp_scalable"] = torch.log(lik["y"]).sum() / den * 0.0
        out["bpp_loss"] = out["bpp_scalable"] + out["bpp_base"] + n_rec * out["bpp_hype"]
        out["loss"] = out["bpp_loss"] + self.weight * (lm * out["mse_loss"]).mean()
        return out


def clip_grad_norm_(model, max_norm: float):
    """``torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)`` (training/step.py:98) — as ONE reduction when the
    gradients of all parameters are views of the first-stage plan's flat buffer (models._FullTrainFn.backward hands them
    out that way; padding between tensors is zero), instead of a walk over 1065 tensors.  Anything else (a frozen
    subset, gradients from another source such as the aux loss, accumulated gradients) takes torch's own routine."""
    plans = [p for k, p in getattr(model, "_plans", {}).items() if k[0] == "full_train" and getattr(p, "handout", None) is not None]
    if len(plans) == 1 and plans[0].handout[2]:
        flat, n_out, _ = plans[0].handout
        base, ok, n = flat.untyped_storage().data_ptr(), True, 0
        for p in model.parameters():
            g = p.grad
            if g is None:
                continue
            n += 1
            if g.untyped_storage().data_ptr() != base:
                ok = False
                break
        if ok and n == n_out:
            total = torch.linalg.vector_norm(flat)
            flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
            return total
    return torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)


def first_train_setup(model):
    """train.py:146-149 / :214-226: the first stage freezes nothing."""
    for p in model.parameters():
        p.requires_grad = True
    return [p for p in model.parameters() if p.requires_grad]


def first_train_step(model, criterion, batch: torch.Tensor, optimizer, list_quality: Sequence[float] = (0, 10),
                     clip_max_norm: float = 1.0, noise=None, aux_optimizer=None) -> dict:
    """One optimisation step of the first-stage schedule on this rank's shard (training/step.py:56-99 with
    ``sampling_training=False``): ``out = model(d, quality=[0, 10])``, criterion, backward, clip, step.  On a multi-GPU
    job the gradients are averaged over the ranks DURING ``backward()``: the plan's flat gradient buffer is cut into
    ~25 MB buckets in the order the backward finishes them, and each bucket's all-reduce (RCCL) is issued on a
    communication stream as soon as its last weight gradient has been enqueued (sharding.BucketReducer); the clip runs
    after the last bucket, on the averaged gradients, so every rank applies the same scale."""
    from . import sharding as S
    optimizer.zero_grad()
    if aux_optimizer is not None:
        aux_optimizer.zero_grad()
    if S.collectives_active() and getattr(model, "grad_reducer", None) is None:
        model.grad_reducer = S.BucketReducer()
    out = model(batch, quality=list(list_quality), training=True, noise=noise)
    crit = criterion(out, batch)
    crit["loss"].backward()
    if aux_optimizer is not None:                      # training/step.py:91-94 (never taken by the reference: "first_strain")
        aux = model.aux_loss()
        aux.backward()
        aux_optimizer.step()
    if clip_max_norm > 0:
        clip_grad_norm_(model, clip_max_norm)
    optimizer.step()
    return crit


def train_one_epoch_first_train(model, criterion, train_dataloader: Iterable[torch.Tensor], optimizer, epoch: int, counter: int,
                                list_quality: Sequence[float] = (0, 10), clip_max_norm: float = 1.0, aux_optimizer=None):
    """``train_one_epoch(model, criterion, loader, optimizer, aux_optimizer, epoch, counter, sampling_training=False,
    list_quality=[0, 10])`` of the reference (training/step.py:32-135).  Returns (counter, mean loss, mean bpp, mean mse,
    mean scalable bpp)."""
    model.train()
    device = next(model.parameters()).device
    tot = {"loss": 0.0, "bpp_loss": 0.0, "mse_loss": 0.0, "bpp_scalable": 0.0}
    n = 0
    park = _PlanObjectsParked()
    for d in train_dataloader:
        crit = first_train_step(model, criterion, d.to(device), optimizer, list_quality, clip_max_norm, aux_optimizer=aux_optimizer)
        for k in tot:
            tot[k] += float(crit[k].detach().mean())
        n += 1
        counter += 1
        park.after_step()
    park.release()
    n = max(n, 1)
    return counter, tot["loss"] / n, tot["bpp_loss"] / n, tot["mse_loss"] / n, tot["bpp_scalable"] / n


# ============================================================================= encoder + decoder refinement (--training_type refine_gs_ga)
class RateDistortionLoss(ScalableRateDistortionLoss):
    """training/loss.py:67-124: ScalableRateDistortionLoss's arithmetic with the Lagrangian passed per call.  The reference
    never sets ``self.lmbda`` (:95-96), so calling it without ``lmbda`` fails there too."""

    def __init__(self, weight=255 ** 2, device="cuda"):
        nn.Module.__init__(self)
        self.weight, self.device = weight, device

    def forward(self, output, target, lmbda=None):
        if lmbda is None:
            raise AttributeError("RateDistortionLoss has no default lmbda (training/loss.py:95-96): pass lmbda=")
        return ScalableRateDistortionLoss.forward(self, output, target, lmbda=lmbda)


def refine_gs_ga_setup(model):
    """train.py:219-222: ``freeze_all(); unfreeze_decoder(); unfreeze_encoder()`` — the progressive decoder g_s[1] and
    the enhancement encoder g_a[1] train (pic.py:171-191)."""
    model.freeze_all()
    model.unfreeze_decoder()
    model.unfreeze_encoder()
    return [p for p in model.parameters() if p.requires_grad]


def refine_gs_ga_lambdas(lmbda_list: Sequence[float], n_qualities: int) -> List[float]:
    """train.py:158-166: one Lagrangian per sampled quality, log-spaced between the two ends of ``--lmbda_list``
    (``torch.logspace(log10(l0), log10(l1), steps=n + 1)[1:]``)."""
    start, end = math.log10(lmbda_list[0]), math.log10(lmbda_list[1])
    return torch.logspace(start, end, steps=n_qualities + 1)[1:].tolist()


def refine_gs_ga_step(model, criterion, batch: torch.Tensor, optimizer, quality: float, lmbda: float,
                      clip_max_norm: float = 1.0, noise=None) -> dict:
    """One step of the ``refine_gs_ga`` schedule: training forward at the sampled quality, rate-distortion loss at that
    quality's Lagrangian, backward (full_train.FullTrainPlan: gradients reach g_a[1] through the likelihoods and both
    straight-through paths), exchange, clip, step.  The REFERENCE's loop cannot run this schedule: with ``lmbda_list`` set,
    training/step.py:84 calls the criterion and discards its result, and the next line reads ``out_criterion`` — a NameError
    on the first batch.  This mirrors what that line evidently means to do (keep the result)."""
    from . import sharding as S
    optimizer.zero_grad()
    if S.collectives_active() and getattr(model, "grad_reducer", None) is None:
        model.grad_reducer = S.BucketReducer()
    out = model.forward_single_quality(batch, quality=quality, training=True, noise=noise)
    crit = criterion(out, batch, lmbda=lmbda)
    crit["loss"].backward()
    if clip_max_norm > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_max_norm)
    optimizer.step()
    return crit
