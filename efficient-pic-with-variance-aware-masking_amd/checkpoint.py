"""Checkpoint I/O compatibility: the counterparts of the reference's
``save_checkpoint`` / ``create_savepath`` / ``initialize_model_from_pretrained`` / ``replace_keys`` /
``configure_optimizers`` / ``AverageMeter`` (utility/functions.py:14-20,23-59,68-86,89-101,107-169) so that the
authors' ``.pth.tar`` files and the ``train.py`` bootstrapping from a single-encoder/decoder model map onto the
``vampic`` models (whose state_dict keys equal the reference's).  Pure host logic: dictionaries of tensors.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Optional

import torch


def create_savepath(base_path: str):
    """(last, very_best) checkpoint paths (utility/functions.py:165-168)."""
    return os.path.join(base_path, "_last.pth.tar"), os.path.join(base_path, "_very_best.pth.tar")


def save_checkpoint(state: dict, is_best: bool, last_pth: str, very_best: str):
    """utility/functions.py:14-20 without the wandb upload."""
    torch.save(state, very_best if is_best else last_pth)


def load_checkpoint(path: str, model=None, map_location="cpu", strict: bool = True, allow_pickle: bool = False) -> dict:
    """``torch.load`` + ``model.load_state_dict(checkpoint["state_dict"])`` as demo.py:46-52 / train.py:96-108 do
    (the REM model's own ``load_state_dict`` takes care of ``post_latent.*``).

    The reference's checkpoints carry ``"args": argparse.Namespace`` beside the tensors (train.py:371-383), which
    ``torch.load`` can only restore by unpickling.  The safe loader is tried first with ``argparse.Namespace``
    allow-listed; full unpickling (arbitrary code execution from an untrusted file) happens only on the explicit
    ``allow_pickle=True``."""
    import argparse
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            ck = torch.load(path, map_location=map_location, weights_only=True)
    except Exception as e:                                  # noqa: BLE001 - any unpickling restriction
        if not allow_pickle:
            raise RuntimeError(f"{path} holds objects the safe loader refuses ({type(e).__name__}: {e}); pass "
                               "allow_pickle=True only for checkpoints you trust") from e
        ck = torch.load(path, map_location=map_location, weights_only=False)
    if model is not None:
        model.load_state_dict(ck["state_dict"] if "state_dict" in ck else ck, strict=strict)
    return ck


def initialize_model_from_pretrained(checkpoint: dict, args, checkpoint_enh: Optional[dict] = None) -> OrderedDict:
    """Spread a single-encoder/decoder/hyperprior state_dict over the multi-component layout
    (utility/functions.py:107-164).  Reference behaviour kept as is: every key survives the first pass (its filter
    expression is always true), ``g_s.*`` / ``g_a.*`` move to component 0 when the target has several,
    all ``h_a`` / ``h_mean_s`` / ``h_scale_s`` keys are dropped and — with ``multiple_hyperprior`` — the two synthesis
    stacks come back as component 0; ``checkpoint_enh`` supplies decoder 1."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in checkpoint.items():
        if "g_s" in k:
            out[("g_s.0." + k[4:]) if args.multiple_decoder else k] = v
        elif "g_a" in k:
            out[("g_a.0." + k[4:]) if args.multiple_encoder else k] = v
        else:
            out[k] = v
    for k in [k for k in out if "h_scale_s" in k or "h_a" in k or "h_mean_s" in k]:
        del out[k]
    if args.multiple_hyperprior:
        for k, v in checkpoint.items():
            if "h_mean_s" in k:
                out["h_mean_s.0." + k[9:]] = v
            elif "h_scale_s" in k:
                out["h_scale_s.0." + k[10:]] = v
    for k in [k for k in out if "h_a" in k]:
        del out[k]
    if checkpoint_enh is not None:
        for k, v in checkpoint_enh.items():
            if "g_s" in k:
                out["g_s.1." + k[4:]] = v
    return out


def replace_keys(checkpoint: dict, multiple_encoder: bool) -> OrderedDict:
    """``g_a_enh.*`` -> ``g_a.1.*`` and bare ``g_a.*`` -> ``g_a.0.*`` for older checkpoints
    (utility/functions.py:68-86)."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    already = "g_a.0.1.beta" in checkpoint
    for k, v in checkpoint.items():
        if multiple_encoder and "g_a_enh." in k:
            out[k.replace("g_a_enh.", "g_a.1.")] = v
        elif multiple_encoder and "g_a." in k and not already:
            out[k.replace("g_a.", "g_a.0.")] = v
        else:
            out[k] = v
    return out


def configure_optimizers(net, args):
    """Adam over everything except ``*.quantiles`` (+ an Adam over the quantiles, returned only for the
    first-stage schedule, as the reference spells it) — utility/functions.py:23-59."""
    named = dict(net.named_parameters())
    main = sorted(n for n in named if not n.endswith(".quantiles"))
    aux = sorted(n for n in named if n.endswith(".quantiles"))
    # ``args.fused_adam`` (not a reference option; default off): torch's single-kernel Adam — the same update, one launch for
    # all 1065 tensors instead of a dozen foreach launches per chunk
    kw = {"fused": True} if getattr(args, "fused_adam", False) else {}
    optimizer = torch.optim.Adam((named[n] for n in main), lr=args.learning_rate, **kw)
    aux_optimizer = torch.optim.Adam((named[n] for n in aux), lr=args.aux_learning_rate, **kw)
    return (optimizer, aux_optimizer) if args.training_type == "first_strain" else (optimizer, None)


class AverageMeter:
    """utility/functions.py:89-101."""

    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
