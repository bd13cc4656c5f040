"""Evaluation drivers and metric helpers (inference half): the counterparts of
``test_epoch`` / ``compress_with_ac`` (reference training/step.py:206-243,259-358) and of
``compute_psnr`` / ``compute_padding`` (utility/functions.py:172-174,191-219) on top of the HIP model.
Rates come from the in-kernel ``log2`` accumulators (a13), squared errors from ``vam_sqdiff_sum``.
"""
from __future__ import annotations

import math
import time
from typing import Iterable, List, Optional, Sequence

import torch

from . import ops


def compute_padding(in_h: int, in_w: int, *, out_h=None, out_w=None, min_div=1):
    """utility/functions.py:191-219: (left, right, top, bottom) pad and un-pad tuples to a multiple of min_div."""
    if out_h is None:
        out_h = (in_h + min_div - 1) // min_div * min_div
    if out_w is None:
        out_w = (in_w + min_div - 1) // min_div * min_div
    if out_h % min_div != 0 or out_w % min_div != 0:
        raise ValueError(f"Padded output height and width are not divisible by min_div={min_div}.")
    left = (out_w - in_w) // 2
    right = out_w - in_w - left
    top = (out_h - in_h) // 2
    bottom = out_h - in_h - top
    return (left, right, top, bottom), (-left, -right, -top, -bottom)


def pad_image(x: torch.Tensor, min_div: int = 64):
    """test/utils.py:7-13: zero-pad to a multiple of 64 (6 stride-2 stages); returns (x_padded, unpad)."""
    pad, unpad = compute_padding(x.size(2), x.size(3), min_div=min_div)
    return torch.nn.functional.pad(x, pad, mode="constant", value=0), unpad


def compute_psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """utility/functions.py:172-174: -10 log10(mean((a-b)^2)) over the whole tensor pair."""
    a, b = a.contiguous(), b.contiguous()
    acc = torch.zeros(1, dtype=torch.float64, device=a.device)
    ops.sqdiff_sum(a, b, acc)
    return -10.0 * math.log10(acc.item() / a.numel())


def estimated_bpp(out: dict, num_pixels: int) -> float:
    """training/loss.py:217-228 (RateLoss): (sum log2 lik_y + sum log2 lik_z) / (-pixels)."""
    return -out["log2_likelihood_sum"].sum().item() / num_pixels


def _checkpoint_for(model, x, p):
    """training/step.py:13-29 extract_quality_ref + ExtractChekpointRepr (REM models only)."""
    levels = getattr(model, "check_levels", None)
    if not levels or p <= levels[0]:
        return None
    q_ref = max(l for l in levels if l < p)
    return model.ExtractChekpointRepr(x, quality=q_ref, rc=False)


def test_epoch(batches: Iterable[torch.Tensor], model, pr_list: Sequence[float], rems: bool = False):
    """training/step.py:206-243: likelihood-estimated (bpp, PSNR) averaged over the batches, per quality."""
    bpp = [[] for _ in pr_list]
    psnr = [[] for _ in pr_list]
    with torch.no_grad():
        for d in batches:
            n_pix = d.shape[0] * d.shape[2] * d.shape[3]
            for j, p in enumerate(pr_list):
                ck = _checkpoint_for(model, d, p) if rems else None
                out = model.forward_single_quality(d, quality=p, training=False, **({"checkpoint_ref": ck} if rems else {}))
                bpp[j].append(estimated_bpp(out, n_pix))
                psnr[j].append(compute_psnr(d, out["x_hat"]))
    return [sum(v) / len(v) for v in bpp], [sum(v) / len(v) for v in psnr]


def compress_with_ac(model, images: Iterable[torch.Tensor], pr_list: Sequence[float], rems: bool = False,
                     with_msssim: bool = False):
    """training/step.py:259-358: real codec evaluation — compress + decompress every (unpadded) image at every
    quality; bpp = 8 * bytes / pixels of the ORIGINAL image, PSNR on the cropped reconstruction.
    Returns (bpp, psnr, enc_seconds, dec_seconds) lists per quality, plus the MS-SSIM in dB
    (-10 log10(1 - ms_ssim), step.py:323-324) as a fifth list when ``with_msssim``."""
    nq = len(pr_list)
    bpp, psnr, t_enc, t_dec = [[] for _ in range(nq)], [[] for _ in range(nq)], [[] for _ in range(nq)], [[] for _ in range(nq)]
    mssim = [[] for _ in range(nq)]
    with torch.no_grad():
        for x in images:
            xp, unpad = pad_image(x)
            for j, p in enumerate(pr_list):
                ck = _checkpoint_for(model, xp, p) if rems else None
                t0 = time.perf_counter()
                enc = model.compress(xp, quality=p, checkpoint_rep=ck)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                dec = model.decompress(enc["strings"], enc["shape"], quality=p, checkpoint_rep=ck)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                x_hat = torch.nn.functional.pad(dec["x_hat"], unpad)
                n_bytes = sum(len(s) for sl in enc["strings"][0] for s in sl) + sum(len(s) for s in enc["strings"][1])
                bpp[j].append(8.0 * n_bytes / (x.shape[0] * x.shape[2] * x.shape[3]))
                psnr[j].append(compute_psnr(x, x_hat))
                if with_msssim:
                    mssim[j].append(-10.0 * math.log10(max(1.0 - compute_msssim(x, x_hat), 1e-12)))
                t_enc[j].append(t1 - t0)
                t_dec[j].append(t2 - t1)
    avg = lambda rows: [sum(v) / len(v) for v in rows]
    if with_msssim:
        return avg(bpp), avg(psnr), avg(t_enc), avg(t_dec), avg(mssim)
    return avg(bpp), avg(psnr), avg(t_enc), avg(t_dec)


def valid_epoch(epoch: int, test_dataloader: Iterable[torch.Tensor], criterion, model, pr_list: Sequence[float] = (0.05,),
                rems: Optional[Sequence[float]] = None):
    """training/step.py:136-202 (without wandb): mean criterion loss over batches x qualities — what drives the
    ReduceLROnPlateau scheduler of train.py:130,279.  ``rems`` = the check levels (REM models) or None."""
    from .finetune import extract_quality_ref
    model.eval()
    device = next(model.parameters()).device
    tot = {"loss": 0.0, "bpp": 0.0, "mse": 0.0, "psnr": 0.0}
    n = 0
    with torch.no_grad():
        for d in test_dataloader:
            d = d.to(device)
            for p in pr_list:
                if rems is None:
                    out = model.forward_single_quality(d, quality=p, training=False)
                else:
                    q_ref = extract_quality_ref(p, rems)
                    ck = None if q_ref is None else model.ExtractChekpointRepr(d, quality=q_ref, rc=False)
                    out = model.forward_single_quality(d, quality=p, training=False, checkpoint_ref=ck)
                crit = criterion(out, d)
                psnr = compute_psnr(d, out["x_hat"])
                tot["loss"] += float(crit["loss"])
                tot["bpp"] += float(crit["bpp_loss"])
                tot["mse"] += 10.0 ** (-psnr / 10.0)
                tot["psnr"] += psnr
                n += 1
    n = max(n, 1)
    return tot["loss"] / n, {k: v / n for k, v in tot.items()}


def read_image(filepath) -> torch.Tensor:
    """utility/functions.py:62-66: RGB image file -> float32 [3,H,W] in [0,1] (what torchvision's ToTensor does to
    an 8-bit image: value / 255, HWC -> CHW)."""
    import numpy as np
    from PIL import Image
    img = Image.open(filepath).convert("RGB")
    a = np.asarray(img, dtype=np.uint8)
    return torch.from_numpy(a.copy()).permute(2, 0, 1).to(torch.float32).div(255.0)


def write_image(x: torch.Tensor, filepath):
    """Inverse of :func:`read_image` for a [3,H,W] or [1,3,H,W] tensor in [0,1] (demo.py saves reconstructions)."""
    import numpy as np
    from PIL import Image
    if x.dim() == 4:
        x = x[0]
    a = (x.detach().clamp(0, 1).mul(255.0).round().to(torch.uint8).permute(1, 2, 0).cpu().numpy())
    Image.fromarray(np.ascontiguousarray(a), mode="RGB").save(filepath)


MS_SSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def compute_msssim(a: torch.Tensor, b: torch.Tensor, data_range: float = 1.0) -> float:
    """utility/functions.py:176-177: ``ms_ssim(a, b, data_range=1.)`` of pytorch_msssim 0.2.1 — 11-tap Gaussian
    (sigma 1.5), 5 scales, default weights, mean over images and channels — on the GPU (``vam_ssim_level``,
    ``vam_avgpool2``).  NCHW fp32 CUDA tensors; the smaller side must exceed 160 pixels."""
    from . import _lib as L
    L.require_gpu()
    if a.shape != b.shape or a.dim() != 4:
        raise ValueError(f"expected two [B,C,H,W] tensors of one shape, got {tuple(a.shape)} and {tuple(b.shape)}")
    if min(a.shape[2:]) <= (11 - 1) * 2 ** 4:
        raise ValueError("image too small for 5 scales with an 11-tap window (smaller side must exceed 160)")
    lib = L.load()
    x, y = a.detach().float().contiguous(), b.detach().float().contiguous()
    B, C_, H, W = x.shape
    planes = B * C_
    coords = torch.arange(11, dtype=torch.float32) - 5
    g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    win = (g / g.sum()).to(x.device)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    vals = []
    for lvl in range(5):
        sums = torch.zeros((2, planes), dtype=torch.float64, device=x.device)
        L.check(lib.vam_ssim_level(x.data_ptr(), y.data_ptr(), planes, H, W, win.data_ptr(), c1, c2, sums[0].data_ptr(),
                                   sums[1].data_ptr(), ops.stream_ptr()), "vam_ssim_level")
        mean = sums / float((H - 10) * (W - 10))
        vals.append(torch.relu(mean[1] if lvl < 4 else mean[0]))
        if lvl < 4:
            ph, pw = H % 2, W % 2
            Ho, Wo = (H + 2 * ph - 2) // 2 + 1, (W + 2 * pw - 2) // 2 + 1
            nx = torch.empty((B, C_, Ho, Wo), dtype=torch.float32, device=x.device)
            ny = torch.empty_like(nx)
            L.check(lib.vam_avgpool2(x.data_ptr(), nx.data_ptr(), planes, H, W, ph, pw, ops.stream_ptr()), "vam_avgpool2")
            L.check(lib.vam_avgpool2(y.data_ptr(), ny.data_ptr(), planes, H, W, ph, pw, ops.stream_ptr()), "vam_avgpool2")
            x, y, H, W = nx, ny, Ho, Wo
    w = torch.tensor(MS_SSIM_WEIGHTS, dtype=torch.float64, device=x.device).reshape(-1, 1)
    return float(torch.prod(torch.stack(vals, 0) ** w, dim=0).mean())
