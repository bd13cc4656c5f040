"""CPU oracle for the variance-aware-masking codec hot path.

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package never does (it fails loudly when the HIP
library is missing instead of falling back to anything in here).

This is an independent restatement (numpy for the integer/rank arithmetic,
plain ATen CPU ops for the dense float arithmetic) of the reference's
``forward_single_quality`` data flow and of every operator on it.  It is
*functional*: it takes a flat ``state_dict`` with the reference's key names and
never instantiates an ``nn.Module``.  Each function cites the reference lines
it follows (paths relative to ``/root/reference/src``).

Pinning: ``oracle/gen_golden.py`` runs the real reference (imported from
``/root/reference`` with the two third-party stand-ins of ``oracle/ref_stubs``)
and this oracle on the same seeded weights/inputs and commits the reference's
outputs under ``tests/golden``; ``tests/test_oracle_golden.py`` re-checks the
oracle against those vectors on every run.  The arithmetic of compressai's
``LowerBound`` / ``NonNegativeParametrizer`` is restated from the published
CompressAI 1.2.4 definition (the package is absent offline), so those two
helpers are pinned only through the stand-ins — see DESIGN.md.
"""
from __future__ import annotations

import math
from fractions import Fraction
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64  # models/pic.py:12-14
SCALE_BOUND = 0.11          # entropy_models.py:535
LIKELIHOOD_BOUND = 1e-9     # entropy_models.py:83
NUM_HEADS = 8               # models/builder.py:9,14,48,52


# --------------------------------------------------------------------------
# A.1  variance mask  (layers/channel_mask.py:132-151, 18-49)
# --------------------------------------------------------------------------
def _fma32(a: np.float32, b: np.float32, c: np.float32) -> np.float32:
    """Correctly rounded fp32 fused multiply-add a*b+c (exact rational arithmetic,
    then one round-to-nearest-even)."""
    if not (np.isfinite(a) and np.isfinite(b) and np.isfinite(c)):
        with np.errstate(all="ignore"):
            return np.float32(np.float64(a) * np.float64(b) + np.float64(c))
    exact = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
    cand = np.float32(float(exact))
    if not np.isfinite(cand):
        return cand
    best, best_err = cand, abs(Fraction(float(cand)) - exact)
    for nb in (np.nextafter(cand, np.float32(-np.inf)), np.nextafter(cand, np.float32(np.inf))):
        if not np.isfinite(nb):
            continue
        err = abs(Fraction(float(nb)) - exact)
        if err < best_err or (err == best_err and (nb.view(np.uint32) & 1) == 0 and (best.view(np.uint32) & 1) == 1):
            best, best_err = nb, err
    return np.float32(best)


def quantile_threshold_np(seg: np.ndarray, q_keep: float) -> np.float32:
    """fp32-exact restatement of ``torch.quantile(seg, 1 - q_keep)`` (linear
    interpolation) for one flattened fp32 segment.

    ``q_keep`` is the python float ``min(pr, 10) * 0.1`` of channel_mask.py:137-139.
    ATen computes rank = float32(q) * float32(n-1), gathers the two neighbours and
    calls lerp_, whose CPU kernel evaluates ``fma(w, b-a, a)`` for w < 0.5 and
    ``fma(w-1, b-a, b)`` otherwise — a FUSED multiply-add (checked against
    torch.quantile on 3120 cases: 0 mismatches fused, 7 unfused).
    """
    s = np.sort(seg.astype(np.float32, copy=False).ravel(), kind="stable")
    n = s.size
    if np.isnan(s).any():
        return np.float32(np.nan)
    qt = np.float32(1.0 - q_keep)               # python double -> fp32 scalar tensor
    rank = np.float32(qt * np.float32(n - 1))   # fp32 multiply
    lo = int(np.floor(rank))
    hi = int(np.ceil(rank))
    w = np.float32(rank - np.float32(lo))
    a, b = s[lo], s[hi]
    with np.errstate(all="ignore"):
        d = np.float32(b - a)
        if w < np.float32(0.5):
            return _fma32(w, d, a)
        return _fma32(np.float32(w - np.float32(1.0)), d, b)


def variance_mask_np(scale: np.ndarray, pr: float) -> np.ndarray:
    """``ChannelMask.forward(scale, pr, "point-based-std")`` for a [B, ...] fp32
    array: one quantile per batch item over all of its elements.  Returns a
    float32 {0,1} array of the same shape (channel_mask.py:132-151)."""
    scale = np.asarray(scale, dtype=np.float32)
    if pr >= 10:
        return np.ones_like(scale)
    if pr == 0:
        return np.zeros_like(scale)
    q_keep = (10 if pr > 10 else pr) * 0.1
    out = np.empty_like(scale)
    for b in range(scale.shape[0]):
        thr = quantile_threshold_np(scale[b], q_keep)
        with np.errstate(invalid="ignore"):
            out[b] = (scale[b] >= thr).astype(np.float32)
    return out


def variance_mask(scale: Tensor, pr: float) -> Tensor:
    return torch.from_numpy(variance_mask_np(scale.detach().cpu().numpy(), pr))


def prog_mask_np(blocks: Sequence[np.ndarray], pr: float) -> np.ndarray:
    """``ChannelMask.ProgMask`` (channel_mask.py:18-49): list of [1,C,h,w]
    blocks -> [len, C, h, w]."""
    res = []
    prf = (10 if pr > 10 else pr) * 0.1
    for blk in blocks:
        blk = np.asarray(blk, dtype=np.float32)
        if prf >= 1:
            res.append(np.ones_like(blk[0]))
        elif prf == 0:
            res.append(np.zeros_like(blk[0]))
        else:
            thr = quantile_threshold_np(blk[0], prf)
            with np.errstate(invalid="ignore"):
                res.append((blk[0] >= thr).astype(np.float32))
    return np.stack(res)


# --------------------------------------------------------------------------
# A.4 / A.5  Gaussian conditional  (entropy_models.py:573-576, 620-659)
# --------------------------------------------------------------------------
def scale_table() -> Tensor:
    """models/pic.py:17-18."""
    return torch.exp(torch.linspace(math.log(SCALES_MIN), math.log(SCALES_MAX), SCALES_LEVELS))


def _phi_c(t: Tensor) -> Tensor:
    return 0.5 * torch.erfc(float(-(2 ** -0.5)) * t)   # entropy_models.py:573-576


def gaussian_likelihood(inputs: Tensor, scales: Tensor, means: Optional[Tensor]) -> Tensor:
    """Eval-mode ``GaussianConditional.forward`` likelihood
    (entropy_models.py:637-652 with quantize "dequantize" :140-149)."""
    if means is not None:
        outputs = torch.round(inputs - means) + means
        values = outputs - means
    else:
        values = torch.round(inputs)
    s = torch.clamp_min(scales, SCALE_BOUND)
    values = values.abs()
    lik = _phi_c((0.5 - values) / s) - _phi_c((-0.5 - values) / s)
    return torch.clamp_min(lik, LIKELIHOOD_BOUND)


def build_indexes(scales: Tensor, table: Optional[Tensor] = None) -> Tensor:
    """entropy_models.py:654-659."""
    table = scale_table() if table is None else table
    s = torch.clamp_min(scales, SCALE_BOUND)
    idx = torch.full(s.shape, len(table) - 1, dtype=torch.int32)
    for t in table[:-1]:
        idx -= (s <= t).int()
    return idx


# --------------------------------------------------------------------------
# A.6  factorised prior for z  (entropy_models.py:403-436, 449-492)
# --------------------------------------------------------------------------
def eb_logits_cumulative(sd: SD, v: Tensor, prefix: str = "entropy_bottleneck.") -> Tensor:
    logits = v
    for i in range(5):
        logits = torch.matmul(F.softplus(sd[f"{prefix}_matrix{i}"]), logits)
        logits = logits + sd[f"{prefix}_bias{i}"]
        if i < 4:
            logits = logits + torch.tanh(sd[f"{prefix}_factor{i}"]) * torch.tanh(logits)
    return logits


def eb_forward(sd: SD, z: Tensor, prefix: str = "entropy_bottleneck."):
    """Eval-mode ``EntropyBottleneck.forward``: returns (z_hat, likelihood)."""
    B, C = z.shape[:2]
    med = sd[f"{prefix}quantiles"][:, :, 1:2]              # [C,1,1]
    vals = z.transpose(0, 1).contiguous().reshape(C, 1, -1)
    out = torch.round(vals - med) + med
    lower = eb_logits_cumulative(sd, out - 0.5, prefix)
    upper = eb_logits_cumulative(sd, out + 0.5, prefix)
    sign = -torch.sign(lower + upper)
    lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
    lik = torch.clamp_min(lik, LIKELIHOOD_BOUND)
    shp = (C, B) + tuple(z.shape[2:])
    return (out.reshape(shp).transpose(0, 1).contiguous(),
            lik.reshape(shp).transpose(0, 1).contiguous())


# --------------------------------------------------------------------------
# layers  (layers/layers.py, layers/gdn.py, layers/win_attention.py, layers/rem.py)
# --------------------------------------------------------------------------
# ---- bf16-storage emulation (BASELINE configs[2] "bf16"; the HIP path's ``model.storage = "bf16"``): tensors of at
# least _BF16_MIN_HW positions per image inside g_a / g_s are rounded to bf16 when they are stored, and a layer that
# reads or writes such a tensor multiplies bf16-rounded inputs by bf16-rounded weights with fp32 accumulation.  None
# (default) = the fp32 reference path, which is what every parity claim refers to.
_BF16_MIN_HW: Optional[int] = None


class bf16_storage:
    def __init__(self, min_hw: int = 4096):
        self.min_hw = min_hw

    def __enter__(self):
        global _BF16_MIN_HW
        self._old, _BF16_MIN_HW = _BF16_MIN_HW, self.min_hw

    def __exit__(self, *a):
        global _BF16_MIN_HW
        _BF16_MIN_HW = self._old


def _q(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def _big(h: int, w: int) -> bool:
    return _BF16_MIN_HW is not None and h * w >= _BF16_MIN_HW


def _st(t: Tensor) -> Tensor:
    """Store rounding of an intermediate feature map."""
    return _q(t) if _big(t.shape[-2], t.shape[-1]) else t


def nonneg(p: Tensor, minimum: float) -> Tensor:
    """compressai NonNegativeParametrizer.forward."""
    pedestal = torch.tensor([(2.0 ** -18) ** 2], dtype=torch.float32)
    # LowerBound (not a plain max): same value, and compressai's gradient rule when the parameters train (A.3)
    return lower_bound(p, (minimum + (2.0 ** -18) ** 2) ** 0.5) ** 2 - pedestal


def gdn(sd: SD, pre: str, x: Tensor, inverse: bool) -> Tensor:
    """layers/gdn.py:62-75."""
    C = x.shape[1]
    beta = nonneg(sd[pre + "beta"], 1e-6)
    gamma = nonneg(sd[pre + "gamma"], 0.0).reshape(C, C, 1, 1)
    if _big(x.shape[-2], x.shape[-1]):
        norm = F.conv2d(_q(x ** 2), _q(gamma), beta)
    else:
        norm = F.conv2d(x ** 2, gamma, beta)
    norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
    return _st(x * norm)


def conv_k(sd: SD, pre: str, x: Tensor, stride: int = 1) -> Tensor:
    w = sd[pre + "weight"]
    H, W = x.shape[-2:]
    if _big(H, W) or _big(-(-H // stride), -(-W // stride)):
        x, w = _q(x), _q(w)
    return F.conv2d(x, w, sd[pre + "bias"], stride=stride, padding=w.shape[-1] // 2)


def deconv_k(sd: SD, pre: str, x: Tensor) -> Tensor:
    """layers/layers.py:14-22 (k5 s2 pad2 outpad1)."""
    w = sd[pre + "weight"]
    k = w.shape[-1]
    if _big(2 * x.shape[-2], 2 * x.shape[-1]):
        x, w = _q(x), _q(w)
    return F.conv_transpose2d(x, w, sd[pre + "bias"], stride=2, padding=k // 2, output_padding=1)


def residual_unit(sd: SD, pre: str, x: Tensor) -> Tensor:
    """layers/layers.py:30-48."""
    o = _st(F.gelu(conv_k(sd, pre + "conv.0.", x)))
    o = _st(F.gelu(conv_k(sd, pre + "conv.2.", o)))
    o = conv_k(sd, pre + "conv.4.", o)
    return _st(F.gelu(o + x))


def _rel_pos_index(ws: int) -> Tensor:
    """win_attention.py:63-72 (own derivation: index = (dy+ws-1)*(2ws-1) + (dx+ws-1))."""
    ys, xs = np.divmod(np.arange(ws * ws), ws)
    dy = ys[:, None] - ys[None, :] + ws - 1
    dx = xs[:, None] - xs[None, :] + ws - 1
    return torch.from_numpy(dy * (2 * ws - 1) + dx).long()


def _shift_mask(H: int, W: int, ws: int, shift: int) -> Tensor:
    """win_attention.py:161-177: region ids on the *shifted* grid, 0 / -100."""
    def ids(n):
        r = np.zeros(n, dtype=np.int64)
        r[n - ws:n - shift] = 1
        r[n - shift:] = 2
        return r
    img = ids(H)[:, None] * 3 + ids(W)[None, :]
    win = img.reshape(H // ws, ws, W // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = win[:, None, :] - win[:, :, None]
    return torch.from_numpy(np.where(diff != 0, -100.0, 0.0).astype(np.float32))


def win_attention(sd: SD, pre: str, x: Tensor, ws: int, shift: int) -> Tensor:
    """``WinBasedAttention.forward`` (win_attention.py:153-207) + ``WindowAttention``
    (:84-115).  x: [B,C,H,W] -> shortcut + attention."""
    B, C, H, W = x.shape
    hd = C // NUM_HEADS
    t = x.permute(0, 2, 3, 1)
    if shift > 0:
        t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
    win = t.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    Bw, N, _ = win.shape
    big = _big(H, W)                                   # bf16-storage emulation: x is bf16-stored, q/k/v and the attention output fp32
    qkv = F.linear(_q(win) if big else win, _q(sd[pre + "attn.qkv.weight"]) if big else sd[pre + "attn.qkv.weight"],
                   sd[pre + "attn.qkv.bias"])
    qkv = qkv.reshape(Bw, N, 3, NUM_HEADS, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[pre + "attn.relative_position_bias_table"][_rel_pos_index(ws).reshape(-1)]
    attn = attn + bias.reshape(N, N, NUM_HEADS).permute(2, 0, 1).unsqueeze(0)
    if shift > 0:
        m = _shift_mask(H, W, ws, shift)
        nW = m.shape[0]
        attn = attn.reshape(Bw // nW, nW, NUM_HEADS, N, N) + m[None, :, None]
        attn = attn.reshape(-1, NUM_HEADS, N, N)
    attn = torch.softmax(attn, dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(Bw, N, C)
    o = F.linear(_q(o) if big else o, _q(sd[pre + "attn.proj.weight"]) if big else sd[pre + "attn.proj.weight"],
                 sd[pre + "attn.proj.bias"])
    o = o.reshape(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return _st(x + o.permute(0, 3, 1, 2))


def attention_block(sd: SD, pre: str, x: Tensor, ws: int) -> Tensor:
    """``Win_noShift_Attention`` (layers/layers.py:50-74); builder passes shift=ws/2."""
    a = x
    for i in range(3):
        a = residual_unit(sd, f"{pre}conv_a.{i}.", a)
    b = win_attention(sd, pre + "conv_b.0.", x, ws, ws // 2)
    for i in (1, 2, 3):
        b = residual_unit(sd, f"{pre}conv_b.{i}.", b)
    b = conv_k(sd, pre + "conv_b.4.", b)
    out = a * torch.sigmoid(b)
    return _st(out + x)


def g_a(sd: SD, pre: str, x: Tensor) -> Tensor:
    """models/builder.py:43-53."""
    x = gdn(sd, pre + "1.", _st(conv_k(sd, pre + "0.", x, 2)), False)
    x = gdn(sd, pre + "3.", _st(conv_k(sd, pre + "2.", x, 2)), False)
    x = attention_block(sd, pre + "4.", x, 8)
    x = gdn(sd, pre + "6.", _st(conv_k(sd, pre + "5.", x, 2)), False)
    x = _st(conv_k(sd, pre + "7.", x, 2))
    return attention_block(sd, pre + "8.", x, 4)


def g_s(sd: SD, pre: str, y: Tensor) -> Tensor:
    """models/builder.py:8-18."""
    y = attention_block(sd, pre + "0.", y, 4)
    y = gdn(sd, pre + "2.", _st(deconv_k(sd, pre + "1.", y)), True)
    y = gdn(sd, pre + "4.", _st(deconv_k(sd, pre + "3.", y)), True)
    y = attention_block(sd, pre + "5.", y, 8)
    y = gdn(sd, pre + "7.", _st(deconv_k(sd, pre + "6.", y)), True)
    return deconv_k(sd, pre + "8.", y)                 # x_hat leaves in fp32


def h_a(sd: SD, y: Tensor) -> Tensor:
    """models/builder.py:72-82."""
    for i, s in ((0, 1), (2, 1), (4, 2), (6, 1)):
        y = F.gelu(conv_k(sd, f"h_a.{i}.", y, s))
    return conv_k(sd, "h_a.8.", y, 2)


def h_s(sd: SD, pre: str, z: Tensor) -> Tensor:
    """models/builder.py:88-98 (subpel = conv3x3 -> PixelShuffle(2))."""
    z = F.gelu(conv_k(sd, pre + "0.", z))
    z = F.gelu(F.pixel_shuffle(conv_k(sd, pre + "2.0.", z), 2))
    z = F.gelu(conv_k(sd, pre + "4.", z))
    z = F.gelu(F.pixel_shuffle(conv_k(sd, pre + "6.0.", z), 2))
    return conv_k(sd, pre + "8.", z)


def cc_stack(sd: SD, pre: str, x: Tensor) -> Tensor:
    """models/pic.py:83-164: five conv3x3 with GELU between."""
    for i in (0, 2, 4, 6):
        x = F.gelu(conv_k(sd, f"{pre}{i}.", x))
    return conv_k(sd, pre + "8.", x)


def rem_resblock(sd: SD, pre: str, x: Tensor) -> Tensor:
    """layers/rem.py:37-66."""
    o = F.leaky_relu(conv_k(sd, pre + "conv1.", x), 0.01)
    o = F.leaky_relu(conv_k(sd, pre + "conv2.", o), 0.01)
    idn = conv_k(sd, pre + "skip.", x) if (pre + "skip.weight") in sd else x
    return o + idn


def rem_block(sd: SD, pre: str, y_ck: Tensor, ep_base: Tensor, ep_prog: Tensor, att: Tensor) -> Tensor:
    """``LatentRateReduction.forward`` (layers/rem.py:130-141)."""
    def seq(name, t):
        i = 0
        while f"{pre}{name}.{i}.conv1.weight" in sd:
            t = rem_resblock(sd, f"{pre}{name}.{i}.", t)
            i += 1
        return t
    f_lat = seq("enc_base_rep", y_ck)
    f_prog = seq("enc_progressive_entropy_params", ep_prog)
    f_base = seq("enc_base_entropy_params", ep_base)
    ret = seq("enc", torch.cat([f_lat, f_base, f_prog], dim=1))
    return ep_prog + ret * att


# --------------------------------------------------------------------------
# model data flow  (models/pic.py:278-298, 497-666; models/rem_pic.py:142-220, 229-422)
# --------------------------------------------------------------------------
def compute_hyperprior(sd: SD, y: Tensor, quality: float, multiple_hyperprior: bool = True):
    """models/pic.py:278-298."""
    z = h_a(sd, y)
    z_hat, z_lik = eb_forward(sd, z)
    if not multiple_hyperprior:                       # one synthesis pair with M output channels
        return h_s(sd, "h_mean_s.", z_hat), h_s(sd, "h_scale_s.", z_hat), z_lik, z_hat
    if quality == 0:
        return h_s(sd, "h_mean_s.0.", z_hat), h_s(sd, "h_scale_s.0.", z_hat), z_lik, z_hat
    means = torch.cat([h_s(sd, "h_mean_s.0.", z_hat), h_s(sd, "h_mean_s.1.", z_hat)], 1)
    scales = torch.cat([h_s(sd, "h_scale_s.0.", z_hat), h_s(sd, "h_scale_s.1.", z_hat)], 1)
    return means, scales, z_lik, z_hat


def find_check_quality(check_levels: Sequence[float], quality: float):
    """models/rem_pic.py:142-165."""
    cl = list(check_levels)
    if quality <= cl[0]:
        return 0, 0, -1
    if len(cl) in (2, 3) and cl[0] < quality <= cl[1]:
        return cl[0], cl[1], 0
    if len(cl) == 2 and quality > cl[1]:
        return cl[1], 10, 1
    if len(cl) == 3 and cl[1] < quality <= cl[2]:
        return cl[1], cl[-1], 1
    return cl[-1], 10, -1


def rem_index(check_levels: Sequence[float], quality: float) -> int:
    """models/rem_pic.py:200-213."""
    cl = list(check_levels)
    if len(cl) == 1:
        return 0
    if len(cl) == 2:
        return 0 if cl[0] < quality <= cl[1] else 1
    if cl[0] < quality <= cl[1]:
        return 0
    if cl[1] < quality <= cl[2]:
        return 1
    return 2


def forward_single_quality(sd: SD, x: Tensor, quality: float, *, div: int = 320, chunk: int = 32,
                           max_support: int = 5, prog_support: int = 5,
                           checkpoint_ref: Optional[Tensor] = None,
                           check_levels: Optional[Sequence[float]] = None,
                           mu_std: bool = True, multiple_encoder: bool = True, multiple_decoder: bool = True,
                           multiple_hyperprior: bool = True, delta_encode: bool = True, total_mu_rep: bool = True,
                           all_scalable: bool = True) -> dict:
    """``VarianceMaskingPIC.forward_single_quality`` (models/pic.py:497-666) and, when
    ``check_levels`` is given, ``VarianceMaskingPICREM.forward`` in eval mode
    (models/rem_pic.py:229-422).  The keyword flags are the constructor's (models/__init__.py:11-55,
    pic.py:27-42); their defaults are the README configuration.
    """
    rem = check_levels is not None
    with torch.no_grad():
        if multiple_encoder:                                              # pic.py:501-508
            y = torch.cat([g_a(sd, "g_a.0.", x), g_a(sd, "g_a.1.", x)], 1)
        else:
            y = g_a(sd, "g_a.", x)
        means_h, scales_h, z_lik, z_hat = compute_hyperprior(sd, y, quality, multiple_hyperprior)
        ns0 = div // chunk
        ys = y.chunk(y.shape[1] // chunk, 1)
        yhat_b: List[Tensor] = []
        lik: List[Tensor] = []
        mu_b, std_b = [], []
        for i in range(ns0):
            sup = yhat_b[:min(max_support, i)]
            msup = torch.cat([means_h[:, :div]] + sup, 1)
            ssup = torch.cat([scales_h[:, :div]] + sup, 1)
            mu = cc_stack(sd, f"cc_mean_transforms.{i}.", msup)
            sc = cc_stack(sd, f"cc_scale_transforms.{i}.", ssup)
            mu_b.append(mu)
            std_b.append(sc)
            lik.append(gaussian_likelihood(ys[i], sc, mu))
            yh = torch.round(ys[i] - mu) + mu
            lrp = cc_stack(sd, f"lrp_transforms.{i}.", torch.cat([msup, yh], 1))
            yhat_b.append(yh + 0.5 * torch.tanh(lrp))
        y_base = torch.cat(yhat_b, 1)
        if quality == 0:
            x_hat = g_s(sd, "g_s.0." if multiple_decoder else "g_s.", y_base).clamp(0, 1)
            return {"x_hat": x_hat, "likelihoods": {"y": torch.cat(lik, 1), "z": z_lik},
                    "y_hat": y_base, "y_base": y_base, "mu_base": torch.cat(mu_b, 1),
                    "std_base": torch.cat(std_b, 1), "z_hat": z_hat, "y": y}

        ck = checkpoint_ref.chunk(10, 1) if checkpoint_ref is not None else None
        mu_tot, std_tot, mu_p, std_p, masks, yhat_p, atts = [], [], [], [], [], [], []
        n_prog = y.shape[1] // chunk - ns0
        for j in range(n_prog):
            r = ys[ns0 + j] - ys[j] if delta_encode else ys[ns0 + j]      # pic.py:583-584
            # determine_support (pic.py:264-270): base slice j + the last min(sp, j) entries of the support vectors
            s = 0 if prog_support == 0 else min(prog_support, j)
            sv_m = mu_tot if all_scalable else yhat_p                     # pic.py:586-587
            sv_s = std_tot if all_scalable else yhat_p
            msup = torch.cat([means_h[:, div:], yhat_b[j]] + sv_m[j - s:j], 1)
            ssup = torch.cat([scales_h[:, div:], yhat_b[j]] + sv_s[j - s:j], 1)
            mu = cc_stack(sd, f"cc_mean_transforms_prog.{j}.", msup)
            sc = cc_stack(sd, f"cc_scale_transforms_prog.{j}.", ssup)
            mu_tot.append(mu + yhat_b[j] if total_mu_rep else mu)         # pic.py:601
            std_tot.append(sc)
            if rem and ck is not None and quality > check_levels[0]:
                att = variance_mask(sc, quality)                      # rem_pic.py:185-192
                atts.append(att)
                if mu_std:
                    att = torch.cat([att, att], 1)
                ri = rem_index(check_levels, quality)
                ep = rem_block(sd, f"post_latent.{ri}.{j}.", ck[j],
                               torch.cat([mu_b[j], std_b[j]], 1),
                               torch.cat([mu, sc], 1) if mu_std else sc, att)
                if mu_std:
                    mu, sc = ep.chunk(2, 1)
                else:
                    sc = ep
            mu_p.append(mu)
            std_p.append(sc)
            m = variance_mask(sc, quality)                               # pic.py:621-622
            masks.append(m)
            lik.append(gaussian_likelihood((r - mu) * m, sc * m, None))  # pic.py:625-628
            rh = torch.round(r - mu) * m + mu                            # pic.py:629
            lrp = cc_stack(sd, f"lrp_transforms_prog.{j}.", torch.cat([msup, rh], 1))
            rh = rh + 0.5 * torch.tanh(lrp)
            yhat_p.append(rh + yhat_b[j])
        y_prog = torch.cat(yhat_p, 1)
        x_hat = g_s(sd, "g_s.1." if multiple_decoder else "g_s.", y_prog).clamp(0, 1)
        res = {"x_hat": x_hat, "likelihoods": {"y": torch.cat(lik, 1), "z": z_lik},
               "y_hat": y_prog, "y_base": y_base, "y_prog": y_prog,
               "mu_base": torch.cat(mu_b, 1), "mu": torch.cat(mu_p, 1),
               "std_base": torch.cat(std_b, 1), "std": torch.cat(std_p, 1),
               "mask": torch.cat(masks, 1), "z_hat": z_hat, "y": y}
        if atts:          # REM: the attention mask (a threshold decision on the UN-refined sigma, rem_pic.py:185-192) and that sigma
            res["att"], res["std_raw"] = torch.cat(atts, 1), torch.cat(std_tot, 1)
        return res


# --------------------------------------------------------------------------
# A.8 metrics  (utility/functions.py:172-174, training/loss.py:217-228)
# --------------------------------------------------------------------------
def psnr(a: Tensor, b: Tensor) -> float:
    mse = F.mse_loss(a, b).item()
    return -10 * math.log10(mse)


def bpp(likelihoods: dict, num_pixels: int) -> float:
    """``RateLoss`` style: sum over y and z of log(lik) / (-ln2 * B*H*W)."""
    tot = 0.0
    for v in likelihoods.values():
        tot += torch.log(v.double()).sum().item() / (-math.log(2) * num_pixels)
    return tot


def log2_sum_per_image(lik: Tensor) -> Tensor:
    return torch.log2(lik.double()).flatten(1).sum(1)


# --------------------------------------------------------------------------
# A.9 REM fine-tune step (BASELINE configs[4]): training-mode forward + autograd
#     (models/rem_pic.py:229-422 with training=True, training/step.py:56-95, training/loss.py:189-229)
# --------------------------------------------------------------------------
class _LowerBoundFn(torch.autograd.Function):
    """compressai.ops.LowerBound (1.2.4, published definition): max(x, bound); the gradient passes where
    x >= bound or where it would push x up (grad < 0)."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        return ((x >= bound) | (g < 0)).to(g.dtype) * g, None


def lower_bound(x: Tensor, bound: float) -> Tensor:
    return _LowerBoundFn.apply(x, torch.tensor([float(bound)]))


def gaussian_likelihood_noise(inputs: Tensor, scales: Tensor, means: Optional[Tensor], noise: Tensor) -> Tensor:
    """Training-mode ``GaussianConditional.forward`` (entropy_models.py:637-652 with quantize "noise" :132-138)."""
    outputs = inputs + noise
    values = outputs - means if means is not None else outputs
    s = lower_bound(scales, SCALE_BOUND)
    values = values.abs()
    lik = _phi_c((0.5 - values) / s) - _phi_c((-0.5 - values) / s)
    return lower_bound(lik, LIKELIHOOD_BOUND)


def eb_aux_loss(sd: SD, prefix: str = "entropy_bottleneck.") -> Tensor:
    """``EntropyBottleneck.loss`` (entropy_models.py:398-401): the density network's parameters are constants
    (stop_gradient=True), only ``quantiles`` carries a gradient."""
    const = {k: (v if k.endswith("quantiles") else v.detach()) for k, v in sd.items() if k.startswith(prefix)}
    logits = eb_logits_cumulative(const, sd[prefix + "quantiles"], prefix)
    target = math.log(2 / 1e-9 - 1)
    return torch.abs(logits - torch.tensor([-target, 0.0, target])).sum()


def eb_likelihood_noise(sd: SD, z: Tensor, noise: Tensor, prefix: str = "entropy_bottleneck.") -> Tensor:
    """Training-mode ``EntropyBottleneck.forward`` likelihood: evaluated at z + noise (entropy_models.py:471-478)."""
    B, C = z.shape[:2]
    out = (z + noise).transpose(0, 1).contiguous().reshape(C, 1, -1)
    lower = eb_logits_cumulative(sd, out - 0.5, prefix)
    upper = eb_logits_cumulative(sd, out + 0.5, prefix)
    sign = -torch.sign(lower + upper)
    lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
    lik = torch.clamp_min(lik, LIKELIHOOD_BOUND)
    return lik.reshape((C, B) + tuple(z.shape[2:])).transpose(0, 1).contiguous()


def rem_training_step(sd: SD, x: Tensor, quality: float, checkpoint_ref: Tensor, noise_y: Tensor, noise_z: Tensor, *,
                      check_levels: Sequence[float], div: int = 320, chunk: int = 32, max_support: int = 5,
                      prog_support: int = 5, mu_std: bool = True) -> dict:
    """One REM fine-tune step's forward + backward: returns the training-mode likelihoods, RateLoss's bpp and
    dLoss/d(post_latent.<r>.*) by autograd.  Everything outside the REM is frozen (train.py:223-226), so it is
    evaluated without a graph; masks carry no gradient (hard comparison).  ``mu_std`` = False: the block sees and refines
    the scale only (rem_pic.py:194-195,214-220; layers/rem.py:86,100)."""
    assert quality > check_levels[0]
    ri = rem_index(check_levels, quality)
    pre = f"post_latent.{ri}."
    names = [k for k in sd if k.startswith(pre)]
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in names}
    sdt = dict(sd)
    sdt.update(leaves)
    ns0 = div // chunk
    with torch.no_grad():
        y = torch.cat([g_a(sd, "g_a.0.", x), g_a(sd, "g_a.1.", x)], 1)
        z = h_a(sd, y)
        means_h, scales_h, _, z_hat = compute_hyperprior(sd, y, quality)
        z_lik = eb_likelihood_noise(sd, z, noise_z)
        ys = y.chunk(y.shape[1] // chunk, 1)
        nys = noise_y.chunk(y.shape[1] // chunk, 1)
        yhat_b, lik, mu_b, std_b = [], [], [], []
        for i in range(ns0):
            sup = yhat_b[:min(max_support, i)]
            msup = torch.cat([means_h[:, :div]] + sup, 1)
            ssup = torch.cat([scales_h[:, :div]] + sup, 1)
            mu = cc_stack(sd, f"cc_mean_transforms.{i}.", msup)
            sc = cc_stack(sd, f"cc_scale_transforms.{i}.", ssup)
            mu_b.append(mu)
            std_b.append(sc)
            lik.append(gaussian_likelihood_noise(ys[i], sc, mu, nys[i]))
            yh = torch.round(ys[i] - mu) + mu
            lrp = cc_stack(sd, f"lrp_transforms.{i}.", torch.cat([msup, yh], 1))
            yhat_b.append(yh + 0.5 * torch.tanh(lrp))
    ck = checkpoint_ref.chunk(10, 1)
    mu_tot, std_tot, masks, mu_f, std_f = [], [], [], [], []
    for j in range(ns0):
        with torch.no_grad():
            r = ys[ns0 + j] - ys[j]
            s = min(prog_support, j)
            msup = torch.cat([means_h[:, div:], yhat_b[j]] + mu_tot[j - s:j], 1)
            ssup = torch.cat([scales_h[:, div:], yhat_b[j]] + std_tot[j - s:j], 1)
            mu = cc_stack(sd, f"cc_mean_transforms_prog.{j}.", msup)
            sc = cc_stack(sd, f"cc_scale_transforms_prog.{j}.", ssup)
            mu_tot.append(mu + yhat_b[j])
            std_tot.append(sc)
            att = variance_mask(sc, quality)
            if mu_std:
                att = torch.cat([att, att], 1)
        if mu_std:
            ep = rem_block(sdt, f"{pre}{j}.", ck[j], torch.cat([mu_b[j], std_b[j]], 1), torch.cat([mu, sc], 1), att)
            mu2, sc2 = ep.chunk(2, 1)
        else:
            mu2, sc2 = mu, rem_block(sdt, f"{pre}{j}.", ck[j], torch.cat([mu_b[j], std_b[j]], 1), sc, att)
        m = variance_mask(sc2.detach(), quality)
        masks.append(m)
        mu_f.append(mu2)
        std_f.append(sc2)
        lik.append(gaussian_likelihood_noise((r - mu2) * m, sc2 * m, None, nys[ns0 + j]))
    lik_y = torch.cat(lik, 1)
    num_pixels = x.shape[0] * x.shape[2] * x.shape[3]
    den = -math.log(2) * num_pixels
    bpp_y = torch.log(lik_y).sum() / den
    bpp_z = torch.log(z_lik).sum() / den
    loss = bpp_y + bpp_z
    loss.backward()
    return {"likelihoods": {"y": lik_y.detach(), "z": z_lik}, "loss": float(loss.detach()),
            "grads": {k: v.grad for k, v in leaves.items()}, "mask": torch.cat(masks, 1),
            "mu": torch.cat(mu_f, 1).detach(), "std": torch.cat(std_f, 1).detach()}


# --------------------------------------------------------------------------
# A.10 decoder refinement step (train.py:150-157,216-218 `--training_type refine_gs`): everything frozen except the
#      synthesis transform of the progressive decoder; DistortionLoss (training/loss.py:126-187)
# --------------------------------------------------------------------------
def refine_gs_training_step(sd: SD, x: Tensor, quality: float, *, lmbda: float = 1e-2, weight: float = 255.0 ** 2,
                            prefix: str = "g_s.1.", lrp: bool = False, **flags):
    """One ``refine_gs`` step: the frozen front end gives y_hat (its STE-rounded values equal the eval pass, pic.py:629),
    x_hat = g_s[1](y_hat).clamp(0, 1), loss = weight * lmbda * mean((x - x_hat)^2).  Returns (loss, mse, x_hat,
    {name: gradient}) for the parameters under ``prefix`` — and, with ``lrp`` (``unfreeze_decoder(lrp=True)``,
    pic.py:171-184), under ``lrp_transforms_prog.``: y_hat_j = rq_j + 0.5 tanh(lrp_j(cat(mean support, rq_j))) + y_base_j
    (pic.py:629-641) is then recomputed under autograd from the front end's (frozen) tensors; README flags only."""
    ref = forward_single_quality(sd, x, quality, **flags)
    trained = (prefix,) + (("lrp_transforms_prog.",) if lrp else ())
    leaves = {k: (v.clone().requires_grad_(True) if k.startswith(trained) and v.dtype.is_floating_point else v)
              for k, v in sd.items()}
    if lrp:
        assert not flags, "the LRP recomputation follows the README configuration"
        div, chunk, sp = 320, 32, 5
        y, yb, mu, mask = ref["y"], ref["y_base"], ref["mu"], ref["mask"]
        means_h = compute_hyperprior(sd, y, quality)[0]
        mu_tot = mu + yb                                                  # pic.py:601
        sl = lambda t, j, n=1: t[:, j * chunk:(j + n) * chunk]
        parts = []
        for j in range(div // chunk):
            s = min(sp, j)
            msup = torch.cat([means_h[:, div:], sl(yb, j)] + ([sl(mu_tot, j - s, s)] if s else []), 1)
            r = sl(y, div // chunk + j) - sl(y, j)
            rh = torch.round(r - sl(mu, j)) * sl(mask, j) + sl(mu, j)
            parts.append(rh + 0.5 * torch.tanh(cc_stack(leaves, f"lrp_transforms_prog.{j}.", torch.cat([msup, rh], 1))) + sl(yb, j))
        y_hat = torch.cat(parts, 1)
        assert torch.equal(y_hat.detach(), ref["y_hat"]), "LRP recomputation must reproduce the forward pass"
    else:
        y_hat = ref["y_hat"].detach()
    x_hat = g_s(leaves, prefix, y_hat).clamp(0, 1)
    mse = F.mse_loss(x, x_hat)
    loss = weight * (lmbda * mse)
    loss.backward()
    grads = {(k[len(prefix):] if k.startswith(prefix) else k): v.grad for k, v in leaves.items()
             if k.startswith(trained) and torch.is_tensor(v) and v.grad is not None}
    return loss.detach(), mse.detach(), x_hat.detach(), grads


# --------------------------------------------------------------------------
# A.11 first-stage training step (BASELINE configs[3], `--training_type first_train`, train.py:146-149):
#      ``VarianceMaskingPIC.forward(x, quality=[0, q], training=True)`` (models/pic.py:301-491) or
#      ``forward_single_quality(x, q, training=True)`` (:497-666) with EVERY parameter trainable,
#      ``ScalableRateDistortionLoss`` (training/loss.py:6-66), autograd over this restatement.
# --------------------------------------------------------------------------
def ste_round(x: Tensor) -> Tensor:
    """models/utils.py:4-5: value round(x), gradient identity."""
    return torch.round(x) - x.detach() + x


def _ste_forced(x: Tensor, q: Optional[Tensor]) -> Tensor:
    """ste_round with the rounding DECISION supplied (teacher forcing for parity tests: a latent within fp32 summation
    noise of x.5 may round the other way on another back end, and every later slice is conditioned on it)."""
    return ste_round(x) if q is None else q - x.detach() + x


def training_forward(sd: SD, x: Tensor, qualities: Sequence[float], noise_y: Tensor, noise_z: Tensor, *,
                     single: bool = False, div: int = 320, chunk: int = 32, max_support: int = 5,
                     prog_support: int = 5, force: Optional[dict] = None, multiple_encoder: bool = True,
                     multiple_decoder: bool = True, multiple_hyperprior: bool = True, delta_encode: bool = True,
                     total_mu_rep: bool = True, all_scalable: bool = True) -> dict:
    """Training-mode forward; delta_encode / total_mu_rep / all_scalable (pic.py:397-405,416) default to the README values; the encoder /
    decoder / hyperprior may each be single (pic.py:285-288,306-311,372,462-466: one g_a with M outputs, one g_s used for
    both reconstructions, one synthesis pair with M outputs).  ``single`` = False: ``forward(x, quality=[0, q])`` — both decoders, no clamp, likelihoods
    {"y": base, "y_prog": [1, B, 640] = base AND progressive (pic.py:389-390,471-472), "z"}.  ``single`` = True:
    ``forward_single_quality(x, q)`` — the decoder in use, ``clamp_(0, 1)``, likelihoods {"y", "z"}.  ``noise_y`` [B,640,h,w]
    / ``noise_z`` [B,192,h/4,w/4]: the U(-.5,.5) draws of quantize("noise") (entropy_models.py:132-138).
    ``force`` (tests only) = {"base_sym", "prog_sym": round(y - mu) [B,320,h,w], "mask": [B,320,h,w], "z_sym"}: the hard
    decisions of another run of the same step, imposed instead of recomputed (values AND gradients then agree to
    rounding noise; without it this is the reference's arithmetic, pinned by tests/golden/first_train_step.npz)."""
    force = force or {}
    fsl = lambda key, j: force[key][:, j * chunk:(j + 1) * chunk] if key in force else None
    qs = list(qualities)
    assert (single and len(qs) == 1) or (not single and len(qs) == 2 and qs[0] == 0)
    q = qs[-1]
    base_only = single and q == 0
    y = torch.cat([g_a(sd, "g_a.0.", x), g_a(sd, "g_a.1.", x)], 1) if multiple_encoder else g_a(sd, "g_a.", x)   # pic.py:306-311
    z = h_a(sd, y)                                                                       # :280
    z_lik = eb_likelihood_noise_bounded(sd, z, noise_z)
    med = sd["entropy_bottleneck.quantiles"][:, 0, 1].reshape(1, -1, 1, 1)
    z_hat = _ste_forced(z - med, force.get("z_sym")) + med                               # :282-284
    if not multiple_hyperprior:                                                          # :285-288: one pair, M channels
        means_h, scales_h = h_s(sd, "h_mean_s.", z_hat), h_s(sd, "h_scale_s.", z_hat)
    elif base_only:                                                                      # :285-288 (quality == 0)
        means_h, scales_h = h_s(sd, "h_mean_s.0.", z_hat), h_s(sd, "h_scale_s.0.", z_hat)
    else:
        means_h = torch.cat([h_s(sd, "h_mean_s.0.", z_hat), h_s(sd, "h_mean_s.1.", z_hat)], 1)
        scales_h = torch.cat([h_s(sd, "h_scale_s.0.", z_hat), h_s(sd, "h_scale_s.1.", z_hat)], 1)
    ns0 = div // chunk
    ys = y.chunk(y.shape[1] // chunk, 1)
    nys = noise_y.chunk(noise_y.shape[1] // chunk, 1)
    yhat_b, lik_b, mu_b, std_b = [], [], [], []
    for i in range(ns0):                                                                 # :330-367
        sup = yhat_b[:min(max_support, i)]
        msup = torch.cat([means_h[:, :div]] + sup, 1)
        ssup = torch.cat([scales_h[:, :div]] + sup, 1)
        mu = cc_stack(sd, f"cc_mean_transforms.{i}.", msup)
        sc = cc_stack(sd, f"cc_scale_transforms.{i}.", ssup)
        mu_b.append(mu)
        std_b.append(sc)
        lik_b.append(gaussian_likelihood_noise(ys[i], sc, mu, nys[i]))
        yh = _ste_forced(ys[i] - mu, fsl("base_sym", i)) + mu
        lrp = cc_stack(sd, f"lrp_transforms.{i}.", torch.cat([msup, yh], 1))
        yhat_b.append(yh + 0.5 * torch.tanh(lrp))
    y_base = torch.cat(yhat_b, 1)
    out = {"y_base": y_base, "y": y, "z": z, "mu_base": torch.cat(mu_b, 1), "std_base": torch.cat(std_b, 1)}
    lik_base = torch.cat(lik_b, 1)
    x_hats = []
    if not single or base_only:
        xb = g_s(sd, "g_s.0." if multiple_decoder else "g_s.", y_base)                   # :372
        x_hats.append(xb.clamp(0, 1) if single else xb)
    if base_only:
        out.update({"x_hat": x_hats[0], "likelihoods": {"y": lik_base, "z": z_lik}, "y_hat": y_base})
        return out
    mu_tot, std_tot, lik_p, yhat_p, masks, mu_p, std_p = [], [], [], [], [], [], []
    for j in range(ns0):                                                                 # :396-457
        r = ys[ns0 + j] - ys[j] if delta_encode else ys[ns0 + j]                         # :397-398
        s = min(prog_support, j)
        sup_m, sup_s = (mu_tot, std_tot) if all_scalable else (yhat_p, yhat_p)           # :400-401
        msup = torch.cat([means_h[:, div:], yhat_b[j]] + sup_m[j - s:j], 1)
        ssup = torch.cat([scales_h[:, div:], yhat_b[j]] + sup_s[j - s:j], 1)
        mu = cc_stack(sd, f"cc_mean_transforms_prog.{j}.", msup)
        sc = cc_stack(sd, f"cc_scale_transforms_prog.{j}.", ssup)
        mu_tot.append(mu + yhat_b[j] if total_mu_rep else mu)                            # :416
        std_tot.append(sc)
        mu_p.append(mu)
        std_p.append(sc)
        m = variance_mask(sc.detach(), q) if "mask" not in force else fsl("mask", j)     # hard comparison: no gradient (channel_mask.py:132-151)
        masks.append(m)
        lik_p.append(gaussian_likelihood_noise((r - mu) * m, sc * m, None, nys[ns0 + j]))
        rh = _ste_forced(r - mu, fsl("prog_sym", j)) * m + mu                            # :443
        lrp = cc_stack(sd, f"lrp_transforms_prog.{j}.", torch.cat([msup, rh], 1))
        yhat_p.append(rh + 0.5 * torch.tanh(lrp) + yhat_b[j])
    y_prog = torch.cat(yhat_p, 1)
    xp = g_s(sd, "g_s.1." if multiple_decoder else "g_s.", y_prog)                       # :462-466
    x_hats.append(xp.clamp(0, 1) if single else xp)
    lik_all = torch.cat([lik_base] + lik_p, 1)
    out.update({"y_hat": y_prog, "y_prog": y_prog, "mask": torch.cat(masks, 1), "mu": torch.cat(mu_p, 1),
                "std": torch.cat(std_p, 1)})
    if single:
        out.update({"x_hat": x_hats[0], "likelihoods": {"y": lik_all, "z": z_lik}})
    else:
        out.update({"x_hat": torch.stack(x_hats, 0), "likelihoods": {"y": lik_base, "y_prog": lik_all.unsqueeze(0), "z": z_lik}})
    return out


def eb_likelihood_noise_bounded(sd: SD, z: Tensor, noise: Tensor, prefix: str = "entropy_bottleneck.") -> Tensor:
    """``eb_likelihood_noise`` with the likelihood bound as compressai's LowerBound (gradient rule) instead of a clamp, and
    the sign detached as the reference detaches it (entropy_models.py:431-432,477-478) — same values."""
    B, C = z.shape[:2]
    out = (z + noise).transpose(0, 1).contiguous().reshape(C, 1, -1)
    lower = eb_logits_cumulative(sd, out - 0.5, prefix)
    upper = eb_logits_cumulative(sd, out + 0.5, prefix)
    sign = (-torch.sign(lower + upper)).detach()
    lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
    lik = lower_bound(lik, LIKELIHOOD_BOUND)
    return lik.reshape((C, B) + tuple(z.shape[2:])).transpose(0, 1).contiguous()


def scalable_rd_loss(out: dict, target: Tensor, lmbda, weight: float = 255.0 ** 2) -> dict:
    """``ScalableRateDistortionLoss.forward`` (training/loss.py:17-66), including its accounting: the base rate enters twice
    when "y_prog" is present and ``batch_size_recon`` = x_hat.shape[0] (the number of LEVELS for the stacked
    output of ``forward``, the number of IMAGES for a single-quality output) multiplies the hyperprior rate."""
    B, _, H, W = target.shape
    n_rec = out["x_hat"].shape[0]
    tgt = target.unsqueeze(0)
    if n_rec != 1 and n_rec != B:
        tgt = tgt.repeat(n_rec, 1, 1, 1, 1)
    lm = torch.as_tensor(lmbda, dtype=torch.float32).reshape(-1)
    res = {"mse_loss": F.mse_loss(tgt, out["x_hat"], reduction="none").mean(dim=(1, 2, 3, 4))}
    den = -math.log(2) * B * H * W
    lik = out["likelihoods"]
    res["bpp_hype"] = torch.log(lik["z"]).sum() / den
    if "y_prog" in lik:
        res["bpp_base"] = torch.log(lik["y"]).sum() / den
        res["bpp_scalable"] = torch.log(lik["y_prog"]).sum() / den
    else:
        res["bpp_base"] = torch.log(lik["y"].squeeze(0)).sum() / den
        res["bpp_scalable"] = torch.log(lik["y"]).sum() / den * 0.0
    res["bpp_loss"] = res["bpp_scalable"] + res["bpp_base"] + n_rec * res["bpp_hype"]
    res["loss"] = res["bpp_loss"] + weight * (lm * res["mse_loss"]).mean()
    return res


def first_train_step(sd: SD, x: Tensor, qualities: Sequence[float], noise_y: Tensor, noise_z: Tensor, lmbda, *,
                     single: bool = False, trainable=None, force: Optional[dict] = None, **variant) -> dict:
    """One optimisation step's forward + backward with every floating-point parameter trainable (``trainable``: a
    predicate on the key name; default all).  Returns the forward outputs (detached), the loss terms and
    {name: gradient} — None where autograd left no gradient (a parameter the pass does not use)."""
    skip = ("entropy_bottleneck._offset", "entropy_bottleneck._quantized_cdf", "entropy_bottleneck._cdf_length",
            "gaussian_conditional.")
    leaves = {}
    for k, v in sd.items():
        if torch.is_tensor(v) and v.dtype.is_floating_point and not k.startswith(skip) and "reparam" not in k and \
                not k.endswith((".target", ".bound", ".pedestal")) and not k.startswith("post_latent.") and \
                (trainable is None or trainable(k)):
            leaves[k] = v.detach().clone().requires_grad_(True)
    sdt = dict(sd)
    sdt.update(leaves)
    out = training_forward(sdt, x, qualities, noise_y, noise_z, single=single, force=force, **variant)
    crit = scalable_rd_loss(out, x, lmbda)
    crit["loss"].backward()
    det = lambda t: t.detach() if torch.is_tensor(t) else t
    return {"out": {k: ({kk: det(vv) for kk, vv in v.items()} if isinstance(v, dict) else det(v)) for k, v in out.items()},
            "crit": {k: det(v) for k, v in crit.items()}, "grads": {k: v.grad for k, v in leaves.items()}}
