"""Single-bitstream progressive container (second half): a base layer plus one
rANS layer per quality step, each layer carrying only the latents that the variance mask ADDS
between two consecutive qualities (``ProgMask(q_k) - ProgMask(q_{k-1})``).

Same container and call surface as the reference harness (src/test/functions_encode.py:15-198,
functions_decode.py:9-229, test/utils.py:16-54) so that ``demo.py`` can import these two functions
instead:  ``bitstreams = {"q_list", "shape", "z", "base", "progressive"}``.  Everything numeric goes
through the model's module-level surface (HIP kernels); this file is orchestration only.

Model variants: like the reference's harness this one follows ``multiple_encoder`` (functions_encode.py:78-83),
``multiple_decoder`` (functions_decode.py:107,224), ``multiple_hyperprior`` (test/utils.py:20-31) and any
``support_progressive_slices``; and like it, it is defined for ``delta_encode=True`` (functions_encode.py:113-114
names the residual only there) and ``all_scalable=True`` (the parameter chain of test/utils.py:35-54 never sees a mask).
"""
from __future__ import annotations

import os
import pickle
from typing import List, Optional, Sequence

import torch

Q_LIST = [0.002, 0.05, 0.5, 0.75, 1, 1.5, 2, 2.5, 3, 4, 5, 5.5, 6, 6.6]     # functions_encode.py:11


def _prog_params(model, j, y_base_slices, mu_hist, std_hist, means_h, scales_h):
    """Entropy parameters of progressive slice j from the decoded base latents and the
    parameter history (test/utils.py:35-54)."""
    d = model.division_dimension[0]
    sup_m = model.determine_support(y_base_slices, j, mu_hist)
    sup_s = model.determine_support(y_base_slices, j, std_hist)
    mean_support = torch.cat([means_h[:, d:]] + sup_m, dim=1)
    scale_support = torch.cat([scales_h[:, d:]] + sup_s, dim=1)
    mu = model.cc_mean_transforms_prog[j](mean_support)
    scale = model.cc_scale_transforms_prog[j](scale_support)
    mu_t = mu + y_base_slices[j] if model.total_mu_rep else mu
    return mean_support, mu, mu_t, scale


def _refine(model, j, mu, scale, mu_base, std_base, y_checkpoints):
    """Optional REM refinement with every available checkpoint (functions_encode.py:126-141)."""
    for r, ck in enumerate(y_checkpoints):
        y_b = ck.chunk(10, 1)[j]
        ms_base = torch.cat([mu_base[j], std_base[j]], dim=1)
        ms_prog = torch.cat([mu, scale], dim=1) if model.mu_std else scale
        nxt = model.check_levels[r + 1] if r < model.num_rems - 1 else 10
        mu, scale = model.apply_latent_enhancement(j, model.check_levels[r], nxt, y_b, ms_base, ms_prog, mu, scale)
    return mu, scale


def _chain(model, y_hat_base, mu_base, std_base, means_h, scales_h, rems, y_checkpoints):
    ns = model.ns0
    yb = list(y_hat_base.chunk(ns, 1))
    mb, sb = list(mu_base.chunk(ns, 1)), list(std_base.chunk(ns, 1))
    mu_hist, std_hist, mus, scales, supports = [], [], [], [], []
    for j in range(ns):
        sup, mu, mu_t, scale = _prog_params(model, j, yb, mu_hist, std_hist, means_h, scales_h)
        if rems and y_checkpoints is not None:
            assert len(y_checkpoints) == model.num_rems
            mu, scale = _refine(model, j, mu, scale, mb, sb, y_checkpoints)
        mu_hist.append(mu_t)
        std_hist.append(scale)
        mus.append(mu)
        scales.append(scale)
        supports.append(sup)
    return yb, mus, scales, supports


def _check_variant(model):
    if not (model.delta_encode and getattr(model, "all_scalable", True)):
        raise NotImplementedError("the progressive container is defined for delta_encode=True and all_scalable=True "
                                  "(reference src/test/functions_encode.py:113-114, test/utils.py:35-54)")


def _synthesis(model, i):
    return model.g_s[i] if model.multiple_decoder else model.g_s


def encode(model, x_padded, save_path=None, rems=False, q_list: Sequence[float] = Q_LIST, y_checkpoints=None):
    """functions_encode.py:15-66.  Returns (bitstreams, [bits_z, bits_base, bits_per_layer])."""
    assert x_padded.shape[0] == 1, "the progressive container is per image (ProgMask squeezes batch 1)"
    _check_variant(model)
    with torch.no_grad():
        base = model.compress(x_padded, quality=0)
        bit = {"q_list": list(q_list), "shape": base["shape"], "z": base["strings"][1], "base": base["strings"][0]}
        bits_z = 8.0 * sum(len(s) for s in bit["z"])
        bits_base = 8.0 * sum(len(s[0]) for s in bit["base"])
        # residual latents and their parameters (functions_encode.py:79-160)
        y = torch.cat([model.g_a[0](x_padded), model.g_a[1](x_padded)], dim=1) if model.multiple_encoder else model.g_a(x_padded)
        means_h, scales_h, _ = model.compute_hyperprior(y)
        yb, mus, scales, _ = _chain(model, base["y_hat_base"], base["mean_base"], base["scale_base"], means_h, scales_h,
                                    rems, y_checkpoints)
        ys = y.chunk(model.num_slices, 1)
        gc = model.gaussian_conditional
        r_sym = torch.stack([gc.quantize((ys[model.ns0 + j] - ys[j]) - mus[j], "symbols") for j in range(model.ns0)]).squeeze(1)
        idx = torch.stack([gc.build_indexes(scales[j]).int() for j in range(model.ns0)]).squeeze(1)       # [10,32,h,w]
        layers, bits = [], []
        for k, q in enumerate(q_list):                                   # functions_encode.py:168-196
            q0 = 0 if k == 0 else q_list[k - 1]
            delta = model.masking.ProgMask(scales, q) - model.masking.ProgMask(scales, q0)
            streams = gc.compress(r_sym * delta, idx * delta, already_quantize=True)
            layers.append(streams)
            bits.append(8.0 * sum(len(s) for s in streams))
        bit["progressive"] = layers
    if save_path is not None:
        os.makedirs(save_path, exist_ok=True)
        with open(os.path.join(save_path, "bits.pkl"), "wb") as f:
            pickle.dump(bit, f)
    return bit, [bits_z, bits_base, bits]


def _decode_hyper(model, z_strings, shape, q_ind):
    """test/utils.py:16-33."""
    z_hat = model.entropy_bottleneck.decompress(z_strings, shape)
    if not model.multiple_hyperprior:          # one hyper-synthesis pair delivers the parameters of both halves
        return z_hat, model.h_mean_s(z_hat), model.h_scale_s(z_hat), [z_hat.shape[2] * 4, z_hat.shape[3] * 4]
    means = [model.h_mean_s[0](z_hat)]
    scales = [model.h_scale_s[0](z_hat)]
    if q_ind != 0:
        means.append(model.h_mean_s[1](z_hat))
        scales.append(model.h_scale_s[1](z_hat))
    return z_hat, torch.cat(means, 1), torch.cat(scales, 1), [z_hat.shape[2] * 4, z_hat.shape[3] * 4]


def _decode_base(model, strings, means_h, scales_h):
    """functions_decode.py:9-55."""
    d, gc = model.division_dimension[0], model.gaussian_conditional
    y_hat, mus, scales = [], [], []
    for i in range(model.ns0):
        sup = y_hat[:min(model.max_support_slices, i)]
        m_sup = torch.cat([means_h[:, :d]] + sup, dim=1)
        s_sup = torch.cat([scales_h[:, :d]] + sup, dim=1)
        mu, sc = model.cc_mean_transforms[i](m_sup), model.cc_scale_transforms[i](s_sup)
        rv = gc.decompress(strings[i], gc.build_indexes(sc)).reshape(mu.shape)
        yh = gc.dequantize(rv, mu)
        yh = yh + 0.5 * torch.tanh(model.lrp_transforms[i](torch.cat([m_sup, yh], dim=1)))
        y_hat.append(yh)
        mus.append(mu)
        scales.append(sc)
    return {"y_hat": torch.cat(y_hat, 1), "scale": torch.cat(scales, 1), "mu": torch.cat(mus, 1)}


def decode(model, bitstreams, q_ind=0, res_base=None, index_hat_slice=None, mean=None, z_data=None, entropy_data=None,
           y_checkpoints=None, rems=False):
    """functions_decode.py:58-229: decode the first ``q_ind`` progressive layers (0 = base only).
    ``z_data`` / ``res_base`` / ``entropy_data`` returned by a previous call can be passed back so
    that moving to the next quality only decodes the new layer's parameters once."""
    q_list, shape = bitstreams["q_list"], bitstreams["shape"]
    assert q_ind <= len(q_list)
    _check_variant(model)
    with torch.no_grad():
        if z_data is None:
            z_data = list(_decode_hyper(model, bitstreams["z"], shape, q_ind))
        z_hat, means_h, scales_h, y_shape = z_data
        if res_base is None:
            res_base = _decode_base(model, bitstreams["base"], means_h, scales_h)
        y_hat_base = res_base["y_hat"]
        if q_ind == 0:
            x_hat = _synthesis(model, 0)(y_hat_base).clamp_(0, 1)
            return {"x_hat": x_hat, "y_hat": y_hat_base, "mu": res_base["mu"], "scale": res_base["scale"],
                    "z_data": z_data, "res_base": res_base}
        if model.multiple_hyperprior and means_h.shape[1] == model.division_dimension[0]:     # z_data came from a q_ind == 0 call
            z_data = list(_decode_hyper(model, bitstreams["z"], shape, q_ind))
            z_hat, means_h, scales_h, y_shape = z_data
        gc = model.gaussian_conditional
        if entropy_data is None:
            yb, mus, scales, supports = _chain(model, y_hat_base, res_base["mu"], res_base["scale"], means_h, scales_h,
                                               rems, y_checkpoints)
            idx = torch.stack([gc.build_indexes(s).int() for s in scales]).squeeze(1)
            entropy_data = [torch.cat(mus, 1).squeeze(0), None, supports, scales, idx]
        mean, _, supports, scales, idx = entropy_data
        M, (h, w) = model.division_channel, y_shape
        acc = torch.zeros(M, h, w, device=mean.device)
        for k, q in enumerate(q_list[:q_ind]):                             # functions_decode.py:186-203
            q0 = 0 if k == 0 else q_list[k - 1]
            delta = model.masking.ProgMask(scales, q) - model.masking.ProgMask(scales, q0)
            sym = gc.decompress(bitstreams["progressive"][k], idx * delta)
            acc += sym.reshape(M, h, w) * delta.reshape(M, h, w)
        r_hat = (acc + mean).reshape(1, M, h, w).chunk(model.ns0, 1)
        yb = y_hat_base.chunk(model.ns0, 1)
        y_prog = []
        for j in range(model.ns0):                                         # functions_decode.py:209-220
            r = r_hat[j] + 0.5 * torch.tanh(model.lrp_transforms_prog[j](torch.cat([supports[j], r_hat[j]], dim=1)))
            y_prog.append(model.merge(r, yb[j]))
        y_prog = torch.cat(y_prog, 1)
        x_hat = _synthesis(model, 1)(y_prog)
    return {"x_hat": x_hat, "z_data": z_data, "entropy_data": entropy_data, "y_hat_base": y_hat_base, "y_prog": y_prog,
            "res_base": res_base}
