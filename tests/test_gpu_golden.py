"""HIP path (through the C ABI) against the committed golden vectors produced by the reference."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vampic                      # noqa: E402
import vampic.synth as synth       # noqa: E402
from vampic import layers as Ly    # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
Q_LEVS = [0, 0.01, 0.05, 0.1, 0.25, 0.5, 0.6, 0.75, 1, 1.5, 2, 2.5, 3, 5, 7.7, 9.99, 10, 12]


def test_mask_bits_equal_reference():
    gold = np.load(os.path.join(GOLD, "variance_mask.npz"))
    cm = Ly.ChannelMask("point-based-std")
    for name, (B, C, h, w), seed in (("s8192", (3, 32, 16, 16), 3), ("s49152", (2, 32, 32, 48), 4), ("s480", (2, 32, 5, 3), 5)):
        s = synth.synth_sigma(B, C * h * w, seed=seed).reshape(B, C, h, w).cuda()
        for q in Q_LEVS:
            m = cm(s, pr=q).cpu().numpy().astype(np.uint8)
            assert np.array_equal(np.packbits(m.reshape(-1)), gold[f"{name}_q{q}"]), (name, q)
    blocks = [synth.synth_sigma(1, 32 * 64, seed=20 + i).reshape(1, 32, 8, 8).cuda() for i in range(10)]
    for q in (0, 0.5, 2.5, 10):
        m = cm.ProgMask(blocks, q).cpu().numpy().astype(np.uint8)
        assert np.array_equal(np.packbits(m.reshape(-1)), gold[f"prog_q{q}"])


def test_thresholds_bit_exact():
    from vampic import ops
    gold = np.load(os.path.join(GOLD, "variance_mask.npz"))
    s = synth.synth_sigma(3, 8192, seed=3).reshape(3, 32, 16, 16).cuda()
    v = ops.from_nchw(s)
    for q in (0.01, 0.5, 2.5, 7.7, 9.99):
        m = ops.new_view(v.B, v.H, v.W, v.C)
        thr = torch.empty(3, device="cuda")
        ops.variance_mask(v, q, m, n_slice=1, thr=thr)
        assert np.array_equal(thr.cpu().numpy(), gold[f"s8192_q{q}_thr"]), q


def _check_against_vectors(tag, x, o, gold, scal, n_pix):
    """Strict when no latent sits on a rounding boundary (the normal case); otherwise the
    flip-aware bound of tests/test_gpu_model.py (fp32 summation order)."""
    ref_y = torch.from_numpy(gold[tag + "_y_hat"])
    flips = int((torch.round(o["y_hat"].cpu() - ref_y).abs() >= 1).sum())
    x_err = (o["x_hat"].cpu() - torch.from_numpy(gold[tag + "_x_hat"])).abs().max().item()
    if flips == 0:
        assert x_err <= 1e-4, tag
        if scal is not None:
            mse = torch.nn.functional.mse_loss(x, o["x_hat"].cpu()).item()
            assert abs(-10 * np.log10(mse) - scal[tag]["psnr"]) <= 1e-4, tag
            bpp = -o["log2_likelihood_sum"].sum().item() / n_pix
            assert abs(bpp - scal[tag]["bpp"]) <= 1e-6 * max(1.0, scal[tag]["bpp"]), (tag, bpp, scal[tag]["bpp"])
        return True
    assert flips <= 0.02 * ref_y.numel() and x_err <= 0.5, (tag, flips, x_err)
    return False


def test_forward_matches_reference_vectors(gpu_model):
    net, _ = gpu_model
    gold = np.load(os.path.join(GOLD, "forward_single_quality.npz"))
    scal = json.load(open(os.path.join(GOLD, "forward_single_quality.json")))
    clean = total = 0
    for seed in (0, 1):
        x = synth.synth_image(1, 64, 64, seed=seed)
        for q in (0, 0.5, 2.5, 10):
            with torch.no_grad():
                o = net.forward_single_quality(x.cuda(), q)
            clean += _check_against_vectors(f"s{seed}_q{q}", x, o, gold, scal, 4096)
            total += 1
    x = synth.synth_image(1, 64, 128, seed=0)
    with torch.no_grad():
        o = net.forward_single_quality(x.cuda(), 2.5, checkpoint_ref=torch.from_numpy(gold["rem_ck"]).cuda())
    clean += _check_against_vectors("rem", x, o, gold, None, 8192)
    total += 1
    assert clean * 2 >= total, f"only {clean}/{total} reference vectors reproduced in every rounding decision"
