"""BASELINE.json full sizes (configs[1]: 32x3x256x256; configs[2] shape: 8x3x512x768, run in fp32).

The whole-network CPU oracle is too slow / too chaotic for these sizes, so each NON-conv stage
of the HIP plan is checked "teacher-forced": the oracle recomputes the stage from the GPU's own
inputs of that stage, which removes upstream float noise and makes the integer decisions
comparable bit for bit; conv stacks are spot-checked the same way."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vampic                    # noqa: E402
import vampic_oracle as O        # noqa: E402


def _rms_rel(got, ref):
    return float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


@pytest.mark.parametrize("shape,q,storage", [((32, 256, 256), 2.5, "fp32"), ((8, 512, 768), 0.75, "fp32"), ((32, 256, 256), 9.99, "fp32"),
                                             ((8, 512, 768), 0.75, "bf16")])
def test_full_size_stagewise(gpu_model, shape, q, storage):
    """``storage = "bf16"`` is BASELINE configs[2] as named (Kodak shape, batch 8, bf16): every fp32 stage of that plan —
    variance mask, quantisation, likelihoods, in-kernel bpp, hyperprior and all entropy-parameter / LRP stacks — is held
    to the SAME bounds as in the fp32 configuration, on the bf16 plan's own inputs; the bf16-stored transforms g_a / g_s
    are compared with the oracle's emulation of the same rounding points (O.bf16_storage: both sides round at bf16
    boundaries, single elements may sit an ulp apart: 2 % of the rms)."""
    net, sd = gpu_model
    bf16 = storage == "bf16"
    if bf16:
        import copy
        net = copy.deepcopy(net)
        net.storage = "bf16"
    B, H, W = shape
    x = vampic.synth.synth_image(B, H, W, seed=11).cuda()
    with torch.no_grad():
        out = net.forward_single_quality(x, q)
        plan = [p for k, p in net._plans.items() if k[:4] == (B, H, W, False) and k[4] is None][0]
        y = plan.y.torch_nchw().cpu()
    cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in out.items()}
    cpu["likelihoods"] = {k: v.cpu() for k, v in out["likelihoods"].items()}
    std, mu, mask = cpu["std"], cpu["mu"], cpu["mask"]
    ns = 10
    # (1) variance mask: bit-exact per (image, slice) segment on the GPU's own sigma
    for j in range(ns):
        ref = O.variance_mask_np(std[:, 32 * j:32 * j + 32].numpy(), q)
        got = mask[:, 32 * j:32 * j + 32].numpy()
        assert np.array_equal(ref, got), f"slice {j}: mask XOR {(ref != got).sum()}"
    keep = mask.mean().item()
    assert q / 10 - 1e-3 <= keep <= q / 10 + 0.02, keep          # ties can only add elements
    # (2) quantisation decisions and likelihood from the GPU's (y, mu, sigma, mask)
    r = y[:, 320:] - y[:, :320]
    lik_ref = O.gaussian_likelihood((r - mu) * mask, std * mask, None)
    lik = cpu["likelihoods"]["y"][:, 320:]
    assert (lik - lik_ref).abs().max().item() < 3e-7
    lik_b_ref = O.gaussian_likelihood(y[:, :320], cpu["std_base"], cpu["mu_base"])
    assert (cpu["likelihoods"]["y"][:, :320] - lik_b_ref).abs().max().item() < 3e-7
    # (3) in-kernel bpp accumulation == sum over the likelihood tensors
    bpp_k = -cpu["log2_likelihood_sum"].sum().item() / (B * H * W)
    bpp_t = O.bpp(cpu["likelihoods"], B * H * W)
    assert abs(bpp_k - bpp_t) <= 1e-9 * max(1.0, bpp_t), (bpp_k, bpp_t)
    # (4) conv stack spot checks on the GPU's own inputs (image 0 and last)
    sel = [0, B - 1]
    mh = None
    z_hat = plan.z_hat.torch_nchw().cpu()[sel]
    ref_m = O.h_s(sd, "h_mean_s.1.", z_hat)
    sup = torch.cat([ref_m, cpu["y_base"][sel][:, :32]], 1)
    ref_mu0 = O.cc_stack(sd, "cc_mean_transforms_prog.0.", sup)
    assert (mu[sel][:, :32] - ref_mu0).abs().max().item() <= 3e-4 * max(1.0, ref_mu0.abs().max().item())
    if bf16:
        with O.bf16_storage():
            ref_x = O.g_s(sd, "g_s.1.", cpu["y_hat"][sel]).clamp(0, 1)
        assert _rms_rel(cpu["x_hat"][sel], ref_x) <= 2e-2
    else:
        ref_x = O.g_s(sd, "g_s.1.", cpu["y_hat"][sel]).clamp(0, 1)
        assert (cpu["x_hat"][sel] - ref_x).abs().max().item() <= 1e-4
    assert cpu["x_hat"].min() >= 0 and cpu["x_hat"].max() <= 1
    # (4b) EVERY conv family at full size, teacher-forced on the GPU's own inputs (images 0 and last):
    #      g_a (both encoders), h_a, the four hyper-synthesis stacks, a base mean / scale / LRP stack, a progressive
    #      mean / scale / LRP stack
    def close(got, ref, tol, what):
        err = (got - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (what, err, ref.abs().max().item())
    xs = x.cpu()[sel]
    ys = y[sel]
    if bf16:
        with O.bf16_storage():
            assert _rms_rel(ys[:, :320], O.g_a(sd, "g_a.0.", xs)) <= 2e-2 and _rms_rel(ys[:, 320:], O.g_a(sd, "g_a.1.", xs)) <= 2e-2
    else:
        close(ys[:, :320], O.g_a(sd, "g_a.0.", xs), 1e-4, "g_a.0")
        close(ys[:, 320:], O.g_a(sd, "g_a.1.", xs), 1e-4, "g_a.1")
    close(plan.z.torch_nchw().cpu()[sel], O.h_a(sd, ys), 1e-4, "h_a")
    mh, sh = plan.means_h.torch_nchw().cpu()[sel], plan.scales_h.torch_nchw().cpu()[sel]
    for k in (0, 1):
        close(mh[:, 320 * k:320 * (k + 1)], O.h_s(sd, f"h_mean_s.{k}.", z_hat), 1e-4, f"h_mean_s.{k}")
        close(sh[:, 320 * k:320 * (k + 1)], O.h_s(sd, f"h_scale_s.{k}.", z_hat), 1e-4, f"h_scale_s.{k}")
    yb, mub, sdb = cpu["y_base"][sel], cpu["mu_base"][sel], cpu["std_base"][sel]
    close(mub[:, 96:128], O.cc_stack(sd, "cc_mean_transforms.3.", torch.cat([mh[:, :320], yb[:, :96]], 1)), 3e-4, "cc_mean.3")
    close(sdb[:, 224:256], O.cc_stack(sd, "cc_scale_transforms.7.", torch.cat([sh[:, :320], yb[:, :160]], 1)), 3e-4, "cc_scale.7")
    yq2 = torch.round(ys[:, 64:96] - mub[:, 64:96]) + mub[:, 64:96]
    lrp = O.cc_stack(sd, "lrp_transforms.2.", torch.cat([mh[:, :320], yb[:, :64], yq2], 1))
    close(yb[:, 64:96], yq2 + 0.5 * torch.tanh(lrp), 3e-4, "lrp.2")
    std_p = plan.std_p.torch_nchw().cpu()[sel]                # progressive sigma chain (before any REM)
    mu_p = plan.mu_p.torch_nchw().cpu()[sel]
    ssup = torch.cat([sh[:, 320:], yb[:, 192:224], std_p[:, 32:192]], 1)
    close(std_p[:, 192:224], O.cc_stack(sd, "cc_scale_transforms_prog.6.", ssup), 3e-4, "cc_scale_prog.6")
    mu_tot = mu_p + yb
    msup9 = torch.cat([mh[:, 320:], yb[:, 288:320], mu_tot[:, 128:288]], 1)
    close(mu_p[:, 288:320], O.cc_stack(sd, "cc_mean_transforms_prog.9.", msup9), 3e-4, "cc_mean_prog.9")
    r9 = ys[:, 608:640] - ys[:, 288:320]
    rq9 = torch.round(r9 - mu[sel][:, 288:320]) * mask[sel][:, 288:320] + mu[sel][:, 288:320]
    lrp = O.cc_stack(sd, "lrp_transforms_prog.9.", torch.cat([msup9, rq9], 1))
    close(cpu["y_hat"][sel][:, 288:320], rq9 + 0.5 * torch.tanh(lrp) + yb[:, 288:320], 3e-4, "lrp_prog.9")
    # (5) determinism: graph replay reproduces every bit
    with torch.no_grad():
        again = net.forward_single_quality(x, q)
    assert torch.equal(again["x_hat"], out["x_hat"]) and torch.equal(again["mask"], out["mask"])
