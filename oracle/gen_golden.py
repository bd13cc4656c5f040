#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE (build container only).

TEST INFRASTRUCTURE.  Imports the reference's own Python sources from
/root/reference/src (read-only) with the two third-party stand-ins of oracle/ref_stubs
(compressai, timm — absent offline), fills it with the deterministic synthetic weights
of vampic.synth and stores the reference's outputs as small data files.  Nothing of the
reference's code is copied; only inputs (re-generated from seeds) and expected outputs
are committed.

    python oracle/gen_golden.py            # writes tests/golden/*.npz / *.json
"""
import argparse
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
sys.path[:0] = [os.path.join(ROOT, "oracle", "ref_stubs"), REF, ROOT]
sys.dont_write_bytecode = True

import vampic.synth as synth  # noqa: E402  (pure python/torch, no GPU needed)

GOLD = os.environ.get("VAMPIC_GOLDEN_DIR", os.path.join(ROOT, "tests", "golden"))     # (another directory: regenerate without touching the committed files)
Q_LEVS = [0, 0.01, 0.05, 0.1, 0.25, 0.5, 0.6, 0.75, 1, 1.5, 2, 2.5, 3, 5, 7.7, 9.99, 10, 12]


# constructor flags away from the README configuration (section 9); shared with the tests through vampic.synth? no —
# the tests carry the same table (tests/config_variants.py) and check it against the fixture's key set
CONFIG_VARIANTS = {
    "single_encoder": dict(multiple_encoder=False),
    "single_decoder": dict(multiple_decoder=False),
    "single_hyperprior": dict(multiple_hyperprior=False),
    "all_single": dict(multiple_encoder=False, multiple_decoder=False, multiple_hyperprior=False),
    "sp0": dict(support_progressive_slices=0),
    "sp2": dict(support_progressive_slices=2),
    "sp8": dict(support_progressive_slices=8),
    "no_delta_no_mu_rep": dict(delta_encode=False, total_mu_rep=False),
    "not_all_scalable": dict(all_scalable=False),
    "rem_big": dict(model="rem", dimension="big"),
    "rem_no_mu_std": dict(model="rem", mu_std=False),
    "rem_not_all_scalable": dict(model="rem", all_scalable=False, support_progressive_slices=3),
}


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def fill(module, seed):
    sd = synth.synth_state_dict(module.state_dict(), seed)
    torch.nn.Module.load_state_dict(module, sd)
    return sd


TRAINED_LIKE_Q = (0, 0.5, 2.5, 10)


def section_trained_like(get_model, args):
    """11. (round 4) The north star's literal tolerances at a realistic rate: the reference's plain ``pic`` model filled with
    the ``trained-like`` profile of the synthetic generator (vampic.synth: the latent's magnitude and the predicted scales
    turned down — y - mu is O(1), sigma 0.3 ... 1.3, 0.9 ... 2.5 bpp instead of the default profile's 20 ... 31), two
    64x64 images and the 256x256 demo image at four qualities.  Stored: mask bit-packs, thresholds, PSNR / bpp scalars
    (float64 sums over the reference's likelihoods), y_hat and strided x_hat samples."""
    pargs = argparse.Namespace(**{**vars(args), "model": "pic"})
    pic = quiet(get_model, pargs, "cpu").eval()
    torch.nn.Module.load_state_dict(pic, synth.synth_state_dict(pic.state_dict(), 0, profile="trained-like"))
    rec, scal = {}, {}
    cases = (("a", synth.synth_image(1, 64, 64, seed=0)), ("b", synth.synth_image(1, 64, 64, seed=1)),
             ("demo", synth.synth_image(1, 256, 256, seed=0)))
    with torch.no_grad():
        for name, x in cases:
            npix = x.shape[2] * x.shape[3]
            for q in TRAINED_LIKE_Q:
                o = pic.forward_single_quality(x, quality=q, training=False)
                tag = f"{name}_q{q}"
                small = name != "demo"
                rec[tag + "_y_hat"] = (o["y_hat"] if small else o["y_hat"][:, ::4, ::2, ::2]).numpy()
                rec[tag + "_x_hat"] = (o["x_hat"][:, :, ::2, ::2] if small else o["x_hat"][:, :, ::8, ::8]).numpy()
                if q > 0:
                    m = torch.cat([pic.masking(s_, pr=q, mask_pol="point-based-std") for s_ in o["std"].chunk(10, 1)], 1)
                    rec[tag + "_mask"] = np.packbits(m.numpy().astype(np.uint8).reshape(-1))
                if 0 < q < 10:
                    rec[tag + "_thr"] = np.array([torch.quantile(s_.ravel(), 1.0 - q * 0.1).item()
                                                  for s_ in o["std"][0].chunk(10, 0)], dtype=np.float32)
                mse = torch.nn.functional.mse_loss(x, o["x_hat"]).item()
                by = torch.log(o["likelihoods"]["y"].double()).sum().item() / (-np.log(2) * npix)
                bz = torch.log(o["likelihoods"]["z"].double()).sum().item() / (-np.log(2) * npix)
                scal[tag] = {"psnr": -10 * np.log10(mse), "bpp": by + bz, "bpp_y": by, "bpp_z": bz,
                             "abs_y_hat_max": float(o["y_hat"].abs().max())}
    np.savez_compressed(os.path.join(GOLD, "trained_like.npz"), **rec)
    with open(os.path.join(GOLD, "trained_like.json"), "w") as f:
        json.dump(scal, f, indent=1)


def section_rem_no_mu_std(get_model, args):
    """12. (round 4) REM fine-tune step of the ``mu_std=False`` variant (the block sees and refines the scale only,
    rem_pic.py:194-195,214-220; layers/rem.py:86,100): the reference's own training-mode forward + RateLoss + backward, noise
    injected as in section 6."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_training_loss3", os.path.join(REF, "training", "loss.py"))
    loss_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(loss_mod)
    a_ = argparse.Namespace(**{**vars(args), "model": "rem", "mu_std": False})
    net = quiet(get_model, a_, "cpu").eval()
    fill(net, 0)
    x = synth.synth_image(1, 64, 128, seed=0)
    with torch.no_grad():
        ck = net.forward_single_quality(x, quality=0.75, training=False)["y_hat"]
    net.train()
    net.freeze_all()
    net.unfreeze_rems()
    ny = synth.uniform((1, 640, 4, 8), 101) - 0.5
    nz = synth.uniform((1, 192, 1, 2), 102) - 0.5
    queue = [nz.transpose(0, 1).reshape(192, 1, -1)] + list(ny.chunk(20, 1))
    real_uniform = torch.Tensor.uniform_

    def fake_uniform(self, a=0.0, b=1.0):
        src = queue.pop(0)
        assert tuple(src.shape) == tuple(self.shape) and (a, b) == (-0.5, 0.5), (src.shape, self.shape, a, b)
        with torch.no_grad():
            return self.copy_(src)

    torch.Tensor.uniform_ = fake_uniform
    try:
        o = net.forward_single_quality(x, quality=2.5, training=True, checkpoint_ref=ck.clone())
    finally:
        torch.Tensor.uniform_ = real_uniform
    assert not queue
    crit = loss_mod.RateLoss()(o, x)
    crit["loss"].backward()
    rec = {"ck": ck.numpy(), "lik_y": o["likelihoods"]["y"].detach().numpy(), "lik_z": o["likelihoods"]["z"].detach().numpy(),
           "loss": np.array([crit["loss"].item(), crit["bpp_loss"].item(), crit["bpp_hype"].item()], dtype=np.float64)}
    names, norms, samples = [], [], []
    for k, p_ in net.post_latent[0].named_parameters():
        gflat = p_.grad.detach().reshape(-1)
        names.append(k)
        norms.append(gflat.double().norm().item())
        samples.append(gflat[::53].numpy())
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms, dtype=np.float64)
    rec["grad_samples"] = np.concatenate(samples).astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "rem_train_step_no_mu_std.npz"), **rec)


def section_config_variants(get_model, args):
    """9. every constructor flag of models/__init__.py:11-55 away from the README values, one 64x64 image each
    (q = 0 and q = 2.5; REM variants with a checkpoint latent at their first check level).  Weights: the same
    name-keyed synthetic generator, so the shared modules carry the same values in every variant."""
    rec, scal = {}, {}
    x = synth.synth_image(1, 64, 64, seed=2)
    for name, over in CONFIG_VARIANTS.items():
        a_ = argparse.Namespace(**{**vars(args), "model": "pic", **over})
        ref_net = quiet(get_model, a_, "cpu").eval()
        fill(ref_net, 0)
        with torch.no_grad():
            ck = None
            if a_.model == "rem":
                ck = ref_net.forward_single_quality(x, quality=a_.check_levels[0], training=False)["y_hat"]
                rec[f"{name}_ck"] = ck.numpy()
            for q in (0, 2.5):
                kw = dict(checkpoint_ref=ck.clone()) if (ck is not None and q > 0) else {}
                o = ref_net.forward_single_quality(x, quality=q, training=False, **kw)
                tag = f"{name}_q{q}"
                rec[tag + "_x_hat"] = o["x_hat"][:, :, ::2, ::2].numpy()
                rec[tag + "_y_hat"] = o["y_hat"].numpy()
                mse = torch.nn.functional.mse_loss(x, o["x_hat"]).item()
                bits = sum(torch.log(v.double()).sum().item() for v in o["likelihoods"].values()) / (-np.log(2) * 64 * 64)
                scal[tag] = {"psnr": -10 * np.log10(mse), "bpp": bits}
    np.savez_compressed(os.path.join(GOLD, "config_variants.npz"), **rec)
    with open(os.path.join(GOLD, "config_variants.json"), "w") as f:
        json.dump(scal, f, indent=1)


TRAIN_VARIANTS = ("single_encoder", "single_decoder", "single_hyperprior", "all_single", "no_delta_no_mu_rep", "not_all_scalable")


def section_first_train_variants(get_model, args):
    """13. (round 4) first-stage training step (section 10's procedure: ``forward(x, [0, 10], training=True)``,
    ScalableRateDistortionLoss, backward of every parameter, injected noise) for the constructor variants with a single
    encoder / decoder / hyperprior (models/__init__.py:11-55; pic.py:285-288,306-311,372,462-466): losses, likelihoods,
    strided reconstructions and every 997th element of every gradient."""
    import importlib.util
    import warnings
    spec = importlib.util.spec_from_file_location("ref_training_loss4", os.path.join(REF, "training", "loss.py"))
    loss_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(loss_mod)
    xt = synth.synth_image(2, 64, 64, seed=5)
    ny = synth.uniform((2, 640, 4, 4), 201) - 0.5
    nz = synth.uniform((2, 192, 1, 1), 202) - 0.5
    rec = {}
    for name in TRAIN_VARIANTS:
        a_ = argparse.Namespace(**{**vars(args), "model": "pic", **CONFIG_VARIANTS[name]})
        net = quiet(get_model, a_, "cpu").train()
        fill(net, 0)
        for p_ in net.parameters():
            p_.requires_grad = True
        queue = [nz.transpose(0, 1).reshape(192, 1, -1)] + list(ny.chunk(20, 1))
        real = torch.Tensor.uniform_

        def fake(self, a=0.0, b=1.0):
            src = queue.pop(0)
            assert tuple(src.shape) == tuple(self.shape) and (a, b) == (-0.5, 0.5), (src.shape, self.shape, a, b)
            with torch.no_grad():
                return self.copy_(src)
        torch.Tensor.uniform_ = fake
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                o = net(xt, quality=[0, 10], training=True)
                crit = loss_mod.ScalableRateDistortionLoss(lmbda_list=[0.0055, 0.04], device="cpu")(o, xt)
        finally:
            torch.Tensor.uniform_ = real
        assert not queue
        crit["loss"].backward()
        rec[name + "_loss"] = np.array([crit[k].mean().item() for k in ("loss", "bpp_loss", "bpp_base", "bpp_scalable", "bpp_hype")], dtype=np.float64)
        rec[name + "_mse"] = crit["mse_loss"].detach().double().numpy()
        rec[name + "_x_hat"] = o["x_hat"].detach()[:, :, :, ::4, ::4].numpy()
        rec[name + "_lik_y"] = o["likelihoods"]["y"].detach().numpy()
        rec[name + "_lik_y_prog"] = o["likelihoods"]["y_prog"].detach().numpy()
        rec[name + "_lik_z"] = o["likelihoods"]["z"].detach().numpy()
        rec[name + "_y_hat_base"] = o["y_hat"][0].detach().numpy()
        names, norms, samples = [], [], []
        for k, p_ in net.named_parameters():
            assert p_.grad is not None, (name, k)
            gflat = p_.grad.detach().reshape(-1)
            names.append(k)
            norms.append(gflat.double().norm().item())
            samples.append(gflat[::997].numpy())
        rec[name + "_grad_names"] = np.array(names)
        rec[name + "_grad_norms"] = np.array(norms, dtype=np.float64)
        rec[name + "_grad_samples"] = np.concatenate(samples).astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "first_train_variants.npz"), **rec)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    from models import get_model                       # the reference
    import layers as RL
    from entropy_models import GaussianConditional, EntropyBottleneck

    args = argparse.Namespace(model="rem", N=192, M=640, multiple_decoder=True, multiple_encoder=True,
                              multiple_hyperprior=True, dim_chunk=32, division_dimension=[320, 640],
                              mask_policy="point-based-std", support_progressive_slices=5, delta_encode=True,
                              total_mu_rep=True, all_scalable=True, check_levels=[0.75], mu_std=True,
                              dimension="middle")
    if os.environ.get("VAMPIC_GOLDEN_ONLY") == "trained_like":      # regenerate this one section (minutes instead of the whole run)
        return section_trained_like(get_model, args)
    if os.environ.get("VAMPIC_GOLDEN_ONLY") == "rem_no_mu_std":
        return section_rem_no_mu_std(get_model, args)
    if os.environ.get("VAMPIC_GOLDEN_ONLY") == "first_train_variants":
        return section_first_train_variants(get_model, args)
    if os.environ.get("VAMPIC_GOLDEN_ONLY") == "config_variants":
        return section_config_variants(get_model, args)
    net = quiet(get_model, args, "cpu").eval()

    # 1. state_dict manifest
    man = {k: [list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()}
    with open(os.path.join(GOLD, "state_dict_manifest.json"), "w") as f:
        json.dump({"n_params": sum(p.numel() for p in net.parameters()), "entries": man}, f)

    # 2. variance mask (reference ChannelMask on operator-level sigma inputs)
    cm = RL.ChannelMask("point-based-std")
    rec = {}
    for name, (B, C, h, w), seed in (("s8192", (3, 32, 16, 16), 3), ("s49152", (2, 32, 32, 48), 4), ("s480", (2, 32, 5, 3), 5)):
        s = synth.synth_sigma(B, C * h * w, seed=seed).reshape(B, C, h, w)
        for q in Q_LEVS:
            m = cm(s, pr=q).numpy().astype(np.uint8)
            rec[f"{name}_q{q}"] = np.packbits(m.reshape(-1))
            if 0 < q < 10:
                rec[f"{name}_q{q}_thr"] = np.array([torch.quantile(s[b].ravel(), 1.0 - q * 0.1).item() for b in range(B)],
                                                   dtype=np.float32)
    blocks = [synth.synth_sigma(1, 32 * 64, seed=20 + i).reshape(1, 32, 8, 8) for i in range(10)]
    for q in (0, 0.5, 2.5, 10):
        rec[f"prog_q{q}"] = np.packbits(cm.ProgMask(blocks, q).numpy().astype(np.uint8).reshape(-1))
    np.savez_compressed(os.path.join(GOLD, "variance_mask.npz"), **rec)

    # 3. Gaussian conditional / entropy bottleneck (operator level)
    g = lambda shape, seed, sc=1.0: synth.normal(shape, seed, sc)
    gc = GaussianConditional(None)
    y, mu = g((2, 32, 16, 16), 30, 6.0), g((2, 32, 16, 16), 31, 4.0)
    sg = synth.synth_sigma(2, 32 * 256, seed=32).reshape(2, 32, 16, 16)
    out, lik = gc(y, sg, mu, training=False)
    out2, lik2 = gc(y, sg, None, training=False)
    gc.scale_table = torch.exp(torch.linspace(np.log(0.11), np.log(256), 64))
    idx = gc.build_indexes(sg)
    eb = EntropyBottleneck(192)
    fill(eb, 40)
    z = g((2, 192, 4, 6), 41, 5.0)
    zh, zl = eb(z, training=False)
    aux = eb.loss()                                    # entropy_models.py:398-401 (aux_loss of models/base.py:22-29)
    aux.backward()
    np.savez_compressed(os.path.join(GOLD, "entropy_ops.npz"), gc_out=out.numpy(), gc_lik=lik.numpy(),
                        gc_out_nomean=out2.numpy(), gc_lik_nomean=lik2.numpy(), gc_idx=idx.numpy().astype(np.int8),
                        eb_zhat=zh.detach().numpy(), eb_lik=zl.detach().numpy(),
                        eb_aux_loss=np.array([aux.item()], dtype=np.float64), eb_aux_dq=eb.quantiles.grad.numpy())

    # 4. layer-level vectors
    rec = {}
    with torch.no_grad():
        for inv in (False, True):
            m = RL.GDN(192, inverse=inv)
            fill(m, 7)
            rec[f"gdn_inv{int(inv)}"] = m(g((2, 192, 8, 8), 8, 2.0)).numpy()
        for dim, ws, hw in ((192, 8, (16, 24)), (320, 4, (8, 8))):
            m = RL.Win_noShift_Attention(dim=dim, num_heads=8, window_size=ws, shift_size=ws // 2)
            fill(m, 9)
            rec[f"attn_{dim}"] = m(g((2, dim) + hw, 10)).numpy()
        m = RL.LatentRateReduction(32, True, "middle")
        fill(m, 11)
        att = (synth.uniform((2, 32, 8, 8), 15) > 0.5).float()
        rec["rem"] = m(g((2, 32, 8, 8), 12, 3.0), g((2, 64, 8, 8), 13), g((2, 64, 8, 8), 14), torch.cat([att, att], 1)).numpy()
    np.savez_compressed(os.path.join(GOLD, "layer_ops.npz"), **rec)

    # 5. end-to-end forward_single_quality of the reference model (64x64 images: tiny tensors)
    sd = fill(net, 0)
    rec = {}
    scal = {}
    with torch.no_grad():
        for seed in (0, 1):
            x = synth.synth_image(1, 64, 64, seed=seed)
            for q in (0, 0.5, 2.5, 10):
                o = net.forward_single_quality(x, quality=q, training=False)
                tag = f"s{seed}_q{q}"
                rec[tag + "_x_hat"] = o["x_hat"].numpy()
                rec[tag + "_y_hat"] = o["y_hat"].numpy()
                rec[tag + "_lik_y"] = o["likelihoods"]["y"].numpy()
                rec[tag + "_lik_z"] = o["likelihoods"]["z"].numpy()
                mse = torch.nn.functional.mse_loss(x, o["x_hat"]).item()
                bits = sum(torch.log(v.double()).sum().item() for v in o["likelihoods"].values()) / (-np.log(2) * 64 * 64)
                scal[tag] = {"psnr": -10 * np.log10(mse), "bpp": bits}
        # REM: checkpoint latent at q=0.75, refined pass at q=2.5
        x = synth.synth_image(1, 64, 128, seed=0)
        ck = net.forward_single_quality(x, quality=0.75, training=False)["y_hat"]
        o = net.forward_single_quality(x, quality=2.5, training=False, checkpoint_ref=ck.clone())
        rec["rem_ck"] = ck.numpy()
        rec["rem_x_hat"] = o["x_hat"].numpy()
        rec["rem_y_hat"] = o["y_hat"].numpy()
        rec["rem_lik_y"] = o["likelihoods"]["y"].numpy()
    np.savez_compressed(os.path.join(GOLD, "forward_single_quality.npz"), **rec)
    with open(os.path.join(GOLD, "forward_single_quality.json"), "w") as f:
        json.dump(scal, f, indent=1)
    # 6. REM fine-tune step: the reference's own training-mode forward + RateLoss + backward
    #    (training/step.py:62-88 with --training_type rems, train.py:223-226).  Its additive noise comes from
    #    Tensor.uniform_; for the duration of the call that method is replaced by a deterministic source
    #    (vampic.synth.uniform, host-independent) so the same draws can be injected on the other side.
    import importlib.util          # training/__init__.py pulls torchvision (absent): load loss.py on its own
    spec = importlib.util.spec_from_file_location("ref_training_loss", os.path.join(REF, "training", "loss.py"))
    loss_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(loss_mod)
    RateLoss = loss_mod.RateLoss
    net.train()
    net.freeze_all()
    net.unfreeze_rems()
    x = synth.synth_image(1, 64, 128, seed=0)
    ny = synth.uniform((1, 640, 4, 8), 101) - 0.5
    nz = synth.uniform((1, 192, 1, 2), 102) - 0.5
    queue = [nz.transpose(0, 1).reshape(192, 1, -1)] + list(ny.chunk(20, 1))
    real_uniform = torch.Tensor.uniform_

    def fake_uniform(self, a=0.0, b=1.0):
        src = queue.pop(0)
        assert tuple(src.shape) == tuple(self.shape) and (a, b) == (-0.5, 0.5), (src.shape, self.shape, a, b)
        with torch.no_grad():
            return self.copy_(src)

    torch.Tensor.uniform_ = fake_uniform
    try:
        o = net.forward_single_quality(x, quality=2.5, training=True, checkpoint_ref=ck.clone())
    finally:
        torch.Tensor.uniform_ = real_uniform
    assert not queue
    crit = RateLoss()(o, x)
    crit["loss"].backward()
    rec = {"lik_y": o["likelihoods"]["y"].detach().numpy(), "lik_z": o["likelihoods"]["z"].detach().numpy(),
           "loss": np.array([crit["loss"].item(), crit["bpp_loss"].item(), crit["bpp_hype"].item()], dtype=np.float64)}
    names, norms, samples = [], [], []
    for k, p_ in net.post_latent[0].named_parameters():
        gflat = p_.grad.detach().reshape(-1)
        names.append(k)
        norms.append(gflat.double().norm().item())
        samples.append(gflat[::53].numpy())
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms, dtype=np.float64)
    rec["grad_samples"] = np.concatenate(samples).astype(np.float32)      # every 53rd element of each gradient
    others = [k for k, p_ in net.named_parameters() if p_.grad is not None and not k.startswith("post_latent.")]
    assert not others, others
    np.savez_compressed(os.path.join(GOLD, "rem_train_step.npz"), **rec)
    net.eval()

    # 7. host utilities (utility/functions.py, imported with the torchvision / pytorch_msssim / wandb stand-ins):
    #    key remapping of initialize_model_from_pretrained / replace_keys, compute_padding, compute_psnr
    spec = importlib.util.spec_from_file_location("ref_utility_functions", os.path.join(REF, "utility", "functions.py"))
    UF = importlib.util.module_from_spec(spec)         # utility/__init__.py pulls seaborn (absent): load the file alone
    spec.loader.exec_module(UF)
    single = ["g_a.0.weight", "g_a.1.beta", "g_a.1.gamma", "g_s.0.weight", "g_s.8.bias", "h_a.0.weight", "h_a.8.bias",
              "h_mean_s.0.weight", "h_mean_s.8.bias", "h_scale_s.0.weight", "h_scale_s.6.0.bias",
              "cc_mean_transforms.0.0.weight", "lrp_transforms.3.8.bias", "gaussian_conditional.scale_table",
              "entropy_bottleneck._matrix0", "entropy_bottleneck.quantiles", "something_else.weight"]
    ck = {k: torch.full((1,), float(i)) for i, k in enumerate(single)}
    enh = {k: torch.full((1,), 100.0 + i) for i, k in enumerate(["g_s.0.weight", "g_s.8.bias", "g_a.0.weight"])}
    util = {"init": {}, "replace": {}, "padding": {}, "psnr": {}}
    for md in (False, True):
        for me in (False, True):
            for mh in (False, True):
                for with_enh in (False, True):
                    a_ = argparse.Namespace(multiple_decoder=md, multiple_encoder=me, multiple_hyperprior=mh)
                    r = quiet(UF.initialize_model_from_pretrained, ck, a_, enh if with_enh else None)
                    util["init"][f"{int(md)}{int(me)}{int(mh)}{int(with_enh)}"] = [[k, float(v)] for k, v in r.items()]
    old = {k: torch.full((1,), float(i)) for i, k in enumerate(["g_a.0.weight", "g_a_enh.0.weight", "g_a.1.beta", "g_s.0.weight"])}
    new = {k: torch.full((1,), float(i)) for i, k in enumerate(["g_a.0.0.weight", "g_a.0.1.beta", "g_a.1.0.weight"])}
    for me in (False, True):
        util["replace"][f"old{int(me)}"] = [[k, float(v)] for k, v in UF.replace_keys(old, me).items()]
        util["replace"][f"new{int(me)}"] = [[k, float(v)] for k, v in UF.replace_keys(new, me).items()]
    for (hh, ww) in ((512, 768), (500, 333), (1, 1), (64, 65), (1200, 1999)):
        util["padding"][f"{hh}x{ww}"] = [list(t) for t in UF.compute_padding(hh, ww, min_div=64)]
    for seed in (1, 2):
        a_, b_ = synth.uniform((2, 3, 16, 24), seed), synth.uniform((2, 3, 16, 24), seed + 10)
        util["psnr"][str(seed)] = UF.compute_psnr(a_, b_)
    util["savepath"] = list(UF.create_savepath("/x/y"))
    with open(os.path.join(GOLD, "host_utils.json"), "w") as f:
        json.dump(util, f)

    # 8. BASELINE configs[0]: the demo's workload — ONE 256x256 image at the 15 quality levels of the demo's parser
    #    (test/parser.py:20).  Stored: mask bit-packs, per-slice thresholds, PSNR / bpp scalars, strided samples.
    DEMO_Q = [0, 0.01, 0.05, 0.1, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9, 1, 2, 3, 4, 4.5, 10]      # base + parser.py:20
    #    Run on the reference's plain "pic" model (demo.py --model pic) with the same weights: its output dict carries
    #    the sigma the mask was computed from ("std", pic.py:661-666); the REM class returns std_base + sigma under
    #    "std_prog" instead (rem_pic.py:343,417-422).
    pargs = argparse.Namespace(**{**vars(args), "model": "pic"})
    pic = quiet(get_model, pargs, "cpu").eval()
    own = pic.state_dict()
    torch.nn.Module.load_state_dict(pic, {k: v for k, v in sd.items() if k in own})
    assert all(k in sd for k in own)
    rec, scal = {}, {}
    x = synth.synth_image(1, 256, 256, seed=0)
    with torch.no_grad():
        for q in DEMO_Q:
            o = pic.forward_single_quality(x, quality=q, training=False)
            o_std = o["std"]
            tag = f"q{q}"
            rec[tag + "_y_hat"] = o["y_hat"][:, ::4, ::2, ::2].numpy()
            rec[tag + "_x_hat"] = o["x_hat"][:, :, ::8, ::8].numpy()
            if q > 0:      # the reference does not return its masks: re-apply ITS ChannelMask to the sigma it returns (pic.py:621-622)
                m = torch.cat([pic.masking(s_, pr=q, mask_pol="point-based-std") for s_ in o_std.chunk(10, 1)], 1)
                rec[tag + "_mask"] = np.packbits(m.numpy().astype(np.uint8).reshape(-1))
            if 0 < q < 10:
                rec[tag + "_thr"] = np.array([torch.quantile(s_.ravel(), 1.0 - q * 0.1).item()
                                              for s_ in o_std[0].chunk(10, 0)], dtype=np.float32)
            mse = torch.nn.functional.mse_loss(x, o["x_hat"]).item()
            bits = sum(torch.log(v.double()).sum().item() for v in o["likelihoods"].values()) / (-np.log(2) * 65536)
            scal[tag] = {"psnr": -10 * np.log10(mse), "bpp": bits}
    np.savez_compressed(os.path.join(GOLD, "demo_256.npz"), **rec)
    with open(os.path.join(GOLD, "demo_256.json"), "w") as f:
        json.dump(scal, f, indent=1)

    # 8b. decoder refinement step (`--training_type refine_gs`, train.py:150-157,216-218): the reference's own
    #     training-mode forward, DistortionLoss (training/loss.py:126-187) and backward with only g_s[1] trainable.
    spec = importlib.util.spec_from_file_location("ref_training_loss2", os.path.join(REF, "training", "loss.py"))
    loss_mod2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(loss_mod2)
    pic.train()
    pic.freeze_all()
    pic.unfreeze_decoder()
    xr = synth.synth_image(1, 64, 64, seed=3)
    torch.manual_seed(1)                                    # the likelihood noise (unused by the distortion loss)
    o = pic.forward_single_quality(xr, quality=2.5, training=True)
    crit = loss_mod2.DistortionLoss(device="cpu")(o, xr)
    crit["loss"].backward()
    rec = {"loss": np.array([crit["loss"].item(), crit["mse_loss"].item()], dtype=np.float64),
           "x_hat": o["x_hat"].detach()[:, :, ::2, ::2].numpy()}
    names, norms, samples = [], [], []
    for k, p_ in pic.g_s[1].named_parameters():
        gflat = p_.grad.detach().reshape(-1)
        names.append(k)
        norms.append(gflat.double().norm().item())
        samples.append(gflat[::97].numpy())
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms, dtype=np.float64)
    rec["grad_samples"] = np.concatenate(samples).astype(np.float32)      # every 97th element of each gradient
    others = [k for k, p_ in pic.named_parameters() if p_.grad is not None and not k.startswith("g_s.1.")]
    assert not others, others
    np.savez_compressed(os.path.join(GOLD, "refine_gs_step.npz"), **rec)

    # 8c. the same step with `--lrp` (train.py:216-218 -> unfreeze_decoder(lrp=True), pic.py:171-184): the ten progressive
    #     latent-residual-prediction stacks train with g_s[1].
    pic.zero_grad(set_to_none=True)
    pic.freeze_all()
    pic.unfreeze_decoder(lrp=True)
    torch.manual_seed(1)
    o = pic.forward_single_quality(xr, quality=2.5, training=True)
    crit = loss_mod2.DistortionLoss(device="cpu")(o, xr)
    crit["loss"].backward()
    rec = {"loss": np.array([crit["loss"].item(), crit["mse_loss"].item()], dtype=np.float64),
           "x_hat": o["x_hat"].detach()[:, :, ::4, ::4].numpy()}
    names, norms, samples = [], [], []
    for k, p_ in pic.named_parameters():
        if p_.grad is None:
            continue
        assert k.startswith("g_s.1.") or k.startswith("lrp_transforms_prog."), k
        gflat = p_.grad.detach().reshape(-1)
        names.append(k)
        norms.append(gflat.double().norm().item())
        samples.append(gflat[::389].numpy())
    assert sum(n.startswith("lrp_transforms_prog.") for n in names) == 100, len(names)
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms, dtype=np.float64)
    rec["grad_samples"] = np.concatenate(samples).astype(np.float32)      # every 389th element of each gradient
    np.savez_compressed(os.path.join(GOLD, "refine_gs_lrp_step.npz"), **rec)
    pic.zero_grad(set_to_none=True)
    pic.eval()

    # 10. first-stage training (BASELINE configs[3], `--training_type first_train`, train.py:146-149): the reference's own
    #     ``forward(x, quality=[0, 10])`` in training mode, ScalableRateDistortionLoss (training/loss.py:6-66, the parser's
    #     default --lmbda_list) and backward with EVERY parameter trainable; then one single-quality training pass
    #     (``forward_single_quality(x, 2.5, training=True)``: mask, clamp and straight-through paths with a non-trivial mask)
    #     under the same criterion with an explicit lmbda.  Noise injected as in section 6.
    pic.train()
    for p_ in pic.parameters():
        p_.requires_grad = True
    pic.zero_grad(set_to_none=True)
    xt = synth.synth_image(2, 64, 64, seed=5)
    ny = synth.uniform((2, 640, 4, 4), 201) - 0.5
    nz = synth.uniform((2, 192, 1, 1), 202) - 0.5

    def with_noise(fn):
        queue = [nz.transpose(0, 1).reshape(192, 1, -1)] + list(ny.chunk(20, 1))
        real = torch.Tensor.uniform_

        def fake(self, a=0.0, b=1.0):
            src = queue.pop(0)
            assert tuple(src.shape) == tuple(self.shape) and (a, b) == (-0.5, 0.5), (src.shape, self.shape, a, b)
            with torch.no_grad():
                return self.copy_(src)
        torch.Tensor.uniform_ = fake
        try:
            r = fn()
        finally:
            torch.Tensor.uniform_ = real
        assert not queue
        return r

    def grad_record(net_, stride):
        names, norms, samples = [], [], []
        for k, p_ in net_.named_parameters():
            if p_.grad is None:
                continue
            gflat = p_.grad.detach().reshape(-1)
            names.append(k)
            norms.append(gflat.double().norm().item())
            samples.append(gflat[::stride].numpy())
        return {"grad_names": np.array(names), "grad_norms": np.array(norms, dtype=np.float64),
                "grad_samples": np.concatenate(samples).astype(np.float32)}

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                # the criterion's broadcasting mse_loss (training/loss.py:41)
        o = with_noise(lambda: pic(xt, quality=[0, 10], training=True))
        crit = loss_mod2.ScalableRateDistortionLoss(lmbda_list=[0.0055, 0.04], device="cpu")(o, xt)
        crit["loss"].backward()
        rec = {"loss": np.array([crit[k].mean().item() for k in ("loss", "bpp_loss", "bpp_base", "bpp_scalable", "bpp_hype")], dtype=np.float64),
               "mse": crit["mse_loss"].detach().double().numpy(),
               "x_hat": o["x_hat"].detach()[:, :, :, ::4, ::4].numpy(), "lik_y": o["likelihoods"]["y"].detach().numpy(),
               "lik_y_prog": o["likelihoods"]["y_prog"].detach().numpy(), "lik_z": o["likelihoods"]["z"].detach().numpy(),
               "y_hat_base": o["y_hat"][0].detach().numpy(), "y_hat_prog": o["y_hat"][1].detach().numpy()}
        rec.update(grad_record(pic, 997))
        unused = [k for k, p_ in pic.named_parameters() if p_.grad is None]
        assert not unused, unused
        np.savez_compressed(os.path.join(GOLD, "first_train_step.npz"), **rec)
        pic.zero_grad(set_to_none=True)
        o = with_noise(lambda: pic.forward_single_quality(xt, quality=2.5, training=True))
        crit = loss_mod2.ScalableRateDistortionLoss(lmbda_list=[0.0055, 0.04], device="cpu")(o, xt, lmbda=0.01)
        crit["loss"].backward()
        rec = {"loss": np.array([crit[k].mean().item() for k in ("loss", "bpp_loss", "bpp_base", "bpp_scalable", "bpp_hype")], dtype=np.float64),
               "mse": crit["mse_loss"].detach().double().numpy(),
               "x_hat": o["x_hat"].detach()[:, :, ::4, ::4].numpy(), "lik_y": o["likelihoods"]["y"].detach().numpy(),
               "lik_z": o["likelihoods"]["z"].detach().numpy(), "y_hat": o["y_hat"].detach().numpy()}
        rec.update(grad_record(pic, 997))
        np.savez_compressed(os.path.join(GOLD, "train_single_step.npz"), **rec)
    pic.zero_grad(set_to_none=True)
    pic.eval()

    section_config_variants(get_model, args)           # 9. constructor-flag variants
    section_trained_like(get_model, args)              # 11. trained-like weight profile (round 4)
    section_rem_no_mu_std(get_model, args)             # 12. REM fine-tune step, mu_std = False (round 4)
    section_first_train_variants(get_model, args)      # 13. first-stage training step, single encoder / decoder / hyperprior (round 4)

    print("golden vectors written to", GOLD)
    for fn in sorted(os.listdir(GOLD)):
        print(f"  {fn}: {os.path.getsize(os.path.join(GOLD, fn)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
