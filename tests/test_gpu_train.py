"""REM fine-tune path (BASELINE configs[4], reference train.py:223-226 + training/step.py:56-95) on the GPU:
backward kernels against torch autograd / the oracle, the full step against the reference's own gradients."""
import copy
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import vampic                              # noqa: E402
import vampic.synth as synth               # noqa: E402
import vampic_oracle as O                  # noqa: E402
from vampic import _lib as L               # noqa: E402
from vampic import ops                     # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("k,segs,n,hw", [(3, (32, 32), 32, (16, 16)), (1, (32, 32, 32), 64, (4, 8)), (3, (64,), 64, (5, 7)),
                                         (3, (32,), 32, (16, 24))])
def test_conv_backward_matches_autograd(k, segs, n, hw):
    """Data gradient (conv kernel on VAM_PACK_CONV_DGRAD weights) and weight / bias gradients (vam_conv_wgrad,
    vam_colsum) of a stride-1 conv over a virtual channel concat.  Tolerance 2e-5 of the max (fp32, different
    summation order than ATen)."""
    B, (H, W) = 2, hw
    cin = sum(segs)
    w = synth.normal((n, cin, k, k), 1, 0.1)
    b = synth.normal((n,), 2, 0.1)
    xs = [synth.normal((B, c, H, W), 3 + i) for i, c in enumerate(segs)]
    dy = synth.normal((B, n, H, W), 9)
    x = torch.cat(xs, 1).requires_grad_(True)
    wt, bt = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.conv2d(x, wt, bt, padding=k // 2).backward(dy)
    xv = [ops.from_nchw(t.cuda()) for t in xs]
    dyv = ops.from_nchw(dy.cuda())
    dw = torch.full((n, cin, k, k), float("nan"), device="cuda")
    db = torch.full((n,), float("nan"), device="cuda")
    ops.conv_wgrad(xv, dyv, dw, db)
    assert _rel(dw, wt.grad) <= 2e-5 and _rel(db, bt.grad) <= 2e-5
    pk = ops.pack_conv_dgrad(w.cuda())
    dx = ops.new_view(B, H, W, cin)
    ops.conv_group([ops.conv_problem(pk, [dyv], dx)])
    assert _rel(dx.torch_nchw(), x.grad) <= 2e-5


def test_leaky_bwd_and_mul():
    a = synth.normal((2, 32, 4, 8), 1)
    g = synth.normal((2, 32, 4, 8), 2)
    act = F.leaky_relu(a, 0.01)
    av, gv = ops.from_nchw(act.cuda()), ops.from_nchw(g.cuda())
    o = ops.new_view(2, 4, 8, 32)
    ops.leaky_bwd(av, gv, o)
    assert torch.equal(o.torch_nchw().cpu(), g * torch.where(a > 0, 1.0, 0.01).float())
    ops.mul(av, gv, o)
    assert torch.equal(o.torch_nchw().cpu(), act * g)


@pytest.mark.parametrize("masked", [False, True])
def test_gauss_train_matches_oracle_autograd(masked):
    """Noisy likelihood forward (<= 1e-6 absolute: erfc implementations and the division differ by an ulp, sigma down to 0.11) and its gradient w.r.t. (mu, sigma)
    incl. both LowerBound rules (sigma below 0.11, likelihood below 1e-9) against the oracle's autograd."""
    shp = (2, 64, 8, 8)
    y, y2 = synth.normal(shp, 1, 4.0), synth.normal(shp, 2, 2.0)
    mu = synth.normal(shp, 3, 2.0).requires_grad_(True)
    sg = (synth.synth_sigma(2, 64 * 64, seed=4).reshape(shp) * 0.5).requires_grad_(True)   # some below the 0.11 bound
    with torch.no_grad():
        y[0, 0, 0, :4] += 80.0                                                              # likelihood below 1e-9
    nz = synth.uniform(shp, 5) - 0.5
    m = (synth.uniform(shp, 6) > 0.4).float()
    g = synth.normal(shp, 7)                                   # both signs: exercises the (grad < 0) pass-through
    if masked:
        lik = O.gaussian_likelihood_noise(((y - y2) - mu) * m, sg * m, None, nz)
    else:
        lik = O.gaussian_likelihood_noise(y, sg, mu, nz)
    lik.backward(g)
    V = lambda t: ops.from_nchw(t.detach().cuda())
    lk, dmu, dsg = ops.new_view(2, 8, 8, 64), ops.new_view(2, 8, 8, 64), ops.new_view(2, 8, 8, 64)
    kw = dict(y2=V(y2), mask=V(m)) if masked else {}
    ops.gauss_train(V(y), V(mu), V(sg), V(nz), lik=lk, **kw)
    ops.gauss_train(V(y), V(mu), V(sg), V(nz), grad_lik=V(g), dmu=dmu, dsigma=dsg, **kw)
    assert float((lk.torch_nchw().cpu() - lik.detach()).abs().max()) <= 1e-6
    for got, ref in ((dmu, mu.grad), (dsg, sg.grad)):
        d = (got.torch_nchw().cpu() - ref).abs()
        assert float(d.max()) <= 2e-5 * float(ref.abs().max()) + 1e-7, float(d.max())
    assert float((sg.grad == 0).float().mean()) > 0.01       # the bounds did clip some gradients


def test_module_level_training_forwards_are_differentiable():
    """``gaussian_conditional(y, sigma, mu, training=True)`` and ``entropy_bottleneck(z, training=True)`` called on the
    modules themselves (as the reference's harness may: entropy_models.py:449-492,637-652): outputs = input + U(-1/2, 1/2),
    likelihoods and their gradients w.r.t. every input / parameter against autograd over the oracle with the same noise."""
    from vampic import entropy_models as EM
    shp = (2, 64, 8, 8)
    y = synth.normal(shp, 1, 4.0).requires_grad_(True)
    mu = synth.normal(shp, 3, 2.0).requires_grad_(True)
    sg = (synth.synth_sigma(2, 64 * 64, seed=4).reshape(shp) * 0.5).requires_grad_(True)
    nz = synth.uniform(shp, 5) - 0.5
    g = synth.normal(shp, 7)
    O.gaussian_likelihood_noise(y, sg, mu, nz).backward(g)
    yg, mg, sgg = (t.detach().clone().cuda().requires_grad_(True) for t in (y, mu, sg))
    out, lik = EM._GaussTrainFn.apply(yg, sgg, mg, nz.cuda())
    lik.backward(g.cuda())
    assert torch.equal(out.cpu(), y.detach() + nz)
    for got, ref in ((yg.grad, y.grad), (mg.grad, mu.grad), (sgg.grad, sg.grad)):
        assert float((got.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-7
    gc = EM.GaussianConditional(None).cuda().train()
    out, lik = gc(yg, sgg, mg)                                   # draws its own noise (torch's generator, as the reference)
    d = (out - yg).detach()
    assert float(d.abs().max()) <= 0.5 and float(d.std()) > 0.2 and bool((lik > 0).all()) and lik.requires_grad
    # entropy bottleneck
    eb = EM.EntropyBottleneck(192)
    sd = synth.synth_state_dict(eb.state_dict(), 40)
    eb.load_state_dict(sd)
    eb = eb.cuda().train()
    z = synth.normal((2, 192, 4, 6), 41, 5.0)
    nzz = synth.uniform((2, 192, 4, 6), 42) - 0.5
    gz = synth.normal((2, 192, 4, 6), 43)
    leaves = {"entropy_bottleneck." + k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and k != "target"}
    zr = z.clone().requires_grad_(True)
    O.eb_likelihood_noise_bounded(leaves, zr, nzz).backward(gz)
    zg = z.clone().cuda().requires_grad_(True)
    out, lik = EM._EbTrainFn.apply(zg, nzz.cuda(), *[getattr(eb, n) for n in EM.EB_PARAM_NAMES])
    lik.backward(gz.cuda())
    assert float((zg.grad.cpu() - zr.grad).abs().max()) <= 5e-5 * float(zr.grad.abs().max()) + 1e-7
    for n in EM.EB_PARAM_NAMES:
        want = leaves["entropy_bottleneck." + n].grad
        got = getattr(eb, n).grad
        if n == "quantiles":
            assert float(got.abs().max()) == 0.0 and (want is None or float(want.abs().max()) == 0.0)
        else:
            assert float((got.cpu() - want).abs().max()) <= 5e-5 * float(want.abs().max()) + 1e-7, n
    out, lik = eb(zg)                                            # module call in training mode
    assert float((out - zg).abs().max()) <= 0.5 and lik.requires_grad and tuple(lik.shape) == tuple(zg.shape)


def test_rem_blocks_backward_teacher_forced():
    """Taped REM forward + backward lowering (engine.lower_rem_blocks_train / lower_rem_backward) on IDENTICAL
    inputs against autograd over the oracle's rem_block: outputs 1e-5, every parameter gradient 2e-5 of its max."""
    from vampic import engine as E, layers as Ly
    K, N, B, H, W = 2, 32, 2, 8, 8
    mods = [Ly.LatentRateReduction(N, True, "middle") for _ in range(K)]
    sds = []
    for k, mod in enumerate(mods):
        sd = synth.synth_state_dict(mod.state_dict(), 11 + k)
        mod.load_state_dict(sd)
        mod.cuda()
        sds.append(sd)
    plan, bw = E.Plan("cuda"), E.Plan("cuda")
    V = lambda t: ops.from_nchw(t.cuda())
    yck = [synth.normal((B, N, H, W), 20 + k, 3.0) for k in range(K)]
    epb = [synth.normal((B, 2 * N, H, W), 30 + k) for k in range(K)]
    epp = [synth.normal((B, 2 * N, H, W), 40 + k) for k in range(K)]
    att = [(synth.uniform((B, N, H, W), 50 + k) > 0.5).float() for k in range(K)]
    dres = [synth.normal((B, 2 * N, H, W), 60 + k) for k in range(K)]
    vb, vp, va, vd = [V(t) for t in epb], [V(t) for t in epp], [V(t) for t in att], [V(t) for t in dres]
    outs = [ops.new_view(B, H, W, 2 * N) for _ in range(K)]
    packs = E.TrainPacks(*E.rem_trained_convs(mods))
    packs.record_refresh(plan)
    tape = E.lower_rem_blocks_train(plan, mods, [V(t) for t in yck], [[v.window(0, N), v.window(N, N)] for v in vb],
                                    [[v.window(0, N), v.window(N, N)] for v in vp], va,
                                    [[o.window(0, N), o.window(N, N)] for o in outs], packs)
    grads = {id(p): torch.full_like(p, float("nan")) for mod in mods for p in mod.parameters()}
    E.lower_rem_backward(bw, tape, mods, [v.window(0, N) for v in vd], [v.window(N, N) for v in vd], va, packs, grads)
    plan.run()
    bw.run()
    torch.cuda.synchronize()
    for k in range(K):
        leaves = {"r." + n: t.clone().requires_grad_(True) for n, t in sds[k].items()}
        res = O.rem_block(leaves, "r.", yck[k], epb[k], epp[k], torch.cat([att[k], att[k]], 1))
        res.backward(dres[k])
        assert _rel(outs[k].torch_nchw(), res) <= 1e-5
        for n, p in mods[k].named_parameters():
            assert _rel(grads[id(p)], leaves["r." + n].grad) <= 2e-5, (k, n)


def _rate_loss(out, x):
    """training/loss.py:196-229 (RateLoss): bpp of y + z."""
    den = -math.log(2) * x.shape[0] * x.shape[2] * x.shape[3]
    return torch.log(out["likelihoods"]["y"]).sum() / den + torch.log(out["likelihoods"]["z"]).sum() / den


@pytest.fixture(scope="module")
def train_model(gpu_model):
    net, sd = gpu_model
    m = copy.deepcopy(net)
    m.train()
    m.freeze_all()
    m.unfreeze_rems()
    return m, sd


@pytest.mark.parametrize("use_graph", [False, True])
def test_rem_train_step_matches_reference_gradients(train_model, use_graph):
    """One fine-tune step (forward in training mode with the fixture's noise, RateLoss, backward) against the
    gradients the REFERENCE computed on a CPU (tests/golden/rem_train_step.npz).  The frozen front end feeds
    (mu, sigma) that differ from the CPU's by fp32 summation order (~1e-5 of their range), and the noisy
    likelihood is a smooth but steep function of them (sigma >= 0.11), so the end-to-end bounds are: likelihoods
    5e-4 absolute, loss 1e-5 relative.  Gradients additionally pass through hard gates (LowerBound's pass-through
    rule at sigma = 0.11, LeakyReLU's sign) that an fp32-order difference can flip for single elements, so they are
    bounded in aggregate: all sampled gradient entries together within 1e-3 of their joint norm, each tensor within
    3e-2 of its own norm.  The backward machinery itself is held to 2e-5 by
    test_rem_blocks_backward_teacher_forced and test_conv_backward_matches_autograd."""
    m, _ = train_model
    m.use_graph = use_graph
    gold = np.load(os.path.join(GOLD, "rem_train_step.npz"))
    ck = torch.from_numpy(np.load(os.path.join(GOLD, "forward_single_quality.npz"))["rem_ck"]).cuda()
    x = synth.synth_image(1, 64, 128, seed=0).cuda()
    noise = {"y": synth.uniform((1, 640, 4, 8), 101) - 0.5, "z": synth.uniform((1, 192, 1, 2), 102) - 0.5}
    for rep in range(2):                                      # second pass replays the captured graphs
        m.zero_grad(set_to_none=True)
        out = m.forward_single_quality(x, quality=2.5, training=True, checkpoint_ref=ck, noise=noise)
        loss = _rate_loss(out, x)
        loss.backward()
        e_y = _rel(out["likelihoods"]["y"], torch.from_numpy(gold["lik_y"]))
        e_z = _rel(out["likelihoods"]["z"], torch.from_numpy(gold["lik_z"]))
        e_l = abs(float(loss.detach()) - gold["loss"][0]) / gold["loss"][0]
        print(f"lik_y err {e_y:.2e}  lik_z err {e_z:.2e}  loss rel err {e_l:.2e}")
        assert e_y <= 5e-4 and e_z <= 1e-5 and e_l <= 1e-5
        samples, off, worst, num, den = gold["grad_samples"], 0, 0.0, 0.0, 0.0
        params = dict(m.post_latent[0].named_parameters())
        for name, norm in zip(gold["grad_names"], gold["grad_norms"]):
            g = params[str(name)].grad.reshape(-1).cpu()
            s = g[::53].numpy()
            ref = samples[off:off + len(s)]
            off += len(s)
            assert abs(float(g.double().norm()) - norm) <= 3e-2 * norm, name
            err = float(np.abs(s - ref).max()) / (norm + 1e-30)
            worst = max(worst, err)
            num += float(((s - ref).astype(np.float64) ** 2).sum())
            den += float((ref.astype(np.float64) ** 2).sum())
            assert err <= 3e-2, (name, err)
        print(f"gradients: joint relative error {(num / den) ** 0.5:.2e}, worst tensor {worst:.2e}")
        assert (num / den) ** 0.5 <= 1e-3
        assert all(p.grad is None for n, p in m.named_parameters() if not n.startswith("post_latent."))
    print(f"worst gradient error / norm: {worst:.2e}")


def test_rem_train_step_without_mu_std_matches_reference_gradients():
    """``--model rem`` without ``--mu_std`` (the block refines the scale only; rem_pic.py:194-195,214-220): one fine-tune
    step against the gradients the REFERENCE computed (tests/golden/rem_train_step_no_mu_std.npz), same bounds as the
    mu_std step above; dL/dmu stops at the block (mu is not refined)."""
    import argparse
    import vampic
    from conftest import README_ARGS
    gold = np.load(os.path.join(GOLD, "rem_train_step_no_mu_std.npz"))
    m = vampic.get_model(argparse.Namespace(model="rem", check_levels=[0.75], mu_std=False, dimension="middle", **README_ARGS), "cpu")
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed=0))
    m = m.cuda().train()
    m.freeze_all()
    m.unfreeze_rems()
    x = synth.synth_image(1, 64, 128, seed=0).cuda()
    noise = {"y": synth.uniform((1, 640, 4, 8), 101) - 0.5, "z": synth.uniform((1, 192, 1, 2), 102) - 0.5}
    ck = torch.from_numpy(gold["ck"]).cuda()
    for use_graph in (False, True):
        m.use_graph = use_graph
        m.zero_grad(set_to_none=True)
        out = m.forward_single_quality(x, quality=2.5, training=True, checkpoint_ref=ck, noise=noise)
        loss = _rate_loss(out, x)
        loss.backward()
        assert _rel(out["likelihoods"]["y"], torch.from_numpy(gold["lik_y"])) <= 5e-4
        assert abs(float(loss.detach()) - gold["loss"][0]) <= 1e-5 * gold["loss"][0]
        samples, off, num, den = gold["grad_samples"], 0, 0.0, 0.0
        params = dict(m.post_latent[0].named_parameters())
        assert len(params) == len(gold["grad_names"])
        for name, norm in zip(gold["grad_names"], gold["grad_norms"]):
            g = params[str(name)].grad.reshape(-1).cpu()
            s = g[::53].numpy()
            ref = samples[off:off + len(s)]
            off += len(s)
            assert abs(float(g.double().norm()) - norm) <= 3e-2 * norm + 1e-12, name
            num += float(((s - ref).astype(np.float64) ** 2).sum())
            den += float((ref.astype(np.float64) ** 2).sum())
        assert (num / den) ** 0.5 <= 1e-3, (num / den) ** 0.5
        assert all(p.grad is None for n, p in m.named_parameters() if not n.startswith("post_latent."))


def test_rem_finetune_loop_reduces_rate(train_model):
    """The reference's loop shape (training/step.py:56-95): checkpoint under no_grad, training forward, RateLoss,
    backward, clip, Adam.  The rate on a fixed batch must go down, the eval plan must see the new weights, and
    nothing outside post_latent may move."""
    m0, _ = train_model
    m = copy.deepcopy(m0)
    m.use_graph = True
    x = synth.synth_image(2, 64, 128, seed=3).cuda()
    noise = {"y": (synth.uniform((2, 640, 4, 8), 7) - 0.5).cuda(), "z": (synth.uniform((2, 192, 1, 2), 8) - 0.5).cuda()}
    frozen = {n: p.detach().clone() for n, p in m.named_parameters() if not n.startswith("post_latent.")}
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    with torch.no_grad():
        ck = m.ExtractChekpointRepr(x, quality=0.75, rc=False)
        before = m.forward_single_quality(x, quality=2.5, training=False, checkpoint_ref=ck)["likelihoods"]["y"].clone()
    losses = []
    for _ in range(6):
        opt.zero_grad()
        out = m.forward_single_quality(x, quality=2.5, training=True, checkpoint_ref=ck, noise=noise)
        loss = _rate_loss(out, x)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        losses.append(float(loss.detach()))
    print("fine-tune losses:", [round(v, 4) for v in losses])
    assert losses[-1] < losses[0]
    with torch.no_grad():
        after = m.forward_single_quality(x, quality=2.5, training=False, checkpoint_ref=ck)["likelihoods"]["y"]
    assert not torch.equal(before, after)                     # cached eval plan was rebuilt on the new weights
    for n, p in m.named_parameters():
        if not n.startswith("post_latent."):
            assert torch.equal(p, frozen[n]), n
    # validation driver (training/step.py:136-202) on the fine-tuned model: same numbers as the manual eval pass
    from vampic.evaluate import valid_epoch
    from vampic.finetune import RateLoss
    m.eval()
    v_loss, v = valid_epoch(0, [x], RateLoss(), m, pr_list=[2.5], rems=[0.75])
    den = -math.log(2) * x.shape[0] * x.shape[2] * x.shape[3]
    with torch.no_grad():
        o = m.forward_single_quality(x, quality=2.5, training=False, checkpoint_ref=ck)
    want = float(torch.log(o["likelihoods"]["y"]).sum() / den + 2 * torch.log(o["likelihoods"]["z"]).sum() / den)
    assert abs(v_loss - want) <= 1e-5 * abs(want) and v["psnr"] > 0
    m.train()
    # random noise path (the reference's uniform_): finite, differentiable
    out = m.forward_single_quality(x, quality=2.5, training=True, checkpoint_ref=ck)
    assert out["likelihoods"]["y"].requires_grad and torch.isfinite(out["likelihoods"]["y"]).all()


def test_fused_finetune_forward_equals_two_pass(train_model):
    """forward_finetune (checkpoint latent derived inside the training plan) against the reference's two calls
    (ExtractChekpointRepr, then forward_single_quality(training=True, checkpoint_ref=...)): same bits out, same
    gradients."""
    m0, _ = train_model
    m = copy.deepcopy(m0)
    x = synth.synth_image(2, 64, 128, seed=5).cuda()
    noise = {"y": (synth.uniform((2, 640, 4, 8), 17) - 0.5).cuda(), "z": (synth.uniform((2, 192, 1, 2), 18) - 0.5).cuda()}
    res = []
    for fused in (False, True):
        m.zero_grad(set_to_none=True)
        if fused:
            out = m.forward_finetune(x, 3.0, noise=noise)
        else:
            with torch.no_grad():
                ck = m.ExtractChekpointRepr(x, quality=0.75, rc=False)
            out = m.forward_single_quality(x, quality=3.0, training=True, checkpoint_ref=ck, noise=noise)
        _rate_loss(out, x).backward()
        res.append((out["likelihoods"]["y"].detach().clone(), out["x_hat"].clone(),
                    torch.cat([p.grad.reshape(-1) for p in m.post_latent.parameters()])))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_rem_finetune_step_full_size_teacher_forced(train_model):
    """BASELINE configs[4] at its stated per-GPU size: one fused fine-tune step (check level 0.75 -> q = 2.5) on
    16 x 3 x 256 x 256.  The step's REM gradients are checked teacher-forced: autograd over the ORACLE's rem_block, fed
    the HIP plan's own REM inputs (checkpoint latent, base / progressive entropy parameters, attention mask) and its own
    dL/d(mu', sigma'), must reproduce the gradient of every parameter of all ten REMs of the level (1e-5 of its max, float64
    reference on the HIP path's LeakyReLU branch decisions) — the sum over all 16 images and 256 latent positions, i.e.
    the full-size reduction the wgrad kernel performs."""
    m0, sd = train_model
    m = copy.deepcopy(m0)
    m.use_graph = True
    B, H, W, q = 16, 256, 256, 2.5
    x = synth.synth_image(B, H, W, seed=21).cuda()
    noise = {"y": (synth.uniform((B, 640, 16, 16), 31) - 0.5).cuda(), "z": (synth.uniform((B, 192, 4, 4), 32) - 0.5).cuda()}
    m.zero_grad(set_to_none=True)
    out = m.forward_finetune(x, q, noise=noise)
    loss = _rate_loss(out, x)
    loss.backward()
    assert torch.isfinite(loss) and float(loss) > 0
    plan = [p for k, p in m._plans.items() if "train" in k and k[:3] == (B, H, W)][0]
    ck, epb, epp, att, _ = plan.rem_io
    N = 32
    nchw = lambda v: v.torch_nchw().detach().cpu().clone()
    recs = {id(r["block"]): r for level in plan.rem_tape["branch"] + plan.rem_tape["enc"] for r in level}
    names = ("enc_base_rep", "enc_progressive_entropy_params", "enc_base_entropy_params", "enc")     # rem.py:133-139
    flipped = 0
    for j in range(10):
        rem = m.post_latent[0][j]
        # The LeakyReLU branch decisions of the HIP forward, in the order the oracle's rem_block evaluates them.  A
        # pre-activation within float noise of 0 takes either branch depending on the summation order, and the two
        # branches' gradients differ by a discrete amount (one position of 16 x 256: ~3e-4 of the weight gradient's
        # max - seen for ATen fp32 against float64 as well as for the HIP path).  So the float64 reference is
        # teacher-forced on the HIP path's own decisions, and every decision that differs from float64's own must be
        # a proven kink event (|pre-activation| <= 1e-5 of the tensor's max, the conv kernels' error bound).
        signs = [nchw(recs[id(rb)][key]) > 0 for nm in names for rb in getattr(rem, nm) for key in ("h1a", "o2")]
        a = nchw(att[j])
        ins = (nchw(ck[j]), torch.cat([nchw(epb[j][0]), nchw(epb[j][1])], 1),
               torch.cat([nchw(epp[j][0]), nchw(epp[j][1])], 1), torch.cat([a, a], 1))
        dres = torch.cat([nchw(plan.dmu.window(j * N, N)), nchw(plan.dsg.window(j * N, N))], 1)
        it = iter(signs)

        def forced_leaky(t, slope):
            nonlocal flipped
            pos = next(it)
            diff = pos != (t.detach() > 0)
            if diff.any():
                flipped += int(diff.sum())
                assert t.detach().abs()[diff].max().item() <= 1e-5 * t.detach().abs().max().item(), "not a kink event"
            return torch.where(pos, t, slope * t)
        leaves = {"r." + n: t.detach().cpu().double().clone().requires_grad_(True) for n, t in rem.state_dict().items()}
        real, O.F.leaky_relu = O.F.leaky_relu, forced_leaky
        try:
            res = O.rem_block(leaves, "r.", *[t.double() for t in ins])
        finally:
            O.F.leaky_relu = real
        assert next(it, None) is None                      # every taped activation was consumed
        res.backward(dres.double())
        # forward agreement of the refined parameters on the same inputs
        got = torch.cat([nchw(plan.mu_f.window(j * N, N)), nchw(plan.std_f.window(j * N, N))], 1)
        assert _rel(got, res.detach()) <= 1e-5, j
        worst = 0.0
        for n, p in rem.named_parameters():
            assert p.grad is not None, (j, n)
            e = _rel(p.grad, leaves["r." + n].grad)
            worst = max(worst, e)
            assert e <= 1e-5, (j, n, e)
        print(f"slice {j}: worst gradient error vs float64 (HIP branch decisions): {worst:.2e}")
    print(f"LeakyReLU decisions that differ from float64's (all proven kink events): {flipped}")
    # the other REMs (other check levels) and everything frozen stay without gradients
    assert all(p.grad is None for n, p in m.named_parameters() if not n.startswith("post_latent.0."))
    # likelihood backward at full size against the oracle's autograd on the plan's own (y, mu', sigma', mask, noise)
    d = 320
    y = plan.y.torch_nchw().detach().cpu()
    mu = nchw(plan.mu_f).requires_grad_(True)
    sg = nchw(plan.std_f).requires_grad_(True)
    mk = nchw(plan.mask)
    lik = O.gaussian_likelihood_noise(((y[:, d:] - y[:, :d]) - mu) * mk, sg * mk, None, noise["y"][:, d:].cpu())
    assert float((lik.detach() - out["likelihoods"]["y"][:, d:].detach().cpu()).abs().max()) <= 1e-6
    den = -math.log(2) * B * H * W
    (torch.log(lik).sum() / den).backward()
    for got, ref in ((plan.dmu, mu.grad), (plan.dsg, sg.grad)):
        dd = (nchw(got) - ref).abs()
        assert float(dd.max()) <= 2e-5 * float(ref.abs().max()) + 1e-9, float(dd.max())


def test_training_outside_rem_fails_loudly(train_model):
    m0, _ = train_model
    m = copy.deepcopy(m0)
    m.unfreeze_decoder()
    x = synth.synth_image(1, 64, 64, seed=0).cuda()
    with pytest.raises(NotImplementedError):
        m.forward_single_quality(x, quality=2.5, training=True)
