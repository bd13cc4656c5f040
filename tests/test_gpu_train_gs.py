"""Decoder refinement (`--training_type refine_gs`, reference train.py:150-157,216-218; SURVEY K14 first schedule) on the
GPU: the backward kernels of the synthesis transform against torch autograd over the oracle, the full step against the
gradients the REFERENCE computed (tests/golden/refine_gs_step.npz)."""
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
from vampic import engine as E, gs_train as G, layers as Ly, ops    # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_elementwise_derivatives_match_autograd():
    shp = (2, 32, 4, 8)
    x = synth.normal(shp, 1, 2.0).requires_grad_(True)
    g = synth.normal(shp, 2)
    V = lambda t: ops.from_nchw(t.detach().cuda())
    N = lambda: ops.new_view(2, 4, 8, 32)
    # exact-erf GELU
    F.gelu(x).backward(g)
    o, d = N(), N()
    ops.ew(L.EW_GELU_FWD, [V(x)], [o])
    ops.ew(L.EW_GELU_BWD, [V(x), V(g)], [d])
    assert _rel(o.torch_nchw(), F.gelu(x)) <= 1e-6 and _rel(d.torch_nchw(), x.grad) <= 1e-5
    # sigmoid gate a * sigmoid(b) + x
    a = synth.normal(shp, 3).requires_grad_(True)
    b = synth.normal(shp, 4, 3.0).requires_grad_(True)
    xs = synth.normal(shp, 5)
    (a * torch.sigmoid(b) + xs).backward(g)
    da, db, fo = N(), N(), N()
    ops.ew(L.EW_GATE_FWD, [V(a), V(b), V(xs)], [fo])
    ops.ew(L.EW_GATE_BWD, [V(a), V(b), V(g)], [da, db])
    assert _rel(fo.torch_nchw(), a * torch.sigmoid(b) + xs) <= 1e-6
    assert _rel(da.torch_nchw(), a.grad) <= 1e-5 and _rel(db.torch_nchw(), b.grad) <= 1e-5
    # clamp_(0, 1) backward from the clamped value
    v = (synth.normal(shp, 6) * 0.7 + 0.5).requires_grad_(True)
    v.clamp(0, 1).backward(g)
    dv = N()
    ops.ew(L.EW_CLAMP_BWD, [V(v.clamp(0, 1)), V(g)], [dv])
    assert torch.equal(dv.torch_nchw().cpu(), v.grad)


def test_grouped_axpy_equals_the_single_updates():
    """vam_train_axpy_group (the backward of the torch.cat in front of a slice stack, pic.py:407-408,452-453): up to eight
    ``dst += coef * src`` updates of channel windows with their own extents and row pitches in one launch — the same bits
    as eight vam_train_elementwise(VAM_EW_AXPY) launches; bad argument lists are refused."""
    B, H, W = 2, 4, 8
    acc = [ops.new_view(B, H, W, c) for c in (320, 32, 160, 64)]
    for k, a in enumerate(acc):
        a.buf.copy_(synth.normal(tuple(a.buf.shape), 40 + k).cuda())
    dx = ops.new_view(B, H, W, 320 + 32 + 96 + 64)
    dx.buf.copy_(synth.normal(tuple(dx.buf.shape), 50).cuda())
    ups = [(acc[0], dx.window(0, 320), 1.0), (acc[1], dx.window(320, 32), 1.0), (acc[2].window(32, 96), dx.window(352, 96), -0.5),
           (acc[3], dx.window(448, 64), 2.0)]
    want = [a.buf.clone() for a in acc]
    ref = [ops.View(w, a.c0, a.C) for w, a in zip(want, acc)]
    ref_ups = [(ref[0], ups[0][1], 1.0), (ref[1], ups[1][1], 1.0), (ref[2].window(32, 96), ups[2][1], -0.5), (ref[3], ups[3][1], 2.0)]
    for dst, src, coef in ref_ups:
        ops.ew(L.EW_AXPY, [dst, src], [dst], coef=coef)
    arr = ops.axpy_jobs(ups)
    ops.axpy_group(arr)
    torch.cuda.synchronize()
    for a, w in zip(acc, want):
        assert torch.equal(a.buf, w)
    assert not torch.equal(acc[2].buf[..., 32:128], synth.normal(tuple(acc[2].buf.shape), 42).cuda()[..., 32:128])     # it did update
    with pytest.raises(AssertionError):
        ops.axpy_jobs([ups[0]] * 9)                                     # more than VAM_MAX_EW_GROUP
    bad = ops.axpy_jobs([ups[0]])
    bad[0].C = 6                                                        # not a multiple of 4
    with pytest.raises(L.VamError):
        ops.axpy_group(bad)


@pytest.mark.parametrize("dim,ws,hw", [(192, 8, (16, 24)), (320, 4, (8, 8))])
def test_attention_block_backward(dim, ws, hw):
    """Win_noShift_Attention (residual units, Swin block with shift / mask / relative-position bias, sigmoid gate): taped
    forward + backward lowering against autograd over the oracle's attention_block: output 1e-5, input gradient and every
    parameter gradient 5e-5 of their max."""
    B = 2
    m = Ly.Win_noShift_Attention(dim=dim, num_heads=8, window_size=ws, shift_size=ws // 2)
    sd = synth.synth_state_dict(m.state_dict(), 9)
    m.load_state_dict(sd)
    m.cuda()
    x = synth.normal((B, dim) + hw, 10)
    dy = synth.normal((B, dim) + hw, 11)
    plan, bw = E.Plan("cuda"), E.Plan("cuda")
    pk = G.TransformPacks(m)
    pk.record_refresh(plan)
    tape = []
    out = G._attention_block_fwd(plan, pk, m, ops.from_nchw(x.cuda()), tape)
    grads = {id(p): torch.full_like(p, float("nan")) for p in m.parameters()}
    dx = G._attention_block_bwd(bw, pk, tape[0], ops.from_nchw(dy.cuda()), grads, need_dx=True)
    plan.run()
    bw.run()
    torch.cuda.synchronize()
    leaves = {"m." + k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ref = O.attention_block(leaves, "m.", xr, ws)
    ref.backward(dy)
    assert _rel(out.torch_nchw(), ref) <= 1e-5
    assert _rel(dx.torch_nchw(), xr.grad) <= 5e-5
    for n, p in m.named_parameters():
        assert _rel(grads[id(p)], leaves["m." + n].grad) <= 5e-5, (n, _rel(grads[id(p)], leaves["m." + n].grad))


@pytest.mark.parametrize("cout", [192, 3])
def test_deconv_and_igdn_backward(cout):
    """ConvTranspose2d(k5, s2, p2, op1) (data gradient = a k5/s2 convolution, weight gradient = stride-2 wgrad with the
    roles exchanged) and IGDN incl. the NonNegativeParametrizer chain, against autograd over the oracle."""
    B, cin, H, W = 2, 192, 6, 8
    dec = Ly.ConvTranspose2d(cin, cout).cuda()
    sd_d = synth.synth_state_dict(dec.state_dict(), 21)
    dec.load_state_dict(sd_d)
    x = synth.normal((B, cin, H, W), 22)
    plan, bw = E.Plan("cuda"), E.Plan("cuda")
    mods = torch.nn.Sequential(dec) if cout == 3 else torch.nn.Sequential(dec, Ly.GDN(cout, inverse=True).cuda())
    if cout != 3:
        sd_g = synth.synth_state_dict(mods[1].state_dict(), 23)
        mods[1].load_state_dict(sd_g)
    pk = G.TransformPacks(mods)
    pk.record_refresh(plan)
    tape = []
    t = G._deconv_fwd(plan, pk, dec, ops.from_nchw(x.cuda()), tape)
    if cout != 3:
        t = G._gdn_fwd(plan, pk, mods[1], t, tape)
    dy = synth.normal((B, cout, 2 * H, 2 * W), 24)
    grads = {id(p): torch.full_like(p, float("nan")) for p in mods.parameters()}
    if cout == 3:                                   # the model's last layer: 3-channel gradient, zero-padded to 16
        d16 = ops.new_view(B, 2 * H, 2 * W, 16, zero=True)
        d16.buf[..., :3].copy_(dy.cuda().permute(0, 2, 3, 1))
        d = d16
    else:
        d = ops.from_nchw(dy.cuda())
        d = G._gdn_bwd(bw, pk, tape[1], d, grads)
    dx = G._deconv_bwd(bw, pk, tape[0], d, grads)
    plan.run()
    bw.run()
    torch.cuda.synchronize()
    leaves = {"d." + k: v.clone().requires_grad_(True) for k, v in sd_d.items()}
    xr = x.clone().requires_grad_(True)
    y = O.deconv_k(leaves, "d.", xr)
    if cout != 3:
        leaves.update({"g." + k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd_g.items()})
        y = O.gdn(leaves, "g.", y, True)
    y.backward(dy)
    assert _rel(t.torch_nchw(), y) <= 1e-5
    assert _rel(dx.torch_nchw(), xr.grad) <= 5e-5
    for n, p in dec.named_parameters():
        assert _rel(grads[id(p)], leaves["d." + n].grad) <= 5e-5, n
    if cout != 3:
        for n, p in mods[1].named_parameters():
            assert _rel(grads[id(p)], leaves["g." + n].grad) <= 5e-5, (n, _rel(grads[id(p)], leaves["g." + n].grad))


@pytest.fixture(scope="module")
def pic_model():
    import argparse
    from conftest import README_ARGS
    net = vampic.get_model(argparse.Namespace(model="pic", **README_ARGS), "cpu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0)
    net.load_state_dict(sd)
    return net.cuda(), sd


def _distortion_loss(out, x, lmbda=1e-2, weight=255.0 ** 2):
    """training/loss.py:126-187 (DistortionLoss): weight * lmbda * mse."""
    return weight * (lmbda * F.mse_loss(x, out["x_hat"]))


@pytest.mark.parametrize("use_graph", [False, True])
def test_refine_gs_step_matches_reference_gradients(pic_model, use_graph):
    """One refine_gs step (freeze_all(); unfreeze_decoder(); training-mode forward; DistortionLoss; backward) against
    the REFERENCE's own run (tests/golden/refine_gs_step.npz): loss 1e-5 relative; every gradient tensor within 2e-3 of
    its norm, all sampled entries jointly within 5e-4 (the frozen front end feeds a y_hat that differs from the CPU's by
    fp32 summation order; g_s itself is held to 5e-5 by the teacher-forced tests above)."""
    net0, sd = pic_model
    m = copy.deepcopy(net0).train()
    m.use_graph = use_graph
    m.freeze_all()
    m.unfreeze_decoder()
    gold = np.load(os.path.join(GOLD, "refine_gs_step.npz"))
    x = synth.synth_image(1, 64, 64, seed=3).cuda()
    for rep in range(2):                                          # the second pass replays the captured graphs
        m.zero_grad(set_to_none=True)
        out = m.forward_single_quality(x, quality=2.5, training=True)
        loss = _distortion_loss(out, x)
        loss.backward()
        assert abs(float(loss.detach()) - gold["loss"][0]) <= 1e-5 * gold["loss"][0], (float(loss), gold["loss"][0])
        assert np.abs(out["x_hat"].detach().cpu()[:, :, ::2, ::2].numpy() - gold["x_hat"]).max() <= 1e-4
        params = dict(m.g_s[1].named_parameters())
        off, num, den, worst = 0, 0.0, 0.0, 0.0
        for name, norm in zip(gold["grad_names"], gold["grad_norms"]):
            g = params[str(name)].grad.reshape(-1).cpu()
            s_ = g[::97].numpy()
            ref = gold["grad_samples"][off:off + len(s_)]
            off += len(s_)
            assert abs(float(g.double().norm()) - norm) <= 2e-3 * norm + 1e-12, (name, float(g.double().norm()), norm)
            worst = max(worst, float(np.abs(s_ - ref).max()) / (norm + 1e-30))
            num += float(((s_ - ref).astype(np.float64) ** 2).sum())
            den += float((ref.astype(np.float64) ** 2).sum())
        print(f"refine_gs gradients: joint relative error {(num / den) ** 0.5:.2e}, worst entry / norm {worst:.2e}")
        assert (num / den) ** 0.5 <= 5e-4
        assert all(p.grad is None for n, p in m.named_parameters() if not n.startswith("g_s.1."))


def test_refine_gs_teacher_forced_and_loop(pic_model):
    """(a) the step's gradients against autograd over the ORACLE's g_s fed the plan's own y_hat (teacher-forced: 5e-5);
    (b) the reference's loop shape (training/step.py:56-99: forward, loss, backward, clip, Adam) lowers the distortion on
    a fixed batch, leaves every frozen parameter untouched, and the eval plan sees the new weights."""
    net0, sd = pic_model
    m = copy.deepcopy(net0).train()
    m.freeze_all()
    m.unfreeze_decoder()
    x = synth.synth_image(2, 64, 128, seed=5).cuda()
    out = m.forward_single_quality(x, quality=1.5, training=True)
    loss = _distortion_loss(out, x)
    loss.backward()
    leaves = {k: (v.clone().requires_grad_(True) if k.startswith("g_s.1.") and v.dtype.is_floating_point else v) for k, v in sd.items()}
    xh = O.g_s(leaves, "g_s.1.", out["y_hat"].detach().cpu()).clamp(0, 1)
    ref_loss = 255.0 ** 2 * (1e-2 * F.mse_loss(x.cpu(), xh))
    ref_loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-5 * float(ref_loss)
    for n, p in m.g_s[1].named_parameters():
        assert _rel(p.grad, leaves["g_s.1." + n].grad) <= 1e-4, (n, _rel(p.grad, leaves["g_s.1." + n].grad))
    frozen = {n: p.detach().clone() for n, p in m.named_parameters() if not n.startswith("g_s.1.")}
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    with torch.no_grad():
        before = m.forward_single_quality(x, 1.5)["x_hat"].clone()
    losses = []
    for _ in range(6):
        opt.zero_grad()
        out = m.forward_single_quality(x, quality=1.5, training=True)
        loss = _distortion_loss(out, x)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        losses.append(float(loss.detach()))
    print("refine_gs losses:", [round(v, 3) for v in losses])
    assert losses[-1] < losses[0]
    with torch.no_grad():
        after = m.forward_single_quality(x, 1.5)["x_hat"]
    assert not torch.equal(before, after)                         # the eval plan was rebuilt on the new weights
    for n, p in m.named_parameters():
        if not n.startswith("g_s.1."):
            assert torch.equal(p, frozen[n]), n
    # refine_gs_ga (train.py:219-222: unfreeze_decoder + unfreeze_encoder): beyond the decoder-only plan -> the complete
    # training plan (full_train.py; parity in tests/test_gpu_first_train.py) — gradients for exactly the trainable set
    m.unfreeze_encoder()
    m.zero_grad(set_to_none=True)
    o = m.forward_single_quality(x, quality=1.5, training=True)
    (o["x_hat"].mean() + torch.log(o["likelihoods"]["y"]).mean()).backward()
    for n, p in m.named_parameters():
        assert (p.grad is not None) == n.startswith(("g_s.1.", "g_a.1.")), n
        assert p.grad is None or torch.isfinite(p.grad).all()


def _train_plan(m, B, H, W):
    return [p for k, p in m._plans.items() if "train_lrp" in k and k[:3] == (B, H, W)][0]


@pytest.mark.parametrize("use_graph", [False, True])
def test_refine_gs_lrp_step_matches_reference_gradients(pic_model, use_graph):
    """`refine_gs --lrp` (train.py:216-218 -> unfreeze_decoder(lrp=True), pic.py:171-184): g_s[1] and the ten progressive
    latent-residual-prediction stacks train.  One step against the REFERENCE's own run (tests/golden/refine_gs_lrp_step.npz:
    100 + 100 gradient tensors): loss 1e-5 relative, every gradient norm within 2e-3, all sampled entries jointly 5e-4."""
    net0, sd = pic_model
    m = copy.deepcopy(net0).train()
    m.use_graph = use_graph
    m.freeze_all()
    m.unfreeze_decoder(lrp=True)
    gold = np.load(os.path.join(GOLD, "refine_gs_lrp_step.npz"))
    x = synth.synth_image(1, 64, 64, seed=3).cuda()
    for rep in range(2):                                          # the second pass replays the captured graphs
        m.zero_grad(set_to_none=True)
        out = m.forward_single_quality(x, quality=2.5, training=True)
        loss = _distortion_loss(out, x)
        loss.backward()
        assert abs(float(loss.detach()) - gold["loss"][0]) <= 1e-5 * gold["loss"][0], (float(loss), gold["loss"][0])
        assert np.abs(out["x_hat"].detach().cpu()[:, :, ::4, ::4].numpy() - gold["x_hat"]).max() <= 1e-4
        params = dict(m.named_parameters())
        err = {"g_s.1.": [0.0, 0.0], "lrp_transforms_prog.": [0.0, 0.0]}
        off = 0
        for name, norm in zip(gold["grad_names"], gold["grad_norms"]):
            name = str(name)
            g = params[name].grad.reshape(-1).cpu()
            s_ = g[::389].numpy()
            ref = gold["grad_samples"][off:off + len(s_)]
            off += len(s_)
            assert abs(float(g.double().norm()) - norm) <= 2e-3 * norm + 1e-12, (name, float(g.double().norm()), norm)
            e = err["g_s.1." if name.startswith("g_s.1.") else "lrp_transforms_prog."]
            e[0] += float(((s_ - ref).astype(np.float64) ** 2).sum())
            e[1] += float((ref.astype(np.float64) ** 2).sum())
        print("refine_gs --lrp gradients, joint relative error:", {k: f"{(v[0] / v[1]) ** 0.5:.2e}" for k, v in err.items()})
        assert all((v[0] / v[1]) ** 0.5 <= 5e-4 for v in err.values())
        trained = ("g_s.1.", "lrp_transforms_prog.")
        assert all((p.grad is not None) == n.startswith(trained) for n, p in m.named_parameters())


def test_refine_gs_lrp_teacher_forced_and_loop(pic_model):
    """(a) the LRP stacks' gradients against autograd over the ORACLE's cc_stack fed the plan's own stack inputs and its
    own dL/dy_hat (teacher-forced: 5e-5 of each tensor's max); the taped forward agrees with the eval plan to fp32
    rounding (a trained stack runs its first layer whole, the eval plan splits off the hyperprior part);
    (b) an Adam loop over both parameter groups lowers the distortion, leaves everything else bit-identical, and the
    eval plan sees the new LRP weights."""
    net0, sd = pic_model
    m = copy.deepcopy(net0).train()
    m.freeze_all()
    m.unfreeze_decoder(lrp=True)
    B, H, W = 2, 64, 128
    x = synth.synth_image(B, H, W, seed=5).cuda()
    with torch.no_grad():
        ev = m.forward_single_quality(x, 1.5)
    out = m.forward_single_quality(x, quality=1.5, training=True)
    assert (out["y_hat"] - ev["y_hat"]).abs().max().item() <= 2e-5 and (out["x_hat"].detach() - ev["x_hat"]).abs().max().item() <= 1e-4
    loss = _distortion_loss(out, x)
    loss.backward()
    plan = _train_plan(m, B, H, W)
    nchw = lambda v: v.torch_nchw().detach().cpu().clone()
    for j in (0, 3, 9):
        tape = plan.lrp_tapes[j]
        inp = torch.cat([nchw(v) for v in tape["x"][0]], 1)
        pre = f"lrp_transforms_prog.{j}."
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(pre)}
        res = 0.5 * torch.tanh(O.cc_stack(leaves, pre, inp))
        res.backward(nchw(plan.d_yhat.window(32 * j, 32)))
        for n, p in m.lrp_transforms_prog[j].named_parameters():
            assert _rel(p.grad, leaves[pre + n].grad) <= 5e-5, (j, n, _rel(p.grad, leaves[pre + n].grad))
    trained = ("g_s.1.", "lrp_transforms_prog.")
    frozen = {n: p.detach().clone() for n, p in m.named_parameters() if not n.startswith(trained)}
    lrp_before = m.lrp_transforms_prog[4][0].weight.detach().clone()
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        o = m.forward_single_quality(x, quality=1.5, training=True)
        ls = _distortion_loss(o, x)
        ls.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        losses.append(float(ls.detach()))
    print("refine_gs --lrp losses:", [round(v, 3) for v in losses])
    assert losses[-1] < losses[0]
    assert not torch.equal(lrp_before, m.lrp_transforms_prog[4][0].weight)
    with torch.no_grad():
        after = m.forward_single_quality(x, 1.5)
    assert not torch.equal(after["y_hat"], ev["y_hat"])            # the eval plan was rebuilt on the new LRP weights
    for n, p in m.named_parameters():
        if not n.startswith(trained):
            assert torch.equal(p, frozen[n]), n
    # an LRP subset, or the LRP stacks at quality 0 (base decoder), is not a schedule of the reference: loud
    for p in m.lrp_transforms_prog[3].parameters():
        p.requires_grad = False
    m.zero_grad(set_to_none=True)                                  # not a schedule of the reference: the complete training plan serves it
    o = m.forward_single_quality(x, quality=1.5, training=True)
    o["x_hat"].mean().backward()
    assert all(p.grad is None for p in m.lrp_transforms_prog[3].parameters())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.lrp_transforms_prog[2].parameters())


def test_refine_gs_epoch_driver(pic_model):
    """vampic.finetune's decoder-refinement driver (train.py:150-157,216-218 + training/step.py:56-99 with
    sampling_training=True): setup freezes / unfreezes as the reference does, an epoch over a fixed batch samples one quality
    per step from the reference's list, applies DistortionLoss and Adam, and the distortion at a fixed quality goes down."""
    import random
    from vampic import finetune as FT
    net0, sd = pic_model
    m = copy.deepcopy(net0)
    params = FT.refine_gs_setup(m, lrp=True)
    assert {n.split(".")[0] for n, p in m.named_parameters() if p.requires_grad} == {"g_s", "lrp_transforms_prog"}
    assert all(n.startswith("g_s.1.") or n.startswith("lrp_transforms_prog.") for n, p in m.named_parameters() if p.requires_grad)
    qs = FT.refine_gs_quality_list()
    assert len(qs) == 254 and qs[0] == 0.015 and qs[-1] == 10.0 and all(b > a for a, b in zip(qs, qs[1:]))
    crit = FT.DistortionLoss(device="cuda")
    x = synth.synth_image(2, 64, 64, seed=8).cuda()

    def fixed_loss():
        with torch.no_grad():
            return float(torch.nn.functional.mse_loss(x, m.eval().forward_single_quality(x, 2.5)["x_hat"]))
    before = fixed_loss()
    opt = torch.optim.Adam(params, lr=1e-4)
    # (a) the sampled schedule runs: one quality of the reference's list per step
    counter, loss, bpp, mse, bpp_s = FT.train_one_epoch_refine_gs(m, crit, [x] * 3, opt, epoch=0, counter=0, rng=random.Random(3))
    assert counter == 3 and math.isfinite(loss) and loss > 0 and bpp > 0 and bpp_s == 0.0 and mse > 0
    # (b) and it optimises: a dozen steps at ONE quality lower that quality's distortion (a handful of Adam steps at
    #     randomly sampled qualities need not lower the distortion at a fixed one)
    counter, loss, _, _, _ = FT.train_one_epoch_refine_gs(m, crit, [x] * 12, opt, epoch=1, counter=counter, list_quality=[2.5])
    assert counter == 15 and math.isfinite(loss)
    assert fixed_loss() < before


def test_training_step_is_reproducible_under_graph_replay(pic_model):
    """Thirty replays of the captured forward + backward graphs at FIXED parameters must give the same gradients every
    time (bit for bit, except the two relative-position-bias tables, which are summed with float atomics: 1e-6).
    Regression test: a hipMemsetAsync node in front of the attention-table atomics intermittently left garbage (1e14 ...
    1e33) in single table elements under graph replay, which zeroed every other gradient through clip_grad_norm_ and made
    the refine_gs --lrp loop diverge after a few steps; the clear is a kernel node now (vam_memset_zero)."""
    net0, sd = pic_model
    m = copy.deepcopy(net0).train()
    m.use_graph = True
    m.freeze_all()
    m.unfreeze_decoder(lrp=True)
    x = synth.synth_image(2, 64, 64, seed=8).cuda()
    ref = None
    for it in range(30):
        m.zero_grad(set_to_none=True)
        out = m.forward_single_quality(x, quality=2.5, training=True)
        _distortion_loss(out, x).backward()
        g = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        junk = [torch.randn(1 << 20, device="cuda") for _ in range(4)]          # allocator churn, as an optimiser causes
        del junk
        if ref is None:
            ref = g
            continue
        for n in g:
            assert torch.isfinite(g[n]).all(), (it, n)
            if n.endswith("relative_position_bias_table"):
                assert (g[n] - ref[n]).abs().max().item() <= 1e-6 * ref[n].abs().max().item(), (it, n)
            else:
                assert torch.equal(g[n], ref[n]), (it, n)


def test_matrix_pipe_attention_agrees_with_the_fma_kernels():
    """8 x 8 windows run on the fp32 matrix pipe (csrc/win_attn.hip win_attn8_mfma_kernel, csrc/train_gs.hip
    win_attn8_bwd_mfma_kernel): exact fp32 products like the FMA kernels they replace, in a different summation order and
    with exp2 instead of expf — forward output and all four gradients (dq, dk, dv, d bias table) agree to 2e-6 of their
    magnitude, with and without the cyclic shift."""
    lib = L.load()
    B, H, W, C_, heads, ws = 2, 16, 24, 192, 8, 8
    qkv = ops.from_nchw(synth.normal((B, 3 * C_, H, W), 71).cuda())
    dout = ops.from_nchw(synth.normal((B, C_, H, W), 72).cuda())
    tab = synth.normal(((2 * ws - 1) ** 2, heads), 73, 0.5).cuda()
    for shift in (0, 4):
        res = {}
        for mode in (1, 0):
            lib.vam_attn_set_mfma(mode)
            try:
                out = ops.new_view(B, H, W, C_, "cuda")
                ops.win_attention(qkv, out, tab, C_, heads, ws, shift)
                dq = ops.new_view(B, H, W, 3 * C_, "cuda")
                dtab = torch.zeros_like(tab)
                wsb = torch.empty(lib.vam_win_attention_bwd_workspace(B, H, W, heads, ws) // 4, device="cuda")
                L.check(lib.vam_win_attention_bwd(qkv.ptr, qkv.ld, dout.ptr, dout.ld, dq.ptr, dq.ld, tab.data_ptr(), dtab.data_ptr(),
                                                  wsb.data_ptr(), B, H, W, C_, heads, ws, shift, ops.stream_ptr()), "vam_win_attention_bwd")
                torch.cuda.synchronize()
                res[mode] = (out.buf.clone(), dq.buf.clone(), dtab.clone())
            finally:
                lib.vam_attn_set_mfma(-1)
        for a, b, what in zip(res[1], res[0], ("out", "dqkv", "dtable")):
            assert torch.isfinite(a).all(), what
            err = (a - b).abs().max().item()
            assert err <= 2e-6 * max(1.0, b.abs().max().item()) * (8 if what == "dtable" else 1), (shift, what, err, b.abs().max().item())
