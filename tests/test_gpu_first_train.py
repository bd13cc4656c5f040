"""First-stage training (BASELINE configs[3], `--training_type first_train`, reference train.py:146-149) on the GPU:
the new backward pieces against torch autograd over the oracle on the same inputs, then the complete step — training
forward (pic.py:301-491 / :497-666), ScalableRateDistortionLoss, backward of every parameter — against the gradients the
REFERENCE computed (tests/golden/first_train_step.npz, train_single_step.npz)."""
import argparse
import os
import warnings

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
from vampic import full_train as FT        # noqa: E402
from conftest import README_ARGS           # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _stack_check(stack, prefix_fn, segs, out_shape_fn, seed, tol=5e-5):
    """Taped forward + backward of ONE conv stack on multi-segment input vs autograd over the oracle function."""
    sd = synth.synth_state_dict(stack.state_dict(), seed)
    stack.load_state_dict(sd)
    stack.cuda()
    xs = [synth.normal(s, seed + 1 + i) for i, s in enumerate(segs)]
    plan, bw = E.Plan("cuda"), E.Plan("cuda")
    pk = G.TransformPacks(stack)
    pk.record_refresh(plan)
    tapes = FT.lower_stacks_train(plan, [stack], [[ops.from_nchw(x.cuda()) for x in xs]], [None], [pk])
    leaves = {"m." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = [x.clone().requires_grad_(True) for x in xs]
    ref = prefix_fn(leaves, torch.cat(xr, 1))
    dy = synth.normal(tuple(ref.shape), seed + 9)
    grads = {id(p): torch.full_like(p, float("nan")) for p in stack.parameters()}
    (dx,) = FT.lower_stacks_backward(bw, tapes, [ops.from_nchw(dy.cuda())], [pk], grads)
    plan.run()
    bw.run()
    torch.cuda.synchronize()
    ref.backward(dy)
    assert _rel(tapes[0]["out"].torch_nchw(), ref) <= 1e-5
    assert _rel(dx.torch_nchw(), torch.cat([x.grad for x in xr], 1)) <= tol
    for n, p in stack.named_parameters():
        e = _rel(grads[id(p)], leaves["m." + n].grad)
        assert e <= tol, (n, e)


def test_slice_stack_backward_with_segmented_input():
    """Five conv3x3 + GELU on cat(hyperprior, support, slice) (pic.py:83-164): weight / bias gradients and the input
    gradient over all segments."""
    from vampic.models import _param_stack
    _stack_check(_param_stack(320 + 64 + 32, 320), lambda sd, x: O.cc_stack(sd, "m.", x),
                 [(2, 320, 4, 8), (2, 64, 4, 8), (2, 32, 4, 8)], None, 31)


@pytest.mark.parametrize("planes", ["0", "1"])
def test_slice_stack_backward_on_the_latent_grid_of_the_full_size_step(planes, monkeypatch):
    """The same stack on a 16 x 16 grid (what a 256 x 256 image gives): weight gradients take the LDS-tiled kernel; with
    VAMPIC_TRAIN_P3=1 the intermediate activations travel as bf16x3 planes (forward convolution and weight gradient both
    read them as planes)."""
    from vampic.models import _param_stack
    monkeypatch.setenv("VAMPIC_TRAIN_P3", planes)
    assert ops.wgrad_reads_planes(16, 16) and ops.train_tape_planes(16, 16) == (planes == "1")
    _stack_check(_param_stack(320 + 64 + 32, 320), lambda sd, x: O.cc_stack(sd, "m.", x),
                 [(2, 320, 16, 16), (2, 64, 16, 16), (2, 32, 16, 16)], None, 33)


def test_hyper_analysis_backward_stride2():
    """h_a (builder.py:72-82): conv3x3 stride 2 — data gradient by zero insertion + the stride-1 data-gradient problem,
    weight gradient by the stride-2 wgrad."""
    from vampic.models import define_hyperprior
    h_a, _, _ = define_hyperprior(True, 640, 192, [320, 640])

    def f(sd, x):
        sd2 = {"h_a." + k[2:]: v for k, v in sd.items()}
        return O.h_a(sd2, x)
    _stack_check(h_a, f, [(2, 640, 8, 16)], None, 41)


def test_hyper_synthesis_backward_pixel_shuffle():
    """h_mean_s / h_scale_s (builder.py:88-98): subpel_conv3x3 = conv + PixelShuffle(2): gradient un-shuffle."""
    from vampic.models import define_hyperprior
    _, hm, _ = define_hyperprior(True, 640, 192, [320, 640])
    _stack_check(hm[0], lambda sd, x: O.h_s(sd, "m.", x), [(2, 192, 2, 3)], None, 51)


def test_analysis_transform_backward():
    """g_a (builder.py:43-53): conv5x5 s2 (data gradient = four sub-pixel phase problems, first layer on the
    space-to-depth input), GDN, both attention blocks — every parameter gradient vs autograd over the oracle."""
    from vampic.models import define_encoder
    enc = define_encoder(True, 192, 640, [320, 640])[0]
    sd = synth.synth_state_dict(enc.state_dict(), 61)
    enc.load_state_dict(sd)
    enc.cuda()
    B, H, W = 2, 64, 128
    x = synth.synth_image(B, H, W, seed=7)
    plan, bw = E.Plan("cuda"), E.Plan("cuda")
    pk = G.TransformPacks(enc)
    pk.record_refresh(plan)
    xs = ops.s2d_input(x.cuda())
    y = ops.new_view(B, H // 16, W // 16, 320)
    tape = FT.lower_g_a_train(plan, enc, xs, y, pk)
    dy = synth.normal((B, 320, H // 16, W // 16), 62)
    grads = {id(p): torch.full_like(p, float("nan")) for p in enc.parameters()}
    FT.lower_g_a_backward(bw, tape, ops.from_nchw(dy.cuda()), pk, grads)
    plan.run()
    bw.run()
    torch.cuda.synchronize()
    leaves = {"m." + k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "reparam" not in k else v) for k, v in sd.items()}
    ref = O.g_a(leaves, "m.", x)
    ref.backward(dy)
    assert _rel(y.torch_nchw(), ref) <= 2e-5
    for n, p in enc.named_parameters():
        e = _rel(grads[id(p)], leaves["m." + n].grad)
        assert e <= 1e-4, (n, e)


def test_entropy_bottleneck_noise_likelihood_backward():
    """vam_eb_train_bwd: dL/dz and the 14 density-network gradients of the additive-noise likelihood
    (entropy_models.py:403-436,449-492) vs autograd over the oracle (quantiles: zero)."""
    from vampic.entropy_models import EntropyBottleneck
    C = 192
    eb = EntropyBottleneck(C)
    sd = synth.synth_state_dict({"entropy_bottleneck." + k: v for k, v in eb.state_dict().items()}, 71)
    eb.load_state_dict({k[len("entropy_bottleneck."):]: v for k, v in sd.items()})
    eb.cuda()
    z = synth.normal((3, C, 2, 4), 72, 3.0)
    nz = synth.uniform((3, C, 2, 4), 73) - 0.5
    g = synth.normal((3, C, 2, 4), 74)
    names = ["_matrix0", "_bias0", "_factor0", "_matrix1", "_bias1", "_factor1", "_matrix2", "_bias2", "_factor2",
             "_matrix3", "_bias3", "_factor3", "_matrix4", "_bias4", "quantiles"]
    params = torch.cat([getattr(eb, n).detach().reshape(-1) for n in names]).contiguous()
    V = lambda t: ops.from_nchw(t.cuda())
    lik, dz = ops.new_view(3, 2, 4, C), ops.new_view(3, 2, 4, C)
    dpar = torch.full_like(params, float("nan"))
    ops.eb_forward(V(z), params, None, lik, None, noise=V(nz))
    ops.eb_train_bwd(V(z), V(nz), params, V(g), dz, dpar)
    torch.cuda.synchronize()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and k.split(".")[-1] in names}
    sdt = dict(sd)
    sdt.update(leaves)
    zr = z.clone().requires_grad_(True)
    ref = O.eb_likelihood_noise_bounded(sdt, zr, nz)
    ref.backward(g)
    assert _rel(lik.torch_nchw(), ref) <= 1e-5
    assert _rel(dz.torch_nchw(), zr.grad) <= 2e-5
    off = 0
    for n in names:
        p = getattr(eb, n)
        got = dpar[off:off + p.numel()].view(p.shape)
        off += p.numel()
        want = leaves["entropy_bottleneck." + n].grad
        if n == "quantiles":
            assert float(got.abs().max()) == 0.0 and (want is None or float(want.abs().max()) == 0.0)
        else:
            assert _rel(got, want) <= 5e-5, (n, _rel(got, want))


# ----------------------------------------------------------------------------------------------- the complete step
def _model():
    net = vampic.get_model(argparse.Namespace(model="pic", **README_ARGS), "cpu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0)
    net.load_state_dict(sd)
    return net.cuda().train(), sd


FAMILIES = ("g_a.0.", "g_a.1.", "g_s.0.", "g_s.1.", "h_a.", "h_mean_s.", "h_scale_s.", "cc_mean_transforms.", "cc_scale_transforms.",
            "lrp_transforms.", "cc_mean_transforms_prog.", "cc_scale_transforms_prog.", "lrp_transforms_prog.", "entropy_bottleneck.")


def _plan_of(net):
    return next(p for k, p in net._plans.items() if k[0] == "full_train")


def _forced_oracle_step(sd, net, x, ny, nz, qualities, lmbda, single, **variant):
    """The oracle's step (pinned to the REFERENCE's own run by tests/test_oracle_golden.py) with the GPU pass's hard
    decisions imposed: a latent within fp32 summation noise of x.5 rounds either way depending on the convolution's
    summation order (|y| reaches 46 with the synthetic weights: 3e-4 absolute noise, a dozen such elements per image), and
    every later slice is conditioned on it — so values and gradients are compared at EQUAL decisions."""
    pl = _plan_of(net)
    nchw = lambda v: v.torch_nchw().detach().cpu()
    force = {"base_sym": torch.round(nchw(pl.yq) - nchw(pl.mu_b)),
             "z_sym": torch.round(nchw(pl.z_hat) - sd["entropy_bottleneck.quantiles"][:, 0, 1].reshape(1, -1, 1, 1))}
    if not pl.base_only:
        force["prog_sym"] = torch.round(nchw(pl.rq) - nchw(pl.mu_p))
        force["mask"] = nchw(pl.mask)
    # the eager and the hipGraph pass of one test take the same decisions (same kernels, same order): one oracle step serves both
    key = (_sd_print(sd), float(x.double().sum()), tuple(qualities), str(lmbda), single, tuple(sorted(variant.items())))
    hit = _ORACLE_CACHE.get("last")
    if hit is not None and hit[0] == key and all(torch.equal(hit[2][k], force[k]) for k in force):
        return hit[1], force
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = O.first_train_step(sd, x, qualities, ny, nz, lmbda, single=single, force=force, **variant)
    _ORACLE_CACHE["last"] = (key, ref, force)
    return ref, force


_ORACLE_CACHE: dict = {}


def _sd_print(sd) -> tuple:
    """Cheap fingerprint of a state dict (tests build the same synthetic model more than once)."""
    ks = [k for k in sorted(sd) if torch.is_tensor(sd[k]) and sd[k].dtype.is_floating_point]
    return (len(sd),) + tuple(float(sd[k].double().sum()) for k in (ks[0], ks[len(ks) // 2], ks[-1]))


def _decision_audit(net, sd, free, force, tol=1e-3):
    """Hard decisions of the GPU's training forward (``force``: the symbols it took) against the oracle's unforced pass
    ``free``, stage by stage in dependency order.  Returns {"first": stage name or None, "explained": boundary events of
    the first differing stage, "violations": its non-boundary differences, "downstream": differences of later stages}."""
    med = sd["entropy_bottleneck.quantiles"][:, 0, 1].reshape(1, -1, 1, 1)
    y, C = free["y"].detach(), 32
    stages = [("z", force["z_sym"], free["z"].detach() - med)]
    for i in range(10):
        stages.append((f"base slice {i}", force["base_sym"][:, i * C:(i + 1) * C], y[:, i * C:(i + 1) * C] - free["mu_base"].detach()[:, i * C:(i + 1) * C]))
    if "prog_sym" in force:
        for j in range(10):
            r = y[:, 320 + j * C:320 + (j + 1) * C] - y[:, j * C:(j + 1) * C]
            stages.append((f"progressive slice {j}", force["prog_sym"][:, j * C:(j + 1) * C], r - free["mu"].detach()[:, j * C:(j + 1) * C]))
    rep = {"first": None, "explained": 0, "violations": 0, "downstream": 0}
    for name, sym, t in stages:
        differ = sym != torch.round(t)
        n = int(differ.sum())
        if n == 0:
            continue
        if rep["first"] is None:
            dist = 0.5 - (t - torch.round(t)).abs()
            bad = int((differ & (dist >= tol)).sum())
            rep.update(first=name, explained=n - bad, violations=bad)
        else:
            rep["downstream"] += n
    return rep


def _compare_grads(net, ref_grads):
    params = dict(net.named_parameters())
    want = sorted(k for k, g in ref_grads.items() if g is not None)
    got = sorted(k for k, p in params.items() if p.grad is not None)
    assert got == want, (set(got) ^ set(want))
    fam = {f: [0.0, 0.0] for f in FAMILIES}
    num = den = 0.0
    worst = ("", 0.0)
    for name in want:
        g, r = params[name].grad.detach().double().cpu(), ref_grads[name].double()
        assert torch.isfinite(g).all(), name
        e, rr = float(((g - r) ** 2).sum()), float((r ** 2).sum())
        num, den = num + e, den + rr
        f = next((f for f in FAMILIES if name.startswith(f)), None)
        if f is None:                                   # single encoder / decoder / hyperprior: "g_a.", "g_s.", ...
            f = name.split(".")[0] + "."
            fam.setdefault(f, [0.0, 0.0])
        fam[f][0] += e
        fam[f][1] += rr
        rel = (e / max(rr, 1e-300)) ** 0.5
        if rr > 1e-12 * den and rel > worst[1]:
            worst = (name, rel)
    return (num / den) ** 0.5, {f: (v[0] / max(v[1], 1e-300)) ** 0.5 for f, v in fam.items()}, worst


def _check_step(net, sd, out, crit, x, ny, nz, qualities, lmbda, single, tag):
    ref, force = _forced_oracle_step(sd, net, x, ny, nz, qualities, lmbda, single)
    for k in ("loss", "bpp_loss", "bpp_base", "bpp_scalable", "bpp_hype"):
        a, b = float(crit[k].detach().mean()), float(ref["crit"][k].mean())
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (k, a, b)
    assert _rel(crit["mse_loss"], ref["crit"]["mse_loss"]) <= 1e-4
    assert _rel(out["x_hat"], ref["out"]["x_hat"]) <= 1e-4
    assert _rel(out["likelihoods"]["y"], ref["out"]["likelihoods"]["y"]) <= 1e-4
    assert _rel(out["likelihoods"]["z"], ref["out"]["likelihoods"]["z"]) <= 1e-4
    joint, fam, worst = _compare_grads(net, ref["grads"])
    print(tag, "joint gradient error", joint, "worst tensor", worst, {k: f"{v:.2e}" for k, v in fam.items()})
    # measured (r03): joint 5.5e-5; g_s / entropy bottleneck / progressive LRP 2e-6 ... 2e-5; the families upstream of the
    # rate term 5e-5 ... 1.6e-4 — the likelihood's argument y + noise - mu is a difference of numbers up to 46 whose fp32
    # summation noise (3e-4 absolute) is 1e-4 of its O(1) value
    assert joint <= 2e-4, (joint, fam)
    for f, v in fam.items():
        assert v <= 5e-4, (f, v, fam)
    return force


@pytest.mark.parametrize("use_graph", [False, True])
def test_first_train_step_matches_reference(use_graph):
    """forward(x, quality=[0, 10], training=True) + ScalableRateDistortionLoss + backward with EVERY parameter trainable:
    loss terms, reconstructions, likelihoods and all 1065 gradient tensors against the oracle's step at equal rounding
    decisions (joint gradient error <= 2e-4, every family <= 5e-4; measured 5.5e-5 / 1.6e-4).  The oracle's step is the reference's: the CPU suite
    holds it to tests/golden/first_train_step.npz (1e-5)."""
    from vampic.finetune import ScalableRateDistortionLoss
    from test_oracle_golden import train_fixture_inputs
    net, sd = _model()
    net.use_graph = use_graph
    x, ny, nz = train_fixture_inputs()
    out = net(x.cuda(), quality=[0, 10], training=True, noise={"y": ny, "z": nz})
    crit = ScalableRateDistortionLoss(lmbda_list=[0.0055, 0.04], device="cuda")(out, x.cuda())
    crit["loss"].backward()
    force = _check_step(net, sd, out, crit, x, ny, nz, [0, 10], [0.0055, 0.04], False, f"first_train graph={use_graph}")
    # how far the decisions themselves are from the reference's run: the oracle's UNFORCED step is the
    # reference's run (CPU suite, tests/golden/first_train_step.npz); walking the hard decisions in dependency order
    # (z, base slices 0..9, progressive slices 0..9), the FIRST stage that differs must differ only in rounding-boundary
    # events of the oracle's own numbers (residual within 1e-3 of x.5); everything after it is conditioned on them and is
    # only counted.  A wrong kernel differs in non-boundary elements of an untainted stage.
    fkey = ("free", _sd_print(sd), float(x.double().sum()))
    if _ORACLE_CACHE.get("free", (None,))[0] != fkey:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _ORACLE_CACHE["free"] = (fkey, O.training_forward(sd, x, [0, 10], ny, nz))
    free = _ORACLE_CACHE["free"][1]
    rep = _decision_audit(net, sd, free, force)
    gold = np.load(os.path.join(GOLD, "first_train_step.npz"))
    flips = int((( _plan_of(net).y_base.torch_nchw().cpu() - torch.from_numpy(gold["y_hat_base"])).abs() > 0.4).sum())
    print(f"first_train graph={use_graph}: decision audit {rep}; base latents differing from the reference's run "
          f"(incl. conditioned ones): {flips} of {gold['y_hat_base'].size}")
    from conftest import record_measurement
    record_measurement(f"first_train step vs the reference's run (graph={use_graph})", first_differing_stage=rep["first"],
                       boundary_events=rep["explained"], violations=rep["violations"], downstream=rep["downstream"],
                       base_latents_differing=f"{flips}/{gold['y_hat_base'].size}")
    assert rep["violations"] == 0, rep
    assert rep["explained"] <= 8, rep                    # measured (r04): see DESIGN.md section 9c
    assert flips <= 0.05 * gold["y_hat_base"].size


def test_single_quality_training_step_matches_reference():
    """forward_single_quality(x, 2.5, training=True) with every parameter trainable (variance mask, straight-through
    rounding under the mask, clamp) against the oracle's step at equal decisions (pinned to the reference by
    tests/golden/train_single_step.npz on the CPU)."""
    from vampic.finetune import ScalableRateDistortionLoss
    from test_oracle_golden import train_fixture_inputs
    net, sd = _model()
    x, ny, nz = train_fixture_inputs()
    out = net.forward_single_quality(x.cuda(), 2.5, training=True, noise={"y": ny, "z": nz})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        crit = ScalableRateDistortionLoss(lmbda_list=[0.0055, 0.04], device="cuda")(out, x.cuda(), lmbda=0.01)
    crit["loss"].backward()
    force = _check_step(net, sd, out, crit, x, ny, nz, [2.5], 0.01, True, "single quality")
    # the mask the GPU used is the oracle's own mask of the GPU's sigma (bit-exact operator)
    pl = _plan_of(net)
    sig = pl.std_p.torch_nchw().cpu()
    assert torch.equal(force["mask"], torch.cat([O.variance_mask(s_, 2.5) for s_ in sig.chunk(10, 1)], 1))


def test_refine_gs_ga_step():
    """`--training_type refine_gs_ga` (train.py:158-166,219-222): g_s[1] and g_a[1] train, rate-distortion loss at the
    sampled quality's Lagrangian.  Gradients of the trainable set against the oracle's step at equal decisions; every other
    parameter keeps grad None and its value through an optimiser step."""
    from vampic import finetune as ft
    from test_oracle_golden import train_fixture_inputs
    net, sd = _model()
    params = ft.refine_gs_ga_setup(net)
    assert {n.split(".")[0] + "." + n.split(".")[1] for n, p in net.named_parameters() if p.requires_grad} == {"g_s.1", "g_a.1"}
    lms = ft.refine_gs_ga_lambdas([0.0055, 0.04], 254)
    assert len(lms) == 254 and abs(lms[-1] - 0.04) < 1e-6 and lms[0] > 0.0055
    x, ny, nz = train_fixture_inputs()
    crit_fn = ft.RateDistortionLoss(device="cuda")
    with pytest.raises(AttributeError):
        crit_fn({"x_hat": x.cuda(), "likelihoods": {"y": x.cuda(), "z": x.cuda()}}, x.cuda())
    out = net.forward_single_quality(x.cuda(), 2.5, training=True, noise={"y": ny, "z": nz})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        crit = crit_fn(out, x.cuda(), lmbda=0.01)
    crit["loss"].backward()
    ref, _ = _forced_oracle_step(sd, net, x, ny, nz, [2.5], 0.01, True)
    want = {k: g for k, g in ref["grads"].items() if k.startswith(("g_s.1.", "g_a.1."))}
    joint, fam, worst = _compare_grads(net, want)
    print("refine_gs_ga joint gradient error", joint, worst)
    assert joint <= 2e-4
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    opt = torch.optim.Adam(params, lr=1e-4)
    ft.refine_gs_ga_step(net, crit_fn, x.cuda(), opt, 2.5, 0.01, noise={"y": ny, "z": nz})
    for n, p in net.named_parameters():
        assert torch.equal(p, before[n]) != n.startswith(("g_s.1.", "g_a.1.")), n


def test_adam_loop_lowers_the_loss():
    """Ten Adam steps of the first_train schedule (training/step.py:56-99: zero_grad, forward([0, 10]), criterion, backward,
    clip 1.0, step) on a fixed batch lower the loss; hipGraph replay across the optimiser's in-place updates."""
    from vampic.finetune import ScalableRateDistortionLoss, first_train_step
    from vampic.checkpoint import configure_optimizers
    net, sd = _model()
    args = argparse.Namespace(learning_rate=1e-4, aux_learning_rate=1e-3, training_type="first_train")
    opt, _ = configure_optimizers(net, args)
    crit = ScalableRateDistortionLoss(lmbda_list=[0.0055, 0.04], device="cuda")
    x = synth.synth_image(2, 64, 64, seed=8).cuda()
    noise = {"y": synth.uniform((2, 640, 4, 4), 301) - 0.5, "z": synth.uniform((2, 192, 1, 1), 302) - 0.5}
    losses = []
    for _ in range(10):
        c = first_train_step(net, crit, x, opt, [0, 10], clip_max_norm=1.0, noise=noise)
        losses.append(float(c["loss"]))
        assert np.isfinite(losses[-1])
    print("first_train losses", losses)
    assert losses[-1] < losses[0] - 1e-3 * abs(losses[0])


def test_first_train_full_size_step_is_reproducible():
    """BASELINE configs[3] per-GPU size (32 x 3 x 256 x 256): one step runs, every gradient is finite, and replaying the
    SAME step (same input, same noise; forward and backward hipGraphs) reproduces loss and gradients bit for bit — the
    backward has no float atomics (fixed-order weight-gradient, bias-table and column-sum reductions)."""
    from vampic.finetune import ScalableRateDistortionLoss
    net, sd = _model()
    B = 32
    x = synth.synth_image(B, 256, 256, seed=11).cuda()
    noise = {"y": synth.uniform((B, 640, 16, 16), 401) - 0.5, "z": synth.uniform((B, 192, 4, 4), 402) - 0.5}
    crit_fn = ScalableRateDistortionLoss(lmbda_list=[0.0055, 0.04], device="cuda")
    runs = []
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        out = net(x, quality=[0, 10], training=True, noise=noise)
        c = crit_fn(out, x)
        c["loss"].backward()
        flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        assert torch.isfinite(flat).all() and float(flat.abs().max()) < 1e6
        runs.append((float(c["loss"]), flat.clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert tuple(out["x_hat"].shape) == (2, B, 3, 256, 256) and tuple(out["likelihoods"]["y_prog"].shape) == (1, B, 640, 16, 16)


@pytest.mark.parametrize("name", ["single_encoder", "single_decoder", "single_hyperprior", "all_single", "no_delta_no_mu_rep", "not_all_scalable"])
def test_first_train_step_variants_match_reference(name):
    """The first-stage training plan with a single encoder / decoder / hyperprior (models/__init__.py:11-55;
    pic.py:285-288,306-311,372,462-466): one g_a with 640 outputs (its last attention block runs 80-dimensional heads), one
    synthesis pair with 640 outputs, ONE g_s reconstructing both levels (two taped passes, gradients summed).  Loss terms,
    likelihoods, reconstructions and every gradient against the oracle's step at equal rounding decisions; the oracle's step is
    the reference's (tests/golden/first_train_variants.npz, CPU suite)."""
    from vampic.finetune import ScalableRateDistortionLoss
    from test_oracle_golden import train_fixture_inputs
    from config_variants import variant_args, oracle_kwargs
    a = variant_args(name)
    net = vampic.get_model(a, "cpu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0)
    net.load_state_dict(sd)
    net = net.cuda().train()
    x, ny, nz = train_fixture_inputs()
    kw = {k: v for k, v in oracle_kwargs(a).items() if k in ("multiple_encoder", "multiple_decoder", "multiple_hyperprior", "delta_encode",
                                                         "total_mu_rep", "all_scalable")}
    for use_graph in (False, True):
        net.use_graph = use_graph
        net.zero_grad(set_to_none=True)
        out = net(x.cuda(), quality=[0, 10], training=True, noise={"y": ny, "z": nz})
        crit = ScalableRateDistortionLoss(lmbda_list=[0.0055, 0.04], device="cuda")(out, x.cuda())
        crit["loss"].backward()
        ref, _ = _forced_oracle_step(sd, net, x, ny, nz, [0, 10], [0.0055, 0.04], False, **kw)
        for k in ("loss", "bpp_loss", "bpp_base", "bpp_scalable", "bpp_hype"):
            a_, b_ = float(crit[k].detach().mean()), float(ref["crit"][k].mean())
            assert abs(a_ - b_) <= 1e-5 * max(1.0, abs(b_)), (k, a_, b_)
        assert _rel(out["x_hat"], ref["out"]["x_hat"]) <= 1e-4
        assert _rel(out["likelihoods"]["y"], ref["out"]["likelihoods"]["y"]) <= 1e-4
        joint, fam, worst = _compare_grads(net, ref["grads"])
        print(name, f"graph={use_graph}", "joint gradient error", joint, "worst", worst)
        assert joint <= 2e-4, (joint, fam)
        for f, v in fam.items():
            assert v <= 5e-4, (f, v, fam)
