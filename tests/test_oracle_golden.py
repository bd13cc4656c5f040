"""Pin the CPU oracle to the reference: the vectors under tests/golden were produced by running the
reference's own code (oracle/gen_golden.py); the oracle must reproduce them.  Rank/index work
is compared bit-exactly, float tensors with 1e-5 relative slack so that a different host CPU
(another MKLDNN code path) cannot fail the suite."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

import vampic.synth as synth
import vampic_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
Q_LEVS = [0, 0.01, 0.05, 0.1, 0.25, 0.5, 0.6, 0.75, 1, 1.5, 2, 2.5, 3, 5, 7.7, 9.99, 10, 12]


def _close(a, b, rtol=1e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape
    assert (a - b).abs().max().item() <= rtol * max(1.0, b.abs().max().item())


def _g(shape, seed, sc=1.0):
    return synth.normal(shape, seed, sc)


def test_variance_mask_matches_reference_bits():
    gold = np.load(os.path.join(GOLD, "variance_mask.npz"))
    for name, (B, C, h, w), seed in (("s8192", (3, 32, 16, 16), 3), ("s49152", (2, 32, 32, 48), 4), ("s480", (2, 32, 5, 3), 5)):
        s = synth.synth_sigma(B, C * h * w, seed=seed).reshape(B, C, h, w).numpy()
        for q in Q_LEVS:
            m = O.variance_mask_np(s, q).astype(np.uint8)
            assert np.array_equal(np.packbits(m.reshape(-1)), gold[f"{name}_q{q}"]), (name, q)
            if 0 < q < 10:
                thr = np.array([O.quantile_threshold_np(s[b], q * 0.1) for b in range(B)], dtype=np.float32)
                assert np.array_equal(thr, gold[f"{name}_q{q}_thr"]), (name, q)      # threshold bit-exact
    blocks = [synth.synth_sigma(1, 32 * 64, seed=20 + i).reshape(1, 32, 8, 8).numpy() for i in range(10)]
    for q in (0, 0.5, 2.5, 10):
        m = O.prog_mask_np(blocks, q).astype(np.uint8)
        assert np.array_equal(np.packbits(m.reshape(-1)), gold[f"prog_q{q}"])


def test_quantile_restatement_equals_torch_quantile():
    rng = np.random.default_rng(0)
    for n in (8192, 333, 2):
        for t in range(6):
            seg = np.exp(rng.uniform(np.log(0.05), np.log(300), n)).astype(np.float32)
            if t % 2:
                seg[rng.integers(0, n, max(1, n // 10))] = seg[0]
            for pr in (0.01, 0.5, 2.5, 5, 9.99):
                ref = np.float32(torch.quantile(torch.from_numpy(seg), 1.0 - pr * 0.1).item())
                assert O.quantile_threshold_np(seg, pr * 0.1) == ref
    assert np.isnan(O.quantile_threshold_np(np.array([1.0, np.nan, 2.0], dtype=np.float32), 0.5))


def test_entropy_ops_match_reference():
    gold = np.load(os.path.join(GOLD, "entropy_ops.npz"))
    y, mu = _g((2, 32, 16, 16), 30, 6.0), _g((2, 32, 16, 16), 31, 4.0)
    sg = synth.synth_sigma(2, 32 * 256, seed=32).reshape(2, 32, 16, 16)
    assert np.array_equal((torch.round(y - mu) + mu).numpy(), gold["gc_out"])
    _close(O.gaussian_likelihood(y, sg, mu), gold["gc_lik"], 1e-6)
    assert np.array_equal(torch.round(y).numpy(), gold["gc_out_nomean"])
    _close(O.gaussian_likelihood(y, sg, None), gold["gc_lik_nomean"], 1e-6)
    assert np.array_equal(O.build_indexes(sg).numpy().astype(np.int8), gold["gc_idx"])
    from vampic.entropy_models import EntropyBottleneck
    eb = EntropyBottleneck(192)
    sd = {"e." + k: v for k, v in synth.synth_state_dict(eb.state_dict(), 40).items()}
    zh, zl = O.eb_forward(sd, _g((2, 192, 4, 6), 41, 5.0), "e.")
    assert np.array_equal(zh.numpy(), gold["eb_zhat"])
    _close(zl, gold["eb_lik"], 1e-6)


def test_layer_ops_match_reference():
    from vampic import layers as Ly
    gold = np.load(os.path.join(GOLD, "layer_ops.npz"))
    for inv in (False, True):
        sd = {"g." + k: v for k, v in synth.synth_state_dict(Ly.GDN(192, inverse=inv).state_dict(), 7).items()}
        _close(O.gdn(sd, "g.", _g((2, 192, 8, 8), 8, 2.0), inv), gold[f"gdn_inv{int(inv)}"])
    for dim, ws, hw in ((192, 8, (16, 24)), (320, 4, (8, 8))):
        m = Ly.Win_noShift_Attention(dim=dim, num_heads=8, window_size=ws, shift_size=ws // 2)
        sd = {"a." + k: v for k, v in synth.synth_state_dict(m.state_dict(), 9).items()}
        _close(O.attention_block(sd, "a.", _g((2, dim) + hw, 10), ws), gold[f"attn_{dim}"])
    m = Ly.LatentRateReduction(32, True, "middle")
    sd = {"r." + k: v for k, v in synth.synth_state_dict(m.state_dict(), 11).items()}
    att = (synth.uniform((2, 32, 8, 8), 15) > 0.5).float()
    _close(O.rem_block(sd, "r.", _g((2, 32, 8, 8), 12, 3.0), _g((2, 64, 8, 8), 13), _g((2, 64, 8, 8), 14),
                       torch.cat([att, att], 1)), gold["rem"])


def test_forward_single_quality_matches_reference(synth_model_cpu):
    _, sd = synth_model_cpu
    gold = np.load(os.path.join(GOLD, "forward_single_quality.npz"))
    scal = json.load(open(os.path.join(GOLD, "forward_single_quality.json")))
    for seed in (0, 1):
        x = synth.synth_image(1, 64, 64, seed=seed)
        for q in (0, 0.5, 2.5, 10):
            o = O.forward_single_quality(sd, x, q)
            tag = f"s{seed}_q{q}"
            _close(o["x_hat"], gold[tag + "_x_hat"])
            _close(o["y_hat"], gold[tag + "_y_hat"])
            _close(o["likelihoods"]["y"], gold[tag + "_lik_y"])
            _close(o["likelihoods"]["z"], gold[tag + "_lik_z"])
            assert abs(O.psnr(x, o["x_hat"]) - scal[tag]["psnr"]) <= 1e-4
            assert abs(O.bpp(o["likelihoods"], 64 * 64) - scal[tag]["bpp"]) <= 1e-6 * max(1.0, scal[tag]["bpp"])
    x = synth.synth_image(1, 64, 128, seed=0)
    ck = O.forward_single_quality(sd, x, 0.75, check_levels=[0.75])["y_hat"]
    _close(ck, gold["rem_ck"])
    o = O.forward_single_quality(sd, x, 2.5, check_levels=[0.75], checkpoint_ref=torch.from_numpy(gold["rem_ck"]))
    _close(o["x_hat"], gold["rem_x_hat"])
    _close(o["y_hat"], gold["rem_y_hat"])
    _close(o["likelihoods"]["y"], gold["rem_lik_y"])


def test_rem_training_step_matches_reference(synth_model_cpu):
    """Training-mode forward + RateLoss + backward of the REM fine-tune step (BASELINE configs[4]) against the
    reference's own autograd run (oracle/gen_golden.py §6): likelihoods, loss and every REM gradient."""
    _, sd = synth_model_cpu
    gold = np.load(os.path.join(GOLD, "rem_train_step.npz"))
    ck = torch.from_numpy(np.load(os.path.join(GOLD, "forward_single_quality.npz"))["rem_ck"])
    x = synth.synth_image(1, 64, 128, seed=0)
    ny = synth.uniform((1, 640, 4, 8), 101) - 0.5
    nz = synth.uniform((1, 192, 1, 2), 102) - 0.5
    o = O.rem_training_step(sd, x, 2.5, ck, ny, nz, check_levels=[0.75])
    _close(o["likelihoods"]["y"], gold["lik_y"])
    _close(o["likelihoods"]["z"], gold["lik_z"])
    assert abs(o["loss"] - gold["loss"][0]) <= 1e-6 * abs(gold["loss"][0])
    samples, off = gold["grad_samples"], 0
    assert len(gold["grad_names"]) == len(o["grads"]) == 420
    for name, norm in zip(gold["grad_names"], gold["grad_norms"]):
        g = o["grads"]["post_latent.0." + str(name)].reshape(-1)
        s = g[::53].numpy()
        ref = samples[off:off + len(s)]
        off += len(s)
        assert abs(float(g.double().norm()) - norm) <= 1e-5 * norm + 1e-12, name
        assert np.abs(s - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-10, name
    assert off == len(samples)


def test_rem_training_step_without_mu_std_matches_reference():
    """The ``mu_std=False`` REM variant (scale-only refinement, rem_pic.py:194-195,214-220): the oracle's fine-tune step
    against the reference's own autograd run (oracle/gen_golden.py section 12)."""
    import argparse
    import vampic
    from conftest import README_ARGS
    gold = np.load(os.path.join(GOLD, "rem_train_step_no_mu_std.npz"))
    net = vampic.get_model(argparse.Namespace(model="rem", check_levels=[0.75], mu_std=False, dimension="middle", **README_ARGS), "cpu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0)
    x = synth.synth_image(1, 64, 128, seed=0)
    ny = synth.uniform((1, 640, 4, 8), 101) - 0.5
    nz = synth.uniform((1, 192, 1, 2), 102) - 0.5
    o = O.rem_training_step(sd, x, 2.5, torch.from_numpy(gold["ck"]), ny, nz, check_levels=[0.75], mu_std=False)
    _close(o["likelihoods"]["y"], gold["lik_y"])
    _close(o["likelihoods"]["z"], gold["lik_z"])
    assert abs(o["loss"] - gold["loss"][0]) <= 1e-6 * abs(gold["loss"][0])
    samples, off = gold["grad_samples"], 0
    assert len(gold["grad_names"]) == len(o["grads"])
    for name, norm in zip(gold["grad_names"], gold["grad_norms"]):
        g = o["grads"]["post_latent.0." + str(name)].reshape(-1)
        s = g[::53].numpy()
        ref = samples[off:off + len(s)]
        off += len(s)
        assert abs(float(g.double().norm()) - norm) <= 1e-5 * norm + 1e-12, name
        assert np.abs(s - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-10, name
    assert off == len(samples)


def test_flip_audit_separates_boundary_events_from_errors(synth_model_cpu):
    """tests/parity_audit.py (the gate of the GPU end-to-end parity tests) on CPU: two oracle runs whose inputs differ
    by 2e-7 play "reference" and "other back-end" — every differing rounding decision must be classified as a
    boundary event; a shifted mean (a genuine error) must be reported as a violation."""
    import copy
    import vampic
    from parity_audit import audit
    _, sd = synth_model_cpu
    x = vampic.synth.synth_image(1, 64, 128, seed=3)
    ref = O.forward_single_quality(sd, x, 2.5)
    oth = O.forward_single_quality(sd, (x + 2e-7 * (vampic.synth.uniform(tuple(x.shape), 9) - 0.5)).float(), 2.5)
    rep = audit(oth["y"], oth, ref, 2.5)
    assert rep["violations"] == [], rep
    assert rep["sym_flips"] + rep["mask_flips"] == rep["explained"] + rep["downstream"]
    same = audit(ref["y"], ref, ref, 2.5)
    assert same["sym_flips"] == 0 and same["mask_flips"] == 0 and same["first"] is None and same["violations"] == []
    bad = copy.deepcopy(ref)
    bad["mu_base"] = ref["mu_base"].clone()
    bad["mu_base"][:, 32:64] += 0.3                      # wrong mean in base slice 1: symbols move, not at boundaries
    rep = audit(ref["y"], bad, ref, 2.5)
    assert rep["violations"] and rep["first"] == "base 1", rep
    near = copy.deepcopy(ref)                            # the kept element closest to the threshold drops out: a threshold event
    near["mask"] = ref["mask"].clone()
    s0 = ref["std"][0, :32]
    thr = float(O.quantile_threshold_np(s0.numpy().ravel(), 0.25))
    pos = torch.nonzero((s0 - thr).abs() == (s0 - thr).abs().min())[0]
    if float((s0 - thr).abs().min()) < 2e-4:
        near["mask"][0, pos[0], pos[1], pos[2]] = 1 - near["mask"][0, pos[0], pos[1], pos[2]]
        rep = audit(ref["y"], near, ref, 2.5)
        assert rep["violations"] == [] and rep["mask_flips"] == 1 and rep["explained"] >= 1, rep
    bad = copy.deepcopy(ref)
    bad["mask"] = ref["mask"].clone()
    bad["mask"][:, :32] = 1 - ref["mask"][:, :32]        # a mask that is simply wrong
    rep = audit(ref["y"], bad, ref, 2.5)
    assert rep["violations"] and "threshold" in rep["violations"][0], rep


def test_oracle_reproduces_reference_demo_image_256(synth_model_cpu):
    """BASELINE configs[0] workload (one 256x256 image, the demo's q_levs, test/parser.py:20): the oracle against the
    vectors the REFERENCE produced (oracle/gen_golden.py section 8) — mask bits and thresholds exactly."""
    import vampic
    _, sd = synth_model_cpu
    gold = np.load(os.path.join(GOLD, "demo_256.npz"))
    scal = json.load(open(os.path.join(GOLD, "demo_256.json")))
    x = vampic.synth.synth_image(1, 256, 256, seed=0)
    for q in (0, 0.01, 0.25, 0.6, 0.9, 2, 4.5, 10):                       # 8 of the 16 stored levels (CPU time)
        tag = f"q{q}"
        o = O.forward_single_quality(sd, x, q)
        _close(o["y_hat"][:, ::4, ::2, ::2], gold[tag + "_y_hat"], 2e-5)
        _close(o["x_hat"][:, :, ::8, ::8], gold[tag + "_x_hat"], 2e-5)
        if q > 0:
            assert np.array_equal(np.packbits(o["mask"].numpy().astype(np.uint8).reshape(-1)), gold[tag + "_mask"]), tag
        if 0 < q < 10:
            thr = np.array([O.quantile_threshold_np(s_.numpy().ravel(), q * 0.1) for s_ in o["std"][0].chunk(10, 0)],
                           dtype=np.float32)
            assert np.array_equal(thr, gold[tag + "_thr"]), tag
        assert abs(O.psnr(x, o["x_hat"]) - scal[tag]["psnr"]) <= 1e-4
        assert abs(O.bpp(o["likelihoods"], 65536) - scal[tag]["bpp"]) <= 1e-6 * max(1.0, scal[tag]["bpp"])


def test_oracle_reproduces_reference_config_variants():
    """Every constructor flag away from the README values (single encoder / decoder / hyperprior, support slices 0 / 2 / 8,
    delta_encode / total_mu_rep / all_scalable off, REM dimension "big", mu_std off): the oracle, fed the state_dict of
    THIS package's model for that variant (same keys as the reference's or the lookup fails), against the vectors the
    REFERENCE produced (oracle/gen_golden.py section 9)."""
    import vampic
    from config_variants import CONFIG_VARIANTS, variant_args, oracle_kwargs
    gold = np.load(os.path.join(GOLD, "config_variants.npz"))
    scal = json.load(open(os.path.join(GOLD, "config_variants.json")))
    x = synth.synth_image(1, 64, 64, seed=2)
    assert {k.rsplit("_q", 1)[0] for k in scal} == set(CONFIG_VARIANTS)
    for name in CONFIG_VARIANTS:
        a = variant_args(name)
        net = vampic.get_model(a, "cpu")
        sd = synth.synth_state_dict(net.state_dict(), seed=0)
        kw = oracle_kwargs(a)
        ck = None
        if a.model == "rem":
            ck = O.forward_single_quality(sd, x, a.check_levels[0], **kw)["y_hat"]
            _close(ck, gold[f"{name}_ck"], 2e-5)
        for q in (0, 2.5):
            o = O.forward_single_quality(sd, x, q, checkpoint_ref=(ck if q > 0 else None), **kw)
            tag = f"{name}_q{q}"
            _close(o["y_hat"], gold[tag + "_y_hat"], 2e-5)
            _close(o["x_hat"][:, :, ::2, ::2], gold[tag + "_x_hat"], 2e-5)
            assert abs(O.psnr(x, o["x_hat"]) - scal[tag]["psnr"]) <= 1e-4, tag
            assert abs(O.bpp(o["likelihoods"], 4096) - scal[tag]["bpp"]) <= 1e-6 * max(1.0, scal[tag]["bpp"]), tag


def test_oracle_aux_loss_matches_reference():
    """EntropyBottleneck.loss and its gradient w.r.t. the quantiles (entropy_models.py:398-401) against the reference's
    own autograd (tests/golden/entropy_ops.npz: eb_aux_loss, eb_aux_dq)."""
    import vampic
    from vampic.entropy_models import EntropyBottleneck
    gold = np.load(os.path.join(GOLD, "entropy_ops.npz"))
    eb = EntropyBottleneck(192)
    sd = {"entropy_bottleneck." + k: v.clone() for k, v in synth.synth_state_dict(eb.state_dict(), 40).items()}
    sd["entropy_bottleneck.quantiles"].requires_grad_(True)
    loss = O.eb_aux_loss(sd)
    loss.backward()
    assert abs(loss.item() - gold["eb_aux_loss"][0]) <= 1e-5 * abs(gold["eb_aux_loss"][0])
    _close(sd["entropy_bottleneck.quantiles"].grad, gold["eb_aux_dq"], 1e-5)


def test_oracle_refine_gs_step_matches_reference():
    """`--training_type refine_gs` (train.py:150-157,216-218): loss and every g_s[1] gradient of the oracle's step against
    the reference's own autograd run (tests/golden/refine_gs_step.npz; incl. the NonNegativeParametrizer / LowerBound rule
    of the IGDN parameters)."""
    import argparse
    import vampic
    from conftest import README_ARGS
    net = vampic.get_model(argparse.Namespace(model="pic", **README_ARGS), "cpu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0)
    gold = np.load(os.path.join(GOLD, "refine_gs_step.npz"))
    x = synth.synth_image(1, 64, 64, seed=3)
    loss, mse, x_hat, grads = O.refine_gs_training_step(sd, x, 2.5)
    assert abs(float(loss) - gold["loss"][0]) <= 1e-6 * gold["loss"][0] and abs(float(mse) - gold["loss"][1]) <= 1e-6 * gold["loss"][1]
    _close(x_hat[:, :, ::2, ::2], gold["x_hat"], 2e-5)
    off = 0
    assert sorted(grads) == sorted(str(n) for n in gold["grad_names"])
    for name, norm in zip(gold["grad_names"], gold["grad_norms"]):
        g = grads[str(name)].reshape(-1)
        s_ = g[::97].numpy()
        ref = gold["grad_samples"][off:off + len(s_)]
        off += len(s_)
        assert abs(float(g.double().norm()) - norm) <= 1e-5 * norm + 1e-12, name
        assert np.abs(s_ - ref).max() <= 1e-5 * norm + 1e-12, name


def test_oracle_refine_gs_lrp_step_matches_reference():
    """`--training_type refine_gs --lrp` (unfreeze_decoder(lrp=True), pic.py:171-184): the oracle's step — y_hat recomputed
    under autograd through the ten progressive LRP stacks — against the reference's own run
    (tests/golden/refine_gs_lrp_step.npz: 100 g_s[1] + 100 LRP gradients)."""
    import argparse
    import vampic
    from conftest import README_ARGS
    net = vampic.get_model(argparse.Namespace(model="pic", **README_ARGS), "cpu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0)
    gold = np.load(os.path.join(GOLD, "refine_gs_lrp_step.npz"))
    x = synth.synth_image(1, 64, 64, seed=3)
    loss, mse, x_hat, grads = O.refine_gs_training_step(sd, x, 2.5, lrp=True)
    assert abs(float(loss) - gold["loss"][0]) <= 1e-6 * gold["loss"][0] and abs(float(mse) - gold["loss"][1]) <= 1e-6 * gold["loss"][1]
    _close(x_hat[:, :, ::4, ::4], gold["x_hat"], 2e-5)
    names = [str(n) for n in gold["grad_names"]]
    key = lambda n: n[len("g_s.1."):] if n.startswith("g_s.1.") else n          # the oracle strips its decoder prefix
    assert sorted(grads) == sorted(key(n) for n in names)
    assert sum(n.startswith("lrp_transforms_prog.") for n in names) == 100
    off = 0
    for name, norm in zip(names, gold["grad_norms"]):
        g = grads[key(name)].reshape(-1)
        s_ = g[::389].numpy()
        ref = gold["grad_samples"][off:off + len(s_)]
        off += len(s_)
        assert abs(float(g.double().norm()) - norm) <= 1e-5 * norm + 1e-12, name
        assert np.abs(s_ - ref).max() <= 1e-5 * norm + 1e-12, name


TRAIN_FIXTURES = (("first_train_step", [0, 10], [0.0055, 0.04], False), ("train_single_step", [2.5], 0.01, True))


def train_fixture_inputs():
    """Inputs of tests/golden/{first_train_step,train_single_step}.npz (oracle/gen_golden.py section 10)."""
    return (synth.synth_image(2, 64, 64, seed=5), synth.uniform((2, 640, 4, 4), 201) - 0.5,
            synth.uniform((2, 192, 1, 1), 202) - 0.5)


@pytest.mark.parametrize("fixture,qualities,lmbda,single", TRAIN_FIXTURES)
def test_oracle_first_train_step_matches_reference(fixture, qualities, lmbda, single):
    """BASELINE configs[3] (`--training_type first_train`): the oracle's training forward (pic.py:301-491 / :497-666 with
    training=True), ScalableRateDistortionLoss (training/loss.py:6-66) and autograd over EVERY parameter against the
    reference's own step — loss terms, likelihoods, reconstructions, and all 1065 (965 for the single-quality pass, which
    leaves g_s[0] unused) gradient tensors (norm + every 997th element)."""
    import argparse
    import warnings
    import vampic
    from conftest import README_ARGS
    net = vampic.get_model(argparse.Namespace(model="pic", **README_ARGS), "cpu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0)
    gold = np.load(os.path.join(GOLD, fixture + ".npz"))
    x, ny, nz = train_fixture_inputs()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = O.first_train_step(sd, x, qualities, ny, nz, lmbda, single=single)
    got = [float(r["crit"][k].mean()) for k in ("loss", "bpp_loss", "bpp_base", "bpp_scalable", "bpp_hype")]
    for a, b in zip(got, gold["loss"]):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b))
    _close(r["crit"]["mse_loss"], gold["mse"], 1e-6)
    _close(r["out"]["likelihoods"]["y"], gold["lik_y"], 1e-5)
    _close(r["out"]["likelihoods"]["z"], gold["lik_z"], 1e-5)
    if single:
        _close(r["out"]["x_hat"][:, :, ::4, ::4], gold["x_hat"], 2e-5)
        _close(r["out"]["y_hat"], gold["y_hat"], 2e-5)
    else:
        _close(r["out"]["x_hat"][:, :, :, ::4, ::4], gold["x_hat"], 2e-5)
        _close(r["out"]["likelihoods"]["y_prog"], gold["lik_y_prog"], 1e-5)
        _close(r["out"]["y_base"], gold["y_hat_base"], 2e-5)
        _close(r["out"]["y_prog"], gold["y_hat_prog"], 2e-5)
    names = [str(n) for n in gold["grad_names"]]
    assert sorted(k for k, g in r["grads"].items() if g is not None) == sorted(names)
    assert len(names) == (965 if single else 1065)
    off, num, den = 0, 0.0, 0.0
    for name, norm in zip(names, gold["grad_norms"]):
        g = r["grads"][name].reshape(-1)
        s_ = g[::997].numpy()
        ref = gold["grad_samples"][off:off + len(s_)]
        off += len(s_)
        assert abs(float(g.double().norm()) - norm) <= 2e-5 * norm + 1e-9, name
        num += float(((s_ - ref).astype(np.float64) ** 2).sum())
        den += float((ref.astype(np.float64) ** 2).sum())
    assert (num / den) ** 0.5 <= 1e-5


def test_oracle_reproduces_reference_trained_like_profile():
    """Round 4: the ``trained-like`` profile of the synthetic generator (rate 0.9 ... 2.5 bpp, |y_hat| <= 2 — a trained
    codec's operating range, where the north star's absolute tolerances apply literally): the oracle against the vectors
    the REFERENCE produced (oracle/gen_golden.py section 11) — masks and thresholds exactly, PSNR 1e-4 dB, bpp 1e-6."""
    import argparse
    import vampic
    from conftest import README_ARGS
    gold = np.load(os.path.join(GOLD, "trained_like.npz"))
    scal = json.load(open(os.path.join(GOLD, "trained_like.json")))
    net = vampic.get_model(argparse.Namespace(model="pic", **README_ARGS), "cpu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0, profile="trained-like")
    assert all(0.3 <= v["bpp"] <= 3.0 and v["abs_y_hat_max"] < 4 for v in scal.values())
    for name, x, qs in (("a", synth.synth_image(1, 64, 64, seed=0), (0, 0.5, 2.5, 10)),
                        ("b", synth.synth_image(1, 64, 64, seed=1), (0, 0.5, 2.5, 10)),
                        ("demo", synth.synth_image(1, 256, 256, seed=0), (2.5,))):
        npix = x.shape[2] * x.shape[3]
        for q in qs:
            tag = f"{name}_q{q}"
            o = O.forward_single_quality(sd, x, q)
            small = name != "demo"
            _close(o["y_hat"] if small else o["y_hat"][:, ::4, ::2, ::2], gold[tag + "_y_hat"], 2e-5)
            _close(o["x_hat"][:, :, ::2, ::2] if small else o["x_hat"][:, :, ::8, ::8], gold[tag + "_x_hat"], 2e-5)
            if q > 0:
                assert np.array_equal(np.packbits(o["mask"].numpy().astype(np.uint8).reshape(-1)), gold[tag + "_mask"]), tag
            if 0 < q < 10:
                thr = np.array([O.quantile_threshold_np(s_.numpy().ravel(), q * 0.1) for s_ in o["std"][0].chunk(10, 0)],
                               dtype=np.float32)
                assert np.array_equal(thr, gold[tag + "_thr"]), tag
            assert abs(O.psnr(x, o["x_hat"]) - scal[tag]["psnr"]) <= 1e-4, tag
            assert abs(O.bpp(o["likelihoods"], npix) - scal[tag]["bpp"]) <= 1e-6, tag


@pytest.mark.parametrize("name", ["single_encoder", "single_decoder", "single_hyperprior", "all_single", "no_delta_no_mu_rep", "not_all_scalable"])
def test_oracle_first_train_step_variants_match_reference(name):
    """First-stage training step with a single encoder / decoder / hyperprior (pic.py:285-288,306-311,372,462-466): the
    oracle's forward([0, 10], training=True) + ScalableRateDistortionLoss + autograd against the REFERENCE's own run
    (oracle/gen_golden.py section 13): losses, likelihoods, reconstructions and every sampled gradient."""
    import vampic
    from config_variants import variant_args, oracle_kwargs
    gold = np.load(os.path.join(GOLD, "first_train_variants.npz"))
    a = variant_args(name)
    net = vampic.get_model(a, "cpu")
    sd = synth.synth_state_dict(net.state_dict(), seed=0)
    x, ny, nz = train_fixture_inputs()
    kw = {k: v for k, v in oracle_kwargs(a).items() if k in ("multiple_encoder", "multiple_decoder", "multiple_hyperprior", "delta_encode",
                                                         "total_mu_rep", "all_scalable")}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = O.first_train_step(sd, x, [0, 10], ny, nz, [0.0055, 0.04], **kw)
    for i, k in enumerate(("loss", "bpp_loss", "bpp_base", "bpp_scalable", "bpp_hype")):
        assert abs(float(r["crit"][k].mean()) - gold[name + "_loss"][i]) <= 1e-5 * max(1.0, abs(gold[name + "_loss"][i])), k
    _close(r["out"]["x_hat"][:, :, :, ::4, ::4], gold[name + "_x_hat"], 2e-5)
    _close(r["out"]["likelihoods"]["y"], gold[name + "_lik_y"], 2e-5)
    _close(r["out"]["likelihoods"]["z"], gold[name + "_lik_z"], 2e-5)
    names = [str(n) for n in gold[name + "_grad_names"]]
    assert sorted(names) == sorted(k for k, g in r["grads"].items() if g is not None)
    samples, off = gold[name + "_grad_samples"], 0
    num = den = 0.0
    for n_, norm in zip(names, gold[name + "_grad_norms"]):
        g = r["grads"][n_].reshape(-1)
        sref = samples[off:off + len(g[::997])]
        off += len(sref)
        assert abs(float(g.double().norm()) - norm) <= 1e-4 * norm + 1e-9, n_
        num += float(((g[::997].numpy() - sref).astype(np.float64) ** 2).sum())
        den += float((sref.astype(np.float64) ** 2).sum())
    assert off == len(samples) and (num / den) ** 0.5 <= 1e-5
