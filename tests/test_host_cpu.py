"""CPU-side checks: the state_dict surface equals the reference's, the C-ABI library loads and
exports every symbol include/vampic.h declares, the synthetic generator is deterministic, and
the product path fails loudly without a GPU (no fallback)."""
import ctypes
import json
import os
import re

import pytest
import torch

import vampic
from vampic import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_state_dict_surface_equals_reference(synth_model_cpu):
    net, _ = synth_model_cpu
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))
    sd = net.state_dict()
    assert list(sd) == list(man["entries"]), "key names/order differ from the reference"
    for k, (shape, dtype) in man["entries"].items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == dtype, k
    assert sum(p.numel() for p in net.parameters()) == man["n_params"] == 153795526


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vampic.h")).read()
    declared = set(re.findall(r"\b(vam_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vam_seg", "vam_aux", "vam_conv"}
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libvampic.so lacks {name}"
    assert declared == set(L.EXPORTED_SYMBOLS), declared ^ set(L.EXPORTED_SYMBOLS)
    assert L.load().vam_version() >= 100
    assert ctypes.sizeof(L.VamConv) == L.load().vam_conv_struct_size()     # ABI layout guard


def test_synth_is_deterministic_and_key_addressed():
    like = torch.empty(8, 4, 3, 3)
    a = vampic.synth.synth_tensor("cc_mean_transforms.0.0.weight", like, 0)
    b = vampic.synth.synth_tensor("cc_mean_transforms.0.0.weight", like, 0)
    c = vampic.synth.synth_tensor("cc_mean_transforms.1.0.weight", like, 0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    s = vampic.synth.synth_sigma(2, 1000, 3)
    assert (s < 0).any() and s.abs().min() >= 2.0 ** -5


def test_no_cpu_fallback(synth_model_cpu):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net, _ = synth_model_cpu
    x = vampic.synth.synth_image(1, 64, 64, 0)
    with pytest.raises(L.VamError):
        net.forward_single_quality(x, 2.5)
    with pytest.raises((L.VamError, AssertionError)):
        net.masking(torch.rand(1, 32, 4, 4), pr=2.5)


def test_unbuilt_rows_raise(synth_model_cpu):
    net, _ = synth_model_cpu
    import copy
    m = copy.deepcopy(net).train()
    with pytest.raises(NotImplementedError):          # gradients outside the REMs need backward kernels that are not built
        m.forward_single_quality(torch.rand(1, 3, 64, 64), 2.5, training=True)
    if not torch.cuda.is_available():
        with pytest.raises(L.VamError):               # the aux loss is a HIP kernel too: no CPU fallback
            net.entropy_bottleneck.loss()
    if not torch.cuda.is_available():
        with pytest.raises(L.VamError):               # bitstream path exists but never falls back to the CPU
            net.compress(torch.rand(1, 3, 64, 64), 2.5)


def test_rem_bookkeeping(synth_model_cpu):
    net, _ = synth_model_cpu
    assert net.find_check_quality(0.5) == (0, 0, -1)
    assert net.find_check_quality(2.5) == (0.75, 10, -1)
    assert net.num_rems == 1 and net.enable_rem == [True] and net._rem_index(2.5) == 0
    import vampic_oracle as O
    for cl in ([0.75], [0.25, 1.75], [0.01, 0.25, 1.75]):
        for q in (0.005, 0.2, 1.0, 5.0):
            if q > cl[0]:
                net2 = type("T", (), {"check_levels": cl, "num_rems": len(cl)})()
                assert vampic.VarianceMaskingPICREM._rem_index(net2, q) == O.rem_index(cl, q)


def test_checkpoint_helpers_match_reference():
    """Key remapping / padding / PSNR / paths against what the reference's utility/functions.py returned
    (tests/golden/host_utils.json, oracle/gen_golden.py §7), plus a save/load round trip."""
    import argparse
    import json
    import math
    import tempfile
    import torch
    import vampic.synth as synth
    from vampic import checkpoint as ck
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "host_utils.json")))
    single = ["g_a.0.weight", "g_a.1.beta", "g_a.1.gamma", "g_s.0.weight", "g_s.8.bias", "h_a.0.weight", "h_a.8.bias",
              "h_mean_s.0.weight", "h_mean_s.8.bias", "h_scale_s.0.weight", "h_scale_s.6.0.bias",
              "cc_mean_transforms.0.0.weight", "lrp_transforms.3.8.bias", "gaussian_conditional.scale_table",
              "entropy_bottleneck._matrix0", "entropy_bottleneck.quantiles", "something_else.weight"]
    base = {k: torch.full((1,), float(i)) for i, k in enumerate(single)}
    enh = {k: torch.full((1,), 100.0 + i) for i, k in enumerate(["g_s.0.weight", "g_s.8.bias", "g_a.0.weight"])}
    for tag, want in gold["init"].items():
        md, me, mh, we = (c == "1" for c in tag)
        a = argparse.Namespace(multiple_decoder=md, multiple_encoder=me, multiple_hyperprior=mh)
        got = ck.initialize_model_from_pretrained(base, a, enh if we else None)
        assert [[k, float(v)] for k, v in got.items()] == want, tag          # same keys, same order, same tensors
    old = {k: torch.full((1,), float(i)) for i, k in enumerate(["g_a.0.weight", "g_a_enh.0.weight", "g_a.1.beta", "g_s.0.weight"])}
    new = {k: torch.full((1,), float(i)) for i, k in enumerate(["g_a.0.0.weight", "g_a.0.1.beta", "g_a.1.0.weight"])}
    for me in (False, True):
        assert [[k, float(v)] for k, v in ck.replace_keys(old, me).items()] == gold["replace"][f"old{int(me)}"]
        assert [[k, float(v)] for k, v in ck.replace_keys(new, me).items()] == gold["replace"][f"new{int(me)}"]
    from vampic.evaluate import compute_padding
    for key, want in gold["padding"].items():
        hh, ww = (int(v) for v in key.split("x"))
        assert [list(t) for t in compute_padding(hh, ww, min_div=64)] == want
    for seed, want in gold["psnr"].items():
        a_, b_ = synth.uniform((2, 3, 16, 24), int(seed)), synth.uniform((2, 3, 16, 24), int(seed) + 10)
        assert abs(-10 * math.log10(torch.mean((a_ - b_) ** 2).item()) - want) < 1e-9   # host formula; GPU path in test_gpu_bitstream
    assert list(ck.create_savepath("/x/y")) == gold["savepath"]
    m = ck.AverageMeter()
    m.update(2.0)
    m.update(4.0, n=3)
    assert m.avg == 3.5 and m.count == 4
    with tempfile.TemporaryDirectory() as d:
        last, best = ck.create_savepath(d)
        lin = torch.nn.Linear(3, 2)
        ck.save_checkpoint({"epoch": 3, "state_dict": lin.state_dict()}, False, last, best)
        ck.save_checkpoint({"epoch": 4, "state_dict": lin.state_dict()}, True, last, best)
        lin2 = torch.nn.Linear(3, 2)
        assert ck.load_checkpoint(best, lin2)["epoch"] == 4 and ck.load_checkpoint(last)["epoch"] == 3
        assert all(torch.equal(a_, b_) for a_, b_ in zip(lin.state_dict().values(), lin2.state_dict().values()))
        # the reference stores its argparse.Namespace beside the tensors (train.py:371-383): the safe loader takes it;
        # anything else needs the explicit opt-in
        ck.save_checkpoint({"epoch": 5, "state_dict": lin.state_dict(), "args": argparse.Namespace(N=192, M=640)}, False, last, best)
        assert ck.load_checkpoint(last)["args"].M == 640
        ck.save_checkpoint({"epoch": 6, "state_dict": lin.state_dict(), "opt": torch.optim.SGD}, False, last, best)
        try:
            ck.load_checkpoint(last)
            raise AssertionError("an arbitrary pickled class must not load without allow_pickle")
        except RuntimeError as e:
            assert "allow_pickle" in str(e)
        assert ck.load_checkpoint(last, allow_pickle=True)["epoch"] == 6
    args = argparse.Namespace(learning_rate=1e-4, aux_learning_rate=1e-3, training_type="rems")
    net = torch.nn.Module()
    net.a = torch.nn.Linear(2, 2)
    net.eb = torch.nn.Module()
    net.eb.quantiles = torch.nn.Parameter(torch.zeros(1))
    opt, aux = ck.configure_optimizers(net, args)
    assert aux is None and len(opt.param_groups[0]["params"]) == 2          # eb.quantiles is not in the main optimiser
    args.training_type = "first_strain"
    opt, aux = ck.configure_optimizers(net, args)
    assert aux is not None and len(aux.param_groups[0]["params"]) == 1


def test_image_io_round_trip(tmp_path):
    import torch
    import vampic.synth as synth
    from vampic.evaluate import read_image, write_image
    x = (synth.uniform((3, 20, 31), 3) * 255).round() / 255
    p = tmp_path / "a.png"
    write_image(x, p)
    y = read_image(p)
    assert y.dtype == torch.float32 and y.shape == (3, 20, 31) and torch.equal(y, x)


def test_finetune_driver_host_logic():
    """Host side of the two training schedules (vampic.finetune): the quality lists of train.py:150-155 / :167-181, the
    check-level lookup of training/step.py:14-32, and the two criteria (training/loss.py:126-187, 189-229) on a synthetic
    output dict — formulas restated inline."""
    import math
    import numpy as np
    import torch
    from vampic import finetune as FT
    q = FT.refine_gs_quality_list()
    exp = list(np.arange(0.015, 1.5, (1.5 - 0.025) / 200)) + [1.5] + list(np.arange(1.6, 10, (10 - 1.6) / 50)) + [10]
    assert q == [float(v) for v in exp] and q[0] == 0.015 and q[-1] == 10.0
    assert FT.extract_quality_ref(0.5, [0.75]) is None and FT.extract_quality_ref(2.5, [0.75]) == 0.75
    assert FT.extract_quality_ref(1.0, [0.75, 2.0]) == 0.75 and FT.extract_quality_ref(5.0, [0.75, 2.0]) == 2.0
    g = torch.Generator().manual_seed(0)
    x = torch.rand((2, 3, 8, 8), generator=g)
    out = {"x_hat": torch.rand((2, 3, 8, 8), generator=g).requires_grad_(True),
           "likelihoods": {"y": torch.rand((2, 4, 2, 2), generator=g) * 0.9 + 0.05, "z": torch.rand((2, 3, 1, 1), generator=g) * 0.9 + 0.05}}
    den = -math.log(2) * 2 * 8 * 8
    d = FT.DistortionLoss(device="cpu")(out, x, lmbda=1e-2)
    mse = torch.nn.functional.mse_loss(x, out["x_hat"])
    assert torch.allclose(d["loss"], 255.0 ** 2 * 1e-2 * mse) and float(d["bpp_scalable"]) == 0.0
    assert torch.allclose(d["bpp_loss"], torch.log(out["likelihoods"]["y"]).sum() / den + 2 * torch.log(out["likelihoods"]["z"]).sum() / den)
    d["loss"].backward()
    assert out["x_hat"].grad is not None                       # the distortion is the trained term
    r = FT.RateLoss(device="cpu")(out, x)
    assert torch.allclose(r["loss"], torch.log(out["likelihoods"]["y"]).sum() / den + 2 * torch.log(out["likelihoods"]["z"]).sum() / den)


def test_memset_zero_is_a_kernel_with_word_granularity():
    """vam_memset_zero has no hipMemsetAsync fallback any more (every node of a captured plan is a kernel node): odd
    pointers / sizes are rejected before anything is launched."""
    import ctypes as C
    from vampic import _lib as L
    lib = L.load()
    assert lib.vam_memset_zero(C.c_void_p(0x1002), 8, None) == -1
    assert lib.vam_memset_zero(C.c_void_p(0x1000), 6, None) == -1
    assert b"multiples of 4" in lib.vam_last_error()


def test_weight_gradient_planning_is_host_arithmetic():
    """``vam_conv_wgrad_plan`` (pixel splits + workspace of a weight-gradient problem) is pure host arithmetic on the
    problem's extents: callable without a GPU, deterministic, and consistent with what the launch will need —
    workspace = splits x (N C taps + N) floats, no split for problems that already fill the chip, never more splits
    than 256-pixel ranges.  (The tile the LDS-tiled kernel picks — csrc/wgrad_lds.hip:wgrad2_tile, e.g. 96x64 where N = 96 —
    enters through the number of weight tiles and the makespan model of wgrad2_splits.)"""
    import ctypes
    lib = L.load()

    def plan(B, H, W, C_, N, k):
        p = L.VamWgrad()
        p.B, p.H, p.W, p.C, p.N, p.kh, p.kw = B, H, W, C_, N, k, k
        nbytes = ctypes.c_size_t(0)
        s = lib.vam_conv_wgrad_plan(ctypes.byref(p), ctypes.byref(nbytes))
        return s, nbytes.value

    for shape in [(32, 64, 64, 192, 192, 5), (32, 64, 64, 96, 96, 3), (32, 16, 16, 320, 224, 3), (32, 128, 128, 192, 192, 1),
                  (32, 16, 16, 64, 32, 3), (1, 16, 16, 64, 32, 3)]:
        B, H, W, C_, N, k = shape
        s, nb = plan(*shape)
        assert (s, nb) == plan(*shape)
        assert 1 <= s <= 256 and s <= max(1, B * H * W // 256), shape       # >= 256 pixels per split (LDS-tiled kernel; 2048: wgrad_kernel)
        assert nb == (0 if s == 1 else s * (N * C_ * k * k + N) * 4), shape
    assert plan(1, 16, 16, 64, 32, 3)[0] == 1                      # 256 pixels: nothing to split
    assert plan(32, 128, 128, 192, 192, 1)[0] > 8                  # a 1x1 layer on 524288 pixels has 2 x 3 weight tiles
    assert ctypes.sizeof(L.VamPackJob) == 48                       # include/vampic.h: two pointers + eight int32
