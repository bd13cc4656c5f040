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
    with pytest.raises(NotImplementedError):          # training needs the backward kernels (not built)
        net.forward_single_quality(torch.rand(1, 3, 64, 64), 2.5, training=True)
    with pytest.raises(NotImplementedError):
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
