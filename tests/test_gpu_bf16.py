"""bf16-storage configuration (BASELINE configs[2] "bf16"; ``model.storage = "bf16"``): the large feature maps of g_a / g_s
are stored in bf16 and multiplied by bf16-rounded weights with fp32 accumulation; entropy-parameter stacks, variance
mask and likelihoods stay fp32.  The kernel is checked against the same arithmetic in torch (exact bf16 products, fp32
sums), the model against the oracle's bf16 emulation, and the distance to the fp32 path is MEASURED (mask XOR, dPSNR,
dbpp) — never claimed to be zero."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import vampic                              # noqa: E402
import vampic.synth as synth               # noqa: E402
import vampic_oracle as O                  # noqa: E402
from vampic import _lib as L, layers as Ly, ops     # noqa: E402

q = lambda t: t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("cin,n,k,stride,hw", [(192, 192, 5, 2, (32, 48)), (16, 192, 3, 1, (24, 40)), (96, 96, 3, 1, (16, 24)),
                                               (192, 576, 1, 1, (16, 16)), (192, 96, 1, 1, (24, 24))])
@pytest.mark.parametrize("in16", [False, True])
@pytest.mark.parametrize("out16", [False, True])
def test_bf16_conv_kernel(cin, n, k, stride, hw, in16, out16):
    """bf16 x bf16 -> fp32 MFMA conv against F.conv2d on the bf16-rounded operands (products exact, sums fp32: 1e-5 of
    the output scale); a bf16 output may differ from the rounded reference by one bf16 ulp where the fp32 value sits on a
    rounding boundary."""
    B = 2
    m = Ly.Conv2d(cin, n, k, stride).cuda()
    sd = synth.synth_state_dict(m.state_dict(), 3)
    m.load_state_dict(sd)
    x = synth.normal((B, cin) + hw, 5)
    res = synth.normal((B, n, hw[0] // stride, hw[1] // stride), 6)
    ref = F.gelu(F.conv2d(q(x), q(sd["weight"]), sd["bias"], stride=stride, padding=k // 2) + q(res))
    if in16:
        xv = ops.new_view16(B, hw[0], hw[1], cin)
        xv.buf.copy_(x.cuda().permute(0, 2, 3, 1))
    else:
        xv = ops.from_nchw(x.cuda())
    rv = ops.new_view16(B, hw[0] // stride, hw[1] // stride, n)
    rv.buf.copy_(res.cuda().permute(0, 2, 3, 1))
    o = (ops.new_view16 if out16 else ops.new_view)(B, hw[0] // stride, hw[1] // stride, n)
    ops.conv_group([ops.conv_problem(m.packed(True), [xv], o, L.ACT_GELU, pre=rv)])
    got = o.torch_nchw().float().cpu()
    scale = ref.abs().max().item()
    if out16:
        assert ((got - q(ref)).abs() <= 2.0 ** -7 * ref.abs() + 1e-5 * scale).all()
        assert ((got - ref).abs() <= 2.0 ** -8 * ref.abs() + 2e-5 * scale).all()          # within half an ulp (+ sum order) of the fp32 value
    else:
        assert (got - ref).abs().max().item() <= 2e-5 * scale


def test_bf16_gdn_and_deconv():
    B, C, H, W = 2, 192, 16, 24
    dec, g = Ly.ConvTranspose2d(C, C).cuda(), Ly.GDN(C, inverse=True).cuda()
    sd_d, sd_g = synth.synth_state_dict(dec.state_dict(), 7), synth.synth_state_dict(g.state_dict(), 8)
    dec.load_state_dict(sd_d)
    g.load_state_dict(sd_g)
    x = synth.normal((B, C, H, W), 9)
    xv = ops.new_view16(B, H, W, C)
    xv.buf.copy_(x.cuda().permute(0, 2, 3, 1))
    t = ops.new_view16(B, 2 * H, 2 * W, C)
    ops.conv_group([ops.conv_problem(pk, [xv], t) for pk in dec.packed(True)])
    y = ops.new_view16(B, 2 * H, 2 * W, C)
    ops.conv_group([ops.conv_problem(g.packed(True), [t], y, L.ACT_SQRT, mul=t, flags=L.CONV_SQUARE_IN)])
    with O.bf16_storage(min_hw=1):
        ref_t = O._st(O.deconv_k({"d." + k: v for k, v in sd_d.items()}, "d.", q(x)))
        ref_y = O.gdn({"g." + k: v for k, v in sd_g.items()}, "g.", ref_t, True)
    gt, gy = t.torch_nchw().cpu(), y.torch_nchw().cpu()
    assert ((gt - ref_t).abs() <= 2.0 ** -7 * ref_t.abs() + 1e-5 * ref_t.abs().max()).all()
    # the IGDN input may already differ by an ulp: compare on the kernel's own input
    with O.bf16_storage(min_hw=1):
        ref_y2 = O.gdn({"g." + k: v for k, v in sd_g.items()}, "g.", gt, True)
    assert ((gy - ref_y2).abs() <= 2.0 ** -7 * ref_y2.abs() + 1e-5 * ref_y2.abs().max()).all()
    assert float((gy - ref_y).abs().max()) <= 0.05 * float(ref_y.abs().max())


@pytest.mark.parametrize("hw,batch", [((16, 16), 2), ((24, 40), 2), ((13, 21), 1), ((64, 64), 3)])
def test_fused_residual_unit_bf16_storage_is_bit_identical(hw, batch, monkeypatch):
    """csrc/resunit.hip resunit192_bf16_kernel: a ResidualUnit on bf16-stored tensors as ONE launch (t1 / t2 rounded to
    bf16 in LDS exactly where the three-launch form stores them).  Bit-identical to three bf16 conv launches, ragged image
    sizes included."""
    from vampic import engine
    m = Ly.ResidualUnit(192).cuda()
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), 31))
    x = synth.normal((batch, 192) + hw, 32)
    outs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("VAMPIC_FUSED_RU", fused)
        xv = ops.new_view16(batch, hw[0], hw[1], 192)
        xv.buf.copy_(x.cuda().permute(0, 2, 3, 1))
        assert ops.resunit_supported(xv) == (fused == "1")
        plan = engine.Plan("cuda")
        plan.act16, plan.act16_min_hw = True, 1
        o = engine.lower_residual_units(plan, [m], [xv])[0]
        assert isinstance(o, ops.View16) and len(plan.steps) == (1 if fused == "1" else 3)
        plan.run()
        torch.cuda.synchronize()
        outs[fused] = o.buf.clone()
    monkeypatch.delenv("VAMPIC_FUSED_RU")
    a, b = outs["1"].float(), outs["0"].float()
    assert torch.isfinite(a).all()
    assert torch.equal(outs["1"], outs["0"]), f"{(a != b).sum().item()} of {a.numel()} elements differ, max {(a - b).abs().max().item():.3e}"
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    with O.bf16_storage(min_hw=1):
        t = O._st(F.gelu(F.conv2d(q(x), q(sd["conv.0.weight"]), sd["conv.0.bias"])))
        t = O._st(F.gelu(F.conv2d(t, q(sd["conv.2.weight"]), sd["conv.2.bias"], padding=1)))
        ref = O._st(F.gelu(F.conv2d(t, q(sd["conv.4.weight"]), sd["conv.4.bias"]) + q(x)))
    got = outs["1"].permute(0, 3, 1, 2).float().cpu()
    assert float((got - ref).abs().max()) <= 0.03 * float(ref.abs().max())      # intermediates one bf16 ulp apart propagate


def test_bf16_storage_model_vs_emulation_and_fp32(gpu_model):
    """forward_single_quality with storage = "bf16" on 2 x 256 x 256: (a) against the oracle's bf16 emulation (same
    rounding points; both sides round at bf16 boundaries, so single elements may sit one ulp apart and cascade: the
    latent must agree to 2 % of its rms, PSNR to 0.2 dB, bpp to 2 %); (b) the distance to the fp32 HIP path is reported:
    mask XOR, dPSNR, dbpp (bounded loosely: a bf16 model is not expected to reproduce fp32 rounding decisions)."""
    import copy
    net, sd = gpu_model
    m16 = copy.deepcopy(net)
    m16.storage = "bf16"
    B, H, W, qual = 2, 256, 256, 2.5
    x = synth.synth_image(B, H, W, seed=4)
    with torch.no_grad():
        o16 = m16.forward_single_quality(x.cuda(), qual)
        o32 = net.forward_single_quality(x.cuda(), qual)
    with O.bf16_storage():
        ref = O.forward_single_quality(sd, x, qual)
    plan = [p for k, p in m16._plans.items() if "bf16" in k][0]
    y16 = plan.y.torch_nchw().cpu()
    rms = float(ref["y"].pow(2).mean().sqrt())
    e_y = float((y16 - ref["y"]).pow(2).mean().sqrt()) / rms
    psnr16, psnr_ref, psnr32 = O.psnr(x, o16["x_hat"].cpu()), O.psnr(x, ref["x_hat"]), O.psnr(x, o32["x_hat"].cpu())
    bpp = lambda o: -float(o["log2_likelihood_sum"].sum()) / (B * H * W)
    bpp_ref = O.bpp(ref["likelihoods"], B * H * W)
    print(f"bf16 storage vs emulation: latent rms err {e_y:.2e}, PSNR {psnr16:.4f} vs {psnr_ref:.4f}, bpp {bpp(o16):.5f} vs {bpp_ref:.5f}")
    assert e_y <= 2e-2 and abs(psnr16 - psnr_ref) <= 0.2 and abs(bpp(o16) - bpp_ref) <= 0.02 * bpp_ref
    xor = int((o16["mask"] != o32["mask"]).sum())
    e32 = float((y16 - plan_y32(net, B, H, W)).pow(2).mean().sqrt()) / rms
    print(f"bf16 storage vs fp32 path: latent rms err {e32:.2e}, mask XOR {xor} of {o32['mask'].numel()}, "
          f"dPSNR {psnr16 - psnr32:+.4f} dB, dbpp {bpp(o16) - bpp(o32):+.5f}")
    assert e32 <= 0.05 and xor <= 0.2 * o32["mask"].numel() and abs(psnr16 - psnr32) <= 1.0 and abs(bpp(o16) - bpp(o32)) <= 0.05 * bpp(o32)
    # bf16 is an inference configuration: the bitstream path and training stay fp32, loudly
    with pytest.raises(NotImplementedError):
        m16.compress(x.cuda(), quality=qual)


def plan_y32(net, B, H, W):
    for k, p in net._plans.items():
        if k[:5] == (B, H, W, False, None) and len(k) == 6:
            return p.y.torch_nchw().cpu()
    raise KeyError
