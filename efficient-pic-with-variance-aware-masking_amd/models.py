"""Host-side mirror of the reference's model classes (``src/models/*.py``).

``get_model(args, device)``, ``models``, ``VarianceMaskingPIC`` and
``VarianceMaskingPICREM`` keep the reference's constructor arguments, attribute names,
sub-module tree and ``state_dict`` keys (so ``src/demo.py`` / ``src/train.py`` style
callers and reference checkpoints map onto them), while ``forward_single_quality`` is
lowered once per input shape into a :class:`engine.Plan` of libvampic launches
(optionally replayed as one hipGraph).

Reference: models/__init__.py:5-55, models/base.py:6-70, models/builder.py:4-136,
models/pic.py:25-666, models/rem_pic.py:8-422.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from . import layers as Ly
from . import ops
from .entropy_models import EntropyBottleneck, GaussianConditional, get_scale_table


# ----------------------------------------------------------------------------- builders
def define_encoder(multiple_encoder, N, M, dimensions_M):
    """models/builder.py:39-67."""
    def one(out):
        return Ly.TransformStack(
            Ly.conv(3, N, 5, 2), Ly.GDN(N), Ly.conv(N, N, 5, 2), Ly.GDN(N),
            Ly.Win_noShift_Attention(dim=N, num_heads=8, window_size=8, shift_size=4),
            Ly.conv(N, N, 5, 2), Ly.GDN(N), Ly.conv(N, out, 5, 2),
            Ly.Win_noShift_Attention(dim=out, num_heads=8, window_size=4, shift_size=2))
    return nn.ModuleList(one(dimensions_M[0]) for _ in range(2)) if multiple_encoder else one(M)


def define_decoder(multiple_decoder, N, M, dimensions_M):
    """models/builder.py:4-32."""
    def one():
        d = dimensions_M[0]
        return Ly.TransformStack(
            Ly.Win_noShift_Attention(dim=d, num_heads=8, window_size=4, shift_size=2),
            Ly.deconv(d, N, 5, 2), Ly.GDN(N, inverse=True), Ly.deconv(N, N, 5, 2), Ly.GDN(N, inverse=True),
            Ly.Win_noShift_Attention(dim=N, num_heads=8, window_size=8, shift_size=4),
            Ly.deconv(N, N, 5, 2), Ly.GDN(N, inverse=True), Ly.deconv(N, 3, 5, 2))
    return nn.ModuleList(one() for _ in range(2)) if multiple_decoder else one()


def _hyper_synthesis(cin, c0, cout):
    return Ly.ConvStack(Ly.conv3x3(cin, c0), Ly.GELU(), Ly.subpel_conv3x3(c0, 224, 2), Ly.GELU(),
                        Ly.conv3x3(224, 256), Ly.GELU(), Ly.subpel_conv3x3(256, 288, 2), Ly.GELU(),
                        Ly.conv3x3(288, cout))


def define_hyperprior(multiple_hyperprior, M, N, dimensions_M):
    """models/builder.py:71-136."""
    h_a = Ly.ConvStack(Ly.conv3x3(M, 320), Ly.GELU(), Ly.conv3x3(320, 288), Ly.GELU(), Ly.conv3x3(288, 256, stride=2),
                       Ly.GELU(), Ly.conv3x3(256, 224), Ly.GELU(), Ly.conv3x3(224, N, stride=2))
    if multiple_hyperprior:
        h_mean_s = nn.ModuleList(_hyper_synthesis(N, 192, dimensions_M[0]) for _ in range(2))
        h_scale_s = nn.ModuleList(_hyper_synthesis(N, 192, dimensions_M[0]) for _ in range(2))
    else:
        h_mean_s = _hyper_synthesis(N, N, M)
        h_scale_s = _hyper_synthesis(192, 192, M)
    return h_a, h_mean_s, h_scale_s


def _param_stack(cin, c_head):
    """Five conv3x3 with GELU between: cin -> 224 -> 176 -> 128 -> 64 -> 32 (models/pic.py:83-164).
    ``c_head``: how many leading input channels are the hyperprior tensor (the rest are support slices).  The fused
    plans compute the first layer as  conv(hyper; W[:, :c_head]) + conv(supports; W[:, c_head:])  (engine.lower_stack_heads);
    a module-level call ``stack(torch.cat([hyper, *supports]))`` must associate the sum the same way, or the decoder of
    the progressive container (module-level calls, test/functions_decode.py) would see a sigma that differs from the
    encoder's (fused plan) in the last bit — enough to desynchronise the range coder."""
    widths = (cin, 224, 176, 128, 64, 32)
    mods = []
    for a, b in zip(widths[:-1], widths[1:]):
        mods += [Ly.conv(a, b, kernel_size=3, stride=1), Ly.GELU()]
    st = Ly.ConvStack(*mods[:-1])
    st.c_head = c_head
    return st


class CompressionModel(nn.Module):
    """models/base.py:6-70 (conv weights kaiming-normal at construction, zero biases)."""

    def __init__(self, init_weights=True):
        super().__init__()

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def update(self, force=False):
        """models/base.py:41-60."""
        updated = False
        for m in self.children():
            if isinstance(m, EntropyBottleneck):
                updated |= m.update(force=force)
        return updated


def _resize_cdf_buffers(module, prefix, names, state_dict):
    """models/utils.py:41-93 'resize_if_empty': let checkpoints carrying CDF tables load."""
    bufs = dict(module.named_buffers())
    for n in names:
        key = f"{prefix}.{n}"
        if key in state_dict and n in bufs and bufs[n].numel() == 0:
            bufs[n].resize_(state_dict[key].size())


class VarianceMaskingPIC(CompressionModel):
    """models/pic.py:25-967."""

    def __init__(self, N=192, M=640, division_dimension=[320, 640], dim_chunk=32, multiple_decoder=True,
                 multiple_encoder=True, multiple_hyperprior=True, support_progressive_slices=5, delta_encode=True,
                 total_mu_rep=True, all_scalable=True, mask_policy="point-based-std", **kwargs):
        super().__init__(**kwargs)
        self.N, self.M, self.dim_chunk = N, M, dim_chunk
        self.num_slices = int(M // dim_chunk)
        self.multiple_encoder, self.multiple_decoder = multiple_encoder, multiple_decoder
        self.multiple_hyperprior = multiple_hyperprior
        self.division_channel = division_dimension[0]
        self.division_dimension = division_dimension
        self.support_progressive_slices = support_progressive_slices
        self.delta_encode, self.total_mu_rep, self.all_scalable = delta_encode, total_mu_rep, all_scalable
        self.mask_policy = mask_policy
        self.quality_list = [0, 10]
        self.max_support_slices = 5
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.gaussian_conditional = GaussianConditional(None)
        self.masking = Ly.ChannelMask(mask_policy)
        self.num_slice_cumulative_list = [p // dim_chunk for p in division_dimension]
        self.ns0, self.ns1 = self.num_slice_cumulative_list[0], self.num_slice_cumulative_list[1]
        d0 = division_dimension[0]
        delta = division_dimension[1] - division_dimension[0]
        sp1 = support_progressive_slices + 1

        self.g_a = define_encoder(multiple_encoder, N, M, division_dimension)
        self.g_s = define_decoder(multiple_decoder, N, M, division_dimension)
        self.h_a, self.h_mean_s, self.h_scale_s = define_hyperprior(multiple_hyperprior, M, N, division_dimension)
        nb, np_ = self.ns0, self.ns1 - self.ns0
        self.cc_mean_transforms = nn.ModuleList(_param_stack(d0 + 32 * min(i, 5), d0) for i in range(nb))
        self.cc_scale_transforms = nn.ModuleList(_param_stack(d0 + 32 * min(i, 5), d0) for i in range(nb))
        self.lrp_transforms = nn.ModuleList(_param_stack(d0 + 32 * min(i + 1, 6), d0) for i in range(nb))
        self.cc_mean_transforms_prog = nn.ModuleList(_param_stack(delta + 32 * min(i + 1, sp1), delta) for i in range(np_))
        self.cc_scale_transforms_prog = nn.ModuleList(_param_stack(delta + 32 * min(i + 1, sp1), delta) for i in range(np_))
        self.lrp_transforms_prog = nn.ModuleList(_param_stack(delta + 32 * min(i + 2, sp1 + 1), delta) for i in range(nb))
        self._plans: Dict[tuple, "_FsqPlan"] = {}
        self._dec_plans: Dict[tuple, "_DecPlan"] = {}
        self.use_graph = True
        # "fp32" (default; every parity claim) or "bf16": BASELINE configs[2] — the large feature maps of g_a / g_s are
        # stored in bf16 and multiplied by bf16-rounded weights (fp32 accumulation); the entropy-parameter stacks, the
        # variance mask and the likelihoods stay fp32.  Differences to the fp32 path are MEASURED (bench.py --dtype bf16).
        self.storage = "fp32"

    # ---- reference helpers kept for the harness
    def freeze_all(self):
        for p in self.parameters():
            p.requires_grad = False

    def unfreeze_decoder(self, lrp=False):
        target = self.g_s if not self.multiple_decoder else self.g_s[1]
        for p in target.parameters():
            p.requires_grad = True
        if lrp:
            for p in self.lrp_transforms_prog.parameters():
                p.requires_grad = True

    def unfreeze_encoder(self):
        target = self.g_s if not self.multiple_encoder else self.g_a[1]   # (sic) pic.py:189-191
        for p in target.parameters():
            p.requires_grad = True

    def print_information(self):
        for name in ("g_a", "h_a", "h_mean_s", "h_scale_s", "cc_mean_transforms", "cc_scale_transforms",
                     "cc_mean_transforms_prog", "cc_scale_transforms_prog", "lrp_transforms", "g_s"):
            print(f" {name}: ", sum(p.numel() for p in getattr(self, name).parameters()))
        tr = sum(p.numel() for p in self.parameters() if p.requires_grad)
        print(" trainable parameters: ", tr)
        print(" freeze parameterss: ", sum(p.numel() for p in self.parameters() if not p.requires_grad))
        return tr

    def update(self, scale_table=None, force=True):
        """models/pic.py:230-237: scale table + CDF tables of both entropy models."""
        if scale_table is None:
            scale_table = get_scale_table()
        self.gaussian_conditional.update_scale_table([float(s) for s in scale_table])
        self.entropy_bottleneck.update(force=force)
        self._drop_plans()
        return True

    def load_state_dict(self, state_dict, strict=True):
        _resize_cdf_buffers(self.gaussian_conditional, "gaussian_conditional",
                            ["_quantized_cdf", "_offset", "_cdf_length", "scale_table"], state_dict)
        _resize_cdf_buffers(self.entropy_bottleneck, "entropy_bottleneck",
                            ["_quantized_cdf", "_offset", "_cdf_length"], state_dict)
        self._drop_plans()
        for em in (self.gaussian_conditional, self.entropy_bottleneck):      # loaded CDF tables replace the cached host copies
            object.__setattr__(em, "_tables_generation", getattr(em, "_tables_generation", 0) + 1)
        return nn.Module.load_state_dict(self, state_dict, strict=strict)

    def _drop_plans(self):
        """Forget every plan: their executable graphs are handed to ops' deferred-destroy list explicitly (not left to
        whenever the garbage collector finds the plans' reference cycles) and destroyed at the next plan entry point,
        after their last replay has finished."""
        for p in list(self._plans.values()) + list(self._dec_plans.values()):
            close = getattr(p, "close", None)
            if close is not None:
                close()
        self._plans.clear()
        self._dec_plans.clear()

    def _apply(self, fn, *a, **k):
        self._drop_plans()
        self.__dict__.pop("_sig_params", None)
        return super()._apply(fn, *a, **k)

    def __deepcopy__(self, memo):
        """Plans hold device pointers, HIP graphs and streams of THIS instance: a copy starts without them."""
        import copy
        held = self._plans, self._dec_plans
        self._plans, self._dec_plans = {}, {}
        self.__dict__.pop("_sig_params", None)
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            new.__dict__ = copy.deepcopy(self.__dict__, memo)
        finally:
            self._plans, self._dec_plans = held
        return new

    def define_quality(self, quality):
        if quality is None:
            return self.quality_list
        if isinstance(quality, list):
            return quality if quality[0] == 0 else [0] + quality
        return [quality]

    def determine_support(self, y_hat_base, current_index, y_hat_quality):
        bi = y_hat_base[current_index]
        if current_index == 0 or self.support_progressive_slices == 0:
            return [bi]
        s = min(self.support_progressive_slices, current_index)
        return [bi] + y_hat_quality[current_index - s:current_index]

    def merge(self, y_base, y_enhanced):
        return y_base + y_enhanced

    def compute_hyperprior(self, y, quality=10):
        """models/pic.py:278-298 on NCHW tensors (module-level path of the harness)."""
        z = self.h_a(y)
        z_hat, z_lik = self.entropy_bottleneck(z, training=False)
        if not self.multiple_hyperprior:
            return self.h_mean_s(z_hat), self.h_scale_s(z_hat), z_lik
        if quality == 0:
            return self.h_mean_s[0](z_hat), self.h_scale_s[0](z_hat), z_lik
        means = torch.cat([self.h_mean_s[0](z_hat), self.h_mean_s[1](z_hat)], dim=1)
        scales = torch.cat([self.h_scale_s[0](z_hat), self.h_scale_s[1](z_hat)], dim=1)
        return means, scales, z_lik

    # ---- the hot path
    def _check_config(self):
        """What the lowering needs.  Every flag of models/__init__.py:11-55 is accepted (single or dual
        encoder / decoder / hyperprior, any support_progressive_slices, delta_encode, total_mu_rep, all_scalable);
        the slice geometry is the one the reference itself can run: 32-channel slices (the stacks end in 32 channels,
        pic.py:83-164) and division [d, 2d] with M = 2d (the progressive stacks read latent_means[:, d:], whose width
        must equal division[1] - division[0], pic.py:75,596-599)."""
        if self.dim_chunk != 32 or self.ns1 != 2 * self.ns0 or self.M != 2 * self.division_dimension[0]:
            raise NotImplementedError(f"slice geometry dim_chunk={self.dim_chunk}, division={self.division_dimension}, M={self.M}: "
                                      "the channel-conditional stacks are built for 32-channel slices and division [d, 2d] "
                                      "with M = 2d (as the reference's own stacks are, pic.py:83-164)")
        if self.support_progressive_slices < 0:
            raise ValueError("support_progressive_slices must be >= 0")

    def _plan(self, x, base_only: bool, rem_idx: Optional[int] = None, symbols: bool = False,
              train: bool = False, own_ck: bool = False, train_gs: bool = False, train_lrp: bool = False) -> "_FsqPlan":
        B, C_, H, W = x.shape
        if C_ != 3 or H % 64 or W % 64:
            raise ValueError(f"expected [B,3,H,W] with H,W multiples of 64 (reference pads to 64), got {tuple(x.shape)}")
        key = (B, H, W, base_only, rem_idx, str(x.device)) + ((True,) if symbols else ()) + (("train",) if train else ()) + \
            (("own_ck",) if own_ck else ()) + (("train_gs",) if train_gs else ()) + (("train_lrp",) if train_lrp else ()) + \
            (("bf16",) if getattr(self, "storage", "fp32") == "bf16" else ())
        if getattr(self, "storage", "fp32") == "bf16" and (train or symbols):
            raise NotImplementedError("bf16 storage is an inference configuration (forward_single_quality): training and "
                                      "the bitstream path run in fp32")
        if ops.f16x2_mode() and (train or symbols):
            raise NotImplementedError("the fp16x2 arithmetic (VAMPIC_CONV=f16x2) is an evaluation-forward configuration: its results depend, in the last bits, on the power-of-two scale of each launch (batch composition, plan structure), so the bitstream path (encoder and decoder must agree bit for bit) and training run in the default bf16x3 arithmetic")
        p = self._plans.get(key)
        if p is not None and rem_idx is not None and not train and p.rem_sig != _version_sig(self.post_latent[rem_idx]):
            p = None                # the REM was fine-tuned since this plan packed its weights
        trained = ([self._decoder_in_use(base_only)] if train_gs else []) + ([self.lrp_transforms_prog] if train_lrp else [])
        wsig = self._weights_sig(trained)
        if p is not None and p.wsig != wsig:
            p = None                # a parameter was edited in place (param.data.copy_, nn.init, optimizer step)
        if p is None:
            old = self._plans.pop(key, None)
            if old is not None:
                old.close()
            p = _FsqPlan(self, B, H, W, base_only, rem_idx, x.device, symbols=symbols, train=train, own_ck=own_ck,
                         train_gs=train_gs, train_lrp=train_lrp)
            p.wsig = wsig
            self._plans[key] = p
        return p

    def _decoder_in_use(self, base_only: bool):
        return (self.g_s[0 if base_only else 1] if self.multiple_decoder else self.g_s)

    def _weights_sig(self, trained: Sequence[nn.Module] = ()):
        """(_version, data_ptr) of every parameter the plans pack ONCE (everything except ``post_latent`` — and except
        ``trained``, the modules a training plan re-packs in place every step): a plan built before an in-place edit
        of a weight must not be replayed."""
        cache = self.__dict__.setdefault("_sig_params", {})
        key = tuple(id(t) for t in trained)
        ps = cache.get(key)
        if ps is None:
            skip = {id(p) for t in trained for p in t.parameters()}
            ps = [p for n, p in self.named_parameters() if not n.startswith("post_latent.") and id(p) not in skip]
            cache[key] = ps
        h = 0
        for p in ps:
            h = (h * 1000003 + p._version * 31 + (p.data_ptr() >> 4)) & 0xFFFFFFFFFFFFFFF
        return h

    def _trainable_outside_rem(self):
        return [n for n, p in self.named_parameters() if p.requires_grad and not n.startswith("post_latent.")]

    def forward_single_quality(self, x, quality, mask_pol="point-based-std", training=False, clone=True, noise=None):
        """models/pic.py:497-666.  Returns the reference's dict; tensors are NCHW-shaped.  ``training=True``: the
        additive-uniform-noise likelihoods of the training forward (entropy_models.py:132-138; the latents themselves
        are STE-rounded, so every other output equals the eval pass) — VALUES only: the transforms outside the REMs have
        no backward kernels in this build, so asking for their gradients fails loudly instead of silently
        returning none.  ``noise`` = {"y": NCHW, "z": NCHW} injects fixed draws."""
        train_gs = train_lrp = False
        if training and torch.is_grad_enabled() and self._trainable_outside_rem():
            # `--training_type refine_gs` (train.py:150-157,216-218): the synthesis transform in use trains, with `--lrp`
            # (unfreeze_decoder(lrp=True), pic.py:171-184) the progressive latent-residual-prediction stacks as well
            dec_ids = {id(p) for p in self._decoder_in_use(quality == 0).parameters()}
            lrp_ps = list(self.lrp_transforms_prog.parameters())
            lrp_ids = {id(p) for p in lrp_ps}
            other = [n for n, p in self.named_parameters()
                     if p.requires_grad and not n.startswith("post_latent.") and id(p) not in dec_ids and id(p) not in lrp_ids]
            n_lrp = sum(p.requires_grad for p in lrp_ps)
            if other or not self.all_scalable or (n_lrp and (quality == 0 or n_lrp != len(lrp_ps))) or \
                    not any(p.requires_grad for p in self._decoder_in_use(quality == 0).parameters()):
                # anything beyond the decoder-refinement subsets (first_train: everything; refine_gs_ga: g_s[1] + g_a[1],
                # train.py:219-222): the complete training plan (full_train.py)
                return self._forward_full_train(x, [quality], mask_pol, noise, single=True)
            train_gs, train_lrp = True, bool(n_lrp)
        mask_pol = self.mask_policy if mask_pol is None else mask_pol
        if mask_pol not in ("point-based-std", "two-levels"):
            raise NotImplementedError()
        Ly._no_autograd(x)
        L.require_gpu()
        self._check_config()
        pr = quality
        if mask_pol == "two-levels" and quality != 0:
            pr = 10                                   # channel_mask.py:152-153: all ones unless pr == 0
        nb = _max_images_per_plan(x)
        if train_gs and x.shape[0] > nb:                # one tape per plan: sub-batches would overwrite each other's
            raise NotImplementedError(f"training with gradients: at most {nb} images of {x.shape[2]}x{x.shape[3]} per step "
                                      "(one plan's 32-bit addressing range); split the batch and accumulate")
        if x.shape[0] > nb:                            # tensors of one plan are addressed with 32-bit byte offsets
            sub = lambda i: None if noise is None else {k: v[i:i + nb] for k, v in noise.items()}
            return _cat_outputs([self.forward_single_quality(x[i:i + nb], quality, mask_pol, training, True, sub(i))
                                 for i in range(0, x.shape[0], nb)])
        plan = self._plan(x.detach(), base_only=(quality == 0), train=bool(training), train_gs=train_gs, train_lrp=train_lrp)
        out = plan.execute(x.detach(), pr, None, self.use_graph, clone, noise=noise)
        if train_gs:
            out["x_hat"] = _GsTrainFn.apply(plan, out["x_hat"], self.use_graph, *plan.gs_params)
        return out

    def forward(self, x, quality=None, mask_pol=None, training=True, noise=None):
        """models/pic.py:301-491: the base pass plus one progressive pass per requested quality (default [0, 10]),
        stacked as the reference stacks them.  ``training=True`` evaluates the likelihoods with additive uniform noise
        (values only, see :meth:`forward_single_quality`); the same noise tensors serve every quality, as one
        ``uniform_`` draw per slice would in a single reference pass."""
        qs = self.define_quality(quality)
        if training and torch.is_grad_enabled() and self._trainable_outside_rem():
            return self._forward_full_train(x, qs, mask_pol, noise, single=False)
        base = self.forward_single_quality(x, 0, mask_pol, training, noise=noise)
        x_hats, y_prog, y_hat_total = [base["x_hat"].unsqueeze(0)], [], [base["y_hat"]]
        out = None
        for q in qs[1:]:
            out = self.forward_single_quality(x, q, mask_pol, training, noise=noise)
            x_hats.append(out["x_hat"].unsqueeze(0))
            y_prog.append(out["likelihoods"]["y"].unsqueeze(0))
            y_hat_total.append(out["y_hat"])
        lik_b = base["likelihoods"]["y"]
        return {"x_hat": torch.cat(x_hats, 0),
                "likelihoods": {"y": lik_b, "y_prog": torch.cat(y_prog, 0) if y_prog else lik_b,
                                "z": base["likelihoods"]["z"]},
                "y_hat": y_hat_total, "y_base": base["y_hat"], "y_prog": out["y_hat"] if out else base["y_hat"]}

    def _forward_full_train(self, x, qs, mask_pol, noise, single: bool):
        """Training forward WITH gradients of every trainable parameter (BASELINE configs[3] `first_train`, train.py:146-149;
        `refine_gs_ga` as a subset): full_train.FullTrainPlan, autograd-connected through :class:`_FullTrainFn`.
        ``single`` = False: ``forward(x, [0, q])`` (pic.py:301-491); True: ``forward_single_quality(x, q)`` (:497-666)."""
        from .full_train import FullTrainPlan
        if ops.f16x2_mode():
            raise NotImplementedError("the fp16x2 arithmetic (VAMPIC_CONV=f16x2) is an evaluation-forward configuration: its results depend, in the last bits, on the power-of-two scale of each launch (batch composition, plan structure), so the bitstream path (encoder and decoder must agree bit for bit) and training run in the default bf16x3 arithmetic")
        mask_pol = self.mask_policy if mask_pol is None else mask_pol
        if mask_pol not in ("point-based-std", "two-levels"):
            raise NotImplementedError()
        if not single and (len(qs) != 2 or qs[0] != 0 or qs[1] == 0):
            raise NotImplementedError("training forward with gradients: quality lists [0, q] (train.py:147: [0, 10]); "
                                      f"got {qs}")
        Ly._no_autograd(x)
        L.require_gpu()
        self._check_config()
        q = qs[-1]
        nb = _max_images_per_plan(x)
        if x.shape[0] > nb:
            raise NotImplementedError(f"training with gradients: at most {nb} images of {x.shape[2]}x{x.shape[3]} per step")
        B, C_, H, W = x.shape
        if C_ != 3 or H % 64 or W % 64:
            raise ValueError(f"expected [B,3,H,W] with H,W multiples of 64 (reference pads to 64), got {tuple(x.shape)}")
        base_only = single and q == 0
        key = ("full_train", B, H, W, "single" if single else "multi", base_only, str(x.device))
        plan = self._plans.get(key)
        if plan is None:
            plan = FullTrainPlan(self, B, H, W, "single" if single else "multi", base_only, x.device)
            self._plans[key] = plan
        pr = 10 if (mask_pol == "two-levels" and q != 0) else q
        raw = plan.execute(x.detach(), pr, self.use_graph, noise=noise)
        x_hat, lik, z_lik = _FullTrainFn.apply(plan, self.use_graph, getattr(self, "grad_reducer", None), raw["x_hat"], raw["lik"],
                                               raw["z_lik"], *plan.params)
        d = self.division_dimension[0]
        if single:
            yh = raw["y_base"] if base_only else raw["y_prog"]
            out = {"x_hat": x_hat[0], "likelihoods": {"y": lik, "z": z_lik}, "y_hat": yh, "y_base": raw["y_base"], "y_prog": yh,
                   "mu_base": raw["mu_base"], "std_base": raw["std_base"]}
            if not base_only:
                out.update({"mu": raw["mu"], "std": raw["std"], "mask": raw["mask"]})
            return out
        return {"x_hat": x_hat, "likelihoods": {"y": lik[:, :d], "y_prog": lik.unsqueeze(0), "z": z_lik},   # pic.py:389-390,471-472,486-491
                "y_hat": [raw["y_base"], raw["y_prog"]], "y_base": raw["y_base"], "y_prog": raw["y_prog"],
                "mu_base": raw["mu_base"], "std_base": raw["std_base"], "mu_prog": raw["mu"], "std_prog": raw["std"]}

    # ---- bitstream path (models/pic.py:671-967; rem_pic.py:425-818)
    def _rem_choice(self, quality, checkpoint_rep):
        return None                                    # no REM in the plain model

    def compress(self, x, quality=0.0, mask_pol=None, checkpoint_rep=None, real_compress=True):
        """One rANS stream per (slice, image) for y and per image for z.  The latents, entropy
        parameters, masks, symbols and table indexes come from the fused HIP plan; only the
        bit-serial coder runs on the host (as in the reference, entropy_models.py:231-239)."""
        mask_pol = self.mask_policy if mask_pol is None else mask_pol
        if mask_pol not in ("point-based-std", "two-levels"):
            raise NotImplementedError()
        Ly._no_autograd(x)
        L.require_gpu()
        self._check_config()
        base_only = quality <= 0
        rem_idx = None if base_only else self._rem_choice(quality, checkpoint_rep)
        pr = 10 if (mask_pol == "two-levels" and quality != 0) else quality
        plan = self._plan(x, base_only=base_only, rem_idx=rem_idx, symbols=True)
        out = plan.execute(x, pr, checkpoint_rep if rem_idx is not None else None, self.use_graph, True)
        B, C = plan.B, self.dim_chunk
        n_sl = self.ns0 if base_only else self.ns1
        y_strings: List[List[bytes]] = []
        if real_compress:
            from . import bitstream as bs
            if plan.idx is None:
                raise ValueError("empty scale table: call model.update() before compress()")
            tg, te = bs.Tables.of(self.gaussian_conditional), bs.Tables.of(self.entropy_bottleneck)
            sym = plan.sym.buf.cpu().numpy()           # [B,h,w,C_lat] int32 (synchronises)
            idx = plan.idx.buf.cpu().numpy()
            zs = plan.z_sym.buf.cpu().numpy()
            for i in range(n_sl):                       # stream order: [C, h, w] per image, as the reference flattens
                sl = slice(i * C, (i + 1) * C)
                y_strings.append([bs.encode(sym[b, :, :, sl].transpose(2, 0, 1), idx[b, :, :, sl].transpose(2, 0, 1), tg)
                                  for b in range(B)])
            zi = torch.arange(self.N, dtype=torch.int32).numpy()[:, None, None]
            z_strings = [bs.encode(zs[b].transpose(2, 0, 1), np.broadcast_to(zi, (self.N,) + zs.shape[1:3]), te)
                         for b in range(B)]
        else:                                           # rem_pic.py:498-500,594-597: the quantised tensors instead of bytes
            sy = plan.sym.buf.permute(0, 3, 1, 2)
            y_strings = [sy[:, i * C:(i + 1) * C].float() for i in range(n_sl)]
            z_strings = [plan.z_sym.buf.permute(0, 3, 1, 2).float()]
        res = {"strings": [y_strings, z_strings], "shape": (plan.H // 64, plan.W // 64),
               "masks": [] if base_only else list(out["mask"].chunk(self.ns0, 1)), "y_hat": out["y_hat"]}
        if base_only:
            res.update({"mean_base": out["mu_base"], "scale_base": out["std_base"], "std_base": out["std_base"],
                        "y_hat_base": out["y_hat"]})
        return res

    def decompress(self, strings, shape, quality, mask_pol=None, checkpoint_rep=None):
        """models/pic.py:838-967: z -> hyper-synthesis -> slice by slice (entropy parameters on the
        GPU, rANS decode on the host, LRP on the GPU) -> g_s."""
        mask_pol = self.mask_policy if mask_pol is None else mask_pol
        if mask_pol not in ("point-based-std", "two-levels"):
            raise NotImplementedError()
        L.require_gpu()
        self._check_config()
        dev = self.entropy_bottleneck.quantiles.device
        B = len(strings[1])
        hz, wz = int(shape[0]), int(shape[1])
        base_only = quality == 0
        rem_idx = None if base_only else self._rem_choice(quality, checkpoint_rep)
        key = (B, hz, wz, base_only, rem_idx, str(dev))
        dp = self._dec_plans.get(key)
        if dp is not None and rem_idx is not None and dp.rem_sig != _version_sig(self.post_latent[rem_idx]):
            dp = None               # REM fine-tuned since the plan packed its weights
        wsig = self._weights_sig()
        if dp is not None and dp.wsig != wsig:
            dp = None               # a weight was edited in place since the plan packed it
        if dp is None:
            if ops.f16x2_mode():
                raise NotImplementedError("the fp16x2 arithmetic (VAMPIC_CONV=f16x2) is an evaluation-forward configuration: its results depend, in the last bits, on the power-of-two scale of each launch (batch composition, plan structure), so the bitstream path (encoder and decoder must agree bit for bit) and training run in the default bf16x3 arithmetic")
            dp = _DecPlan(self, B, hz, wz, base_only, rem_idx, dev)
            dp.wsig = wsig
            dp.rem_sig = _version_sig(self.post_latent[rem_idx]) if rem_idx is not None else None
            self._dec_plans[key] = dp
        pr = 10 if (mask_pol == "two-levels" and quality != 0) else quality
        return {"x_hat": dp.decode(strings, pr, checkpoint_rep if rem_idx is not None else None)}


class VarianceMaskingPICREM(VarianceMaskingPIC):
    """models/rem_pic.py:8-818."""

    def __init__(self, N=192, M=640, division_dimension=[320, 416], dim_chunk=32, multiple_decoder=True,
                 multiple_encoder=True, multiple_hyperprior=True, support_progressive_slices=5, delta_encode=True,
                 total_mu_rep=True, all_scalable=True, mask_policy="point-based-std", check_levels=[0.01, 0.25, 1.75],
                 mu_std=True, dimension="big", **kwargs):
        super().__init__(N=N, M=M, division_dimension=division_dimension, dim_chunk=dim_chunk,
                         multiple_decoder=multiple_decoder, multiple_encoder=multiple_encoder,
                         multiple_hyperprior=multiple_hyperprior, support_progressive_slices=support_progressive_slices,
                         delta_encode=delta_encode, total_mu_rep=total_mu_rep, all_scalable=all_scalable,
                         mask_policy=mask_policy, **kwargs)
        self.dimension = dimension
        self.check_levels = check_levels
        self.num_rems = len(check_levels)
        self.enable_rem = [True] * self.num_rems
        self.mu_std = mu_std
        self.post_latent = nn.ModuleList(
            nn.ModuleList(Ly.LatentRateReduction(dim_chunk=dim_chunk, mu_std=mu_std, dimension=dimension) for _ in range(10))
            for _ in range(self.num_rems))

    def unfreeze_rems(self):
        for p in self.post_latent.parameters():
            p.requires_grad = True

    def load_state_dict(self, state_dict, strict=True):
        """Loads parent keys non-strictly and ``post_latent.*`` strictly, as rem_pic.py:66-78 intends
        (the reference forgets to strip the ``post_latent.`` prefix and fails on its own checkpoints;
        both prefixed and stripped keys are accepted here)."""
        own = self.state_dict()
        parent = {k: v for k, v in state_dict.items() if k in own and "post_latent" not in k}
        res = super().load_state_dict(parent, strict=False)
        post = {(k[len("post_latent."):] if k.startswith("post_latent.") else k): v
                for k, v in state_dict.items() if "post_latent" in k}
        if post:
            self.post_latent.load_state_dict(post, strict=True)
            self.enable_rem = [True] * self.num_rems
        else:
            print("This model does not have trained REMs.  self.enable_rem will be set to False")
            self.enable_rem = [False] * self.num_rems
        return res

    def find_check_quality(self, quality):
        """models/rem_pic.py:142-165."""
        cl = self.check_levels
        if quality <= cl[0]:
            return 0, 0, -1
        if len(cl) in (2, 3) and cl[0] < quality <= cl[1]:
            return cl[0], cl[1], 0
        if len(cl) == 2 and quality > cl[1]:
            return cl[1], 10, 1
        if len(cl) == 3 and cl[1] < quality <= cl[2]:
            return cl[1], cl[-1], 1
        return cl[-1], 10, -1

    def _rem_index(self, quality):
        """models/rem_pic.py:200-213."""
        cl = self.check_levels
        if self.num_rems == 1:
            return 0
        if self.num_rems == 2:
            return 0 if cl[0] < quality <= cl[1] else 1
        if cl[0] < quality <= cl[1]:
            return 0
        if cl[1] < quality <= cl[2]:
            return 1
        return 2

    def forward_single_quality(self, x, quality, mask_pol="point-based-std", training=False, checkpoint_ref=None,
                               clone=True, noise=None):
        return self.forward(x=x, quality=quality, mask_pol=mask_pol, training=training, checkpoint_ref=checkpoint_ref,
                            clone=clone, noise=noise)

    def forward(self, x, mask_pol="point-based-std", quality=0, training=True, checkpoint_ref=None, clone=True,
                noise=None):
        """models/rem_pic.py:229-422.  ``training=True`` is the REM fine-tune forward (BASELINE configs[4]):
        additive-noise likelihoods, autograd-connected to the ``post_latent`` parameters only."""
        if training:
            return self._forward_train(x, mask_pol, quality, checkpoint_ref, noise)
        mask_pol = self.mask_policy if mask_pol is None else mask_pol
        if mask_pol not in ("point-based-std", "two-levels"):
            raise NotImplementedError()
        Ly._no_autograd(x)
        L.require_gpu()
        self._check_config()
        nb = _max_images_per_plan(x)
        if x.shape[0] > nb:                            # tensors of one plan are addressed with 32-bit byte offsets
            return _cat_outputs([self.forward(x[i:i + nb], mask_pol, quality, False,
                                              None if checkpoint_ref is None else checkpoint_ref[i:i + nb], True)
                                 for i in range(0, x.shape[0], nb)])
        rem_idx = self._rem_choice(quality, checkpoint_ref) if quality != 0 else None
        pr = 10 if (mask_pol == "two-levels" and quality != 0) else quality
        plan = self._plan(x, base_only=(quality == 0), rem_idx=rem_idx)
        return plan.execute(x, pr, checkpoint_ref if rem_idx is not None else None, self.use_graph, clone)

    def forward_finetune(self, x, quality, mask_pol="point-based-std", noise=None):
        """One fused plan for the fine-tune step's two forward passes: bit-identical to
        ``ck = ExtractChekpointRepr(x, q_ref, rc=False); forward_single_quality(x, quality, training=True,
        checkpoint_ref=ck)`` (training/step.py:67-76) because everything up to the progressive (mu, sigma) chain
        does not depend on the quality — the checkpoint latent is derived from the same front end instead of a
        second run of g_a / hyperprior / slices."""
        from .finetune import extract_quality_ref
        q_ref = extract_quality_ref(quality, self.check_levels)
        if q_ref is None:
            return self._forward_train(x, mask_pol, quality, None, noise)
        return self._forward_train(x, mask_pol, quality, "own", noise, ck_pr=q_ref)

    def _forward_train(self, x, mask_pol, quality, checkpoint_ref, noise, ck_pr=None):
        """Training-mode forward (rem_pic.py:229-422 with training=True; training/step.py:62-76).  Everything
        outside ``post_latent`` must be frozen (``freeze_all(); unfreeze_rems()``): those transforms have no
        backward kernels in this build, and silently dropping their gradients would be wrong."""
        rem_ids = {id(p) for p in self.post_latent.parameters()}
        loose = [n for n, p in self.named_parameters() if p.requires_grad and id(p) not in rem_ids]
        if loose:
            raise NotImplementedError("training-mode forward is built for --training_type rems only (call freeze_all(); "
                                      f"unfreeze_rems()); trainable non-REM parameters: {loose[:3]}...")
        mask_pol = self.mask_policy if mask_pol is None else mask_pol
        if mask_pol not in ("point-based-std", "two-levels"):
            raise NotImplementedError()
        L.require_gpu()
        self._check_config()
        own = isinstance(checkpoint_ref, str)
        if checkpoint_ref is not None and not own:
            checkpoint_ref = checkpoint_ref.detach()
        rem_idx = self._rem_choice(quality, checkpoint_ref) if quality != 0 else None
        pr = 10 if (mask_pol == "two-levels" and quality != 0) else quality
        own = own and rem_idx is not None
        if own and mask_pol == "two-levels":
            ck_pr = 10 if ck_pr != 0 else 0
        plan = self._plan(x.detach(), base_only=(quality == 0), rem_idx=rem_idx, train=True, own_ck=own)
        out = plan.execute(x.detach(), pr, checkpoint_ref if (rem_idx is not None and not own) else None, self.use_graph,
                           True, noise=noise, ck_pr=ck_pr if own else None)
        if rem_idx is not None and torch.is_grad_enabled() and any(p.requires_grad for p in plan.rem_params):
            out["likelihoods"]["y"] = _RemTrainFn.apply(plan, out["likelihoods"]["y"], self.use_graph, *plan.rem_params)
        return out

    def apply_latent_enhancement(self, current_index, quality, quality_bar, y_b_hat, mu_scale_base, mu_scale_enh,
                                 mu, scale, training=False, mask_pol="point-based-std", attention_mask=None):
        """models/rem_pic.py:167-220 on NCHW tensors (module-level surface used by the progressive
        harness, test/functions_encode.py:126-141): refine (mu, scale) of one progressive slice."""
        if attention_mask is None:
            attention_mask = self.masking(scale, pr=quality, mask_pol=mask_pol)
        if self.mu_std:
            attention_mask = torch.cat([attention_mask, attention_mask], dim=1)
        if quality <= self.check_levels[0]:
            return mu, scale
        block = self.post_latent[self._rem_index(quality)][current_index]
        enhanced = block(y_b_hat, mu_scale_base, mu_scale_enh, attention_mask)
        if self.mu_std:
            mu, scale = enhanced.chunk(2, 1)
            return mu, scale
        return mu, enhanced

    def _rem_choice(self, quality, checkpoint_rep):
        """Which REM (if any) refines the entropy parameters (rem_pic.py:197-213,363,566)."""
        if checkpoint_rep is None or quality <= self.check_levels[0]:
            return None
        _, _, right = self.find_check_quality(quality)
        return self._rem_index(quality) if self.enable_rem[right] else None

    def ExtractChekpointRepr(self, x, quality, rc=True, y_check=None):
        """rem_pic.py:121-132: compress(...)["y_hat"].  The coder is lossless, so y_hat does not
        depend on ``rc``; with rc=False no host coding happens at all."""
        if rc and self.gaussian_conditional._quantized_cdf.numel() == 0:
            rc = False                                 # tables not built: y_hat is the same without the bytes
        return self.compress(x, quality=quality, mask_pol="point-based-std", real_compress=rc,
                             checkpoint_rep=y_check)["y_hat"]


# ----------------------------------------------------------------------------- the fused plan
# Largest tensor of a plan: the 192-channel feature map at half resolution (and the 576-channel qkv at quarter
# resolution, smaller).  The kernels address a tensor with 32-bit BYTE offsets, so one plan holds at most
# 2^31 / (192 ch * 4 B) / (1/4) pixels of input; larger batches run as several plans over sub-batches.
MAX_PLAN_PIXELS = int((2 ** 31 - 2 ** 20) // (192 * 4) * 4)


def _max_images_per_plan(x) -> int:
    return max(1, MAX_PLAN_PIXELS // (x.shape[2] * x.shape[3]))


def _cat_outputs(outs):
    """Concatenate per-sub-batch result dicts along the batch dimension (every image is an independent unit)."""
    def cat(vals):
        v0 = vals[0]
        if isinstance(v0, dict):
            return {k: cat([v[k] for v in vals]) for k in v0}
        if isinstance(v0, (list, tuple)):
            return type(v0)(cat([v[i] for v in vals]) for i in range(len(v0)))
        if torch.is_tensor(v0):
            if v0.dim() == 2 and v0.dtype == torch.float64:          # log2_likelihood_sum [2, B]
                return torch.cat(vals, dim=1)
            return torch.cat(vals, dim=0)
        return v0
    return cat(outs)


def _slice_stack_heads(plan, m, means_h, scales_h, which):
    """Hyperprior part of the first layer of every slice stack (engine.lower_stack_heads): base stacks read the first
    ``d`` channels of the hyper tensors, progressive ones the second (pic.py:528-529,598-599); the LRP stacks share the
    MEAN support of their slice (pic.py:548,635).  ``which`` = "base" or "prog".  Encoder and decoder plans both call
    this, so both associate the first-layer sum the same way."""
    d, ns = m.division_dimension[0], m.ns0
    stacks, hyp, sup = [], [], []
    if which == "base":
        mh0, sh0 = means_h.window(0, d), scales_h.window(0, d)
        for i in range(ns):
            stacks += [m.cc_mean_transforms[i], m.cc_scale_transforms[i], m.lrp_transforms[i]]
            hyp += [mh0, sh0, mh0]
            sup += [i > 0, i > 0, True]
    else:
        mh1, sh1 = means_h.window(d, d), scales_h.window(d, d)
        for j in range(ns):
            stacks += [m.cc_mean_transforms_prog[j], m.cc_scale_transforms_prog[j], m.lrp_transforms_prog[j]]
            hyp += [mh1, sh1, mh1]
            sup += [True, True, True]
    return E.lower_stack_heads(plan, stacks, hyp, sup)


def _lower_hyper_synthesis(plan, m, z_hat, base_only):
    """compute_hyperprior's synthesis half (pic.py:285-298): with multiple_hyperprior two (mean, scale) pairs of d
    channels each — only the first at quality 0 —, otherwise ONE pair with M channels whose halves feed the base and
    the progressive stacks.  Returns (means_h, scales_h)."""
    d = m.division_dimension[0]
    B, hz, wz = z_hat.B, z_hat.H, z_hat.W
    if m.multiple_hyperprior:
        nh = 1 if base_only else 2
        means_h, scales_h = plan.buf(B, 4 * hz, 4 * wz, nh * d), plan.buf(B, 4 * hz, 4 * wz, nh * d)
        E.lower_stacks(plan, [m.h_mean_s[k] for k in range(nh)] + [m.h_scale_s[k] for k in range(nh)], [[z_hat]] * (2 * nh),
                       [means_h.window(k * d, d) for k in range(nh)] + [scales_h.window(k * d, d) for k in range(nh)])
    else:
        means_h, scales_h = plan.buf(B, 4 * hz, 4 * wz, m.M), plan.buf(B, 4 * hz, 4 * wz, m.M)
        E.lower_stacks(plan, [m.h_mean_s, m.h_scale_s], [[z_hat]] * 2, [means_h, scales_h])
    return means_h, scales_h


def _version_sig(mod: nn.Module):
    return tuple(p._version for p in mod.parameters())


class _RemTrainFn(torch.autograd.Function):
    """likelihoods["y"] of the training-mode REM forward as a differentiable function of the REM parameters
    (the only trainable ones under ``--training_type rems``, train.py:223-226)."""

    @staticmethod
    def forward(ctx, plan, lik_y, use_graph, *params):
        ctx.plan, ctx.use_graph = plan, use_graph
        ctx.generation = plan.generation      # the tape lives in the plan's buffers: it belongs to ONE execute()
        return lik_y.clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.plan.generation != ctx.generation:
            raise RuntimeError(
                "the training plan for this shape ran again before this backward(): its tape (activations, noise, "
                "mask) now belongs to the later forward.  Call loss.backward() before the next training forward of "
                "the same shape (gradient accumulation: backward after every forward).")
        grads = ctx.plan.backward(g, ctx.use_graph)
        return (None, None, None) + tuple(gr if need else None for gr, need in zip(grads, ctx.needs_input_grad[3:]))


class _GsTrainFn(torch.autograd.Function):
    """x_hat of the training-mode forward as a differentiable function of the synthesis transform's parameters (the only
    trainable ones under ``--training_type refine_gs``, train.py:216-218)."""

    @staticmethod
    def forward(ctx, plan, x_hat, use_graph, *params):
        ctx.plan, ctx.use_graph, ctx.generation = plan, use_graph, plan.generation
        return x_hat.clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.plan.generation != ctx.generation:
            raise RuntimeError("the training plan for this shape ran again before this backward(): its tape now belongs to "
                               "the later forward — call loss.backward() before the next training forward of this shape")
        grads = ctx.plan.backward_gs(g.contiguous(), ctx.use_graph)
        return (None, None, None) + tuple(gr if need else None for gr, need in zip(grads, ctx.needs_input_grad[3:]))


class _FullTrainFn(torch.autograd.Function):
    """(x_hat, likelihoods y, likelihoods z) of the complete training plan as differentiable functions of every parameter
    on the path.  backward() runs the plan's backward (and, when the model carries a ``grad_reducer``, the bucketed
    gradient exchange of a multi-GPU job while it runs) and hands each parameter its slice of the flat buffer."""

    @staticmethod
    def forward(ctx, plan, use_graph, reducer, x_hat, lik, z_lik, *params):
        ctx.plan, ctx.use_graph, ctx.reducer, ctx.generation = plan, use_graph, reducer, plan.generation
        return x_hat, lik, z_lik

    @staticmethod
    def backward(ctx, g_xhat, g_lik, g_z):
        if ctx.plan.generation != ctx.generation:
            raise RuntimeError("the training plan for this shape ran again before this backward(): its tape now belongs to "
                               "the later forward — call loss.backward() before the next training forward of this shape")
        plan = ctx.plan
        plan.backward(g_xhat, g_lik, g_z, ctx.use_graph, ctx.reducer)
        need = ctx.needs_input_grad[6:]
        # ONE copy of the flat gradient buffer (the plan overwrites its own at the next backward); every parameter's
        # gradient is a view of the copy, so the clip can run as one reduction over it (finetune.clip_grad_norm_)
        flat = plan.flat.clone()
        plan.handout = (flat, sum(1 for n in need if n), all(need))
        return (None,) * 6 + tuple(flat[o:o + p.numel()].view(p.shape) if n else None
                                   for o, p, n in zip(plan.offsets, plan.params, need))


class _FsqPlan:
    """``forward_single_quality`` for one (B,H,W) lowered to libvampic launches."""

    def __init__(self, m: VarianceMaskingPIC, B, H, W, base_only, rem_idx, device, symbols=False, train=False,
                 own_ck=False, train_gs=False, train_lrp=False):
        self.m, self.B, self.H, self.W = m, B, H, W
        self.train_gs = train_gs    # the synthesis transform in use is being trained (refine_gs): taped g_s + backward plan
        self.train_lrp = train_lrp  # ... and the progressive LRP stacks with it (refine_gs --lrp)
        self.own_ck, self.ck_pr = own_ck, 0.0     # fine-tune: derive the checkpoint latent inside this plan
        self.base_only, self.rem_idx = base_only, rem_idx
        self.symbols = symbols
        self.train = train          # additive-noise likelihoods (+ taped REM and a backward plan when rem_idx is set)
        self.bwd = None
        self.generation = 0         # bumped by every execute(): which forward the training tape belongs to
        self.pr = 0.0
        self.graphs: Dict[float, ops.Graph] = {}
        self.stream = None
        plan = self.plan = E.Plan(device)
        h, w = H // 16, W // 16
        d = m.division_dimension[0]
        ns = m.ns0
        f32 = dict(dtype=torch.float32, device=device)
        self.x_in = torch.empty((B, 3, H, W), **f32)
        self.x_hat = torch.empty((B, 3, H, W), **f32)
        self.log2sum = torch.zeros((2, B), dtype=torch.float64, device=device)   # [y, z] per image
        plan.keep += [self.x_in, self.x_hat, self.log2sum]
        ls_y, ls_z = self.log2sum[0], self.log2sum[1]
        plan.call(lambda: ops.memset_zero(self.log2sum))

        # ---- analysis transforms (both encoders in lockstep)                      pic.py:506-508
        x_s2d = plan.buf(B, H // 2, W // 2, 16)
        plan.call(lambda: L.check(L.load().vam_s2d_input(self.x_in.data_ptr(), x_s2d.ptr, B, H, W, ops.stream_ptr()),
                                  "vam_s2d_input"))
        y = self.y = plan.buf(B, h, w, 2 * d)
        plan.set_class("g_a")
        act16 = getattr(m, "storage", "fp32") == "bf16"
        plan.act16 = act16
        if m.multiple_encoder:
            E.lower_g_a(plan, [m.g_a[0], m.g_a[1]], x_s2d, [y.window(0, d), y.window(d, d)])
        else:                                            # one encoder with M output channels (builder.py:56-67)
            E.lower_g_a(plan, [m.g_a], x_s2d, [y])
        plan.act16 = False

        # ---- hyperprior                                                            pic.py:278-298
        z = plan.buf(B, h // 4, w // 4, m.N)
        plan.set_class("hyperprior")
        self.z = z
        E.lower_stacks(plan, [m.h_a], [[y]], [z])
        self.z_hat = plan.buf(B, h // 4, w // 4, m.N)
        self.z_lik = plan.buf(B, h // 4, w // 4, m.N)
        self.z_sym = ops.new_iview(B, h // 4, w // 4, m.N, device) if symbols else None
        plan.keep.append(self.z_sym)
        self.noise_z = plan.buf(B, h // 4, w // 4, m.N) if train else None
        self.noise_y = plan.buf(B, h, w, d if base_only else 2 * d) if train else None
        plan.call(lambda: ops.eb_forward(z, m.entropy_bottleneck.packed_params(), self.z_hat, self.z_lik, ls_z, sym=self.z_sym,
                                         noise=self.noise_z))
        means_h, scales_h = _lower_hyper_synthesis(plan, m, self.z_hat, base_only)
        self.means_h, self.scales_h = means_h, scales_h

        # ---- base slices                                                           pic.py:522-554
        C = m.dim_chunk
        yq = plan.buf(B, h, w, d)                      # round(y-mu)+mu before the LRP correction
        yb = self.y_base = plan.buf(B, h, w, d)        # base y_hat (after LRP)
        self.mu_b = plan.buf(B, h, w, d)
        self.std_b = plan.buf(B, h, w, d)
        self.lik = plan.buf(B, h, w, d if base_only else 2 * d)
        # entropy-coder inputs (compress only): quantised symbols and scale-table indexes
        self.sym = ops.new_iview(B, h, w, d if base_only else 2 * d, device) if symbols else None
        table = m.gaussian_conditional.scale_table
        indexes = symbols and table.numel() > 0        # without update() only real_compress=False is possible
        self.idx = ops.new_iview(B, h, w, d if base_only else 2 * d, device) if indexes else None
        plan.keep += [self.sym, self.idx]
        sl = lambda v, i, n=1: v.window(i * C, n * C)
        mh0, sh0 = means_h.window(0, d), scales_h.window(0, d)
        hyper_done = plan.record() if not base_only else None
        plan.set_class("stack_heads")
        heads = _slice_stack_heads(plan, m, means_h, scales_h, "base")     # hyperprior part of every first layer, up front
        plan.set_class("slice_chain")

        def base_group(idx: List[int]):
            sup = [sl(yb, 0, min(m.max_support_slices, idx[0]))] if idx[0] > 0 else []
            E.lower_stacks(plan, [m.cc_mean_transforms[i] for i in idx] + [m.cc_scale_transforms[i] for i in idx],
                           [sup] * (2 * len(idx)),
                           [sl(self.mu_b, i) for i in idx] + [sl(self.std_b, i) for i in idx], heads=heads)
            i0, n = idx[0], len(idx)
            plan.call(lambda: ops.gauss_tail(sl(y, i0, n), sl(self.mu_b, i0, n), sl(self.std_b, i0, n),
                                             yhat=sl(yq, i0, n), lik=sl(self.lik, i0, n), log2sum=ls_y,
                                             sym=sl(self.sym, i0, n) if symbols else None))
            if train:        # quantize "noise": likelihood at y + U(-.5,.5) - mu (entropy_models.py:132-138,643-651)
                plan.call(lambda: ops.gauss_train(sl(y, i0, n), sl(self.mu_b, i0, n), sl(self.std_b, i0, n),
                                                  sl(self.noise_y, i0, n), lik=sl(self.lik, i0, n)))
            if indexes:                                                               # pic.py:737
                plan.call(lambda: ops.build_indexes(sl(self.std_b, i0, n), table, out=sl(self.idx, i0, n)))
            E.lower_stacks(plan, [m.lrp_transforms[i] for i in idx], [sup + [sl(yq, i)] for i in idx],
                           [sl(yb, i) for i in idx],
                           [dict(act=L.ACT_HALF_TANH, post=sl(yq, i)) for i in idx], heads=heads)

        base_done = {}                                   # slice -> event "its y_hat_base is final"
        for i in range(min(ns, m.max_support_slices)):
            base_group([i])
            if not base_only:
                base_done[i] = plan.record()
        if ns > m.max_support_slices:
            base_group(list(range(m.max_support_slices, ns)))   # slices 5..9 only see slices 0..4
            if not base_only:
                ev = plan.record()
                for i in range(m.max_support_slices, ns):
                    base_done[i] = ev

        if base_only:
            if not symbols:                              # compress() does not decode (pic.py:671-860)
                plan.set_class("g_s")
                if train_gs:
                    self._lower_g_s_train(plan, m.g_s[0] if m.multiple_decoder else m.g_s, yb)
                else:
                    plan.act16 = act16
                    E.lower_g_s(plan, [m.g_s[0] if m.multiple_decoder else m.g_s], [yb], [self.x_hat])
                    plan.act16 = False
            return

        # ---- progressive slices                                                    pic.py:577-643
        self.mu_p = plan.buf(B, h, w, d)
        self.std_p = plan.buf(B, h, w, d)
        # support vector of the mean chain: mu + y_hat_base with total_mu_rep (pic.py:601), else mu itself
        mu_tot = plan.buf(B, h, w, d) if m.total_mu_rep else self.mu_p
        sp = m.support_progressive_slices
        mu_std = getattr(m, "mu_std", True)
        y_top = y.window(d, d)
        y_sub = y.window(0, d) if m.delta_encode else None                        # pic.py:583-584
        yp = self.y_prog = plan.buf(B, h, w, d)
        g_s = m.g_s[1] if m.multiple_decoder else m.g_s
        if train and not m.all_scalable:
            raise NotImplementedError("training-mode plans are built for all_scalable=True (README config)")

        def supports(j):
            """determine_support (pic.py:264-270): base slice j + the last min(sp, j) entries of the support vectors
            (mu_total / std_total with all_scalable, the decoded progressive slices otherwise, pic.py:586-587)."""
            s = min(sp, j)
            sm, ss_ = (mu_tot, self.std_p) if m.all_scalable else (yp, yp)
            return ([sl(yb, j)] + ([sl(sm, j - s, s)] if s else []), [sl(yb, j)] + ([sl(ss_, j - s, s)] if s else []))

        if not m.all_scalable:
            self._lower_prog_sequential(plan, heads, means_h, scales_h, hyper_done, supports, y_top, y_sub, yb, yp, ls_y,
                                        table if indexes else None, symbols)
            if not symbols:
                plan.set_class("g_s")
                E.lower_g_s(plan, [g_s], [yp], [self.x_hat])
            return
        msups, ssups = [], []
        # With all_scalable the progressive mu/sigma chain only needs y_hat_base[j] and its own history
        # (pic.py:586-612), so it runs on a second HIP stream concurrently with base slices > j.
        plan.branch(1)
        plan.wait(hyper_done)
        plan.set_class("stack_heads")
        heads.update(_slice_stack_heads(plan, m, means_h, scales_h, "prog"))   # on the chain's stream, beside base slice 0
        plan.set_class("slice_chain")
        for j in range(ns):
            plan.wait(base_done[j])
            ms, ss = supports(j)                                               # the hyperprior part is in `heads`
            msups.append(ms)
            ssups.append(ss)
            E.lower_stacks(plan, [m.cc_mean_transforms_prog[j], m.cc_scale_transforms_prog[j]], [ms, ss],
                           [sl(self.mu_p, j), sl(self.std_p, j)], heads=heads)
            if m.total_mu_rep:
                plan.call(lambda j=j: ops.add(sl(self.mu_p, j), sl(yb, j), sl(mu_tot, j)))   # pic.py:601
        chain_done = plan.record()
        plan.branch(0)
        plan.wait(chain_done)

        mu_f, std_f = self.mu_p, self.std_p
        if rem_idx is not None:                                                       # rem_pic.py:363-377
            plan.set_class("rem")
            self.ck = plan.buf(B, h, w, d)
            if own_ck:
                # y_hat at the check level from the SAME front end (everything up to here is quality independent;
                # at q <= check_levels[0] no REM applies): what ExtractChekpointRepr(x, q_ref) returns, pic.py:621-641
                m_ck, rq_ck, junk = plan.buf(B, h, w, d), plan.buf(B, h, w, d), plan.buf(B, h, w, d)
                plan.call(lambda: ops.variance_mask(self.std_p, self.ck_pr, m_ck, n_slice=ns))
                plan.call(lambda: ops.gauss_tail(y_top, self.mu_p, self.std_p, y2=y_sub, mask=m_ck, yhat=rq_ck, lik=junk))
                E.lower_stacks(plan, [m.lrp_transforms_prog[j] for j in range(ns)],
                               [msups[j] + [sl(rq_ck, j)] for j in range(ns)], [sl(self.ck, j) for j in range(ns)],
                               [dict(act=L.ACT_HALF_TANH, post=sl(rq_ck, j), post2=sl(yb, j)) for j in range(ns)], heads=heads)
            att = self.att = plan.buf(B, h, w, d)
            plan.call(lambda: ops.variance_mask(self.std_p, self.pr, att, n_slice=ns))
            std_f = plan.buf(B, h, w, d)
            mu_f = plan.buf(B, h, w, d) if mu_std else self.mu_p      # without mu_std only sigma is refined (rem_pic.py:214-218)
            mods = [m.post_latent[rem_idx][j] for j in range(ns)]
            rem_io = ([sl(self.ck, j) for j in range(ns)],
                      [[sl(self.mu_b, j), sl(self.std_b, j)] for j in range(ns)],
                      [([sl(self.mu_p, j)] if mu_std else []) + [sl(self.std_p, j)] for j in range(ns)],
                      [sl(att, j) for j in range(ns)],
                      [([sl(mu_f, j)] if mu_std else []) + [sl(std_f, j)] for j in range(ns)])
            if train:
                self.rem_params = [p for mod in mods for p in mod.parameters()]
                self.packs = E.TrainPacks(*E.rem_trained_convs(mods))
                self.packs.record_refresh(plan)
                tape = E.lower_rem_blocks_train(plan, mods, *rem_io, self.packs)
            else:
                self.rem_sig = _version_sig(m.post_latent[rem_idx])
                E.lower_rem_blocks(plan, mods, *rem_io)
        self.mu_f, self.std_f = mu_f, std_f
        plan.set_class("lrp_prog")
        self.mask = plan.buf(B, h, w, d)
        self.thr = torch.empty((B * ns,), **f32)
        plan.keep.append(self.thr)
        plan.call(lambda: ops.variance_mask(std_f, self.pr, self.mask, n_slice=ns, thr=self.thr))   # pic.py:621-622
        rq = plan.buf(B, h, w, d)
        plan.call(lambda: ops.gauss_tail(y_top, mu_f, std_f, y2=y_sub, mask=self.mask, yhat=rq,
                                         lik=self.lik.window(d, d), log2sum=ls_y,
                                         sym=self.sym.window(d, d) if symbols else None))           # pic.py:625-629
        if train:
            yr, y0, nz = y_top, y_sub, self.noise_y.window(d, d)
            plan.call(lambda: ops.gauss_train(yr, mu_f, std_f, nz, y2=y0, mask=self.mask, lik=self.lik.window(d, d)))
            if rem_idx is not None:
                # ---- backward plan: dL/dlik (progressive half) -> (dmu', dsigma') -> REM parameters
                bw = self.bwd = E.Plan(device)
                self.glik = bw.buf(B, h, w, d)
                dmu, dsg = bw.buf(B, h, w, d), bw.buf(B, h, w, d)
                self.dmu, self.dsg, self.rem_io, self.rem_tape = dmu, dsg, rem_io, tape     # kept for teacher-forced gradient checks
                flat = torch.zeros(sum(p.numel() for p in self.rem_params), **f32)
                self.gflat, self.gviews, off = flat, [], 0
                for p in self.rem_params:
                    self.gviews.append(flat[off:off + p.numel()].view(p.shape))
                    off += p.numel()
                grads = {id(p): g for p, g in zip(self.rem_params, self.gviews)}
                bw.keep += [flat, self.gviews]
                bw.call(lambda: ops.gauss_train(yr, mu_f, std_f, nz, y2=y0, mask=self.mask, grad_lik=self.glik,
                                                dmu=dmu, dsigma=dsg), "likelihood backward")
                E.lower_rem_backward(bw, tape, mods, [sl(dmu, j) for j in range(ns)], [sl(dsg, j) for j in range(ns)],
                                     rem_io[3], self.packs, grads)
        if indexes:                                                                   # pic.py:813
            plan.call(lambda: ops.build_indexes(std_f, table, mask=self.mask, out=self.idx.window(d, d)))
        lrp = None
        if train_lrp:
            from . import gs_train as G
            stacks = [m.lrp_transforms_prog[j] for j in range(ns)]
            lpk = [G.TransformPacks(st) for st in stacks]
            for pk_ in lpk:
                pk_.record_refresh(plan)
            mh1 = means_h.window(d, d)
            tapes = G.lower_lrp_stacks_train(plan, stacks, [[mh1] + msups[j] + [sl(rq, j)] for j in range(ns)],
                                             [sl(rq, j) for j in range(ns)], [sl(yb, j) for j in range(ns)],
                                             [sl(yp, j) for j in range(ns)], lpk)
            lrp = dict(tapes=tapes, packs=lpk, params=[p for st in stacks for p in st.parameters()])
        else:
            E.lower_stacks(plan, [m.lrp_transforms_prog[j] for j in range(ns)], [msups[j] + [sl(rq, j)] for j in range(ns)],
                           [sl(yp, j) for j in range(ns)],
                           [dict(act=L.ACT_HALF_TANH, post=sl(rq, j), post2=sl(yb, j)) for j in range(ns)], heads=heads)   # :635-641
        if not symbols:
            plan.set_class("g_s")
            if train_gs:
                self._lower_g_s_train(plan, g_s, yp, lrp)
            else:
                plan.act16 = act16
                E.lower_g_s(plan, [g_s], [yp], [self.x_hat])
                plan.act16 = False

    def _lower_g_s_train(self, plan, dec, y_in, lrp=None):
        """refine_gs: taped synthesis transform + its backward plan (gs_train.py); ``lrp`` = the taped progressive LRP
        stacks when they train too (refine_gs --lrp): their gradients come from dL/dy_hat, the input gradient of g_s."""
        from . import gs_train as G
        dev = self.x_in.device
        self.gs_params = list(dec.parameters()) + (lrp["params"] if lrp else [])
        self.gs_packs = G.TransformPacks(dec)
        self.gs_packs.record_refresh(plan)
        tape = G.lower_g_s_train(plan, dec, y_in, self.x_hat, self.gs_packs)
        bw = self.gs_bwd = E.Plan(dev)
        self.g_xhat = torch.zeros_like(self.x_hat)
        offs, tot = [], 0
        for p in self.gs_params:                       # every gradient view 16-byte aligned inside one flat bucket
            offs.append(tot)
            tot += (p.numel() + 3) // 4 * 4
        self.gs_flat = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.gs_views = [self.gs_flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, self.gs_params)]
        grads = {id(p): g for p, g in zip(self.gs_params, self.gs_views)}
        bw.keep += [self.g_xhat, self.gs_flat, self.gs_views]
        d_y = G.lower_g_s_backward(bw, tape, self.x_hat, self.g_xhat, self.gs_packs, grads, need_input_grad=lrp is not None)
        if lrp is not None:
            self.lrp_tapes, self.d_yhat = lrp["tapes"], d_y                    # kept for teacher-forced gradient checks
            G.lower_lrp_stacks_backward(bw, lrp["tapes"], [d_y.window(32 * j, 32) for j in range(len(lrp["tapes"]))],
                                        lrp["packs"], grads)

    def backward_gs(self, grad_x_hat: torch.Tensor, use_graph: bool):
        """dL/d(parameters of the trained synthesis transform) for dL/dx_hat; fresh tensors in ``gs_params`` order."""
        cur = torch.cuda.current_stream(self.x_in.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.g_xhat.copy_(grad_x_hat)
            if use_graph:
                if getattr(self, "_gs_bwd_graph", None) is None:
                    self.gs_bwd.run()
                    self.stream.synchronize()
                    self._gs_bwd_graph = ops.Graph()
                    self._gs_bwd_graph.capture(self.gs_bwd.run)
                self._gs_bwd_graph.launch()
            else:
                self.gs_bwd.run()
            out = [v.clone() for v in self.gs_views]
        cur.wait_stream(self.stream)
        return out

    def _lower_prog_sequential(self, plan, heads, means_h, scales_h, hyper_done, supports, y_top, y_sub, yb, yp, ls_y,
                               table, symbols):
        """all_scalable = False (pic.py:586-587): the (mu, sigma) stacks of progressive slice j read the DECODED
        progressive slices j-sp..j-1, so mask, quantisation and LRP of a slice must finish before the next slice's
        stacks start — one slice at a time, on the caller's stream."""
        m, C, ns = self.m, self.m.dim_chunk, self.m.ns0
        B, h, w, d = self.B, self.H // 16, self.W // 16, self.m.division_dimension[0]
        sl = lambda v, i, n=1: v.window(i * C, n * C)
        mu_std = getattr(m, "mu_std", True)
        rem_idx = self.rem_idx
        plan.set_class("stack_heads")
        heads.update(_slice_stack_heads(plan, m, means_h, scales_h, "prog"))
        plan.set_class("slice_chain")
        self.mask = plan.buf(B, h, w, d)
        self.thr = None                                   # per-slice launches: thresholds are not collected
        rq = plan.buf(B, h, w, d)
        mu_f, std_f = self.mu_p, self.std_p
        if rem_idx is not None:
            self.ck = plan.buf(B, h, w, d)
            att = self.att = plan.buf(B, h, w, d)
            std_f = plan.buf(B, h, w, d)
            mu_f = plan.buf(B, h, w, d) if mu_std else self.mu_p
            self.rem_sig = _version_sig(m.post_latent[rem_idx])
        self.mu_f, self.std_f = mu_f, std_f
        for j in range(ns):
            ms, ss = supports(j)
            E.lower_stacks(plan, [m.cc_mean_transforms_prog[j], m.cc_scale_transforms_prog[j]], [ms, ss],
                           [sl(self.mu_p, j), sl(self.std_p, j)], heads=heads)
            if rem_idx is not None:                                                   # rem_pic.py:363-377
                plan.call(lambda j=j: ops.variance_mask(sl(self.std_p, j), self.pr, sl(att, j), n_slice=1))
                E.lower_rem_blocks(plan, [m.post_latent[rem_idx][j]], [sl(self.ck, j)], [[sl(self.mu_b, j), sl(self.std_b, j)]],
                                   [([sl(self.mu_p, j)] if mu_std else []) + [sl(self.std_p, j)]], [sl(att, j)],
                                   [([sl(mu_f, j)] if mu_std else []) + [sl(std_f, j)]])
            plan.call(lambda j=j: ops.variance_mask(sl(std_f, j), self.pr, sl(self.mask, j), n_slice=1))          # pic.py:621-622
            plan.call(lambda j=j: ops.gauss_tail(sl(y_top, j), sl(mu_f, j), sl(std_f, j), y2=sl(y_sub, j) if y_sub is not None else None,
                                                 mask=sl(self.mask, j), yhat=sl(rq, j), lik=sl(self.lik, ns + j), log2sum=ls_y,
                                                 sym=sl(self.sym, ns + j) if symbols else None))                  # pic.py:625-629
            if table is not None:
                plan.call(lambda j=j: ops.build_indexes(sl(std_f, j), table, mask=sl(self.mask, j), out=sl(self.idx, ns + j)))
            E.lower_stacks(plan, [m.lrp_transforms_prog[j]], [ms + [sl(rq, j)]], [sl(yp, j)],
                           [dict(act=L.ACT_HALF_TANH, post=sl(rq, j), post2=sl(yb, j))], heads=heads)              # pic.py:635-641

    # -------------------------------------------------------------------------------------------
    def close(self):
        """Give up the executable graphs of this plan (ops.Graph.close: destroyed at the next safe point)."""
        for g in list(self.graphs.values()) + [getattr(self, "_bwd_graph", None), getattr(self, "_gs_bwd_graph", None),
                                               getattr(self.plan, "_graph", None)]:
            if g is not None:
                g.close()
        self.graphs.clear()
        self._bwd_graph = self._gs_bwd_graph = None

    def set_noise(self, noise=None):
        """Training: U(-.5,.5) for the likelihood proxies; ``noise`` = {"y": NCHW, "z": NCHW} injects fixed draws
        (parity tests), otherwise torch's generator fills the buffers like the reference's ``uniform_``."""
        for key, v in (("y", self.noise_y), ("z", self.noise_z)):
            if noise is not None and key in noise:
                v.buf.copy_(noise[key].to(v.buf.device).permute(0, 2, 3, 1))
            else:
                v.buf.uniform_(-0.5, 0.5)

    def backward(self, grad_lik_y: torch.Tensor, use_graph: bool):
        """dL/d(REM parameters) for dL/dlikelihoods["y"] (NCHW); returns fresh tensors in ``rem_params`` order."""
        d = self.m.division_dimension[0]
        cur = torch.cuda.current_stream(self.x_in.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.glik.buf.copy_(grad_lik_y[:, d:].permute(0, 2, 3, 1))
            if use_graph:
                if getattr(self, "_bwd_graph", None) is None:
                    self.bwd.run()
                    self.stream.synchronize()
                    self._bwd_graph = ops.Graph()
                    self._bwd_graph.capture(self.bwd.run)
                self._bwd_graph.launch()
            else:
                self.bwd.run()
            flat = self.gflat.clone()
        cur.wait_stream(self.stream)
        out, off = [], 0
        for p in self.rem_params:
            out.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return out

    def execute(self, x, pr, checkpoint_ref, use_graph, clone, noise=None, ck_pr=None):
        """Run the plan on the model's own HIP stream (hipGraph capture is not allowed on the
        legacy default stream), ordered after / before the caller's current stream."""
        self.pr = float(pr)
        self.generation += 1
        self.ck_pr = float(ck_pr) if ck_pr is not None else 0.0
        if self.train_gs:
            sig = tuple(p.data_ptr() for p in self.gs_params)
            if getattr(self, "_gs_ptr_sig", sig) != sig:       # parameter storage replaced: captured pointers are stale
                self.close()
            self._gs_ptr_sig = sig
        if self.train and self.rem_idx is not None:
            sig = tuple(p.data_ptr() for p in self.rem_params)
            if getattr(self, "_ptr_sig", sig) != sig:          # parameter storage replaced: captured pointers are stale
                self.close()
            self._ptr_sig = sig
        ops.drain_graveyard()                          # dropped plans' graphs: destroyed here, outside any capture
        cur = torch.cuda.current_stream(self.x_in.device)
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.x_in.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.x_in.copy_(x)
            if self.train:
                self.set_noise(noise)
            if checkpoint_ref is not None:
                ck = ops.from_nchw(checkpoint_ref.to(self.x_in.device))
                self.ck.buf.copy_(ck.buf[..., ck.c0:ck.c0 + ck.C])
            if use_graph:
                g = self.graphs.get((self.pr, self.ck_pr))
                if g is None:
                    self.plan.run()                      # warm-up: every code object loaded before capture
                    self.stream.synchronize()
                    g = ops.Graph()
                    g.capture(self.plan.run)
                    if len(self.graphs) > 32:
                        for g_ in self.graphs.values():
                            g_.close()
                        self.graphs.clear()
                    self.graphs[(self.pr, self.ck_pr)] = g
                g.launch()
            else:
                self.plan.run()
        cur.wait_stream(self.stream)
        fin = (lambda t: t.clone()) if clone else (lambda t: t)
        nchw = lambda v: fin(v.torch_nchw())
        out = {"x_hat": fin(self.x_hat),
               "likelihoods": {"y": nchw(self.lik), "z": nchw(self.z_lik)},
               "log2_likelihood_sum": fin(self.log2sum)}
        if self.base_only:
            yh = nchw(self.y_base)
            out.update({"y_hat": yh, "y_base": yh, "y_prog": yh, "mu": nchw(self.mu_b), "std": nchw(self.std_b),
                        "mu_base": nchw(self.mu_b), "std_base": nchw(self.std_b), "mu_prog": [], "std_prog": []})
        else:
            yh = nchw(self.y_prog)
            out.update({"y_hat": yh, "y_base": nchw(self.y_base), "y_prog": yh, "mu_base": nchw(self.mu_b),
                        "mu": nchw(self.mu_f), "std_base": nchw(self.std_b), "std": nchw(self.std_f),
                        "mask": nchw(self.mask)})
        return out


class _DecPlan:
    """``decompress`` for one (B, z-shape): the same kernels as the encoder's plan, cut where the
    host rANS decoder has to deliver the symbols of a slice (models/pic.py:862-960).  The conv
    kernel's K order is canonical, so mu / sigma / masks / indexes are bit-identical to the
    encoder's although the launches are grouped differently."""

    def __init__(self, m: VarianceMaskingPIC, B, hz, wz, base_only, rem_idx, device):
        self.m, self.B, self.base_only, self.rem_idx = m, B, base_only, rem_idx
        self.device = torch.device(device)
        self.pr = 0.0
        self.stream = None
        h, w = hz * 4, wz * 4
        self.h, self.w, self.hz, self.wz = h, w, hz, wz
        d, C, ns = m.division_dimension[0], m.dim_chunk, m.ns0
        f32 = dict(dtype=torch.float32, device=device)
        table = m.gaussian_conditional.scale_table
        if table.numel() == 0:
            raise ValueError("empty scale table: call model.update() before decompress()")
        nv, ni = (lambda c, hh=h, ww=w: ops.new_view(B, hh, ww, c, device)), (lambda c, hh=h, ww=w: ops.new_iview(B, hh, ww, c, device))
        sl = lambda v, i, n=1: v.window(i * C, n * C)
        self.x_hat = torch.empty((B, 3, h * 16, w * 16), **f32)
        # ---- z
        self.z_sym = ni(m.N, hz, wz)
        med = nv(m.N, hz, wz)
        med.buf.copy_(m.entropy_bottleneck._get_medians().detach().reshape(1, 1, 1, -1).expand_as(med.buf))
        z_hat = nv(m.N, hz, wz)
        P = self.p_hyper = E.Plan(device)
        P.call(lambda: ops.dequantize(self.z_sym, med, z_hat))                       # entropy_models.py:520-525
        means_h, scales_h = _lower_hyper_synthesis(P, m, z_hat, base_only)
        heads = _slice_stack_heads(P, m, means_h, scales_h, "base")                 # same association as the encoder's plan
        if not base_only:
            heads.update(_slice_stack_heads(P, m, means_h, scales_h, "prog"))
        # ---- base slices
        yq, yb, mu_b, std_b = nv(d), nv(d), nv(d), nv(d)
        self.idx_b, self.sym_b = ni(d), ni(d)
        self.mu_b, self.std_b = mu_b, std_b
        self.p_base = []
        for i in range(ns):
            sup = [sl(yb, 0, min(m.max_support_slices, i))] if i > 0 else []
            Pa, Pb = E.Plan(device), E.Plan(device)
            E.lower_stacks(Pa, [m.cc_mean_transforms[i], m.cc_scale_transforms[i]], [sup, sup],
                           [sl(mu_b, i), sl(std_b, i)], heads=heads)
            Pa.call(lambda i=i: ops.build_indexes(sl(std_b, i), table, out=sl(self.idx_b, i)))          # pic.py:879
            Pb.call(lambda i=i: ops.dequantize(sl(self.sym_b, i), sl(mu_b, i), sl(yq, i)))               # pic.py:884
            E.lower_stacks(Pb, [m.lrp_transforms[i]], [sup + [sl(yq, i)]], [sl(yb, i)],
                           [dict(act=L.ACT_HALF_TANH, post=sl(yq, i))], heads=heads)
            self.p_base.append((Pa, Pb))
        self.p_syn = E.Plan(device)
        if base_only:
            E.lower_g_s(self.p_syn, [m.g_s[0] if m.multiple_decoder else m.g_s], [yb], [self.x_hat])
            return
        # ---- progressive slices
        mu_p, std_p, mask, rq, yp = nv(d), nv(d), nv(d), nv(d), nv(d)
        mu_tot = nv(d) if m.total_mu_rep else mu_p                                    # pic.py:601
        mu_std = getattr(m, "mu_std", True)
        std_f = nv(d) if rem_idx is not None else std_p
        mu_f = nv(d) if (rem_idx is not None and mu_std) else mu_p
        self.ck = nv(d) if rem_idx is not None else None
        att = nv(d) if rem_idx is not None else None
        self.idx_p, self.sym_p = ni(d), ni(d)
        sp = m.support_progressive_slices
        self.p_prog = []
        for j in range(ns):
            s_ = min(sp, j)
            sm, ss_v = (mu_tot, std_p) if m.all_scalable else (yp, yp)                # pic.py:586-587
            ms = [sl(yb, j)] + ([sl(sm, j - s_, s_)] if s_ else [])
            ss = [sl(yb, j)] + ([sl(ss_v, j - s_, s_)] if s_ else [])
            Pa, Pb = E.Plan(device), E.Plan(device)
            E.lower_stacks(Pa, [m.cc_mean_transforms_prog[j], m.cc_scale_transforms_prog[j]], [ms, ss], [sl(mu_p, j), sl(std_p, j)],
                           heads=heads)
            if m.total_mu_rep:
                Pa.call(lambda j=j: ops.add(sl(mu_p, j), sl(yb, j), sl(mu_tot, j)))
            if rem_idx is not None:
                Pa.call(lambda j=j: ops.variance_mask(sl(std_p, j), self.pr, sl(att, j), n_slice=1))
                E.lower_rem_blocks(Pa, [m.post_latent[rem_idx][j]], [sl(self.ck, j)], [[sl(mu_b, j), sl(std_b, j)]],
                                   [([sl(mu_p, j)] if mu_std else []) + [sl(std_p, j)]], [sl(att, j)],
                                   [([sl(mu_f, j)] if mu_std else []) + [sl(std_f, j)]])
            Pa.call(lambda j=j: ops.variance_mask(sl(std_f, j), self.pr, sl(mask, j), n_slice=1))       # pic.py:942
            Pa.call(lambda j=j: ops.build_indexes(sl(std_f, j), table, mask=sl(mask, j), out=sl(self.idx_p, j)))  # :945
            Pb.call(lambda j=j: ops.dequantize(sl(self.sym_p, j), sl(mu_f, j), sl(rq, j)))               # :948
            E.lower_stacks(Pb, [m.lrp_transforms_prog[j]], [ms + [sl(rq, j)]], [sl(yp, j)],
                           [dict(act=L.ACT_HALF_TANH, post=sl(rq, j), post2=sl(yb, j))], heads=heads)
            self.p_prog.append((Pa, Pb))
        E.lower_g_s(self.p_syn, [m.g_s[1] if m.multiple_decoder else m.g_s], [yp], [self.x_hat])

    def _decode_slice(self, strings, idx_view: ops.IView, sym_view: ops.IView, tables, C):
        """indexes GPU -> host, rANS decode per image, symbols host -> GPU (NHWC window)."""
        from . import bitstream as bs
        B, h, w = self.B, idx_view.buf.shape[1], idx_view.buf.shape[2]
        self.stream.synchronize()
        idx = idx_view.buf[..., idx_view.c0:idx_view.c0 + C].cpu().numpy()          # [B,h,w,C]
        out = np.empty((B, h, w, C), dtype=np.int32)
        for b in range(B):
            dec = bs.decode(strings[b], idx[b].transpose(2, 0, 1), tables)           # stream order [C,h,w]
            out[b] = dec.reshape(C, h, w).transpose(1, 2, 0)
        sym_view.buf[..., sym_view.c0:sym_view.c0 + C].copy_(torch.from_numpy(out).to(self.device))

    def decode(self, strings, pr, checkpoint_rep):
        from . import bitstream as bs
        m = self.m
        self.pr = float(pr)
        y_strings, z_strings = strings[0], strings[1]
        n_need = m.ns0 if self.base_only else m.ns1
        if len(y_strings) < n_need or len(z_strings) != self.B:
            raise ValueError(f"expected {n_need} slice streams x {self.B} images, got {len(y_strings)} x {len(z_strings)}")
        tg, te = bs.Tables.of(m.gaussian_conditional), bs.Tables.of(m.entropy_bottleneck)
        cur = torch.cuda.current_stream(self.device)
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.device)
        self.stream.wait_stream(cur)
        C = m.dim_chunk
        with torch.cuda.stream(self.stream):
            if checkpoint_rep is not None:
                ck = ops.from_nchw(checkpoint_rep.to(self.device))
                self.ck.buf.copy_(ck.buf[..., ck.c0:ck.c0 + ck.C])
            zi = np.broadcast_to(np.arange(m.N, dtype=np.int32)[:, None, None], (m.N, self.hz, self.wz))
            zs = np.stack([bs.decode(z_strings[b], zi, te).reshape(m.N, self.hz, self.wz).transpose(1, 2, 0)
                           for b in range(self.B)])
            self.z_sym.buf.copy_(torch.from_numpy(zs).to(self.device))
            self.p_hyper.run()
            for i, (Pa, Pb) in enumerate(self.p_base):
                Pa.run()
                self._decode_slice(y_strings[i], self.idx_b.window(i * C, C), self.sym_b.window(i * C, C), tg, C)
                Pb.run()
            if not self.base_only:
                for j, (Pa, Pb) in enumerate(self.p_prog):
                    Pa.run()
                    self._decode_slice(y_strings[m.ns0 + j], self.idx_p.window(j * C, C), self.sym_p.window(j * C, C), tg, C)
                    Pb.run()
            self.p_syn.run()
        cur.wait_stream(self.stream)
        return self.x_hat.clone()


models = {"pic": VarianceMaskingPIC, "rem": VarianceMaskingPICREM}


def get_model(args, device):
    """models/__init__.py:11-55 (``cnn`` = the legacy WACNN baseline, out of scope: SURVEY §2 #11)."""
    common = dict(N=args.N, M=args.M, multiple_decoder=args.multiple_decoder, multiple_encoder=args.multiple_encoder,
                  multiple_hyperprior=args.multiple_hyperprior, dim_chunk=args.dim_chunk,
                  division_dimension=args.division_dimension, mask_policy=args.mask_policy,
                  support_progressive_slices=args.support_progressive_slices, delta_encode=args.delta_encode,
                  total_mu_rep=args.total_mu_rep, all_scalable=args.all_scalable)
    if args.model == "pic":
        net = VarianceMaskingPIC(**common)
    elif args.model == "rem":
        net = VarianceMaskingPICREM(**common, check_levels=args.check_levels, mu_std=args.mu_std,
                                    dimension=args.dimension)
    else:
        raise NotImplementedError
    return net.to(device)
