"""Host-side mirror of the reference's layer modules (``src/layers/*.py``).

These classes keep the reference's attribute names, constructor arguments and
``state_dict`` keys so its checkpoints load unchanged; they own parameters only.
All arithmetic is issued through :mod:`.engine` (libvampic kernels) — calling a
module's ``forward`` builds a small plan for that module alone and runs it, so
``model.g_a[0](x)``, ``model.cc_mean_transforms[i](t)``, ``model.masking(...)`` work as
the reference harness expects (NCHW fp32 in, NCHW fp32 out).

Factory names follow reference layers/layers.py: conv, deconv, conv1x1, conv3x3,
subpel_conv3x3, ResidualUnit, Win_noShift_Attention; layers/gdn.py: GDN;
layers/win_attention.py: WindowAttention, WinBasedAttention; layers/rem.py:
ResidualBlock, LatentRateReduction; layers/channel_mask.py: ChannelMask.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops


def _no_autograd(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "this module-level call runs the evaluation kernels only: gradients flow through the model-level training plans "
            "(model(x, quality=[0, q], training=True), forward_single_quality(..., training=True)) and the training-mode "
            "entropy models — call it under torch.no_grad()")


class _Packable(nn.Module):
    """Caches the kernel-layout copy of the parameters; invalidated when they change."""

    def _key(self):
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in self.parameters(recurse=False))

    def packed(self, bf16: bool = False):
        """Kernel-layout weights; ``bf16=True``: rounded to bf16 for the bf16-storage kernel (BASELINE configs[2])."""
        key = self._key()
        if bf16:
            if getattr(self, "_pk16_key", None) != key:
                object.__setattr__(self, "_pk16", self._pack(True))
                object.__setattr__(self, "_pk16_key", key)
            return self._pk16
        if getattr(self, "_pk_key", None) != key:
            object.__setattr__(self, "_pk", self._pack())
            object.__setattr__(self, "_pk_key", key)
        return self._pk

    def _pack(self, bf16: bool = False):
        raise NotImplementedError


class Conv2d(_Packable):
    """nn.Conv2d(in, out, k, stride, padding=k//2) parameters (layers/layers.py:5-12)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = kernel_size, stride
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.kaiming_normal_(self.weight)          # models/base.py:31-36

    def _pack(self, bf16: bool = False):
        if self.in_channels == 3 and self.kernel_size == 5 and self.stride == 2:
            return ops.pack_conv5s2_rgb(self.weight, self.bias, bf16)
        return ops.pack_conv(self.weight, self.bias, self.stride, bf16)

    @property
    def is_rgb_s2d(self):
        return self.in_channels == 3 and self.kernel_size == 5 and self.stride == 2

    def packed_split(self, c_head: int):
        """(head, tail): the layer as  conv(x[:, :c_head]; W[:, :c_head], bias) + conv(x[:, c_head:]; W[:, c_head:]).
        ``tail`` is None when the layer has no input channels beyond ``c_head``.  Used to hoist the hyperprior part of
        every slice stack's first layer out of the sequential slice loop (engine.lower_stack_heads)."""
        key = (self._key(), c_head)
        if getattr(self, "_pk_split_key", None) != key:
            w = self.weight.detach()
            head = ops.pack_conv(w[:, :c_head].contiguous(), self.bias, self.stride)
            tail = ops.pack_conv(w[:, c_head:].contiguous(), None, self.stride) if self.in_channels > c_head else None
            object.__setattr__(self, "_pk_split", (head, tail))
            object.__setattr__(self, "_pk_split_key", key)
        return self._pk_split

    def forward(self, x):
        from . import engine
        return engine.run_module(self, x)


class SubpelConv(nn.Sequential):
    """conv3x3(in, out*r^2) -> PixelShuffle(r)  (layers/layers.py:82-86); keys "0.weight","0.bias"."""

    def __init__(self, in_ch: int, out_ch: int, r: int = 2):
        assert r == 2
        super().__init__(Conv2d(in_ch, out_ch * r * r, 3), PixelShuffle(r))
        self.out_ch = out_ch

    def packed(self):
        c = self[0]
        key = c._key()
        if getattr(self, "_pk_key", None) != key:
            object.__setattr__(self, "_pk", ops.pack_subpel(c.weight, c.bias))
            object.__setattr__(self, "_pk_key", key)
        return self._pk

    def forward(self, x):
        from . import engine
        return engine.run_module(self, x)


class ConvTranspose2d(_Packable):
    """nn.ConvTranspose2d(in, out, 5, stride 2, padding 2, output_padding 1) (layers/layers.py:14-22)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 5, stride: int = 2):
        super().__init__()
        assert kernel_size == 5 and stride == 2, "only the reference's k5/s2 transposed conv is built"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, 5, 5))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.kaiming_normal_(self.weight)

    def _pack(self, bf16: bool = False):
        return ops.pack_deconv(self.weight, self.bias, bf16)

    def forward(self, x):
        from . import engine
        return engine.run_module(self, x)


class Linear(_Packable):
    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)

    def _pack(self, bf16: bool = False):
        return ops.pack_linear(self.weight, self.bias, bf16)


class _Marker(nn.Module):
    """Parameter-free placeholder keeping the reference's nn.Sequential indices; the
    operation itself is fused into the neighbouring conv's epilogue."""

    def forward(self, x):
        raise RuntimeError(f"{type(self).__name__} is fused into the preceding convolution; call the enclosing module")


class GELU(_Marker):
    pass


class LeakyReLU(_Marker):
    pass


class PixelShuffle(_Marker):
    def __init__(self, r: int = 2):
        super().__init__()
        self.r = r


def conv(in_channels, out_channels, kernel_size=5, stride=2):
    return Conv2d(in_channels, out_channels, kernel_size, stride)


def deconv(in_channels, out_channels, kernel_size=5, stride=2):
    return ConvTranspose2d(in_channels, out_channels, kernel_size, stride)


def conv1x1(in_ch, out_ch, stride=1):
    assert stride == 1
    return Conv2d(in_ch, out_ch, 1, 1)


def conv3x3(in_ch, out_ch, stride=1):
    return Conv2d(in_ch, out_ch, 3, stride)


def subpel_conv3x3(in_ch, out_ch, r=1):
    return SubpelConv(in_ch, out_ch, r)


class ConvStack(nn.Sequential):
    """nn.Sequential of conv / GELU / subpel layers (models/pic.py:83-164, builder.py:72-135)."""

    def forward(self, x):
        from . import engine
        return engine.run_module(self, x)


# ------------------------------------------------------------------ GDN
class _LowerBound(nn.Module):
    def __init__(self, bound: float):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))


class NonNegativeParametrizer(nn.Module):
    """Buffer layout of compressai's parametrizer (pedestal, lower_bound.bound); the
    arithmetic max(p, bound)^2 - pedestal happens in vam_pack_* (SURVEY A.3)."""

    def __init__(self, minimum: float = 0.0, reparam_offset: float = 2 ** -18):
        super().__init__()
        self.minimum, self.reparam_offset = float(minimum), float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = _LowerBound((self.minimum + pedestal) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))


class GDN(_Packable):
    """layers/gdn.py:26-75."""

    def __init__(self, in_channels: int, inverse: bool = False, beta_min: float = 1e-6, gamma_init: float = 0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.in_channels = in_channels
        assert float(beta_min) == 1e-6, "vam_pack_bias(GDN) is built for the reference's beta_min = 1e-6"
        self.beta_reparam = NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def _pack(self, bf16: bool = False):
        return ops.pack_gdn(self.beta, self.gamma, bf16)

    def forward(self, x):
        from . import engine
        return engine.run_module(self, x)


# ------------------------------------------------------------------ attention
class WindowAttention(nn.Module):
    """layers/win_attention.py:37-115 (parameters + relative_position_index buffer)."""

    def __init__(self, dim=192, window_size=(8, 8), num_heads=8):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        ws = window_size[0]
        assert window_size[0] == window_size[1]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        ys = torch.arange(ws * ws) // ws
        xs = torch.arange(ws * ws) % ws
        idx = (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)
        self.register_buffer("relative_position_index", idx)
        self.qkv = Linear(dim, dim * 3)
        self.proj = Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class WinBasedAttention(nn.Module):
    """layers/win_attention.py:118-207."""

    def __init__(self, dim=192, num_heads=8, window_size=8, shift_size=0):
        super().__init__()
        assert 0 <= shift_size < window_size, "shift_size must in 0-window_size"
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.attn = WindowAttention(dim, (window_size, window_size), num_heads)

    def forward(self, x):
        from . import engine
        return engine.run_module(self, x)


class ResidualUnit(nn.Module):
    """layers/layers.py:30-48."""

    def __init__(self, N: int):
        super().__init__()
        self.conv = nn.Sequential(conv1x1(N, N // 2), GELU(), conv3x3(N // 2, N // 2), GELU(), conv1x1(N // 2, N))
        self.relu = GELU()

    def forward(self, x):
        from . import engine
        return engine.run_module(self, x)


class Win_noShift_Attention(nn.Module):
    """layers/layers.py:50-74 (the builder passes shift_size = window/2)."""

    def __init__(self, dim, num_heads=8, window_size=8, shift_size=0):
        super().__init__()
        N = dim
        self.conv_a = nn.Sequential(ResidualUnit(N), ResidualUnit(N), ResidualUnit(N))
        self.conv_b = nn.Sequential(
            WinBasedAttention(dim=dim, num_heads=num_heads, window_size=window_size, shift_size=shift_size),
            ResidualUnit(N), ResidualUnit(N), ResidualUnit(N), conv1x1(N, N))

    def forward(self, x):
        from . import engine
        return engine.run_module(self, x)


class TransformStack(nn.Sequential):
    """g_a / g_s nn.Sequential (models/builder.py:8-18,43-53)."""

    def forward(self, x):
        from . import engine
        return engine.run_module(self, x)


# ------------------------------------------------------------------ REM
class ResidualBlock(nn.Module):
    """layers/rem.py:37-66."""

    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.nonlin = LeakyReLU()
        self.conv2 = conv3x3(out_ch, out_ch)
        self.skip = conv1x1(in_ch, out_ch) if in_ch != out_ch else None


class LatentRateReduction(nn.Module):
    """layers/rem.py:69-141."""

    def __init__(self, dim_chunk=32, mu_std=True, dimension="middle"):
        super().__init__()
        self.dim_block, self.mu_std = dim_chunk, mu_std
        N = dim_chunk
        extra = 1 if dimension == "big" else 0
        Seq = nn.Sequential
        RB = ResidualBlock
        pin = 2 * N if mu_std else N
        self.enc_base_entropy_params = Seq(RB(2 * N, N), *[RB(N, N) for _ in range(1 + extra)])
        self.enc_progressive_entropy_params = Seq(RB(pin, N), *[RB(N, N) for _ in range(1 + extra)])
        self.enc_base_rep = Seq(*[RB(N, N) for _ in range(2 + extra)])
        self.enc = Seq(RB(3 * N, 2 * N), *[RB(2 * N, 2 * N) for _ in range(1 + extra)], RB(2 * N, 2 * N if mu_std else N))

    def forward(self, x_base, entropy_params_base, entropy_params_prog, att_mask):
        from . import engine
        return engine.run_rem_module(self, x_base, entropy_params_base, entropy_params_prog, att_mask)


# ------------------------------------------------------------------ variance mask
class ChannelMask(nn.Module):
    """layers/channel_mask.py:9-156 — "point-based-std" and "two-levels" policies."""

    def __init__(self, mask_policy):
        super().__init__()
        self.mask_policy = mask_policy

    def apply_noise(self, mask, training):
        # round() is the identity on a {0,1} mask (channel_mask.py:81-86); the STE form has
        # the same value and only matters for autograd, which this build does not provide.
        return mask

    def forward(self, scale, pr=0, mask_pol="point-based-std", ravel=False, cust_map=None):
        if cust_map is not None or ravel:
            raise NotImplementedError("cust_map / ravel branches are dead code in the reference (channel_mask.py:96-121)")
        if mask_pol is None:
            mask_pol = self.mask_policy
        if mask_pol == "two-levels":
            return torch.zeros_like(scale) if pr == 0 else torch.ones_like(scale)
        if mask_pol != "point-based-std":
            raise NotImplementedError()
        assert scale is not None
        _no_autograd(scale)
        v = ops.from_nchw(scale)
        m = ops.new_view(v.B, v.H, v.W, v.C)
        ops.variance_mask(v, pr, m, n_slice=1)
        return m.torch_nchw()

    def ProgMask(self, scale: List[torch.Tensor], pr):
        """list of [1,C,h,w] blocks -> [len,C,h,w] (channel_mask.py:18-49)."""
        blocks = [ops.from_nchw(b) for b in scale]
        v0 = blocks[0]
        n = len(blocks)
        assert all(b.B == 1 and b.C == v0.C and b.H == v0.H and b.W == v0.W for b in blocks)
        cat = ops.new_view(1, v0.H, v0.W, v0.C * n)
        for i, b in enumerate(blocks):     # gather the blocks into one NHWC buffer (memcpy only)
            cat.buf[..., i * v0.C:(i + 1) * v0.C].copy_(b.buf[..., b.c0:b.c0 + b.C])
        m = ops.new_view(1, v0.H, v0.W, v0.C * n)
        ops.variance_mask(cat, pr, m, n_slice=n)
        return m.buf.reshape(v0.H, v0.W, n, v0.C).permute(2, 3, 0, 1)
