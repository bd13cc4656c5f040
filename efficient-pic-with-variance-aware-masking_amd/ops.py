"""Thin host-side wrappers over the C ABI.

``View`` is a channel window of an NHWC fp32 device buffer (``[B,H,W,ld]`` with
``C <= ld`` channels starting at ``c0``).  torch only owns the memory and the stream;
every arithmetic op below is a libvampic call.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib as L


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


@dataclass
class View:
    buf: torch.Tensor   # [B, H, W, ld] fp32 contiguous, cuda
    c0: int
    C: int

    @property
    def B(self): return self.buf.shape[0]
    @property
    def H(self): return self.buf.shape[1]
    @property
    def W(self): return self.buf.shape[2]
    @property
    def ld(self): return self.buf.shape[3]
    @property
    def ptr(self): return self.buf.data_ptr() + 4 * self.c0
    @property
    def n_pix(self): return self.B * self.H * self.W

    def window(self, c0: int, C_: int) -> "View":
        assert 0 <= c0 and c0 + C_ <= self.C, (c0, C_, self.C)
        return View(self.buf, self.c0 + c0, C_)

    def torch_nchw(self) -> torch.Tensor:
        """Logical NCHW tensor sharing this window's memory (channels_last strides)."""
        return self.buf[..., self.c0:self.c0 + self.C].permute(0, 3, 1, 2)


@dataclass
class View3:
    """A channel window of a bf16x3-plane ("P3") tensor: [B, H, W, groups][plane 3][8 bf16], 48 bytes per 8-channel
    group (stored in an fp32 tensor of 12 floats per group).  Written by a conv launch with VAM_CONV_OUT_BF3 and read
    only by convolutions (VAM_CONV_IN_BF3): the exact hi/mid/lo split of each fp32 value, made once by the producer."""
    buf: torch.Tensor   # [B, H, W, 12 * groups] fp32 storage
    g0: int
    C: int

    @property
    def B(self): return self.buf.shape[0]
    @property
    def H(self): return self.buf.shape[1]
    @property
    def W(self): return self.buf.shape[2]
    @property
    def ld(self): return self.buf.shape[3] // 12      # groups per pixel
    @property
    def ptr(self): return self.buf.data_ptr() + 48 * self.g0
    @property
    def n_pix(self): return self.B * self.H * self.W

    def window(self, c0: int, C_: int) -> "View3":
        assert c0 % 8 == 0 and C_ % 8 == 0 and 0 <= c0 and c0 + C_ <= self.C
        return View3(self.buf, self.g0 + c0 // 8, C_)

    def to_float(self) -> torch.Tensor:
        """Decode to a [B,H,W,C] fp32 tensor (hi + mid + lo; test / debug helper)."""
        B, H, W = self.B, self.H, self.W
        raw = self.buf.view(torch.int16).view(B, H, W, self.ld, 3, 8)[..., self.g0:self.g0 + self.C // 8, :, :]
        planes = (raw.to(torch.int32) << 16).view(torch.float32)
        return (planes[..., 0, :] + planes[..., 1, :] + planes[..., 2, :]).reshape(B, H, W, self.C)


@dataclass
class View16:
    """A channel window of a bf16 NHWC tensor (bf16-storage mode, BASELINE configs[2]): activations of the large feature
    maps of g_a / g_s are kept in bf16; ``ld`` counts bf16 elements."""
    buf: torch.Tensor   # [B, H, W, ld] torch.bfloat16, cuda
    c0: int
    C: int

    @property
    def B(self): return self.buf.shape[0]
    @property
    def H(self): return self.buf.shape[1]
    @property
    def W(self): return self.buf.shape[2]
    @property
    def ld(self): return self.buf.shape[3]
    @property
    def ptr(self): return self.buf.data_ptr() + 2 * self.c0
    @property
    def n_pix(self): return self.B * self.H * self.W

    def window(self, c0: int, C_: int) -> "View16":
        assert c0 % 8 == 0 and C_ % 8 == 0 and 0 <= c0 and c0 + C_ <= self.C
        return View16(self.buf, self.c0 + c0, C_)

    def torch_nchw(self) -> torch.Tensor:
        return self.buf[..., self.c0:self.c0 + self.C].permute(0, 3, 1, 2).float()


def new_view16(B: int, H: int, W: int, C_: int, device="cuda") -> View16:
    assert C_ % 8 == 0
    return View16(torch.empty((B, H, W, C_), dtype=torch.bfloat16, device=device), 0, C_)


def new_view3(B: int, H: int, W: int, C_: int, device="cuda") -> View3:
    assert C_ % 8 == 0
    return View3(torch.zeros((B, H, W, (C_ // 8) * 12), dtype=torch.float32, device=device), 0, C_)


def split_mode() -> bool:
    """True when the convolution kernel runs on split bf16x3 operands (the default)."""
    return L.load().vam_conv_get_mode() == 1


def f16x2_mode() -> bool:
    """True when the convolution kernel runs on fp16x2 operands (opt-in: VAMPIC_CONV=f16x2 / vam_conv_set_mode(3))."""
    return L.load().vam_conv_get_mode() == 3


def amax_prepare(chunk: Sequence["L.VamConv"]):
    """fp16x2 mode, eager launches (``conv_group``): give every input segment of every problem of one grouped launch a
    max-|x| cell and return (cells, fn) where ``fn()`` zeroes the cells and reduces each segment into its cell on the
    current stream.  Plans avoid the re-read where the producer was a convolution (engine.Plan: ``out_amax`` cells)."""
    lib = L.load()
    cells = torch.zeros(len(chunk) * L.VAM_MAX_SEG, dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
    base = cells.data_ptr()
    todo = []
    for i, c in enumerate(chunk):
        c._amax_cells = cells            # the structs point into ``cells``: it lives as long as they do
        if c.flags & (L.CONV_W_BF16 | L.CONV_IN_BF3):
            continue
        for k in range(c.n_seg):
            cell = base + 4 * (i * L.VAM_MAX_SEG + k)
            c.in_amax[k] = cell
            todo.append((c, k, cell))

    def fill():
        if not todo:
            return
        s = stream_ptr()
        L.check(lib.vam_memset_zero(base, 4 * cells.numel(), s), "vam_memset_zero")
        for c, k, cell in todo:
            L.check(lib.vam_absmax(C.byref(c.seg[k]), 1, c.B * c.H * c.W, cell, s), "vam_absmax")
    return cells, fill


def new_view(B: int, H: int, W: int, C_: int, device="cuda", zero: bool = False) -> View:
    f = torch.zeros if zero else torch.empty
    return View(f((B, H, W, C_), dtype=torch.float32, device=device), 0, C_)


def from_nchw(x: torch.Tensor) -> View:
    """NCHW (any strides) -> NHWC view.  Uses the layout kernel for plain-contiguous inputs;
    a channels_last tensor is already NHWC in memory and is wrapped without a copy."""
    assert x.dim() == 4 and x.dtype == torch.float32 and x.is_cuda, "expected a cuda fp32 NCHW tensor"
    B, C_, H, W = x.shape
    xp = x.permute(0, 2, 3, 1)
    if xp.is_contiguous():
        return View(xp, 0, C_)
    x = x.contiguous()
    out = new_view(B, H, W, C_)
    L.check(L.load().vam_nchw_to_nhwc(x.data_ptr(), out.ptr, B, C_, H, W, out.ld, stream_ptr()), "vam_nchw_to_nhwc")
    return out


def to_nchw(v: View) -> torch.Tensor:
    out = torch.empty((v.B, v.C, v.H, v.W), dtype=torch.float32, device=v.buf.device)
    L.check(L.load().vam_nhwc_to_nchw(v.ptr, v.ld, out.data_ptr(), v.B, v.C, v.H, v.W, stream_ptr()), "vam_nhwc_to_nchw")
    return out


# --------------------------------------------------------------------------- weights
@dataclass
class Packed:
    """A conv problem's constant part: packed weights + bias and its geometry."""
    w: torch.Tensor
    b: Optional[torch.Tensor]
    kh: int
    kw: int
    cin: int
    n: int
    stride: int = 1
    pad_y: int = 0
    pad_x: int = 0
    ps2_cq: int = 0          # >0: PixelShuffle-style phase scatter with Cq channels per phase
    osy: int = 1
    osx: int = 1
    ooy: int = 0
    oox: int = 0
    w16: bool = False        # weights rounded to bf16 (vam_pack_conv_weights_bf16): the bf16-storage kernel


def pack_weights(src: torch.Tensor, mode: int, phase: int, kh: int, kw: int, cin: int, n: int, bf16: bool = False) -> torch.Tensor:
    lib = L.load()
    src = src.detach().to(dtype=torch.float32).contiguous()
    assert src.is_cuda
    if bf16:
        dst = torch.empty(lib.vam_conv_wpack_bf16_bytes(kh, kw, cin, n) // 4, dtype=torch.float32, device=src.device)
        L.check(lib.vam_pack_conv_weights_bf16(src.data_ptr(), dst.data_ptr(), mode, phase, kh, kw, cin, n, stream_ptr()),
                "vam_pack_conv_weights_bf16")
        return dst
    dst = torch.empty(lib.vam_conv_wpack_floats(kh, kw, cin, n), dtype=torch.float32, device=src.device)
    L.check(lib.vam_pack_conv_weights(src.data_ptr(), dst.data_ptr(), mode, phase, kh, kw, cin, n, stream_ptr()),
            "vam_pack_conv_weights")
    return dst


_PACK_JOBS: Optional[list] = None      # inside ``pack_batch()``: deferred (job, keep-alive) pairs


class pack_batch:
    """``with ops.pack_batch(): ...`` — the repack calls inside are collected and issued as grouped launches
    (``vam_pack_group``) when the block ends, on the stream that is current then.  Nothing may read the packed buffers
    inside the block."""

    def __enter__(self):
        global _PACK_JOBS
        self.prev, _PACK_JOBS = _PACK_JOBS, []
        return self

    def __exit__(self, *exc):
        global _PACK_JOBS
        jobs, _PACK_JOBS = _PACK_JOBS, self.prev
        if exc[0] is None and jobs:
            arr = (L.VamPackJob * len(jobs))(*[j for j, _ in jobs])
            L.check(L.load().vam_pack_group(arr, len(jobs), stream_ptr()), "vam_pack_group")
        return False


def _pack_job(src: torch.Tensor, dst: torch.Tensor, bias: bool, mode: int, phase: int, kh: int, kw: int, cin: int, n: int) -> bool:
    if _PACK_JOBS is None:
        return False
    j = L.VamPackJob()
    j.src, j.dst, j.bias, j.mode, j.phase, j.kh, j.kw, j.cin, j.n = src.data_ptr(), dst.data_ptr(), int(bias), mode, phase, kh, kw, cin, n
    _PACK_JOBS.append((j, (src, dst)))
    return True


def repack_conv(conv_weight: torch.Tensor, conv_bias: Optional[torch.Tensor], pk: "Packed", dgrad: bool = False):
    """Refresh ``pk`` IN PLACE from the (trained) parameters: plans keep pointing at the same packed buffers
    while the optimiser updates the weights between steps."""
    lib = L.load()
    assert conv_weight.is_cuda and conv_weight.is_contiguous() and conv_weight.dtype == torch.float32
    mode = L.PACK_CONV_DGRAD if dgrad else L.PACK_CONV
    if not _pack_job(conv_weight, pk.w, False, mode, 0, pk.kh, pk.kw, pk.cin, pk.n):
        L.check(lib.vam_pack_conv_weights(conv_weight.data_ptr(), pk.w.data_ptr(), mode, 0, pk.kh, pk.kw, pk.cin, pk.n,
                                          stream_ptr()), "vam_pack_conv_weights")
    if not dgrad and conv_bias is not None:
        if not _pack_job(conv_bias, pk.b, True, L.PACK_CONV, 0, 0, 0, 0, pk.n):
            L.check(lib.vam_pack_bias(conv_bias.data_ptr(), pk.b.data_ptr(), L.PACK_CONV, pk.n, stream_ptr()), "vam_pack_bias")


def pack_bias(src: torch.Tensor, mode: int, n: int) -> torch.Tensor:
    lib = L.load()
    src = src.detach().to(dtype=torch.float32).contiguous()
    dst = torch.empty(n, dtype=torch.float32, device=src.device)
    L.check(lib.vam_pack_bias(src.data_ptr(), dst.data_ptr(), mode, n, stream_ptr()), "vam_pack_bias")
    return dst


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, bf16: bool = False) -> Packed:
    """nn.Conv2d(k, stride, padding=k//2)  (reference layers/layers.py:5-12,24-26,77-79)."""
    n, cin, kh, kw = weight.shape
    return Packed(pack_weights(weight, L.PACK_CONV, 0, kh, kw, cin, n, bf16),
                  None if bias is None else pack_bias(bias, L.PACK_CONV, n),
                  kh, kw, cin, n, stride, kh // 2, kw // 2, w16=bf16)


def pack_linear(weight: torch.Tensor, bias: Optional[torch.Tensor], bf16: bool = False) -> Packed:
    n, cin = weight.shape
    return Packed(pack_weights(weight, L.PACK_CONV, 0, 1, 1, cin, n, bf16),
                  None if bias is None else pack_bias(bias, L.PACK_CONV, n), 1, 1, cin, n, w16=bf16)


def pack_subpel(weight: torch.Tensor, bias: torch.Tensor) -> Packed:
    """conv3x3(Cin -> 4*Cq) + PixelShuffle(2)  (layers/layers.py:82-86)."""
    n, cin, kh, kw = weight.shape
    return Packed(pack_weights(weight, L.PACK_PS2, 0, kh, kw, cin, n), pack_bias(bias, L.PACK_PS2, n),
                  kh, kw, cin, n, 1, kh // 2, kw // 2, ps2_cq=n // 4)


def pack_conv5s2_rgb(weight: torch.Tensor, bias: torch.Tensor, bf16: bool = False) -> Packed:
    """conv5x5 s2 pad2 on 3 input channels as a 3x3 s1 problem over the space-to-depth
    input of vam_s2d_input (16 channels = (py,px,c) + 4 zeros).  Pure re-indexing of the
    weights (a gather of existing values and zeros): w'[n,(py,px,c),ty,tx] = w[n,c,2ty+py,2tx+px]."""
    n, cin, kh, kw = weight.shape
    assert cin == 3 and kh == 5 and kw == 5
    w6 = torch.zeros((n, 3, 6, 6), dtype=torch.float32, device=weight.device)
    w6[:, :, :5, :5] = weight.detach()
    w = w6.reshape(n, 3, 3, 2, 3, 2).permute(0, 3, 5, 1, 2, 4).reshape(n, 12, 3, 3)   # [n,(py,px,c),ty,tx]
    w16 = torch.zeros((n, 16, 3, 3), dtype=torch.float32, device=weight.device)
    w16[:, :12] = w
    return Packed(pack_weights(w16, L.PACK_CONV, 0, 3, 3, 16, n, bf16), pack_bias(bias, L.PACK_CONV, n),
                  3, 3, 16, n, 1, 1, 1, w16=bf16)


def pack_deconv(weight: torch.Tensor, bias: torch.Tensor, bf16: bool = False) -> List[Packed]:
    """ConvTranspose2d(k5,s2,p2,op1) (layers/layers.py:14-22) as sub-pixel phase problems.
    Cout % 4 == 0: four problems (taps 3x3,3x2,2x3,2x2) writing interleaved output pixels.
    Otherwise (Cout = 3): one merged 3x3 problem with phase-major channels + PS2 scatter."""
    cin, cout, kh, kw = weight.shape
    assert kh == 5 and kw == 5
    if cout % 4 == 0:
        b = pack_bias(bias, L.PACK_CONV, cout)
        out = []
        for ph in range(4):
            py, px = ph >> 1, ph & 1
            k_h, k_w = (2 if py else 3), (2 if px else 3)
            out.append(Packed(pack_weights(weight, L.PACK_DECONV5S2, ph, k_h, k_w, cin, cout, bf16), b, k_h, k_w, cin, cout,
                              1, 0 if py else 1, 0 if px else 1, osy=2, osx=2, ooy=py, oox=px, w16=bf16))
        return out
    n = 4 * cout
    return [Packed(pack_weights(weight, L.PACK_DECONV5S2, -1, 3, 3, cin, n, bf16), pack_bias(bias, L.PACK_DECONV5S2, n),
                   3, 3, cin, n, 1, 1, 1, ps2_cq=cout, w16=bf16)]


def pack_gdn(beta: torch.Tensor, gamma: torch.Tensor, bf16: bool = False) -> Packed:
    """GDN norm pool as a 1x1 problem on x^2 with the reparametrised gamma/beta (layers/gdn.py:62-69)."""
    c = beta.shape[0]
    return Packed(pack_weights(gamma, L.PACK_GDN, 0, 1, 1, c, c, bf16), pack_bias(beta, L.PACK_GDN, c), 1, 1, c, c, w16=bf16)


# --------------------------------------------------------------------------- conv problems
def _aux(v: Optional[View]) -> L.VamAux:
    a = L.VamAux()
    if v is not None:
        a.ptr = v.ptr
        a.ld = v.ld
    return a


def conv_problem(pk: Packed, inputs: Sequence[View], out: View, act: int = L.ACT_NONE, *,
                 pre: Optional[View] = None, mul: Optional[View] = None, post: Optional[View] = None,
                 post2: Optional[View] = None, flags: int = 0, out_nchw: Optional[torch.Tensor] = None,
                 preact: Optional[View] = None, gelu_z: Optional[View] = None) -> L.VamConv:
    """Describe one problem.  ``inputs`` are concatenated along channels (virtually).  Training plans: ``preact`` = a second
    output holding the value the activation is applied to (the taped input of a GELU); ``gelu_z`` = the pre-activation of
    the GELU in FRONT of this (data-gradient) layer — the result is multiplied by gelu'(z): (conv + bias + pre) * gelu'(z)."""
    if gelu_z is not None:
        assert mul is None and act == L.ACT_NONE and type(gelu_z) is View
        mul, flags = gelu_z, flags | L.CONV_MUL_GELU_GRAD
    c = L.VamConv()
    assert 1 <= len(inputs) <= L.VAM_MAX_SEG
    in3 = isinstance(inputs[0], View3)
    assert all(isinstance(v, View3) == in3 for v in inputs), "a conv problem reads either fp32 or bf16x3-plane segments"
    in16 = isinstance(inputs[0], View16)
    assert all(isinstance(v, View16) == in16 for v in inputs), "a conv problem reads either fp32 or bf16 segments"
    if in3:
        flags |= L.CONV_IN_BF3
    if isinstance(out, View3):
        assert out_nchw is None
        flags |= L.CONV_OUT_BF3
    auxs = [a for a in (pre, mul, post, post2) if a is not None]
    aux16 = bool(auxs) and isinstance(auxs[0], View16)
    assert all(isinstance(a, View16) == aux16 for a in auxs), "epilogue operands of one problem share a storage type"
    if in16 or isinstance(out, View16) or aux16 or pk.w16:
        assert pk.w16, "bf16-stored tensors are read / written by the bf16-weight kernel: pack the layer with bf16=True"
        flags |= L.CONV_W_BF16 | (L.CONV_IN_BF16 if in16 else 0) | (L.CONV_OUT_BF16 if isinstance(out, View16) else 0) | \
            (L.CONV_AUX_BF16 if aux16 else 0)
    B, H, W = inputs[0].B, inputs[0].H, inputs[0].W
    cin = 0
    for i, v in enumerate(inputs):
        assert (v.B, v.H, v.W) == (B, H, W), "all input segments must share the spatial extent"
        c.seg[i].ptr = v.ptr
        c.seg[i].C = v.C
        c.seg[i].ld = v.ld
        cin += v.C
    assert cin == pk.cin, f"conv expects {pk.cin} input channels, got {cin}"
    c.n_seg = len(inputs)
    c.B, c.H, c.W = B, H, W
    c.kh, c.kw, c.stride, c.pad_y, c.pad_x = pk.kh, pk.kw, pk.stride, pk.pad_y, pk.pad_x
    if pk.stride == 1:
        Ho, Wo = H, W
    else:
        Ho = (H + 2 * pk.pad_y - pk.kh) // pk.stride + 1
        Wo = (W + 2 * pk.pad_x - pk.kw) // pk.stride + 1
    c.Ho, c.Wo = Ho, Wo
    c.N = pk.n
    c.wpack = pk.w.data_ptr()
    c.bias = pk.b.data_ptr() if pk.b is not None else None
    c.osy, c.osx, c.ooy, c.oox = pk.osy, pk.osx, pk.ooy, pk.oox
    c.Cq = pk.ps2_cq
    c.act = act
    c.flags = flags | (L.CONV_PS2 if pk.ps2_cq else 0)
    if out_nchw is not None:
        assert out_nchw.is_contiguous()
        c.flags |= L.CONV_OUT_NCHW
        c.out = out_nchw.data_ptr()
        c.Hf, c.Wf = out_nchw.shape[2], out_nchw.shape[3]
        c.ldo = 0
        cexp = pk.ps2_cq if pk.ps2_cq else pk.n
        assert out_nchw.shape[0] == B and out_nchw.shape[1] == cexp
    else:
        c.out = out.ptr
        c.ldo = out.ld
        c.Hf, c.Wf = out.H, out.W
        cexp = pk.ps2_cq if pk.ps2_cq else pk.n
        assert out.B == B and out.C == cexp, f"output window has {out.C} channels, problem writes {cexp}"
    if pk.ps2_cq:
        assert (c.Hf, c.Wf) == (2 * Ho, 2 * Wo)
    else:
        assert (Ho - 1) * pk.osy + pk.ooy < c.Hf and (Wo - 1) * pk.osx + pk.oox < c.Wf
    c.pre, c.mul, c.post, c.post2 = _aux(pre), _aux(mul), _aux(post), _aux(post2)
    if preact is not None:
        assert type(preact) is View and out_nchw is None and preact.B == B and (preact.H, preact.W) == (c.Hf, c.Wf) and \
            preact.C == (pk.ps2_cq if pk.ps2_cq else pk.n), "preact is an fp32 NHWC tensor of the output's shape"
        c.preact = _aux(preact)
    return c


def conv_group(problems: Sequence[L.VamConv]):
    lib = L.load()
    s = stream_ptr()
    for i in range(0, len(problems), L.VAM_MAX_GROUP):
        chunk = problems[i:i + L.VAM_MAX_GROUP]
        if f16x2_mode():
            cells, fill = amax_prepare(chunk)
            fill()
        arr = (L.VamConv * len(chunk))(*chunk)
        L.check(lib.vam_conv_group(arr, len(chunk), s), "vam_conv_group")


# --------------------------------------------------------------------------- fused residual unit
def resunit_supported(x) -> bool:
    """A fused kernel exists for this unit's shape (csrc/resunit.hip: C = 192; fp32 tensors in the split-operand mode, or
    bf16-stored tensors of the bf16-storage mode)."""
    if os.environ.get("VAMPIC_FUSED_RU", "1") == "0":
        return False
    if type(x) is View:
        if not split_mode():
            return False
    elif type(x) is not View16:
        return False
    return bool(L.load().vam_resunit_supported(x.C, x.H, x.W))


def resunit_problem(p1: Packed, p2: Packed, p3: Packed, x, out) -> "L.VamResunit":
    """One ResidualUnit (layers/layers.py:30-48) as a single launch; p1 / p2 / p3 are the ordinary packed weights of
    its 1x1 (C -> C/2), 3x3 (C/2 -> C/2) and 1x1 (C/2 -> C) convolutions — bf16-packed ones (``m.packed(True)``) when
    x / out are bf16-stored tensors."""
    b16 = type(x) is View16
    assert type(out) is type(x) and type(x) in (View, View16), "the fused residual unit reads and writes fp32 or bf16 NHWC tensors"
    assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, x.C)
    C_ = x.C
    assert (p1.kh, p1.cin, p1.n) == (1, C_, C_ // 2) and (p2.kh, p2.kw, p2.cin, p2.n, p2.stride) == (3, 3, C_ // 2, C_ // 2, 1) and \
        (p3.kh, p3.cin, p3.n) == (1, C_ // 2, C_), "residual unit: conv1x1(C, C/2), conv3x3(C/2, C/2), conv1x1(C/2, C)"
    assert p1.w16 == p2.w16 == p3.w16 == b16 and p1.b is not None and p2.b is not None and p3.b is not None
    c = L.VamResunit()
    c.x, c.out, c.ldx, c.ldo = x.ptr, out.ptr, x.ld, out.ld
    c.B, c.H, c.W, c.C = x.B, x.H, x.W, C_
    c.w1, c.b1 = p1.w.data_ptr(), p1.b.data_ptr()
    c.w2, c.b2 = p2.w.data_ptr(), p2.b.data_ptr()
    c.w3, c.b3 = p3.w.data_ptr(), p3.b.data_ptr()
    c.flags = L.RESUNIT_BF16 if b16 else 0
    return c


def resunit_group(problems: Sequence["L.VamResunit"]):
    lib = L.load()
    s = stream_ptr()
    for i in range(0, len(problems), L.VAM_MAX_GROUP):
        chunk = problems[i:i + L.VAM_MAX_GROUP]
        arr = (L.VamResunit * len(chunk))(*chunk)
        L.check(lib.vam_resunit_group(arr, len(chunk), s), "vam_resunit_group")


# --------------------------------------------------------------------------- other ops
def s2d_input(x: torch.Tensor) -> View:
    B, C_, H, W = x.shape
    assert C_ == 3 and x.is_cuda and x.dtype == torch.float32
    x = x.contiguous()
    out = new_view(B, H // 2, W // 2, 16)
    L.check(L.load().vam_s2d_input(x.data_ptr(), out.ptr, B, H, W, stream_ptr()), "vam_s2d_input")
    return out


def win_attention(qkv: View, out: View, table: torch.Tensor, C_: int, heads: int, ws: int, shift: int):
    L.check(L.load().vam_win_attention(qkv.ptr, qkv.ld, out.ptr, out.ld, table.data_ptr(), qkv.B, qkv.H, qkv.W, C_, heads,
                                       ws, shift, stream_ptr()), "vam_win_attention")


def variance_mask(sigma: View, pr: float, mask: View, n_slice: int = 1, slice_C: Optional[int] = None,
                  thr: Optional[torch.Tensor] = None):
    """sigma/mask windows hold n_slice consecutive slices of slice_C channels each; one
    quantile per (batch item, slice)."""
    slice_C = slice_C or sigma.C // n_slice
    assert slice_C * n_slice == sigma.C == mask.C
    hw = sigma.H * sigma.W
    L.check(L.load().vam_variance_mask(sigma.ptr, sigma.ld, hw * sigma.ld, slice_C, sigma.B, n_slice, hw, slice_C, float(pr),
                                       mask.ptr, mask.ld, hw * mask.ld, slice_C,
                                       thr.data_ptr() if thr is not None else None, stream_ptr()), "vam_variance_mask")


@dataclass
class IView:
    """int32 NHWC channel window (symbols / table indexes)."""
    buf: torch.Tensor   # [B,H,W,ld] int32
    c0: int
    C: int

    @property
    def ptr(self): return self.buf.data_ptr() + 4 * self.c0
    @property
    def ld(self): return self.buf.shape[3]

    def window(self, c0, C_): return IView(self.buf, self.c0 + c0, C_)


def new_iview(B, H, W, C_, device="cuda") -> IView:
    return IView(torch.empty((B, H, W, C_), dtype=torch.int32, device=device), 0, C_)


def gauss_tail(y: View, mu: View, sigma: View, *, y2: Optional[View] = None, mask: Optional[View] = None,
               yhat: Optional[View] = None, lik: Optional[View] = None, sym=None,
               log2sum: Optional[torch.Tensor] = None):
    n_pix = y.n_pix
    def p(v): return (v.ptr, v.ld) if v is not None else (None, 0)
    if sym is None:
        sym_ptr, sym_ld = None, 0
    elif isinstance(sym, IView):
        sym_ptr, sym_ld = sym.ptr, sym.ld
    else:
        sym_ptr, sym_ld = sym.data_ptr(), sym.shape[-1]
    L.check(L.load().vam_gauss_tail(*p(y), *p(y2), *p(mu), *p(sigma), *p(mask), *p(yhat), *p(lik), sym_ptr, sym_ld,
                                    log2sum.data_ptr() if log2sum is not None else None, y.H * y.W, n_pix, y.C,
                                    stream_ptr()), "vam_gauss_tail")


def build_indexes(sigma: View, table: torch.Tensor, mask: Optional[View] = None, out: Optional[IView] = None) -> torch.Tensor:
    if out is None:
        out = new_iview(sigma.B, sigma.H, sigma.W, sigma.C, sigma.buf.device)
    L.check(L.load().vam_build_indexes(sigma.ptr, sigma.ld, mask.ptr if mask is not None else None,
                                       mask.ld if mask is not None else 0, table.data_ptr(), table.numel(),
                                       out.ptr, out.ld, sigma.n_pix, sigma.C, stream_ptr()), "vam_build_indexes")
    return out.buf


def dequantize(sym: IView, mu: Optional[View], out: View):
    L.check(L.load().vam_dequantize(sym.ptr, sym.ld, mu.ptr if mu is not None else None, mu.ld if mu is not None else 0,
                                    out.ptr, out.ld, out.n_pix, out.C, stream_ptr()), "vam_dequantize")


def eb_forward(z: View, params: torch.Tensor, zhat: Optional[View], lik: Optional[View],
               log2sum: Optional[torch.Tensor] = None, sym: Optional[IView] = None, noise: Optional[View] = None):
    def p(v): return (v.ptr, v.ld) if v is not None else (None, 0)
    L.check(L.load().vam_eb_forward_noise(z.ptr, z.ld, params.data_ptr(), z.C, *p(zhat), *p(lik), *p(sym),
                                          log2sum.data_ptr() if log2sum is not None else None, z.H * z.W, z.n_pix,
                                          *p(noise), stream_ptr()), "vam_eb_forward")


def add(a: View, b: View, out: View):
    L.check(L.load().vam_add(a.ptr, a.ld, b.ptr, b.ld, out.ptr, out.ld, a.n_pix, a.C, stream_ptr()), "vam_add")


# ---- REM fine-tune backward pieces (csrc/train.hip)
def pack_conv_dgrad(weight: torch.Tensor) -> Packed:
    """Weights of the data-gradient conv of a stride-1 ``nn.Conv2d`` (taps flipped, channel roles swapped)."""
    n_out, c_in, kh, kw = weight.shape
    w = pack_weights(weight, L.PACK_CONV_DGRAD, 0, kh, kw, n_out, c_in)
    return Packed(w, None, kh, kw, n_out, c_in, 1, kh // 2, kw // 2)


def wgrad_problems(x_segs: Sequence[View], dy: View, dw: torch.Tensor, db: Optional[torch.Tensor],
                   stride: int = 1) -> List[L.VamWgrad]:
    """dw (OIHW) / db of a conv (stride 1 pad k/2, or the k5/s2/p2 one) whose input was the channel concat of
    ``x_segs``: one problem per segment.  dw may be a 2-D [N, C] tensor (a 1x1 layer / nn.Linear)."""
    if dw.dim() == 2:
        n, cin_total, kh, kw = dw.shape[0], dw.shape[1], 1, 1
    else:
        n, cin_total, kh, kw = dw.shape
    assert dy.C == n and sum(v.C for v in x_segs) == cin_total and dw.is_contiguous() and kh == kw
    out, off = [], 0
    for v in x_segs:
        assert v.B == dy.B and (v.H, v.W) == (stride * dy.H, stride * dy.W)
        p = L.VamWgrad()
        p.x, p.dy, p.dw = v.ptr, dy.ptr, dw.data_ptr()
        p.db = db.data_ptr() if (db is not None and off == 0) else None
        p.ld_x, p.ld_dy, p.B, p.H, p.W, p.kh, p.kw, p.C, p.N = v.ld, dy.ld, dy.B, dy.H, dy.W, kh, kw, v.C, n
        if isinstance(v, View3):          # taped activation stored as bf16x3 planes: the LDS-tiled kernel copies instead of splitting
            assert kh == 3 and stride == 1 and wgrad_reads_planes(dy.H, dy.W), "plane inputs: k3 stride-1 layers on an LDS-kernel grid"
            p.flags = L.WGRAD_X_P3
        p.cin_total, p.c_off = cin_total, off
        p.stride, p.Hx, p.Wx = stride, v.H, v.W
        p._dev = dy.buf.device
        p._keep = (v, dy, dw, db)         # the struct holds raw pointers: the tensors live at least as long as it does
        out.append(p)
        off += v.C
    wgrad_plan(out)                       # alone; a grouped launch re-plans with each problem's share of the group
    return out


def wgrad_reads_planes(H: int, W: int) -> bool:
    """True when the weight gradient of a k3 stride-1 layer on an H x W grid takes the LDS-tiled kernel, which can read its
    input as bf16x3 planes (``View3``)."""
    return split_mode() and bool(L.load().vam_conv_wgrad_lds_grid(int(H), int(W)))


def train_tape_planes(H: int, W: int) -> bool:
    """Policy: write the taped activations of the training slice stacks as bf16x3 planes where forward convolution and
    weight gradient can both read them.  OFF by default — built, bit-identical (tests/test_gpu_wgrad_lds.py), and measured
    no faster on the first_train step (interleaved on one box: 140.2 / 140.3 ms with, 140.1 / 139.9 without,
    profiles/r04_first_train_ab.txt: the forward gains what the 1.5x bytes of the plane stores cost the backward).
    ``VAMPIC_TRAIN_P3=1`` switches it on."""
    return os.environ.get("VAMPIC_TRAIN_P3", "0") == "1" and wgrad_reads_planes(H, W)


def wgrad_plan(problems: Sequence[L.VamWgrad]):
    """Pixel splits (vam_conv_wgrad_plan) and the caller-owned scratch of the problems of ONE launch.  A problem that
    shares the launch with others plans against its share of the chip (its fraction of the group's FLOPs)."""
    lib = L.load()
    work = [float(p.B) * p.H * p.W * p.C * p.N * p.kh * p.kw for p in problems]
    tot = sum(work) or 1.0
    # VAMPIC_WGRAD_SHARE (experiment): the fraction of the chip a weight-gradient launch is planned against.  On a branch of
    # its own (engine.Plan.wgrad_branch) a launch shares the chip with the data-gradient chain.
    chip = float(os.environ.get("VAMPIC_WGRAD_SHARE", "1.0"))
    for p, w in zip(problems, work):
        share = 1.0 if len(problems) == 1 else max(w / tot, 1e-3)
        share *= chip
        p.slot_share = 0.0 if share == 1.0 else max(share, 1e-3)
        nbytes = C.c_size_t(0)
        p.splits = lib.vam_conv_wgrad_plan(C.byref(p), C.byref(nbytes))
        if p.splits > 1:                  # caller-owned scratch for the pixel-split partial tiles; lives with the problem
            if getattr(p, "_ws", None) is None or p._ws.numel() * 4 < nbytes.value:
                p._ws = torch.empty(nbytes.value // 4, dtype=torch.float32, device=p._dev)
            p.workspace = p._ws.data_ptr()
        else:
            p._ws, p.workspace = None, None


def wgrad_group(problems: Sequence[L.VamWgrad]):
    lib = L.load()
    for i in range(0, len(problems), L.VAM_MAX_WGRAD_GROUP):
        chunk = list(problems[i:i + L.VAM_MAX_WGRAD_GROUP])
        if len(chunk) > 1:
            wgrad_plan(chunk)
        arr = (L.VamWgrad * len(chunk))(*chunk)
        L.check(lib.vam_conv_wgrad_group(arr, len(chunk), stream_ptr()), "vam_conv_wgrad_group")


def conv_wgrad(x_segs: Sequence[View], dy: View, dw: torch.Tensor, db: Optional[torch.Tensor]):
    wgrad_group(wgrad_problems(x_segs, dy, dw, db))


def leaky_bwd(act: View, dy: View, dx: View):
    L.check(L.load().vam_leaky_bwd(act.ptr, act.ld, dy.ptr, dy.ld, dx.ptr, dx.ld, dx.n_pix, dx.C, stream_ptr()), "vam_leaky_bwd")


def mul(a: View, b: View, out: View):
    L.check(L.load().vam_mul(a.ptr, a.ld, b.ptr, b.ld, out.ptr, out.ld, out.n_pix, out.C, stream_ptr()), "vam_mul")


def gauss_train(y: View, mu: View, sigma: View, noise: View, *, y2: Optional[View] = None, mask: Optional[View] = None,
                lik: Optional[View] = None, grad_lik: Optional[View] = None, dmu: Optional[View] = None,
                dsigma: Optional[View] = None):
    """Noisy-likelihood forward (``lik``) or backward (``grad_lik`` -> ``dmu``, ``dsigma``)."""
    def p(v): return (v.ptr, v.ld) if v is not None else (None, 0)
    L.check(L.load().vam_gauss_train(*p(y), *p(y2), *p(mu), *p(sigma), *p(mask), *p(noise), *p(grad_lik), *p(lik),
                                     *p(dmu), *p(dsigma), y.n_pix, y.C, stream_ptr()), "vam_gauss_train")


# ---- transform backward pieces (csrc/train_gs.hip)
def ew(op: int, ins: Sequence[View], outs: Sequence[View], coef: float = 0.0, flag: int = 0):
    """One element-wise derivative kernel (enum vam_ew_op) over channel windows of equal extent."""
    e = L.VamEw()
    v0 = ins[0]
    for k, v in enumerate(ins):
        assert v.n_pix == v0.n_pix and v.C == v0.C
        e.inp[k].ptr, e.inp[k].ld = v.ptr, v.ld
    for k, v in enumerate(outs):
        assert v.n_pix == v0.n_pix and v.C == v0.C
        e.out[k].ptr, e.out[k].ld = v.ptr, v.ld
    e.n_pix, e.C, e.flag, e.coef = v0.n_pix, v0.C, int(flag), float(coef)
    L.check(L.load().vam_train_elementwise(op, C.byref(e), stream_ptr()), "vam_train_elementwise")


def stack_tail_ok(x, m4, m5, out: Optional[View], kw: dict) -> bool:
    """``out`` None = a fresh buffer.  Can the last two layers (m4: 128 -> 64 + GELU, m5: 64 -> 32) of a slice stack run as ONE launch (csrc/stack_tail.hip)?"""
    if os.environ.get("VAMPIC_STACK_TAIL", "1") != "1" or not split_mode():
        return False
    if not (isinstance(x, View3) and x.C == 128 and x.g0 == 0 and x.ld == 16 and x.W == 16 and x.H % 4 == 0):
        return False
    if (out is not None and (type(out) is not View or out.C != 32)) or any(kw.get(k) is not None for k in ("pre", "mul")) or kw.get("flags"):
        return False
    if kw.get("act", L.ACT_NONE) not in (L.ACT_NONE, L.ACT_HALF_TANH) or any(v is not None and type(v) is not View for v in (kw.get("post"), kw.get("post2"))):
        return False
    ok = lambda m, ci, co: (type(m).__name__ == "Conv2d" and m.in_channels == ci and m.out_channels == co and m.kernel_size == 3 and m.stride == 1)
    return ok(m4, 128, 64) and ok(m5, 64, 32)


def stack_tail_problem(x: "View3", pk4: Packed, pk5: Packed, out: View, act: int = L.ACT_NONE, post: Optional[View] = None,
                       post2: Optional[View] = None) -> "L.VamStackTail":
    t = L.VamStackTail()
    assert pk4.b is not None and pk5.b is not None and not pk4.w16 and not pk5.w16
    t.x, t.w4, t.b4, t.w5, t.b5, t.out = x.ptr, pk4.w.data_ptr(), pk4.b.data_ptr(), pk5.w.data_ptr(), pk5.b.data_ptr(), out.ptr
    t.B, t.H, t.W, t.x_groups, t.ld_out, t.act = x.B, x.H, x.W, x.ld, out.ld, int(act)
    for name, v in (("post", post), ("post2", post2)):
        if v is not None:
            assert v.n_pix == out.n_pix and v.C == out.C
            a = getattr(t, name)
            a.ptr, a.ld = v.ptr, v.ld
    t._keep = (x, pk4, pk5, out, post, post2)
    return t


def stack_tail_group(problems: Sequence["L.VamStackTail"]):
    arr = (L.VamStackTail * len(problems))(*problems)
    L.check(L.load().vam_stack_tail_group(arr, len(problems), stream_ptr()), "vam_stack_tail_group")


def axpy_jobs(updates: Sequence[tuple]):
    """Pre-marshalled jobs of :func:`axpy_group`: updates = [(dst, src, coef)] means dst += coef * src on channel windows of
    equal extent.  Returns the ctypes array (keep it alive as long as a plan step refers to it)."""
    assert 1 <= len(updates) <= L.VAM_MAX_EW_GROUP
    arr = (L.VamEw * len(updates))()
    for e, (dst, src, coef) in zip(arr, updates):
        assert dst.n_pix == src.n_pix and dst.C == src.C
        e.inp[0].ptr, e.inp[0].ld = dst.ptr, dst.ld
        e.inp[1].ptr, e.inp[1].ld = src.ptr, src.ld
        e.out[0].ptr, e.out[0].ld = dst.ptr, dst.ld
        e.n_pix, e.C, e.flag, e.coef = dst.n_pix, dst.C, 0, float(coef)
    return arr


def axpy_group(arr):
    """Several ``dst += coef * src`` updates of windows that do not overlap, ONE launch (vam_train_axpy_group)."""
    L.check(L.load().vam_train_axpy_group(arr, len(arr), stream_ptr()), "vam_train_axpy_group")


def flat_view(t: torch.Tensor, width: int = 4) -> View:
    """A contiguous fp32 tensor of any shape as an [n/width, width] channel window (element-wise kernels only)."""
    assert t.is_contiguous() and t.dtype == torch.float32 and t.numel() % width == 0
    return View(t.view(1, 1, t.numel() // width, width), 0, width)


def win_attention_bwd_workspace(qkv: View, heads: int, ws: int) -> torch.Tensor:
    """Caller-owned scratch of vam_win_attention_bwd (per-block partial bias-table gradients)."""
    n = L.load().vam_win_attention_bwd_workspace(qkv.B, qkv.H, qkv.W, heads, ws)
    return torch.empty(n // 4, dtype=torch.float32, device=qkv.buf.device)


def win_attention_bwd(qkv: View, dout: View, dqkv: View, table: torch.Tensor, dtable: torch.Tensor, C_: int, heads: int,
                      ws: int, shift: int, workspace: Optional[torch.Tensor] = None):
    assert dtable.shape == table.shape and dtable.is_contiguous()
    if workspace is None:
        workspace = win_attention_bwd_workspace(qkv, heads, ws)
    assert workspace.numel() * 4 >= L.load().vam_win_attention_bwd_workspace(qkv.B, qkv.H, qkv.W, heads, ws)
    L.check(L.load().vam_win_attention_bwd(qkv.ptr, qkv.ld, dout.ptr, dout.ld, dqkv.ptr, dqkv.ld, table.data_ptr(),
                                           dtable.data_ptr(), workspace.data_ptr(), qkv.B, qkv.H, qkv.W, C_, heads, ws, shift,
                                           stream_ptr()), "vam_win_attention_bwd")


def colsum_workspace(dy: View) -> torch.Tensor:
    return torch.empty(L.load().vam_colsum_workspace(dy.n_pix, dy.C) // 4, dtype=torch.float32, device=dy.buf.device)


def colsum(dy: View, out: torch.Tensor, workspace: Optional[torch.Tensor] = None):
    if workspace is None:
        workspace = colsum_workspace(dy)
    L.check(L.load().vam_colsum(dy.ptr, dy.ld, dy.n_pix, dy.C, out.data_ptr(), workspace.data_ptr(), stream_ptr()), "vam_colsum")


def repack_weights(src: torch.Tensor, dst: torch.Tensor, mode: int, phase: int, kh: int, kw: int, cin: int, n: int):
    """vam_pack_conv_weights into an EXISTING packed buffer (training: the optimiser changed ``src`` in place)."""
    assert src.is_cuda and src.is_contiguous() and src.dtype == torch.float32
    if _pack_job(src, dst, False, mode, phase, kh, kw, cin, n):
        return
    L.check(L.load().vam_pack_conv_weights(src.data_ptr(), dst.data_ptr(), mode, phase, kh, kw, cin, n, stream_ptr()),
            "vam_pack_conv_weights")


def repack_bias(src: torch.Tensor, dst: torch.Tensor, mode: int, n: int):
    if _pack_job(src, dst, True, mode, 0, 0, 0, 0, n):
        return
    L.check(L.load().vam_pack_bias(src.data_ptr(), dst.data_ptr(), mode, n, stream_ptr()), "vam_pack_bias")


def eb_train_bwd(z: View, noise: View, params: torch.Tensor, grad_lik: View, dz: View, dparams: torch.Tensor):
    """Backward of the training-mode entropy bottleneck (vam_eb_train_bwd): dz written, dparams in ``params`` layout."""
    assert dparams.numel() == params.numel() and dparams.is_contiguous()
    L.check(L.load().vam_eb_train_bwd(z.ptr, z.ld, noise.ptr, noise.ld, params.data_ptr(), z.C, grad_lik.ptr, grad_lik.ld,
                                      dz.ptr, dz.ld, dparams.data_ptr(), z.n_pix, stream_ptr()), "vam_eb_train_bwd")


def ps2_unshuffle(src: View, dst: View):
    """Gradient of PixelShuffle(2): src [B,2H,2W,Cq] -> dst [B,H,W,4Cq] in the convolution's channel order c*4+i*2+j."""
    assert src.C * 4 == dst.C and (src.H, src.W) == (2 * dst.H, 2 * dst.W) and src.B == dst.B
    L.check(L.load().vam_ps2_unshuffle(src.ptr, src.ld, dst.ptr, dst.ld, dst.B, dst.H, dst.W, src.C, stream_ptr()), "vam_ps2_unshuffle")


def upsample2_zero(src: View, dst: View):
    """dst[b,2y,2x] = src[b,y,x], zeros elsewhere (data gradient of a stride-2 3x3 convolution, first half)."""
    assert src.C == dst.C and (dst.H, dst.W) == (2 * src.H, 2 * src.W) and src.B == dst.B
    L.check(L.load().vam_upsample2_zero(src.ptr, src.ld, dst.ptr, dst.ld, src.B, src.H, src.W, src.C, stream_ptr()), "vam_upsample2_zero")


def memset_zero(t: torch.Tensor):
    L.check(L.load().vam_memset_zero(t.data_ptr(), t.numel() * t.element_size(), stream_ptr()), "vam_memset_zero")


def sqdiff_sum(a: torch.Tensor, b: torch.Tensor, acc: torch.Tensor):
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    L.check(L.load().vam_sqdiff_sum(a.data_ptr(), b.data_ptr(), a.numel(), acc.data_ptr(), stream_ptr()), "vam_sqdiff_sum")


# --------------------------------------------------------------------------- graphs / profiling
_CAPTURING = 0                      # depth of stream captures in progress on this (single) host thread
_GRAVEYARD: list = []               # (exec handle, torch stream it was last launched on) of dropped graphs


class Graph:
    """hipGraph capture of a sequence of libvampic launches on the current stream.

    Lifetime.  A dropped graph (``close()`` or garbage collection) only parks its handle: no HIP call from ``__del__``,
    which can run inside another plan's capture.  :func:`drain_graveyard` — called by the plans at their entry points,
    never during a capture — synchronises the stream each parked handle was last launched on and then RETIRES it: the
    handle is kept for the life of the process and ``hipGraphExecDestroy`` is not called.

    Why not destroyed (round 3, demonstrated; ROCm 7.2 / gfx950).  The host segfault of round 2 (``hipGraphLaunch``
    inside tests/test_gpu_bitstream.py, right after ``net.update()`` had dropped the model's plans) came back
    DETERMINISTICALLY with the deferred, synchronised, return-code-checked destruction of round 3 in place, in the test
    order ops -> model -> golden -> config_variants -> bitstream (two runs of two; ~20 executable graphs destroyed at
    that entry point, then the capture and launch of a new plan faults inside the runtime, gpurun_out/r3_segv.log).  The
    same order with the handles retired instead of destroyed: 73 passed (gpurun_out/r3_segv2.log).  Every destroyed
    graph had finished (its stream synchronised), none was destroyed twice (``close`` clears the handle), no capture was
    active: the library's side of the contract holds, and the only difference between crash and no crash is the call
    to hipGraphExecDestroy.  A retired executable graph costs host memory for its kernel arguments (~0.3 MB for a
    230-launch plan); at most ``VAMPIC_GRAPH_RETIRE_MAX`` (256) handles are kept, older ones are destroyed after all.
    ``VAMPIC_GRAPH_DESTROY=1`` restores immediate destruction.  Round 4 could not make the fault reappear with the
    destruction on (stand-alone probe with every ingredient, the library's own sequence, the faulting test order twice:
    DESIGN.md section 5), so the cause is unknown and retirement stays the default."""

    def __init__(self):
        self.exec = C.c_void_p(None)
        self.last_stream = None

    def capture(self, fn):
        global _CAPTURING
        drain_graveyard()
        lib = L.load()
        s = stream_ptr()
        dev = torch.cuda.current_device()
        n_alloc = torch.cuda.memory_stats(dev).get("allocation.all.allocated", 0)
        L.check(lib.vam_graph_begin(s), "vam_graph_begin")
        _CAPTURING += 1
        try:
            fn()
        finally:
            _CAPTURING -= 1
            rc = lib.vam_graph_end(s, C.byref(self.exec))
        L.check(rc, "vam_graph_end")
        # The capture is a raw hipStreamBeginCapture torch's caching allocator knows nothing about: a tensor allocated by a
        # captured step would be returned to the pool while the graph keeps its address.  Plans allocate while they are
        # BUILT, never while they run — enforce it.
        n_new = torch.cuda.memory_stats(dev).get("allocation.all.allocated", 0) - n_alloc
        if n_new:
            self.close()
            raise RuntimeError(f"{n_new} device allocation(s) inside a hipGraph capture: a captured step created a torch temporary "
                               "(pre-allocate it when the plan is built and write through views)")

    def launch(self):
        self.last_stream = torch.cuda.current_stream()
        L.check(L.load().vam_graph_launch(self.exec, self.last_stream.cuda_stream), "vam_graph_launch")

    def close(self):
        """Give the executable graph up; it is destroyed at the next safe point."""
        if self.exec:
            _GRAVEYARD.append((self.exec.value, self.last_stream))
            self.exec = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()                # no HIP call here: see the class docstring
        except Exception:
            pass


def drain_graveyard():
    """Retire (or, with VAMPIC_GRAPH_DESTROY=1, destroy) the executable graphs dropped since the last call, after
    synchronising the streams they last ran on.  No-op while a capture is in progress."""
    if _CAPTURING or not _GRAVEYARD:
        return
    dead = list(_GRAVEYARD)
    del _GRAVEYARD[:]
    seen = set()
    for _, st in dead:
        if st is not None and st.cuda_stream not in seen:
            seen.add(st.cuda_stream)
            st.synchronize()            # the last replay of every dropped graph has finished
    lib = L.load()
    if os.environ.get("VAMPIC_GRAPH_DESTROY", "0") != "1":   # default: see the class docstring
        _RETIRED.extend(h for h, _ in dead)
        # bounded: a loop that changes the weights and runs an eval forward every step drops a plan per step.
        # Beyond the cap the OLDEST retired handles are destroyed after all — their last launch is long finished, and round
        # 4's probes found the destruction harmless in every set-up tried.
        cap = int(os.environ.get("VAMPIC_GRAPH_RETIRE_MAX", "256"))
        while len(_RETIRED) > cap:
            L.check(lib.vam_graph_destroy(C.c_void_p(_RETIRED.pop(0))), "vam_graph_destroy")
        return
    for handle, _ in dead:
        L.check(lib.vam_graph_destroy(C.c_void_p(handle)), "vam_graph_destroy")


_RETIRED: list = []      # handles of dropped executable graphs, kept alive on purpose


def graveyard_size() -> int:
    return len(_GRAVEYARD)


def retired_graphs() -> int:
    return len(_RETIRED)


_PROF_ON = False


def prof_enable(on: bool):
    global _PROF_ON
    _PROF_ON = bool(on)
    L.check(L.load().vam_prof_enable(1 if on else 0))


def prof_on() -> bool:
    return _PROF_ON


def prof_set_class(name_or_idx):
    """Launch class of the convolution launches that follow (event profiler only; see L.PROF_CLASSES)."""
    i = L.PROF_CLASSES.index(name_or_idx) if isinstance(name_or_idx, str) else int(name_or_idx)
    L.check(L.load().vam_prof_set_class(i), "vam_prof_set_class")


def prof_read_classes():
    lib = L.load()
    out = {}
    for i, name in enumerate(L.PROF_CLASSES):
        ms, n, fl, by = C.c_double(), C.c_long(), C.c_double(), C.c_double()
        L.check(lib.vam_prof_read_class(i, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
    return out


def prof_reset():
    L.check(L.load().vam_prof_reset())


def prof_read():
    lib = L.load()
    out = {}
    for fam, name in enumerate(L.FAMILY_NAMES):
        ms, n, fl, by = C.c_double(), C.c_long(), C.c_double(), C.c_double()
        L.check(lib.vam_prof_read(fam, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
    return out
