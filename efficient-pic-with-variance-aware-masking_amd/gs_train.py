"""Training-mode lowering of a synthesis transform (first schedule: ``--training_type refine_gs``,
reference train.py:150-157,216-218 — everything frozen except ``g_s[1]``; loss = DistortionLoss, training/loss.py:126-187).

Forward: the same kernels as the eval lowering (engine.lower_g_s), but every tensor the backward needs is kept ("tape"):
pre-activations of the GELUs, the pre-sigmoid gate operand, the IGDN norm pool, qkv.  Activations that the eval path
applies in the producing convolution's epilogue are applied by an element-wise kernel with the SAME formula here, so the
forward values are bit-identical to the eval path.

Backward (its own plan): clamp -> transposed-conv data / weight gradients -> IGDN -> attention block -> ..., in reverse.
  * data gradient of a stride-1 conv / Linear: the conv kernel on VAM_PACK_CONV_DGRAD weights;
  * data gradient of ConvTranspose2d(k5, s2, p2, op1): a k5/s2/p2 convolution with the SAME weight tensor read as OIHW;
  * its weight gradient: vam_conv_wgrad (stride 2) with the roles of input and output gradient exchanged;
  * IGDN:  y = x * sqrt(n), n = beta' + gamma' x^2:  dn = dy x / (2 sqrt n);  dx = dy sqrt n + 2 x (gamma'^T dn);
           dgamma' = dn (x^2)^T, dbeta' = sum dn, then the NonNegativeParametrizer chain (LowerBound rule);
  * window attention: csrc/train_gs.hip (softmax backward, dq/dk/dv, relative-position-bias gradient).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from . import layers as Ly
from . import ops
from .ops import View

_GAMMA_BOUND = 2.0 ** -18                        # NonNegativeParametrizer(): sqrt(0 + 2^-36)
_BETA_BOUND = (1e-6 + 2.0 ** -36) ** 0.5         # NonNegativeParametrizer(minimum=1e-6)


class TransformPacks:
    """Forward and data-gradient packed weights of every layer of ONE transform stack, re-packed IN PLACE at the head of
    every step (the optimiser changes the parameters between steps; plans and graphs keep their pointers)."""

    def __init__(self, stack: nn.Module):
        self.f: Dict[int, object] = {}
        self.d: Dict[int, ops.Packed] = {}
        self._refresh = []
        self.keep = []
        self._walk(stack)

    def _walk(self, mod):
        if isinstance(mod, Ly.SubpelConv):             # conv3x3 + PixelShuffle(2): a leaf (its inner Conv2d packs phase-major)
            self._add_subpel(mod)
        elif isinstance(mod, Ly.Conv2d):
            if mod.is_rgb_s2d:
                self._add_conv_rgb(mod)
            elif mod.stride == 2:
                self._add_conv_s2(mod)
            else:
                self._add_conv(mod)
        elif isinstance(mod, Ly.Linear):
            self._add_linear(mod)
        elif isinstance(mod, Ly.ConvTranspose2d):
            self._add_deconv(mod)
        elif isinstance(mod, Ly.GDN):
            self._add_gdn(mod)
        else:
            for ch in mod.children():
                self._walk(ch)

    # ---- first-stage training (analysis transforms, hyperprior): the remaining layer kinds
    def _add_conv_s2(self, c):
        """Stride-2 convolution (k5: g_a, layers/layers.py:5-12; k3: h_a, builder.py:72-82).  Data gradient of the k5 one =
        the transposed convolution k5/s2/p2/op1 with the SAME tensor read as IOHW (I = Cout, O = Cin): four sub-pixel phase
        problems; of the k3 one = VAM_PACK_CONV_DGRAD on the zero-inserted output gradient (vam_upsample2_zero)."""
        n, cin, k = c.out_channels, c.in_channels, c.kernel_size
        f = ops.pack_conv(c.weight, c.bias, 2)
        if k == 5:
            assert cin % 4 == 0
            zb = torch.zeros(cin, dtype=torch.float32, device=c.weight.device)
            d = ops.pack_deconv(c.weight, zb)
            self.keep.append(zb)
        else:
            d = ops.pack_conv_dgrad(c.weight)
        self.f[id(c)], self.d[id(c)] = f, d

        def refresh():
            ops.repack_weights(c.weight, f.w, L.PACK_CONV, 0, k, k, cin, n)
            ops.repack_bias(c.bias, f.b, L.PACK_CONV, n)
            if k == 5:
                for ph, pk in enumerate(d):
                    ops.repack_weights(c.weight, pk.w, L.PACK_DECONV5S2, ph, pk.kh, pk.kw, n, cin)
            else:
                ops.repack_weights(c.weight, d.w, L.PACK_CONV_DGRAD, 0, k, k, n, cin)
        self._refresh.append(refresh)

    def _add_conv_rgb(self, c):
        """conv5x5 s2 (3 -> N) run as a 3x3 problem on the space-to-depth input (ops.pack_conv5s2_rgb): the re-indexed
        16-channel weight tensor is rebuilt from the parameter at every refresh; the image has no data gradient."""
        n = c.out_channels
        w16 = torch.zeros((n, 16, 3, 3), dtype=torch.float32, device=c.weight.device)
        w6 = torch.zeros((n, 3, 6, 6), dtype=torch.float32, device=c.weight.device)
        self.keep += [w16, w6]

        def gather():           # runs inside the captured refresh step: strided copies between views, no temporaries
            w6[:, :, :5, :5].copy_(c.weight.detach())
            w16[:, :12].view(n, 2, 2, 3, 3, 3).copy_(w6.view(n, 3, 3, 2, 3, 2).permute(0, 3, 5, 1, 2, 4))
        gather()
        f = ops.Packed(ops.pack_weights(w16, L.PACK_CONV, 0, 3, 3, 16, n), ops.pack_bias(c.bias, L.PACK_CONV, n), 3, 3, 16, n, 1, 1, 1)
        self.f[id(c)] = f

        def refresh():
            gather()
            ops.repack_weights(w16, f.w, L.PACK_CONV, 0, 3, 3, 16, n)
            ops.repack_bias(c.bias, f.b, L.PACK_CONV, n)
        self._refresh.append(refresh)

    def _add_subpel(self, m):
        c = m[0]
        n, cin = c.out_channels, c.in_channels
        f = ops.pack_subpel(c.weight, c.bias)
        d = ops.pack_conv_dgrad(c.weight)                # on the un-shuffled gradient, reference channel order
        self.f[id(m)], self.d[id(m)] = f, d

        def refresh():
            ops.repack_weights(c.weight, f.w, L.PACK_PS2, 0, 3, 3, cin, n)
            ops.repack_bias(c.bias, f.b, L.PACK_PS2, n)
            ops.repack_weights(c.weight, d.w, L.PACK_CONV_DGRAD, 0, 3, 3, n, cin)
        self._refresh.append(refresh)

    def _add_conv(self, c):
        n, cin, k = c.out_channels, c.in_channels, c.kernel_size
        f = ops.pack_conv(c.weight, c.bias, 1)
        d = ops.pack_conv_dgrad(c.weight)
        self.f[id(c)], self.d[id(c)] = f, d

        def refresh():
            ops.repack_weights(c.weight, f.w, L.PACK_CONV, 0, k, k, cin, n)
            ops.repack_bias(c.bias, f.b, L.PACK_CONV, n)
            ops.repack_weights(c.weight, d.w, L.PACK_CONV_DGRAD, 0, k, k, n, cin)
        self._refresh.append(refresh)

    def _add_linear(self, l):
        n, cin = l.out_features, l.in_features
        f = ops.pack_linear(l.weight, l.bias)
        d = ops.Packed(ops.pack_weights(l.weight.detach().reshape(n, cin, 1, 1), L.PACK_CONV_DGRAD, 0, 1, 1, n, cin), None,
                       1, 1, n, cin)
        self.f[id(l)], self.d[id(l)] = f, d

        def refresh():
            ops.repack_weights(l.weight, f.w, L.PACK_CONV, 0, 1, 1, cin, n)
            ops.repack_bias(l.bias, f.b, L.PACK_CONV, n)
            ops.repack_weights(l.weight, d.w, L.PACK_CONV_DGRAD, 0, 1, 1, n, cin)
        self._refresh.append(refresh)

    def _add_deconv(self, m):
        ci, co = m.in_channels, m.out_channels
        f = ops.pack_deconv(m.weight, m.bias)
        # data gradient: dx[ci] = conv5x5 s2 p2 over dy[co] with W[ci][co][ky][kx] read as OIHW (O = ci, I = co); the
        # conv kernel wants input channels in multiples of 16, so a 3-channel output gradient is zero-padded
        cp = (co + 15) // 16 * 16
        wsrc = m.weight if cp == co else torch.zeros((ci, cp, 5, 5), dtype=torch.float32, device=m.weight.device)
        if cp != co:
            wsrc[:, :co].copy_(m.weight.detach())
            self.keep.append(wsrc)
        d = ops.Packed(ops.pack_weights(wsrc, L.PACK_CONV, 0, 5, 5, cp, ci), None, 5, 5, cp, ci, 2, 2, 2)
        self.f[id(m)], self.d[id(m)] = f, d

        def refresh():
            if len(f) == 4:
                for ph, pk in enumerate(f):
                    ops.repack_weights(m.weight, pk.w, L.PACK_DECONV5S2, ph, pk.kh, pk.kw, ci, co)
                ops.repack_bias(m.bias, f[0].b, L.PACK_CONV, co)
            else:
                ops.repack_weights(m.weight, f[0].w, L.PACK_DECONV5S2, -1, 3, 3, ci, 4 * co)
                ops.repack_bias(m.bias, f[0].b, L.PACK_DECONV5S2, 4 * co)
            if cp != co:
                wsrc[:, :co].copy_(m.weight.detach())
            ops.repack_weights(wsrc, d.w, L.PACK_CONV, 0, 5, 5, cp, ci)
        self._refresh.append(refresh)

    def _add_gdn(self, g):
        c = g.in_channels
        f = ops.pack_gdn(g.beta, g.gamma)
        d = ops.Packed(ops.pack_weights(g.gamma, L.PACK_GDN_T, 0, 1, 1, c, c), None, 1, 1, c, c)
        self.f[id(g)], self.d[id(g)] = f, d

        def refresh():
            ops.repack_weights(g.gamma, f.w, L.PACK_GDN, 0, 1, 1, c, c)
            ops.repack_bias(g.beta, f.b, L.PACK_GDN, c)
            ops.repack_weights(g.gamma, d.w, L.PACK_GDN_T, 0, 1, 1, c, c)
        self._refresh.append(refresh)

    def record_refresh(self, plan: E.Plan):
        fns = list(self._refresh)
        def run():
            with ops.pack_batch():               # one grouped launch per 32 repacks instead of one launch each
                for fn in fns:
                    fn()
        plan.call(run, f"repack {len(fns)} trained layers")


# ============================================================================= forward (taped)
def _gelu(plan, pre: View) -> View:
    o = plan.buf(pre.B, pre.H, pre.W, pre.C)
    plan.call(lambda: ops.ew(L.EW_GELU_FWD, [pre], [o]), "gelu")
    return o


def _ru_fwd(plan, pk: TransformPacks, ru: Ly.ResidualUnit, x: View, tape: list) -> View:
    """GELU(conv1x1(GELU(conv3x3(GELU(conv1x1(x))))) + x)   (layers/layers.py:30-48).  Every launch applies its GELU in the
    epilogue (as the eval plans do) and keeps the pre-activation as a second output (the backward needs gelu'(z))."""
    c1, c2, c3 = ru.conv[0], ru.conv[2], ru.conv[4]
    h1p, h1 = plan.buf(x.B, x.H, x.W, c1.out_channels), plan.buf(x.B, x.H, x.W, c1.out_channels)
    plan.conv([ops.conv_problem(pk.f[id(c1)], [x], h1, L.ACT_GELU, preact=h1p)])
    h2p, h2 = plan.buf(x.B, x.H, x.W, c2.out_channels), plan.buf(x.B, x.H, x.W, c2.out_channels)
    plan.conv([ops.conv_problem(pk.f[id(c2)], [h1], h2, L.ACT_GELU, preact=h2p)])
    op, o = plan.buf(x.B, x.H, x.W, c3.out_channels), plan.buf(x.B, x.H, x.W, c3.out_channels)
    plan.conv([ops.conv_problem(pk.f[id(c3)], [h2], o, L.ACT_GELU, pre=x, preact=op)])
    tape.append(dict(kind="ru", mod=ru, x=x, h1p=h1p, h1=h1, h2p=h2p, h2=h2, op=op))
    return o


def _attention_block_fwd(plan, pk, blk: Ly.Win_noShift_Attention, x: View, tape: list, out: Optional[View] = None) -> View:
    """a * sigmoid(b) + x  (layers/layers.py:50-74); ``out``: write the result into this window."""
    rec = dict(kind="attn", mod=blk, x=x, a_tape=[], b_tape=[])
    wa = blk.conv_b[0]
    qkv = plan.buf(x.B, x.H, x.W, 3 * x.C)
    plan.conv([ops.conv_problem(pk.f[id(wa.attn.qkv)], [x], qkv)])
    att = plan.buf(x.B, x.H, x.W, x.C)
    tab = wa.attn.relative_position_bias_table
    plan.call(lambda: ops.win_attention(qkv, att, tab, x.C, wa.num_heads, wa.window_size, wa.shift_size), "win_attention")
    b = plan.buf(x.B, x.H, x.W, x.C)
    plan.conv([ops.conv_problem(pk.f[id(wa.attn.proj)], [att], b, post=x)])
    rec.update(qkv=qkv, att=att)
    for i in (1, 2, 3):
        b = _ru_fwd(plan, pk, blk.conv_b[i], b, rec["b_tape"])
    b4p = plan.buf(x.B, x.H, x.W, x.C)
    plan.conv([ops.conv_problem(pk.f[id(blk.conv_b[4])], [b], b4p)])
    a = x
    for i in range(3):
        a = _ru_fwd(plan, pk, blk.conv_a[i], a, rec["a_tape"])
    out = out if out is not None else plan.buf(x.B, x.H, x.W, x.C)
    plan.call(lambda: ops.ew(L.EW_GATE_FWD, [a, b4p, x], [out]), "gate")
    rec.update(a=a, b3=b, b4p=b4p)
    tape.append(rec)
    return out


def _deconv_fwd(plan, pk, m: Ly.ConvTranspose2d, x: View, tape: list, out_nchw: Optional[torch.Tensor] = None,
                act: int = L.ACT_NONE) -> Optional[View]:
    o = None if out_nchw is not None else plan.buf(x.B, 2 * x.H, 2 * x.W, m.out_channels)
    plan.conv([ops.conv_problem(p, [x], o, act, out_nchw=out_nchw) for p in pk.f[id(m)]])
    tape.append(dict(kind="deconv", mod=m, x=x))
    return o


def _gdn_fwd(plan, pk, g: Ly.GDN, x: View, tape: list) -> View:
    # one launch, as the eval plans: y = x * rsqrt(norm) (sqrt: inverse) in the epilogue; the norm pool itself — what the
    # backward needs — leaves as the second output
    n, y = plan.buf(x.B, x.H, x.W, x.C), plan.buf(x.B, x.H, x.W, x.C)
    plan.conv([ops.conv_problem(pk.f[id(g)], [x], y, L.ACT_SQRT if g.inverse else L.ACT_RSQRT, mul=x, flags=L.CONV_SQUARE_IN,
                                preact=n)])
    tape.append(dict(kind="gdn", mod=g, x=x, n=n))
    return y


def lower_g_s_train(plan: E.Plan, dec: nn.Sequential, y: View, x_hat: torch.Tensor, pk: TransformPacks, clamp: bool = True) -> list:
    """models/builder.py:8-18 + clamp_(0, 1) (pic.py:651; ``forward`` of pic.py:372,462 does not clamp), keeping the tape.
    Values equal engine.lower_g_s."""
    tape: list = []
    t = _attention_block_fwd(plan, pk, dec[0], y, tape)
    t = _deconv_fwd(plan, pk, dec[1], t, tape)
    t = _gdn_fwd(plan, pk, dec[2], t, tape)
    t = _deconv_fwd(plan, pk, dec[3], t, tape)
    t = _gdn_fwd(plan, pk, dec[4], t, tape)
    t = _attention_block_fwd(plan, pk, dec[5], t, tape)
    t = _deconv_fwd(plan, pk, dec[6], t, tape)
    t = _gdn_fwd(plan, pk, dec[7], t, tape)
    _deconv_fwd(plan, pk, dec[8], t, tape, out_nchw=x_hat, act=L.ACT_CLAMP01 if clamp else L.ACT_NONE)
    return tape


# ============================================================================= backward
def _conv_bwd(bw, pk, layer, x: View, dy: View, grads, need_dx: bool = True, dx_post: Optional[View] = None,
              gelu_z: Optional[View] = None) -> Optional[View]:
    """Weight / bias gradient of a stride-1 layer and (``need_dx``) its data gradient + ``dx_post``.  ``gelu_z``: the layer's
    input was GELU(z) — the data-gradient launch multiplies by gelu'(z) in its epilogue and returns dL/dz."""
    bw.wgrad(ops.wgrad_problems([x], dy, grads[id(layer.weight)], grads[id(layer.bias)]))
    if not need_dx:
        return None
    dx = bw.buf(x.B, x.H, x.W, x.C)
    if gelu_z is not None:
        bw.conv([ops.conv_problem(pk.d[id(layer)], [dy], dx, pre=dx_post, gelu_z=gelu_z)])
    else:
        bw.conv([ops.conv_problem(pk.d[id(layer)], [dy], dx, post=dx_post)])
    return dx


def _gelu_bwd(bw, pre: View, dy: View) -> View:
    o = bw.buf(pre.B, pre.H, pre.W, pre.C)
    bw.call(lambda: ops.ew(L.EW_GELU_BWD, [pre, dy], [o]), "gelu bwd")
    return o


def _ru_bwd(bw, pk, r: dict, d_o: View, grads, need_dx: bool = True, d_is_dz: bool = False,
            in_gelu_z: Optional[View] = None) -> Optional[View]:
    """``d_is_dz``: ``d_o`` is already dL/d(pre-activation of the unit's last GELU) (its producer applied gelu');
    ``in_gelu_z``: the unit's input is GELU(z) (the unit in front of it) — return dL/dz instead of dL/dx."""
    ru = r["mod"]
    c1, c2, c3 = ru.conv[0], ru.conv[2], ru.conv[4]
    dop = d_o if d_is_dz else _gelu_bwd(bw, r["op"], d_o)
    dh2p = _conv_bwd(bw, pk, c3, r["h2"], dop, grads, gelu_z=r["h2p"])
    dh1p = _conv_bwd(bw, pk, c2, r["h1"], dh2p, grads, gelu_z=r["h1p"])
    return _conv_bwd(bw, pk, c1, r["x"], dh1p, grads, need_dx, dx_post=dop, gelu_z=in_gelu_z)     # + the residual path


def _attention_block_bwd(bw, pk, r: dict, dout: View, grads, need_dx: bool) -> Optional[View]:
    blk = r["mod"]
    x = r["x"]
    da, db4 = bw.buf(x.B, x.H, x.W, x.C), bw.buf(x.B, x.H, x.W, x.C)
    bw.call(lambda: ops.ew(L.EW_GATE_BWD, [r["a"], r["b4p"], dout], [da, db4]), "gate bwd")
    bt, at = r["b_tape"], r["a_tape"]
    d = _conv_bwd(bw, pk, blk.conv_b[4], r["b3"], db4, grads, gelu_z=bt[-1]["op"])
    for i in range(len(bt) - 1, -1, -1):
        d = _ru_bwd(bw, pk, bt[i], d, grads, d_is_dz=True, in_gelu_z=bt[i - 1]["op"] if i > 0 else None)
    wa = blk.conv_b[0]
    d_att = _conv_bwd(bw, pk, wa.attn.proj, r["att"], d, grads)
    dqkv = bw.buf(x.B, x.H, x.W, 3 * x.C)
    tab = wa.attn.relative_position_bias_table
    dtab = grads[id(tab)]
    wsp = ops.win_attention_bwd_workspace(r["qkv"], wa.num_heads, wa.window_size)
    bw.keep.append(wsp)
    bw.call(lambda: ops.win_attention_bwd(r["qkv"], d_att, dqkv, tab, dtab, x.C, wa.num_heads, wa.window_size, wa.shift_size, wsp),
            "win_attention bwd")
    dx_b = _conv_bwd(bw, pk, wa.attn.qkv, x, dqkv, grads, need_dx, dx_post=d)          # + the shortcut of the Swin block
    d = da
    for i in range(len(at) - 1, -1, -1):
        d = _ru_bwd(bw, pk, at[i], d, grads, need_dx or i > 0, d_is_dz=i < len(at) - 1, in_gelu_z=at[i - 1]["op"] if i > 0 else None)
    if not need_dx:
        return None
    t, dx = bw.buf(x.B, x.H, x.W, x.C), bw.buf(x.B, x.H, x.W, x.C)
    bw.call(lambda: (ops.ew(L.EW_AXPY, [dout, d], [t], coef=1.0), ops.ew(L.EW_AXPY, [t, dx_b], [dx], coef=1.0)), "sum of the three paths")
    return dx


def _deconv_bwd(bw, pk, r: dict, dy: View, grads, need_dx: bool = True) -> Optional[View]:
    m = r["mod"]
    x = r["x"]
    ci, co = m.in_channels, m.out_channels
    gw, gb = grads[id(m.weight)], grads[id(m.bias)]
    if dy.C == co:
        bw.wgrad(ops.wgrad_problems([dy], x, gw, None, stride=2))
        wsp = ops.colsum_workspace(dy)
        bw.keep.append(wsp)
        with bw.off_path():
            bw.call(lambda: ops.colsum(dy, gb, wsp), "deconv bias grad")
    else:                                            # zero-padded 3-channel output gradient
        tw = torch.zeros((ci, dy.C, 5, 5), dtype=torch.float32, device=gw.device)
        tb = torch.zeros((dy.C,), dtype=torch.float32, device=gw.device)
        wsp = ops.colsum_workspace(dy)
        bw.keep += [tw, tb, wsp]
        with bw.off_path():
            bw.wgrad(ops.wgrad_problems([dy], x, tw, None, stride=2), now=True)      # tw is un-padded into gw right below
            # (torch.mul(.., 1.0, out=..) instead of a contiguous copy_: an element-wise KERNEL node under graph capture, not a
            # device-to-device memcpy node — see vam_memset_zero for what a non-kernel node did in these graphs)
            bw.call(lambda: (ops.colsum(dy, tb, wsp), gw.copy_(tw[:, :co]), torch.mul(tb[:co], 1.0, out=gb)), "deconv bias grad + unpad")
    if not need_dx:
        return None
    dx = bw.buf(x.B, x.H, x.W, ci)
    bw.conv([ops.conv_problem(pk.d[id(m)], [dy], dx)])
    return dx


def _gdn_bwd(bw, pk, r: dict, dy: View, grads) -> View:
    g = r["mod"]
    x, n = r["x"], r["n"]
    s, dx0, x2 = (bw.buf(x.B, x.H, x.W, x.C) for _ in range(3))
    bw.call(lambda: ops.ew(L.EW_GDN_BWD_PREP, [x, n, dy], [s, dx0, x2], flag=1 if g.inverse else 0), "gdn bwd prep")
    dx = bw.buf(x.B, x.H, x.W, x.C)                    # dx = dx0 + x * 2 (gamma'^T dL/dnorm): the sum in the launch's epilogue
    bw.conv([ops.conv_problem(pk.d[id(g)], [s], dx, L.ACT_DOUBLE, mul=x, post=dx0)])
    C_ = x.C
    tg = torch.zeros((C_, C_), dtype=torch.float32, device=x.buf.device)
    tb = torch.zeros((C_,), dtype=torch.float32, device=x.buf.device)
    bw.keep += [tg, tb]
    gg, gbeta = grads[id(g.gamma)], grads[id(g.beta)]
    with bw.off_path():                   # nothing on the data-gradient path reads gamma's / beta's gradients
        bw.wgrad(ops.wgrad_problems([x2], s, tg, tb), now=True)       # d gamma' [j][i] = sum dn_j x_i^2 ; d beta' = sum dn (read right below)
        bw.call(lambda: (ops.ew(L.EW_REPARAM_BWD, [ops.flat_view(g.gamma.detach()), ops.flat_view(tg)], [ops.flat_view(gg)], coef=_GAMMA_BOUND),
                         ops.ew(L.EW_REPARAM_BWD, [ops.flat_view(g.beta.detach()), ops.flat_view(tb)], [ops.flat_view(gbeta)], coef=_BETA_BOUND)),
                "NonNegativeParametrizer backward")
    return dx


def lower_g_s_backward(bw: E.Plan, tape: list, x_hat: torch.Tensor, g_xhat: torch.Tensor, pk: TransformPacks, grads,
                       need_input_grad: bool = False, clamp: bool = True) -> Optional[View]:
    """dL/d(parameters of the transform) from dL/dx_hat (NCHW, ``g_xhat``).  Nothing upstream of the transform's input is
    trainable under plain ``refine_gs`` (train.py:216-218); with ``--lrp`` the latent-residual-prediction stacks are
    (models/pic.py:171-184), and ``need_input_grad`` returns dL/dy_hat for them."""
    B, _, H, W = x_hat.shape
    gcl = torch.zeros_like(x_hat)
    d16 = bw.buf(B, H, W, 16, zero=True)
    bw.keep.append(gcl)
    if clamp:
        bw.call(lambda: (ops.ew(L.EW_CLAMP_BWD, [ops.flat_view(x_hat), ops.flat_view(g_xhat)], [ops.flat_view(gcl)]),
                         L.check(L.load().vam_nchw_to_nhwc(gcl.data_ptr(), d16.ptr, B, 3, H, W, d16.ld, ops.stream_ptr()), "vam_nchw_to_nhwc")),
                "clamp backward + NCHW -> NHWC")
    else:
        bw.call(lambda: L.check(L.load().vam_nchw_to_nhwc(g_xhat.data_ptr(), d16.ptr, B, 3, H, W, d16.ld, ops.stream_ptr()), "vam_nchw_to_nhwc"),
                "NCHW -> NHWC")
    d: Optional[View] = d16
    bw.defer_wgrad()                      # the transform's weight gradients leave grouped by shape (engine.Plan.flush_wgrad)
    for i, r in enumerate(reversed(tape)):
        first = i == len(tape) - 1
        if r["kind"] == "deconv":
            d = _deconv_bwd(bw, pk, r, d, grads)
        elif r["kind"] == "gdn":
            d = _gdn_bwd(bw, pk, r, d, grads)
        else:
            d = _attention_block_bwd(bw, pk, r, d, grads, need_dx=(not first) or need_input_grad)
    bw.flush_wgrad()
    return d


# ============================================================================= latent-residual-prediction stacks (--lrp)
def lower_lrp_stacks_train(plan: E.Plan, stacks: Sequence[nn.Sequential], inputs: Sequence[Sequence[View]],
                           rqs: Sequence[View], bases: Sequence[View], outs: Sequence[View],
                           packs: Sequence[TransformPacks]) -> List[dict]:
    """K progressive LRP stacks in lockstep with a tape (pic.py:635-641: y_hat_j = rq_j + 0.5 tanh(stack_j(cat(supports,
    rq_j))) + base_j).  Same kernels as the eval lowering, but every layer writes its pre-activation and the GELU / the
    0.5 tanh tail are element-wise launches with the epilogue's formulas.  The eval plan computes the hyperprior part of
    the first layer ahead of the slice loop (engine.lower_stack_heads: same sum, different association); a TRAINED stack
    runs its first layer whole, so its forward agrees with the eval plan to fp32 rounding, not bit for bit."""
    K = len(stacks)
    lay = [E.conv_layers(s) for s in stacks]
    depth = len(lay[0])
    cur: List[List[View]] = [list(i) for i in inputs]
    tapes = [dict(stack=stacks[k], x=[], z=[]) for k in range(K)]
    for d in range(depth):
        zs = []
        probs = []
        nxt: List[List[View]] = []
        for k in range(K):
            m, act = lay[k][d]
            assert isinstance(m, Ly.Conv2d) and m.stride == 1 and act == (L.ACT_NONE if d == depth - 1 else L.ACT_GELU)
            v0 = cur[k][0]
            z = plan.buf(v0.B, v0.H, v0.W, m.out_channels)
            if d < depth - 1:           # GELU in the epilogue, pre-activation kept as the second output
                # (as planes where the next layer's convolution and weight gradient both read planes: full_train.lower_stacks_train)
                p3 = m.out_channels % 8 == 0 and ops.train_tape_planes(v0.H, v0.W) and v0.B * v0.H * v0.W <= E.P3_MAX_PIXELS
                a = plan.buf3(v0.B, v0.H, v0.W, m.out_channels) if p3 else plan.buf(v0.B, v0.H, v0.W, m.out_channels)
                probs.append(ops.conv_problem(packs[k].f[id(m)], cur[k], a, L.ACT_GELU, preact=z))
                nxt.append([a])
            else:
                probs.append(ops.conv_problem(packs[k].f[id(m)], cur[k], z))
            tapes[k]["x"].append(list(cur[k]))
            tapes[k]["z"].append(z)
            zs.append(z)
        plan.conv(probs)
        if d < depth - 1:
            cur = nxt
        else:
            for z, rq, yb, o in zip(zs, rqs, bases, outs):
                plan.call(lambda z=z, rq=rq, yb=yb, o=o: ops.ew(L.EW_HTANH_FWD, [z, rq, yb], [o]), "lrp tail")
    return tapes


def lower_lrp_stacks_backward(bw: E.Plan, tapes: Sequence[dict], d_outs: Sequence[View], packs: Sequence[TransformPacks], grads):
    """dL/d(parameters of the LRP stacks) from dL/dy_hat_j.  The stacks' inputs (hyperprior means, supports, rq_j) have no
    trainable producer in this schedule, so the first layer needs no data gradient."""
    K = len(tapes)
    lay = [E.conv_layers(t["stack"]) for t in tapes]
    depth = len(lay[0])
    dz = []
    for t, dy in zip(tapes, d_outs):
        z = t["z"][-1]
        o = bw.buf(z.B, z.H, z.W, z.C)
        bw.call(lambda z=z, dy=dy, o=o: ops.ew(L.EW_HTANH_BWD, [z, dy], [o]), "lrp tail bwd")
        dz.append(o)
    for d in range(depth - 1, -1, -1):
        wg = []
        for k in range(K):
            m = lay[k][d][0]
            wg += ops.wgrad_problems(tapes[k]["x"][d], dz[k], grads[id(m.weight)], grads[id(m.bias)])
        bw.wgrad(wg)
        if d == 0:
            break
        das = [bw.buf(t["z"][d - 1].B, t["z"][d - 1].H, t["z"][d - 1].W, t["z"][d - 1].C) for t in tapes]
        bw.conv([ops.conv_problem(packs[k].d[id(lay[k][d][0])], [dz[k]], das[k], gelu_z=tapes[k]["z"][d - 1]) for k in range(K)])
        dz = das                          # the launch applied gelu'(z) of the GELU in front of the layer
