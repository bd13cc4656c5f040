"""First-stage training plan (BASELINE configs[3]): the reference's training-mode forward
``VarianceMaskingPIC.forward(x, quality=[0, q], training=True)`` (models/pic.py:301-491) — or
``forward_single_quality(x, q, training=True)`` (:497-666) — lowered WITH A TAPE, and its complete backward as a second
plan: every parameter on the path trains (train.py:146-149 ``first_train``; ``refine_gs_ga`` is a subset).

Data flow of the backward (reverse of pic.py:301-491; d* = dL/d*):
    dx_hat[1] -> g_s[1] -> dy_prog -> progressive LRP stacks -> (d rq, d supports)
    likelihood (additive-noise Gaussian, entropy_models.py:620-652) -> (dmu, dsigma);  dy = -dmu
    straight-through rounding  rq = ste_round(r - mu) * m + mu  (models/utils.py:4-5, pic.py:443):  dr = d rq * m,
        dmu += d rq * (1 - m);  the variance mask itself carries no gradient (hard comparison, channel_mask.py:132-151)
    progressive (mu, sigma) chain j = 9..0 -> supports (mu_total / std history, y_hat_base[j], hyperprior halves)
    dx_hat[0] -> g_s[0] -> dy_base;  base slices 9..0: LRP stack, likelihood, (mu, sigma) stacks
    hyper-synthesis stacks -> dz_hat = dz (straight-through);  entropy bottleneck (noise likelihood);  h_a -> dy
    g_a[0], g_a[1]  (conv5x5 s2 data gradient = the four sub-pixel phase problems of the transposed convolution)
Gradients of shared tensors are accumulated in zero-initialised buffers by element-wise launches, so every convolution
problem owns its output (grouped launches never race).  All parameter gradients live in ONE flat fp32 buffer laid out in
the order the backward produces them; ``bucket_bounds`` cuts it into ~25 MB buckets and ``bucket_ready`` gives, for each
bucket, the step of the backward plan after which it is final — the exchange of a multi-GPU job (RCCL all-reduce, one
per bucket, issued while the rest of the backward runs) hangs on these (sharding.BucketReducer).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from . import gs_train as G
from . import layers as Ly
from . import ops
from .ops import View

BUCKET_BYTES = 25 * 1024 * 1024
WGRAD_BRANCH = 7                # stream branch of the backward plan's weight-gradient launches
WGRAD_SIDE_DEFAULT = "3"
GS_BASE_BRANCH = 1              # the base decoder beside the progressive slice chain (forward and backward)


# ============================================================================= generic conv stacks with a tape
def _layer_out(m, v: View):
    if isinstance(m, Ly.SubpelConv):
        return 2 * v.H, 2 * v.W, m.out_ch
    if m.stride == 2:
        return v.H // 2, v.W // 2, m.out_channels
    return v.H, v.W, m.out_channels


def lower_stacks_train(plan: E.Plan, stacks: Sequence[nn.Sequential], inputs: Sequence[Sequence[View]],
                       outs: Sequence[Optional[View]], packs: Sequence[G.TransformPacks]) -> List[dict]:
    """K structurally identical conv stacks (conv / subpel layers with GELU between, no activation after the last:
    models/pic.py:83-164, builder.py:72-135) in lockstep, keeping every layer's input segments and pre-activation.
    The last layer writes ``outs[k]`` when given.  A trained stack runs its first layer whole (the eval plans hoist the
    hyperprior part, engine.lower_stack_heads: same sum, different association — fp32 rounding apart)."""
    K = len(stacks)
    lay = [E.conv_layers(s) for s in stacks]
    depth = len(lay[0])
    assert all(len(l) == depth for l in lay)
    cur: List[List[View]] = [list(i) for i in inputs]
    tapes = [dict(stack=stacks[k], x=[], z=[], out=None) for k in range(K)]
    for d in range(depth):
        probs, zs, nxt = [], [], []
        last = d == depth - 1
        for k in range(K):
            m, act = lay[k][d]
            assert act == (L.ACT_NONE if last else L.ACT_GELU)
            v0 = cur[k][0]
            Ho, Wo, Co = _layer_out(m, v0)
            z = outs[k] if (last and outs[k] is not None) else plan.buf(v0.B, Ho, Wo, Co)
            if last:
                probs.append(ops.conv_problem(packs[k].f[id(m)], cur[k], z))
            else:                       # GELU in the epilogue (as the eval plans), pre-activation kept as the second output
                # the activation is read by the next layer's convolution and by that layer's weight gradient only: where
                # both take bf16x3 planes it is written as planes (one split, by the producer), as in the eval stacks
                nm = lay[k][d + 1][0]
                p3 = Co % 8 == 0 and isinstance(m, Ly.Conv2d) and isinstance(nm, Ly.Conv2d) and nm.kernel_size == 3 and \
                    nm.stride == 1 and ops.train_tape_planes(Ho, Wo) and v0.B * Ho * Wo <= E.P3_MAX_PIXELS
                a = plan.buf3(v0.B, Ho, Wo, Co) if p3 else plan.buf(v0.B, Ho, Wo, Co)
                probs.append(ops.conv_problem(packs[k].f[id(m)], cur[k], a, L.ACT_GELU, preact=z))
                nxt.append([a])
            tapes[k]["x"].append(list(cur[k]))
            tapes[k]["z"].append(z)
            zs.append(z)
        plan.conv(probs)
        if not last:
            cur = nxt
        else:
            for k in range(K):
                tapes[k]["out"] = zs[k]
    return tapes


def lower_stacks_backward(bw: E.Plan, tapes: Sequence[dict], d_outs: Sequence[View], packs: Sequence[G.TransformPacks],
                          grads, need_dx: bool = True) -> Optional[List[View]]:
    """Backward of :func:`lower_stacks_train`: weight / bias gradients of every layer (``grads[id(param)]``) and, with
    ``need_dx``, dL/d(first layer's concatenated input) per stack — one fresh [B,H,W,C_in] buffer each, whose channel
    ranges the caller adds into the accumulators of the segments."""
    K = len(tapes)
    lay = [E.conv_layers(t["stack"]) for t in tapes]
    depth = len(lay[0])
    dz = list(d_outs)
    for d in range(depth - 1, -1, -1):
        ms = [lay[k][d][0] for k in range(K)]
        m0 = ms[0]
        g_conv = dz                                              # gradient at the convolution's own output grid
        if isinstance(m0, Ly.SubpelConv):                        # PixelShuffle backward
            g_conv = []
            for k in range(K):
                x0 = tapes[k]["x"][d][0]
                u = bw.buf(x0.B, x0.H, x0.W, 4 * dz[k].C)
                bw.call(lambda s=dz[k], u=u: ops.ps2_unshuffle(s, u), "pixel un-shuffle")
                g_conv.append(u)
        wg = []
        for k in range(K):
            c = ms[k][0] if isinstance(ms[k], Ly.SubpelConv) else ms[k]
            wg += ops.wgrad_problems(tapes[k]["x"][d], g_conv[k], grads[id(c.weight)], grads[id(c.bias)], stride=c.stride)
        bw.wgrad(wg)
        if d == 0 and not need_dx:
            return None
        gz = (lambda k: tapes[k]["z"][d - 1]) if d > 0 else (lambda k: None)     # pre-activation of the GELU in front of layer d
        if not isinstance(m0, Ly.SubpelConv) and m0.stride == 2:
            if m0.kernel_size == 5:                              # transposed convolution: four phase problems per stack
                das, probs = [], []
                for k in range(K):
                    x0 = tapes[k]["x"][d][0]
                    assert len(tapes[k]["x"][d]) == 1
                    o = bw.buf(x0.B, x0.H, x0.W, x0.C)
                    probs += [ops.conv_problem(p_, [g_conv[k]], o, gelu_z=gz(k)) for p_ in packs[k].d[id(ms[k])]]
                    das.append(o)
                bw.conv(probs)
            else:                                                # k3 s2: zero insertion + stride-1 data-gradient problem
                ups = []
                for k in range(K):
                    x0 = tapes[k]["x"][d][0]
                    u = bw.buf(x0.B, x0.H, x0.W, g_conv[k].C)
                    bw.call(lambda s=g_conv[k], u=u: ops.upsample2_zero(s, u), "zero insertion")
                    ups.append(u)
                das = [bw.buf(u.B, u.H, u.W, sum(v.C for v in tapes[k]["x"][d])) for k, u in enumerate(ups)]
                bw.conv([ops.conv_problem(packs[k].d[id(ms[k])], [ups[k]], das[k], gelu_z=gz(k)) for k in range(K)])
        else:
            das = [bw.buf(g_conv[k].B, g_conv[k].H, g_conv[k].W, sum(v.C for v in tapes[k]["x"][d])) for k in range(K)]
            bw.conv([ops.conv_problem(packs[k].d[id(ms[k])], [g_conv[k]], das[k], gelu_z=gz(k)) for k in range(K)])
        if d == 0:
            return das
        dz = das                          # the data-gradient launches applied gelu'(z) of the GELU in front of the layer
    return None


# ============================================================================= analysis transform with a tape
def _conv5_fwd(plan, pk, m: Ly.Conv2d, x: View, tape: list) -> View:
    if m.is_rgb_s2d:
        o = plan.buf(x.B, x.H, x.W, m.out_channels)              # x is the space-to-depth input: 3x3 s1 problem
    else:
        o = plan.buf(x.B, x.H // 2, x.W // 2, m.out_channels)
    plan.conv([ops.conv_problem(pk.f[id(m)], [x], o)])
    tape.append(dict(kind="conv5", mod=m, x=x))
    return o


def lower_g_a_train(plan: E.Plan, enc: nn.Sequential, x_s2d: View, y_out: View, pk: G.TransformPacks) -> list:
    """models/builder.py:43-53 keeping the tape (values as engine.lower_g_a up to fp32 rounding: activations are applied
    by the element-wise kernels with the epilogue's formulas)."""
    tape: list = []
    t = _conv5_fwd(plan, pk, enc[0], x_s2d, tape)
    t = G._gdn_fwd(plan, pk, enc[1], t, tape)
    t = _conv5_fwd(plan, pk, enc[2], t, tape)
    t = G._gdn_fwd(plan, pk, enc[3], t, tape)
    t = G._attention_block_fwd(plan, pk, enc[4], t, tape)
    t = _conv5_fwd(plan, pk, enc[5], t, tape)
    t = G._gdn_fwd(plan, pk, enc[6], t, tape)
    t = _conv5_fwd(plan, pk, enc[7], t, tape)
    G._attention_block_fwd(plan, pk, enc[8], t, tape, out=y_out)
    return tape


def _conv5_bwd(bw, pk, r: dict, dy: View, grads, keep: list) -> Optional[View]:
    m, x = r["mod"], r["x"]
    gw, gb = grads[id(m.weight)], grads[id(m.bias)]
    if m.is_rgb_s2d:
        # weight gradient w.r.t. the re-indexed 16-channel 3x3 tensor, then gathered back: w16[n,(py,px,c),ty,tx] =
        # w[n,c,2ty+py,2tx+px] (ops.pack_conv5s2_rgb); the image has no gradient
        n = m.out_channels
        t16 = torch.zeros((n, 16, 3, 3), dtype=torch.float32, device=gw.device)
        g6 = torch.zeros((n, 3, 6, 6), dtype=torch.float32, device=gw.device)
        keep += [t16, g6]
        def scatter():          # captured step: strided copies between pre-allocated buffers, no temporaries
            g6.view(n, 3, 3, 2, 3, 2).copy_(t16[:, :12].view(n, 2, 2, 3, 3, 3).permute(0, 3, 4, 1, 5, 2))
            gw.copy_(g6[:, :, :5, :5])
        with bw.off_path():
            bw.wgrad(ops.wgrad_problems([x], dy, t16, gb), now=True)      # t16 is re-indexed into gw right below
            bw.call(scatter, "first-layer weight gradient: s2d -> 5x5")
        return None
    bw.wgrad(ops.wgrad_problems([x], dy, gw, gb, stride=2))
    dx = bw.buf(x.B, x.H, x.W, x.C)
    bw.conv([ops.conv_problem(p_, [dy], dx) for p_ in pk.d[id(m)]])
    return dx


def lower_g_a_backward(bw: E.Plan, tape: list, d_y: View, pk: G.TransformPacks, grads):
    d: Optional[View] = d_y
    bw.defer_wgrad()                      # grouped by shape at the end of the transform (engine.Plan.flush_wgrad)
    for r in reversed(tape):
        if r["kind"] == "conv5":
            d = _conv5_bwd(bw, pk, r, d, grads, bw.keep)
        elif r["kind"] == "gdn":
            d = G._gdn_bwd(bw, pk, r, d, grads)
        else:
            d = G._attention_block_bwd(bw, pk, r, d, grads, need_dx=True)
    bw.flush_wgrad()


# ============================================================================= the plan
class FullTrainPlan:
    """Training forward + backward for one (B, H, W, mode).  ``mode`` = "multi": ``forward(x, [0, q])`` — both decoders,
    no clamp, likelihoods y / y_prog / z;  "single": ``forward_single_quality(x, q)`` — one decoder, clamp."""

    def __init__(self, m, B: int, H: int, W: int, mode: str, base_only: bool, device, trainable_ids: Optional[set] = None):
        assert mode in ("multi", "single") and not (mode == "multi" and base_only)
        delta, mu_rep, scal = bool(m.delta_encode), bool(m.total_mu_rep), bool(m.all_scalable)
        me, md, mh = bool(m.multiple_encoder), bool(m.multiple_decoder), bool(m.multiple_hyperprior)
        self.m, self.B, self.H, self.W, self.mode, self.base_only = m, B, H, W, mode, base_only
        self.device = torch.device(device)
        self.pr = 10.0
        self.generation = 0
        self.stream = None
        self.fwd_graphs: Dict[float, ops.Graph] = {}
        self.bwd_graphs: Dict[tuple, List[ops.Graph]] = {}
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        h, w, d, ns, C = H // 16, W // 16, m.division_dimension[0], m.ns0, m.dim_chunk
        sl = lambda v, i, n=1: v.window(i * C, n * C)
        nh = 1 if base_only else 2
        multi = mode == "multi"
        clamp = not multi
        self.clamp = clamp
        dec_base = multi or base_only
        dec_prog = not base_only

        # ------------------------------------------------------------------ modules on the path and their packs
        # single encoder: one g_a with M outputs (pic.py:306-307); single hyperprior: one synthesis pair with M outputs,
        # whatever the quality (pic.py:285-288); single decoder: ONE g_s reconstructs both levels (pic.py:372,462-466)
        self.enc = [m.g_a[0], m.g_a[1]] if me else [m.g_a]
        self.hs_m = [m.h_mean_s[k] for k in range(nh)] if mh else [m.h_mean_s]
        self.hs_s = [m.h_scale_s[k] for k in range(nh)] if mh else [m.h_scale_s]
        nhc = nh if mh else 2                                    # d-channel halves the hyper-synthesis tensors hold
        gs_base, gs_prog = (m.g_s[0], m.g_s[1]) if md else (m.g_s, m.g_s)
        base_st = [m.cc_mean_transforms[i] for i in range(ns)] + [m.cc_scale_transforms[i] for i in range(ns)] + \
                  [m.lrp_transforms[i] for i in range(ns)]
        prog_st = [] if base_only else ([m.cc_mean_transforms_prog[j] for j in range(ns)] +
                                        [m.cc_scale_transforms_prog[j] for j in range(ns)] +
                                        [m.lrp_transforms_prog[j] for j in range(ns)])
        self.decs = ([gs_base] if dec_base else []) + ([gs_prog] if (dec_prog and not (dec_base and gs_prog is gs_base)) else [])
        mods = self.enc + [m.h_a] + self.hs_m + self.hs_s + base_st + prog_st + self.decs
        self.pk: Dict[int, G.TransformPacks] = {id(mod): G.TransformPacks(mod) for mod in mods}
        pk = lambda mod: self.pk[id(mod)]

        # ------------------------------------------------------------------ forward plan
        P = self.plan = E.Plan(dev)
        pack_side = os.environ.get("VAMPIC_TRAIN_OVERLAP", "1") == "1"
        for mod in self.enc:                                     # the analysis transforms' packs first ...
            self.pk[id(mod)].record_refresh(P)
        if pack_side:                                            # ... everything else's beside g_a (joined in front of h_a)
            P.branch(WGRAD_BRANCH)
        for mod in mods[len(self.enc):]:
            self.pk[id(mod)].record_refresh(P)
        P.branch(0)
        self.x_in = torch.empty((B, 3, H, W), **f32)
        n_rec = (1 if dec_base else 0) + (1 if dec_prog else 0)
        self.x_hat = torch.empty((n_rec, B, 3, H, W), **f32)
        P.keep += [self.x_in, self.x_hat]
        x_s2d = P.buf(B, H // 2, W // 2, 16)
        P.call(lambda: L.check(L.load().vam_s2d_input(self.x_in.data_ptr(), x_s2d.ptr, B, H, W, ops.stream_ptr()), "vam_s2d_input"))
        y = self.y = P.buf(B, h, w, 2 * d)
        self.t_ga = [lower_g_a_train(P, e, x_s2d, y.window(k * d, d) if me else y, pk(e)) for k, e in enumerate(self.enc)]

        P.join(WGRAD_BRANCH)
        z = self.z = P.buf(B, h // 4, w // 4, m.N)
        self.t_ha = lower_stacks_train(P, [m.h_a], [[y]], [z], [pk(m.h_a)])
        self.z_hat, self.z_lik, self.noise_z = (P.buf(B, h // 4, w // 4, m.N) for _ in range(3))
        self.noise_y = P.buf(B, h, w, nh * d)
        eb = m.entropy_bottleneck
        self.eb_names = ["_matrix0", "_bias0", "_factor0", "_matrix1", "_bias1", "_factor1", "_matrix2", "_bias2", "_factor2",
                         "_matrix3", "_bias3", "_factor3", "_matrix4", "_bias4", "quantiles"]
        self.eb_params = torch.empty(sum(getattr(eb, n).numel() for n in self.eb_names), **f32)
        P.keep.append(self.eb_params)

        def eb_pack():          # refreshed every step: the optimiser moves the density network's tensors
            torch.cat([getattr(eb, n).detach().reshape(-1) for n in self.eb_names], out=self.eb_params)
        P.call(eb_pack, "entropy-bottleneck parameter block")
        P.call(lambda: ops.eb_forward(z, self.eb_params, self.z_hat, self.z_lik, None, noise=self.noise_z), "eb forward (noise)")
        means_h, scales_h = P.buf(B, h, w, nhc * d), P.buf(B, h, w, nhc * d)
        nst = len(self.hs_m)
        self.t_hs = lower_stacks_train(P, self.hs_m + self.hs_s, [[self.z_hat]] * (2 * nst),
                                       ([means_h.window(k * d, d) for k in range(nh)] + [scales_h.window(k * d, d) for k in range(nh)])
                                       if mh else [means_h, scales_h],
                                       [pk(s_) for s_ in self.hs_m + self.hs_s])
        mh0, sh0 = means_h.window(0, d), scales_h.window(0, d)

        # ---- base slices (pic.py:330-367)
        yq, yb = P.buf(B, h, w, d), P.buf(B, h, w, d)
        self.y_base, self.yq = yb, yq
        self.mu_b, self.std_b = P.buf(B, h, w, d), P.buf(B, h, w, d)
        self.lik = P.buf(B, h, w, nh * d)
        zero32 = P.buf(B, h, w, C, zero=True)
        junk = P.buf(B, h, w, d)
        self.base_groups = [[i] for i in range(min(ns, m.max_support_slices))] + \
                           ([list(range(m.max_support_slices, ns))] if ns > m.max_support_slices else [])
        self.t_base = []
        for idx in self.base_groups:
            sup = [sl(yb, 0, min(m.max_support_slices, idx[0]))] if idx[0] > 0 else []
            st = [m.cc_mean_transforms[i] for i in idx] + [m.cc_scale_transforms[i] for i in idx]
            t_ms = lower_stacks_train(P, st, [[mh0] + sup] * len(idx) + [[sh0] + sup] * len(idx),
                                      [sl(self.mu_b, i) for i in idx] + [sl(self.std_b, i) for i in idx], [pk(s_) for s_ in st])
            i0, n = idx[0], len(idx)
            P.call(lambda i0=i0, n=n: ops.gauss_tail(sl(y, i0, n), sl(self.mu_b, i0, n), sl(self.std_b, i0, n),
                                                     yhat=sl(yq, i0, n), lik=sl(junk, i0, n)), "quantise (base)")
            P.call(lambda i0=i0, n=n: ops.gauss_train(sl(y, i0, n), sl(self.mu_b, i0, n), sl(self.std_b, i0, n),
                                                      sl(self.noise_y, i0, n), lik=sl(self.lik, i0, n)), "noise likelihood (base)")
            lst = [m.lrp_transforms[i] for i in idx]
            t_l = lower_stacks_train(P, lst, [[mh0] + sup + [sl(yq, i)] for i in idx], [None] * n, [pk(s_) for s_ in lst])
            for k, i in enumerate(idx):
                P.call(lambda k=k, i=i, t_l=t_l: ops.ew(L.EW_HTANH_FWD, [t_l[k]["out"], sl(yq, i), zero32], [sl(yb, i)]), "lrp tail")
            self.t_base.append(dict(idx=idx, sup=sup, ms=t_ms, lrp=t_l))
        # The base reconstruction needs y_hat_base only and nothing of this step needs it: with both levels on the path it
        # runs on a branch of its own beside the progressive slice chain (ten dependent stacks of five small launches that
        # leave most of the chip idle) — forward here, backward below.  VAMPIC_TRAIN_OVERLAP=0: one stream as in round 3.
        overlap = dec_base and dec_prog and os.environ.get("VAMPIC_TRAIN_OVERLAP", "1") == "1"
        if dec_base:
            if overlap:
                ev = P.record()
                P.branch(GS_BASE_BRANCH)
                P.wait(ev)
            self.t_gs0 = G.lower_g_s_train(P, gs_base, yb, self.x_hat[0], pk(gs_base), clamp=clamp)
            P.branch(0)

        # ---- progressive slices (pic.py:396-457)
        if dec_prog:
            mh1, sh1 = means_h.window(d, d), scales_h.window(d, d)
            self.mu_p, self.std_p = P.buf(B, h, w, d), P.buf(B, h, w, d)
            mu_tot = P.buf(B, h, w, d) if mu_rep else self.mu_p      # pic.py:416: mu_total = mu (+ y_hat_base with total_mu_rep)
            sp = m.support_progressive_slices
            y_top, y_sub = y.window(d, d), (y.window(0, d) if delta else None)      # pic.py:397-398: r = y_top - y_base under delta_encode
            self.t_chain = []
            self.mask = P.buf(B, h, w, d)
            rq = self.rq = P.buf(B, h, w, d)
            yp = self.y_prog = P.buf(B, h, w, d)
            nz_p = self.noise_y.window(d, d)
            lst = [m.lrp_transforms_prog[j] for j in range(ns)]
            # all_scalable (pic.py:400-401): the supports of slice j are mu_total / std_total of the slices before it, so the
            # (mu, sigma) chain runs through all ten slices before anything is quantised; without it they are the DECODED
            # slices, and each slice is finished (mask, quantisation, LRP, merge) before the next one's stacks run
            sup_m, sup_s = (mu_tot, self.std_p) if scal else (yp, yp)
            self.t_lrp_p = []
            for j in range(ns):
                s_ = min(sp, j)
                ms = [mh1, sl(yb, j)] + ([sl(sup_m, j - s_, s_)] if s_ else [])
                ss = [sh1, sl(yb, j)] + ([sl(sup_s, j - s_, s_)] if s_ else [])
                st = [m.cc_mean_transforms_prog[j], m.cc_scale_transforms_prog[j]]
                t_ = lower_stacks_train(P, st, [ms, ss], [sl(self.mu_p, j), sl(self.std_p, j)], [pk(s) for s in st])
                if mu_rep and scal:
                    P.call(lambda j=j: ops.add(sl(self.mu_p, j), sl(yb, j), sl(mu_tot, j)), "mu_total")   # pic.py:416
                self.t_chain.append(dict(j=j, s=s_, t=t_, msup=ms))
                if not scal:
                    P.call(lambda j=j: ops.variance_mask(sl(self.std_p, j), self.pr, sl(self.mask, j), n_slice=1), "variance mask")
                    P.call(lambda j=j: ops.gauss_tail(sl(y_top, j), sl(self.mu_p, j), sl(self.std_p, j), y2=sl(y_sub, j) if delta else None,
                                                      mask=sl(self.mask, j), yhat=sl(rq, j), lik=sl(junk, j)), "quantise (prog)")
                    (tl,) = lower_stacks_train(P, [lst[j]], [ms + [sl(rq, j)]], [None], [pk(lst[j])])
                    self.t_lrp_p.append(tl)
                    P.call(lambda j=j, tl=tl: ops.ew(L.EW_HTANH_FWD, [tl["out"], sl(rq, j), sl(yb, j)], [sl(yp, j)]), "lrp tail (prog)")
            if scal:
                P.call(lambda: ops.variance_mask(self.std_p, self.pr, self.mask, n_slice=ns), "variance mask")   # pic.py:430
                P.call(lambda: ops.gauss_tail(y_top, self.mu_p, self.std_p, y2=y_sub, mask=self.mask, yhat=rq, lik=junk), "quantise (prog)")
            P.call(lambda: ops.gauss_train(y_top, self.mu_p, self.std_p, nz_p, y2=y_sub, mask=self.mask, lik=self.lik.window(d, d)),
                   "noise likelihood (prog)")
            if scal:
                self.t_lrp_p = lower_stacks_train(P, lst, [self.t_chain[j]["msup"] + [sl(rq, j)] for j in range(ns)], [None] * ns,
                                                  [pk(s) for s in lst])
                for j in range(ns):
                    P.call(lambda j=j: ops.ew(L.EW_HTANH_FWD, [self.t_lrp_p[j]["out"], sl(rq, j), sl(yb, j)], [sl(yp, j)]), "lrp tail (prog)")
            self.t_gs1 = G.lower_g_s_train(P, gs_prog, yp, self.x_hat[n_rec - 1], pk(gs_prog), clamp=clamp)

        # ------------------------------------------------------------------ parameters and the flat gradient buffer
        # order = the order in which the backward FINISHES them (so that buckets complete front to back)
        order: List[nn.Parameter] = []
        add = lambda mod: order.extend(mod.parameters())
        shared_dec = dec_prog and dec_base and gs_prog is gs_base     # one decoder, two passes: final after the second
        if dec_prog:
            if not shared_dec:
                add(gs_prog)
            if scal:
                for j in range(ns):
                    add(m.lrp_transforms_prog[j])
            for j in range(ns - 1, -1, -1):
                if not scal:
                    add(m.lrp_transforms_prog[j])
                add(m.cc_mean_transforms_prog[j])
                add(m.cc_scale_transforms_prog[j])
        if dec_base:
            add(gs_base)
        for idx in reversed(self.base_groups):
            for i in idx:
                add(m.lrp_transforms[i])
            for i in idx:
                add(m.cc_mean_transforms[i])
            for i in idx:
                add(m.cc_scale_transforms[i])
        for s_ in self.hs_m + self.hs_s:
            add(s_)
        eb_ps = [getattr(eb, n) for n in self.eb_names]
        order.extend(eb_ps)
        add(m.h_a)
        for e in self.enc:
            add(e)
        self.params = order
        self.trainable_ids = trainable_ids
        offs, tot = [], 0
        for p in order:
            offs.append(tot)
            tot += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(tot, **f32)
        self.views = [self.flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, order)]
        self.offsets = offs
        grads = {id(p): g for p, g in zip(order, self.views)}

        # ------------------------------------------------------------------ backward plan
        bw = self.bwd = E.Plan(dev)
        # weight gradients on a branch of their own (engine.Plan.wgrad_branch).  VAMPIC_WGRAD_SIDE: 0 = one stream as in
        # round 3, 1 = the entropy-parameter / LRP / hyperprior stacks only, 2 = every transform, 3 = also the GDN / bias /
        # first-layer gradient sub-chains (engine.Plan.off_path)
        side = int(os.environ.get("VAMPIC_WGRAD_SIDE", WGRAD_SIDE_DEFAULT))
        bw.wgrad_branch = WGRAD_BRANCH if side >= 1 else None
        bw.wgrad_units = side >= 3
        bw.wgrad_branches = int(os.environ.get("VAMPIC_WGRAD_BRANCHES", "1"))

        def transform_bwd(fn, *a, **kw):            # g_a / g_s: their launches fill the chip on their own
            keep_ = bw.wgrad_branch
            if side < 2:
                bw.wgrad_branch = None
            try:
                return fn(*a, **kw)
            finally:
                bw.wgrad_branch = keep_
        bw.keep += [self.flat, self.views]
        self.g_xhat = torch.zeros_like(self.x_hat)
        self.glik = bw.buf(B, h, w, nh * d, zero=True)
        self.glik_z = bw.buf(B, h // 4, w // 4, m.N, zero=True)
        bw.keep.append(self.g_xhat)
        D_y = bw.buf(B, h, w, 2 * d, zero=True)
        D_yb = bw.buf(B, h, w, d, zero=True)
        D_mh, D_sh = bw.buf(B, h, w, nhc * d, zero=True), bw.buf(B, h, w, nhc * d, zero=True)
        accs = [D_y, D_yb, D_mh, D_sh]
        if dec_prog:
            D_mutot, D_stdp = bw.buf(B, h, w, d, zero=True), bw.buf(B, h, w, d, zero=True)
            accs += [D_mutot, D_stdp]
        bw.call(lambda: [ops.memset_zero(a.buf) for a in accs], "clear the gradient accumulators")
        self.param_done: Dict[int, int] = {}

        def done(mod_or_params):
            ps = mod_or_params.parameters() if isinstance(mod_or_params, nn.Module) else mod_or_params
            for p in ps:
                self.param_done[id(p)] = len(bw.steps)

        def acc(dst: View, src: View, coef: float = 1.0):
            bw.call(lambda: ops.ew(L.EW_AXPY, [dst, src], [dst], coef=coef), "accumulate")

        def acc_many(updates: Sequence[tuple], desc: str):
            """[(dst, src, coef)]: dst += coef * src on windows that do not overlap each other — one launch."""
            if os.environ.get("VAMPIC_AXPY_GROUP", "1") != "1":          # A/B arm: one launch per update, as in round 3
                for dst, src, coef in updates:
                    acc(dst, src, coef)
                return
            for i in range(0, len(updates), L.VAM_MAX_EW_GROUP):
                arr = ops.axpy_jobs(updates[i:i + L.VAM_MAX_EW_GROUP])
                bw.keep.append(arr)
                bw.call(lambda arr=arr: ops.axpy_group(arr), desc)

        def scatter(dx: View, segs: Sequence[tuple]):
            """Add the channel ranges of a first-layer input gradient into the accumulators: segs = [(accumulator view or
            None, channels)] in concatenation order (the backward of the torch.cat in front of the stack; the accumulators
            of one call are different tensors or different channel ranges: one launch)."""
            off, ups = 0, []
            for dst, c in segs:
                if dst is not None:
                    ups.append((dst, dx.window(off, c), 1.0))
                off += c
            assert off == dx.C
            if ups:
                acc_many(ups, "scatter of a first-layer input gradient")

        def gs_base_bwd():
            g0 = grads
            if shared_dec:
                # the same parameters took part in the progressive pass above: this pass writes its gradients into a scratch
                # copy of the decoder's range of the flat buffer, and ONE element-wise launch adds the two
                ps = list(gs_base.parameters())
                pos = {id(p_): i for i, p_ in enumerate(order)}
                lo_ = offs[pos[id(ps[0])]]
                hi_ = offs[pos[id(ps[-1])]] + (ps[-1].numel() + 3) // 4 * 4
                tmp = torch.zeros(hi_ - lo_, **f32)
                bw.keep.append(tmp)
                g0 = dict(grads)
                for p_ in ps:
                    o_ = offs[pos[id(p_)]] - lo_
                    g0[id(p_)] = tmp[o_:o_ + p_.numel()].view(p_.shape)
                self._shared_sum = (self.flat[lo_:hi_], tmp)
            return transform_bwd(G.lower_g_s_backward, bw, self.t_gs0, self.x_hat[0], self.g_xhat[0], pk(gs_base), g0,
                                 need_input_grad=True, clamp=clamp)

        d_yb0 = None
        if dec_prog:
            g1 = self.g_xhat[n_rec - 1]
            d_yp = transform_bwd(G.lower_g_s_backward, bw, self.t_gs1, self.x_hat[n_rec - 1], g1, pk(gs_prog), grads, need_input_grad=True, clamp=clamp)
            if not shared_dec:
                done(gs_prog)
            if overlap:                             # the base decoder's backward beside the progressive chain's (see the forward)
                ev = bw.record()
                bw.branch(GS_BASE_BRANCH)
                bw.wait(ev)
                d_yb0 = gs_base_bwd()
                bw.branch(0)
            if scal:
                acc(D_yb, d_yp)                                                       # merge: y_hat = r_hat + y_hat_base (pic.py:451)
                dzl = []
                for j in range(ns):
                    zt = self.t_lrp_p[j]["out"]
                    o = bw.buf(zt.B, zt.H, zt.W, zt.C)
                    bw.call(lambda j=j, zt=zt, o=o: ops.ew(L.EW_HTANH_BWD, [zt, sl(d_yp, j)], [o]), "lrp tail bwd")
                    dzl.append(o)
                lst = [m.lrp_transforms_prog[j] for j in range(ns)]
                dxs = lower_stacks_backward(bw, self.t_lrp_p, dzl, [pk(s) for s in lst], grads)
                for s in lst:
                    done(s)
                d_rq = bw.buf(B, h, w, d)
                for j in range(ns):
                    s_ = self.t_chain[j]["s"]
                    scatter(dxs[j], [(D_mh.window(d, d), d), (sl(D_yb, j), C)] + ([(sl(D_mutot, j - s_, s_), C * s_)] if s_ else []) + [(None, C)])
                    bw.call(lambda j=j, dx=dxs[j]: ops.ew(L.EW_AXPY, [sl(d_yp, j), dx.window(dx.C - C, C)], [sl(d_rq, j)], coef=1.0), "d rq")
                # likelihood + straight-through rounding of the ten progressive slices, one launch each
                dmu_l, dsg_l = bw.buf(B, h, w, d), bw.buf(B, h, w, d)
                bw.call(lambda: ops.gauss_train(y_top, self.mu_p, self.std_p, nz_p, y2=y_sub, mask=self.mask,
                                                grad_lik=self.glik.window(d, d), dmu=dmu_l, dsigma=dsg_l), "likelihood backward (prog)")
                d_r, G_mu = bw.buf(B, h, w, d), bw.buf(B, h, w, d)
                bw.call(lambda: ops.ew(L.EW_MASK_SPLIT, [d_rq, self.mask], [d_r, G_mu]), "straight-through rounding under the mask")
                acc(G_mu, dmu_l)                                                          # dL/dmu_p before the chain
                acc(d_r, dmu_l, -1.0)                                                     # dL/dr = d rq * m - dmu_lik
                acc(D_y.window(d, d), d_r)
                if delta:
                    acc(D_y.window(0, d), d_r, -1.0)                                      # delta_encode: r = y_top - y_sub
                for j in range(ns - 1, -1, -1):
                    rec = self.t_chain[j]
                    s_ = rec["s"]
                    acc_many([(sl(G_mu, j), sl(D_mutot, j), 1.0)] +                      # mu_total_j = mu_j + y_hat_base_j
                             ([(sl(D_yb, j), sl(D_mutot, j), 1.0)] if mu_rep else []) +
                             [(sl(dsg_l, j), sl(D_stdp, j), 1.0)], "supports' gradients of slice j")
                    st = [m.cc_mean_transforms_prog[j], m.cc_scale_transforms_prog[j]]
                    dxm, dxs_ = lower_stacks_backward(bw, rec["t"], [sl(G_mu, j), sl(dsg_l, j)], [pk(s) for s in st], grads)
                    done(st[0])
                    done(st[1])
                    scatter(dxm, [(D_mh.window(d, d), d), (sl(D_yb, j), C)] + ([(sl(D_mutot, j - s_, s_), C * s_)] if s_ else []))
                    scatter(dxs_, [(D_sh.window(d, d), d), (sl(D_yb, j), C)] + ([(sl(D_stdp, j - s_, s_), C * s_)] if s_ else []))
            else:
                # all_scalable off: slice j's stacks and LRP read the DECODED slices before it, so dL/dy_hat_j collects the
                # input gradients of the later slices before slice j's own backward runs (d_yp is accumulated in place:
                # it is the data gradient g_s handed back, no weight gradient reads it)
                dmu_l, dsg_l = bw.buf(B, h, w, d), bw.buf(B, h, w, d)
                bw.call(lambda: ops.gauss_train(y_top, self.mu_p, self.std_p, nz_p, y2=y_sub, mask=self.mask,
                                                grad_lik=self.glik.window(d, d), dmu=dmu_l, dsigma=dsg_l), "likelihood backward (prog)")
                d_rq, d_r, G_mu = bw.buf(B, h, w, d), bw.buf(B, h, w, d), bw.buf(B, h, w, d)
                for j in range(ns - 1, -1, -1):
                    rec = self.t_chain[j]
                    s_ = rec["s"]
                    hist = [(sl(d_yp, j - s_, s_), C * s_)] if s_ else []
                    acc(sl(D_yb, j), sl(d_yp, j))                                     # merge: y_hat_j = r_hat_j + lrp_j + y_hat_base_j
                    zt = self.t_lrp_p[j]["out"]
                    dz = bw.buf(zt.B, zt.H, zt.W, zt.C)
                    bw.call(lambda j=j, zt=zt, dz=dz: ops.ew(L.EW_HTANH_BWD, [zt, sl(d_yp, j)], [dz]), "lrp tail bwd")
                    (dx,) = lower_stacks_backward(bw, [self.t_lrp_p[j]], [dz], [pk(m.lrp_transforms_prog[j])], grads)
                    done(m.lrp_transforms_prog[j])
                    scatter(dx, [(D_mh.window(d, d), d), (sl(D_yb, j), C)] + hist + [(None, C)])
                    bw.call(lambda j=j, dx=dx: ops.ew(L.EW_AXPY, [sl(d_yp, j), dx.window(dx.C - C, C)], [sl(d_rq, j)], coef=1.0), "d rq")
                    bw.call(lambda j=j: ops.ew(L.EW_MASK_SPLIT, [sl(d_rq, j), sl(self.mask, j)], [sl(d_r, j), sl(G_mu, j)]),
                            "straight-through rounding under the mask")
                    acc(sl(G_mu, j), sl(dmu_l, j))
                    acc(sl(d_r, j), sl(dmu_l, j), -1.0)
                    acc(sl(D_y.window(d, d), j), sl(d_r, j))
                    if delta:
                        acc(sl(D_y.window(0, d), j), sl(d_r, j), -1.0)
                    st = [m.cc_mean_transforms_prog[j], m.cc_scale_transforms_prog[j]]
                    dxm, dxs_ = lower_stacks_backward(bw, rec["t"], [sl(G_mu, j), sl(dsg_l, j)], [pk(s) for s in st], grads)
                    done(st[0])
                    done(st[1])
                    scatter(dxm, [(D_mh.window(d, d), d), (sl(D_yb, j), C)] + hist)
                    scatter(dxs_, [(D_sh.window(d, d), d), (sl(D_yb, j), C)] + hist)
        if dec_base:
            if d_yb0 is None:
                d_yb0 = gs_base_bwd()
            else:
                bw.join(GS_BASE_BRANCH)
            if shared_dec:
                bw.join_wgrad()                                 # both passes' weight gradients are final
                main, tmp = self._shared_sum
                bw.call(lambda: ops.ew(L.EW_AXPY, [ops.flat_view(main), ops.flat_view(tmp)], [ops.flat_view(main)], coef=1.0),
                        "single decoder: sum of the two passes' gradients")
            done(gs_base)
            acc(D_yb, d_yb0)
        # ---- base slices, last group first
        dmu_b, dsg_b = bw.buf(B, h, w, d), bw.buf(B, h, w, d)
        nz_b = self.noise_y.window(0, d)
        y_b = y.window(0, d)
        bw.call(lambda: ops.gauss_train(y_b, self.mu_b, self.std_b, nz_b, grad_lik=self.glik.window(0, d), dmu=dmu_b, dsigma=dsg_b),
                "likelihood backward (base)")
        acc(D_y.window(0, d), dmu_b, -1.0)                                            # lik at y + noise - mu: dy = -dmu
        for rec in reversed(self.t_base):
            idx, sup = rec["idx"], rec["sup"]
            n_sup = sup[0].C if sup else 0
            dzl = []
            for k, i in enumerate(idx):
                zt = rec["lrp"][k]["out"]
                o = bw.buf(zt.B, zt.H, zt.W, zt.C)
                bw.call(lambda i=i, zt=zt, o=o: ops.ew(L.EW_HTANH_BWD, [zt, sl(D_yb, i)], [o]), "lrp tail bwd")
                dzl.append(o)
            lst = [m.lrp_transforms[i] for i in idx]
            dxs = lower_stacks_backward(bw, rec["lrp"], dzl, [pk(s) for s in lst], grads)
            for s in lst:
                done(s)
            for k, i in enumerate(idx):
                # y_hat_i = yq_i + 0.5 tanh(.): d yq_i = d y_hat_i + (first-layer segment); yq = ste_round(y - mu) + mu: dy = d yq
                scatter(dxs[k], [(D_mh.window(0, d), d)] + ([(sl(D_yb, 0, n_sup // C), n_sup)] if n_sup else []) + [(sl(D_y, i), C)])
                acc(sl(D_y, i), sl(D_yb, i))
            st = [m.cc_mean_transforms[i] for i in idx] + [m.cc_scale_transforms[i] for i in idx]
            dxs = lower_stacks_backward(bw, rec["ms"], [sl(dmu_b, i) for i in idx] + [sl(dsg_b, i) for i in idx], [pk(s) for s in st],
                                        grads)
            for s in st:
                done(s)
            for k in range(len(idx)):
                scatter(dxs[k], [(D_mh.window(0, d), d)] + ([(sl(D_yb, 0, n_sup // C), n_sup)] if n_sup else []))
                scatter(dxs[len(idx) + k], [(D_sh.window(0, d), d)] + ([(sl(D_yb, 0, n_sup // C), n_sup)] if n_sup else []))
        # ---- hyperprior
        hs = self.hs_m + self.hs_s
        dxs = lower_stacks_backward(bw, self.t_hs, ([D_mh.window(k * d, d) for k in range(nh)] + [D_sh.window(k * d, d) for k in range(nh)])
                                    if mh else [D_mh, D_sh], [pk(s) for s in hs], grads)
        for s in hs:
            done(s)
        D_z = bw.buf(B, h // 4, w // 4, m.N)
        self.eb_dparams = torch.zeros_like(self.eb_params)
        bw.keep.append(self.eb_dparams)
        bw.call(lambda: ops.eb_train_bwd(z, self.noise_z, self.eb_params, self.glik_z, D_z, self.eb_dparams), "entropy bottleneck backward")

        def eb_unpack():
            off = 0
            for p in eb_ps:
                torch.mul(self.eb_dparams[off:off + p.numel()].view(p.shape), 1.0, out=grads[id(p)])
                off += p.numel()
        bw.call(eb_unpack, "entropy-bottleneck gradients")
        done(eb_ps)
        for dx in dxs:                                                                # z_hat = ste_round(z - med) + med: dz = dz_hat
            acc(D_z, dx)
        (dxy,) = lower_stacks_backward(bw, self.t_ha, [D_z], [pk(m.h_a)], grads)
        done(m.h_a)
        acc(D_y, dxy)
        for k, e in enumerate(self.enc):
            transform_bwd(lower_g_a_backward, bw, self.t_ga[k], D_y.window(k * d, d) if me else D_y, pk(e), grads)
            done(e)
        # ------------------------------------------------------------------ buckets
        from .sharding import bucket_partition
        self.bucket_bounds, self.bucket_ready = bucket_partition(offs, [p.numel() for p in order],
                                                                 [self.param_done[id(p)] for p in order], tot, len(bw.steps),
                                                                 BUCKET_BYTES)

    # ------------------------------------------------------------------------------------------- execution
    def _own_stream(self):
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.device)
        return self.stream

    def close(self):
        for g in list(self.fwd_graphs.values()) + [g for gs in self.bwd_graphs.values() for g in gs]:
            g.close()
        self.fwd_graphs.clear()
        self.bwd_graphs.clear()

    def set_noise(self, noise=None):
        for key, v in (("y", self.noise_y), ("z", self.noise_z)):
            if noise is not None and key in noise:
                src = noise[key].to(v.buf.device)
                v.buf.copy_(src[:, :v.C].permute(0, 2, 3, 1))
            else:
                v.buf.uniform_(-0.5, 0.5)

    def execute(self, x: torch.Tensor, pr: float, use_graph: bool, noise=None) -> dict:
        self.pr = float(pr)
        self.generation += 1
        sig = tuple(p.data_ptr() for p in self.params)
        if getattr(self, "_ptr_sig", sig) != sig:          # parameter storage replaced: captured pointers are stale
            self.close()
        self._ptr_sig = sig
        ops.drain_graveyard()
        cur = torch.cuda.current_stream(self.device)
        st = self._own_stream()
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            self.x_in.copy_(x)
            self.set_noise(noise)
            if use_graph:
                g = self.fwd_graphs.get(self.pr)
                if g is None:
                    self.plan.run()
                    st.synchronize()
                    g = ops.Graph()
                    g.capture(self.plan.run)
                    self.fwd_graphs[self.pr] = g
                g.launch()
            else:
                self.plan.run()
        cur.wait_stream(st)
        d = self.m.division_dimension[0]
        nchw = lambda v: v.torch_nchw().clone()
        out = {"x_hat": self.x_hat.clone(), "lik": nchw(self.lik), "z_lik": nchw(self.z_lik), "y_base": nchw(self.y_base),
               "mu_base": nchw(self.mu_b), "std_base": nchw(self.std_b)}
        if not self.base_only:
            out.update({"y_prog": nchw(self.y_prog), "mask": nchw(self.mask), "mu": nchw(self.mu_p), "std": nchw(self.std_p)})
        return out

    def backward(self, g_xhat: Optional[torch.Tensor], g_lik: Optional[torch.Tensor], g_z: Optional[torch.Tensor], use_graph: bool,
                 reducer=None) -> List[torch.Tensor]:
        """Run the backward plan for dL/dx_hat [n_rec,B,3,H,W], dL/dlik [B, n*d, h, w] (NCHW) and dL/dlik_z; returns the
        parameter gradients (views of the flat buffer, ``self.params`` order).  ``reducer`` (sharding.BucketReducer):
        called with (bucket index, flat slice, stream) as each bucket becomes final."""
        cur = torch.cuda.current_stream(self.device)
        st = self._own_stream()
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            if g_xhat is None:
                self.g_xhat.zero_()
            else:
                self.g_xhat.copy_(g_xhat)
            for buf, g in ((self.glik, g_lik), (self.glik_z, g_z)):
                if g is None:
                    buf.buf.zero_()
                else:
                    buf.buf.copy_(g.permute(0, 2, 3, 1))
            cuts = [0] + (list(self.bucket_ready) if reducer is not None else [len(self.bwd.steps)])
            cuts = sorted(set(cuts))
            key = tuple(cuts)
            graphs = self.bwd_graphs.get(key) if use_graph else None
            if use_graph and graphs is None:
                self._run_bwd_segment(0, len(self.bwd.steps))                      # warm-up (code objects loaded before capture)
                st.synchronize()
                graphs = []
                for a, b in zip(cuts[:-1], cuts[1:]):
                    g = ops.Graph()
                    g.capture(lambda a=a, b=b: self._run_bwd_segment(a, b))
                    graphs.append(g)
                self.bwd_graphs[key] = graphs
            nb = 0
            for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                if use_graph:
                    graphs[k].launch()
                else:
                    self._run_bwd_segment(a, b)
                if reducer is not None:
                    while nb < len(self.bucket_ready) and self.bucket_ready[nb] <= b:
                        lo, hi = self.bucket_bounds[nb]
                        reducer(nb, self.flat[lo:hi], st)
                        nb += 1
            if reducer is not None:
                reducer.finish(st)
        cur.wait_stream(st)
        return self.views

    def _run_bwd_segment(self, a: int, b: int):
        self.bwd.run_range(a, b)        # (joins the weight-gradient branch at the end: a segment ends where a bucket is final)
