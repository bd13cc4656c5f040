"""Flip audit for end-to-end parity of ``forward_single_quality`` (TEST INFRASTRUCTURE).

The HIP path and the CPU reference sum their convolutions in different orders, so y - mu differs by ~1e-5 of its
magnitude.  A latent whose residual lies that close to a rounding boundary (x.5) gets the other symbol, a sigma that
close to the quantile threshold gets the other mask bit, and every slice conditioned on that one then differs
legitimately.  "Legitimately" is checked here instead of being assumed:

* slices are visited in dependency order (models/pic.py:522-554: base slice i reads base slices 0..min(i,5)-1;
  pic.py:577-643: progressive slice j reads base slice j and the progressive (mu, sigma) chain of j-5..j-1, hence base
  slices 0..j; with all_scalable the progressive symbols feed nothing but their own LRP);
* in a slice none of whose inputs has flipped yet ("untainted"), EVERY differing symbol must be a boundary event in the
  ORACLE's own numbers:  0.5 - |t - round(t)| < BOUNDARY_TOL  for t = y - mu (base) or (y_top - y_base) - mu (progressive),
  and every differing mask bit must have  |sigma - threshold| < THRESH_TOL * max(1, |sigma|) ;
* differences in tainted slices are downstream of a verified boundary event and are only counted.

A wrong kernel produces differences that are not boundary events in an untainted slice (a uniformly random residual
lies within BOUNDARY_TOL of a boundary with probability 2e-3), so it cannot pass as "fp32 noise".
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

import vampic_oracle as O

BOUNDARY_TOL = 1e-3      # distance of the oracle's residual from x.5, in latent units (|y - mu| reaches ~50)
THRESH_TOL = 2e-4        # relative distance of the oracle's sigma from the quantile threshold


def audit(y_gpu: torch.Tensor, out: Dict[str, torch.Tensor], ref: Dict[str, torch.Tensor], q: float, *,
          div: int = 320, chunk: int = 32, max_support: int = 5, all_scalable: bool = True, delta_encode: bool = True,
          att_gpu: torch.Tensor = None) -> dict:
    """``y_gpu``: the HIP path's latent y [B, 2*div, h, w]; ``out``: its output dict (cpu tensors); ``ref``: the oracle's
    dict (holds "y").  Returns counts plus ``violations`` (list of strings; empty = every difference is explained).
    REM variants (``ref`` holds "att" / "std_raw"): the REM's attention mask is a SECOND threshold decision, taken on the
    un-refined sigma (rem_pic.py:185-192) — ``att_gpu`` (the HIP plan's attention mask) is audited by the same rule, and a
    slice with a (proven) attention-mask event counts as tainted: its refined (mu, sigma) legitimately differ."""
    ns = div // chunk
    sl = lambda t, i: t[:, i * chunk:(i + 1) * chunk]
    yg, yo = y_gpu.float().cpu(), ref["y"]
    rep = {"sym_flips": 0, "mask_flips": 0, "first": None, "violations": [], "explained": 0, "downstream": 0}

    def boundary_check(tag, t_o, differ):
        n = int(differ.sum())
        if n == 0:
            return
        dist = 0.5 - (t_o - torch.round(t_o)).abs()
        bad = differ & (dist >= BOUNDARY_TOL)
        rep["explained"] += n - int(bad.sum())
        if bad.any():
            rep["violations"].append(f"{tag}: {int(bad.sum())} of {n} differing symbols are NOT rounding-boundary events "
                                     f"(largest distance from x.5: {float(dist[differ].max()):.3e})")

    # ---- base slices
    flipped_b: List[bool] = []
    for i in range(ns):
        t_g = sl(yg, i) - sl(out["mu_base"], i)
        t_o = sl(yo, i) - sl(ref["mu_base"], i)
        differ = torch.round(t_g) != torch.round(t_o)
        n = int(differ.sum())
        rep["sym_flips"] += n
        deps = range(min(i, max_support))
        tainted = any(flipped_b[k] for k in deps)
        if n and rep["first"] is None:
            rep["first"] = f"base {i}"
        if tainted:
            rep["downstream"] += n
        else:
            boundary_check(f"base slice {i}", t_o, differ)
        flipped_b.append(n > 0)
    if "mask" not in ref:
        return rep
    # ---- progressive slices
    prog_flipped = False
    for j in range(ns):
        r_g = sl(yg, ns + j) - sl(yg, j) if delta_encode else sl(yg, ns + j)
        r_o = sl(yo, ns + j) - sl(yo, j) if delta_encode else sl(yo, ns + j)
        m_g, m_o = sl(out["mask"], j), sl(ref["mask"], j)
        s_o = sl(ref["std"], j)
        mdiff = m_g != m_o
        t_g = r_g - sl(out["mu"], j)
        t_o = r_o - sl(ref["mu"], j)
        sdiff = (torch.round(t_g) * m_g != torch.round(t_o) * m_o) & ~mdiff
        nm, nsym = int(mdiff.sum()), int(sdiff.sum())
        rep["mask_flips"] += nm
        rep["sym_flips"] += nsym
        if (nm or nsym) and rep["first"] is None:
            rep["first"] = f"prog {j}"
        tainted = any(flipped_b[k] for k in range(j + 1))
        if "att" in ref and att_gpu is not None and not tainted:
            adiff = sl(att_gpu, j) != sl(ref["att"], j)
            if adiff.any():
                rep.setdefault("att_flips", 0)
                rep["att_flips"] += int(adiff.sum())
                sr = sl(ref["std_raw"], j)
                for b in range(sr.shape[0]):
                    if not adiff[b].any():
                        continue
                    thr = float(O.quantile_threshold_np(sr[b].numpy().ravel(), min(q, 10) * 0.1))
                    sv = sr[b][adiff[b]]
                    bad = (sv - thr).abs() >= THRESH_TOL * torch.clamp(sv.abs(), min=1.0)
                    rep["explained"] += int(adiff[b].sum()) - int(bad.sum())
                    if bad.any():
                        rep["violations"].append(f"prog slice {j} image {b}: {int(bad.sum())} differing REM attention-mask bits are "
                                                 f"NOT threshold events of the un-refined sigma (thr {thr:.6g})")
                tainted = True                      # the refined entropy parameters of this slice follow the attention mask
        if not all_scalable:          # pic.py:586-587: the (mu, sigma) stacks read the decoded progressive slices before j
            tainted = tainted or prog_flipped
            prog_flipped = prog_flipped or bool(nm or nsym)
        if tainted:
            rep["downstream"] += nm + nsym
            continue
        if nm and 0 < q < 10:
            for b in range(s_o.shape[0]):
                if not mdiff[b].any():
                    continue
                thr = float(O.quantile_threshold_np(s_o[b].numpy().ravel(), min(q, 10) * 0.1))     # q_keep = pr * 0.1
                sv = s_o[b][mdiff[b]]
                bad = (sv - thr).abs() >= THRESH_TOL * torch.clamp(sv.abs(), min=1.0)
                rep["explained"] += int(mdiff[b].sum()) - int(bad.sum())
                if bad.any():
                    rep["violations"].append(f"prog slice {j} image {b}: {int(bad.sum())} differing mask bits are NOT "
                                             f"threshold events (thr {thr:.6g}, sigma {sv[bad][:3].tolist()})")
        elif nm:
            rep["violations"].append(f"prog slice {j}: mask differs at q={q} where it is all-zero / all-one")
        boundary_check(f"prog slice {j}", t_o, sdiff)
    return rep


def gpu_latent(net, B: int, H: int, W: int, base_only: bool, rem_idx=None) -> torch.Tensor:
    """The latent y of the most recent eval plan of this shape (NCHW view of the plan's buffer)."""
    for k, p in net._plans.items():
        if k[:5] == (B, H, W, base_only, rem_idx) and len(k) == 6:        # no symbols / train / own_ck suffix
            return p.y.torch_nchw().cpu()
    raise KeyError((B, H, W, base_only, rem_idx))
