#!/usr/bin/env python3
"""Numerical probe (CPU, numpy) for a round-4 candidate: fp32 operands split into TWO fp16 terms with a power-of-two
per-tensor scale (3 MFMA products hh + hl + lh) instead of THREE bf16 terms (6 products).

What is measured: the error of a K-term dot product caused by the operand representation ALONE (products and sums in
float64), relative to the rms magnitude of the result, next to the error fp32 accumulation itself adds (sequential fp32
sum of float64-exact products, which is what the MFMA's fp32 accumulator does at best).  Activations: GELU of a Gaussian
with a log-normal per-pixel gain (so one tensor spans several decades); weights: Gaussian, per-output-channel scale.

    python scratch/f16x2_probe.py
"""
import numpy as np


def bf16_round(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split_bf16x3(x):
    h = bf16_round(x)
    m = bf16_round(x - h)
    lo = bf16_round(x - h - m)
    return [h.astype(np.float64), m.astype(np.float64), lo.astype(np.float64)]


def split_f16x2(x, scale_exp, lo_shift):
    """h = fp16(x 2^s), l = fp16((x 2^s - h) 2^lo_shift); value = (h + l 2^-lo_shift) 2^-s."""
    xs = np.ldexp(x.astype(np.float64), scale_exp)
    with np.errstate(over="ignore"):
        h = xs.astype(np.float16)
    assert np.isfinite(h).all()
    l = np.ldexp(xs - h.astype(np.float64), lo_shift).astype(np.float16)
    return h.astype(np.float64), np.ldexp(l.astype(np.float64), -lo_shift), -scale_exp


def main():
    rng = np.random.default_rng(0)
    P, K, N = 2048, 1728, 64
    gain = np.exp(rng.normal(0, 2.0, (P, 1)))                       # decades of per-pixel dynamic range
    pre = rng.normal(0, 1, (P, K)) * gain
    a = (0.5 * pre * (1 + np.vectorize(__import__("math").erf)(pre / np.sqrt(2)))).astype(np.float32)
    w = (rng.normal(0, 1, (K, N)) * np.exp(rng.normal(-4, 1, (1, N)))).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    rms = np.sqrt((ref ** 2).mean(axis=0, keepdims=True))           # per output channel

    def report(name, got):
        e = np.abs(got - ref)
        print(f"{name:58s} max|err|/rms {np.max(e / rms):.3e}   rms err/rms {np.sqrt(((e / rms) ** 2).mean()):.3e}   "
              f"max rel (|ref|>rms/100) {np.max((e / np.abs(ref))[np.abs(ref) > rms / 100]):.3e}")

    # fp32 accumulation alone (operands exact): sequential fp32 sum over k in chunks of 16 (one MFMA k-step each)
    acc = np.zeros((P, N), np.float32)
    a64, w64 = a.astype(np.float64), w.astype(np.float64)
    for k0 in range(0, K, 16):
        acc = (acc.astype(np.float64) + a64[:, k0:k0 + 16] @ w64[k0:k0 + 16]).astype(np.float32)
    report("fp32 accumulate, exact operands (16-wide k-steps)", acc.astype(np.float64))

    ah, am, al = split_bf16x3(a)
    wh, wm, wl = split_bf16x3(w)
    report("bf16x3, 6 products (today)", ah @ wh + ah @ wm + am @ wh + ah @ wl + al @ wh + am @ wm)
    report("bf16x3, 3 products (hh hm mh)", ah @ wh + ah @ wm + am @ wh)

    amax = float(np.abs(a).max())
    sa = 14 - int(np.floor(np.log2(amax)))                          # max |a| lands in [2^14, 2^15)
    wmax = np.abs(w).max(axis=0)
    sw = (14 - np.floor(np.log2(wmax))).astype(np.int64)            # per output channel
    for lo_shift, tag in ((0, "one accumulator"), (11, "low terms scaled 2^11, second accumulator")):
        h, l, _ = split_f16x2(a, sa, lo_shift)
        whs = np.empty_like(w64)
        wls = np.empty_like(w64)
        for n in range(N):
            hh, ll, _ = split_f16x2(w[:, n], int(sw[n]), lo_shift)
            whs[:, n], wls[:, n] = hh, ll
        got = (h @ whs + h @ wls + l @ whs) * np.ldexp(1.0, -sa) * np.ldexp(1.0, -sw)[None, :]
        report(f"f16x2, 3 products, per-tensor 2^s ({tag})", got)
    print(f"activation range: max {amax:.3g}, median |a| {np.median(np.abs(a)):.3g}, 1st percentile {np.percentile(np.abs(a), 1):.3g}")


if __name__ == "__main__":
    main()
