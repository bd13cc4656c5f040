"""Pure-Python restatement of the CDF quantiser and rANS coder of CompressAI 1.2.4.

TEST INFRASTRUCTURE — used only by tests/ to cross-check `csrc/rans.cpp` (an independent
implementation of the same published algorithm; compressai itself is absent offline, so the
wire format is "unpinned" — see DESIGN.md).  Python integers are unbounded, so the 64-bit
state arithmetic is written out literally.  Cites: compressai `cpp_exts/ops/ops.cpp`
(pmf_to_quantized_cdf), `cpp_exts/rans/rans_interface.cpp` + ryg_rans `rans64.h`; reference
call sites entropy_models.py:61-64,231-239,280-290.
"""
from __future__ import annotations

import struct
from typing import List, Sequence

PRECISION = 16
BYPASS = 4
MAX_BYPASS = (1 << BYPASS) - 1
RANS_L = 1 << 31


def pmf_to_quantized_cdf(pmf: Sequence[float], precision: int = 16) -> List[int]:
    import numpy as np
    f32 = np.float32
    cdf = [0] + [int(np.round(f32(p) * f32(1 << precision))) for p in pmf]
    total = sum(cdf)
    assert total > 0
    cdf = [((1 << precision) * c) // total for c in cdf]
    for i in range(1, len(cdf)):
        cdf[i] += cdf[i - 1]
    cdf[-1] = 1 << precision
    n = len(cdf) - 1
    for i in range(n):
        if cdf[i] == cdf[i + 1]:
            best_freq, best = None, -1
            for j in range(n):
                fr = cdf[j + 1] - cdf[j]
                if fr > 1 and (best_freq is None or fr < best_freq):
                    best_freq, best = fr, j
            assert best >= 0
            if best < i:
                for j in range(best + 1, i + 1):
                    cdf[j] -= 1
            else:
                for j in range(i + 1, best + 1):
                    cdf[j] += 1
    return cdf


def encode(symbols, indexes, cdfs, cdf_sizes, offsets) -> bytes:
    syms = []
    cdfs = [[int(v) for v in c] for c in cdfs]
    for s, ci in zip(symbols, indexes):
        s, ci = int(s), int(ci)
        cdf = cdfs[ci]
        max_value = int(cdf_sizes[ci]) - 2
        value = s - int(offsets[ci])
        raw = 0
        if value < 0:
            raw = -2 * value - 1
            value = max_value
        elif value >= max_value:
            raw = 2 * (value - max_value)
            value = max_value
        syms.append((cdf[value], cdf[value + 1] - cdf[value], False))
        if value == max_value:
            nb = 0
            while (raw >> (nb * BYPASS)) != 0:
                nb += 1
            val = nb
            while val >= MAX_BYPASS:
                syms.append((MAX_BYPASS, MAX_BYPASS + 1, True))
                val -= MAX_BYPASS
            syms.append((val, val + 1, True))
            for j in range(nb):
                v = (raw >> (j * BYPASS)) & MAX_BYPASS
                syms.append((v, v + 1, True))
    x = RANS_L
    words: List[int] = []                     # emitted back to front
    for start, rng, bypass in reversed(syms):
        if bypass:
            freq = 1 << (16 - BYPASS)
            x_max = ((RANS_L >> 16) << 32) * freq
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = (x << BYPASS) | start
        else:
            x_max = ((RANS_L >> PRECISION) << 32) * rng
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x // rng) << PRECISION) + (x % rng) + start
    words.append(x >> 32)
    words.append(x & 0xFFFFFFFF)
    words.reverse()
    return struct.pack(f"<{len(words)}I", *words)


def decode(stream: bytes, indexes, cdfs, cdf_sizes, offsets) -> List[int]:
    words = list(struct.unpack(f"<{len(stream) // 4}I", stream))
    x = words[0] | (words[1] << 32)
    pos = 2

    def get_bits(nbits):
        nonlocal x, pos
        val = x & ((1 << nbits) - 1)
        x >>= nbits
        if x < RANS_L:
            x = (x << 32) | words[pos]
            pos += 1
        return val

    out = []
    cdfs = [[int(v) for v in c] for c in cdfs]
    for ci in indexes:
        ci = int(ci)
        cdf = cdfs[ci]
        sz = int(cdf_sizes[ci])
        max_value = sz - 2
        cum = x & ((1 << PRECISION) - 1)
        s = 0
        while s + 1 < sz and cdf[s + 1] <= cum:
            s += 1
        start, freq = cdf[s], cdf[s + 1] - cdf[s]
        x = freq * (x >> PRECISION) + cum - start
        if x < RANS_L:
            x = (x << 32) | words[pos]
            pos += 1
        value = s
        if value == max_value:
            val = get_bits(BYPASS)
            nb = val
            while val == MAX_BYPASS:
                val = get_bits(BYPASS)
                nb += val
            raw = 0
            for j in range(nb):
                raw |= get_bits(BYPASS) << (j * BYPASS)
            value = raw >> 1
            value = -value - 1 if (raw & 1) else value + max_value
        out.append(value + int(offsets[ci]))
    return out
