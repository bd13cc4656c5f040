// Bitstream layer: CDF quantisation and the rANS entropy coder that the
// reference reaches through compressai 1.2.4 (`_CXX.pmf_to_quantized_cdf`, `ans.RansEncoder` /
// `ans.RansDecoder`; call sites reference entropy_models.py:61-64,175-183,206-294).
//
// compressai is absent offline, so this is a restatement of its PUBLISHED algorithm (which itself
// ports ryg_rans' rans64): 64-bit state, 32-bit renormalisation words, 16-bit CDF precision,
// 4-bit bypass chunks for out-of-range symbols, symbols pushed in order and encoded in reverse.
// Wire compatibility with compressai cannot be checked here (no vectors, no library) — DESIGN.md
// marks it "unpinned"; what IS tested is: round trip, agreement with the independent pure-Python
// restatement in oracle/rans_oracle.py, and the rate against -sum(log2 p).
//
// Host code by nature (bit-serial, one stream per image and slice); it runs beside the GPU path,
// exactly where the reference runs it (SURVEY §1 "sits beside the path").
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../include/vampic.h"

namespace vam {
void set_error(const char* fmt, ...);
}
using vam::set_error;

namespace {

constexpr int kPrecision = 16;
constexpr int kBypassBits = 4;
constexpr uint32_t kMaxBypass = (1u << kBypassBits) - 1;
constexpr uint64_t kRansL = 1ull << 31;

struct Sym {
  uint16_t start;
  uint16_t range;   // 0 means 65536 is impossible here: ranges are < 2^16 by construction
  bool bypass;
};

inline void enc_put(uint64_t& x, uint32_t*& p, uint32_t start, uint32_t freq, uint32_t scale_bits) {
  const uint64_t x_max = ((kRansL >> scale_bits) << 32) * freq;
  if (x >= x_max) {
    *--p = (uint32_t)x;
    x >>= 32;
  }
  x = ((x / freq) << scale_bits) + (x % freq) + start;
}

inline void enc_put_bits(uint64_t& x, uint32_t*& p, uint32_t val, uint32_t nbits) {
  const uint32_t freq = 1u << (16 - nbits);
  const uint64_t x_max = ((kRansL >> 16) << 32) * freq;
  if (x >= x_max) {
    *--p = (uint32_t)x;
    x >>= 32;
  }
  x = (x << nbits) | val;
}

inline uint32_t dec_get_bits(uint64_t& x, const uint32_t*& p, const uint32_t* end, uint32_t nbits, bool& ok) {
  const uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
  x >>= nbits;
  if (x < kRansL) {
    if (p >= end) { ok = false; return 0; }
    x = (x << 32) | *p++;
  }
  return val;
}

}  // namespace

extern "C" {

int vam_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int32_t* cdf_out) {
  if (!pmf || !cdf_out || n < 1 || precision < 1 || precision > 16) {
    set_error("vam_pmf_to_quantized_cdf: bad arguments");
    return VAM_EINVAL;
  }
  for (int i = 0; i < n; ++i)
    if (!(pmf[i] >= 0.f) || !std::isfinite(pmf[i])) {
      set_error("vam_pmf_to_quantized_cdf: invalid pmf value at %d", i);
      return VAM_EINVAL;
    }
  std::vector<uint32_t> cdf(n + 1);
  cdf[0] = 0;
  for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)std::round(pmf[i] * (float)(1 << precision));
  uint32_t total = 0;
  for (uint32_t v : cdf) total += v;
  if (total == 0) {
    set_error("vam_pmf_to_quantized_cdf: pmf sums to zero");
    return VAM_EINVAL;
  }
  for (auto& v : cdf) v = (uint32_t)((((uint64_t)1 << precision) * v) / total);
  for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
  cdf[n] = 1u << precision;
  for (int i = 0; i < n; ++i) {
    if (cdf[i] == cdf[i + 1]) {
      // steal one count from the least frequent symbol that can spare it
      uint32_t best_freq = ~0u;
      int best = -1;
      for (int j = 0; j < n; ++j) {
        uint32_t f = cdf[j + 1] - cdf[j];
        if (f > 1 && f < best_freq) { best_freq = f; best = j; }
      }
      if (best < 0) {
        set_error("vam_pmf_to_quantized_cdf: cannot give every symbol a non-zero frequency");
        return VAM_EINVAL;
      }
      if (best < i) for (int j = best + 1; j <= i; ++j) cdf[j]--;
      else for (int j = i + 1; j <= best; ++j) cdf[j]++;
    }
  }
  for (int i = 0; i <= n; ++i) cdf_out[i] = (int32_t)cdf[i];
  return VAM_OK;
}

long vam_rans_encode(const int32_t* symbols, const int32_t* indexes, long n, const int32_t* cdfs, int cdf_stride,
                     const int32_t* cdf_sizes, const int32_t* offsets, int n_cdfs, uint8_t* out, long out_cap) {
  if (!symbols || !indexes || !cdfs || !cdf_sizes || !offsets || !out || n < 0 || n_cdfs < 1) {
    set_error("vam_rans_encode: bad arguments");
    return VAM_EINVAL;
  }
  std::vector<Sym> syms;
  syms.reserve((size_t)n + 16);
  for (long i = 0; i < n; ++i) {
    const int ci = indexes[i];
    if (ci < 0 || ci >= n_cdfs) {
      set_error("vam_rans_encode: index %d out of range at %ld", ci, i);
      return VAM_EINVAL;
    }
    const int32_t* cdf = cdfs + (long)ci * cdf_stride;
    const int max_value = cdf_sizes[ci] - 2;
    if (max_value < 0 || max_value + 1 >= cdf_stride) {
      set_error("vam_rans_encode: cdf size %d invalid for table %d", cdf_sizes[ci], ci);
      return VAM_EINVAL;
    }
    int32_t value = symbols[i] - offsets[ci];
    uint32_t raw = 0;
    if (value < 0) {
      raw = (uint32_t)(-2 * (int64_t)value - 1);
      value = max_value;
    } else if (value >= max_value) {
      raw = (uint32_t)(2 * ((int64_t)value - max_value));
      value = max_value;
    }
    syms.push_back({(uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), false});
    if (value == max_value) {                      // bypass mode: Golomb-like count + raw 4-bit chunks
      int n_bypass = 0;
      while ((raw >> (n_bypass * kBypassBits)) != 0) ++n_bypass;
      int32_t val = n_bypass;
      while (val >= (int32_t)kMaxBypass) {
        syms.push_back({(uint16_t)kMaxBypass, (uint16_t)(kMaxBypass + 1), true});
        val -= kMaxBypass;
      }
      syms.push_back({(uint16_t)val, (uint16_t)(val + 1), true});
      for (int j = 0; j < n_bypass; ++j) {
        const uint32_t v = (raw >> (j * kBypassBits)) & kMaxBypass;
        syms.push_back({(uint16_t)v, (uint16_t)(v + 1), true});
      }
    }
  }
  std::vector<uint32_t> buf(syms.size() + 4);
  uint32_t* p = buf.data() + buf.size();
  uint64_t x = kRansL;
  for (size_t k = syms.size(); k-- > 0;) {
    const Sym& s = syms[k];
    if (s.bypass) enc_put_bits(x, p, s.start, kBypassBits);
    else {
      if (s.range == 0) {
        set_error("vam_rans_encode: zero-frequency symbol (cdf table not normalised)");
        return VAM_EINVAL;
      }
      enc_put(x, p, s.start, s.range, kPrecision);
    }
  }
  *--p = (uint32_t)(x >> 32);
  *--p = (uint32_t)x;
  const long nbytes = (long)((buf.data() + buf.size()) - p) * 4;
  if (nbytes > out_cap) {
    set_error("vam_rans_encode: output buffer too small (%ld > %ld)", nbytes, out_cap);
    return VAM_EINVAL;
  }
  std::memcpy(out, p, (size_t)nbytes);
  return nbytes;
}

int vam_rans_decode(const uint8_t* in, long n_bytes, const int32_t* indexes, long n, const int32_t* cdfs,
                    int cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets, int n_cdfs, int32_t* out) {
  if (!in || !indexes || !cdfs || !cdf_sizes || !offsets || !out || n < 0 || n_bytes < 8 || (n_bytes & 3)) {
    set_error("vam_rans_decode: bad arguments (stream of %ld bytes)", n_bytes);
    return VAM_EINVAL;
  }
  std::vector<uint32_t> words((size_t)n_bytes / 4);
  std::memcpy(words.data(), in, (size_t)n_bytes);
  const uint32_t* p = words.data();
  const uint32_t* end = p + words.size();
  uint64_t x = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
  p += 2;
  bool ok = true;
  for (long i = 0; i < n; ++i) {
    const int ci = indexes[i];
    if (ci < 0 || ci >= n_cdfs) {
      set_error("vam_rans_decode: index %d out of range at %ld", ci, i);
      return VAM_EINVAL;
    }
    const int32_t* cdf = cdfs + (long)ci * cdf_stride;
    const int sz = cdf_sizes[ci];
    const int max_value = sz - 2;
    const uint32_t cum = (uint32_t)(x & ((1u << kPrecision) - 1));
    int s = 0;                                   // first entry > cum, minus one
    while (s + 1 < sz && (uint32_t)cdf[s + 1] <= cum) ++s;
    const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
    x = (uint64_t)freq * (x >> kPrecision) + cum - start;
    if (x < kRansL) {
      if (p >= end) { ok = false; break; }
      x = (x << 32) | *p++;
    }
    int32_t value = s;
    if (value == max_value) {
      int32_t val = (int32_t)dec_get_bits(x, p, end, kBypassBits, ok);
      int32_t n_bypass = val;
      while (ok && val == (int32_t)kMaxBypass) {
        val = (int32_t)dec_get_bits(x, p, end, kBypassBits, ok);
        n_bypass += val;
      }
      uint32_t raw = 0;
      for (int j = 0; ok && j < n_bypass; ++j) raw |= dec_get_bits(x, p, end, kBypassBits, ok) << (j * kBypassBits);
      value = (int32_t)(raw >> 1);
      if (raw & 1) value = -value - 1;
      else value += max_value;
    }
    if (!ok) break;
    out[i] = value + offsets[ci];
  }
  if (!ok) {
    set_error("vam_rans_decode: bitstream truncated");
    return VAM_EINVAL;
  }
  return VAM_OK;
}

}  // extern "C"
