// Variance-aware mask: per (image, slice) segment, threshold = torch.quantile(sigma.ravel(),
// 1 - 0.1*pr) with linear interpolation, mask = sigma >= threshold
// (reference layers/channel_mask.py:132-151; ProgMask :18-49 is the same rule per block).
//
// Bit-exact restatement of ATen's quantile without a sort:
//   rank = float32(1 - 0.1*pr) * float32(n-1);  lo = floor(rank); hi = ceil(rank); w = rank - lo
//   a = sorted[lo], b = sorted[hi]            -> two order statistics by radix select on
//                                                 order-preserving uint32 keys (4 passes x 8 bit)
//   thr = w < 0.5 ? fma(w, b-a, a) : fma(w-1, b-a, b)    (ATen's lerp kernel is a fused fma)
//   any NaN in the segment -> thr = NaN -> mask all zero.
// One 1024-thread workgroup per segment; the segment's elements are loaded ONCE (16 B per
// lane, coalesced over the NHWC channel window) and stay in registers for every pass, so HBM
// traffic is the algorithmic 4 B read + 4 B written per element.  Histograms live in LDS.
#include "common.h"
#include <cmath>

namespace vam {

struct MaskArgs {
  const float* sigma;
  float* mask;
  float* thr;
  long batch_stride, slice_stride, mask_batch_stride, mask_slice_stride;
  int ld, ld_mask, n_slice, n_pix, C4;
  int k_lo, k_hi;
  float w;
  int mode;  // 0 = quantile, 1 = all zero (pr == 0), 2 = all one (pr >= 10)
};

__device__ __forceinline__ unsigned f2key(float f) {
  unsigned u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  unsigned u = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  return __uint_as_float(u);
}

// MAXV = float4 per thread kept in registers (0 = stream from memory every pass)
template <int MAXV>
__global__ __launch_bounds__(1024) void variance_mask_kernel(const MaskArgs a) {
  __shared__ unsigned hist[256];
  __shared__ unsigned sh_prefix, sh_k, sh_cnt, sh_min;
  __shared__ int sh_nan;

  const int seg = blockIdx.x;
  const int b = seg / a.n_slice, j = seg - b * a.n_slice;
  const float* src = a.sigma + b * a.batch_stride + j * a.slice_stride;
  float* dst = a.mask + b * a.mask_batch_stride + j * a.mask_slice_stride;
  const int nvec = a.n_pix * a.C4;
  const int tid = threadIdx.x;

  auto vec_ptr = [&](int i) -> const float* {
    int p = i / a.C4;
    return src + (long)p * a.ld + (i - p * a.C4) * 4;
  };
  auto out_ptr = [&](int i) -> float* {
    int p = i / a.C4;
    return dst + (long)p * a.ld_mask + (i - p * a.C4) * 4;
  };

  if (a.mode != 0) {
    const float v = a.mode == 2 ? 1.f : 0.f;
    for (int i = tid; i < nvec; i += 1024) *reinterpret_cast<float4*>(out_ptr(i)) = make_float4(v, v, v, v);
    if (tid == 0 && a.thr) a.thr[seg] = a.mode == 2 ? -INFINITY : INFINITY;
    return;
  }

  float4 reg[MAXV > 0 ? MAXV : 1];
  if (MAXV > 0) {
#pragma unroll
    for (int r = 0; r < MAXV; ++r) {
      int i = tid + r * 1024;
      reg[r] = (i < nvec) ? *reinterpret_cast<const float4*>(vec_ptr(i)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // visit every element of the segment: f(float)
  auto for_each = [&](auto&& f) {
    if (MAXV > 0) {
#pragma unroll
      for (int r = 0; r < MAXV; ++r) {
        if (tid + r * 1024 < nvec) { f(reg[r].x); f(reg[r].y); f(reg[r].z); f(reg[r].w); }
      }
    } else {
      for (int i = tid; i < nvec; i += 1024) {
        float4 v = *reinterpret_cast<const float4*>(vec_ptr(i));
        f(v.x); f(v.y); f(v.z); f(v.w);
      }
    }
  };

  if (tid == 0) sh_nan = 0;
  __syncthreads();
  {
    int nan = 0;
    for_each([&](float x) { nan |= (x != x) ? 1 : 0; });
    if (nan) atomicOr(&sh_nan, 1);
  }

  // ---- radix select of rank k_lo (ascending, 0-based)
  if (tid == 0) { sh_prefix = 0u; sh_k = (unsigned)a.k_lo; }
  for (int pass = 3; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const unsigned prefix = sh_prefix;
    const int shift = pass * 8;
    const unsigned hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for_each([&](float x) {
      unsigned k = f2key(x);
      if ((k & hi_mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
    });
    __syncthreads();
    if (tid < 64) {
      unsigned h0 = hist[tid * 4], h1 = hist[tid * 4 + 1], h2 = hist[tid * 4 + 2], h3 = hist[tid * 4 + 3];
      unsigned c = h0 + h1 + h2 + h3;
      unsigned incl = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        unsigned t = __shfl_up(incl, o, 64);
        if (tid >= o) incl += t;
      }
      unsigned excl = incl - c;
      const unsigned k = sh_k;
      if (k >= excl && k < incl) {
        unsigned kk = k - excl;
        unsigned bin;
        if (kk < h0) bin = 0;
        else if ((kk -= h0) < h1) bin = 1;
        else if ((kk -= h1) < h2) bin = 2;
        else { kk -= h2; bin = 3; }
        sh_prefix = prefix | ((unsigned)(tid * 4 + bin) << shift);
        sh_k = kk;
      }
    }
    __syncthreads();
  }
  const unsigned key_lo = sh_prefix;
  unsigned key_hi = key_lo;
  if (a.k_hi != a.k_lo) {
    // sorted[k_lo+1]: equals key_lo when more than k_lo+1 elements are <= key_lo, else the
    // smallest key above it.
    if (tid == 0) { sh_cnt = 0u; sh_min = 0xFFFFFFFFu; }
    __syncthreads();
    unsigned cnt = 0u, mn = 0xFFFFFFFFu;
    for_each([&](float x) {
      unsigned k = f2key(x);
      if (k <= key_lo) ++cnt;
      else mn = mn < k ? mn : k;
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      cnt += __shfl_down(cnt, o, 64);
      unsigned t = __shfl_down(mn, o, 64);
      mn = mn < t ? mn : t;
    }
    if ((tid & 63) == 0) { atomicAdd(&sh_cnt, cnt); atomicMin(&sh_min, mn); }
    __syncthreads();
    key_hi = (sh_cnt > (unsigned)a.k_lo + 1u) ? key_lo : sh_min;
  }
  const float lo_v = key2f(key_lo), hi_v = key2f(key_hi);
  const float d = hi_v - lo_v;
  float thr = (a.w < 0.5f) ? __builtin_fmaf(a.w, d, lo_v) : __builtin_fmaf(a.w - 1.0f, d, hi_v);
  if (sh_nan) thr = __uint_as_float(0x7FC00000u);
  if (tid == 0 && a.thr) a.thr[seg] = thr;

  if (MAXV > 0) {
#pragma unroll
    for (int r = 0; r < MAXV; ++r) {
      int i = tid + r * 1024;
      if (i < nvec) {
        float4 v = reg[r];
        *reinterpret_cast<float4*>(out_ptr(i)) = make_float4(v.x >= thr ? 1.f : 0.f, v.y >= thr ? 1.f : 0.f,
                                                              v.z >= thr ? 1.f : 0.f, v.w >= thr ? 1.f : 0.f);
      }
    }
  } else {
    for (int i = tid; i < nvec; i += 1024) {
      float4 v = *reinterpret_cast<const float4*>(vec_ptr(i));
      *reinterpret_cast<float4*>(out_ptr(i)) = make_float4(v.x >= thr ? 1.f : 0.f, v.y >= thr ? 1.f : 0.f,
                                                            v.z >= thr ? 1.f : 0.f, v.w >= thr ? 1.f : 0.f);
    }
  }
}

}  // namespace vam

using namespace vam;

extern "C" int vam_variance_mask(const float* sigma, int ld, long batch_stride, long slice_stride, int n_batch,
                                 int n_slice, int n_pix, int C, double pr, float* mask_out, int ld_mask,
                                 long mask_batch_stride, long mask_slice_stride, float* thr_out, void* stream) {
  VAM_REQUIRE(sigma && mask_out && n_batch > 0 && n_slice > 0 && n_pix > 0 && C > 0, "vam_variance_mask: bad arguments");
  VAM_REQUIRE(C % 4 == 0 && ld % 4 == 0 && ld_mask % 4 == 0 && batch_stride % 4 == 0 && slice_stride % 4 == 0 && mask_batch_stride % 4 == 0 && mask_slice_stride % 4 == 0, "vam_variance_mask: C and strides must be multiples of 4");
  VAM_REQUIRE((((uintptr_t)sigma) & 15) == 0 && (((uintptr_t)mask_out) & 15) == 0, "vam_variance_mask: 16-byte alignment");
  VAM_REQUIRE(ld >= C && ld_mask >= C, "vam_variance_mask: pixel stride < C");
  const long n = (long)n_pix * C;
  // torch.quantile rejects inputs above 16M elements (ATen Sorting.cpp); so do we
  VAM_REQUIRE(n <= 16000000L, "vam_variance_mask: segment of %ld elements exceeds torch.quantile's 16M limit", n);
  VAM_REQUIRE(pr >= 0.0 && pr == pr, "vam_variance_mask: pr must be >= 0");
  MaskArgs a;
  a.sigma = sigma; a.mask = mask_out; a.thr = thr_out;
  a.batch_stride = batch_stride; a.slice_stride = slice_stride;
  a.mask_batch_stride = mask_batch_stride; a.mask_slice_stride = mask_slice_stride;
  a.ld = ld; a.ld_mask = ld_mask; a.n_slice = n_slice; a.n_pix = n_pix; a.C4 = C / 4;
  a.k_lo = a.k_hi = 0; a.w = 0.f;
  if (pr >= 10.0) a.mode = 2;                 // channel_mask.py:133-134
  else if (pr == 0.0) a.mode = 1;             // :135-136
  else {
    a.mode = 0;
    const double q_keep = pr * 0.1;            // python float arithmetic of :138-139
    // volatile: each step must round to fp32 exactly like ATen's tensor ops; a host-side
    // contraction of (qt*(n-1)) - lo into one fma changes w in the 5th digit and the
    // threshold by an ulp (caught by tests/golden thresholds).
    volatile float qt = (float)(1.0 - q_keep);       // scalar_tensor(q, float32)
    volatile float last = (float)(n - 1);
    volatile float rank = qt * last;                 // fp32 multiply (q * last_index)
    const float lo = floorf(rank);
    a.k_lo = (int)lo;
    a.k_hi = (int)ceilf(rank);
    a.w = rank - lo;
    VAM_REQUIRE(a.k_lo >= 0 && a.k_hi < n, "vam_variance_mask: rank out of range");
  }
  const int segs = n_batch * n_slice;
  const int nvec = n_pix * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(VAM_FAM_MASK, s, 0, 8.0 * (double)n * segs);
  if (nvec <= 4 * 1024)
    hipLaunchKernelGGL((variance_mask_kernel<4>), dim3(segs), dim3(1024), 0, s, a);
  else if (nvec <= 16 * 1024)
    hipLaunchKernelGGL((variance_mask_kernel<16>), dim3(segs), dim3(1024), 0, s, a);
  else
    hipLaunchKernelGGL((variance_mask_kernel<0>), dim3(segs), dim3(1024), 0, s, a);
  return check_launch("variance_mask_kernel");
}
