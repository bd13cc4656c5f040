// Shared host-side helpers of libvampic (error reporting, launch bracketing for the
// HIP-event profiler).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../../include/vampic.h"

namespace vam {

void set_error(const char* fmt, ...);

#define VAM_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      vam::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VAM_EHIP;                                                             \
    }                                                                              \
  } while (0)

#define VAM_REQUIRE(cond, ...)                                                     \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      vam::set_error(__VA_ARGS__);                                                 \
      return VAM_EINVAL;                                                           \
    }                                                                              \
  } while (0)

// Event bracketing: when profiling is on, record an event before and after the launch
// on the SAME stream the kernel runs on; vam_prof_read sums the elapsed times.
struct ProfScope {
  ProfScope(int family, hipStream_t s, double flops, double bytes);
  ~ProfScope();
  int family;
  hipStream_t stream;
  bool active;
  size_t slot;
};

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("launch of %s failed: %s", what, hipGetErrorString(e));
    return VAM_EHIP;
  }
  return VAM_OK;
}

inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

// nn.GELU() (exact erf form, reference models/pic.py:86 and layers/layers.py:35-41):  0.5 v (1 + erf(v / sqrt 2)).
// ocml's erff costs ~45 instructions with two divergent branches (|x| < 1: odd polynomial; else 1 - exp(-poly)), and the
// path evaluates 1.3 G GELUs per 32 x 256 x 256 step — in the fused residual unit they are a third of a tile's cycles.
// One branch-free form serves GELU as well:  erf(|x|) = 1 - 2^(-|x| P(|x|)),  P = degree-9 fit of -log2(erfc(a)) / a on
// [0, 4] weighted by erfc(a) a (the sensitivity of erf to P), |x| clamped at 4 (erfc(4) = 1.5e-8 rounds away in 1 - e).
// 18 instructions.  It gives up RELATIVE accuracy of erf near 0 (absolute error stays 8.3e-8 = 1.4 ulp of 1), which GELU
// does not need: 1 + erf is formed anyway.  Measured over 2 M points of [-8, 8] against float64 (scratch/gelu_fit.py):
// GELU max relative error 3.5e-7 for v > -1 (a correctly rounded erf in the same formula: 2.1e-7), max absolute error
// 4.5e-7 (the same), bit-identical to the correctly-rounded-erf result on 96 % of the points, monotone for v > -0.7.
// Every GELU of the library goes through this function (conv epilogues, the fused residual unit, the taped training
// forward), so plans that must agree bit for bit still do.
__device__ __forceinline__ float vam_gelu(float v) {
  const float x = v * 0.70710678118654752440f;
  const float a = fminf(fabsf(x), 4.0f);
  float p = 1.9250594505137997e-06f;
  p = fmaf(p, a, -3.2931573514360934e-05f);
  p = fmaf(p, a, 0.00025072888820432127f);
  p = fmaf(p, a, -0.0010894045699387789f);
  p = fmaf(p, a, 0.0026234956458210945f);
  p = fmaf(p, a, -0.00038660463178530335f);
  p = fmaf(p, a, -0.027571382001042366f);
  p = fmaf(p, a, 0.14826533198356628f);
  p = fmaf(p, a, 0.9184485077857971f);
  p = fmaf(p, a, 1.6279070377349854f);
  const float e = __builtin_amdgcn_exp2f(-(p * a));       // v_exp_f32; the argument is in [-26.1, 0]: no denormals
  const float r = copysignf(1.0f - e, x);
  return (v * 0.5f) * (1.0f + r);
}

// d/dv [ v Phi(v) ] = Phi(v) + v phi(v): the derivative of nn.GELU() (autograd of pic.py:86 / layers/layers.py:35-41).
// One definition for the element-wise backward kernel (train_gs.hip) and the data-gradient launches that apply it in
// their epilogue (VAM_CONV_MUL_GELU_GRAD): the two routes give the same bits.
__device__ __forceinline__ float vam_gelu_grad(float v) {
  return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.3989422804014327f * expf(-0.5f * v * v);
}

}  // namespace vam
