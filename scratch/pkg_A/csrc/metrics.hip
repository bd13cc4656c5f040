// MS-SSIM building blocks (reference utility/functions.py:176-177 -> pytorch_msssim.ms_ssim, used by the real-codec
// evaluation training/step.py:323-326): one SSIM level (11x11 Gaussian window, "valid" positions) and the 2x2 average
// pooling between levels.  A metric, not the hot path: a direct 121-tap window per output position out of L2 is
// ~0.1 ms for a Kodak image; sums are accumulated in double (wave-aggregated atomics, like the rate sums).
#include "common.h"

namespace vam {

__global__ __launch_bounds__(256) void ssim_level_kernel(const float* __restrict__ x, const float* __restrict__ y, int H,
                                                         int W, const float* __restrict__ win, float c1, float c2,
                                                         double* __restrict__ ssim_sum, double* __restrict__ cs_sum) {
  __shared__ float g[11];
  __shared__ double red[2][4];
  if (threadIdx.x < 11) g[threadIdx.x] = win[threadIdx.x];
  __syncthreads();
  const int plane = blockIdx.y;
  const int Ho = H - 10, Wo = W - 10;
  const float* xp = x + (size_t)plane * H * W;
  const float* yp = y + (size_t)plane * H * W;
  double s_ssim = 0.0, s_cs = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)Ho * Wo; i += (long)gridDim.x * blockDim.x) {
    const int oy = (int)(i / Wo), ox = (int)(i - (long)oy * Wo);
    // separable order of the reference implementation: filter along H first, then along W
    float mx = 0.f, my = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
    for (int dx = 0; dx < 11; ++dx) {
      float cx = 0.f, cy = 0.f, cxx = 0.f, cyy = 0.f, cxy = 0.f;
      for (int dy = 0; dy < 11; ++dy) {
        const float a = xp[(size_t)(oy + dy) * W + ox + dx], b = yp[(size_t)(oy + dy) * W + ox + dx];
        const float w = g[dy];
        cx += w * a; cy += w * b; cxx += w * (a * a); cyy += w * (b * b); cxy += w * (a * b);
      }
      const float w = g[dx];
      mx += w * cx; my += w * cy; sxx += w * cxx; syy += w * cyy; sxy += w * cxy;
    }
    const float vx = sxx - mx * mx, vy = syy - my * my, cov = sxy - mx * my;
    const float cs = (2.f * cov + c2) / (vx + vy + c2);
    const float ss = (2.f * mx * my + c1) / (mx * mx + my * my + c1) * cs;
    s_ssim += (double)ss;
    s_cs += (double)cs;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s_ssim += __shfl_down(s_ssim, o);
    s_cs += __shfl_down(s_cs, o);
  }
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wid] = s_ssim; red[1][wid] = s_cs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&ssim_sum[plane], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&cs_sum[plane], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// F.avg_pool2d(kernel 2, stride 2, padding (ph, pw), zeros counted in the mean)
__global__ void avgpool2_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int Ho, int Wo, int ph,
                                int pw, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    long r = i / Wo;
    const int oy = (int)(r % Ho);
    const long plane = r / Ho;
    const float* xp = x + plane * H * W;
    float s = 0.f;
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx) {
        const int iy = 2 * oy - ph + dy, ix = 2 * ox - pw + dx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) s += xp[(size_t)iy * W + ix];
      }
    out[i] = s * 0.25f;
  }
}

}  // namespace vam

using namespace vam;

extern "C" {

int vam_ssim_level(const float* x, const float* y, int planes, int H, int W, const float* win11, float c1, float c2,
                   double* ssim_sum, double* cs_sum, void* stream) {
  VAM_REQUIRE(x && y && win11 && ssim_sum && cs_sum && planes > 0, "vam_ssim_level: bad arguments");
  VAM_REQUIRE(H > 10 && W > 10, "vam_ssim_level: plane %dx%d smaller than the 11x11 window", H, W);
  const long n = (long)(H - 10) * (W - 10);
  unsigned gx = (unsigned)((n + 255) / 256);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(ssim_level_kernel, dim3(gx, planes), dim3(256), 0, (hipStream_t)stream, x, y, H, W, win11, c1, c2,
                     ssim_sum, cs_sum);
  return check_launch("ssim_level_kernel");
}

int vam_avgpool2(const float* x, float* out, int planes, int H, int W, int pad_h, int pad_w, void* stream) {
  VAM_REQUIRE(x && out && planes > 0 && H > 0 && W > 0 && pad_h >= 0 && pad_h <= 1 && pad_w >= 0 && pad_w <= 1, "vam_avgpool2: bad arguments");
  const int Ho = (H + 2 * pad_h - 2) / 2 + 1, Wo = (W + 2 * pad_w - 2) / 2 + 1;
  const long total = (long)planes * Ho * Wo;
  unsigned g = (unsigned)((total + 255) / 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(avgpool2_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, out, H, W, Ho, Wo, pad_h, pad_w, total);
  return check_launch("avgpool2_kernel");
}

}  // extern "C"
