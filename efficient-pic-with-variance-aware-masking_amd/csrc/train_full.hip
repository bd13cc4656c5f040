// First-stage training (BASELINE configs[3], reference train.py:146-149 `--training_type first_train`: every parameter
// trains, models/pic.py:301-491 forward(quality=[0,10], training=True), training/loss.py:6-66) — the backward pieces that
// the decoder-refinement (csrc/train_gs.hip) and REM (csrc/train.hip) schedules did not need:
//   * entropy-bottleneck noise likelihood: gradient w.r.t. z and the 14 density-network tensors
//     (entropy_models.py:403-436,449-492; LowerBound rule on the 1e-9 likelihood bound);
//   * PixelShuffle(2) backward (layers/layers.py:82-86) = pixel un-shuffle of the output gradient into the conv's
//     channel order c*4 + i*2 + j;
//   * zero-insertion up-sampling: the data gradient of a 3x3 stride-2 convolution (h_a, models/builder.py:72-82) is the
//     stride-1 data-gradient convolution (VAM_PACK_CONV_DGRAD) of the output gradient with zeros between its samples.
// These tensors live on the 4x4 ... 16x16 grids of the hyperprior: a few hundred KB per launch, written for clarity and
// determinism (fixed reduction order, no float atomics).
#include "common.h"

namespace vam {

__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_t(float x) { return 1.0f / (1.0f + expf(-x)); }

// Per-channel parameter block (58 values): layer 0: m[3] b[3] f[3]; layers 1..3: m[9] b[3] f[3]; layer 4: m[3] b[1].
// `raw` keeps the stored values (for softplus' = sigmoid(raw m), tanh' = 1 - tanh(raw f)^2), `sp` the transformed ones.
struct EbNet {
  float m0[3], b0[3], f0[3];
  float m[3][9], b[3][3], f[3][3];
  float m4[3], b4;
};

// forward of the 1-3-3-3-3-1 network at x keeping what the backward needs
struct EbTrace {
  float th0[3], l0[3];          // tanh(v) and output of layer 0
  float th[3][3], l[3][3];      // layers 1..3
};

__device__ __forceinline__ float eb_fwd(const EbNet& n, float x, EbTrace& t) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float v = n.m0[k] * x + n.b0[k];
    t.th0[k] = tanhf(v);
    t.l0[k] = v + n.f0[k] * t.th0[k];
  }
  const float* in = t.l0;
#pragma unroll
  for (int L = 0; L < 3; ++L) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v = n.m[L][k * 3 + 0] * in[0];
      v = v + n.m[L][k * 3 + 1] * in[1];
      v = v + n.m[L][k * 3 + 2] * in[2];
      v = v + n.b[L][k];
      t.th[L][k] = tanhf(v);
      t.l[L][k] = v + n.f[L][k] * t.th[L][k];
    }
    in = t.l[L];
  }
  float v = n.m4[0] * in[0];
  v = v + n.m4[1] * in[1];
  v = v + n.m4[2] * in[2];
  return v + n.b4;
}

// backward of one evaluation: g = dL/dlogit; accumulates dL/d(transformed parameters) into `acc` (same struct layout,
// still w.r.t. softplus(m) / tanh(f): the chain to the stored values is applied once per channel at the end), returns dL/dx
__device__ __forceinline__ float eb_bwd(const EbNet& n, float x, const EbTrace& t, float g, EbNet& acc) {
  float gin[3];
  const float* in = t.l[2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    acc.m4[j] += g * in[j];
    gin[j] = g * n.m4[j];
  }
  acc.b4 += g;
#pragma unroll
  for (int L = 2; L >= 0; --L) {
    const float* prev = L == 0 ? t.l0 : t.l[L - 1];
    float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float th = t.th[L][k];
      acc.f[L][k] += gin[k] * th;
      const float gv = gin[k] * (1.0f + n.f[L][k] * (1.0f - th * th));
      acc.b[L][k] += gv;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc.m[L][k * 3 + j] += gv * prev[j];
        gp[j] += gv * n.m[L][k * 3 + j];
      }
    }
    gin[0] = gp[0]; gin[1] = gp[1]; gin[2] = gp[2];
  }
  float gx = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float th = t.th0[k];
    acc.f0[k] += gin[k] * th;
    const float gv = gin[k] * (1.0f + n.f0[k] * (1.0f - th * th));
    acc.b0[k] += gv;
    acc.m0[k] += gv * x;
    gx += gv * n.m0[k];
  }
  return gx;
}

constexpr int EB_NPAR = 58;
constexpr int EB_THREADS = 128;

// One block per channel.  params as for vam_eb_forward (tensor-major, channel-major inside a tensor).
__global__ __launch_bounds__(EB_THREADS) void eb_train_bwd_kernel(const float* __restrict__ z, int ld_z,
                                                                  const float* __restrict__ noise, int ld_noise,
                                                                  const float* __restrict__ params, int C,
                                                                  const float* __restrict__ glik, int ld_g,
                                                                  float* __restrict__ dz, int ld_dz,
                                                                  float* __restrict__ dparams, long n_pix) {
  __shared__ float red[EB_NPAR][EB_THREADS + 1];
  const int c = blockIdx.x;
  // tensor offsets inside the parameter block
  const long o_m0 = 0, o_b0 = 3L * C, o_f0 = 6L * C;
  long o_m[3], o_b[3], o_f[3];
  long o = 9L * C;
  for (int L = 0; L < 3; ++L) { o_m[L] = o; o_b[L] = o + 9L * C; o_f[L] = o + 12L * C; o += 15L * C; }
  const long o_m4 = o, o_b4 = o + 3L * C, o_q = o + 4L * C;
  EbNet n, raw;
  for (int k = 0; k < 3; ++k) {
    raw.m0[k] = params[o_m0 + c * 3 + k]; n.m0[k] = softplus_t(raw.m0[k]);
    n.b0[k] = params[o_b0 + c * 3 + k];
    raw.f0[k] = params[o_f0 + c * 3 + k]; n.f0[k] = tanhf(raw.f0[k]);
    raw.m4[k] = params[o_m4 + c * 3 + k]; n.m4[k] = softplus_t(raw.m4[k]);
  }
  n.b4 = params[o_b4 + c];
  for (int L = 0; L < 3; ++L) {
    for (int k = 0; k < 9; ++k) { raw.m[L][k] = params[o_m[L] + c * 9 + k]; n.m[L][k] = softplus_t(raw.m[L][k]); }
    for (int k = 0; k < 3; ++k) {
      n.b[L][k] = params[o_b[L] + c * 3 + k];
      raw.f[L][k] = params[o_f[L] + c * 3 + k]; n.f[L][k] = tanhf(raw.f[L][k]);
    }
  }
  EbNet acc;
  {
    float* a = reinterpret_cast<float*>(&acc);
    for (int i = 0; i < EB_NPAR; ++i) a[i] = 0.f;
  }
  for (long p = threadIdx.x; p < n_pix; p += EB_THREADS) {
    const float x = z[p * ld_z + c] + noise[p * ld_noise + c];
    EbTrace tl, tu;
    const float lower = eb_fwd(n, x - 0.5f, tl);
    const float upper = eb_fwd(n, x + 0.5f, tu);
    const float sum = lower + upper;
    const float sign = sum > 0.f ? -1.f : (sum < 0.f ? 1.f : 0.f);          // detached (entropy_models.py:431-432)
    const float su = sigmoid_t(sign * upper), sl = sigmoid_t(sign * lower);
    const float diff = su - sl;
    const float lik_raw = fabsf(diff);
    float g = glik[p * ld_g + c];
    if (!(lik_raw >= 1e-9f || g < 0.f)) g = 0.f;                            // LowerBound(1e-9) gradient rule
    const float gd = g * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));     // torch.abs backward (0 at 0)
    const float gu = gd * su * (1.0f - su) * sign;
    const float gl = -gd * sl * (1.0f - sl) * sign;
    const float gx = eb_bwd(n, x + 0.5f, tu, gu, acc) + eb_bwd(n, x - 0.5f, tl, gl, acc);
    dz[p * ld_dz + c] = gx;
  }
  // chain to the stored values: d softplus(m)/dm = sigmoid(m); d tanh(f)/df = 1 - tanh(f)^2
  for (int k = 0; k < 3; ++k) {
    acc.m0[k] *= sigmoid_t(raw.m0[k]);
    acc.f0[k] *= 1.0f - n.f0[k] * n.f0[k];
    acc.m4[k] *= sigmoid_t(raw.m4[k]);
  }
  for (int L = 0; L < 3; ++L) {
    for (int k = 0; k < 9; ++k) acc.m[L][k] *= sigmoid_t(raw.m[L][k]);
    for (int k = 0; k < 3; ++k) acc.f[L][k] *= 1.0f - n.f[L][k] * n.f[L][k];
  }
  {
    const float* a = reinterpret_cast<const float*>(&acc);
    for (int i = 0; i < EB_NPAR; ++i) red[i][threadIdx.x] = a[i];
  }
  __syncthreads();
  if (threadIdx.x < EB_NPAR) {
    float s = 0.f;
    for (int t = 0; t < EB_THREADS; ++t) s += red[threadIdx.x][t];         // fixed order
    // EbNet member order: m0[3] b0[3] f0[3] | m[3][9] b[3][3] f[3][3] | m4[3] b4
    const int i = threadIdx.x;
    long dst;
    if (i < 3) dst = o_m0 + c * 3 + i;
    else if (i < 6) dst = o_b0 + c * 3 + (i - 3);
    else if (i < 9) dst = o_f0 + c * 3 + (i - 6);
    else if (i < 36) { const int L = (i - 9) / 9, k = (i - 9) % 9; dst = o_m[L] + c * 9 + k; }
    else if (i < 45) { const int L = (i - 36) / 3, k = (i - 36) % 3; dst = o_b[L] + c * 3 + k; }
    else if (i < 54) { const int L = (i - 45) / 3, k = (i - 45) % 3; dst = o_f[L] + c * 3 + k; }
    else if (i < 57) dst = o_m4 + c * 3 + (i - 54);
    else dst = o_b4 + c;
    dparams[dst] = s;
  }
  if (threadIdx.x < 3) dparams[o_q + c * 3 + threadIdx.x] = 0.f;            // the noise likelihood does not see the quantiles
}

// dst[b, y, x, c*4 + i*2 + j] = src[b, 2y+i, 2x+j, c]      (gradient of PixelShuffle(2); H, W = extent of dst)
__global__ void ps2_unshuffle_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, int B, int H,
                                     int W, int Cq) {
  const long total = (long)B * H * W * Cq * 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % (4 * Cq));
    long p = i / (4 * Cq);
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    const int c = n >> 2, ph = n & 3;
    const long sp = ((long)b * 2 * H + 2 * y + (ph >> 1)) * (2 * W) + 2 * x + (ph & 1);
    dst[(((long)b * H + y) * W + x) * ld_dst + n] = src[sp * ld_src + c];
  }
}

// dst[b, 2y, 2x, :] = src[b, y, x, :], zero elsewhere      (H, W = extent of src; C % 4 == 0)
__global__ void upsample2_zero_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, int B, int H,
                                      int W, int C4) {
  const long total = (long)B * 2 * H * 2 * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long p = i / C4;
    const int x = (int)(p % (2 * W));
    p /= 2 * W;
    const int y = (int)(p % (2 * H));
    const int b = (int)(p / (2 * H));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(x & 1) && !(y & 1)) v = *reinterpret_cast<const float4*>(src + (((long)b * H + (y >> 1)) * W + (x >> 1)) * ld_src + c);
    *reinterpret_cast<float4*>(dst + (((long)b * 2 * H + y) * (2 * W) + x) * ld_dst + c) = v;
  }
}

static inline unsigned fgrid(long n, int block) {
  long g = (n + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace vam

using namespace vam;

extern "C" {

int vam_eb_train_bwd(const float* z, int ld_z, const float* noise, int ld_noise, const float* params, int C,
                     const float* grad_lik, int ld_glik, float* dz, int ld_dz, float* dparams, long n_pix, void* stream) {
  VAM_REQUIRE(z && noise && params && grad_lik && dz && dparams && C > 0 && n_pix > 0, "vam_eb_train_bwd: bad arguments");
  VAM_REQUIRE(ld_z >= C && ld_noise >= C && ld_glik >= C && ld_dz >= C, "vam_eb_train_bwd: pixel strides");
  hipLaunchKernelGGL(eb_train_bwd_kernel, dim3(C), dim3(EB_THREADS), 0, (hipStream_t)stream, z, ld_z, noise, ld_noise, params, C,
                     grad_lik, ld_glik, dz, ld_dz, dparams, n_pix);
  return check_launch("eb_train_bwd_kernel");
}

int vam_ps2_unshuffle(const float* src, int ld_src, float* dst, int ld_dst, int B, int H, int W, int Cq, void* stream) {
  VAM_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && Cq > 0 && ld_src >= Cq && ld_dst >= 4 * Cq, "vam_ps2_unshuffle: bad arguments");
  const long total = (long)B * H * W * Cq * 4;
  hipLaunchKernelGGL(ps2_unshuffle_kernel, dim3(fgrid(total, 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, B,
                     H, W, Cq);
  return check_launch("ps2_unshuffle_kernel");
}

int vam_upsample2_zero(const float* src, int ld_src, float* dst, int ld_dst, int B, int H, int W, int C, void* stream) {
  VAM_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ld_src >= C && ld_dst >= C, "vam_upsample2_zero: bad arguments");
  VAM_REQUIRE(ld_src % 4 == 0 && ld_dst % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "vam_upsample2_zero: alignment");
  const long total = (long)B * 2 * H * 2 * W * (C / 4);
  hipLaunchKernelGGL(upsample2_zero_kernel, dim3(fgrid(total, 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, B,
                     H, W, C / 4);
  return check_launch("upsample2_zero_kernel");
}

}  // extern "C"
