// Backward kernels for the REM fine-tune step (BASELINE configs[4]: `--training_type rems`,
// reference train.py:223-226 freezes everything except `post_latent`, loss = RateLoss).  Only the
// Rate-Enhancement blocks train, so the backward pass is: noisy Gaussian likelihood -> (mu', sigma')
// -> LatentRateReduction (conv3x3 / LeakyReLU / 1x1 skips, reference layers/rem.py:37-141).
//   * data gradients of a convolution reuse conv_igemm_kernel with weights repacked by
//     vam_pack_conv_weights(VAM_PACK_CONV_DGRAD) (channels swapped, taps flipped);
//   * weight gradients: wgrad_kernel below (GEMM over pixels on the fp32 matrix cores);
//   * element-wise pieces: LeakyReLU backward, products, and the likelihood's forward (with
//     additive uniform noise, entropy_models.py:132-138,620-652) and backward (incl. compressai's
//     LowerBound gradient rule).
// These layers are 32-96 channels on a 16x16 latent grid: 0.3 % of the step's FLOPs, so the
// kernels are written for clarity and determinism (fixed reduction order, no float atomics).
#include "common.h"

namespace vam {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// dW[n][c_off + c][ty][tx] = sum_p dY[p][n] * X[pix(p) + (ty - pad, tx - pad)][c]      (stride 1)
// grid = (taps * N/32 tiles * C/32 tiles * pixel splits [max over the group], problems).  8 waves split the pixels of the
// block's range in batches of 32 (16 MFMAs with all 32 loads of a batch in flight: the loop is latency-bound otherwise);
// fixed-order LDS reduction.  The block of tap 0 / channel tile 0 / c_off 0 also produces db[n] = sum_p dY[p][n].
// Pixel splits (first-stage training: 1x1 layers on 131072 ... 524288 pixels have only (N/32)(C/32) = 18 ... 36 weight
// tiles, far fewer than the chip's 256 CUs): split s of S handles a contiguous pixel range and writes its partial tile to
// the caller-owned workspace [S][N][C][taps] (+ [S][N] for the bias); wgrad_reduce_kernel adds the S partials in split
// order.  Deterministic either way: no float atomics.
constexpr int WG_WAVES = 8, WG_BATCH = 16;   // MFMAs per batch; each covers 2 pixels


struct WgradArgs {
  vam_wgrad p[VAM_MAX_WGRAD_GROUP];
};

// TN x TC = 32x32 blocks of the weight tile one workgroup owns.  With a 32x32 tile every MFMA needs two vector loads (one
// row of dY, one of X per pixel pair) and a [192->192 k5] layer reads its two tensors 25 x 6 times over (30 GB from L2 for
// one launch, 41 TF/s); a 64x64 tile issues four MFMAs per four loads and halves that traffic.  Chosen per launch by
// wgrad_tile(): 2x2 where both channel counts are multiples of 64, 3x1 for the 96-channel layers, 2x1 / 1x2 / 1x1 otherwise.
// SPLIT (default; VAMPIC_WGRAD=f32 keeps the fp32-pipe loop): the same sum on the bf16 matrix pipe.  K of this GEMM is the
// pixel axis and BOTH operands are activations, so both are split on the fly: a lane of v_mfma_f32_32x32x16_bf16 holds 8
// consecutive pixels of one channel — 8 dword loads (each still a coalesced 128-byte row across the 32 lanes, the same number
// of load instructions per pixel as the fp32 loop), split exactly into hi / mid / lo bf16 planes (truncation, as the
// convolution kernel's), six products per 16 pixels and block instead of eight fp32 MFMAs of twice the length: 192 vs 512
// matrix-pipe cycles.  Products are exact either way; the fp32 accumulation order differs, the result is as deterministic.
template <int TN, int TC, int SPLIT>      // SPLIT: 0 = fp32 pipe, n > 0 = bf16 pipe with n 16-pixel steps per loop iteration (1 is used)
__global__ __launch_bounds__(WG_WAVES * 64) void wgrad_kernel(const WgradArgs args) {
  __shared__ float red[WG_WAVES][32][33];
  __shared__ float redb[WG_WAVES][2][32];
  const vam_wgrad& pr = args.p[blockIdx.y];
  const int kh = pr.kh, kw = pr.kw, taps = kh * kw;
  const int n_tiles = (pr.N + 32 * TN - 1) / (32 * TN), c_tiles = (pr.C + 32 * TC - 1) / (32 * TC);
  const int per = taps * n_tiles * c_tiles;
  const int S = pr.splits > 1 ? pr.splits : 1;
  int bid = blockIdx.x;
  if (bid >= per * S) return;
  const int split = bid / per;
  bid -= split * per;
  const int tap = bid % taps;
  bid /= taps;
  const int n0 = (bid % n_tiles) * (32 * TN), c0 = (bid / n_tiles) * (32 * TC);
  const int ty = tap / kw, tx = tap % kw, pad_y = kh / 2, pad_x = kw / 2;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int H = pr.H, W = pr.W, HW = H * W;
  const int stride = pr.stride == 2 ? 2 : 1;                       // 2: a stride-2 / pad k/2 convolution (x is [B, Hx, Wx])
  const int Hx = stride == 2 ? pr.Hx : H, Wx = stride == 2 ? pr.Wx : W;
  const long HWx = (long)Hx * Wx;
  const long P = (long)pr.B * HW;
  const long chunk = ((P + S - 1) / S + 2 * WG_BATCH - 1) / (2 * WG_BATCH) * (2 * WG_BATCH);
  const long p_begin = (long)split * chunk;
  const long p_end = p_begin + chunk < P ? p_begin + chunk : P;
  const float* __restrict__ x = pr.x;
  const float* __restrict__ dy = pr.dy;
  const int ld_x = pr.ld_x, ld_dy = pr.ld_dy;
  bool n_ok[TN], c_ok[TC];
#pragma unroll
  for (int u = 0; u < TN; ++u) n_ok[u] = n0 + 32 * u + l31 < pr.N;
#pragma unroll
  for (int u = 0; u < TC; ++u) c_ok[u] = c0 + 32 * u + l31 < pr.C;
  const bool want_db = pr.db != nullptr && tap == 0 && c0 == 0 && pr.c_off == 0;
  f32x16 acc[TN][TC];
#pragma unroll
  for (int u = 0; u < TN; ++u)
#pragma unroll
    for (int v = 0; v < TC; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;
  float bsum[TN];
#pragma unroll
  for (int u = 0; u < TN; ++u) bsum[u] = 0.f;
  if constexpr (SPLIT) {
    // exact hi / mid / lo planes of 8 values (truncation split), packed two bf16 per dword
    auto split8 = [](const float (&f)[8], bf16x8 (&pl)[3]) {
      unsigned hb[8], mb[8], lb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        hb[e] = __float_as_uint(f[e]);
        const float r1 = f[e] - __uint_as_float(hb[e] & 0xFFFF0000u);
        mb[e] = __float_as_uint(r1);
        lb[e] = __float_as_uint(r1 - __uint_as_float(mb[e] & 0xFFFF0000u));
      }
      u32x4 h, m, l;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h[q] = __builtin_amdgcn_perm(hb[2 * q + 1], hb[2 * q], 0x07060302u);
        m[q] = __builtin_amdgcn_perm(mb[2 * q + 1], mb[2 * q], 0x07060302u);
        l[q] = __builtin_amdgcn_perm(lb[2 * q + 1], lb[2 * q], 0x07060302u);
      }
      pl[0] = __builtin_bit_cast(bf16x8, h);
      pl[1] = __builtin_bit_cast(bf16x8, m);
      pl[2] = __builtin_bit_cast(bf16x8, l);
    };
    // 16-pixel MFMA steps per loop iteration, all their loads issued first.  Two (SPLIT = 2) measured no better over a
    // first_train step: 88.2 vs 82.3 ms of weight gradients — the deep-pixel layers gain 5 %, the pixel-split ones (a few
    // hundred pixels per wave) lose 30 %; choosing per launch landed in between (86.4 ms).  One it is.
    constexpr int NH = SPLIT;
    // loads without branches: 32-bit byte offsets against wave-uniform descriptors, anything outside the pixel range, the image
    // or the channel count gets the out-of-range offset and reads 0 (the host sends tensors of 2 GiB or more down the fp32 loop)
    auto desc = [](const void* q) {
      const unsigned long long a = reinterpret_cast<unsigned long long>(q);
      return __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a)), 0, 0x7FFFFFFF, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t r_dy = desc(dy), r_x = desc(x);
    const unsigned OOB = 0x80000000u;
    for (long pb = p_begin + (long)wid * (16 * NH); pb < p_end; pb += (long)WG_WAVES * 16 * NH) {
      float af[NH][TN][8], bf[NH][TC][8];
#pragma unroll
      for (int hf = 0; hf < NH; ++hf) {
        // this lane's 8 pixels of the step: pb + 16 hf + 8 lh + j; position of the first by division, the rest by carry
        const long pf = pb + 16 * hf + 8 * lh;
        int bi = (int)(pf / HW);
        const int r0 = (int)(pf - (long)bi * HW);
        int oy = r0 / W, ox = r0 - oy * W;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long p = pf + j;
          const bool pin = p < p_end;
          const unsigned o_dy = (unsigned)((int)p * ld_dy + n0 + l31) << 2;
#pragma unroll
          for (int u = 0; u < TN; ++u)
            af[hf][u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_dy, (int)((pin && n_ok[u]) ? o_dy + 128u * u : OOB), 0, 0));
          const int iy = oy * stride - pad_y + ty, ix = ox * stride - pad_x + tx;
          const bool xin = pin && (unsigned)iy < (unsigned)Hx && (unsigned)ix < (unsigned)Wx;
          const unsigned o_x = (unsigned)((bi * (int)HWx + iy * Wx + ix) * ld_x + c0 + l31) << 2;
#pragma unroll
          for (int v = 0; v < TC; ++v)
            bf[hf][v][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_x, (int)((xin && c_ok[v]) ? o_x + 128u * v : OOB), 0, 0));
          ++ox;
          if (ox == W) { ox = 0; ++oy; if (oy == H) { oy = 0; ++bi; } }
        }
      }
#pragma unroll
      for (int hf = 0; hf < NH; ++hf) {
        bf16x8 fa[TN][3], fb[TC][3];
#pragma unroll
        for (int u = 0; u < TN; ++u) {
          split8(af[hf][u], fa[u]);
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[u] += af[hf][u][j];
        }
#pragma unroll
        for (int v = 0; v < TC; ++v) split8(bf[hf][v], fb[v]);
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
          for (int v = 0; v < TC; ++v) {
            // smallest terms first, as in the convolution kernel: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
            acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][0], fb[v][2], acc[u][v], 0, 0, 0);
            acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][2], fb[v][0], acc[u][v], 0, 0, 0);
            acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][1], fb[v][1], acc[u][v], 0, 0, 0);
            acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][0], fb[v][1], acc[u][v], 0, 0, 0);
            acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][1], fb[v][0], acc[u][v], 0, 0, 0);
            acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][0], fb[v][0], acc[u][v], 0, 0, 0);
          }
      }
    }
  } else
  for (long pb = p_begin + (long)wid * (2 * WG_BATCH); pb < p_end; pb += (long)WG_WAVES * 2 * WG_BATCH) {
    float a[WG_BATCH][TN], b[WG_BATCH][TC];
#pragma unroll
    for (int i = 0; i < WG_BATCH; ++i) {
      const long p = pb + 2 * i + lh;
#pragma unroll
      for (int u = 0; u < TN; ++u) a[i][u] = 0.f;
#pragma unroll
      for (int v = 0; v < TC; ++v) b[i][v] = 0.f;
      if (p < p_end) {
#pragma unroll
        for (int u = 0; u < TN; ++u)
          if (n_ok[u]) a[i][u] = dy[p * ld_dy + n0 + 32 * u + l31];
        const int bi = (int)(p / HW);
        const int r = (int)(p - (long)bi * HW);
        const int oy = r / W, ox = r - oy * W;
        const int iy = oy * stride - pad_y + ty, ix = ox * stride - pad_x + tx;
        if ((unsigned)iy < (unsigned)Hx && (unsigned)ix < (unsigned)Wx) {
          const float* xp = x + ((long)bi * HWx + (long)iy * Wx + ix) * ld_x + c0 + l31;
#pragma unroll
          for (int v = 0; v < TC; ++v)
            if (c_ok[v]) b[i][v] = xp[32 * v];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WG_BATCH; ++i) {
#pragma unroll
      for (int u = 0; u < TN; ++u) {
#pragma unroll
        for (int v = 0; v < TC; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][u], b[i][v], acc[u][v], 0, 0, 0);
        bsum[u] += a[i][u];
      }
    }
  }
  float* part = S > 1 ? pr.workspace + (size_t)split * ((size_t)pr.N * pr.C * taps + pr.N) : nullptr;
  // the 32x32 blocks of the tile go through the reduction buffer one after the other (fixed order over the 8 waves)
#pragma unroll
  for (int u = 0; u < TN; ++u) {
#pragma unroll
    for (int v = 0; v < TC; ++v) {
      if (u + v > 0) __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wid][(r & 3) + 8 * (r >> 2) + 4 * lh][l31] = acc[u][v][r];
      if (v == 0) redb[wid][lh][l31] = bsum[u];
      __syncthreads();
      const int nb = n0 + 32 * u, cb = c0 + 32 * v;
      for (int i = threadIdx.x; i < 32 * 32; i += WG_WAVES * 64) {
        const int n = i >> 5, c = i & 31;
        if (nb + n < pr.N && cb + c < pr.C) {
          float t = red[0][n][c];
#pragma unroll
          for (int k = 1; k < WG_WAVES; ++k) t += red[k][n][c];
          if (part) part[((size_t)(nb + n) * pr.C + cb + c) * taps + tap] = t;
          else pr.dw[(((long)(nb + n) * pr.cin_total + pr.c_off + cb + c) * kh + ty) * kw + tx] = t;
        }
      }
      if (v == 0 && want_db && threadIdx.x < 32 && nb + threadIdx.x < pr.N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < WG_WAVES; ++k) t += redb[k][0][threadIdx.x] + redb[k][1][threadIdx.x];
        if (part) part[(size_t)pr.N * pr.C * taps + nb + threadIdx.x] = t;
        else pr.db[nb + threadIdx.x] = t;
      }
    }
  }
}

// dw / db = sum over the pixel splits, in split order (problems without splits: nothing to do)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs args) {
  const vam_wgrad& pr = args.p[blockIdx.y];
  const int S = pr.splits;
  if (S <= 1) return;
  const int taps = pr.kh * pr.kw;
  const size_t n_w = (size_t)pr.N * pr.C * taps, stride = n_w + pr.N;
  const bool want_db = pr.db != nullptr && pr.c_off == 0;
  const size_t total = n_w + (want_db ? pr.N : 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float v = pr.workspace[i];
    for (int s = 1; s < S; ++s) v += pr.workspace[(size_t)s * stride + i];
    if (i < n_w) {
      const int tap = (int)(i % taps);
      const size_t nc = i / taps;
      const int c = (int)(nc % pr.C), n = (int)(nc / pr.C);
      pr.dw[((size_t)n * pr.cin_total + pr.c_off + c) * taps + tap] = v;
    } else {
      pr.db[i - n_w] = v;
    }
  }
}

// db[n] = sum_p dY[p][n]: blockIdx.y-th slice of the pixels into partial[y][n] (fixed order inside), then colsum_reduce
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, int ld, long P, int N,
                                                     float* __restrict__ out) {
  __shared__ float red[8][32];
  const int n = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  const long chunk = (P + gridDim.y - 1) / gridDim.y;
  const long p0 = (long)blockIdx.y * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
  float s = 0.f;
  if (n < N)
    for (long p = p0 + g; p < p1; p += 8) s += dy[p * ld + n];
  red[g][threadIdx.x & 31] = s;
  __syncthreads();
  if (g == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
    out[(size_t)blockIdx.y * N + n] = t;
  }
}

__global__ void colsum_reduce_kernel(const float* __restrict__ partial, int S, int N, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float v = partial[n];
  for (int s = 1; s < S; ++s) v += partial[(size_t)s * N + n];
  out[n] = v;
}

__global__ void leaky_bwd_kernel(const float* __restrict__ act, int ld_a, const float* __restrict__ dy, int ld_dy,
                                 float* __restrict__ dx, int ld_dx, long n_vec, int C4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long)gridDim.x * blockDim.x) {
    long p = i / C4;
    int c = (int)(i - p * C4) * 4;
    float4 a = *reinterpret_cast<const float4*>(act + p * ld_a + c);
    float4 g = *reinterpret_cast<const float4*>(dy + p * ld_dy + c);
    *reinterpret_cast<float4*>(dx + p * ld_dx + c) =
        make_float4(g.x * (a.x > 0.f ? 1.f : 0.01f), g.y * (a.y > 0.f ? 1.f : 0.01f), g.z * (a.z > 0.f ? 1.f : 0.01f),
                    g.w * (a.w > 0.f ? 1.f : 0.01f));
  }
}

__global__ void mul_kernel(const float* __restrict__ a, int ld_a, const float* __restrict__ b, int ld_b,
                           float* __restrict__ out, int ld_out, long n_vec, int C4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long)gridDim.x * blockDim.x) {
    long p = i / C4;
    int c = (int)(i - p * C4) * 4;
    float4 x = *reinterpret_cast<const float4*>(a + p * ld_a + c);
    float4 y = *reinterpret_cast<const float4*>(b + p * ld_b + c);
    *reinterpret_cast<float4*>(out + p * ld_out + c) = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
  }
}

// ---- training-mode Gaussian likelihood (additive-noise quantisation proxy)
//   v   = ((y - y2) - mu) * m + noise          (progressive: pic.py:437-442 / rem_pic.py:387-390; m, y2 optional)
//   s   = max(sigma * m, 0.11)                  LowerBound
//   lik = max(Phi((.5-|v|)/s) - Phi((-.5-|v|)/s), 1e-9)
struct LikArgs {
  const float *y, *y2, *mu, *sigma, *mask, *noise, *glik;
  float *lik, *dmu, *dsigma;
  int ld_y, ld_y2, ld_mu, ld_sigma, ld_mask, ld_noise, ld_glik, ld_lik, ld_dmu, ld_dsigma, C4;
  long n_vec;
};

__device__ __forceinline__ float phi_cdf(float t) { return 0.5f * erfcf(-0.70710678118654752440f * t); }
__device__ __forceinline__ float phi_pdf(float t) { return 0.3989422804014327f * expf(-0.5f * t * t); }

template <bool BWD>
__global__ void gauss_train_kernel(const LikArgs a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_vec; i += (long)gridDim.x * blockDim.x) {
    long p = i / a.C4;
    int c = (int)(i - p * a.C4) * 4;
    float4 y4 = *reinterpret_cast<const float4*>(a.y + p * a.ld_y + c);
    if (a.y2) {
      float4 t = *reinterpret_cast<const float4*>(a.y2 + p * a.ld_y2 + c);
      y4.x -= t.x; y4.y -= t.y; y4.z -= t.z; y4.w -= t.w;
    }
    const float4 mu4 = *reinterpret_cast<const float4*>(a.mu + p * a.ld_mu + c);
    const float4 sg4 = *reinterpret_cast<const float4*>(a.sigma + p * a.ld_sigma + c);
    const float4 nz4 = *reinterpret_cast<const float4*>(a.noise + p * a.ld_noise + c);
    float4 m4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.mask) m4 = *reinterpret_cast<const float4*>(a.mask + p * a.ld_mask + c);
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BWD) g4 = *reinterpret_cast<const float4*>(a.glik + p * a.ld_glik + c);
    const float yv[4] = {y4.x, y4.y, y4.z, y4.w}, mv[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, sv[4] = {sg4.x, sg4.y, sg4.z, sg4.w};
    const float nv[4] = {nz4.x, nz4.y, nz4.z, nz4.w}, kv[4] = {m4.x, m4.y, m4.z, m4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
    float o0[4], o1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = yv[k] - mv[k];
      const float v = a.mask ? d * kv[k] + nv[k] : d + nv[k];
      const float raw_s = a.mask ? sv[k] * kv[k] : sv[k];
      const float s = fmaxf(raw_s, 0.11f);
      const float av = fabsf(v);
      const float u = (0.5f - av) / s, l = (-0.5f - av) / s;
      const float lik_raw = phi_cdf(u) - phi_cdf(l);
      if (!BWD) {
        o0[k] = fmaxf(lik_raw, 1e-9f);
      } else {
        // LowerBound backward: pass where x >= bound or the incoming gradient is negative
        float g = gv[k];
        if (!(lik_raw >= 1e-9f || g < 0.f)) g = 0.f;
        const float pu = phi_pdf(u), pl = phi_pdf(l);
        const float dlik_dv = (av == 0.f ? 0.f : (v > 0.f ? 1.f : -1.f)) * (pl - pu) / s;      // d|v|/dv * dlik/d|v|
        float gs = g * (pl * l - pu * u) / s;                                                 // dlik/ds = (-pu*u + pl*l)/s
        if (!(raw_s >= 0.11f || gs < 0.f)) gs = 0.f;                                          // LowerBound(0.11) rule
        o0[k] = -g * dlik_dv * kv[k];        // d/dmu : v = (.. - mu) * m
        o1[k] = gs * kv[k];                  // d/dsigma: s = sigma * m
      }
    }
    if (!BWD) {
      *reinterpret_cast<float4*>(a.lik + p * a.ld_lik + c) = make_float4(o0[0], o0[1], o0[2], o0[3]);
    } else {
      *reinterpret_cast<float4*>(a.dmu + p * a.ld_dmu + c) = make_float4(o0[0], o0[1], o0[2], o0[3]);
      *reinterpret_cast<float4*>(a.dsigma + p * a.ld_dsigma + c) = make_float4(o1[0], o1[1], o1[2], o1[3]);
    }
  }
}

// LDS-tiled kernel for the k3 / k5 layers (wgrad_lds.hip)
bool wgrad2_eligible(const vam_wgrad& p);
bool wgrad2_grid(int H, int W);
int wgrad2_splits(const vam_wgrad& p);
int wgrad2_launch_class(const vam_wgrad* probs, int n, hipStream_t stream);

static inline unsigned sgrid(long n, int block) {
  long g = (n + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
static bool a16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace vam

using namespace vam;

extern "C" {

// 32x32 blocks per workgroup tile (wgrad_kernel<TN, TC>): wide where it costs little padding
static void wgrad_tile(const vam_wgrad& p, int* tn, int* tc) {
  auto pad_ok = [](int n, int t) { return (double)(cdiv(n, 32 * t) * 32 * t) <= 1.15 * n; };
  // bf16-pipe loop: every fragment is split by the wave that loads it, so operand reuse inside the tile pays twice —
  // 96 x 64 (five fragments for six block pairs) where N is a multiple of 96 (VAMPIC_WGRAD_TILE32=0 keeps 64 x 64 there)
  static int wide = -1;
  if (wide < 0) { const char* e = getenv("VAMPIC_WGRAD_TILE32"); wide = (e && e[0] == '0') ? 0 : 1; }
  const bool wide_here = wide && p.N % 96 == 0 && pad_ok(p.C, 2);    // (a 16-channel input gains nothing from the taller tile)
  int n_ = (p.N % 96 == 0 && (wide_here || p.N % 64 != 0)) ? 3 : (pad_ok(p.N, 2) ? 2 : 1);
  int c_ = ((n_ < 3 || wide_here) && pad_ok(p.C, 2)) ? 2 : 1;
  *tn = n_;
  *tc = c_;
}

static bool wg_split_on() {             // VAMPIC_WGRAD=f32: the fp32-pipe loop (A/B measurements)
  static int wg_split = -1;
  if (wg_split < 0) {
    const char* e = getenv("VAMPIC_WGRAD");
    wg_split = (e && (e[0] == 'f' || e[0] == 'F')) ? 0 : 1;
  }
  return wg_split != 0;
}

static int wgrad_splits(const vam_wgrad& p) {
  if (wgrad2_eligible(p)) return wgrad2_splits(p);
  // enough blocks to fill the chip (4 blocks of 8 waves per CU x 256 CUs, twice over), at least 2048 pixels per split
  int tn, tc;
  wgrad_tile(p, &tn, &tc);
  const long per = (long)p.kh * p.kw * cdiv(p.N, 32 * tn) * cdiv(p.C, 32 * tc);
  const long P = (long)p.B * p.H * p.W;
  long s = (2048 + per - 1) / per;
  const long cap = P / 2048;
  if (s > cap) s = cap;
  if (s > 256) s = 256;
  return s < 2 ? 1 : (int)s;
}

int vam_conv_wgrad_lds_grid(int H, int W) { return wgrad2_grid(H, W) ? 1 : 0; }

int vam_conv_wgrad_plan(const vam_wgrad* p, size_t* workspace_bytes) {
  if (!p || p->B <= 0 || p->H <= 0 || p->W <= 0 || p->C <= 0 || p->N <= 0 || p->kh <= 0) return 1;
  const int s = wgrad_splits(*p);
  if (workspace_bytes) *workspace_bytes = s > 1 ? (size_t)s * ((size_t)p->N * p->C * p->kh * p->kw + p->N) * sizeof(float) : 0;
  return s;
}

int vam_conv_wgrad_group(const vam_wgrad* probs, int n_probs, void* stream) {
  VAM_REQUIRE(probs && n_probs >= 1 && n_probs <= VAM_MAX_WGRAD_GROUP, "vam_conv_wgrad_group: 1..%d problems", VAM_MAX_WGRAD_GROUP);
  long max_red = 0;
  double flops = 0;
  int tns[VAM_MAX_WGRAD_GROUP], tcs[VAM_MAX_WGRAD_GROUP];
  bool small = true;                   // the bf16-pipe loop addresses dY and x with 32-bit byte offsets
  for (int i = 0; i < n_probs; ++i) {
    const vam_wgrad& p = probs[i];
    VAM_REQUIRE(p.x && p.dy && p.dw && p.B > 0 && p.H > 0 && p.W > 0 && p.C > 0 && p.N > 0, "vam_conv_wgrad_group: problem %d: bad arguments", i);
    VAM_REQUIRE((p.kh == 1 || p.kh == 3 || p.kh == 5) && p.kw == p.kh, "vam_conv_wgrad_group: square odd kernels (pad k/2)");
    VAM_REQUIRE(p.stride == 0 || p.stride == 1 || (p.stride == 2 && p.kh >= 3 && p.Hx == 2 * p.H && p.Wx == 2 * p.W),
                "vam_conv_wgrad_group: problem %d: stride %d (1, or 2 with k3 / k5 and x of extent 2H x 2W)", i, p.stride);
    VAM_REQUIRE(p.c_off >= 0 && p.c_off + p.C <= p.cin_total && p.ld_x * ((p.flags & VAM_WGRAD_X_P3) ? 8 : 1) >= p.C && p.ld_dy >= p.N,
                "vam_conv_wgrad_group: problem %d: channel window", i);
    VAM_REQUIRE(p.splits >= 0 && p.splits <= 256 && (p.splits <= 1 || p.workspace), "vam_conv_wgrad_group: problem %d: %d pixel splits need a workspace "
                "(vam_conv_wgrad_plan)", i, p.splits);
    VAM_REQUIRE(!(p.flags & VAM_WGRAD_X_P3) || (wg_split_on() && wgrad2_eligible(p)),
                "vam_conv_wgrad_group: problem %d: a plane (P3) input needs the LDS-tiled kernel (k3 stride 1, C %% 8 == 0, a grid "
                "vam_conv_wgrad_lds_grid accepts, VAMPIC_WGRAD_LDS / VAMPIC_WGRAD not switched off)", i);
    wgrad_tile(p, &tns[i], &tcs[i]);
    if (p.splits > 1) {
      const long tot = (long)p.N * p.C * p.kh * p.kw + p.N;
      max_red = tot > max_red ? tot : max_red;
    }
    flops += 2.0 * p.B * p.H * p.W * (double)p.C * p.N * p.kh * p.kw;
    const double px = (double)p.B * (p.stride == 2 ? (double)p.Hx * p.Wx : (double)p.H * p.W);
    if ((double)p.B * p.H * p.W * p.ld_dy * 4.0 >= 2147483648.0 || px * p.ld_x * 4.0 >= 2147483648.0) small = false;
  }
  const bool use_split = wg_split_on() && small;
  WgradArgs wa;
  for (int i = 0; i < n_probs; ++i) wa.p[i] = probs[i];
  ProfScope ps(VAM_FAM_CONV, (hipStream_t)stream, flops, 0);
  // One launch per tile shape: the problems of a group that share a shape go together.  (A group used to fall back to
  // 32 x 32 tiles as soon as its problems differed — and the first layer of every slice stack is such a group: its
  // input segments are the 320-channel hyper-latents and 32 ... 160 channels of y_hat, so the 320-channel problem ran
  // on the smallest tile too.)
  bool done[VAM_MAX_WGRAD_GROUP] = {};
  // k3 / k5 problems on the supported grids: the LDS-tiled kernel, one launch per (kernel size, stride) class
  for (int i0 = 0; i0 < n_probs; ++i0) {
    if (done[i0] || !use_split || !wgrad2_eligible(probs[i0])) continue;
    vam_wgrad cls[VAM_MAX_WGRAD_GROUP];
    int n_cls = 0;
    for (int i = i0; i < n_probs; ++i) {
      if (done[i] || !wgrad2_eligible(probs[i]) || probs[i].kh != probs[i0].kh || (probs[i].stride == 2) != (probs[i0].stride == 2)) continue;
      done[i] = true;
      cls[n_cls++] = probs[i];
    }
    if (int rc = wgrad2_launch_class(cls, n_cls, (hipStream_t)stream)) return rc;
  }
  for (int i0 = 0; i0 < n_probs; ++i0) {
    if (done[i0]) continue;
    const int tn = tns[i0], tc = tcs[i0];
    WgradArgs sub;
    int n_sub = 0, max_blocks = 0;
    for (int i = i0; i < n_probs; ++i) {
      if (done[i] || tns[i] != tn || tcs[i] != tc) continue;
      done[i] = true;
      const vam_wgrad& p = probs[i];
      sub.p[n_sub++] = p;
      const int S = p.splits > 1 ? p.splits : 1;
      const int nb = p.kh * p.kw * cdiv(p.N, 32 * tn) * cdiv(p.C, 32 * tc) * S;
      max_blocks = nb > max_blocks ? nb : max_blocks;
    }
#define VAM_WG(TN_, TC_) \
    if (tn == TN_ && tc == TC_) {                                                                                                      \
      if (use_split) hipLaunchKernelGGL((wgrad_kernel<TN_, TC_, 1>), dim3(max_blocks, n_sub), dim3(WG_WAVES * 64), 0, (hipStream_t)stream, sub); \
      else hipLaunchKernelGGL((wgrad_kernel<TN_, TC_, 0>), dim3(max_blocks, n_sub), dim3(WG_WAVES * 64), 0, (hipStream_t)stream, sub);           \
    }
    VAM_WG(1, 1) VAM_WG(2, 1) VAM_WG(1, 2) VAM_WG(2, 2) VAM_WG(3, 1) VAM_WG(3, 2)
#undef VAM_WG
    if (int rc = check_launch("wgrad_kernel")) return rc;
  }
  if (max_red > 0) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(max_red, 256) > 1024 ? 1024 : cdiv(max_red, 256), n_probs), dim3(256), 0,
                       (hipStream_t)stream, wa);
    return check_launch("wgrad_reduce_kernel");
  }
  return VAM_OK;
}

int vam_conv_wgrad(const float* x, int ld_x, const float* dy, int ld_dy, int B, int H, int W, int kh, int kw, int C,
                   int N, float* dw, float* db, int cin_total, int c_off, void* stream) {
  vam_wgrad p;
  p.x = x; p.dy = dy; p.dw = dw; p.db = db;
  p.ld_x = ld_x; p.ld_dy = ld_dy; p.B = B; p.H = H; p.W = W; p.kh = kh; p.kw = kw; p.C = C; p.N = N;
  p.cin_total = cin_total; p.c_off = c_off;
  p.stride = 1; p.Hx = H; p.Wx = W;
  p.splits = 1; p.workspace = nullptr;
  return vam_conv_wgrad_group(&p, 1, stream);
}

static int colsum_splits(long n_pix) {
  long s = n_pix / 4096;
  return s < 1 ? 1 : (s > 512 ? 512 : (int)s);
}

size_t vam_colsum_workspace(long n_pix, int N) { return (size_t)colsum_splits(n_pix) * (N > 0 ? N : 0) * sizeof(float); }

int vam_colsum(const float* dy, int ld, long n_pix, int N, float* out, float* workspace, void* stream) {
  VAM_REQUIRE(dy && out && workspace && n_pix > 0 && N > 0 && ld >= N, "vam_colsum: bad arguments");
  const int S = colsum_splits(n_pix);
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 32), S), dim3(256), 0, (hipStream_t)stream, dy, ld, n_pix, N, workspace);
  if (int rc = check_launch("colsum_kernel")) return rc;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, workspace, S, N, out);
  return check_launch("colsum_reduce_kernel");
}

int vam_leaky_bwd(const float* act, int ld_a, const float* dy, int ld_dy, float* dx, int ld_dx, long n_pix, int C,
                  void* stream) {
  VAM_REQUIRE(act && dy && dx && n_pix > 0 && C > 0 && C % 4 == 0 && ld_a % 4 == 0 && ld_dy % 4 == 0 && ld_dx % 4 == 0 && a16(act) && a16(dy) && a16(dx), "vam_leaky_bwd: bad arguments");
  long n_vec = n_pix * (C / 4);
  hipLaunchKernelGGL(leaky_bwd_kernel, dim3(sgrid(n_vec, 256)), dim3(256), 0, (hipStream_t)stream, act, ld_a, dy, ld_dy, dx, ld_dx, n_vec, C / 4);
  return check_launch("leaky_bwd_kernel");
}

int vam_mul(const float* a, int ld_a, const float* b, int ld_b, float* out, int ld_out, long n_pix, int C, void* stream) {
  VAM_REQUIRE(a && b && out && n_pix > 0 && C > 0 && C % 4 == 0 && ld_a % 4 == 0 && ld_b % 4 == 0 && ld_out % 4 == 0 && a16(a) && a16(b) && a16(out), "vam_mul: bad arguments");
  long n_vec = n_pix * (C / 4);
  hipLaunchKernelGGL(mul_kernel, dim3(sgrid(n_vec, 256)), dim3(256), 0, (hipStream_t)stream, a, ld_a, b, ld_b, out, ld_out, n_vec, C / 4);
  return check_launch("mul_kernel");
}

int vam_gauss_train(const float* y, int ld_y, const float* y2, int ld_y2, const float* mu, int ld_mu, const float* sigma,
                    int ld_sigma, const float* mask, int ld_mask, const float* noise, int ld_noise, const float* grad_lik,
                    int ld_glik, float* lik, int ld_lik, float* dmu, int ld_dmu, float* dsigma, int ld_dsigma, long n_pix,
                    int C, void* stream) {
  VAM_REQUIRE(y && mu && sigma && noise && n_pix > 0 && C > 0 && C % 4 == 0, "vam_gauss_train: bad arguments");
  const bool bwd = grad_lik != nullptr;
  VAM_REQUIRE(bwd ? (dmu && dsigma) : (lik != nullptr), "vam_gauss_train: forward needs lik, backward needs dmu and dsigma");
  LikArgs a;
  a.y = y; a.y2 = y2; a.mu = mu; a.sigma = sigma; a.mask = mask; a.noise = noise; a.glik = grad_lik;
  a.lik = lik; a.dmu = dmu; a.dsigma = dsigma;
  a.ld_y = ld_y; a.ld_y2 = ld_y2; a.ld_mu = ld_mu; a.ld_sigma = ld_sigma; a.ld_mask = ld_mask; a.ld_noise = ld_noise;
  a.ld_glik = ld_glik; a.ld_lik = ld_lik; a.ld_dmu = ld_dmu; a.ld_dsigma = ld_dsigma;
  a.C4 = C / 4; a.n_vec = n_pix * (C / 4);
  VAM_REQUIRE(a16(y) && a16(y2) && a16(mu) && a16(sigma) && a16(mask) && a16(noise) && a16(grad_lik) && a16(lik) && a16(dmu) && a16(dsigma), "vam_gauss_train: alignment");
  VAM_REQUIRE(ld_y % 4 == 0 && ld_mu % 4 == 0 && ld_sigma % 4 == 0 && ld_noise % 4 == 0 && (!y2 || ld_y2 % 4 == 0) && (!mask || ld_mask % 4 == 0) && (!bwd || (ld_glik % 4 == 0 && ld_dmu % 4 == 0 && ld_dsigma % 4 == 0)) && (bwd || ld_lik % 4 == 0), "vam_gauss_train: strides");
  ProfScope ps(VAM_FAM_TAIL, (hipStream_t)stream, 0, 4.0 * (double)n_pix * C * 8);
  if (bwd) hipLaunchKernelGGL((gauss_train_kernel<true>), dim3(sgrid(a.n_vec, 256)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((gauss_train_kernel<false>), dim3(sgrid(a.n_vec, 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("gauss_train_kernel");
}

}  // extern "C"
