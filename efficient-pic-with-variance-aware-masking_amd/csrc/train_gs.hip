// Backward pieces of the synthesis / analysis transforms (reference train.py:150-157,216-218
// `--training_type refine_gs`: everything frozen except g_s[1], loss = DistortionLoss, training/loss.py:126-187).
//   * convolution data gradients reuse conv_igemm_kernel (stride-1 layers: VAM_PACK_CONV_DGRAD weights; the k5/s2
//     transposed convolution's data gradient IS a k5/s2 convolution with the same weight tensor read as OIHW);
//   * weight gradients: wgrad_kernel (csrc/train.hip), generalised to stride 2 for the transposed convolutions;
//   * this file: the element-wise derivatives (exact-erf GELU, sigmoid gate, GDN / IGDN, clamp + MSE) and the
//     window-attention backward (softmax, q/k/v and relative-position-bias gradients).
// Formulas mirror what autograd derives for layers/layers.py:30-74, layers/gdn.py:62-75, layers/win_attention.py:84-115.
#include "common.h"
#include <cstdlib>

namespace vam {

struct EwArgs {
  const float* in[4];
  float* out[3];
  int ld_in[4], ld_out[3];
  int C4;
  long n_vec;
  float coef;
  int flag;
};

__device__ __forceinline__ float gelu_f(float v) { return vam_gelu(v); }
// d/dv [ v * Phi(v) ] = Phi(v) + v * phi(v)
__device__ __forceinline__ float gelu_d(float v) { return vam_gelu_grad(v); }

enum { EW_GELU_FWD = 0, EW_GELU_BWD, EW_GATE_BWD, EW_GDN_APPLY, EW_GDN_BWD_PREP, EW_GDN_BWD_FIN, EW_CLAMP_BWD, EW_AXPY, EW_GATE_FWD,
       EW_REPARAM_BWD, EW_HTANH_FWD, EW_HTANH_BWD, EW_MASK_SPLIT };

template <int OP>
__global__ void ew_kernel(const EwArgs a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_vec; i += (long)gridDim.x * blockDim.x) {
    const long p = i / a.C4;
    const int c = (int)(i - p * a.C4) * 4;
    auto ld4 = [&](int k) { return *reinterpret_cast<const float4*>(a.in[k] + p * a.ld_in[k] + c); };
    auto st4 = [&](int k, const float* v) { *reinterpret_cast<float4*>(a.out[k] + p * a.ld_out[k] + c) = make_float4(v[0], v[1], v[2], v[3]); };
    float4 t0 = ld4(0);
    const float x0[4] = {t0.x, t0.y, t0.z, t0.w};
    float o0[4], o1[4], o2[4];
    if constexpr (OP == EW_GELU_FWD) {
      for (int k = 0; k < 4; ++k) o0[k] = gelu_f(x0[k]);
      st4(0, o0);
    } else if constexpr (OP == EW_GELU_BWD) {                 // in0 = pre-activation, in1 = dy
      float4 t1 = ld4(1);
      const float g[4] = {t1.x, t1.y, t1.z, t1.w};
      for (int k = 0; k < 4; ++k) o0[k] = g[k] * gelu_d(x0[k]);
      st4(0, o0);
    } else if constexpr (OP == EW_GATE_BWD) {                 // out = a * sigmoid(b) + x: in0 = a, in1 = b (pre-sigmoid), in2 = dout
      float4 t1 = ld4(1), t2 = ld4(2);
      const float b[4] = {t1.x, t1.y, t1.z, t1.w}, g[4] = {t2.x, t2.y, t2.z, t2.w};
      for (int k = 0; k < 4; ++k) {
        const float s = 1.0f / (1.0f + expf(-b[k]));
        o0[k] = g[k] * s;                                     // d a
        o1[k] = g[k] * x0[k] * (s * (1.0f - s));              // d b
      }
      st4(0, o0);
      st4(1, o1);
    } else if constexpr (OP == EW_GDN_APPLY) {                // in0 = x, in1 = norm: y = x * sqrt(norm) (flag) or x * rsqrt(norm)
      float4 t1 = ld4(1);
      const float n[4] = {t1.x, t1.y, t1.z, t1.w};
      for (int k = 0; k < 4; ++k) o0[k] = a.flag ? sqrtf(n[k]) * x0[k] : (1.0f / sqrtf(n[k])) * x0[k];
      st4(0, o0);
    } else if constexpr (OP == EW_GDN_BWD_PREP) {             // in0 = x, in1 = norm, in2 = dy -> out0 = dL/dnorm, out1 = dy * dy/dx|norm, out2 = x^2
      float4 t1 = ld4(1), t2 = ld4(2);
      const float n[4] = {t1.x, t1.y, t1.z, t1.w}, g[4] = {t2.x, t2.y, t2.z, t2.w};
      for (int k = 0; k < 4; ++k) {
        const float r = sqrtf(n[k]);
        if (a.flag) {                                         // inverse: y = x * sqrt(n)
          o0[k] = g[k] * x0[k] / (2.0f * r);
          o1[k] = g[k] * r;
        } else {                                              // y = x / sqrt(n)
          o0[k] = -g[k] * x0[k] / (2.0f * n[k] * r);
          o1[k] = g[k] / r;
        }
        o2[k] = x0[k] * x0[k];
      }
      st4(0, o0);
      st4(1, o1);
      st4(2, o2);
    } else if constexpr (OP == EW_GDN_BWD_FIN) {              // in0 = dy*dy/dx|norm, in1 = x, in2 = gamma^T dL/dnorm: dx = in0 + 2 x in2
      float4 t1 = ld4(1), t2 = ld4(2);
      const float x[4] = {t1.x, t1.y, t1.z, t1.w}, u[4] = {t2.x, t2.y, t2.z, t2.w};
      for (int k = 0; k < 4; ++k) o0[k] = x0[k] + 2.0f * x[k] * u[k];
      st4(0, o0);
    } else if constexpr (OP == EW_CLAMP_BWD) {                // in0 = clamp_(v, 0, 1), in1 = dL/d(clamped): passes inside the clamp
      float4 t1 = ld4(1);
      const float g[4] = {t1.x, t1.y, t1.z, t1.w};
      for (int k = 0; k < 4; ++k) o0[k] = (x0[k] > 0.0f && x0[k] < 1.0f) ? g[k] : 0.0f;
      st4(0, o0);
    } else if constexpr (OP == EW_GATE_FWD) {                 // in0 = a, in1 = b (pre-sigmoid), in2 = x: a * sigmoid(b) + x (layers.py:72-74)
      float4 t1 = ld4(1), t2 = ld4(2);
      const float b[4] = {t1.x, t1.y, t1.z, t1.w}, x[4] = {t2.x, t2.y, t2.z, t2.w};
      for (int k = 0; k < 4; ++k) o0[k] = x0[k] * (1.0f / (1.0f + expf(-b[k]))) + x[k];
      st4(0, o0);
    } else if constexpr (OP == EW_HTANH_FWD) {
      // latent residual prediction (pic.py:635-641): in0 = stack output z, in1 = quantised residual, in2 = base latent:
      // (0.5 tanh(z) + in1) + in2, the order of the convolution epilogue (act, post, post2)
      float4 t1 = ld4(1), t2 = ld4(2);
      const float p1[4] = {t1.x, t1.y, t1.z, t1.w}, p2[4] = {t2.x, t2.y, t2.z, t2.w};
      for (int k = 0; k < 4; ++k) o0[k] = (0.5f * tanhf(x0[k]) + p1[k]) + p2[k];
      st4(0, o0);
    } else if constexpr (OP == EW_HTANH_BWD) {                // in0 = z, in1 = dy: dz = dy * 0.5 (1 - tanh(z)^2)
      float4 t1 = ld4(1);
      const float g[4] = {t1.x, t1.y, t1.z, t1.w};
      for (int k = 0; k < 4; ++k) {
        const float t = tanhf(x0[k]);
        o0[k] = g[k] * (0.5f * (1.0f - t * t));
      }
      st4(0, o0);
    } else if constexpr (OP == EW_MASK_SPLIT) {               // in0 = g, in1 = m: out0 = g * m, out1 = g * (1 - m)
      float4 t1 = ld4(1);
      const float m[4] = {t1.x, t1.y, t1.z, t1.w};
      for (int k = 0; k < 4; ++k) {
        o0[k] = x0[k] * m[k];
        o1[k] = x0[k] * (1.0f - m[k]);
      }
      st4(0, o0);
      st4(1, o1);
    } else if constexpr (OP == EW_REPARAM_BWD) {
      // NonNegativeParametrizer (compressai 1.2.4): value = max(p, bound)^2 - pedestal, LowerBound's gradient rule on the
      // max.  in0 = stored parameter p, in1 = dL/dvalue, coef = bound -> dL/dp
      float4 t1 = ld4(1);
      const float g[4] = {t1.x, t1.y, t1.z, t1.w};
      for (int k = 0; k < 4; ++k) {
        const float go = g[k] * 2.0f * fmaxf(x0[k], a.coef);          // gradient arriving at the LowerBound's output
        o0[k] = (x0[k] >= a.coef || go < 0.0f) ? go : 0.0f;
      }
      st4(0, o0);
    } else {                                                  // EW_AXPY: out = in0 + coef * in1
      float4 t1 = ld4(1);
      const float y[4] = {t1.x, t1.y, t1.z, t1.w};
      for (int k = 0; k < 4; ++k) o0[k] = x0[k] + a.coef * y[k];
      st4(0, o0);
    }
  }
}

// ---------------------------------------------------------------- window attention backward
// One wave per (window, 64/ws^2 heads), lane = (head_sub, token i) as in the forward kernel (csrc/win_attn.hip).
//   S = (q*scale) k^T + bias + shift mask ; P = softmax(S) ; O = P v
//   dV_j = sum_i P_ij dO_i ; dP_ij = dO_i . v_j ; dS_ij = P_ij (dP_ij - sum_j' P_ij' dP_ij') ;
//   dq_i = scale * sum_j dS_ij k_j ; dk_j = sum_i dS_ij (scale q_i) ; dtable[ridx(i,j)][head] += dS_ij
// The relative-position-bias gradient is DETERMINISTIC: a block (one wave) sums its own contributions in LDS (ds_add of a
// single wave executes in lane order), writes them to its row of `partial` [blocks of one head group][HPW * NT], and
// bias_grad_reduce_kernel adds the rows in block order — no global float atomics, so refine_gs / first_train gradients
// are bit-reproducible run to run and rank to rank.
template <int WS, int HD>
__global__ __launch_bounds__(64) void win_attn_bwd_kernel(const float* __restrict__ qkv, int ld_qkv, const float* __restrict__ dout,
                                                          int ld_do, float* __restrict__ dqkv, int ld_dq,
                                                          const float* __restrict__ table, float* __restrict__ partial,
                                                          int B, int H, int W, int C, int heads, int shift, float scale) {
  constexpr int N = WS * WS;
  constexpr int HPW = 64 / N;
  constexpr int LD = HD + 4;
  constexpr int NT = (2 * WS - 1) * (2 * WS - 1);
  extern __shared__ float sm[];
  float* sK = sm;                         // [HPW*N][LD]
  float* sV = sK + HPW * N * LD;
  float* sQ = sV + HPW * N * LD;          // q * scale
  float* sG = sQ + HPW * N * LD;          // dO
  float* sP = sG + HPW * N * LD;          // [HPW][N][N+1]  P, later reused
  float* sS = sP + HPW * N * (N + 1);     // [HPW][N][N+1]  dS
  float* sT = sS + HPW * N * (N + 1);     // [HPW][NT] bias-table gradient of this block

  const int lane = threadIdx.x;
  const int hs = lane / N, tok = lane % N;
  const int groups = heads / HPW;
  int bid = blockIdx.x;
  const int hg = bid % groups;
  bid /= groups;
  const int nWx = W / WS, nWy = H / WS;
  const int wx = bid % nWx;
  bid /= nWx;
  const int wy = bid % nWy;
  const int b = bid / nWy;
  const int head = hg * HPW + hs;
  const int ti = tok / WS, tj = tok % WS;
  const int sy = wy * WS + ti, sx = wx * WS + tj;
  int oy = sy + shift, ox = sx + shift;
  if (oy >= H) oy -= H;
  if (ox >= W) ox -= W;
  const size_t pix = ((size_t)b * H + oy) * W + ox;
  const float* base = qkv + pix * ld_qkv + head * HD;
  const float* gbase = dout + pix * ld_do + head * HD;
  float* mK = sK + (hs * N + tok) * LD;
  float* mV = sV + (hs * N + tok) * LD;
  float* mQ = sQ + (hs * N + tok) * LD;
  float* mG = sG + (hs * N + tok) * LD;
  float q[HD], g[HD];
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const float4 qv = *reinterpret_cast<const float4*>(base + d);
    const float4 kv = *reinterpret_cast<const float4*>(base + C + d);
    const float4 vv = *reinterpret_cast<const float4*>(base + 2 * C + d);
    const float4 gv = *reinterpret_cast<const float4*>(gbase + d);
    q[d] = qv.x * scale; q[d + 1] = qv.y * scale; q[d + 2] = qv.z * scale; q[d + 3] = qv.w * scale;
    g[d] = gv.x; g[d + 1] = gv.y; g[d + 2] = gv.z; g[d + 3] = gv.w;
    mK[d] = kv.x; mK[d + 1] = kv.y; mK[d + 2] = kv.z; mK[d + 3] = kv.w;
    mV[d] = vv.x; mV[d + 1] = vv.y; mV[d + 2] = vv.z; mV[d + 3] = vv.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) { mQ[d + e] = q[d + e]; mG[d + e] = g[d + e]; }
  }
  for (int i = lane; i < HPW * NT; i += 64) sT[i] = 0.f;
  __syncthreads();
  auto rid1 = [&](int s, int n) { return shift > 0 ? (s < n - WS ? 0 : (s < n - shift ? 1 : 2)) : 0; };
  const int my_rid = rid1(sy, H) * 3 + rid1(sx, W);
  float s[N];
  float mx = -3.0e38f;
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const float* kr = sK + (hs * N + u) * LD;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc = fmaf(q[d], kr[d], acc);
    const int ui = u / WS, uj = u % WS;
    const int ridx = (ti - ui + WS - 1) * (2 * WS - 1) + (tj - uj + WS - 1);
    acc = acc + table[ridx * heads + head];
    const int urid = rid1(wy * WS + ui, H) * 3 + rid1(wx * WS + uj, W);
    acc = acc + (urid != my_rid ? -100.0f : 0.0f);
    s[u] = acc;
    mx = fmaxf(mx, acc);
  }
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < N; ++u) { s[u] = expf(s[u] - mx); sum += s[u]; }
  const float inv = 1.0f / sum;
  float dsum = 0.f;
  float dp[N];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    s[u] *= inv;                                       // P_iu
    const float* vr = sV + (hs * N + u) * LD;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc = fmaf(g[d], vr[d], acc);
    dp[u] = acc;                                       // dP_iu
    dsum = fmaf(s[u], acc, dsum);
  }
  float dq[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) dq[d] = 0.f;
  float* rowP = sP + (hs * N + tok) * (N + 1);
  float* rowS = sS + (hs * N + tok) * (N + 1);
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const float ds = s[u] * (dp[u] - dsum);            // dS_iu
    rowP[u] = s[u];
    rowS[u] = ds;
    const float* kr = sK + (hs * N + u) * LD;
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, kr[d], dq[d]);
    const int ui = u / WS, uj = u % WS;
    const int ridx = (ti - ui + WS - 1) * (2 * WS - 1) + (tj - uj + WS - 1);
    atomicAdd(sT + hs * NT + ridx, ds);                // LDS atomic: within a block every (i, j) pair maps to some ridx
  }
  __syncthreads();
  // this lane as KEY / VALUE token `tok`: reductions over the query tokens i
  float dk[HD], dv[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  for (int i = 0; i < N; ++i) {
    const float ds = sS[(hs * N + i) * (N + 1) + tok];
    const float pj = sP[(hs * N + i) * (N + 1) + tok];
    const float* qr = sQ + (hs * N + i) * LD;
    const float* gr = sG + (hs * N + i) * LD;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      dk[d] = fmaf(ds, qr[d], dk[d]);
      dv[d] = fmaf(pj, gr[d], dv[d]);
    }
  }
  float* dst = dqkv + pix * ld_dq + head * HD;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    *reinterpret_cast<float4*>(dst + d) = make_float4(dq[d] * scale, dq[d + 1] * scale, dq[d + 2] * scale, dq[d + 3] * scale);
    *reinterpret_cast<float4*>(dst + C + d) = make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]);
    *reinterpret_cast<float4*>(dst + 2 * C + d) = make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]);
  }
  float* row = partial + (size_t)blockIdx.x * (HPW * NT);
  for (int i = lane; i < HPW * NT; i += 64) row[i] = sT[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// 8 x 8 windows: the backward on the fp32 matrix pipe (round 3).  The kernel above needs 63 KB of LDS per wave (K, V, q,
// dO and two 64 x 65 score images for the transposed pass), i.e. TWO waves per CU: 2.6 ms per launch at 32x64x64x192,
// 21 ms of a first_train step.  Here every product is a v_mfma_f32_32x32x2_f32 chain (exact fp32, as in the forward
// kernel csrc/win_attn.hip win_attn8_mfma_kernel, same operand / accumulator conventions) and the scores are computed in
// BOTH orientations, so that every reduction runs over the row index of an accumulator and nothing is transposed:
//   phase A, accumulators [key][query]:   S^T = K Qs^T,  dP^T = V dO^T  ->  P^T, row sums, dS^T (in the lane's registers:
//            a lane holds one query);  dQs^T = K^T dS^T;  bias-table gradient (LDS adds of one wave: lane order);
//   phase B, accumulators [query][key]:   S = Qs K^T (the same bits as S^T: the MFMA is symmetric in its operands),
//            dP = dO V^T  ->  P, dS with the row statistics of phase A (LDS);  dV^T = dO^T P,  dK^T = Qs^T dS.
// 384 MFMAs per (window, head); LDS 26 KB per wave (q, k, v, dO images + statistics + the bias-gradient table).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int HD>
__global__ __launch_bounds__(64) void win_attn8_bwd_mfma_kernel(const float* __restrict__ qkv, int ld_qkv, const float* __restrict__ dout,
                                                               int ld_do, float* __restrict__ dqkv, int ld_dq,
                                                               const float* __restrict__ table, float* __restrict__ partial,
                                                               int B, int H, int W, int C, int heads, int shift, float scale) {
  static_assert(HD % 4 == 0 && HD <= 32, "head dim: a multiple of 4, one 32-row block");
  constexpr int WS = 8, N = 64, KS = HD / 2, NT = (2 * WS - 1) * (2 * WS - 1);
  __shared__ __attribute__((aligned(16))) float sQ[N * HD], sK[N * HD], sV[N * HD], sG[N * HD];
  __shared__ __attribute__((aligned(16))) float sM[N], sI[N], sR[N];
  __shared__ float sTab[NT], sT[NT];

  const int lane = threadIdx.x, r31 = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int head = bid % heads;
  bid /= heads;
  const int nWx = W / WS, nWy = H / WS;
  const int wx = bid % nWx;
  bid /= nWx;
  const int wy = bid % nWy;
  const int b = bid / nWy;
  for (int i = lane; i < NT; i += 64) { sTab[i] = table[i * heads + head]; sT[i] = 0.f; }

  size_t pix[2];
  int ti[2], tj[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int tok = 32 * i + r31;
    ti[i] = tok >> 3;
    tj[i] = tok & 7;
    int oy = wy * WS + ti[i] + shift, ox = wx * WS + tj[i] + shift;
    if (oy >= H) oy -= H;
    if (ox >= W) ox -= W;
    pix[i] = ((size_t)b * H + oy) * W + ox;
  }
  // LDS images [token][d] of q * scale, k, v, dO: half h of the wave copies the float4s d/4 = h, h+2, ...
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float* base = qkv + pix[i] * ld_qkv + head * HD;
    const float* gb = dout + pix[i] * ld_do + head * HD;
#pragma unroll
    for (int d = 4 * h; d < HD; d += 8) {
      const float4 qv = *reinterpret_cast<const float4*>(base + d);
      const int o = (32 * i + r31) * HD + d;
      *reinterpret_cast<float4*>(sQ + o) = make_float4(qv.x * scale, qv.y * scale, qv.z * scale, qv.w * scale);
      *reinterpret_cast<float4*>(sK + o) = *reinterpret_cast<const float4*>(base + C + d);
      *reinterpret_cast<float4*>(sV + o) = *reinterpret_cast<const float4*>(base + 2 * C + d);
      *reinterpret_cast<float4*>(sG + o) = *reinterpret_cast<const float4*>(gb + d);
    }
  }
  __syncthreads();
  // operand of lane (r31, h) for k-step t: element 2t + h of token 32 i + r31
  auto operands = [&](const float* img, float (&a)[2][KS]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < KS; ++t) a[i][t] = img[(32 * i + r31) * HD + 2 * t + h];
  };
  auto product = [&](f32x16 (&acc)[2][2], const float (&ra)[2][KS], const float (&cb)[2][KS]) {   // acc[i][j] = rows of a-block i x columns of b-block j
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i][t], cb[j][t], acc[i][j], 0, 0, 0);
  };
  auto rid1 = [&](int sgrid, int n) { return shift > 0 ? (sgrid < n - WS ? 0 : (sgrid < n - shift ? 1 : 2)) : 0; };
  // token index of accumulator row e of block i on lane half h
  auto rowtok = [&](int i, int e) { return 32 * i + 8 * (e >> 2) + 4 * h + (e & 3); };
  // out^T[d][col token] = sum over the accumulator rows of img[row token][d] * acc[row][col]; stores float4s of d
  auto reduce_rows = [&](const float* img, const f32x16 (&acc)[2][2], float mul, float* dst_base, int part_off) {
    f32x16 o[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float at = r31 < HD ? img[rowtok(i, e) * HD + r31] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) o[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(at, acc[i][j][e], o[j], 0, 0, 0);
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float* dst = dst_base + pix[j] * ld_dq + part_off + head * HD;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 8 * g + 4 * h;
        if (d < HD)
          *reinterpret_cast<float4*>(dst + d) = make_float4(o[j][4 * g] * mul, o[j][4 * g + 1] * mul, o[j][4 * g + 2] * mul, o[j][4 * g + 3] * mul);
      }
    }
  };

  float opa[2][KS], opb[2][KS];
  f32x16 aS[2][2], aD[2][2];
  // ---------------------------------------------------------------- phase A: [key][query]
  operands(sK, opa);
  operands(sQ, opb);
  product(aS, opa, opb);
  operands(sV, opa);
  operands(sG, opb);
  product(aD, opa, opb);
#pragma unroll
  for (int j = 0; j < 2; ++j) {                      // this lane's query 32 j + r31
    const int my_rid = rid1(wy * WS + ti[j], H) * 3 + rid1(wx * WS + tj[j], W);
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ui = 4 * i + (e >> 2), uj = 4 * h + (e & 3);
        float v = aS[i][j][e] + sTab[(ti[j] - ui + WS - 1) * (2 * WS - 1) + (tj[j] - uj + WS - 1)];
        const int urid = rid1(wy * WS + ui, H) * 3 + rid1(wx * WS + uj, W);
        v = v + (urid != my_rid ? -100.0f : 0.0f);
        aS[i][j][e] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f((aS[i][j][e] - mx) * 1.4426950408889634f);
        aS[i][j][e] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    float rs = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = aS[i][j][e] * inv;
        aS[i][j][e] = p;
        rs = fmaf(p, aD[i][j][e], rs);
      }
    rs += __shfl_xor(rs, 32);
    if (h == 0) { sM[32 * j + r31] = mx; sI[32 * j + r31] = inv; sR[32 * j + r31] = rs; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float ds = aS[i][j][e] * (aD[i][j][e] - rs);                 // dS^T[key][query]
        aS[i][j][e] = ds;
        const int ui = 4 * i + (e >> 2), uj = 4 * h + (e & 3);
        atomicAdd(sT + (ti[j] - ui + WS - 1) * (2 * WS - 1) + (tj[j] - uj + WS - 1), ds);
      }
  }
  reduce_rows(sK, aS, scale, dqkv, 0);               // dq = scale * sum_key dS K   (columns = queries)
  __syncthreads();                                   // statistics visible
  // ---------------------------------------------------------------- phase B: [query][key]
  operands(sQ, opa);
  operands(sK, opb);
  product(aS, opa, opb);
  operands(sG, opa);
  operands(sV, opb);
  product(aD, opa, opb);
#pragma unroll
  for (int j = 0; j < 2; ++j) {                      // this lane's key 32 j + r31
    const int key_rid = rid1(wy * WS + ti[j], H) * 3 + rid1(wx * WS + tj[j], W);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int q0 = 32 * i + 8 * g + 4 * h;       // queries of registers 4g .. 4g+3
        const float4 m4 = *reinterpret_cast<const float4*>(sM + q0), i4 = *reinterpret_cast<const float4*>(sI + q0),
                     r4 = *reinterpret_cast<const float4*>(sR + q0);
        const float mq[4] = {m4.x, m4.y, m4.z, m4.w}, iq[4] = {i4.x, i4.y, i4.z, i4.w}, rq[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int e = 4 * g + c;
          const int qi = 4 * i + g, qj = 4 * h + c;  // window coordinates of query q0 + c
          float v = aS[i][j][e] + sTab[(qi - ti[j] + WS - 1) * (2 * WS - 1) + (qj - tj[j] + WS - 1)];
          const int qrid = rid1(wy * WS + qi, H) * 3 + rid1(wx * WS + qj, W);
          v = v + (key_rid != qrid ? -100.0f : 0.0f);
          const float p = __builtin_amdgcn_exp2f((v - mq[c]) * 1.4426950408889634f) * iq[c];
          aS[i][j][e] = p;                                                   // P[query][key]
          aD[i][j][e] = p * (aD[i][j][e] - rq[c]);                           // dS[query][key]
        }
      }
  }
  reduce_rows(sG, aS, 1.0f, dqkv, 2 * C);            // dv = sum_query P dO     (columns = keys)
  reduce_rows(sQ, aD, 1.0f, dqkv, C);                // dk = sum_query dS (q scale)
  __syncthreads();                                   // (LDS adds of the bias gradient complete)
  float* row = partial + (size_t)blockIdx.x * NT;
  for (int i = lane; i < NT; i += 64) row[i] = sT[i];
}

// dtable[r][hg * hpw + h2] = sum over the blocks of head group hg (block id = win * groups + hg), in block order
__global__ __launch_bounds__(256) void bias_grad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dtable, long n_win,
                                                               int groups, int hpw, int nt, int heads) {
  __shared__ float red[256];
  const int o = blockIdx.x;                       // output element: (hg, h2, r)
  const int hg = o / (hpw * nt), i = o - hg * (hpw * nt);
  const int h2 = i / nt, r = i - h2 * nt;
  float s = 0.f;
  for (long w = threadIdx.x; w < n_win; w += 256) s += partial[((size_t)w * groups + hg) * (hpw * nt) + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {          // fixed tree
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) dtable[r * heads + hg * hpw + h2] = red[0];
}

static inline unsigned sgrid2(long n, int block) {
  long g = (n + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// several AXPY updates of independent windows in one launch (blockIdx.y = job): what ew_kernel<EW_AXPY> computes, per job
struct AxpyJob {
  const float* a;
  const float* b;
  float* o;
  int lda, ldb, ldo, C4;
  long n_vec;
  float coef;
};
struct AxpyGroupArgs {
  AxpyJob j[VAM_MAX_EW_GROUP];
};

__global__ __launch_bounds__(256) void axpy_group_kernel(const AxpyGroupArgs g) {
  const AxpyJob& J = g.j[blockIdx.y];
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < J.n_vec; v += (long)gridDim.x * 256) {
    const long p = v / J.C4;
    const int c = (int)(v - p * J.C4) * 4;
    const float4 x = *reinterpret_cast<const float4*>(J.a + p * J.lda + c), y = *reinterpret_cast<const float4*>(J.b + p * J.ldb + c);
    *reinterpret_cast<float4*>(J.o + p * J.ldo + c) = make_float4(x.x + J.coef * y.x, x.y + J.coef * y.y, x.z + J.coef * y.z, x.w + J.coef * y.w);
  }
}

template <int OP>
static int launch_ew(const EwArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((ew_kernel<OP>), dim3(sgrid2(a.n_vec, 256)), dim3(256), 0, s, a);
  return check_launch("ew_kernel");
}

}  // namespace vam

using namespace vam;

extern "C" {

int vam_train_axpy_group(const vam_ew* jobs, int n, void* stream) {
  VAM_REQUIRE(jobs && n >= 1 && n <= VAM_MAX_EW_GROUP, "vam_train_axpy_group: 1 .. %d jobs", VAM_MAX_EW_GROUP);
  AxpyGroupArgs g;
  long most = 0;
  for (int i = 0; i < n; ++i) {
    const vam_ew& e = jobs[i];
    VAM_REQUIRE(e.n_pix > 0 && e.C > 0 && e.C % 4 == 0, "vam_train_axpy_group: job %d extent", i);
    VAM_REQUIRE(e.in[0].ptr && e.in[1].ptr && e.out[0].ptr, "vam_train_axpy_group: job %d needs in0, in1, out0", i);
    VAM_REQUIRE(e.in[0].ld % 4 == 0 && e.in[1].ld % 4 == 0 && e.out[0].ld % 4 == 0 && e.in[0].ld >= e.C && e.in[1].ld >= e.C && e.out[0].ld >= e.C,
                "vam_train_axpy_group: job %d row pitches", i);
    VAM_REQUIRE((((uintptr_t)e.in[0].ptr | (uintptr_t)e.in[1].ptr | (uintptr_t)e.out[0].ptr) & 15) == 0, "vam_train_axpy_group: job %d alignment", i);
    AxpyJob& J = g.j[i];
    J.a = e.in[0].ptr; J.b = e.in[1].ptr; J.o = const_cast<float*>(e.out[0].ptr);
    J.lda = e.in[0].ld; J.ldb = e.in[1].ld; J.ldo = e.out[0].ld;
    J.C4 = e.C / 4;
    J.n_vec = e.n_pix * (e.C / 4);
    J.coef = e.coef;
    most = J.n_vec > most ? J.n_vec : most;
  }
  for (int i = n; i < VAM_MAX_EW_GROUP; ++i) g.j[i] = AxpyJob{nullptr, nullptr, nullptr, 0, 0, 0, 1, 0, 0.f};
  hipLaunchKernelGGL(axpy_group_kernel, dim3(sgrid2(most, 256), n), dim3(256), 0, (hipStream_t)stream, g);
  return check_launch("axpy_group_kernel");
}

int vam_train_elementwise(int op, const vam_ew* e, void* stream) {
  VAM_REQUIRE(e && e->n_pix > 0 && e->C > 0 && e->C % 4 == 0, "vam_train_elementwise: bad arguments");
  EwArgs a;
  for (int k = 0; k < 4; ++k) {
    a.in[k] = e->in[k].ptr;
    a.ld_in[k] = e->in[k].ld;
    if (a.in[k]) VAM_REQUIRE(a.ld_in[k] % 4 == 0 && ((uintptr_t)a.in[k] & 15) == 0, "vam_train_elementwise: input %d alignment", k);
  }
  for (int k = 0; k < 3; ++k) {
    a.out[k] = const_cast<float*>(e->out[k].ptr);
    a.ld_out[k] = e->out[k].ld;
    if (a.out[k]) VAM_REQUIRE(a.ld_out[k] % 4 == 0 && ((uintptr_t)a.out[k] & 15) == 0, "vam_train_elementwise: output %d alignment", k);
  }
  a.C4 = e->C / 4;
  a.n_vec = e->n_pix * (e->C / 4);
  a.coef = e->coef;
  a.flag = e->flag;
  static const int n_in[] = {1, 2, 3, 2, 3, 3, 2, 2, 3, 2, 3, 2, 2}, n_out[] = {1, 1, 2, 1, 3, 1, 1, 1, 1, 1, 1, 1, 2};
  VAM_REQUIRE(op >= 0 && op <= VAM_EW_MASK_SPLIT, "vam_train_elementwise: op %d", op);
  for (int k = 0; k < n_in[op]; ++k) VAM_REQUIRE(a.in[k], "vam_train_elementwise: op %d needs %d inputs", op, n_in[op]);
  for (int k = 0; k < n_out[op]; ++k) VAM_REQUIRE(a.out[k], "vam_train_elementwise: op %d needs %d outputs", op, n_out[op]);
  hipStream_t s = (hipStream_t)stream;
  switch (op) {
    case VAM_EW_GELU_FWD: return launch_ew<EW_GELU_FWD>(a, s);
    case VAM_EW_GELU_BWD: return launch_ew<EW_GELU_BWD>(a, s);
    case VAM_EW_GATE_BWD: return launch_ew<EW_GATE_BWD>(a, s);
    case VAM_EW_GDN_APPLY: return launch_ew<EW_GDN_APPLY>(a, s);
    case VAM_EW_GDN_BWD_PREP: return launch_ew<EW_GDN_BWD_PREP>(a, s);
    case VAM_EW_GDN_BWD_FIN: return launch_ew<EW_GDN_BWD_FIN>(a, s);
    case VAM_EW_CLAMP_BWD: return launch_ew<EW_CLAMP_BWD>(a, s);
    case VAM_EW_GATE_FWD: return launch_ew<EW_GATE_FWD>(a, s);
    case VAM_EW_REPARAM_BWD: return launch_ew<EW_REPARAM_BWD>(a, s);
    case VAM_EW_HTANH_FWD: return launch_ew<EW_HTANH_FWD>(a, s);
    case VAM_EW_HTANH_BWD: return launch_ew<EW_HTANH_BWD>(a, s);
    case VAM_EW_MASK_SPLIT: return launch_ew<EW_MASK_SPLIT>(a, s);
    default: return launch_ew<EW_AXPY>(a, s);
  }
}

size_t vam_win_attention_bwd_workspace(int B, int H, int W, int heads, int ws) {
  if (B <= 0 || H <= 0 || W <= 0 || heads <= 0 || (ws != 4 && ws != 8)) return 0;
  const size_t nt = (size_t)(2 * ws - 1) * (2 * ws - 1);
  return (size_t)B * (H / ws) * (W / ws) * heads * nt * sizeof(float);      // [windows * head groups][hpw * nt]
}

int vam_win_attention_bwd(const float* qkv, int ld_qkv, const float* dout, int ld_do, float* dqkv, int ld_dq,
                          const float* table, float* dtable, float* workspace, int B, int H, int W, int C, int heads, int ws,
                          int shift, void* stream) {
  VAM_REQUIRE(qkv && dout && dqkv && table && dtable && workspace && B > 0 && H > 0 && W > 0, "vam_win_attention_bwd: bad arguments");
  VAM_REQUIRE((ws == 4 || ws == 8) && H % ws == 0 && W % ws == 0 && shift >= 0 && shift < ws, "vam_win_attention_bwd: window");
  VAM_REQUIRE(heads > 0 && C % heads == 0 && ld_qkv >= 3 * C && ld_dq >= 3 * C && ld_do >= C, "vam_win_attention_bwd: channels");
  VAM_REQUIRE(ld_qkv % 4 == 0 && ld_dq % 4 == 0 && ld_do % 4 == 0 && (((uintptr_t)qkv | (uintptr_t)dout | (uintptr_t)dqkv) & 15) == 0, "vam_win_attention_bwd: alignment");
  const int hd = C / heads, hpw = 64 / (ws * ws), n = ws * ws, nt = (2 * ws - 1) * (2 * ws - 1);
  VAM_REQUIRE(heads % hpw == 0, "vam_win_attention_bwd: heads %d not a multiple of %d", heads, hpw);
  const float scale = (float)(1.0 / sqrt((double)hd));
  long nblk = (long)B * (H / ws) * (W / ws) * (heads / hpw);
  VAM_REQUIRE(nblk < (1L << 31), "vam_win_attention_bwd: grid too large");
  const size_t smem = sizeof(float) * ((size_t)4 * hpw * n * (hd + 4) + (size_t)2 * hpw * n * (n + 1) + (size_t)hpw * nt);
  hipStream_t s = (hipStream_t)stream;
  {
    if (ws == 8 && hd == 24 && vam_attn_mfma()) {
      hipLaunchKernelGGL((win_attn8_bwd_mfma_kernel<24>), dim3((unsigned)nblk), dim3(64), 0, s, qkv, ld_qkv, dout, ld_do, dqkv, ld_dq, table,
                         workspace, B, H, W, C, heads, shift, scale);
      if (int rc_ = check_launch("win_attn8_bwd_mfma_kernel")) return rc_;
      hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3((unsigned)(heads * nt)), dim3(256), 0, s, workspace, dtable,
                         (long)B * (H / ws) * (W / ws), heads / hpw, hpw, nt, heads);
      return check_launch("bias_grad_reduce_kernel");
    }
  }
#define VAM_ATT_BWD(WS_, HD_)                                                                                              \
  if (ws == WS_ && hd == HD_) {                                                                                            \
    static bool attr = false;                                                                                              \
    if (!attr) { (void)hipFuncSetAttribute((const void*)win_attn_bwd_kernel<WS_, HD_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr = true; } \
    hipLaunchKernelGGL((win_attn_bwd_kernel<WS_, HD_>), dim3((unsigned)nblk), dim3(64), smem, s, qkv, ld_qkv, dout, ld_do, dqkv, \
                       ld_dq, table, workspace, B, H, W, C, heads, shift, scale);                                          \
    if (int rc_ = check_launch("win_attn_bwd_kernel")) return rc_;                                                         \
    hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3((unsigned)(heads * nt)), dim3(256), 0, s, workspace, dtable,          \
                       (long)B * (H / ws) * (W / ws), heads / hpw, hpw, nt, heads);                                        \
    return check_launch("bias_grad_reduce_kernel");                                                                        \
  }
  VAM_ATT_BWD(8, 24) VAM_ATT_BWD(4, 40) VAM_ATT_BWD(4, 80)
#undef VAM_ATT_BWD
  set_error("vam_win_attention_bwd: unsupported (ws=%d, head_dim=%d)", ws, hd);
  return VAM_EINVAL;
}

}  // extern "C"
