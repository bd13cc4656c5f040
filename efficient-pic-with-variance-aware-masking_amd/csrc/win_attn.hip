// Windowed multi-head self-attention core of the Swin block used in g_a / g_s
// (reference layers/win_attention.py:84-115 WindowAttention.forward and :153-207
// WinBasedAttention.forward).  The qkv / proj Linear layers run as 1x1 problems of the
// implicit-GEMM kernel; this kernel does, per (window, head):
//     S = (q*scale) k^T + rel_pos_bias + shift_mask(0/-100) ; P = softmax(S) ; O = P v
// torch.roll, window_partition and window_reverse are folded into the token addressing:
// token (i,j) of window (wy,wx) lives at original pixel ((wy*ws+i+shift)%H, (wx*ws+j+shift)%W)
// and its output goes back to the same pixel.
//
// One wave handles 64/ws^2 heads of one window; lane = (head_sub, query token).  K and V of
// the window/head sit in LDS and are read as wave-broadcasts; scores stay in registers.
// Tokens per window <= 64 and head_dim <= 40, so this is VALU work (0.5 % of the FLOPs),
// not an MFMA shape.
#include "common.h"
#include <cstdlib>

namespace vam {

typedef float f32x2 __attribute__((ext_vector_type(2)));   // pairs of fp32: v_pk_fma_f32 does two fmas per instruction

template <int WS, int HD>
__global__ __launch_bounds__(64) void win_attn_kernel(const float* __restrict__ qkv, int ld_qkv,
                                                      float* __restrict__ out, int ld_out,
                                                      const float* __restrict__ table, int B, int H, int W,
                                                      int C, int heads, int shift, float scale) {
  constexpr int N = WS * WS;        // tokens per window
  constexpr int HPW = 64 / N;       // heads per wave
  constexpr int LD = HD + 4;       // rows stay 16-byte aligned: K / V rows are read as wave-broadcast ds_read_b128
  __shared__ float sK[HPW * N * LD];
  __shared__ float sV[HPW * N * LD];

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
  const int sy = wy * WS + ti, sx = wx * WS + tj;          // shifted-grid coordinates
  int oy = sy + shift, ox = sx + shift;                    // roll(-shift): shifted[y] = x[(y+shift)%H]
  if (oy >= H) oy -= H;
  if (ox >= W) ox -= W;
  const size_t pix = ((size_t)b * H + oy) * W + ox;
  const float* base = qkv + pix * ld_qkv + head * HD;

  float q[HD];
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    float4 v = *reinterpret_cast<const float4*>(base + d);
    q[d] = v.x * scale; q[d + 1] = v.y * scale; q[d + 2] = v.z * scale; q[d + 3] = v.w * scale;
  }
  float* kd = sK + (hs * N + tok) * LD;
  float* vd = sV + (hs * N + tok) * LD;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    float4 kv = *reinterpret_cast<const float4*>(base + C + d);
    float4 vv = *reinterpret_cast<const float4*>(base + 2 * C + d);
    kd[d] = kv.x; kd[d + 1] = kv.y; kd[d + 2] = kv.z; kd[d + 3] = kv.w;
    vd[d] = vv.x; vd[d + 1] = vv.y; vd[d + 2] = vv.z; vd[d + 3] = vv.w;
  }
  __syncthreads();

  // region ids of win_attention.py:163-173 on the shifted grid
  auto rid1 = [&](int s, int n) { return shift > 0 ? (s < n - WS ? 0 : (s < n - shift ? 1 : 2)) : 0; };
  const int my_rid = rid1(sy, H) * 3 + rid1(sx, W);

  float s[N];
  float mx = -3.0e38f;
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const float* kr = sK + (hs * N + u) * LD;
    // even / odd head-dim elements accumulate in the two halves of a packed fma, summed at the end
    f32x2 acc2 = {0.f, 0.f};
#pragma unroll
    for (int d = 0; d < HD; d += 2) {
      const f32x2 qq = {q[d], q[d + 1]}, kk = {kr[d], kr[d + 1]};
      acc2 = __builtin_elementwise_fma(qq, kk, acc2);
    }
    float acc = acc2.x + acc2.y;
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
  for (int u = 0; u < N; ++u) {
    s[u] = expf(s[u] - mx);
    sum += s[u];
  }
  const float inv = 1.0f / sum;
  f32x2 o2[HD / 2];
#pragma unroll
  for (int d = 0; d < HD / 2; ++d) o2[d] = f32x2{0.f, 0.f};
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const float* vr = sV + (hs * N + u) * LD;
    const float p = s[u] * inv;
    const f32x2 pp = {p, p};
#pragma unroll
    for (int d = 0; d < HD / 2; ++d) {
      const f32x2 vv = {vr[2 * d], vr[2 * d + 1]};
      o2[d] = __builtin_elementwise_fma(pp, vv, o2[d]);
    }
  }
  float o[HD];
#pragma unroll
  for (int d = 0; d < HD / 2; ++d) { o[2 * d] = o2[d].x; o[2 * d + 1] = o2[d].y; }
  float* dst = out + pix * ld_out + head * HD;
#pragma unroll
  for (int d = 0; d < HD; d += 4)
    *reinterpret_cast<float4*>(dst + d) = make_float4(o[d], o[d + 1], o[d + 2], o[d + 3]);
}


// ---------------------------------------------------------------------------------------------------------------------
// 8 x 8 windows on the matrix pipe (round 3).  The kernel above keeps K and V of a (window, head) in LDS and every lane
// reads every row as a wave-broadcast ds_read_b128: 1,536 of them per wave pace it (222 us per launch at 32x64x64x192,
// 0 % MFMA).  Here one wave still owns one (window, head), but both products run on v_mfma_f32_32x32x2_f32 — fp32
// operands, an exact fma chain, so no bf16 split and the same arithmetic as the FMA loops:
//   S^T = K (Q scale)^T   as 2 x 2 blocks of 32 keys x 32 queries, 12 k-steps of two head-dim elements: the A / B operand
//         of lane (r, h) in step t is K[key r][2t + h] / Q[query r][2t + h] — registers loaded straight from q / k;
//   the block's accumulator holds, per lane, ONE query (its column) and 16 keys (rows (e&3) + 8(e>>2) + 4h): bias, shift
//         mask and the softmax run in the lane's own registers plus one exchange with lane ^ 32;
//   O^T = V^T P^T   sums over keys = the ROW index of the S^T accumulators, so P^T is the B operand of step e as it stands
//         (register e of the block, no conversion, no lane movement); the A operand is V[key of (e, h)][d = r], read from
//         a 6 KB LDS image of V (bank-conflict free: 24-float rows);
//   the O^T accumulator holds one query and head-dim elements in runs of four: 16-byte stores.
// 112 MFMAs of 64 cycles per (window, head): 55 us of matrix time per launch at 32x64x64x192.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int HD>
__global__ __launch_bounds__(64) void win_attn8_mfma_kernel(const float* __restrict__ qkv, int ld_qkv, float* __restrict__ out,
                                                           int ld_out, const float* __restrict__ table, int B, int H, int W,
                                                           int C, int heads, int shift, float scale) {
  static_assert(HD % 4 == 0 && HD <= 32, "head dim: a multiple of 4, one 32-row block of O^T");
  constexpr int WS = 8, N = 64, KS = HD / 2;
  constexpr int NT = (2 * WS - 1) * (2 * WS - 1);           // relative-position table entries of one head
  __shared__ __attribute__((aligned(16))) float sV[N * HD];
  __shared__ float sTab[NT];

  const int lane = threadIdx.x, r31 = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int head = bid % heads;
  bid /= heads;
  const int nWx = W / WS, nWy = H / WS;
  const int wx = bid % nWx;
  bid /= nWx;
  const int wy = bid % nWy;
  const int b = bid / nWy;

  for (int i = lane; i < NT; i += 64) sTab[i] = table[i * heads + head];

  // tokens 32 i + r31 (i = 0, 1): original pixel of window token (ti, tj) under the cyclic shift
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
  // operands of S^T: element 2t + h of this lane's two tokens (q pre-scaled), and this half's share of V -> LDS
  float qa[2][KS], ka[2][KS];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float* base = qkv + pix[i] * ld_qkv + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4 qv = *reinterpret_cast<const float4*>(base + d);
      const float4 kv = *reinterpret_cast<const float4*>(base + C + d);
      qa[i][d / 2] = (h ? qv.y : qv.x) * scale;
      qa[i][d / 2 + 1] = (h ? qv.w : qv.z) * scale;
      ka[i][d / 2] = h ? kv.y : kv.x;
      ka[i][d / 2 + 1] = h ? kv.w : kv.z;
    }
    // V row of token 32 i + r31: half h copies the float4s d/4 = h, h + 2, ...
#pragma unroll
    for (int d = 4 * h; d < HD; d += 8)
      *reinterpret_cast<float4*>(sV + (32 * i + r31) * HD + d) = *reinterpret_cast<const float4*>(base + 2 * C + d);
  }
  f32x16 acc[2][2];          // [key block][query block]
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
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[i][t], qa[j][t], acc[i][j], 0, 0, 0);
  __syncthreads();           // sV, sTab complete

  // ---- bias, shift mask (win_attention.py:163-173), softmax over the keys of each query
  auto rid1 = [&](int sgrid, int n) { return shift > 0 ? (sgrid < n - WS ? 0 : (sgrid < n - shift ? 1 : 2)) : 0; };
  float inv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int my_rid = rid1(wy * WS + ti[j], H) * 3 + rid1(wx * WS + tj[j], W);
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ui = 4 * i + (e >> 2), uj = 4 * h + (e & 3);       // key = 32 i + 8 (e >> 2) + 4 h + (e & 3) = 8 ui + uj
        float v = acc[i][j][e] + sTab[(ti[j] - ui + WS - 1) * (2 * WS - 1) + (tj[j] - uj + WS - 1)];
        const int urid = rid1(wy * WS + ui, H) * 3 + rid1(wx * WS + uj, W);
        v = v + (urid != my_rid ? -100.0f : 0.0f);
        acc[i][j][e] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f((acc[i][j][e] - mx) * 1.4426950408889634f);
        acc[i][j][e] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32);
    inv[j] = 1.0f / sum;
  }
  // ---- O^T = V^T P^T: k-step (i, e) pairs key 32 i + 8 (e >> 2) + 4 h + (e & 3) of half h with register e of P^T
  f32x16 o[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[j][e] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = 32 * i + 8 * (e >> 2) + 4 * h + (e & 3);
      const float vt = r31 < HD ? sV[key * HD + r31] : 0.f;          // V^T[d = r31][key]
#pragma unroll
      for (int j = 0; j < 2; ++j) o[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vt, acc[i][j][e], o[j], 0, 0, 0);
    }
  // ---- store: lane = query 32 j + r31, registers 4g .. 4g+3 = head-dim elements 8 g + 4 h + 0..3
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float* dst = out + pix[j] * ld_out + head * HD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 8 * g + 4 * h;
      if (d < HD)
        *reinterpret_cast<float4*>(dst + d) = make_float4(o[j][4 * g] * inv[j], o[j][4 * g + 1] * inv[j], o[j][4 * g + 2] * inv[j],
                                                          o[j][4 * g + 3] * inv[j]);
    }
  }
}

}  // namespace vam

using namespace vam;

// 1 (default): 8 x 8 windows run on the fp32 matrix pipe (forward and backward); 0: the FMA kernels (A/B measurements and
// the equivalence test); -1: follow the environment variable VAMPIC_ATTN_MFMA
static int g_attn_mfma = -1;
extern "C" int vam_attn_set_mfma(int mode) {
  g_attn_mfma = mode < 0 ? -1 : (mode ? 1 : 0);
  return VAM_OK;
}
extern "C" int vam_attn_mfma(void) {
  if (g_attn_mfma < 0) {
    const char* e = getenv("VAMPIC_ATTN_MFMA");
    g_attn_mfma = (e && e[0] == '0') ? 0 : 1;
  }
  return g_attn_mfma;
}

extern "C" int vam_win_attention(const float* qkv, int ld_qkv, float* out, int ld_out, const float* table, int B,
                                 int H, int W, int C, int heads, int ws, int shift, void* stream) {
  VAM_REQUIRE(qkv && out && table && B > 0 && H > 0 && W > 0, "vam_win_attention: bad arguments");
  VAM_REQUIRE(ws == 4 || ws == 8, "vam_win_attention: window size %d (4 or 8)", ws);
  VAM_REQUIRE(H % ws == 0 && W % ws == 0, "vam_win_attention: %dx%d not a multiple of the window %d", H, W, ws);
  VAM_REQUIRE(shift >= 0 && shift < ws, "vam_win_attention: shift %d", shift);
  VAM_REQUIRE(heads > 0 && C % heads == 0 && ld_qkv >= 3 * C && ld_out >= C, "vam_win_attention: channels");
  VAM_REQUIRE(ld_qkv % 4 == 0 && ld_out % 4 == 0 && (((uintptr_t)qkv) & 15) == 0 && (((uintptr_t)out) & 15) == 0, "vam_win_attention: alignment");
  const int hd = C / heads;
  const int hpw = 64 / (ws * ws);
  VAM_REQUIRE(heads % hpw == 0, "vam_win_attention: heads %d not a multiple of %d", heads, hpw);
  const float scale = (float)(1.0 / sqrt((double)hd));
  long nblk = (long)B * (H / ws) * (W / ws) * (heads / hpw);
  VAM_REQUIRE(nblk < (1L << 31), "vam_win_attention: grid too large");
  hipStream_t s = (hipStream_t)stream;
  double tokens = (double)B * H * W;
  ProfScope ps(VAM_FAM_ATTN, s, 4.0 * tokens * ws * ws * C, 4.0 * tokens * 4 * C);
  if (ws == 8 && hd == 24 && vam_attn_mfma()) {
    // one wave per (window, head): fp32 products on the matrix pipe (win_attn8_mfma_kernel)
    const long nb = (long)B * (H / ws) * (W / ws) * heads;
    VAM_REQUIRE(nb < (1L << 31), "vam_win_attention: grid too large");
    hipLaunchKernelGGL((win_attn8_mfma_kernel<24>), dim3((unsigned)nb), dim3(64), 0, s, qkv, ld_qkv, out, ld_out, table, B, H, W, C, heads, shift, scale);
    return check_launch("win_attn8_mfma_kernel");
  }
  if (ws == 8 && hd == 24)
    hipLaunchKernelGGL((win_attn_kernel<8, 24>), dim3((unsigned)nblk), dim3(64), 0, s, qkv, ld_qkv, out, ld_out, table, B, H, W, C, heads, shift, scale);
  else if (ws == 4 && hd == 40)
    hipLaunchKernelGGL((win_attn_kernel<4, 40>), dim3((unsigned)nblk), dim3(64), 0, s, qkv, ld_qkv, out, ld_out, table, B, H, W, C, heads, shift, scale);
  else if (ws == 8 && hd == 40)
    hipLaunchKernelGGL((win_attn_kernel<8, 40>), dim3((unsigned)nblk), dim3(64), 0, s, qkv, ld_qkv, out, ld_out, table, B, H, W, C, heads, shift, scale);
  else if (ws == 4 && hd == 24)
    hipLaunchKernelGGL((win_attn_kernel<4, 24>), dim3((unsigned)nblk), dim3(64), 0, s, qkv, ld_qkv, out, ld_out, table, B, H, W, C, heads, shift, scale);
  else if (ws == 4 && hd == 80)      // single-encoder / single-decoder models: the last attention block has M = 640 channels
    hipLaunchKernelGGL((win_attn_kernel<4, 80>), dim3((unsigned)nblk), dim3(64), 0, s, qkv, ld_qkv, out, ld_out, table, B, H, W, C, heads, shift, scale);
  else {
    set_error("vam_win_attention: unsupported (ws=%d, head_dim=%d); built for head_dim 24, 40 and (ws 4) 80", ws, hd);
    return VAM_EINVAL;
  }
  return check_launch("win_attn_kernel");
}
