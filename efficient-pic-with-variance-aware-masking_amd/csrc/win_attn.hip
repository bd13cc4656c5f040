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

}  // namespace vam

using namespace vam;

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
