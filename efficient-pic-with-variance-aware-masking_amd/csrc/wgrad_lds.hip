// Weight gradients of the k3 / k5 convolutions, LDS-tiled (round 4; first-stage training, BASELINE configs[3]:
// autograd of F.conv2d in the reference's training/step.py:32-135).
//
//   dW[n][c_off + c][ty][tx] = sum_p dY[p][n] * X[pix(p) * stride + (ty - pad, tx - pad)][c]
//
// is a GEMM whose K axis is the PIXEL axis, so the matrix pipe wants, per lane, 8 consecutive pixels of one channel —
// the transpose of how NHWC tensors lie in memory.  wgrad_kernel (train.hip) gathers them with 8 dword loads per fragment
// and re-splits every fp32 value into its bf16x3 planes in registers, per wave, per tile and PER TAP: ~45 vector
// instructions per fragment pace the loop (r03: 65 ms of a 160 ms step).  Here a workgroup owns a 128 (n) x 64 (c) tile of
// ONE kernel row ty with all kw taps, and walks its pixel range in chunks of 64 output pixels:
//   * the chunk of dY (64 pixels x 128 channels) and the input rows the kernel row touches ((SEGW - 1) stride + kw pixels
//     per image row x 64 channels) are loaded ONCE with coalesced 16-byte loads, split exactly into hi / mid / lo bf16
//     planes (truncation split, as conv_igemm_kernel) and stored in LDS as [32-channel block][plane][pixel][32 bf16];
//   * operand fragments come out of LDS already transposed: ds_read_b64_tr_b16 hands lane (channel i, pixel group) four
//     consecutive pixels of its channel (64-byte pixel rows: four rows x 64 B = all 64 banks, conflict-free); a tap is a
//     pixel-row offset of the same image, so the kw taps share one staged tile (stride 2: even and odd input columns are
//     kept apart so that a tap's pixels stay consecutive);
//   * each of the 8 waves owns one 32 x 32 block of the tile for all kw taps: per 16-pixel step one dY fragment and kw
//     input fragments feed 6 kw MFMAs (v_mfma_f32_32x32x16_bf16; the six partial products of weight <= 2, smallest first).
// Every element is split once per (tile, kernel row) instead of once per (wave tile, tap), and the loop's vector work is
// what the loader does between two barriers.  Products are exact; the accumulation order is fixed (pixel chunks in order,
// pixel splits reduced in split order by wgrad_reduce_kernel): deterministic, no float atomics.
#include "common.h"

namespace vam {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Wgrad2Args {
  vam_wgrad p[VAM_MAX_WGRAD_GROUP];
};

// A workgroup is WN x WC waves, each owning one 32 x 32 block of the (32 WN) x (32 WC) weight tile for all kw taps.  The
// grid of waves is a template parameter: layers whose channel counts are multiples of 96 (96 -> 96, 192 -> 192, 176 padded
// to 192) waste a quarter of a 128-row tile, 16-channel inputs three quarters of a 64-column one (wgrad2_tile()).

// LDS pixels of the input tile: R image rows of XWL pixel slots each (stride 2: even columns, then odd columns)
__host__ __device__ constexpr int w2_xw(int segw, int kw, int stride) { return (segw - 1) * stride + kw; }
__host__ __device__ constexpr int w2_xwl(int segw, int kw, int stride) {
  return stride == 2 ? 2 * ((w2_xw(segw, kw, stride) + 1) / 2) : w2_xw(segw, kw, stride);
}
__host__ __device__ constexpr int w2_xp_max(int kw, int stride, int kp) {   // over SEGW in {16, 32, 64} <= KP, R = KP / SEGW rows
  const int a = (kp / 16) * w2_xwl(16, kw, stride), b = kp >= 32 ? (kp / 32) * w2_xwl(32, kw, stride) : 0,
            c = kp >= 64 ? w2_xwl(64, kw, stride) : 0;
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

__device__ __forceinline__ void w2_split4(const u32x4 v, uint2 (&pl)[3]) {
  unsigned hb[4], mb[4], lb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float f = __uint_as_float(v[e]);
    hb[e] = v[e];
    const float r1 = f - __uint_as_float(hb[e] & 0xFFFF0000u);
    mb[e] = __float_as_uint(r1);
    lb[e] = __float_as_uint(r1 - __uint_as_float(mb[e] & 0xFFFF0000u));
  }
  pl[0] = make_uint2(__builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u), __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u));
  pl[1] = make_uint2(__builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u), __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u));
  pl[2] = make_uint2(__builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u), __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u));
}

// XP3: the input x arrives as bf16x3 planes ("P3": [pixel][8-channel group][plane][8 bf16], 48 B per group, written by the
// producing convolution's epilogue, VAM_CONV_OUT_BF3 — the taped activations of the slice stacks): its tile is staged by
// plain 16-byte copies, no split.  dY is always fp32 (its column sums are the bias gradient).
template <int KW, int STRIDE, int WN, int WC, int W2_KP, int TN = 1, int TC = 1, int XP3 = 0>
__global__ __launch_bounds__(64 * WN * WC) void wgrad2_kernel(const Wgrad2Args args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // a wave owns TN x TC blocks of 32 x 32 (1 x 1 for the k3 / k5 layers, whose kw taps already give it kw block products
  // per dY fragment; 2 x 1 / 1 x 2 / 2 x 2 for the 1x1 layers, which have one tap)
  constexpr int W2_NB = 32 * WN * TN, W2_CB = 32 * WC * TC, W2_NT = 64 * WN * WC;
  constexpr int LGKP = W2_KP == 64 ? 6 : 5;                   // chunk = 64 or 32 output pixels
  constexpr int W2_SY_BYTES = (W2_NB / 32) * 3 * W2_KP * 64;
  constexpr int UY = W2_NB / 4, UX = W2_CB / 4;               // 16-byte units per pixel of the dY / input tile
  constexpr int PY = W2_NT / UY, PX = W2_NT / UX;             // pixels one pass of the workgroup stages (8 WC, 8 WN)
  constexpr int NYU = (W2_KP + PY - 1) / PY;                  // passes over the 64 dY pixels
  constexpr int XP_MAX = w2_xp_max(KW, STRIDE, W2_KP);
  constexpr int NXU = (XP_MAX + PX - 1) / PX;                 // passes over the input tile's pixels
  constexpr int PAD = KW / 2;
  unsigned char* sY = smem;
  unsigned char* sX = smem + W2_SY_BYTES;
  const vam_wgrad& pr = args.p[blockIdx.y];
  const int N = pr.N, C = pr.C;
  const int n_tiles = (N + W2_NB - 1) / W2_NB, c_tiles = (C + W2_CB - 1) / W2_CB;
  const int per = KW * n_tiles * c_tiles;
  const int S = pr.splits > 1 ? pr.splits : 1;
  int bid = blockIdx.x;
  if (bid >= per * S) return;                                   // whole block leaves: no barrier is skipped by a part of it
  const int split = bid / per;
  bid -= split * per;
  const int ty = bid % KW;
  bid /= KW;
  const int n0 = (bid % n_tiles) * W2_NB, c0 = (bid / n_tiles) * W2_CB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int H = pr.H, W = pr.W, HW = H * W;
  const int Hx = STRIDE == 2 ? pr.Hx : H, Wx = STRIDE == 2 ? pr.Wx : W;
  const int SEGW = W < W2_KP ? W : W2_KP;                       // host: W in {16, 32} or a multiple of 64; H * W % 64 == 0
  const int lg = SEGW == 64 ? 6 : (SEGW == 32 ? 5 : 4);
  const int R = W2_KP >> lg;
  const int XW = (SEGW - 1) * STRIDE + KW;
  const int XWH = (XW + 1) >> 1;
  const int XWL = STRIDE == 2 ? 2 * XWH : XW;
  const int XPL = R * XWL;                                      // LDS pixel slots per (channel block, plane)
  const int n_chunks = (int)(((long)pr.B * HW) >> LGKP);
  const int chunks_per = (n_chunks + S - 1) / S;
  const int q_begin = split * chunks_per;
  const int q_end = q_begin + chunks_per < n_chunks ? q_begin + chunks_per : n_chunks;

  auto desc = [](const void* q) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(q);
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                                (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a)), 0, 0x7FFFFFFF, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t r_dy = desc(pr.dy), r_x = desc(pr.x);
  const unsigned OOB = 0x80000000u;

  // ---- staging roles (fixed per thread over all chunks)
  // dY: unit u = tid + NT i -> pixel tid / UY + PY i, channels 4 (tid % UY) .. + 3 of the tile
  const int ycc = (tid % UY) * 4;                               // channel inside the tile
  const int ypx = tid / UY;
  const bool y_ok = n0 + ycc < N;                               // (host: N % 4 == 0)
  const unsigned y_src = (unsigned)((ypx * pr.ld_dy + n0 + ycc) << 2);                 // + (p0 + PY i) * ld_dy * 4
  const unsigned y_dst = (unsigned)(((ycc >> 5) * 3 * W2_KP + ypx) * 64 + (ycc & 31) * 2);   // + plane * 64 * 64 + PY i * 64
  // X: unit u = tid + NT i -> tile pixel tid / UX + PX i, channels 4 (tid % UX) .. + 3
  const int xcc = (tid % UX) * 4;
  const bool xc_ok = c0 + xcc < C;                              // (host: C % 4 == 0)
  int x_r[NXU], x_j[NXU];
  unsigned x_dst[NXU];
#pragma unroll
  for (int i = 0; i < NXU; ++i) {
    const int xp = tid / UX + PX * i;                           // pixel of the tile in raster order: row r, column j
    const int r = xp / XW, j = xp - r * XW;
    x_r[i] = xp < R * XW ? r : -1;
    x_j[i] = j;
    const int lp = r * XWL + (STRIDE == 2 ? (j & 1) * XWH + (j >> 1) : j);
    x_dst[i] = (unsigned)(((xcc >> 5) * 3 * XPL + lp) * 64 + (xcc & 31) * 2);          // + plane * XPL * 64
  }

  // X as planes: 16-byte unit u = tid + NT i -> tile pixel u / UPX, unit e = u % UPX of the pixel = (group e / 3, plane e % 3)
  constexpr int UPX = (W2_CB / 8) * 3;
  constexpr int NXP = XP3 ? (XP_MAX * UPX + W2_NT - 1) / W2_NT : 1;
  int x3_r[NXP], x3_j[NXP];
  unsigned x3_src[NXP], x3_dst[NXP];
  if constexpr (XP3) {
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
      const int u = tid + W2_NT * i;
      const int xp = u / UPX, e = u - xp * UPX;
      const int gq = e / 3, pl = e - gq * 3;
      const int r = xp / XW, j = xp - r * XW;
      x3_r[i] = (xp < R * XW && c0 + 8 * gq < C) ? r : -1;     // (host: C % 8 == 0 for plane tensors)
      x3_j[i] = j;
      x3_src[i] = (unsigned)((c0 / 8 + gq) * 48 + pl * 16);
      const int lp = r * XWL + (STRIDE == 2 ? (j & 1) * XWH + (j >> 1) : j);
      x3_dst[i] = (unsigned)((((gq >> 2) * 3 + pl) * XPL + lp) * 64 + (gq & 3) * 16);
    }
  }

  u32x4 ry[NYU], rx[XP3 ? NXP : NXU];
  auto issue = [&](int q) {
    const int p0 = q << LGKP;
    const int b = p0 / HW;
    const int rem = p0 - b * HW;
    const int oy0 = rem / W, ox0 = rem - oy0 * W;
#pragma unroll
    for (int i = 0; i < NYU; ++i) {
      const unsigned o = y_src + (unsigned)((p0 + PY * i) * pr.ld_dy) * 4u;
      const bool ok = y_ok && (W2_KP % PY == 0 || ypx + PY * i < W2_KP);
      ry[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_dy, (int)(ok ? o : OOB), 0, 0));
    }
    const int ix0 = ox0 * STRIDE - PAD;
    if constexpr (XP3) {
#pragma unroll
      for (int i = 0; i < NXP; ++i) {
        const int iy = (oy0 + x3_r[i]) * STRIDE - PAD + ty, ix = ix0 + x3_j[i];
        const bool ok = x3_r[i] >= 0 && (unsigned)iy < (unsigned)Hx && (unsigned)ix < (unsigned)Wx;
        const unsigned o = (unsigned)(((b * Hx + iy) * Wx + ix) * pr.ld_x) * 48u + x3_src[i];      // ld_x counts groups
        rx[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)(ok ? o : OOB), 0, 0));
      }
    } else {
#pragma unroll
      for (int i = 0; i < NXU; ++i) {
        const int iy = (oy0 + x_r[i]) * STRIDE - PAD + ty, ix = ix0 + x_j[i];
        const bool ok = xc_ok && x_r[i] >= 0 && (unsigned)iy < (unsigned)Hx && (unsigned)ix < (unsigned)Wx;
        const unsigned o = (unsigned)(((b * Hx + iy) * Wx + ix) * pr.ld_x + c0 + xcc) << 2;
        rx[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)(ok ? o : OOB), 0, 0));
      }
    }
  };
  const bool want_db = pr.db != nullptr && ty == 0 && c0 == 0 && pr.c_off == 0;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NYU; ++i) {
      if (W2_KP % PY == 0 || ypx + PY * i < W2_KP) {
        uint2 pl[3];
        w2_split4(ry[i], pl);
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2*>(sY + y_dst + k * (W2_KP * 64) + i * (PY * 64)) = pl[k];
        if (want_db) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bs[e] += __uint_as_float(ry[i][e]);
        }
      }
    }
    if constexpr (XP3) {
#pragma unroll
      for (int i = 0; i < NXP; ++i)
        if (tid + W2_NT * i < R * XW * UPX)                       // every unit of the tile is written (zeros where out of range)
          *reinterpret_cast<u32x4*>(sX + x3_dst[i]) = rx[i];
    } else {
#pragma unroll
      for (int i = 0; i < NXU; ++i) {
        if (x_r[i] >= 0) {
          uint2 pl[3];
          w2_split4(rx[i], pl);
#pragma unroll
          for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2*>(sX + x_dst[i] + k * (XPL * 64)) = pl[k];
        }
      }
    }
  };

  // ---- compute roles: wave = TN x TC blocks of 32 x 32 of the tile, all KW taps
  const int nb0 = (wid % WN) * TN, cb0 = (wid / WN) * TC;
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  // transposed read: lane 4 q + p of 16-lane group g supplies pixel row q, channels 16 (g & 1) + 4 p .. + 3; the group
  // receives pixels 8 (g >> 1) + 4 h + {0..3} of channel 16 (g & 1) + (lane & 15)
  const unsigned lane_off = (unsigned)((8 * (g >> 1) + qq) * 64 + (16 * (g & 1) + 4 * pp) * 2);
  const unsigned a_base = lane_off + (unsigned)(nb0 * 3 * W2_KP * 64);
  const unsigned b_base = lane_off + (unsigned)(cb0 * 3 * XPL * 64);
  f32x16 acc[TN][TC][KW];
#pragma unroll
  for (int u = 0; u < TN; ++u)
#pragma unroll
    for (int v = 0; v < TC; ++v)
#pragma unroll
      for (int t = 0; t < KW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][v][t][r] = 0.f;

  auto tr8 = [&](const unsigned char* base, unsigned off) -> bf16x8 {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off + 4 * 64));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < W2_KP / 16; ++ks) {
      const int r = (ks * 16) >> lg, oxs = (ks * 16) & (SEGW - 1);
      bf16x8 fa[TN][3];
#pragma unroll
      for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int k = 0; k < 3; ++k) fa[u][k] = tr8(sY, a_base + (unsigned)(((u * 3 + k) * W2_KP + ks * 16) * 64));
      const unsigned row = (unsigned)(r * XWL + oxs);
#pragma unroll
      for (int t = 0; t < KW; ++t) {
        const unsigned px = row + (STRIDE == 2 ? (unsigned)((t & 1) * XWH + (t >> 1)) : (unsigned)t);
#pragma unroll
        for (int v = 0; v < TC; ++v) {
          bf16x8 fb[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) fb[k] = tr8(sX, b_base + (unsigned)((v * 3 + k) * XPL * 64) + px * 64u);
#pragma unroll
          for (int u = 0; u < TN; ++u) {
            // smallest terms first, as in the convolution kernel: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
            acc[u][v][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][0], fb[2], acc[u][v][t], 0, 0, 0);
            acc[u][v][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][2], fb[0], acc[u][v][t], 0, 0, 0);
            acc[u][v][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][1], fb[1], acc[u][v][t], 0, 0, 0);
            acc[u][v][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][0], fb[1], acc[u][v][t], 0, 0, 0);
            acc[u][v][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][1], fb[0], acc[u][v][t], 0, 0, 0);
            acc[u][v][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][0], fb[0], acc[u][v][t], 0, 0, 0);
          }
        }
      }
    }
  };

  // LDS slots the loader never writes must read as zero: pixel slots of the de-interleaved rows beyond XW
  if (STRIDE == 2) {
    for (int i = tid; i < (W2_CB / 32) * 3 * XPL * 4; i += W2_NT) reinterpret_cast<uint4*>(sX)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
  }
  if (q_begin < q_end) {
    issue(q_begin);
    stage();
    __syncthreads();
    for (int q = q_begin; q < q_end; ++q) {
      const bool more = q + 1 < q_end;                          // block-uniform
      if (more) issue(q + 1);                                   // in flight under the MFMAs
      compute();
      __syncthreads();                                          // every wave has read this chunk
      if (more) stage();
      __syncthreads();
    }
  }

  // ---- results: dW (or this split's partial tile), db
  const int taps = KW * KW;
  float* part = S > 1 ? pr.workspace + (size_t)split * ((size_t)N * C * taps + N) : nullptr;
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int u = 0; u < TN; ++u)
#pragma unroll
    for (int v = 0; v < TC; ++v) {
      const int c = c0 + 32 * (cb0 + v) + l31;
#pragma unroll
      for (int t = 0; t < KW; ++t) {
        const int tap = ty * KW + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + 32 * (nb0 + u) + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (n < N && c < C) {
            if (part) part[((size_t)n * C + c) * taps + tap] = acc[u][v][t][r];
            else pr.dw[((size_t)n * pr.cin_total + pr.c_off + c) * taps + tap] = acc[u][v][t][r];
          }
        }
      }
    }
  if (want_db) {                                                // block-uniform
    float* red = reinterpret_cast<float*>(smem);                // [PY pixel groups][NB channels]; the tiles are dead
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ypx * W2_NB + ycc + e] = bs[e];
    __syncthreads();
    if (tid < W2_NB && n0 + tid < N) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < PY; ++k) t += red[k * W2_NB + tid];
      if (part) part[(size_t)N * C * taps + n0 + tid] = t;
      else pr.db[n0 + tid] = t;
    }
  }
}

static size_t w2_lds_bytes(int kw, int stride, int nbk, int cbk, int kp) {      // nbk / cbk: 32-channel blocks of the tile
  return (size_t)nbk * 3 * kp * 64 + (size_t)cbk * 3 * w2_xp_max(kw, stride, kp) * 64;
}

// ---------------------------------------------------------------------------------------------------- host side
// A problem takes this kernel when its geometry is the one the chunking assumes; everything else (1x1 layers, the small
// grids of the hyperprior, odd channel counts) stays with wgrad_kernel.
static bool w2_grid_ok(int H, int W) {
  if (!(W == 16 || W == 32 || (W >= 64 && W % 64 == 0))) return false;
  if (((long)H * W) % 64 != 0) return false;
  if (W < 64 && H % (64 / W) != 0) return false;
  return true;
}
static bool w2_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("VAMPIC_WGRAD_LDS"); on = (e && e[0] == '0') ? 0 : 1; }
  return on != 0;
}
bool wgrad2_grid(int H, int W) { return w2_enabled() && w2_grid_ok(H, W); }

bool wgrad2_eligible(const vam_wgrad& p) {
  if (!w2_enabled()) return false;
  if (!((p.kh == 1 || p.kh == 3 || p.kh == 5) && p.kw == p.kh)) return false;
  if (p.kh == 1 && p.stride == 2) return false;
  const int stride = p.stride == 2 ? 2 : 1;
  if (!w2_grid_ok(p.H, p.W)) return false;
  const bool xp3 = (p.flags & VAM_WGRAD_X_P3) != 0;
  if (xp3 && !(p.kh == 3 && stride == 1 && p.C % 8 == 0)) return false;
  if (p.N % 4 || p.C % 4 || (!xp3 && p.ld_x % 4) || p.ld_dy % 4) return false;
  if ((((uintptr_t)p.x) | ((uintptr_t)p.dy)) & 15) return false;
  const double px = (double)p.B * (stride == 2 ? (double)p.Hx * p.Wx : (double)p.H * p.W);
  if ((double)p.B * p.H * p.W * p.ld_dy * 4.0 >= 2147483648.0 || px * p.ld_x * (xp3 ? 48.0 : 4.0) >= 2147483648.0) return false;
  if (p.C < 16 || p.N < 32) return false;
  return true;
}

// tile of a problem: the candidate with the least padded work (ties: the larger tile, less re-staging)
struct W2Tile { int wn, wc, tn, tc, kp; int nb() const { return 32 * wn * tn; } int cb() const { return 32 * wc * tc; } };
static const W2Tile w2_tiles_k35[] = {{4, 2, 1, 1, 64}, {3, 2, 1, 1, 64}, {2, 2, 1, 1, 64}, {4, 1, 1, 1, 64}, {3, 3, 1, 1, 64}};
// (tiles that hold ALL of N and C of the 96 <-> 192 layers — 192 x 96 / 96 x 192, each operand read once — measured slower:
// 42 vs 48 - 51 TF/s, gpurun_out/r4_wgbench4.log; the instantiations stay for the sweep, the choice does not use them)
static const W2Tile w2_tiles_k1[] = {{3, 2, 2, 1, 32}, {3, 2, 1, 2, 32}, {2, 2, 2, 2, 32}};
static int w2_force_kp() {                                      // VAMPIC_WGRAD_KP=32|64: chunk size (measurements)
  static int kp = -1;
  if (kp < 0) { const char* e = getenv("VAMPIC_WGRAD_KP"); kp = e ? atoi(e) : 0; }
  return kp == 32 || kp == 64 ? kp : 0;
}
W2Tile wgrad2_tile(const vam_wgrad& p) {
  static int fn = -1, fc = -1;                                  // VAMPIC_WGRAD_TILE="wn,wc": force one wave grid (measurements; k3 / k5)
  if (fn < 0) {
    fn = fc = 0;
    if (const char* e = getenv("VAMPIC_WGRAD_TILE")) sscanf(e, "%d,%d", &fn, &fc);
  }
  if (p.kh == 1) {
    W2Tile best = w2_tiles_k1[0];
    double best_cost = 1e300;
    for (const W2Tile& t : w2_tiles_k1) {
      const double cost = (double)cdiv(p.N, t.nb()) * t.nb() * cdiv(p.C, t.cb()) * t.cb() * (t.tn * t.tc == 4 ? 0.97 : 1.0);
      if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return best;
  }
  // Measured choice (scratch/r4_run4.sh: every wave grid x chunk size forced over the k3 / k5 shapes of a first_train step,
  // profiles/r04_wgrad_tile_sweep.txt; TF/s): small tiles with 32-pixel chunks win almost everywhere — their LDS footprint
  // (25 - 40 KB) keeps three or four workgroups on a CU, whose staging phases and barriers then overlap (occupancy beats
  // operand reuse on this chip, as in the convolution kernel): 2 x 2 waves, 64 x 64 tile is the default;
  //   C <= 32 (the 16-channel first / last layers, the 32-channel support segments): 4 x 1 waves, 128 x 32 tile — half or
  //     three quarters of a 64-column tile would be padding;
  //   N a multiple of 96 with few n tiles (96 -> 96 of the residual units): 3 x 2 waves, 96 x 64 tile;
  //   k5 stride 2 below 65536 output pixels: 4 x 2 waves with 64-pixel chunks (the 2 SEGW + 5 input pixels per row make
  //     32-pixel chunks pay two staging rounds per MFMA block there: 136 vs 119 TF/s).
  const bool xp3 = (p.flags & VAM_WGRAD_X_P3) != 0;             // plane inputs: the automatic choice only (three instantiations)
  const int fkp = xp3 ? 0 : w2_force_kp();
  if (fn > 0 && !xp3)
    for (const W2Tile& t : w2_tiles_k35)
      if (t.wn == fn && t.wc == fc) return W2Tile{t.wn, t.wc, 1, 1, fkp ? fkp : (p.stride == 2 ? 32 : 64)};
  const long P = (long)p.B * p.H * p.W;
  W2Tile t{2, 2, 1, 1, 32};
  if (p.C <= 32) t = W2Tile{4, 1, 1, 1, 32};
  else if (p.stride == 2 && p.kh == 5 && P < 65536) t = W2Tile{4, 2, 1, 1, 64};
  else if (p.N % 96 == 0 && p.N <= 96) t = W2Tile{3, 2, 1, 1, 32};
  if (fkp) t.kp = fkp;
  return t;
}

int wgrad2_splits(const vam_wgrad& p) {
  // Pixel splits of ONE problem.  A workgroup slot is a CU x (workgroups that fit its LDS); with `per` tiles (kernel rows x
  // n tiles x c tiles) and S splits the launch runs ceil(per S / slots) rounds of ceil(chunks / S) chunks each.  Pick the S
  // with the shortest makespan (r04: 30 tiles x 18 splits = 540 workgroups on 256 slots ran a third, nearly empty round;
  // 17 splits do not), plus what a split costs: its partial tile is written once and read back by the reduce kernel.
  const W2Tile t = wgrad2_tile(p);
  const int stride = p.stride == 2 ? 2 : 1;
  const long per = (long)p.kh * cdiv(p.N, t.nb()) * cdiv(p.C, t.cb());
  const long chunks = ((long)p.B * p.H * p.W) / t.kp;
  const size_t lds = w2_lds_bytes(p.kh, stride, t.nb() / 32, t.cb() / 32, t.kp);
  long wg_per_cu = (long)(160 * 1024 / lds);
  const long by_waves = 16 / (t.wn * t.wc);                     // four waves per SIMD at most for these register counts
  if (wg_per_cu > by_waves) wg_per_cu = by_waves;
  if (wg_per_cu < 1) wg_per_cu = 1;
  long slots = 256 * wg_per_cu;                                 // MI355X: 256 CUs
  if (p.slot_share > 0.f && p.slot_share != 1.f) {              // the problem shares its launch with others (or, > 1: plans finer)
    slots = (long)(slots * (double)p.slot_share + 0.5);
    if (slots < 1) slots = 1;
  }
  const long min_chunks = 256 / t.kp;                           // at least 256 pixels per split
  long cap = chunks / min_chunks;
  if (cap > 256) cap = 256;
  // time in chunk units: rounds x (a workgroup's fixed cost [descriptors, first loads, its partial tile] + its chunks) + what
  // a split adds beyond that (the reduce kernel reads every partial tile back).  Calibrated on 6x[320 -> 224 k3] at 8192
  // pixels: 8 splits 491 us, 16 splits 648 us.
  // (1x1 layers: a chunk is a third of a k3 chunk's work and the partial tile a third of its bytes, while the fixed cost is
  // the same — more, cheaper splits: 2 weight tiles x 64 splits left half of the CUs idle on the 96-channel layers)
  const double t_fix = (p.kh == 1 ? 10.0 : 6.0) * 32 / t.kp, t_split = (p.kh == 1 ? 0.3 : 1.0) * 32 / t.kp;
  long best_s = 1;
  double best_cost = 1e300;
  for (long s = 1; s <= (cap < 1 ? 1 : cap); ++s) {
    const double rounds = (double)((per * s + slots - 1) / slots);
    const double cost = rounds * (t_fix + (double)((chunks + s - 1) / s)) + (s > 1 ? t_split * s : 0.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best_s = s; }
  }
  return (int)best_s;
}

template <int KW, int STRIDE, int WN, int WC, int KP, int TN = 1, int TC = 1, int XP3 = 0>
static int w2_launch(const Wgrad2Args& a, int max_blocks, int n_sub, hipStream_t s) {
  static bool attr = false;
  const size_t lds = w2_lds_bytes(KW, STRIDE, WN * TN, WC * TC, KP);
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)wgrad2_kernel<KW, STRIDE, WN, WC, KP, TN, TC, XP3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((wgrad2_kernel<KW, STRIDE, WN, WC, KP, TN, TC, XP3>), dim3(max_blocks, n_sub), dim3(64 * WN * WC), lds, s, a);
  return check_launch("wgrad2_kernel");
}

template <int WN, int WC>
static int w2_launch_ks(int kw, int st, int kp, const Wgrad2Args& a, int max_blocks, int n, hipStream_t stream) {
#define W2_KS(KW_, ST_)                                                                      \
  if (kw == KW_ && st == ST_)                                                                \
    return kp == 32 ? w2_launch<KW_, ST_, WN, WC, 32>(a, max_blocks, n, stream) : w2_launch<KW_, ST_, WN, WC, 64>(a, max_blocks, n, stream);
  W2_KS(3, 1) W2_KS(5, 1) W2_KS(3, 2)
#undef W2_KS
  return kp == 32 ? w2_launch<5, 2, WN, WC, 32>(a, max_blocks, n, stream) : w2_launch<5, 2, WN, WC, 64>(a, max_blocks, n, stream);
}

// eligible problems of one (kernel size, stride) class: one launch per tile shape
int wgrad2_launch_class(const vam_wgrad* probs, int n, hipStream_t stream) {
  bool done[VAM_MAX_WGRAD_GROUP] = {};
  const int kw = probs[0].kh, st = probs[0].stride == 2 ? 2 : 1;
  for (int i0 = 0; i0 < n; ++i0) {
    if (done[i0]) continue;
    const W2Tile t = wgrad2_tile(probs[i0]);
    Wgrad2Args a;
    int n_sub = 0, max_blocks = 0;
    for (int i = i0; i < n; ++i) {
      if (done[i]) continue;
      const W2Tile ti = wgrad2_tile(probs[i]);
      if (ti.wn != t.wn || ti.wc != t.wc || ti.kp != t.kp || ti.tn != t.tn || ti.tc != t.tc ||
          ((probs[i].flags ^ probs[i0].flags) & VAM_WGRAD_X_P3)) continue;
      done[i] = true;
      const vam_wgrad& p = probs[i];
      a.p[n_sub++] = p;
      const int S = p.splits > 1 ? p.splits : 1;
      const int nb = p.kh * cdiv(p.N, t.nb()) * cdiv(p.C, t.cb()) * S;
      max_blocks = nb > max_blocks ? nb : max_blocks;
    }
    int rc;
    if (a.p[0].flags & VAM_WGRAD_X_P3) {                        // k3 stride 1, automatic tile, 32-pixel chunks
      if (t.wn == 2 && t.wc == 2) rc = w2_launch<3, 1, 2, 2, 32, 1, 1, 1>(a, max_blocks, n_sub, stream);
      else if (t.wn == 4 && t.wc == 1) rc = w2_launch<3, 1, 4, 1, 32, 1, 1, 1>(a, max_blocks, n_sub, stream);
      else rc = w2_launch<3, 1, 3, 2, 32, 1, 1, 1>(a, max_blocks, n_sub, stream);
    } else if (kw == 1) {
      if (t.tn == 2 && t.tc == 1) rc = w2_launch<1, 1, 3, 2, 32, 2, 1>(a, max_blocks, n_sub, stream);
      else if (t.tn == 1 && t.tc == 2) rc = w2_launch<1, 1, 3, 2, 32, 1, 2>(a, max_blocks, n_sub, stream);
      else if (t.tn == 3 && t.tc == 1) rc = w2_launch<1, 1, 2, 3, 32, 3, 1>(a, max_blocks, n_sub, stream);
      else if (t.tn == 1 && t.tc == 3) rc = w2_launch<1, 1, 3, 2, 32, 1, 3>(a, max_blocks, n_sub, stream);
      else rc = w2_launch<1, 1, 2, 2, 32, 2, 2>(a, max_blocks, n_sub, stream);
    } else if (t.wn == 4 && t.wc == 2) rc = w2_launch_ks<4, 2>(kw, st, t.kp, a, max_blocks, n_sub, stream);
    else if (t.wn == 3 && t.wc == 2) rc = w2_launch_ks<3, 2>(kw, st, t.kp, a, max_blocks, n_sub, stream);
    else if (t.wn == 2 && t.wc == 2) rc = w2_launch_ks<2, 2>(kw, st, t.kp, a, max_blocks, n_sub, stream);
    else if (t.wn == 4 && t.wc == 1) rc = w2_launch_ks<4, 1>(kw, st, t.kp, a, max_blocks, n_sub, stream);
    else rc = w2_launch_ks<3, 3>(kw, st, t.kp, a, max_blocks, n_sub, stream);
    if (rc) return rc;
  }
  return VAM_OK;
}

}  // namespace vam
