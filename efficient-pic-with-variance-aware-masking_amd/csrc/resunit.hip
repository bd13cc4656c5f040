// One launch per ResidualUnit (reference layers/layers.py:30-48):
//
//   out = GELU( x + conv1x1_{C/2->C}( GELU( conv3x3_{C/2->C/2}( GELU( conv1x1_{C->C/2}(x) ) ) ) ) )
//
// The three convolutions of a unit used to be three launches of the implicit-GEMM kernel with both C/2-channel
// intermediates making a round trip through HBM (3840 B per pixel moved for 1536 B of input + output).  Here a
// workgroup owns an 8 x 16 pixel tile of one image and keeps both intermediates in LDS:
//
//   GEMM1  t1 = GELU(W1 x + b1) on the tile plus a one-pixel halo (10 x 18 = 180 positions; positions outside the
//          image are ZERO, the 3x3's padding), written to LDS as bf16x3 planes — the split-operand format the
//          matrix pipe reads (conv_igemm.hip, MODE 1), so the 3x3 stages nothing: its A operand of tap (dy,dx) is the
//          same LDS image read at a row offset;
//   GEMM2  t2 = GELU(W2 * t1 + b2), 9 taps x 3 channel groups, t2 written over t1 as planes;
//   GEMM3  out = GELU(W3 t2 + b3 + x), stored once.
//
// Arithmetic is the split-operand scheme of conv_igemm.hip (every fp32 operand = hi + mid + lo bf16 terms, six exact
// partial products per 32x32x16 block, fp32 accumulate) in the SAME canonical K order (32-channel group outer, tap
// inner, two 16-channel steps per group, the six products smallest first), the same bias / GELU / split formulas, and
// the same packed weights (vam_pack_conv_weights): results are bit-identical to the three-launch path
// (tests/test_gpu_ops.py::test_fused_residual_unit_is_bit_identical).
//
// The MFMA runs "transposed" (A operand = weights, B operand = pixels): the accumulator then holds, per lane, one
// pixel and 16 channels in runs of four — so t1 / t2 leave the accumulators as 8-byte plane stores and the output as
// 16-byte row segments without a transpose through LDS.
//
// Weights stream through a ring of three 18 KB LDS slots ("slab" = 96 output channels x 32 input channels of one tap,
// 192 B per row, the packed layout verbatim) by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round trip, no
// ds_write; the XOR swizzle of the LDS image is applied to the per-lane SOURCE address), two slabs ahead of the
// MFMAs, one raw s_barrier per slab behind a counted s_waitcnt vmcnt.  Only x (fp32 in HBM) is staged through
// registers, because it has to be split.  12 waves per workgroup (3 per SIMD): GEMM2's 12 (channel tile, pixel tile)
// pairs are one per wave; its operand fragments are double-buffered in registers (the LDS reads of slab k+1 are in flight
// under the MFMAs of slab k: 51.5k -> 41.7k cycles for the 27 slabs).  LDS: 180 x 576 B of planes + 3 x 18 KB = 155 KB,
// one workgroup per CU.
//
// Where a tile's 110k cycles go (s_memtime stamps, scratch/ru_phases.py; matrix-pipe time alone would be 53k):
// GEMM1 23.5k (x from HBM, 1.41x halo) | t1 conversion 12.9k | GEMM2 41.7k | t2 conversion 5.7k | GEMM3 10.0k |
// epilogue 16.7k.  The conversion phases are vector-pipe bound (45 instructions per erff GELU, 54k of them per tile,
// + the bf16x3 split) and, with one workgroup per CU, nothing runs beside them.  A persistent variant that deferred the
// epilogue and two thirds of the t1 conversion into GEMM2's idle issue slots (one wave of each SIMD per slab, exact
// compile-time s_waitcnt counts around the LDS-DMA ring) was built, is bit-identical and spill-free, and is NOT faster
// (988 vs 955 us per four units: a lone wave issues vector instructions at half the pipe's rate and the slab waits for
// it; git history, "experiment: persistent fused-ResidualUnit kernel").
#include "common.h"

namespace vam {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct RuP {
  const float* x;      // (bf16-storage problems: bf16 tensors behind these two pointers, strides in bf16 elements)
  float* out;
  const float* w1; const float* b1;
  const float* w2; const float* b2;
  const float* w3; const float* b3;
  int ldx, ldo, B, H, W, tiles_x, tiles_y;
};

struct RuArgs {
  long long* dbg;      // diagnostic builds of the measurement scripts: [block][8] s_memtime stamps at the phase boundaries (NULL = off)
  int nprob;
  int tile_start[VAM_MAX_GROUP + 1];
  RuP p[VAM_MAX_GROUP];
};

__device__ __forceinline__ float ru_gelu(float v) { return vam_gelu(v); }

// exact 3-way split by truncation of four fp32 values into packed bf16 pairs (conv_igemm.hip, same formula)
__device__ __forceinline__ void ru_split4(const float (&v)[4], uint2& h, uint2& m, uint2& l) {
  unsigned hb[4], mb[4], lb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    hb[k] = __float_as_uint(v[k]);
    const float r1 = v[k] - __uint_as_float(hb[k] & 0xFFFF0000u);
    mb[k] = __float_as_uint(r1);
    lb[k] = __float_as_uint(r1 - __uint_as_float(mb[k] & 0xFFFF0000u));
  }
  h = make_uint2(__builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u), __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u));
  m = make_uint2(__builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u), __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u));
  l = make_uint2(__builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u), __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u));
}

#define RU_MFMA6(acc, w, p)                                                                  \
  do {                                                                                       \
    /* (activation plane, weight plane): (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi) */ \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], p[0], acc, 0, 0, 0);                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], p[2], acc, 0, 0, 0);                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], p[1], acc, 0, 0, 0);                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], p[0], acc, 0, 0, 0);                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], p[1], acc, 0, 0, 0);                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], p[0], acc, 0, 0, 0);                 \
  } while (0)

constexpr int RU_C = 192, RU_TH = 8, RU_TW = 16, RU_HW = RU_TW + 2, RU_NHALO = (RU_TH + 2) * (RU_TW + 2);
constexpr int RU_NT = 768;                       // 12 waves
constexpr int RU_TROW = 576;                     // bytes of one t1 / t2 row: three 32-channel groups x 192 B
constexpr int RU_TBYTES = RU_NHALO * RU_TROW;    // 103,680
constexpr int RU_SLOT = 96 * 192;                // one weight slab: 96 rows x 192 B
constexpr int RU_LDS = RU_TBYTES + 3 * RU_SLOT;  // 158,976

// DMA: 1 = weight slabs of GEMM2 / GEMM3 by LDS-DMA through the three-slot ring; 0 = register-staged, two LDS buffers
// (the A/B arm, and what GEMM1 always uses for W1 beside the x staging)
template <int DMA>
__global__ __launch_bounds__(RU_NT, 3) void resunit192_kernel(const RuArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sT = smem;                      // t1 (180 rows) / t2 (128 rows) planes; during GEMM1: x planes [2][192][192 B]
  unsigned char* sR = smem + RU_TBYTES;          // weight ring, 3 slots

  // ---- which problem / tile: every XCD gets a contiguous eighth of each problem's tiles (conv_igemm.hip)
  int pi = -1, t = 0;
  {
    const int xcd = blockIdx.x & 7;
    int idx = blockIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < VAM_MAX_GROUP; ++i) {
      if (i < args.nprob && pi < 0) {
        const int T = args.tile_start[i + 1] - args.tile_start[i];
        const int q = T >> 3, r = T & 7;
        const int c = q + (xcd < r ? 1 : 0);
        if (idx < c) {
          pi = i;
          t = xcd * q + (xcd < r ? xcd : r) + idx;
        } else {
          idx -= c;
        }
      }
    }
  }
  if (pi < 0) return;
  const RuP& P = args.p[pi];
  const int u_H = __builtin_amdgcn_readfirstlane(P.H), u_W = __builtin_amdgcn_readfirstlane(P.W);
  const int u_ldx = __builtin_amdgcn_readfirstlane(P.ldx), u_ldo = __builtin_amdgcn_readfirstlane(P.ldo);
  const int tpi = P.tiles_x * P.tiles_y;
  const int img = t / tpi;
  const int tr = t - img * tpi;
  const int tyi = tr / P.tiles_x;
  const int y0 = tyi * RU_TH, x0 = (tr - tyi * P.tiles_x) * RU_TW;

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  long long* const dbg = args.dbg;
  auto stamp = [&](int i) {
    if (dbg != nullptr && tid == 0) dbg[(size_t)blockIdx.x * 8 + i] = (long long)__builtin_amdgcn_s_memtime();
  };
  stamp(0);

  auto desc = [&](const void* q) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(q);
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                                (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a)), 0, 0x7FFFFFFF, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t r_x = desc(P.x), r_w1 = desc(P.w1), r_w2 = desc(P.w2), r_w3 = desc(P.w3);

  // ---- per-thread staging roles
  // x: halo row (tid >> 2) < 192, 8-channel unit (tid & 3) of the current 32-channel group; rows >= 180 and positions
  // outside the image load zeros (offset 2^31 is out of the descriptor's range)
  const int xrow = tid >> 2, xg = tid & 3;
  unsigned xoff;
  {
    const int hy = xrow / RU_HW, hx = xrow - hy * RU_HW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool in = xrow < RU_NHALO && (unsigned)iy < (unsigned)u_H && (unsigned)ix < (unsigned)u_W;
    xoff = in ? (unsigned)(((img * u_H + iy) * u_W + ix) * u_ldx * 4 + xg * 32) : 0x80000000u;
  }
  const int xst = xrow * 192 + ((xg ^ ((xrow >> 2) & 3)) << 4);          // + plane * 64
  // weight slab, register-staged: 1152 16-byte units, two per thread for tid < 384
  unsigned wgo[2];
  int wlo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = tid + i * RU_NT;
    const int row = u / 12, c = u - row * 12;
    const bool ok = u < 96 * 12;
    wgo[i] = ok ? (unsigned)(u * 16) : 0x80000000u;
    wlo[i] = ok ? row * 192 + (((c & ~3) | ((c & 3) ^ ((row >> 2) & 3))) << 4) : -1;
  }
  // weight slab by LDS-DMA: 18 wave-instructions of 1 KB; this wave issues j0 = 2*wid and 2*wid+1 (mod 18: waves 9..11
  // repeat six of them, same bytes to the same place).  Lane `lane` of instruction j fills the LDS unit j*64 + lane =
  // (row, physical chunk c'), whose content is the slab's logical chunk c = c' with the low two bits XOR (row >> 2) & 3.
  unsigned dgo[2];
  int dlo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int j = wid * 2 + i;
    j = j >= 18 ? j - 18 : j;
    const int u = j * 64 + lane;
    const int row = u / 12, cp = u - row * 12;
    const int c = (cp & ~3) | ((cp & 3) ^ ((row >> 2) & 3));
    dgo[i] = (unsigned)((row * 12 + c) * 16);
    dlo[i] = j * 1024;
  }
  auto dma_slab = [&](const __amdgpu_buffer_rsrc_t& r, unsigned slab_byte, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sR + slot * RU_SLOT + dlo[i]), 16,
                                               (int)dgo[i], (int)slab_byte, 0, 0);
  };

  const int rsw = (l31 >> 2) & 3;                                      // swizzle term of rows (32-row tile base) + l31
  const int fsub[2] = {((lh ^ rsw) << 4), (((2 + lh) ^ rsw) << 4)};    // byte offset of this lane's k-group, steps 0 / 1

  // =============================================================== GEMM1: t1 = GELU(W1 x + b1) on the halo tile
  // 18 (channel tile, pixel tile) pairs over 12 waves: waves 0..5 own pixel tile wid with channel tiles 0 and 2,
  // waves 6..11 pixel tile wid-6 with channel tile 1
  const int px1 = wid < 6 ? wid : wid - 6;
  const bool two1 = wid < 6;
  const int c1a = two1 ? 0 : 1;
  f32x16 acc1[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[q][r] = 0.f;
  {
    u32x4 xr0[2], xr1[2], wr0[2], wr1[2];
    auto gload1 = [&](int k, u32x4 (&xr)[2], u32x4 (&wr)[2]) {
      xr[0] = __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)(xoff + (unsigned)k * 128u), 0, 0);
      xr[1] = __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)(xoff + (unsigned)k * 128u + 16u), 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) wr[i] = __builtin_amdgcn_raw_buffer_load_b128(r_w1, (int)(wgo[i] + (unsigned)k * (unsigned)RU_SLOT), 0, 0);
    };
    auto sstore1 = [&](int buf, const u32x4 (&xr)[2], const u32x4 (&wr)[2]) {
      const float lo4[4] = {__uint_as_float(xr[0].x), __uint_as_float(xr[0].y), __uint_as_float(xr[0].z), __uint_as_float(xr[0].w)};
      const float hi4[4] = {__uint_as_float(xr[1].x), __uint_as_float(xr[1].y), __uint_as_float(xr[1].z), __uint_as_float(xr[1].w)};
      uint2 h0, m0, l0, h1, m1, l1;
      ru_split4(lo4, h0, m0, l0);
      ru_split4(hi4, h1, m1, l1);
      unsigned char* d = sT + buf * (192 * 192) + xst;
      u32x4 v;
      v.x = h0.x; v.y = h0.y; v.z = h1.x; v.w = h1.y;
      *reinterpret_cast<u32x4*>(d) = v;
      v.x = m0.x; v.y = m0.y; v.z = m1.x; v.w = m1.y;
      *reinterpret_cast<u32x4*>(d + 64) = v;
      v.x = l0.x; v.y = l0.y; v.z = l1.x; v.w = l1.y;
      *reinterpret_cast<u32x4*>(d + 128) = v;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (wlo[i] >= 0) *reinterpret_cast<u32x4*>(sR + buf * RU_SLOT + wlo[i]) = wr[i];
    };
    auto compute1 = [&](int buf) {
      const unsigned char* pb = sT + buf * (192 * 192) + (px1 * 32 + l31) * 192;
      const unsigned char* wb = sR + buf * RU_SLOT + l31 * 192;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 p[3], w[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) p[pl] = *reinterpret_cast<const bf16x8*>(pb + pl * 64 + fsub[ks]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const bf16x8*>(wb + c1a * (32 * 192) + pl * 64 + fsub[ks]);
        RU_MFMA6(acc1[0], w, p);
        if (two1) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const bf16x8*>(wb + 2 * (32 * 192) + pl * 64 + fsub[ks]);
          RU_MFMA6(acc1[1], w, p);
        }
      }
    };
    constexpr int K1 = RU_C / 32;   // 6 items
    gload1(0, xr0, wr0);
    gload1(1, xr1, wr1);                             // (in flight while chunk 0 is split and stored)
    sstore1(0, xr0, wr0);
    gload1(2, xr0, wr0);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < K1; k += 2) {
      sstore1(1, xr1, wr1);                          // item k+1
      if (k + 3 < K1) gload1(k + 3, xr1, wr1);
      compute1(0);
      __syncthreads();
      if (k + 2 < K1) sstore1(0, xr0, wr0);          // item k+2
      if (k + 4 < K1) gload1(k + 4, xr0, wr0);
      compute1(1);
      __syncthreads();
    }
  }
  // every wave is past its last read of the GEMM1 buffers
  stamp(1);

  // ---- item machinery of GEMM2 / GEMM3 (weight slabs only)
  // GEMM2 item k = g*9 + tap -> slab ((tap*3 + g) * 96) rows; GEMM3 item k = g*2 + half -> slab (g*192 + half*96) rows
  auto slab2 = [](int k) -> unsigned { const int g = k / 9, tap = k - g * 9; return (unsigned)((tap * 3 + g) * 96) * 192u; };
  auto slab3 = [](int k) -> unsigned { return (unsigned)((k >> 1) * 192 + (k & 1) * 96) * 192u; };

  if constexpr (DMA) {
    dma_slab(r_w2, slab2(0), 0);
    dma_slab(r_w2, slab2(1), 1);
  }

  // ---- t1 out of the accumulators: lane = pixel (halo row px1*32 + l31), registers 4j..4j+3 = channels 8j + 4lh + 0..3
  {
    const int row = px1 * 32 + l31;
    const int hy = row / RU_HW, hx = row - hy * RU_HW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool store = row < RU_NHALO;
    const bool in = (unsigned)iy < (unsigned)u_H && (unsigned)ix < (unsigned)u_W;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 1 && !two1) break;
      const int ct = q == 0 ? c1a : 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bb = *reinterpret_cast<const float4*>(P.b1 + ct * 32 + 8 * j + 4 * lh);
        float v[4] = {acc1[q][4 * j] + bb.x, acc1[q][4 * j + 1] + bb.y, acc1[q][4 * j + 2] + bb.z, acc1[q][4 * j + 3] + bb.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = in ? ru_gelu(v[i]) : 0.f;
        uint2 h, m, l;
        ru_split4(v, h, m, l);
        if (store) {
          unsigned char* d = sT + row * RU_TROW + ct * 192 + ((j ^ ((row >> 2) & 3)) << 4) + lh * 8;
          *reinterpret_cast<uint2*>(d) = h;
          *reinterpret_cast<uint2*>(d + 64) = m;
          *reinterpret_cast<uint2*>(d + 128) = l;
        }
      }
    }
  }

  stamp(2);
  // =============================================================== GEMM2: t2 = GELU(W2 * t1 + b2), 27 items
  const int px2 = wid & 3, ct2 = wid >> 2;           // one (channel tile, pixel tile) pair per wave
  // Which of the 32 pixels of a pixel tile (two image rows x 16) a lane slot holds.  A ds_read_b128 is served in groups of
  // 16 lanes — {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} — and the XOR swizzle of the plane image is conflict-free for
  // 16 rows that are distinct mod 16.  With the raster order (lanes 0-15 = first row) a group straddles both image rows,
  // whose halo rows lie 18 apart: two of its sixteen lanes collide (measured LDS conflict ratio 0.25).  So each GROUP gets
  // one image row: 16 consecutive halo rows for every tap.  t2 rows and the epilogue use the same slot -> pixel map.
  const int py2 = (0xF00F0FF0u >> l31) & 1;          // image row (0 / 1) inside the pixel tile
  const int pxx2 = l31 - (l31 < 4 ? 0 : l31 < 12 ? 4 : l31 < 20 ? 8 : l31 < 28 ? 12 : 16);   // column 0..15
  f32x16 acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  const int hb2 = (px2 * 2 + py2) * RU_HW + pxx2;                        // halo row of this lane's pixel, tap (0,0)
  // identity (x at this lane's pixel, the channels of its two GEMM3 accumulators): requested when GEMM2 is done
  float4 xv[2][4];
  const int oiy = y0 + px2 * 2 + py2, oix = x0 + pxx2;
  const bool opix_ok = oiy < u_H && oix < u_W;
  const size_t opix = (size_t)(img * u_H + (opix_ok ? oiy : 0)) * u_W + (opix_ok ? oix : 0);
  auto load_identity = [&]() {
    const float* xp = P.x + opix * u_ldx;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[q][j] = *reinterpret_cast<const float4*>(xp + (q * 3 + ct2) * 32 + 8 * j + 4 * lh);
  };
  auto compute2 = [&](int k, const unsigned char* slot) {
    const int g = k / 9, tap = k - g * 9;
    const int ty = tap / 3, tx = tap - ty * 3;
    const int hrow = hb2 + ty * RU_HW + tx;
    const int sw = (hrow >> 2) & 3;
    const unsigned char* pb = sT + hrow * RU_TROW + g * 192;
    const unsigned char* wb = slot + (ct2 * 32 + l31) * 192;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 p[3], w[3];
      const int po = ((2 * ks + lh) ^ sw) << 4;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) p[pl] = *reinterpret_cast<const bf16x8*>(pb + pl * 64 + po);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const bf16x8*>(wb + pl * 64 + fsub[ks]);
      RU_MFMA6(acc2, w, p);
    }
  };
  // register-staged arm: the slab of item k sits in ring slot k & 1
  u32x4 wra[2], wrb[2];
  auto gloadw = [&](const __amdgpu_buffer_rsrc_t& r, unsigned slab_byte, u32x4 (&wr)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) wr[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(wgo[i] + slab_byte), 0, 0);
  };
  auto sstorew = [&](int buf, const u32x4 (&wr)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (wlo[i] >= 0) *reinterpret_cast<u32x4*>(sR + buf * RU_SLOT + wlo[i]) = wr[i];
  };
  // operand fragments of one item: [k-step][plane]; double-buffered (FA / FB) so that the LDS reads of item k+1 are in
  // flight while the MFMAs of item k run
  auto read2 = [&](int k, const unsigned char* slot, bf16x8 (&p)[2][3], bf16x8 (&w)[2][3]) {
    const int g = k / 9, tap = k - g * 9;
    const int ty = tap / 3, tx = tap - ty * 3;
    const int hrow = hb2 + ty * RU_HW + tx;
    const int sw = (hrow >> 2) & 3;
    const unsigned char* pb = sT + hrow * RU_TROW + g * 192;
    const unsigned char* wb = slot + (ct2 * 32 + l31) * 192;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int po = ((2 * ks + lh) ^ sw) << 4;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) p[ks][pl] = *reinterpret_cast<const bf16x8*>(pb + pl * 64 + po);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) w[ks][pl] = *reinterpret_cast<const bf16x8*>(wb + pl * 64 + fsub[ks]);
    }
  };
  auto mma = [&](f32x16& acc, bf16x8 (&p)[2][3], bf16x8 (&w)[2][3]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) RU_MFMA6(acc, w[ks], p[ks]);
  };
  bf16x8 pA[2][3], wA[2][3], pB[2][3], wB[2][3];
#define RU_BAR(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
  constexpr int K2 = 27;
  if constexpr (DMA) {
    // "barrier for slab j": my share of slab j has landed (slab j+1's two instructions may be in flight), then everyone's
    // has — and every wave has the fragments of slab j-1 in registers, so slab j+2 may take that slot
    RU_BAR(2);                                         // slab 0 (and t1) visible
    dma_slab(r_w2, slab2(2), 2);
    read2(0, sR, pA, wA);
    int s = 0;                                         // k % 3
#pragma unroll 1
    for (int k = 0; k + 2 < K2; k += 2) {
      const int s1 = s == 2 ? 0 : s + 1, s2 = s1 == 2 ? 0 : s1 + 1;
      RU_BAR(2);                                       // slab k+1
      if (k + 3 < K2) dma_slab(r_w2, slab2(k + 3), s);
      read2(k + 1, sR + s1 * RU_SLOT, pB, wB);
      mma(acc2, pA, wA);                               // item k
      if (k + 3 < K2) RU_BAR(2); else RU_BAR(0);       // slab k+2
      if (k + 4 < K2) dma_slab(r_w2, slab2(k + 4), s1);
      read2(k + 2, sR + s2 * RU_SLOT, pA, wA);
      mma(acc2, pB, wB);                               // item k+1
      s = s2;
    }
    mma(acc2, pA, wA);                                 // item 26
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // all reads of t1 and of the ring are done
    load_identity();                                   // (ahead of the LDS-DMA instructions: the first GEMM3 barrier covers them,
    dma_slab(r_w3, slab3(0), 0);                       //  and their latency runs under the t2 conversion instead of the epilogue)
    dma_slab(r_w3, slab3(1), 1);
  } else {
    gloadw(r_w2, slab2(0), wra);
    sstorew(0, wra);
    gloadw(r_w2, slab2(1), wrb);
    gloadw(r_w2, slab2(2), wra);
    __syncthreads();                                   // t1 and slab 0 visible
#pragma unroll 1
    for (int k = 0; k < K2; k += 2) {
      if (k + 1 < K2) sstorew(1, wrb);
      if (k + 3 < K2) gloadw(r_w2, slab2(k + 3), wrb);
      compute2(k, sR);
      __syncthreads();
      if (k + 1 >= K2) break;
      if (k + 2 < K2) sstorew(0, wra);
      if (k + 4 < K2) gloadw(r_w2, slab2(k + 4), wra);
      compute2(k + 1, sR + RU_SLOT);
      __syncthreads();
    }
    load_identity();
    gloadw(r_w3, slab3(0), wra);
    gloadw(r_w3, slab3(1), wrb);
  }

  stamp(3);
  // ---- t2 out of the accumulators, over t1 (rows 0..127)
  {
    const int row = px2 * 32 + l31;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 bb = *reinterpret_cast<const float4*>(P.b2 + ct2 * 32 + 8 * j + 4 * lh);
      float v[4] = {acc2[4 * j] + bb.x, acc2[4 * j + 1] + bb.y, acc2[4 * j + 2] + bb.z, acc2[4 * j + 3] + bb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = ru_gelu(v[i]);
      uint2 h, m, l;
      ru_split4(v, h, m, l);
      unsigned char* d = sT + row * RU_TROW + ct2 * 192 + ((j ^ rsw) << 4) + lh * 8;
      *reinterpret_cast<uint2*>(d) = h;
      *reinterpret_cast<uint2*>(d + 64) = m;
      *reinterpret_cast<uint2*>(d + 128) = l;
    }
  }

  stamp(4);
  // =============================================================== GEMM3: out = GELU(W3 t2 + b3 + x), 6 items
  // item (g, half): weight rows half*96 .. +96; this wave's pair = (channel tile half*3 + ct2, pixel tile px2)
  f32x16 acc3[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[q][r] = 0.f;
  auto compute3 = [&](int g, f32x16& acc, const unsigned char* slot) {
    const unsigned char* pb = sT + (px2 * 32 + l31) * RU_TROW + g * 192;
    const unsigned char* wb = slot + (ct2 * 32 + l31) * 192;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 p[3], w[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) p[pl] = *reinterpret_cast<const bf16x8*>(pb + pl * 64 + fsub[ks]);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const bf16x8*>(wb + pl * 64 + fsub[ks]);
      RU_MFMA6(acc, w, p);
    }
  };
  auto read3 = [&](int g, const unsigned char* slot, bf16x8 (&p)[2][3], bf16x8 (&w)[2][3]) {
    const unsigned char* pb = sT + (px2 * 32 + l31) * RU_TROW + g * 192;
    const unsigned char* wb = slot + (ct2 * 32 + l31) * 192;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) p[ks][pl] = *reinterpret_cast<const bf16x8*>(pb + pl * 64 + fsub[ks]);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) w[ks][pl] = *reinterpret_cast<const bf16x8*>(wb + pl * 64 + fsub[ks]);
    }
  };
  if constexpr (DMA) {
    // items 0..5 = (g, half) in ring slots 0,1,2,0,1,2; the accumulator of an item is acc3[item & 1]
    RU_BAR(2);                                         // slab 0 (and t2) visible
    dma_slab(r_w3, slab3(2), 2);
    read3(0, sR, pA, wA);
    RU_BAR(2);                                         // slab 1
    dma_slab(r_w3, slab3(3), 0);
    read3(0, sR + RU_SLOT, pB, wB);
    mma(acc3[0], pA, wA);
    RU_BAR(2);                                         // slab 2
    dma_slab(r_w3, slab3(4), 1);
    read3(1, sR + 2 * RU_SLOT, pA, wA);
    mma(acc3[1], pB, wB);
    RU_BAR(2);                                         // slab 3
    dma_slab(r_w3, slab3(5), 2);
    read3(1, sR, pB, wB);
    mma(acc3[0], pA, wA);
    RU_BAR(2);                                         // slab 4
    read3(2, sR + RU_SLOT, pA, wA);
    mma(acc3[1], pB, wB);
    RU_BAR(0);                                         // slab 5
    read3(2, sR + 2 * RU_SLOT, pB, wB);
    mma(acc3[0], pA, wA);
    mma(acc3[1], pB, wB);
  } else {
    sstorew(0, wra);
    gloadw(r_w3, slab3(2), wra);
    __syncthreads();                                   // t2 and slab 0 visible
#pragma unroll
    for (int k = 0; k < 6; k += 2) {
      sstorew(1, wrb);
      if (k + 3 < 6) gloadw(r_w3, slab3(k + 3), wrb);
      compute3(k >> 1, acc3[0], sR);
      __syncthreads();
      if (k + 2 < 6) sstorew(0, wra);
      if (k + 4 < 6) gloadw(r_w3, slab3(k + 4), wra);
      compute3(k >> 1, acc3[1], sR + RU_SLOT);
      __syncthreads();
    }
  }

  stamp(5);
  // ---- epilogue: out = GELU(acc + b3 + x); lane = pixel, 16-byte row segments
  if (opix_ok) {
    float* op = P.out + opix * u_ldo;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ch = (q * 3 + ct2) * 32 + 8 * j + 4 * lh;
        const float4 bb = *reinterpret_cast<const float4*>(P.b3 + ch);
        float4 o;
        o.x = ru_gelu((acc3[q][4 * j] + bb.x) + xv[q][j].x);
        o.y = ru_gelu((acc3[q][4 * j + 1] + bb.y) + xv[q][j].y);
        o.z = ru_gelu((acc3[q][4 * j + 2] + bb.z) + xv[q][j].z);
        o.w = ru_gelu((acc3[q][4 * j + 3] + bb.w) + xv[q][j].w);
        *reinterpret_cast<float4*>(op + ch) = o;
      }
    }
  }
  stamp(6);
}

// ======================================================================================================================
// bf16-storage variant (BASELINE configs[2] "bf16", model.storage = "bf16"; conv_igemm.hip MODE 2): x, t1, t2 and the
// output are bf16 tensors, the weights are rounded to bf16 once at pack time (vam_pack_conv_weights_bf16: 64 B per
// (output channel, 32-channel chunk)), ONE v_mfma_f32_32x32x16_bf16 per 32x32x16 block, fp32 accumulation, the
// intermediates rounded to bf16 (nearest even) exactly where the three-launch form stores them — bit-identical to it.
// A sixth of the matrix work and a third of the LDS bytes of the fp32 kernel: t1 / t2 rows are 192 B (34.5 KB for the
// halo tile), a weight slab is 6 KB, the ring 18 KB -> 54 KB per workgroup, THREE workgroups per CU, so here the GELU /
// rounding phases of one tile do run beside the MFMA and load phases of the others.  6 waves: GEMM1's 18 pairs are
// three per wave (one pixel tile, three channel tiles), GEMM2's 12 and each half of GEMM3's 24 two per wave (one
// channel tile, two pixel tiles).  x needs no split, so its chunks are plain 16-byte copies; the weight slabs come by
// LDS-DMA (one 1 KB instruction per wave and slab).
constexpr int RB_NT = 384;                        // 6 waves
constexpr int RB_TROW = 192;                      // bytes of one t1 / t2 row: 96 channels of bf16 = three 64-byte groups
constexpr int RB_TBYTES = RU_NHALO * RB_TROW;     // 34,560
constexpr int RB_SLOT = 96 * 64;                  // one weight slab: 96 rows x 64 B
constexpr int RB_LDS = RB_TBYTES + 3 * RB_SLOT + (96 + 96 + 192) * 4;   // 54,528: t1/t2 | ring | b1 b2 b3

__device__ __forceinline__ unsigned rb_pk2(float lo, float hi) {         // two fp32 -> two bf16 (nearest even), lo in the low half
  const __bf16 a = (__bf16)lo, b = (__bf16)hi;
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

__global__ __launch_bounds__(RB_NT, 5) void resunit192_bf16_kernel(const RuArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sT = smem;                        // t1 (180 rows) / t2 (128 rows); during GEMM1: x chunks [2][192][64 B]
  unsigned char* sR = smem + RB_TBYTES;            // weight ring, 3 slots
  float* sBias = reinterpret_cast<float*>(smem + RB_TBYTES + 3 * RB_SLOT);

  int pi = -1, t = 0;
  {
    const int xcd = blockIdx.x & 7;
    int idx = blockIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < VAM_MAX_GROUP; ++i) {
      if (i < args.nprob && pi < 0) {
        const int T = args.tile_start[i + 1] - args.tile_start[i];
        const int q = T >> 3, r = T & 7;
        const int c = q + (xcd < r ? 1 : 0);
        if (idx < c) {
          pi = i;
          t = xcd * q + (xcd < r ? xcd : r) + idx;
        } else {
          idx -= c;
        }
      }
    }
  }
  if (pi < 0) return;
  const RuP& P = args.p[pi];
  const int u_H = __builtin_amdgcn_readfirstlane(P.H), u_W = __builtin_amdgcn_readfirstlane(P.W);
  const int u_ldx = __builtin_amdgcn_readfirstlane(P.ldx), u_ldo = __builtin_amdgcn_readfirstlane(P.ldo);
  const int tpi = P.tiles_x * P.tiles_y;
  const int img = t / tpi;
  const int tr = t - img * tpi;
  const int tyi = tr / P.tiles_x;
  const int y0 = tyi * RU_TH, x0 = (tr - tyi * P.tiles_x) * RU_TW;

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  auto desc = [&](const void* q) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(q);
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                                (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a)), 0, 0x7FFFFFFF, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t r_x = desc(P.x), r_w1 = desc(P.w1), r_w2 = desc(P.w2), r_w3 = desc(P.w3);
  if (tid < 96) sBias[tid] = P.b1[tid];
  else if (tid < 192) sBias[tid] = P.b2[tid - 96];
  else sBias[tid] = P.b3[tid - 192];               // 192 threads left = the 192 biases of the last layer

  // ---- GEMM1 staging roles: x units (8 channels = 16 B) tid and tid + 384 of the 768 of a chunk; W1 unit tid of 384
  unsigned xoff[2];
  int xst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = tid + i * RB_NT;
    const int row = u >> 2, g = u & 3;
    const int hy = row / RU_HW, hx = row - hy * RU_HW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool in = row < RU_NHALO && (unsigned)iy < (unsigned)u_H && (unsigned)ix < (unsigned)u_W;
    xoff[i] = in ? (unsigned)(((img * u_H + iy) * u_W + ix) * u_ldx * 2 + g * 16) : 0x80000000u;
    xst[i] = row * 64 + ((g ^ ((row >> 2) & 3)) << 4);
  }
  const int wrow = tid >> 2, wc = tid & 3;
  const unsigned wgo = (unsigned)(tid * 16);
  const int wlo = wrow * 64 + ((wc ^ ((wrow >> 2) & 3)) << 4);
  // LDS-DMA of a slab: instruction `wid` of six; lane -> LDS unit wid*64 + lane = (row, physical chunk c')
  unsigned dgo;
  {
    const int u = wid * 64 + lane;
    const int row = u >> 2, cp = u & 3;
    dgo = (unsigned)((row * 4 + (cp ^ ((row >> 2) & 3))) * 16);
  }
  auto dma_slab = [&](const __amdgpu_buffer_rsrc_t& r, unsigned slab_byte, int slot) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sR + slot * RB_SLOT + wid * 1024), 16, (int)dgo,
                                             (int)slab_byte, 0, 0);
  };
  const int rsw = (l31 >> 2) & 3;
  const int fsub[2] = {((lh ^ rsw) << 4), (((2 + lh) ^ rsw) << 4)};
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

  // =============================================================== GEMM1: wave = pixel tile wid, channel tiles 0..2
  f32x16 acc1[3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[q][r] = 0.f;
  {
    u32x4 xr0[2], xr1[2], wr0, wr1;
    auto gload1 = [&](int k, u32x4 (&xr)[2], u32x4& wr) {
#pragma unroll
      for (int i = 0; i < 2; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)(xoff[i] + (unsigned)k * 64u), 0, 0);
      wr = __builtin_amdgcn_raw_buffer_load_b128(r_w1, (int)(wgo + (unsigned)k * (unsigned)RB_SLOT), 0, 0);
    };
    auto sstore1 = [&](int buf, const u32x4 (&xr)[2], const u32x4& wr) {
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(sT + buf * (192 * 64) + xst[i]) = xr[i];
      *reinterpret_cast<u32x4*>(sR + buf * RB_SLOT + wlo) = wr;
    };
    auto compute1 = [&](int buf) {
      const unsigned char* pb = sT + buf * (192 * 64) + (wid * 32 + l31) * 64;
      const unsigned char* wb = sR + buf * RB_SLOT + l31 * 64;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 p = *reinterpret_cast<const bf16x8*>(pb + fsub[ks]);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const bf16x8 w = *reinterpret_cast<const bf16x8*>(wb + q * (32 * 64) + fsub[ks]);
          acc1[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, p, acc1[q], 0, 0, 0);
        }
      }
    };
    constexpr int K1 = RU_C / 32;
    gload1(0, xr0, wr0);
    gload1(1, xr1, wr1);
    sstore1(0, xr0, wr0);
    gload1(2, xr0, wr0);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < K1; k += 2) {
      sstore1(1, xr1, wr1);
      if (k + 3 < K1) gload1(k + 3, xr1, wr1);
      compute1(0);
      __syncthreads();
      if (k + 2 < K1) sstore1(0, xr0, wr0);
      if (k + 4 < K1) gload1(k + 4, xr0, wr0);
      compute1(1);
      __syncthreads();
    }
  }
  auto slab2 = [](int k) -> unsigned { const int g = k / 9, tap = k - g * 9; return (unsigned)((tap * 3 + g) * 96) * 64u; };
  auto slab3 = [](int k) -> unsigned { return (unsigned)((k >> 1) * 192 + (k & 1) * 96) * 64u; };
  dma_slab(r_w2, slab2(0), 0);
  dma_slab(r_w2, slab2(1), 1);
  // ---- t1: lane = halo row wid*32 + l31, registers 4j..4j+3 of channel tile q = channels q*32 + 8j + 4lh + 0..3
  {
    const int row = wid * 32 + l31;
    const int hy = row / RU_HW, hx = row - hy * RU_HW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool in = (unsigned)iy < (unsigned)u_H && (unsigned)ix < (unsigned)u_W;
    if (row < RU_NHALO) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 bb = *reinterpret_cast<const float4*>(sBias + q * 32 + 8 * j + 4 * lh);
          float v[4] = {acc1[q][4 * j] + bb.x, acc1[q][4 * j + 1] + bb.y, acc1[q][4 * j + 2] + bb.z, acc1[q][4 * j + 3] + bb.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = in ? ru_gelu(v[i]) : 0.f;
          *reinterpret_cast<uint2*>(sT + row * RB_TROW + q * 64 + ((j ^ ((row >> 2) & 3)) << 4) + lh * 8) =
              make_uint2(rb_pk2(v[0], v[1]), rb_pk2(v[2], v[3]));
        }
    }
  }
  // =============================================================== GEMM2 / GEMM3: wave = channel tile wid % 3, pixel tiles 2 (wid / 3) + {0, 1}
  const int ctw = wid % 3, pxp = (wid / 3) * 2;
  const int py2 = (0xF00F0FF0u >> l31) & 1;        // slot -> pixel map of a pixel tile: one image row per ds_read_b128 lane group
  const int pxx2 = l31 - (l31 < 4 ? 0 : l31 < 12 ? 4 : l31 < 20 ? 8 : l31 < 28 ? 12 : 16);
  f32x16 acc2[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[q][r] = 0.f;
#define RB_BAR(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
  {
    int hb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) hb[q] = ((pxp + q) * 2 + py2) * RU_HW + pxx2;
    int slot = 0;
#pragma unroll 1
    for (int k = 0; k < 27; ++k) {
      if (k + 1 < 27) RB_BAR(1); else RB_BAR(0);   // slab k landed everywhere (slab k+1 may be in flight); slab k-1's readers are done
      if (k + 2 < 27) dma_slab(r_w2, slab2(k + 2), slot >= 1 ? slot - 1 : slot + 2);
      const int g = k / 9, tap = k - g * 9;
      const int toff = (tap / 3) * RU_HW + (tap % 3);
      const unsigned char* wb = sR + slot * RB_SLOT + (ctw * 32 + l31) * 64;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wb + fsub[ks]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int hrow = hb[q] + toff;
          const bf16x8 p = *reinterpret_cast<const bf16x8*>(sT + hrow * RB_TROW + g * 64 + (((2 * ks + lh) ^ ((hrow >> 2) & 3)) << 4));
          acc2[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, p, acc2[q], 0, 0, 0);
        }
      }
      slot = slot == 2 ? 0 : slot + 1;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // all reads of t1 and of the ring are done
  dma_slab(r_w3, slab3(0), 0);
  dma_slab(r_w3, slab3(1), 1);
  // ---- t2 over t1 (rows = pixel-tile slots)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (pxp + q) * 32 + l31;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 bb = *reinterpret_cast<const float4*>(sBias + 96 + ctw * 32 + 8 * j + 4 * lh);
      const float v0 = ru_gelu(acc2[q][4 * j] + bb.x), v1 = ru_gelu(acc2[q][4 * j + 1] + bb.y), v2 = ru_gelu(acc2[q][4 * j + 2] + bb.z),
                  v3 = ru_gelu(acc2[q][4 * j + 3] + bb.w);
      *reinterpret_cast<uint2*>(sT + row * RB_TROW + ctw * 64 + ((j ^ rsw) << 4) + lh * 8) = make_uint2(rb_pk2(v0, v1), rb_pk2(v2, v3));
    }
  }
  // =============================================================== GEMM3: items (g, half); accumulators [half][pixel tile]
  f32x16 acc3[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[a][q][r] = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (k + 1 < 6) RB_BAR(1); else RB_BAR(0);
    if (k + 2 < 6) dma_slab(r_w3, slab3(k + 2), (k + 2) % 3);
    const unsigned char* wb = sR + (k % 3) * RB_SLOT + (ctw * 32 + l31) * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(wb + fsub[ks]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const bf16x8 p = *reinterpret_cast<const bf16x8*>(sT + ((pxp + q) * 32 + l31) * RB_TROW + (k >> 1) * 64 + fsub[ks]);
        acc3[k & 1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, p, acc3[k & 1][q], 0, 0, 0);
      }
    }
  }
#undef RB_BAR
  // ---- epilogue: out = bf16( GELU((acc + b3) + x) ); lane = pixel, four channels = 8 bytes
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int iy = y0 + (pxp + q) * 2 + py2, ix = x0 + pxx2;
    if (iy < u_H && ix < u_W) {
      const size_t pix = (size_t)(img * u_H + iy) * u_W + ix;
      const unsigned short* xp = reinterpret_cast<const unsigned short*>(P.x) + pix * u_ldx;
      unsigned short* op = reinterpret_cast<unsigned short*>(P.out) + pix * u_ldo;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ch = (a * 3 + ctw) * 32 + 8 * j + 4 * lh;
          const uint2 xu = *reinterpret_cast<const uint2*>(xp + ch);
          const float4 bb = *reinterpret_cast<const float4*>(sBias + 192 + ch);
          const float x0f = __uint_as_float(xu.x << 16), x1f = __uint_as_float(xu.x & 0xFFFF0000u), x2f = __uint_as_float(xu.y << 16),
                      x3f = __uint_as_float(xu.y & 0xFFFF0000u);
          const float o0 = ru_gelu((acc3[a][q][4 * j] + bb.x) + x0f), o1 = ru_gelu((acc3[a][q][4 * j + 1] + bb.y) + x1f),
                      o2 = ru_gelu((acc3[a][q][4 * j + 2] + bb.z) + x2f), o3 = ru_gelu((acc3[a][q][4 * j + 3] + bb.w) + x3f);
          *reinterpret_cast<uint2*>(op + ch) = make_uint2(rb_pk2(o0, o1), rb_pk2(o2, o3));
        }
    }
  }
}

static long long* g_ru_dbg = nullptr;
static int g_ru_dma = -1;   // 1 = weight slabs by LDS-DMA through the three-slot ring (default), 0 = register-staged, two LDS
                            // buffers (VAMPIC_RU_DMA=0: the A/B arm); -1 = not chosen yet

}  // namespace vam

using namespace vam;

extern "C" {

int vam_resunit_set_dma(int mode) {
  g_ru_dma = mode < 0 ? -1 : (mode ? 1 : 0);
  return VAM_OK;
}

/* measurement scripts only (scratch/ru_phases.py): device buffer of 8 stamps per workgroup, NULL = off */
int vam_resunit_set_debug(void* buf) {
  g_ru_dbg = reinterpret_cast<long long*>(buf);
  return VAM_OK;
}

int vam_resunit_supported(int C, int H, int W) {
  (void)H; (void)W;
  return C == RU_C ? 1 : 0;
}

size_t vam_resunit_struct_size(void) { return sizeof(vam_resunit); }

int vam_resunit_group(const vam_resunit* probs, int nprob, void* stream) {
  VAM_REQUIRE(probs && nprob >= 1 && nprob <= VAM_MAX_GROUP, "vam_resunit_group: 1..%d problems", VAM_MAX_GROUP);
  const bool bf16 = (probs[0].flags & VAM_RESUNIT_BF16) != 0;
  VAM_REQUIRE(bf16 || vam_conv_get_mode() == 1, "vam_resunit_group: the fp32 fused residual unit is built for the split-operand mode");
  RuArgs ga;
  ga.dbg = g_ru_dbg;
  ga.nprob = nprob;
  int total = 0;
  double flops = 0, bytes = 0;
  for (int i = 0; i < nprob; ++i) {
    const vam_resunit& c = probs[i];
    RuP& p = ga.p[i];
    VAM_REQUIRE(c.C == RU_C, "resunit[%d]: C = %d (this build fuses C = %d units)", i, c.C, RU_C);
    VAM_REQUIRE(((c.flags & VAM_RESUNIT_BF16) != 0) == bf16 && (c.flags & ~VAM_RESUNIT_BF16) == 0, "resunit[%d]: a group is all fp32 or all bf16 storage (flags 0x%x)", i, c.flags);
    VAM_REQUIRE(c.B > 0 && c.H > 0 && c.W > 0, "resunit[%d]: bad extent", i);
    VAM_REQUIRE(c.x && c.out && c.w1 && c.b1 && c.w2 && c.b2 && c.w3 && c.b3, "resunit[%d]: null pointer", i);
    VAM_REQUIRE(c.x != c.out, "resunit[%d]: in-place is not supported (neighbouring tiles read the halo)", i);
    VAM_REQUIRE(c.ldx >= c.C && c.ldo >= c.C && c.ldx % (bf16 ? 8 : 4) == 0 && c.ldo % (bf16 ? 8 : 4) == 0, "resunit[%d]: pixel strides", i);
    auto al = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    VAM_REQUIRE(al(c.x) && al(c.out) && al(c.w1) && al(c.w2) && al(c.w3) && al(c.b1) && al(c.b2) && al(c.b3), "resunit[%d]: pointers must be 16-byte aligned", i);
    VAM_REQUIRE((double)c.B * c.H * c.W * c.ldx * (bf16 ? 2.0 : 4.0) < 2147483648.0, "resunit[%d]: input spans 2 GiB or more (split the batch)", i);
    p.x = c.x; p.out = c.out; p.w1 = c.w1; p.b1 = c.b1; p.w2 = c.w2; p.b2 = c.b2; p.w3 = c.w3; p.b3 = c.b3;
    p.ldx = c.ldx; p.ldo = c.ldo; p.B = c.B; p.H = c.H; p.W = c.W;
    p.tiles_x = (c.W + RU_TW - 1) / RU_TW;
    p.tiles_y = (c.H + RU_TH - 1) / RU_TH;
    ga.tile_start[i] = total;
    total += c.B * p.tiles_x * p.tiles_y;
    const double px = (double)c.B * c.H * c.W;
    flops += 2.0 * px * (double)(c.C * (c.C / 2) * 2 + (c.C / 2) * (c.C / 2) * 9);
    bytes += (bf16 ? 2.0 : 4.0) * px * 2.0 * c.C;
  }
  for (int i = nprob; i <= VAM_MAX_GROUP; ++i) ga.tile_start[i] = total;
  if (g_ru_dma < 0) {
    const char* e = getenv("VAMPIC_RU_DMA");
    g_ru_dma = (e && e[0] == '0') ? 0 : 1;
  }
  int per_xcd = 0;
  for (int i = 0; i < nprob; ++i) per_xcd += (ga.tile_start[i + 1] - ga.tile_start[i] + 7) / 8;
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set[2] = {false, false};
  ProfScope ps(VAM_FAM_CONV, s, flops, bytes);
  if (bf16) {
    static bool attr16 = false;
    if (!attr16) {
      (void)hipFuncSetAttribute((const void*)resunit192_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RB_LDS);
      attr16 = true;
    }
    hipLaunchKernelGGL(resunit192_bf16_kernel, dim3(8 * per_xcd), dim3(RB_NT), RB_LDS, s, ga);
    return check_launch("resunit192_bf16_kernel");
  }
  if (g_ru_dma) {
    if (!attr_set[1]) {
      (void)hipFuncSetAttribute((const void*)resunit192_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, RU_LDS);
      attr_set[1] = true;
    }
    hipLaunchKernelGGL((resunit192_kernel<1>), dim3(8 * per_xcd), dim3(RU_NT), RU_LDS, s, ga);
  } else {
    if (!attr_set[0]) {
      (void)hipFuncSetAttribute((const void*)resunit192_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, RU_LDS);
      attr_set[0] = true;
    }
    hipLaunchKernelGGL((resunit192_kernel<0>), dim3(8 * per_xcd), dim3(RU_NT), RU_LDS, s, ga);
  }
  return check_launch("resunit192_kernel");
}

}  // extern "C"
