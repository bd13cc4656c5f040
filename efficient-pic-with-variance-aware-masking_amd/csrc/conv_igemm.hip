// NHWC im2col-free convolution as an implicit GEMM on the CDNA4 matrix cores.
//
//   out[p, n] = post2 + post + mul * act( bias[n] + pre + sum_{ty,tx,c} in[pix(p)+(ty,tx), c] * W[ty,tx,c,n] )
//
// Two arithmetic modes, one kernel template (MODE):
//   MODE 1 (default): every fp32 operand is split EXACTLY into three bf16 terms (hi + mid + lo) and the product is
//           formed from six exact bf16 x bf16 partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//           fp32 accuracy (error vs float64 no larger than the fp32 fma chain's) at 6/16 of the fp32 pipe's matrix time.
//           Weights are split once at pack time ([tap][32-channel chunk][n][plane 3][group 4][8 bf16], 192 B per row,
//           copied to LDS verbatim); fp32 activations are split while they are staged into LDS, or arrive already
//           split as bf16x3 planes from the producing launch's epilogue (AIN = 1).  K step 32 channels; LDS rows of
//           12 x 16 B, the group index XOR-swizzled with the row so that staging writes and operand reads are both
//           bank-conflict free; tiles with BM + BN > 128 single-buffer LDS, smaller ones double-buffer.
//   MODE 0 (VAMPIC_CONV=f32): fp32 operands on v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain, 64 FLOP/clk/SIMD),
//           weights packed [tap][16-channel chunk][n][16] fp32, K step 16 or 32, double-buffered XOR-swizzled LDS.
// Common to both:
// * M = output positions p=(b,oy,ox), N = output channels, K = taps x concatenated input channels.  The input is a
//   virtual channel-concat of up to 4 NHWC "segments" (pointer + channel count + pixel stride), so torch.cat of the
//   reference (models/pic.py:528-529,548,598-599,635) never materialises.
// * Each K chunk (one tap, BK consecutive channels) of the A tile is BM rows of BK contiguous channels in HBM/L2 ->
//   coalesced 16-byte buffer loads, zero-filled at the halo by the hardware range check, staged through LDS; the next
//   chunk is prefetched into registers under the MFMAs.
// * Canonical K order (32-channel group outer, tap inner): a layer gives bit-identical results however it is tiled
//   or grouped (the decoder must reproduce the encoder's sigma exactly).  16-channel inputs (the space-to-depth RGB
//   layer) put two consecutive taps into one 32-channel chunk instead of padding each tap with 16 zero channels.
// * Grouped launch: up to 8 independent problems share one grid; every XCD gets a contiguous eighth of EACH problem's
//   tiles (blocks b and b+8 share an XCD/L2) so neighbours reuse A rows and halos.
// * The C tile is staged through LDS in 32-row slabs so the epilogue issues 16-byte loads / stores.
//
// Replaces nn.Conv2d / nn.ConvTranspose2d (per sub-pixel phase) / nn.Linear plus the
// surrounding element-wise ops of reference layers/layers.py:5-86, layers/gdn.py:62-75,
// layers/rem.py:52-66,130-141, models/pic.py:528-551,598-641.
#include "common.h"
#include <cstdlib>

namespace vam {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct ConvP {
  const float* seg_ptr[VAM_MAX_SEG];
  int seg_ld[VAM_MAX_SEG];
  int seg_end[VAM_MAX_SEG];  // cumulative channel end of each segment
  int n_seg;
  int H, W, HW;
  int kh, kw, stride, pad_y, pad_x;
  int Ho, Wo, HoWo;
  int P;        // B*Ho*Wo
  int N, Npad;  // output channels, padded to 32
  int Cin, Kc, Kc16;  // total input channels, K chunks (of the kernel's BK) per tap, 16-channel packing chunks per tap
  const float* wpack;
  const float* bias;
  float* out;
  int ldo, Hf, Wf, osy, osx, ooy, oox, Cq, act, flags;
  const float* pre;
  const float* mul;
  const float* post;
  const float* post2;
  int ld_pre, ld_mul, ld_post, ld_post2;
  float* preact;        // optional second output: the value the activation is applied to (bias + pre + conv), fp32, indexed like the output
  int ld_preact;
  int tiles_n;
  const int* in_amax[VAM_MAX_SEG];   // MODE 3: per input segment, a device cell holding the bits of an upper bound of max |x| (nullptr: unused)
  int* out_amax;        // any mode: if set, the epilogue folds max |stored value| into this cell (integer atomicMax on the float's bits)
};

constexpr int VAM_CONVI_DUAL = 1 << 29;     // internal ConvP flag: 16-channel input, two taps share one 32-channel K chunk (split-operand mode)
constexpr int VAM_CONVI_STAGED = 1 << 30;   // internal ConvP flag: tensor extents beyond the direct epilogue's 32-bit window

struct GroupArgs {
  int nprob;
  int staged_epilogue;   // 1: every problem takes the LDS-staged epilogue (VAMPIC_EPILOGUE=staged: A/B measurements, bit-identity tests)
  int tile_start[VAM_MAX_GROUP + 1];
  ConvP p[VAM_MAX_GROUP];
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case VAM_ACT_GELU: return vam_gelu(v);
    case VAM_ACT_LEAKY: return v > 0.f ? v : v * 0.01f;
    case VAM_ACT_HALF_TANH: return 0.5f * tanhf(v);
    case VAM_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case VAM_ACT_CLAMP01: return fminf(fmaxf(v, 0.f), 1.f);
    case VAM_ACT_RSQRT: return 1.0f / sqrtf(v);
    case VAM_ACT_SQRT: return sqrtf(v);
    case VAM_ACT_DOUBLE: return 2.0f * v;
    default: return v;
  }
}

constexpr int PK = 16;  // packing granularity of the weight buffer along K

// MODE 0: fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
// MODE 1: every fp32 operand is split EXACTLY into three bf16 terms (x = hi + mid + lo, 8 + 8 + 8 mantissa bits) and
//         the product is formed from the six bf16 x bf16 partial products of weight <= 2 (each exact in fp32) on
//         v_mfma_f32_32x32x16_bf16, accumulating in fp32.  Dropped terms are below 2^-24 of the product — the size of
//         ONE fp32 rounding; measured error vs float64 is slightly smaller than the fp32 fma chain's
//         (scratch/probe/bf16x3.hip: rms 1.06e-6 vs 1.22e-6 at K = 4800).  The bf16 pipe runs 16x the fp32 MFMA rate,
//         so six products cost 6/16 of the fp32 matrix time.  Weights are split once at pack time, activations when
//         they are staged into LDS.
// AIN (MODE 1 only): 0 = fp32 NHWC input segments, split into bf16x3 while staging; 1 = input already stored as bf16x3
// planes ("P3": [pixel][8-channel group][plane 3][8 bf16], 48 B per group) by the producing launch's epilogue
// (VAM_CONV_OUT_BF3) — the staging is then a pure copy, like the weights'.  Inside conv stacks every intermediate is
// consumed by exactly one convolution, which otherwise re-splits each element once per tap and per N tile.
// SPEC (MODE 1 only): wave-specialised block of 2 x WGM x WGN waves.  Waves 0..NW-1 ("consumers") do nothing but operand
// reads and MFMAs; waves NW..2NW-1 ("loaders") do nothing but global loads, the bf16x3 split and the LDS stores, two
// K chunks ahead, into a double-buffered LDS tile.  A workgroup's waves are dealt to the 4 SIMDs cyclically, so every
// SIMD hosts one consumer and one loader: the matrix pipe and the vector / memory pipes run side by side instead of
// taking turns inside each wave (measured on the 128x192 tile: MFMA-only time 1.42 ms, staging-only 1.03 ms, the
// un-specialised kernel 2.45 ms = their SUM).  One s_barrier per K chunk.
// (Tried on top and measured no better, so not kept: 8 consumer waves (4x2) beside 4 loaders on the 128x192 / 128x128
// tiles — two consumers per SIMD to cover each other's operand-read latency: 2350 vs 2287 us on the 192->192 5x5 layer.)
template <int BM, int BN, int BK, int WGM, int WGN, int MODE, int AIN = 0, int SPEC = 0>
__global__ __launch_bounds__(WGM * WGN * 64 * (SPEC ? 2 : 1), SPEC ? ((BM + BN <= 192) ? 4 : 2) : (((MODE == 1 || MODE == 3) && BM == 128 && BN == 128) ? 3 : 2)) void conv_igemm_kernel(const GroupArgs args) {
  static_assert(SPEC == 0 || MODE == 1 || MODE == 3, "wave specialisation is built for the split-operand modes");
  static_assert(MODE != 3 || AIN == 0, "MODE 3 (fp16x2) takes fp32 inputs");
  static_assert(MODE != 2 || AIN <= 1, "MODE 2: AIN 0 = fp32 input rounded while staged, 1 = bf16 input");
  constexpr int NT = WGM * WGN * 64;         // threads of one role group (= threads per block without SPEC)
  constexpr int NTC = NT;
  constexpr int NTB = NT * (SPEC ? 2 : 1);   // threads per block
  constexpr int LDS_LD = BK;                 // floats per LDS row: no padding, the 16-byte chunks of a row are
                                             // XOR-swizzled instead (below) so that both the staging writes and
                                             // the MFMA operand reads are bank-conflict free
  constexpr int TM = BM / WGM / 32;          // 32x32 tiles per wave along M
  constexpr int TN = BN / WGN / 32;
  constexpr int CPR = MODE ? 4 : BK / 4;     // per-thread load units per row (MODE 0: float4; MODE 1 / 2: 8 channels)
  constexpr int LDW = MODE ? 8 : 4;          // floats per load unit
  static_assert(MODE == 0 || BK == 32, "the bf16 paths step K by 32 channels");
  // chunk c of row r lives at chunk c ^ ((r >> SW_SHIFT) & (CPR-1)): 16 consecutive lanes of a ds_write_b128
  // (16/CPR whole rows) and of a ds_read_b128 (16 rows, one logical chunk) each cover all 64 banks once
  constexpr int SW_SHIFT = (CPR == 4) ? 2 : 1;
  constexpr int RPP = NT / CPR;              // rows covered per pass of the block
  constexpr int NA = (BM + RPP - 1) / RPP;   // A float4 per thread
  constexpr int NB = (BN + RPP - 1) / RPP;
  constexpr bool A_FULL = (BM % RPP) == 0, B_FULL = (BN % RPP) == 0;
  static_assert(TM >= 1 && TN >= 1 && BM == TM * WGM * 32 && BN == TN * WGN * 32, "tile/wave layout");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                          // [2][BM][LDS_LD]
  float* sB = smem + 2 * BM * LDS_LD;        // [2][BN][LDS_LD]

  // ---- which problem / tile.  The hardware deals consecutive block ids round-robin over the 8
  // XCDs (blocks b and b+8 share an L2).  Every XCD gets a contiguous 1/8 of EACH problem's tiles:
  // neighbouring tiles (shared A rows / halos) stay on one L2, and problems of unequal cost (the
  // 9/6/6/4-tap phases of a transposed conv) are balanced across XCDs.  The grid is padded to
  // 8 x max-per-XCD; surplus blocks exit.
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
  const ConvP& P = args.p[pi];
  // Block-uniform problem fields used by the main loop, pinned to SGPRs.  Without this hipcc re-loads some of
  // them inside the (lane-divergent) row-decode branch below, the value comes out of that branch in a VGPR, and
  // everything derived from it — tap count, chunk count, the chunk state, the loop condition — turns into
  // vector code with an exec-masked loop (seen in the ISA: v_cmp loop exit, v_cndmask chunk state).
  const int u_kh = __builtin_amdgcn_readfirstlane(P.kh), u_kw = __builtin_amdgcn_readfirstlane(P.kw);
  const int u_W = __builtin_amdgcn_readfirstlane(P.W), u_Kc = __builtin_amdgcn_readfirstlane(P.Kc);
  const int u_kc16 = __builtin_amdgcn_readfirstlane(P.Kc16), u_Npad = __builtin_amdgcn_readfirstlane(P.Npad);
  const int u_se0 = __builtin_amdgcn_readfirstlane(P.seg_end[0]), u_se1 = __builtin_amdgcn_readfirstlane(P.seg_end[1]);
  const int u_se2 = __builtin_amdgcn_readfirstlane(P.seg_end[2]);
  const int tn = t % P.tiles_n, tm = t / P.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // SPEC: tid indexes a thread inside its role group (consumers 0..NT-1, loaders 0..NT-1)
  const bool is_loader = SPEC && (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) >= WGM * WGN);
  const int tid = (SPEC && (int)threadIdx.x >= NTC) ? (int)threadIdx.x - NTC : (int)threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  const int ld_row = tid / CPR;              // row within a pass
  const int ld_col = (tid % CPR) * LDW;      // float offset within the chunk

  // ---- per-thread A rows: decode the output position once.  a_pix0 = input pixel index of tap
  // (0,0); a_mask bit t = tap t of this row lies inside the image (zero padding otherwise).
  int a_pix0[NA];
  unsigned a_mask[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int p = m0 + ld_row + i * RPP;
    a_pix0[i] = 0;
    a_mask[i] = 0u;
    if (p < P.P && (A_FULL || ld_row + i * RPP < BM)) {
      int b = p / P.HoWo;
      int r = p - b * P.HoWo;
      int oy = r / P.Wo;
      int ox = r - oy * P.Wo;
      const int iy0 = oy * P.stride - P.pad_y, ix0 = ox * P.stride - P.pad_x;
      a_pix0[i] = b * P.HW + iy0 * P.W + ix0;
      unsigned m = 0u;
      for (int ty = 0; ty < u_kh; ++ty)
        for (int tx = 0; tx < u_kw; ++tx)
          if ((unsigned)(iy0 + ty) < (unsigned)P.H && (unsigned)(ix0 + tx) < (unsigned)P.W) m |= 1u << (ty * u_kw + tx);
      a_mask[i] = m;
    }
  }
  f32x16 acc[TM][TN];
  if constexpr (!SPEC) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  const int l31 = lane & 31, lh = lane >> 5;
  const unsigned long long wpa = reinterpret_cast<unsigned long long>(P.wpack);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(wpa >> 32)) << 32) |
                              (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)wpa)), 0, 0x7FFFFFFF, 0x00020000);
  const bool sq = (P.flags & VAM_CONV_SQUARE_IN) != 0;

  // chunk state (block-uniform)
  int c_ty = 0, c_tx = 0, c_kc = 0, c_tap = 0;
  int s_begin = 0, s_end = 0, s_ld4 = 0;     // current input segment (see gload)
  unsigned s_lo = 0, s_hi = 0;
  const int n_taps = u_kh * u_kw;
  // 16-channel inputs (the space-to-depth RGB layer): a 32-channel K chunk would be half zeros, so two consecutive taps
  // share one (units 0-1 of a row: tap 2c, units 2-3: tap 2c+1); the weights are packed the same way
  const bool dual = (MODE == 1 || MODE == 3) && (P.flags & VAM_CONVI_DUAL) != 0;
  const int n_chunks = dual ? (n_taps + 1) / 2 : n_taps * u_Kc;        // Kc = K chunks of BK per tap
  // fp32 NHWC outputs with fp32 epilogue operands leave straight from the accumulators (epilogue, "direct" path) in the
  // one configuration where that measured faster (see there)
  constexpr bool DIRECT_CFG = !SPEC && BM == 128 && BN == 64;
  const bool direct_out = DIRECT_CFG && !args.staged_epilogue &&
                          (P.flags & (VAM_CONV_OUT_NCHW | VAM_CONV_OUT_BF3 | VAM_CONV_OUT_BF16 | VAM_CONV_AUX_BF16 | VAM_CONVI_STAGED)) == 0;
  const int kc16 = u_kc16;                   // 16-channel packing chunks per tap
  if constexpr (MODE == 0) {
    // B rows: byte offset inside one [16-chunk][Npad][16] slab of the packed weights (constant per thread)
    unsigned b_off[NB];
  #pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = ld_row + i * RPP;
      const bool ok = (B_FULL || r < BN) && (n0 + r < P.Npad);
      b_off[i] = ok ? (unsigned)((((ld_col >> 4) * u_Npad + n0 + r) * PK + (ld_col & 15)) * 4) : 0x80000000u;
    }
    // Two register stages: the loads of chunk k+2 are issued before chunk k is computed and are
    // written to LDS one iteration later, so a load has two compute phases to land (L2-miss
    // latency is ~2 us under load; one K chunk of MFMAs is 0.2-1.3 us).
    float4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];


    // Global -> register stage through buffer loads: 32-bit per-lane byte offsets against a
    // wave-uniform descriptor, and the halo/tail zero-fill comes from the hardware range check
    // (an offset of 2^31 is out of range and returns 0) instead of branches.
    auto gload = [&](float4 (&ra)[NA], float4 (&rb)[NB]) {
      const int cc0 = c_kc * BK;
      // Input segment holding channel cc0.  Its descriptor is loop-carried SGPR state, reloaded (scalar loads) only
      // when cc0 leaves [s_begin, s_end): a handful of times per tile, instead of two dependent scalar loads and
      // their s_waitcnt between every pair of MFMA phases.
      if (cc0 < s_begin || cc0 >= s_end) {
        const int k = (cc0 >= u_se0 ? 1 : 0) + (cc0 >= u_se1 ? 1 : 0) + (cc0 >= u_se2 ? 1 : 0);
        s_begin = __builtin_amdgcn_readfirstlane(k ? P.seg_end[k - 1] : 0);
        s_end = __builtin_amdgcn_readfirstlane(P.seg_end[k]);
        const unsigned long long spa = reinterpret_cast<unsigned long long>(P.seg_ptr[k]);
        s_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)spa);          // the builtin returns a signed int:
        s_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(spa >> 32));  // keep the halves unsigned
        s_ld4 = __builtin_amdgcn_readfirstlane(P.seg_ld[k]) * 4;
      }
      const int seg_begin = s_begin;
      const int sld4 = s_ld4;
      const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(((unsigned long long)s_hi << 32) | s_lo), 0, 0x7FFFFFFF, 0x00020000);
      const int tap_pix = c_ty * u_W + c_tx;                         // uniform
      const int col4 = (cc0 - seg_begin + ld_col) * 4;
  #pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool ok = (a_mask[i] >> c_tap) & 1u;
        const unsigned off = (unsigned)((a_pix0[i] + tap_pix) * sld4 + col4);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)(ok ? off : 0x80000000u), 0, 0);
        float4 f = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        ra[i] = sq ? make_float4(f.x * f.x, f.y * f.y, f.z * f.z, f.w * f.w) : f;   // GDN pools x^2 (select, no branch)
      }
      // weights: [tap][kc16][Npad][16]; a BK=32 chunk is two consecutive 16-chunks
      const unsigned wbase = (unsigned)((c_tap * kc16 + c_kc * (BK / PK)) * u_Npad * (PK * 4));   // uniform
  #pragma unroll
      for (int i = 0; i < NB; ++i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)(b_off[i] + wbase), 0, 0);
        rb[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
      // advance chunk state (branch-free: the steady-state loop body must stay one basic block so
      // the steady state stays straight-line code)
      // K order (canonical, identical for every tile shape and for BK = 16 or 32, so a layer gives
      // bit-identical results however it is tiled or grouped — the decoder must reproduce the
      // encoder's sigma exactly):  32-channel group OUTER, tap MIDDLE, 16-channel half INNER.
      // All kh*kw taps of one group touch the same input pixels (128 B of each), so a block's
      // working set between re-touches is ~100 KB instead of the whole halo x all channels and the
      // re-reads hit the XCD's L2 instead of going back out to the fabric.
      if (BK == 16) {
        const int has_half = (((c_kc & 1) == 0) && (c_kc + 1 < kc16)) ? 1 : 0;   // second half of this group exists
        const int adv_tap = has_half ? 0 : 1;
        c_kc += has_half;                      // (g, tap, 0) -> (g, tap, 1)
        c_tap += adv_tap;
        c_tx += adv_tap;
        const int wx = (c_tx == u_kw) ? 1 : 0;
        c_tx = wx ? 0 : c_tx;
        c_ty += wx;
        const int wt = (c_tap == n_taps) ? 1 : 0;
        c_tap = wt ? 0 : c_tap;
        c_ty = wt ? 0 : c_ty;
        // next tap of the same group restarts at the group's first half; next group starts after it
        c_kc = adv_tap ? (wt ? (c_kc | 1) + 1 : (c_kc & ~1)) : c_kc;
      } else {
        ++c_tap;
        ++c_tx;
        const int wx = (c_tx == u_kw) ? 1 : 0;
        c_tx = wx ? 0 : c_tx;
        c_ty += wx;
        const int wt = (c_tap == n_taps) ? 1 : 0;
        c_tap = wt ? 0 : c_tap;
        c_ty = wt ? 0 : c_ty;
        c_kc += wt;
      }
    };
    static_assert(RPP % 16 == 0, "the swizzle of a thread's rows must not depend on the pass");
    const int st_col = (((ld_col >> 2) ^ ((ld_row >> SW_SHIFT) & (CPR - 1))) << 2);   // swizzled float offset in the row
    auto sstore = [&](int buf, const float4 (&ra)[NA], const float4 (&rb)[NB]) {
      float* a = sA + buf * BM * LDS_LD;
      float* b = sB + buf * BN * LDS_LD;
  #pragma unroll
      for (int i = 0; i < NA; ++i)
        if (A_FULL || ld_row + i * RPP < BM)
          *reinterpret_cast<float4*>(a + (ld_row + i * RPP) * LDS_LD + st_col) = ra[i];
  #pragma unroll
      for (int i = 0; i < NB; ++i)
        if (B_FULL || ld_row + i * RPP < BN)
          *reinterpret_cast<float4*>(b + (ld_row + i * RPP) * LDS_LD + st_col) = rb[i];
    };

    // rows of a wave's 32-row groups differ by multiples of 32, so the swizzle term depends on l31 only
    const int a_row0 = (wm * TM * 32 + l31) * LDS_LD;
    const int b_row0 = (wn * TN * 32 + l31) * LDS_LD;
    const int rd_sw = (l31 >> SW_SHIFT) & (CPR - 1);
    int rd_col[BK / 8];
  #pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) rd_col[kk] = ((kk * 2 + lh) ^ rd_sw) << 2;

    auto compute = [&](int buf) {
      const float* a = sA + buf * BM * LDS_LD + a_row0;
      const float* b = sB + buf * BN * LDS_LD + b_row0;
      // operand fragments are double-buffered: the ds_reads of sub-step kk+1 are issued before the MFMAs of kk,
      // so their LDS latency hides under 4*TM*TN MFMAs instead of being waited for
      float4 fa[2][TM], fb[2][TN];
  #pragma unroll
      for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const float4*>(a + i * 32 * LDS_LD + rd_col[0]);
  #pragma unroll
      for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const float4*>(b + j * 32 * LDS_LD + rd_col[0]);
  #pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BK / 8) {
  #pragma unroll
          for (int i = 0; i < TM; ++i) fa[nxt][i] = *reinterpret_cast<const float4*>(a + i * 32 * LDS_LD + rd_col[kk + 1]);
  #pragma unroll
          for (int j = 0; j < TN; ++j) fb[nxt][j] = *reinterpret_cast<const float4*>(b + j * 32 * LDS_LD + rd_col[kk + 1]);
        }
  #pragma unroll
        for (int i = 0; i < TM; ++i)
  #pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].x, fb[cur][j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].y, fb[cur][j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].z, fb[cur][j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].w, fb[cur][j].w, acc[i][j], 0, 0, 0);
          }
      }
    };

    // prologue: chunk 0 -> LDS[0]; chunk 1 -> stage 0 registers
    gload(ra0, rb0);
    sstore(0, ra0, rb0);
    if (n_chunks > 1) gload(ra0, rb0);
    __syncthreads();
    // steady state, unrolled by two so that the register stages are named statically.  At the top of
    // step ch the LDS buffer ch&1 holds chunk ch and one register stage holds chunk ch+1 (loaded a
    // whole compute phase ago).  The stage is written to the other LDS buffer FIRST (that buffer's
    // last readers passed the barrier that ended step ch-1), then chunk ch+2 is requested, then the
    // MFMAs run: neither the LDS stores nor the global loads sit between the MFMAs and the barrier.
    int ch = 0;
    // steady state (no conditionals: chunks ch+1..ch+3 all exist)
    for (; ch + 3 < n_chunks; ch += 2) {
      sstore(1, ra0, rb0);
      gload(ra1, rb1);
      compute(0);
      __syncthreads();
      sstore(0, ra1, rb1);
      gload(ra0, rb0);
      compute(1);
      __syncthreads();
    }
    // tail: at most three chunks left
    for (; ch < n_chunks; ch += 2) {
      if (ch + 1 < n_chunks) sstore(1, ra0, rb0);
      if (ch + 2 < n_chunks) gload(ra1, rb1);
      compute(0);
      __syncthreads();
      if (ch + 1 >= n_chunks) break;
      if (ch + 2 < n_chunks) sstore(0, ra1, rb1);
      if (ch + 3 < n_chunks) gload(ra0, rb0);
      compute(1);
      __syncthreads();
    }

  } else if constexpr (MODE == 1 || MODE == 3) {
    // =============================================================== MODE 1: bf16x3 operands   (MODE 3: fp16x2, below)
    // LDS row (one pixel / one output channel, 32 input channels of one tap): 12 chunks of 16 B = 192 B, no padding;
    //   logical chunk (p*4 + g) = 8 bf16 of plane p (0 hi, 1 mid, 2 lo) and channel group g (channels 8g..8g+7),
    //   stored at chunk p*4 + (g ^ ((row >> 2) & 3)).  An MFMA k-step s (16 channels) takes group 2s from lanes 0-31
    //   and 2s+1 from lanes 32-63.  With the XOR term both the staging writes (16 lanes = 4 rows x 4 groups of one
    //   plane) and the operand reads (16 rows of one chunk) touch every LDS bank exactly once.
    // MODE 3 (fp16x2, opt-in): x 2^s = h + l with h = fp16(x 2^s), l = fp16(x 2^s - h) (both round-to-nearest; 22 of the
    //   24 significand bits), s ONE power of two per launch input (max |x| of the input tensors, published by their
    //   producers / vam_absmax, lands in [2^14, 2^15): no overflow, 18 binades at full precision) and one per output channel
    //   of the weights (made at pack time); three products hh + hl + lh on v_mfma_f32_32x32x16_f16; the scales are taken out
    //   of the accumulators, exactly, before the epilogue.  LDS row: 8 chunks of 16 B = 128 B; logical chunk (p*4 + g) is
    //   stored at chunk (p*4 + g) ^ ((row >> 1) & 7): rows alternate between the two halves of the 64 banks, so the 8 rows
    //   of one parity among 16 consecutive ones must take 8 different chunks.
    constexpr bool F16 = MODE == 3;
    constexpr int NPL = F16 ? 2 : 3;                 // operand planes
    constexpr int UPR = NPL * 4;                     // 16-byte chunks per LDS / packed-weight row
    constexpr int RS = NPL * 16;
    constexpr int NBUF = SPEC ? 2 : ((BM + BN <= 128) ? 2 : 1);   // LDS buffers: 1.5x the bytes of fp32 rows, so wide tiles single-buffer (wave-specialised blocks always double-buffer)
    constexpr int NBC = (BN * UPR + NT - 1) / NT;    // 16-byte weight chunks per thread and K chunk
    float* sA1 = smem;                               // [NBUF][BM][RS]
    float* sB1 = smem + NBUF * BM * RS;              // [NBUF][BN][RS]
    unsigned b_goff[NBC];
    int b_loff[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) {
      const int idx = tid + j * NT;
      const int row = idx / UPR, c = idx - row * UPR;
      const bool in_tile = idx < BN * UPR;
      b_goff[j] = (in_tile && n0 + row < u_Npad) ? (unsigned)((n0 + row) * (UPR * 16) + c * 16) : 0x80000000u;
      b_loff[j] = in_tile ? row * RS + (F16 ? (c ^ ((row >> 1) & 7)) : ((c & ~3) | ((c & 3) ^ ((row >> 2) & 3)))) * 4 : -1;
    }
    const int u_Cin = __builtin_amdgcn_readfirstlane(P.Cin);
    // MODE 3: the launch input's power-of-two scale from the published max |x| (of x^2 for GDN's pooled squares)
    float a_scale = 1.f, a_unscale = 1.f;
    if constexpr (F16) {
      int mb = 0;                                    // non-negative floats: integer order = float order
#pragma unroll
      for (int k = 0; k < VAM_MAX_SEG; ++k)
        if (P.in_amax[k]) mb = max(mb, *P.in_amax[k]);
      float m = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(mb));
      if (sq) m = m * m;
      const int e = (int)((__float_as_uint(m) >> 23) & 0xFFu) - 127;             // floor(log2 m) of a normal m
      int sh = (m > 0.f && e > -127 && e < 128) ? 14 - e : 0;
      sh = sh > 126 ? 126 : (sh < -126 ? -126 : sh);
      a_scale = __uint_as_float((unsigned)(sh + 127) << 23);
      a_unscale = __uint_as_float((unsigned)(127 - sh) << 23);
    }
    // register stages: chunk c lives in stage c & 1 between its global loads and its LDS store.  Double-buffered
    // (small) tiles use both stages, so a load has TWO compute phases to land — their phases are only 12-24 MFMAs long;
    // single-buffered (wide) tiles use stage 0 only.
    constexpr int NAR = AIN ? 3 : 2;                 // 16-byte registers per staged A unit (P3: three planes; fp32: 8 floats)
    u32x4 ra0[NA][NAR], ra1[NA][NAR];
    u32x4 rb0[NBC], rb1[NBC];

    auto gload = [&](u32x4 (&ra)[NA][NAR], u32x4 (&rb)[NBC]) {
      const int cc0 = c_kc * 32;
      if (cc0 < s_begin || cc0 >= s_end) {           // input segment changed (rare; see MODE 0)
        const int k = (cc0 >= u_se0 ? 1 : 0) + (cc0 >= u_se1 ? 1 : 0) + (cc0 >= u_se2 ? 1 : 0);
        s_begin = __builtin_amdgcn_readfirstlane(k ? P.seg_end[k - 1] : 0);
        s_end = __builtin_amdgcn_readfirstlane(P.seg_end[k]);
        const unsigned long long spa = reinterpret_cast<unsigned long long>(P.seg_ptr[k]);
        s_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)spa);
        s_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(spa >> 32));
        s_ld4 = __builtin_amdgcn_readfirstlane(P.seg_ld[k]) * (AIN ? 48 : 4);   // bytes per pixel (P3: ld counts 8-channel groups)
      }
      const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(((unsigned long long)s_hi << 32) | s_lo), 0, 0x7FFFFFFF, 0x00020000);
      int tap_pix = c_ty * u_W + c_tx;
      // byte offset of this thread's 8 channels inside a pixel: 32 B of fp32, or one 48-byte P3 group
      int col4 = AIN ? ((cc0 - s_begin + ld_col) >> 3) * 48 : (cc0 - s_begin + ld_col) * 4;
      bool ch_ok = cc0 + ld_col < u_Cin;             // the last chunk of a 16-mod-32 channel count is half empty
      int my_tap = c_tap;
      if (!AIN && dual) {                            // (block-uniform) this thread's half of the chunk belongs to tap c_tap or c_tap + 1
        const bool second = ld_col >= 16;
        const int txb = (c_tx + 1 == u_kw) ? 0 : c_tx + 1, tyb = (c_tx + 1 == u_kw) ? c_ty + 1 : c_ty;
        tap_pix = second ? tyb * u_W + txb : tap_pix;
        my_tap = second ? c_tap + 1 : c_tap;
        col4 = (ld_col & 15) * 4;
        ch_ok = my_tap < n_taps;
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool ok = ch_ok && ((a_mask[i] >> my_tap) & 1u);
        const unsigned off = ok ? (unsigned)((a_pix0[i] + tap_pix) * s_ld4 + col4) : 0x80000000u;
#pragma unroll
        for (int q = 0; q < NAR; ++q) ra[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)(off + 16u * q), 0, 0);
      }
      // weights: [tap][32-channel chunk][Npad][12 chunks of 8 bf16] = 192 B per (n, chunk), pre-split at pack time
      // (dual: [tap pair][Npad][...], the pair's two 16-channel halves side by side)
      const unsigned wbase = (unsigned)((dual ? (c_tap >> 1) : (c_tap * u_Kc + c_kc)) * u_Npad) * (unsigned)(UPR * 16);
#pragma unroll
      for (int j = 0; j < NBC; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)(b_goff[j] + wbase), 0, 0);
      // canonical K order: 32-channel group OUTER, tap INNER
      for (int step = 0; step < (dual ? 2 : 1); ++step) {
        ++c_tap;
        ++c_tx;
        const int wx = (c_tx == u_kw) ? 1 : 0;
        c_tx = wx ? 0 : c_tx;
        c_ty += wx;
      }
      const int wt = (c_tap >= n_taps) ? 1 : 0;
      c_tap = wt ? 0 : c_tap;
      c_ty = wt ? 0 : c_ty;
      c_tx = wt ? 0 : c_tx;
      c_kc += wt;
    };
    static_assert(RPP % 16 == 0, "the swizzle of a thread's rows must not depend on the pass");
    const int st1_col = (((ld_col >> 3) ^ ((ld_row >> 2) & 3)) << 2);
    const int st3_col[2] = {((((ld_col >> 3)) ^ ((ld_row >> 1) & 7)) << 2), (((4 + (ld_col >> 3)) ^ ((ld_row >> 1) & 7)) << 2)};   // MODE 3
    auto sstore = [&](int buf, const u32x4 (&ra)[NA][NAR], const u32x4 (&rb)[NBC]) {
      float* a = sA1 + buf * BM * RS;
      float* b = sB1 + buf * BN * RS;
#pragma unroll
      for (int i = 0; i < NA; ++i)
        if (A_FULL || ld_row + i * RPP < BM) {
          if constexpr (AIN) {                       // planes arrive ready-made: three straight copies
            float* dst = a + (ld_row + i * RPP) * RS + st1_col;
            *reinterpret_cast<u32x4*>(dst) = ra[i][0];
            *reinterpret_cast<u32x4*>(dst + 16) = ra[i][1];
            *reinterpret_cast<u32x4*>(dst + 32) = ra[i][NAR - 1];
            continue;
          }
          float x[8] = {__uint_as_float(ra[i][0].x), __uint_as_float(ra[i][0].y), __uint_as_float(ra[i][0].z), __uint_as_float(ra[i][0].w),
                        __uint_as_float(ra[i][1].x), __uint_as_float(ra[i][1].y), __uint_as_float(ra[i][1].z), __uint_as_float(ra[i][1].w)};
          if (sq) {                                  // GDN pools x^2 (block-uniform branch)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = x[e] * x[e];
          }
          if constexpr (F16) {
            // x 2^s = h + l: h nearest fp16, the remainder (exact in fp32) rounded to fp16 once more
            unsigned hw2[4], lw2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float x0 = x[2 * q] * a_scale, x1 = x[2 * q + 1] * a_scale;
              const f16x2 h = {(_Float16)x0, (_Float16)x1};
              const f16x2 l = {(_Float16)(x0 - (float)h[0]), (_Float16)(x1 - (float)h[1])};
              hw2[q] = __builtin_bit_cast(unsigned, h);
              lw2[q] = __builtin_bit_cast(unsigned, l);
            }
            float* drow = a + (ld_row + i * RPP) * RS;
            u32x4 t;
            t.x = hw2[0]; t.y = hw2[1]; t.z = hw2[2]; t.w = hw2[3];
            *reinterpret_cast<u32x4*>(drow + st3_col[0]) = t;
            t.x = lw2[0]; t.y = lw2[1]; t.z = lw2[2]; t.w = lw2[3];
            *reinterpret_cast<u32x4*>(drow + st3_col[1]) = t;
            continue;
          }
          // exact 3-way split by truncation: hi = top 16 bits of x, mid = top 16 bits of (x - hi), lo = x - hi - mid
          // (at most 8 significant bits are left, so its top 16 bits hold it exactly); two AND + two SUB per element
          // and one byte-permute per plane and element pair
          unsigned hb[8], mb[8], lb[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            hb[e] = __float_as_uint(x[e]);
            const float r1 = x[e] - __uint_as_float(hb[e] & 0xFFFF0000u);
            mb[e] = __float_as_uint(r1);
            lb[e] = __float_as_uint(r1 - __uint_as_float(mb[e] & 0xFFFF0000u));
          }
          unsigned hw[4], mw[4], lw[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            hw[q] = __builtin_amdgcn_perm(hb[2 * q + 1], hb[2 * q], 0x07060302u);   // {hi16(odd), hi16(even)}
            mw[q] = __builtin_amdgcn_perm(mb[2 * q + 1], mb[2 * q], 0x07060302u);
            lw[q] = __builtin_amdgcn_perm(lb[2 * q + 1], lb[2 * q], 0x07060302u);
          }
          float* dst = a + (ld_row + i * RPP) * RS + st1_col;             // swizzled chunk of plane 0; planes are 16 floats apart
          u32x4 t;
          t.x = hw[0]; t.y = hw[1]; t.z = hw[2]; t.w = hw[3];
          *reinterpret_cast<u32x4*>(dst) = t;
          t.x = mw[0]; t.y = mw[1]; t.z = mw[2]; t.w = mw[3];
          *reinterpret_cast<u32x4*>(dst + 16) = t;
          t.x = lw[0]; t.y = lw[1]; t.z = lw[2]; t.w = lw[3];
          *reinterpret_cast<u32x4*>(dst + 32) = t;
        }
#pragma unroll
      for (int j = 0; j < NBC; ++j) {
        if (b_loff[j] >= 0) *reinterpret_cast<u32x4*>(b + b_loff[j]) = rb[j];
      }
    };
    const int a_row1 = (wm * TM * 32 + l31) * RS;
    const int b_row1 = (wn * TN * 32 + l31) * RS;
    const int rsw = (l31 >> 2) & 3;                  // rows of a wave's 32-row groups differ by multiples of 32
    const int rd1[2] = {((lh ^ rsw) << 2), (((2 + lh) ^ rsw) << 2)};
    const int rsw3 = (l31 >> 1) & 7;
    const int rd3[2][2] = {{((lh ^ rsw3) << 2), (((4 + lh) ^ rsw3) << 2)}, {(((2 + lh) ^ rsw3) << 2), (((6 + lh) ^ rsw3) << 2)}};   // MODE 3: [k-step][plane]
    auto compute = [&](int buf) {
      const float* a = sA1 + buf * BM * RS + a_row1;
      const float* b = sB1 + buf * BN * RS + b_row1;
      if constexpr (F16) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f16x8 fa[TM][2], fb[TN][2];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fa[i][pl] = *reinterpret_cast<const f16x8*>(a + i * 32 * RS + rd3[ks][pl]);
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(b + j * 32 * RS + rd3[ks][pl]);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              // fixed order, smallest terms first: (h,l) (l,h) (h,h)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
            }
        }
        return;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fa[i][pl] = *reinterpret_cast<const bf16x8*>(a + i * 32 * RS + pl * 16 + rd1[ks]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fb[j][pl] = *reinterpret_cast<const bf16x8*>(b + j * 32 * RS + pl * 16 + rd1[ks]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            // fixed order, smallest terms first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
          }
        // seven column groups per wave: 112 accumulator + 96 fragment registers; keep hipcc from hoisting the next
        // sub-step's 96 fragment registers above this one's MFMAs (39 spilled registers otherwise)
        if constexpr (TN >= 7) __builtin_amdgcn_sched_barrier(0);
      }
    };
    // MODE 3: take the two scales out of the accumulators (powers of two: exact).  The per-channel factors 2^-s_n sit
    // behind the packed weights ([taps][Kc][Npad] rows of 128 B, then Npad floats).
    auto unscale = [&]() {
      if constexpr (F16) {
        const float* wsc = P.wpack + (size_t)n_taps * u_Kc * u_Npad * 32;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + (wn * TN + j) * 32 + l31;
          const float f = (n < u_Npad ? wsc[n] : 0.f) * a_unscale;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
        }
      }
    };
    if constexpr (SPEC) {
      // chunk c sits in register stage c & 1 between its global loads and its LDS store, and in LDS buffer c & 1
      // afterwards.  Iteration ch: consumers compute from buffer ch & 1 while loaders store chunk ch+1 into the other
      // buffer (its last readers passed the barrier that ended iteration ch-1) and request chunk ch+3.  Both roles run
      // the same number of barriers.
      if (is_loader) {
        gload(ra0, rb0);
        if (n_chunks > 1) gload(ra1, rb1);
        sstore(0, ra0, rb0);
        if (n_chunks > 2) gload(ra0, rb0);
        __syncthreads();
        for (int ch = 0; ch < n_chunks; ch += 2) {
          if (ch + 1 < n_chunks) sstore(1, ra1, rb1);
          if (ch + 3 < n_chunks) gload(ra1, rb1);
          __syncthreads();
          if (ch + 1 >= n_chunks) break;
          if (ch + 2 < n_chunks) sstore(0, ra0, rb0);
          if (ch + 4 < n_chunks) gload(ra0, rb0);
          __syncthreads();
        }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // (the consumers' operand reads and MFMAs are left to hipcc's scheduler: a hand-pipelined variant that issued
        // the next sub-step's fragment reads ahead of each MFMA block and pinned the order with sched_barrier(0) measured
        // 10-25 % SLOWER on every tile — the pinned order keeps the compiler from interleaving reads and MFMAs)
        __syncthreads();
        for (int ch = 0; ch < n_chunks; ch += 2) {
          compute(0);
          __syncthreads();
          if (ch + 1 >= n_chunks) break;
          compute(1);
          __syncthreads();
        }
        unscale();
        // the whole C tile goes to LDS (it fits in the pipeline buffers, which every wave has left behind the last
        // barrier); C layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        if (!direct_out) {
          float* sCf = smem;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                sCf[row * (BN + 4) + wn * TN * 32 + j * 32 + l31] = acc[i][j][r];
              }
        }
      }
    } else if constexpr (NBUF == 2) {
      // one barrier per chunk; at the top of step ch: LDS[ch&1] = chunk ch, stage (ch+1)&1 = chunk ch+1 (requested two
      // steps ago), stage ch&1 = chunk ch+2 (requested one step ago)
      gload(ra0, rb0);
      sstore(0, ra0, rb0);
      if (n_chunks > 1) gload(ra1, rb1);
      if (n_chunks > 2) gload(ra0, rb0);
      __syncthreads();
      int ch = 0;
      for (; ch + 1 < n_chunks; ch += 2) {
        sstore(1, ra1, rb1);                          // chunk ch+1
        if (ch + 3 < n_chunks) gload(ra1, rb1);       // chunk ch+3
        compute(0);
        __syncthreads();
        if (ch + 2 < n_chunks) sstore(0, ra0, rb0);   // chunk ch+2
        if (ch + 4 < n_chunks) gload(ra0, rb0);       // chunk ch+4
        compute(1);
        __syncthreads();
      }
      if (ch < n_chunks) {                            // odd chunk count: the last chunk sits in LDS[0]
        compute(0);
        __syncthreads();
      }
      unscale();
    } else {
      // single LDS buffer (wide tiles): registers hold chunk ch+1 while chunk ch is computed
      gload(ra0, rb0);
      for (int ch = 0; ch < n_chunks; ++ch) {
        sstore(0, ra0, rb0);
        __syncthreads();
        if (ch + 1 < n_chunks) gload(ra0, rb0);
        compute(0);
        __syncthreads();
      }
      unscale();
    }
  } else {
    // =============================================================== MODE 2: bf16 storage (BASELINE configs[2] "bf16")
    // Activations of the large feature maps are STORED as bf16 (AIN = 1: [pixel][C] bf16, 16-byte loads of 8 channels)
    // or arrive as fp32 and are rounded while they are staged (AIN = 0); the weights of these layers are rounded to
    // bf16 once at pack time ([tap][32-channel chunk][n][32 bf16], 64 B per row).  One v_mfma_f32_32x32x16_bf16 per
    // 32x32x16 block, fp32 accumulation: a sixth of the split-operand kernel's matrix work and a third of its LDS
    // bytes.  LDS row = 4 chunks of 16 B (8 channels each), chunk g stored at g ^ ((row >> 2) & 3): staging writes and
    // operand reads both conflict-free.  Double-buffered LDS, two register stages.
    constexpr int RS = 16;
    constexpr int NBC = (BN * 4 + NT - 1) / NT;
    float* sA2 = smem;                               // [2][BM][RS]
    float* sB2 = smem + 2 * BM * RS;                 // [2][BN][RS]
    unsigned b_goff[NBC];
    int b_loff[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) {
      const int idx = tid + j * NT;
      const int row = idx >> 2, c = idx & 3;
      const bool in_tile = idx < BN * 4;
      b_goff[j] = (in_tile && n0 + row < u_Npad) ? (unsigned)((n0 + row) * 64 + c * 16) : 0x80000000u;
      b_loff[j] = in_tile ? row * RS + ((c ^ ((row >> 2) & 3)) << 2) : -1;
    }
    const int u_Cin = __builtin_amdgcn_readfirstlane(P.Cin);
    constexpr int NAR = AIN ? 1 : 2;                 // 16-byte registers per staged A unit of 8 channels
    u32x4 ra0[NA][NAR], ra1[NA][NAR];
    u32x4 rb0[NBC], rb1[NBC];
    auto gload = [&](u32x4 (&ra)[NA][NAR], u32x4 (&rb)[NBC]) {
      const int cc0 = c_kc * 32;
      if (cc0 < s_begin || cc0 >= s_end) {
        const int k = (cc0 >= u_se0 ? 1 : 0) + (cc0 >= u_se1 ? 1 : 0) + (cc0 >= u_se2 ? 1 : 0);
        s_begin = __builtin_amdgcn_readfirstlane(k ? P.seg_end[k - 1] : 0);
        s_end = __builtin_amdgcn_readfirstlane(P.seg_end[k]);
        const unsigned long long spa = reinterpret_cast<unsigned long long>(P.seg_ptr[k]);
        s_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)spa);
        s_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(spa >> 32));
        s_ld4 = __builtin_amdgcn_readfirstlane(P.seg_ld[k]) * (AIN ? 2 : 4);     // bytes per pixel
      }
      const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(((unsigned long long)s_hi << 32) | s_lo), 0, 0x7FFFFFFF, 0x00020000);
      const int tap_pix = c_ty * u_W + c_tx;
      const int col4 = (cc0 - s_begin + ld_col) * (AIN ? 2 : 4);
      const bool ch_ok = cc0 + ld_col < u_Cin;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool ok = ch_ok && ((a_mask[i] >> c_tap) & 1u);
        const unsigned off = ok ? (unsigned)((a_pix0[i] + tap_pix) * s_ld4 + col4) : 0x80000000u;
#pragma unroll
        for (int q = 0; q < NAR; ++q) ra[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)(off + 16u * q), 0, 0);
      }
      const unsigned wbase = (unsigned)((c_tap * u_Kc + c_kc) * u_Npad) * 64u;
#pragma unroll
      for (int j = 0; j < NBC; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)(b_goff[j] + wbase), 0, 0);
      ++c_tap;
      ++c_tx;
      const int wx = (c_tx == u_kw) ? 1 : 0;
      c_tx = wx ? 0 : c_tx;
      c_ty += wx;
      const int wt = (c_tap == n_taps) ? 1 : 0;
      c_tap = wt ? 0 : c_tap;
      c_ty = wt ? 0 : c_ty;
      c_kc += wt;
    };
    auto pk2 = [](float lo, float hi) -> unsigned {  // two fp32 -> two bf16 (round to nearest even), lo in the low half
      const __bf16 a = (__bf16)lo, b = (__bf16)hi;
      return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
    };
    const int st2_col = (((ld_col >> 3) ^ ((ld_row >> 2) & 3)) << 2);
    auto sstore = [&](int buf, const u32x4 (&ra)[NA][NAR], const u32x4 (&rb)[NBC]) {
      float* a = sA2 + buf * BM * RS;
      float* b = sB2 + buf * BN * RS;
#pragma unroll
      for (int i = 0; i < NA; ++i)
        if (A_FULL || ld_row + i * RPP < BM) {
          u32x4 t;
          if constexpr (AIN) {
            t = ra[i][0];
            if (sq) {                                // GDN pools x^2: square in fp32, round back to bf16
              unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float lo = __uint_as_float(w[q] << 16), hi = __uint_as_float(w[q] & 0xFFFF0000u);
                w[q] = pk2(lo * lo, hi * hi);
              }
              t.x = w[0]; t.y = w[1]; t.z = w[2]; t.w = w[3];
            }
          } else {
            float x[8] = {__uint_as_float(ra[i][0].x), __uint_as_float(ra[i][0].y), __uint_as_float(ra[i][0].z), __uint_as_float(ra[i][0].w),
                          __uint_as_float(ra[i][NAR - 1].x), __uint_as_float(ra[i][NAR - 1].y), __uint_as_float(ra[i][NAR - 1].z), __uint_as_float(ra[i][NAR - 1].w)};
            if (sq) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = x[e] * x[e];
            }
            t.x = pk2(x[0], x[1]); t.y = pk2(x[2], x[3]); t.z = pk2(x[4], x[5]); t.w = pk2(x[6], x[7]);
          }
          *reinterpret_cast<u32x4*>(a + (ld_row + i * RPP) * RS + st2_col) = t;
        }
#pragma unroll
      for (int j = 0; j < NBC; ++j)
        if (b_loff[j] >= 0) *reinterpret_cast<u32x4*>(b + b_loff[j]) = rb[j];
    };
    const int a_row2 = (wm * TM * 32 + l31) * RS;
    const int b_row2 = (wn * TN * 32 + l31) * RS;
    const int rsw = (l31 >> 2) & 3;
    const int rd2[2] = {((lh ^ rsw) << 2), (((2 + lh) ^ rsw) << 2)};
    auto compute = [&](int buf) {
      const float* a = sA2 + buf * BM * RS + a_row2;
      const float* b = sB2 + buf * BN * RS + b_row2;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * RS + rd2[ks]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * RS + rd2[ks]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    };
    gload(ra0, rb0);
    sstore(0, ra0, rb0);
    if (n_chunks > 1) gload(ra1, rb1);
    if (n_chunks > 2) gload(ra0, rb0);
    __syncthreads();
    int ch = 0;
    for (; ch + 1 < n_chunks; ch += 2) {
      sstore(1, ra1, rb1);
      if (ch + 3 < n_chunks) gload(ra1, rb1);
      compute(0);
      __syncthreads();
      if (ch + 2 < n_chunks) sstore(0, ra0, rb0);
      if (ch + 4 < n_chunks) gload(ra0, rb0);
      compute(1);
      __syncthreads();
    }
    if (ch < n_chunks) {
      compute(0);
      __syncthreads();
    }
  }

  // ---- epilogue.  The block's C tile goes through LDS one 32-row slab per wave-row at a time
  // (C layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)), then is
  // streamed out with one float4 of 4 consecutive channels per lane: 16-byte aux loads and
  // stores, and the activation code exists once instead of once per accumulator register.
  constexpr int LDC = BN + 4;
  constexpr int SROWS = SPEC ? BM : WGM * 32;   // rows of C held in LDS at a time (SPEC: the whole tile)
  float* sC = smem;  // the pipeline buffers are dead after the last barrier of the loop
  int* sPix = reinterpret_cast<int*>(smem + SROWS * LDC);   // per LDS row: output pixel index, batch index
  const bool ps2 = (P.flags & VAM_CONV_PS2) != 0;
  const bool nchw = (P.flags & VAM_CONV_OUT_NCHW) != 0;
  const bool dense = !ps2 && !nchw && P.osy == 1 && P.osx == 1 && P.ooy == 0 && P.oox == 0 &&
                     P.Hf == P.Ho && P.Wf == P.Wo;
  const bool vec_ok = !nchw && (!ps2 || (P.Cq & 3) == 0);
  const bool out_p3 = (P.flags & VAM_CONV_OUT_BF3) != 0;     // host guarantees: split mode, vec_ok, no PS2
  const bool out16 = (P.flags & VAM_CONV_OUT_BF16) != 0;     // bf16 NHWC output (ldo counts bf16 elements); host guarantees vec_ok
  const bool aux16 = (P.flags & VAM_CONV_AUX_BF16) != 0;     // pre / mul / post / post2 are bf16 NHWC tensors
#ifdef VAM_NO_TRAIN_EPILOGUE     // A/B build (scratch/ab_epilogue.sh): the training-only epilogue features compiled out
  const bool mulg = false;
#define VAM_PREACT(P) false
#else
  const bool mulg = (P.flags & VAM_CONV_MUL_GELU_GRAD) != 0; // the mul operand is a GELU's pre-activation z: multiply by gelu'(z)
#define VAM_PREACT(P) ((P).preact != nullptr)
#endif
  auto ld_aux = [&](const float* base, size_t off) -> float4 {   // four consecutive channels of an epilogue operand
    if (aux16) {
      const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
      return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                         __uint_as_float(u.y & 0xFFFF0000u));
    }
    return *reinterpret_cast<const float4*>(base + off);
  };
  const int Cc = ps2 ? P.Cq : P.N;
  const size_t HfWf = (size_t)P.Hf * P.Wf;
  // output position of a tile row (the integer divisions happen once per row, not once per element): pixel index
  // within the full output, and batch index
  auto decode_row = [&](int row, int slot) {
    const int p = m0 + row;
    int pix = p, ob = 0;
    if (!dense && p < P.P) {
      ob = p / P.HoWo;
      int rr = p - ob * P.HoWo;
      int oy = rr / P.Wo;
      int ox = rr - oy * P.Wo;
      if (ps2) pix = (ob * P.Hf + 2 * oy) * P.Wf + 2 * ox;
      else pix = (ob * P.Hf + oy * P.osy + P.ooy) * P.Wf + ox * P.osx + P.oox;
    }
    sPix[2 * slot] = pix;
    sPix[2 * slot + 1] = ob;
  };
  // max |stored value| of this thread, folded into P.out_amax at the end (the consumer of this tensor may run in MODE 3)
  float omax = 0.f;
  const bool pub = P.out_amax != nullptr;
  auto publish = [&]() {
    if (pub) {
      float m = omax;
#pragma unroll
      for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      // |x| >= 0: integer order = float order.  Thousands of waves fold into one cell: look first (a relaxed atomic load is
      // served by L2, where the atomics land) and skip the read-modify-write unless this wave raises the maximum —
      // same-address atomics serialise (measured: 49k of them cost 0.7 ms of a 0.6 ms launch)
      if ((threadIdx.x & 63) == 0 && m > 0.f && __float_as_int(m) > __hip_atomic_load(P.out_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(P.out_amax, __float_as_int(m));
    }
  };
  // four consecutive channels (c4..c4+3) of tile row `row`, held in LDS row `srow`
  auto emit = [&](int srow, int row, int c4) {
    const int p = m0 + row;
    const int n = n0 + c4;
    if (p >= P.P || n >= P.N) return;
    const float4 av = *reinterpret_cast<const float4*>(sC + srow * LDC + c4);
    float v[4] = {av.x, av.y, av.z, av.w};
    const int pixb = sPix[2 * srow], ob = sPix[2 * srow + 1];
    if (vec_ok) {
      int cch = n;
      size_t opix = (size_t)pixb;
      if (ps2) {
        int ph = n / P.Cq;
        cch = n - ph * P.Cq;
        opix += (size_t)(ph >> 1) * P.Wf + (ph & 1);
      }
      if (P.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(P.bias + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
      }
      if (P.pre) {
        const float4 t4 = ld_aux(P.pre, opix * P.ld_pre + cch);
        v[0] += t4.x; v[1] += t4.y; v[2] += t4.z; v[3] += t4.w;
      }
      if (VAM_PREACT(P)) *reinterpret_cast<float4*>(P.preact + opix * P.ld_preact + cch) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k], P.act);
      if (P.mul) {
        float4 t4 = ld_aux(P.mul, opix * P.ld_mul + cch);
        if (mulg) t4 = make_float4(vam_gelu_grad(t4.x), vam_gelu_grad(t4.y), vam_gelu_grad(t4.z), vam_gelu_grad(t4.w));
        v[0] *= t4.x; v[1] *= t4.y; v[2] *= t4.z; v[3] *= t4.w;
      }
      if (P.post) {
        const float4 t4 = ld_aux(P.post, opix * P.ld_post + cch);
        v[0] += t4.x; v[1] += t4.y; v[2] += t4.z; v[3] += t4.w;
      }
      if (P.post2) {
        const float4 t4 = ld_aux(P.post2, opix * P.ld_post2 + cch);
        v[0] += t4.x; v[1] += t4.y; v[2] += t4.z; v[3] += t4.w;
      }
      if (pub) omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      if (out_p3) {
        // bf16x3 planes for the consuming convolution: [pixel][8-channel group][plane][8 bf16]; this lane owns
        // channels cch..cch+3 = 8 bytes of each plane
        unsigned hb[4], mb[4], lb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          hb[k] = __float_as_uint(v[k]);
          const float r1 = v[k] - __uint_as_float(hb[k] & 0xFFFF0000u);
          mb[k] = __float_as_uint(r1);
          lb[k] = __float_as_uint(r1 - __uint_as_float(mb[k] & 0xFFFF0000u));
        }
        char* o3 = reinterpret_cast<char*>(P.out) + (opix * P.ldo + (size_t)(cch >> 3)) * 48 + (cch & 7) * 2;
        *reinterpret_cast<uint2*>(o3) = make_uint2(__builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u), __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u));
        *reinterpret_cast<uint2*>(o3 + 16) = make_uint2(__builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u), __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u));
        *reinterpret_cast<uint2*>(o3 + 32) = make_uint2(__builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u), __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u));
      } else if (out16) {
        const __bf16 h0 = (__bf16)v[0], h1 = (__bf16)v[1], h2 = (__bf16)v[2], h3 = (__bf16)v[3];
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(P.out) + opix * P.ldo + cch) =
            make_uint2((unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16),
                       (unsigned)__builtin_bit_cast(unsigned short, h2) | ((unsigned)__builtin_bit_cast(unsigned short, h3) << 16));
      } else {
        *reinterpret_cast<float4*>(P.out + opix * P.ldo + cch) = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else {
      // scalar path: model-edge NCHW store and phase groups that are not multiples of 4
      for (int k = 0; k < 4; ++k) {
        const int nn = n + k;
        if (nn >= P.N) break;
        int cch = nn;
        size_t opix = (size_t)pixb;
        if (ps2) {
          int ph = nn / P.Cq;
          cch = nn - ph * P.Cq;
          opix += (size_t)(ph >> 1) * P.Wf + (ph & 1);
        }
        float x = v[k] + (P.bias ? P.bias[nn] : 0.f);
        if (P.pre) x = x + P.pre[opix * P.ld_pre + cch];
        if (VAM_PREACT(P)) P.preact[opix * P.ld_preact + cch] = x;
        x = apply_act(x, P.act);
        if (P.mul) x = x * (mulg ? vam_gelu_grad(P.mul[opix * P.ld_mul + cch]) : P.mul[opix * P.ld_mul + cch]);
        if (P.post) x = x + P.post[opix * P.ld_post + cch];
        if (P.post2) x = x + P.post2[opix * P.ld_post2 + cch];
        if (pub) omax = fmaxf(omax, fabsf(x));
        if (nchw) P.out[((size_t)ob * Cc + cch) * HfWf + (opix - (size_t)ob * HfWf)] = x;
        else P.out[opix * P.ldo + cch] = x;
      }
    }
  };
  if constexpr (DIRECT_CFG) if (direct_out) {
    // ---- direct path: every accumulator register of the 32x32 MFMA is two 128-byte row segments (32 consecutive
    // channels of rows R and R+4), which the memory pipe takes at its full rate as one dword store per lane
    // (MI355X_MICROARCH.md, "plain stores of the same shape").  No LDS round trip and no barrier (one, for the row
    // table, when the output is a strided view): the operand loads of a 32x32 block are all issued before the first
    // one is needed, where the staged path below waits for one bias / operand load after the other, slab by slab.
    // Same operations in the same order per element as the staged path: bit-identical results
    // (tests/test_gpu_ops.py::test_direct_and_staged_epilogues_are_bit_identical).
    // (The training-only epilogue features — second output `preact`, gelu'(mul) — live in the staged path alone: the host
    // sends problems that use them there.  Compiled into this path they cost the inference step 0.07 ms, interleaved A/B,
    // gpurun_out/r4_epi*.log.)
    // Kept for the 128x64 one-role tile only, the tile of the short-K, store-bound layers (1x1 convolutions of GDN,
    // residual units and attention, the 16-channel first layer), where it measured 6-17 % faster (scratch/ab_epi*.sh:
    // 4x[96->192 1x1] 250 -> 208 us, 2x[192->192 1x1 @128] 667 -> 596 us).  Elsewhere it measured no better or worse:
    // equal on deep-K one-role tiles, 11 % slower on the 4x1-wave 128x96 tile with 384-byte rows, and 8-22 % slower in
    // wave-specialised blocks, whose loader waves share the staged epilogue's work but hold no accumulators.
    int* sRow = reinterpret_cast<int*>(smem);
    if (!dense) {
      if ((int)threadIdx.x < BM) {
        const int p = m0 + (int)threadIdx.x;
        int pix = p;
        if (p < P.P) {
          const int ob = p / P.HoWo;
          const int rr = p - ob * P.HoWo;
          const int oy = rr / P.Wo;
          const int ox = rr - oy * P.Wo;
          if (ps2) pix = (ob * P.Hf + 2 * oy) * P.Wf + 2 * ox;
          else pix = (ob * P.Hf + oy * P.osy + P.ooy) * P.Wf + ox * P.osx + P.oox;
        }
        sRow[threadIdx.x] = pix;
      }
      __syncthreads();
    }
    if (is_loader) return;
    // 32-bit byte offsets against wave-uniform descriptors (the host sends tensors beyond the 2^31-byte window, or with
    // 2^22 pixels or more, down the staged path): one v_mad_u32_u24 per element and operand, and rows / columns
    // outside the problem are dropped by the hardware range check instead of branches
    auto desc = [&](const void* q) {
      const unsigned long long a = reinterpret_cast<unsigned long long>(q);
      return __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a)), 0, 0x7FFFFFFF, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t r_out = desc(P.out), r_pre = desc(P.pre), r_mul = desc(P.mul), r_post = desc(P.post),
                                 r_post2 = desc(P.post2);
    const unsigned u_ldo = (unsigned)__builtin_amdgcn_readfirstlane(P.ldo), u_ldpre = (unsigned)__builtin_amdgcn_readfirstlane(P.ld_pre);
    const unsigned u_ldmul = (unsigned)__builtin_amdgcn_readfirstlane(P.ld_mul), u_ldpost = (unsigned)__builtin_amdgcn_readfirstlane(P.ld_post);
    const unsigned u_ldpost2 = (unsigned)__builtin_amdgcn_readfirstlane(P.ld_post2);
    const int u_act = __builtin_amdgcn_readfirstlane(P.act);
    const bool full_rows = m0 + BM <= P.P;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row0 = wm * TM * 32 + i * 32 + 4 * lh;
      unsigned pix4[16];                     // 4 x output pixel index of the 16 rows this lane holds of block row i
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int4 px = make_int4(0, 0, 0, 0);
        if (!dense) px = *reinterpret_cast<const int4*>(sRow + row0 + 8 * g);
        const int pxs[4] = {px.x, px.y, px.z, px.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int p = m0 + row0 + 8 * g + k;
          pix4[g * 4 + k] = (full_rows || p < P.P) ? (unsigned)(dense ? p : pxs[k]) << 2 : 0xFFFFFFFFu;
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + l31;
        if (n < P.N) {
          unsigned cch = (unsigned)n, poff = 0;
          if (ps2) {
            const unsigned ph = cch / (unsigned)P.Cq;
            cch -= ph * (unsigned)P.Cq;
            poff = (ph >> 1) * (unsigned)P.Wf + (ph & 1);
          }
          // byte offset of element (row r, this lane's channel) of an operand with row pitch ld
          auto off = [&](int r, unsigned ld, unsigned c4) -> int {
            const unsigned o = __umul24(pix4[r], ld) + c4;
            return (int)((full_rows || pix4[r] != 0xFFFFFFFFu) ? o : 0x80000000u);
          };
          float v[16], t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
          if (P.bias) {
            const float bv = P.bias[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += bv;
          }
          if (P.pre) {
            const unsigned c4 = (poff * u_ldpre + cch) << 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_pre, off(r, u_ldpre, c4), 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += t[r];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = apply_act(v[r], u_act);
          if (P.mul) {
            const unsigned c4 = (poff * u_ldmul + cch) << 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_mul, off(r, u_ldmul, c4), 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= t[r];
          }
          if (P.post) {
            const unsigned c4 = (poff * u_ldpost + cch) << 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_post, off(r, u_ldpost, c4), 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += t[r];
          }
          if (P.post2) {
            const unsigned c4 = (poff * u_ldpost2 + cch) << 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_post2, off(r, u_ldpost2, c4), 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += t[r];
          }
          const unsigned c4o = (poff * u_ldo + cch) << 2;
          if (pub) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (full_rows || pix4[r] != 0xFFFFFFFFu) omax = fmaxf(omax, fabsf(v[r]));
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), r_out, off(r, u_ldo, c4o), 0, 0);
        }
      }
    }
    publish();
    return;
  }
  if constexpr (SPEC) {
    if ((int)threadIdx.x < BM) decode_row((int)threadIdx.x, (int)threadIdx.x);
    __syncthreads();                         // the consumers' C tile and the row table are in LDS
    for (int it = (int)threadIdx.x; it < BM * (BN / 4); it += NTB) {
      const int row = it / (BN / 4);
      emit(row, row, (it - row * (BN / 4)) * 4);
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i > 0) __syncthreads();
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int srow = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          sC[srow * LDC + wn * TN * 32 + j * 32 + l31] = acc[i][j][r];
        }
      if (tid < SROWS) decode_row((tid >> 5) * (TM * 32) + i * 32 + (tid & 31), tid);
      __syncthreads();
      for (int it = tid; it < SROWS * (BN / 4); it += NT) {
        const int srow = it / (BN / 4);
        emit(srow, (srow >> 5) * (TM * 32) + i * 32 + (srow & 31), (it - srow * (BN / 4)) * 4);
      }
    }
  }
  publish();
}

// ---------------------------------------------------------------- weight packing
__device__ __forceinline__ float pack_value(const float* __restrict__ src, int mode, int phase, int kh, int kw, int cin,
                                            int n, int nn, int cc, int ty, int tx) {
  float v = 0.f;
  if (nn < n && cc < cin) {
    if (mode == VAM_PACK_CONV) {
      v = src[(((size_t)nn * cin + cc) * kh + ty) * kw + tx];
    } else if (mode == VAM_PACK_PS2) {
      int cq = n / 4;
      int ph = nn / cq, c = nn - ph * cq;
      v = src[(((size_t)(c * 4 + ph) * cin + cc) * kh + ty) * kw + tx];
    } else if (mode == VAM_PACK_CONV_DGRAD) {
      // data gradient of a stride-1 conv = correlation of dY with the taps flipped and the channel roles
      // swapped: here `cin` = forward Cout (channels of dY), `n` = forward Cin; src is the forward OIHW tensor
      v = src[(((size_t)cc * n + nn) * kh + (kh - 1 - ty)) * kw + (kw - 1 - tx)];
    } else if (mode == VAM_PACK_GDN || mode == VAM_PACK_GDN_T) {
      float g = mode == VAM_PACK_GDN ? src[(size_t)nn * cin + cc] : src[(size_t)cc * n + nn];   // _T: gamma transposed
      const float bound = 3.814697265625e-06f;       // 2^-18 = sqrt(0 + 2^-36)
      const float ped = 1.4551915228366852e-11f;     // 2^-36
      g = fmaxf(g, bound);
      v = g * g - ped;
    } else if (mode == VAM_PACK_DECONV5S2) {
      if (phase >= 0) {
        int py = phase >> 1, px = phase & 1;
        int dy = ty - (py ? 0 : 1), dx = tx - (px ? 0 : 1);
        int ky = py + 2 - 2 * dy, kx = px + 2 - 2 * dx;
        if (ky >= 0 && ky < 5 && kx >= 0 && kx < 5)
          v = src[(((size_t)cc * n + nn) * 5 + ky) * 5 + kx];
      } else {
        int cout = n / 4;
        int ph = nn / cout, c = nn - ph * cout;
        int py = ph >> 1, px = ph & 1;
        int dy = ty - 1, dx = tx - 1;
        int ky = py + 2 - 2 * dy, kx = px + 2 - 2 * dx;
        if (ky >= 0 && ky < 5 && kx >= 0 && kx < 5)
          v = src[(((size_t)cc * cout + c) * 5 + ky) * 5 + kx];
      }
    }
  }
  return v;
}

__global__ void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int mode,
                                    int phase, int kh, int kw, int cin, int n, int npad, int bk, int kc,
                                    long total) {
  long d = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= total) return;
  int kk = (int)(d % bk);
  long r = d / bk;
  int nn = (int)(r % npad);
  r /= npad;
  int c_chunk = (int)(r % kc);
  int tap = (int)(r / kc);
  int ty = tap / kw, tx = tap % kw;
  int cc = c_chunk * bk + kk;
  dst[d] = pack_value(src, mode, phase, kh, kw, cin, n, nn, cc, ty, tx);
}

// bf16x3 layout: [tap][32-channel chunk][Npad][12 chunks][8 bf16]; chunk (p*4 + g) holds plane p (hi/mid/lo) of
// channels 8g..8g+7 of the 32-channel chunk.  The three planes sum to the fp32 weight exactly.
__device__ __forceinline__ void pack_bf3_element(long d, const float* __restrict__ src, unsigned short* __restrict__ dst, int mode,
                                                 int phase, int kh, int kw, int cin, int n, int npad, int kc32, int dual) {
  int kk = (int)(d % 32);
  long r = d / 32;
  int nn = (int)(r % npad);
  r /= npad;
  int c_chunk = (int)(r % kc32);
  int tap = (int)(r / kc32);
  int cc = c_chunk * 32 + kk;
  if (dual) {                                    // 16 input channels: chunk = tap pair, [tap 2c: 16 ch | tap 2c+1: 16 ch]
    tap = 2 * tap + (kk >> 4);
    cc = kk & 15;
  }
  int ty = tap / kw, tx = tap % kw;
  const float v = (tap < kh * kw) ? pack_value(src, mode, phase, kh, kw, cin, n, nn, cc, ty, tx) : 0.f;
  // exact split by truncation (same as the activations' in the kernel): hi + mid + lo == v
  const unsigned hb = __float_as_uint(v);
  const float r1 = v - __uint_as_float(hb & 0xFFFF0000u);
  const unsigned mb = __float_as_uint(r1);
  const unsigned lb = __float_as_uint(r1 - __uint_as_float(mb & 0xFFFF0000u));
  const size_t row = (size_t)(d / 32) * 96;                                  // 96 bf16 = 192 B per (chunk, n) row
  const int g = kk >> 3, e = kk & 7;
  dst[row + (0 * 4 + g) * 8 + e] = (unsigned short)(hb >> 16);
  dst[row + (1 * 4 + g) * 8 + e] = (unsigned short)(mb >> 16);
  dst[row + (2 * 4 + g) * 8 + e] = (unsigned short)(lb >> 16);
}

__global__ void pack_weights_bf3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int mode,
                                        int phase, int kh, int kw, int cin, int n, int npad, int kc32, long total, int dual) {
  long d = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= total) return;
  pack_bf3_element(d, src, dst, mode, phase, kh, kw, cin, n, npad, kc32, dual);
}

// fp16x2 (MODE 3): per output channel n the power of two 2^s_n that maps max |w[n, :]| into [2^14, 2^15); out[n] = 2^-s_n
// (1 for an all-zero channel).  One block per channel.
__global__ void wscale_f16x2_kernel(const float* __restrict__ src, float* __restrict__ out, int mode, int phase, int kh, int kw,
                                    int cin, int n, int npad) {
  const int nn = blockIdx.x;
  float m = 0.f;
  for (int i = threadIdx.x; i < kh * kw * cin; i += blockDim.x) {
    const int tap = i / cin, cc = i - tap * cin;
    m = fmaxf(m, fabsf(pack_value(src, mode, phase, kh, kw, cin, n, nn, cc, tap / kw, tap % kw)));
  }
  __shared__ float red[256];
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    m = red[0];
    const int e = (int)((__float_as_uint(m) >> 23) & 0xFFu) - 127;
    int sh = (m > 0.f && e > -127 && e < 128) ? 14 - e : 0;
    sh = sh > 126 ? 126 : (sh < -126 ? -126 : sh);
    out[nn] = __uint_as_float((unsigned)(127 - sh) << 23);
  }
  (void)npad;
}

// fp16x2 layout: [tap][32-channel chunk][Npad][8 chunks][8 fp16]; chunk (p*4 + g) holds plane p (h / l) of channels
// 8g..8g+7; w 2^s_n = h + l up to 2^-22 |w|.  (dual: as in the bf16x3 layout)
__global__ void pack_weights_f16x2_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, const float* __restrict__ wsc,
                                          int mode, int phase, int kh, int kw, int cin, int n, int npad, int kc32, long total, int dual) {
  long d = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= total) return;
  int kk = (int)(d % 32);
  long r = d / 32;
  int nn = (int)(r % npad);
  r /= npad;
  int c_chunk = (int)(r % kc32);
  int tap = (int)(r / kc32);
  int cc = c_chunk * 32 + kk;
  if (dual) {
    tap = 2 * tap + (kk >> 4);
    cc = kk & 15;
  }
  int ty = tap / kw, tx = tap % kw;
  const float v = (tap < kh * kw) ? pack_value(src, mode, phase, kh, kw, cin, n, nn, cc, ty, tx) : 0.f;
  const float vs = v / wsc[nn];                      // 2^-s_n is a power of two: the division is exact
  const _Float16 h = (_Float16)vs;
  const _Float16 l = (_Float16)(vs - (float)h);
  const size_t row = (size_t)(d / 32) * 64;          // 64 fp16 = 128 B per (chunk, n) row
  const int g = kk >> 3, e = kk & 7;
  dst[row + (0 * 4 + g) * 8 + e] = __builtin_bit_cast(unsigned short, h);
  dst[row + (1 * 4 + g) * 8 + e] = __builtin_bit_cast(unsigned short, l);
}

// max |x| over an NHWC window, folded into *cell (the float's bits; integer atomicMax).  float4 loads, grid-stride.
struct AbsmaxArgs { const float* ptr[VAM_MAX_SEG]; int C[VAM_MAX_SEG]; int ld[VAM_MAX_SEG]; int n_seg; long n_pix; int* cell; };
__global__ __launch_bounds__(256) void absmax_kernel(const AbsmaxArgs a) {
  float m = 0.f;
  for (int s = 0; s < a.n_seg; ++s) {
    const int c4 = a.C[s] >> 2;
    const long total = a.n_pix * c4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const long pix = i / c4;
      const int c = (int)(i - pix * c4);
      const float4 v = *reinterpret_cast<const float4*>(a.ptr[s] + pix * a.ld[s] + 4 * c);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f && __float_as_int(m) > __hip_atomic_load(a.cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(a.cell, __float_as_int(m));
}

// bf16 storage mode: [tap][32-channel chunk][Npad][32 bf16] = 64 B per row, each weight rounded to nearest-even bf16
__global__ void pack_weights_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int mode,
                                         int phase, int kh, int kw, int cin, int n, int npad, int kc32, long total) {
  long d = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= total) return;
  int kk = (int)(d % 32);
  long r = d / 32;
  int nn = (int)(r % npad);
  r /= npad;
  int c_chunk = (int)(r % kc32);
  int tap = (int)(r / kc32);
  int ty = tap / kw, tx = tap % kw;
  const float v = pack_value(src, mode, phase, kh, kw, cin, n, nn, c_chunk * 32 + kk, ty, tx);
  const __bf16 h = (__bf16)v;
  dst[d] = __builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ void pack_bias_element(int i, const float* __restrict__ src, float* __restrict__ dst, int mode, int n) {
  float v;
  if (mode == VAM_PACK_PS2) {
    int cq = n / 4;
    int ph = i / cq, c = i - ph * cq;
    v = src[c * 4 + ph];
  } else if (mode == VAM_PACK_DECONV5S2) {
    int cout = n / 4;
    v = src[i % cout];
  } else if (mode == VAM_PACK_GDN) {
    // beta: NonNegativeParametrizer(minimum=1e-6): bound = sqrt(1e-6 + 2^-36) rounded to fp32
    const float bound = (float)1.0000072759311445e-03;
    const float ped = 1.4551915228366852e-11f;
    float b = fmaxf(src[i], bound);
    v = b * b - ped;
  } else {
    v = src[i];
  }
  dst[i] = v;
}

__global__ void pack_bias_kernel(const float* __restrict__ src, float* __restrict__ dst, int mode, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pack_bias_element(i, src, dst, mode, n);
}

// Many small repacks as one launch (training: every trained layer is repacked after every optimiser step — 1,250 launches
// of ~4.5 us per first_train step when issued one by one).  blockIdx.y = job; a job's blocks stride over its elements.
struct PackJobD {
  const float* src;
  void* dst;
  long total;
  int kind, mode, phase, kh, kw, cin, n, npad, kc32, dual;
};
struct PackGroupArgs {
  PackJobD j[VAM_MAX_PACK_GROUP];
};
__global__ __launch_bounds__(256) void pack_group_kernel(const PackGroupArgs a) {
  const PackJobD& j = a.j[blockIdx.y];
  for (long d = (long)blockIdx.x * 256 + threadIdx.x; d < j.total; d += (long)gridDim.x * 256) {
    if (j.kind == 0) pack_bf3_element(d, j.src, reinterpret_cast<unsigned short*>(j.dst), j.mode, j.phase, j.kh, j.kw, j.cin, j.n, j.npad, j.kc32, j.dual);
    else pack_bias_element((int)d, j.src, reinterpret_cast<float*>(j.dst), j.mode, j.n);
  }
}

static int g_force[3] = {0, 0, 0};   // tuning hook: forced BM / BN / BK (0 = automatic)
static int g_staged = -1;             // tuning / test hook: 1 = staged epilogue everywhere, 0 = automatic, -1 = VAMPIC_EPILOGUE
static int g_last[3] = {0, 0, 0};   // tile configuration of the most recent launch (diagnostics)

static inline bool dual_tap(int cin, int taps) { return cin == 16 && taps > 1; }   // split-operand mode: see VAM_CONVI_DUAL

static inline int bk_for(int cin) { return (cin % 32 == 0) ? 32 : 16; }   // kernel K step (packing is always 16-granular)

static int g_mode = -1;              // 0 = fp32 MFMA, 1 = bf16x3 MFMA, 3 = fp16x2 MFMA (opt-in); -1 = not chosen yet (VAMPIC_CONV, default bf16x3)

static int conv_mode() {
  if (g_mode < 0) {
    const char* e = getenv("VAMPIC_CONV");
    g_mode = (e && (e[0] == 'f' || e[0] == 'F')) ? ((e[1] == '1') ? 3 : 0) : 1;      // "f32" -> 0, "f16x2" -> 3
  }
  return g_mode;
}

template <int BM, int BN, int BK, int WGM, int WGN, int MODE, int AIN = 0, int SPEC = 0>
static int launch_cfg(const GroupArgs& ga, int total_tiles, hipStream_t s) {
  constexpr size_t pipe = MODE == 2 ? (size_t)2 * (BM + BN) * 16 * sizeof(float)
                          : MODE ? (size_t)((SPEC || BM + BN <= 128) ? 2 : 1) * (BM + BN) * (MODE == 3 ? 32 : 48) * sizeof(float)
                                 : (size_t)2 * (BM + BN) * BK * sizeof(float);
  constexpr int crows = SPEC ? BM : WGM * 32;              // rows of C staged through LDS at a time
  constexpr size_t ctile = (size_t)crows * (BN + 4) * sizeof(float) + (size_t)crows * 2 * sizeof(int);
  constexpr size_t smem = pipe > ctile ? pipe : ctile;
  static_assert(smem <= 160 * 1024, "tile does not fit the CU's LDS");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, BK, WGM, WGN, MODE, AIN, SPEC>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  // grid: 8 XCD groups x the largest per-XCD share (see the kernel's tile mapping)
  int per_xcd = 0;
  for (int i = 0; i < ga.nprob; ++i) per_xcd += (ga.tile_start[i + 1] - ga.tile_start[i] + 7) / 8;
  (void)total_tiles;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WGM, WGN, MODE, AIN, SPEC>), dim3(8 * per_xcd), dim3(WGM * WGN * 64 * (SPEC ? 2 : 1)), smem, s, ga);
  return check_launch("conv_igemm_kernel");
}

}  // namespace vam

using namespace vam;

extern "C" {

int vam_conv_last_tile(int* bm, int* bn, int* bk) {
  if (bm) *bm = g_last[0];
  if (bn) *bn = g_last[1];
  if (bk) *bk = g_last[2];
  return VAM_OK;
}

int vam_conv_force_tile(int bm, int bn, int bk) {
  g_force[0] = bm; g_force[1] = bn; g_force[2] = bk;
  return VAM_OK;
}

int vam_conv_force_epilogue(int staged) {
  g_staged = staged < 0 ? -1 : (staged ? 1 : 0);
  return VAM_OK;
}

int vam_conv_set_mode(int mode) {
  if (mode != 0 && mode != 1 && mode != 3) {
    set_error("vam_conv_set_mode: mode %d (0 = fp32 MFMA, 1 = bf16x3 MFMA, 3 = fp16x2 MFMA)", mode);
    return VAM_EINVAL;
  }
  g_mode = mode;
  return VAM_OK;
}

int vam_conv_get_mode(void) { return conv_mode(); }

size_t vam_conv_wpack_floats(int kh, int kw, int cin, int n) {
  int npad = (n + 31) / 32 * 32;
  if (conv_mode() == 1) return (size_t)kh * kw * ((cin + 31) / 32) * npad * 48;   // 96 bf16 per (n, 32-channel chunk)
  if (conv_mode() == 3) return (size_t)kh * kw * ((cin + 31) / 32) * npad * 32 + npad;   // 64 fp16 per row, then 2^-s_n per channel
  int bk = PK;
  int kc = (cin + bk - 1) / bk;
  return (size_t)kh * kw * kc * npad * bk;
}

int vam_pack_conv_weights(const float* src, float* dst, int mode, int phase, int kh, int kw, int cin, int n,
                          void* stream) {
  VAM_REQUIRE(src && dst && kh > 0 && kw > 0 && cin > 0 && n > 0, "vam_pack_conv_weights: bad arguments");
  VAM_REQUIRE(mode >= VAM_PACK_CONV && mode <= VAM_PACK_GDN_T, "vam_pack_conv_weights: bad mode %d", mode);
  if (mode == VAM_PACK_PS2) VAM_REQUIRE(n % 4 == 0, "PS2 pack needs N %% 4 == 0");
  if (mode == VAM_PACK_DECONV5S2 && phase < 0) VAM_REQUIRE(n % 4 == 0 && kh == 3 && kw == 3, "merged deconv pack needs 3x3, N=4*Cout");
  if (mode == VAM_PACK_DECONV5S2 && phase >= 0)
    VAM_REQUIRE(phase < 4 && kh == ((phase >> 1) ? 2 : 3) && kw == ((phase & 1) ? 2 : 3), "deconv phase %d needs kh/kw = 3|2", phase);
  if (mode == VAM_PACK_GDN || mode == VAM_PACK_GDN_T) VAM_REQUIRE(kh == 1 && kw == 1 && cin == n, "GDN pack is 1x1 and square");
  int npad = (n + 31) / 32 * 32;
  if (conv_mode() == 3) {
    const int kc32 = (cin + 31) / 32;
    const int dual = dual_tap(cin, kh * kw) ? 1 : 0;
    const long total1 = dual ? (long)((kh * kw + 1) / 2) * npad * 32 : (long)kh * kw * kc32 * npad * 32;
    float* wsc = dst + (size_t)kh * kw * kc32 * npad * 32;
    hipLaunchKernelGGL(wscale_f16x2_kernel, dim3(npad), dim3(256), 0, (hipStream_t)stream, src, wsc, mode, phase, kh, kw, cin, n, npad);
    hipLaunchKernelGGL(pack_weights_f16x2_kernel, dim3(cdiv(total1, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<unsigned short*>(dst), wsc, mode, phase, kh, kw, cin, n, npad, kc32, total1, dual);
    return check_launch("pack_weights_f16x2_kernel");
  }
  if (conv_mode() == 1) {
    const int kc32 = (cin + 31) / 32;
    const int dual = dual_tap(cin, kh * kw) ? 1 : 0;       // 16-channel inputs: two taps per 32-channel chunk (kernel: VAM_CONVI_DUAL)
    const long total1 = dual ? (long)((kh * kw + 1) / 2) * npad * 32 : (long)kh * kw * kc32 * npad * 32;
    hipLaunchKernelGGL(pack_weights_bf3_kernel, dim3(cdiv(total1, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<unsigned short*>(dst), mode, phase, kh, kw, cin, n, npad, kc32, total1, dual);
    return check_launch("pack_weights_bf3_kernel");
  }
  int bk = PK;
  int kc = (cin + bk - 1) / bk;
  long total = (long)kh * kw * kc * npad * bk;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, mode,
                     phase, kh, kw, cin, n, npad, bk, kc, total);
  return check_launch("pack_weights_kernel");
}

size_t vam_conv_wpack_bf16_bytes(int kh, int kw, int cin, int n) {
  return (size_t)kh * kw * ((cin + 31) / 32) * ((n + 31) / 32 * 32) * 64;
}

int vam_pack_conv_weights_bf16(const float* src, void* dst, int mode, int phase, int kh, int kw, int cin, int n, void* stream) {
  VAM_REQUIRE(src && dst && kh > 0 && kw > 0 && cin > 0 && n > 0, "vam_pack_conv_weights_bf16: bad arguments");
  VAM_REQUIRE(mode >= VAM_PACK_CONV && mode <= VAM_PACK_GDN_T, "vam_pack_conv_weights_bf16: bad mode %d", mode);
  if (mode == VAM_PACK_PS2) VAM_REQUIRE(n % 4 == 0, "PS2 pack needs N %% 4 == 0");
  if (mode == VAM_PACK_DECONV5S2 && phase < 0) VAM_REQUIRE(n % 4 == 0 && kh == 3 && kw == 3, "merged deconv pack needs 3x3, N=4*Cout");
  if (mode == VAM_PACK_DECONV5S2 && phase >= 0)
    VAM_REQUIRE(phase < 4 && kh == ((phase >> 1) ? 2 : 3) && kw == ((phase & 1) ? 2 : 3), "deconv phase %d needs kh/kw = 3|2", phase);
  const int npad = (n + 31) / 32 * 32, kc32 = (cin + 31) / 32;
  const long total = (long)kh * kw * kc32 * npad * 32;
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     reinterpret_cast<unsigned short*>(dst), mode, phase, kh, kw, cin, n, npad, kc32, total);
  return check_launch("pack_weights_bf16_kernel");
}

int vam_pack_bias(const float* src, float* dst, int mode, int n, void* stream) {
  VAM_REQUIRE(src && dst && n > 0, "vam_pack_bias: bad arguments");
  hipLaunchKernelGGL(pack_bias_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, mode, n);
  return check_launch("pack_bias_kernel");
}

int vam_conv_group(const vam_conv* probs, int nprob, void* stream) {
  VAM_REQUIRE(probs && nprob >= 1 && nprob <= VAM_MAX_GROUP, "vam_conv_group: 1..%d problems", VAM_MAX_GROUP);
  GroupArgs ga;
  ga.nprob = nprob;
  {
    static int staged_env = -1;
    if (staged_env < 0) {
      const char* e = getenv("VAMPIC_EPILOGUE");
      staged_env = (e && e[0] == 's') ? 1 : 0;
    }
    ga.staged_epilogue = g_staged >= 0 ? g_staged : staged_env;
  }
  bool in_p3 = false, w16 = false, in16 = false;
  int bk = 0;
  long max_p = 0;
  int max_n = 0;
  double flops = 0, bytes = 0;
  for (int i = 0; i < nprob; ++i) {
    const vam_conv& c = probs[i];
    ConvP& p = ga.p[i];
    VAM_REQUIRE(c.n_seg >= 1 && c.n_seg <= VAM_MAX_SEG, "conv[%d]: n_seg %d", i, c.n_seg);
    VAM_REQUIRE(c.B > 0 && c.H > 0 && c.W > 0 && c.Ho > 0 && c.Wo > 0 && c.N > 0, "conv[%d]: bad extent", i);
    VAM_REQUIRE(c.kh >= 1 && c.kh <= 5 && c.kw >= 1 && c.kw <= 5 && (c.stride == 1 || c.stride == 2), "conv[%d]: kernel %dx%d stride %d", i, c.kh, c.kw, c.stride);
    VAM_REQUIRE(c.wpack && c.out, "conv[%d]: null weights/output", i);
    int cin = 0;
    const bool p3_in = (c.flags & VAM_CONV_IN_BF3) != 0, p3_out = (c.flags & VAM_CONV_OUT_BF3) != 0;
    const bool c_w16 = (c.flags & VAM_CONV_W_BF16) != 0, c_in16 = (c.flags & VAM_CONV_IN_BF16) != 0;
    const bool c_out16 = (c.flags & VAM_CONV_OUT_BF16) != 0, c_aux16 = (c.flags & VAM_CONV_AUX_BF16) != 0;
    if (i == 0) { w16 = c_w16; in16 = c_in16; }
    VAM_REQUIRE(c_w16 == w16 && c_in16 == in16, "conv group mixes bf16-storage and fp32 problems");
    VAM_REQUIRE(!c_in16 || c_w16, "conv[%d]: bf16 input needs bf16-packed weights (VAM_CONV_W_BF16)", i);
    if (c_w16) VAM_REQUIRE(!p3_in && !p3_out, "conv[%d]: bf16 storage and bf16x3 planes do not combine", i);
    if (c_out16) VAM_REQUIRE(!(c.flags & (VAM_CONV_PS2 | VAM_CONV_OUT_NCHW)) && c.ldo % 4 == 0 && (((uintptr_t)c.out) & 7) == 0, "conv[%d]: bf16 output needs plain NHWC placement, ldo %% 4 == 0", i);
    if (c_aux16) VAM_REQUIRE(!(c.flags & (VAM_CONV_OUT_NCHW)) && (!(c.flags & VAM_CONV_PS2) || (c.Cq % 4) == 0), "conv[%d]: bf16 epilogue operands need the vector epilogue", i);
    if (i == 0) in_p3 = p3_in;
    VAM_REQUIRE(p3_in == in_p3, "conv group mixes fp32 and bf16x3-plane inputs");
    if (conv_mode() == 3 && !c_w16)
      for (int sg = 0; sg < c.n_seg; ++sg)
        VAM_REQUIRE(c.in_amax[sg] != nullptr, "conv[%d]: the fp16x2 mode needs in_amax[%d] (max |x| of that input segment: vam_absmax, or its producer's out_amax)", i, sg);
    if (p3_in || p3_out) VAM_REQUIRE(conv_mode() == 1, "conv[%d]: bf16x3-plane tensors need the split-operand mode", i);
    if (p3_in) VAM_REQUIRE(!(c.flags & VAM_CONV_SQUARE_IN), "conv[%d]: SQUARE_IN needs fp32 input", i);
    if (p3_out) VAM_REQUIRE(!(c.flags & (VAM_CONV_PS2 | VAM_CONV_OUT_NCHW)) && c.N % 8 == 0 && c.ldo * 8 >= c.N, "conv[%d]: bf16x3-plane output needs plain NHWC placement, N %% 8 == 0 and ldo (groups) >= N/8", i);
    for (int s = 0; s < VAM_MAX_SEG; ++s) {
      if (s < c.n_seg) {
        if (c_in16) {     // ld counts bf16 elements
          VAM_REQUIRE(c.seg[s].ptr && c.seg[s].C > 0 && c.seg[s].ld >= c.seg[s].C && c.seg[s].ld % 8 == 0 && c.seg[s].C % 8 == 0, "conv[%d]: bf16 segment %d invalid", i, s);
          VAM_REQUIRE((((uintptr_t)c.seg[s].ptr) % 16) == 0, "conv[%d]: segment %d not 16-byte aligned", i, s);
        } else if (p3_in) {      // ld counts 8-channel groups (48 bytes each)
          VAM_REQUIRE(c.seg[s].ptr && c.seg[s].C > 0 && c.seg[s].C % 8 == 0 && c.seg[s].ld * 8 >= c.seg[s].C, "conv[%d]: bf16x3 segment %d invalid", i, s);
          VAM_REQUIRE((((uintptr_t)c.seg[s].ptr) % 16) == 0, "conv[%d]: segment %d not 16-byte aligned", i, s);
        } else {
        VAM_REQUIRE(c.seg[s].ptr && c.seg[s].C > 0 && c.seg[s].ld >= c.seg[s].C, "conv[%d]: segment %d invalid", i, s);
        VAM_REQUIRE((c.seg[s].ld % 4) == 0 && (((uintptr_t)c.seg[s].ptr) % 16) == 0, "conv[%d]: segment %d not 16-byte aligned", i, s);
        }
        cin += c.seg[s].C;
        p.seg_ptr[s] = c.seg[s].ptr;
        p.seg_ld[s] = c.seg[s].ld;
        p.seg_end[s] = cin;
      } else {
        p.seg_ptr[s] = nullptr;
        p.seg_ld[s] = 0;
        p.seg_end[s] = 1 << 30;
      }
    }
    const int mode1 = conv_mode() == 1 || conv_mode() == 3 || c_w16;
    int pbk = mode1 ? 32 : bk_for(cin);
    VAM_REQUIRE(cin % 16 == 0, "conv[%d]: Cin %d not a multiple of 16", i, cin);
    for (int sgi = 0; sgi < c.n_seg; ++sgi)
      VAM_REQUIRE((double)c.B * c.H * c.W * c.seg[sgi].ld * (p3_in ? 48.0 : (c_in16 ? 2.0 : 4.0)) < 2147000000.0, "conv[%d]: input window larger than 2 GiB (32-bit buffer offsets)", i);
    VAM_REQUIRE((double)vam_conv_wpack_floats(c.kh, c.kw, cin, c.N) * 4.0 < 2147000000.0, "conv[%d]: packed weights larger than 2 GiB", i);
    for (int s = 0; s + 1 < c.n_seg; ++s)
      VAM_REQUIRE(p.seg_end[s] % pbk == 0, "conv[%d]: segment boundary %d not a multiple of BK=%d", i, p.seg_end[s], pbk);
    if (i == 0) bk = pbk;
    VAM_REQUIRE(pbk == bk, "conv group mixes BK=%d and BK=%d problems", bk, pbk);
    // the last input position touched must be consistent with the declared geometry
    VAM_REQUIRE((c.Ho - 1) * c.stride - c.pad_y <= c.H - 1 + c.kh && (c.Wo - 1) * c.stride - c.pad_x <= c.W - 1 + c.kw, "conv[%d]: output grid larger than the input allows", i);
    if (c.flags & VAM_CONV_PS2) {
      VAM_REQUIRE(c.Cq > 0 && c.N == 4 * c.Cq && c.Hf == 2 * c.Ho && c.Wf == 2 * c.Wo, "conv[%d]: PS2 geometry", i);
    } else {
      VAM_REQUIRE(c.osy >= 1 && c.osx >= 1 && c.ooy >= 0 && c.oox >= 0 && (c.Ho - 1) * c.osy + c.ooy < c.Hf && (c.Wo - 1) * c.osx + c.oox < c.Wf, "conv[%d]: output placement outside Hf x Wf", i);
    }
    VAM_REQUIRE(c.N % 4 == 0, "conv[%d]: N %d not a multiple of 4", i, c.N);
    {
      const bool vec = !(c.flags & VAM_CONV_OUT_NCHW) && (!(c.flags & VAM_CONV_PS2) || (c.Cq % 4) == 0);
      auto al = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
      if (vec) {
        auto al8 = [](const void* q) { return (((uintptr_t)q) & 7) == 0; };
        VAM_REQUIRE((p3_out || c.ldo % 4 == 0) && (c_out16 ? al8(c.out) : al(c.out)) && al(c.bias), "conv[%d]: output/bias not 16-byte aligned", i);
        auto aok = [&](const vam_aux& a) { return !a.ptr || (a.ld % 4 == 0 && (c_aux16 ? al8(a.ptr) : al(a.ptr))); };
        VAM_REQUIRE(aok(c.pre) && aok(c.mul) && aok(c.post) && aok(c.post2), "conv[%d]: epilogue operand not aligned", i);
      }
    }
    if (!(c.flags & VAM_CONV_OUT_NCHW) && !p3_out) VAM_REQUIRE(c.ldo >= ((c.flags & VAM_CONV_PS2) ? c.Cq : c.N), "conv[%d]: ldo %d < channels", i, c.ldo);
    p.n_seg = c.n_seg;
    p.H = c.H; p.W = c.W; p.HW = c.H * c.W;
    p.kh = c.kh; p.kw = c.kw; p.stride = c.stride; p.pad_y = c.pad_y; p.pad_x = c.pad_x;
    p.Ho = c.Ho; p.Wo = c.Wo; p.HoWo = c.Ho * c.Wo;
    long P = (long)c.B * c.Ho * c.Wo;
    VAM_REQUIRE(P < (1L << 30) && (long)c.B * c.H * c.W < (1L << 30), "conv[%d]: too many pixels", i);
    p.P = (int)P;
    p.N = c.N; p.Npad = (c.N + 31) / 32 * 32;
    p.Cin = cin; p.Kc = (cin + pbk - 1) / pbk; p.Kc16 = (cin + PK - 1) / PK;
    p.wpack = c.wpack; p.bias = c.bias; p.out = c.out;
    p.ldo = c.ldo; p.Hf = c.Hf; p.Wf = c.Wf; p.osy = c.osy; p.osx = c.osx; p.ooy = c.ooy; p.oox = c.oox;
    constexpr int PUBLIC_FLAGS = VAM_CONV_SQUARE_IN | VAM_CONV_PS2 | VAM_CONV_OUT_NCHW | VAM_CONV_IN_BF3 | VAM_CONV_OUT_BF3 |
                                 VAM_CONV_W_BF16 | VAM_CONV_IN_BF16 | VAM_CONV_OUT_BF16 | VAM_CONV_AUX_BF16 | VAM_CONV_MUL_GELU_GRAD;
    VAM_REQUIRE((c.flags & ~PUBLIC_FLAGS) == 0, "conv[%d]: unknown flag bits 0x%x", i, c.flags & ~PUBLIC_FLAGS);
    p.Cq = c.Cq; p.act = c.act; p.flags = c.flags & PUBLIC_FLAGS;      // bits 29 / 30 are the kernel's own
    // 16-channel multi-tap problems: vam_pack_conv_weights lays the weights out two taps per 32-channel chunk in the
    // split-operand mode (dual_tap), and ONLY the dual-tap indexing reads that layout — so every such problem must be one
    // the dual path takes (one fp32 segment); anything else would read dual-packed weights with plain indexing
    if ((conv_mode() == 1 || conv_mode() == 3) && !c_w16 && dual_tap(cin, c.kh * c.kw)) {
      VAM_REQUIRE(!p3_in && c.n_seg == 1, "conv[%d]: a 16-channel multi-tap problem takes one fp32 input segment (its weights are "
                  "packed two taps per chunk)", i);
      p.flags |= VAM_CONVI_DUAL;
    }
    {
      // the kernel addresses every input segment with 32-bit byte offsets inside a 2^31-byte window
      const double in_pix = (double)c.B * c.H * c.W;
      for (int s = 0; s < c.n_seg; ++s)
        VAM_REQUIRE(in_pix * c.seg[s].ld * (p3_in ? 48.0 : c_in16 ? 2.0 : 4.0) < 2147483648.0, "conv[%d]: input segment %d spans 2 GiB or more (split the batch)", i, s);
      // the direct epilogue does the same for the output and the epilogue operands, with 24-bit pixel arithmetic;
      // anything larger takes the staged epilogue (64-bit addresses)
      const double out_pix = (double)c.B * c.Hf * c.Wf;
      auto fits = [&](const void* q, int ld) { return !q || (ld < (1 << 24) && out_pix * ld * 4.0 < 2147483648.0); };
      if (!(out_pix < (double)(1 << 22) && fits(c.out, c.ldo) && fits(c.pre.ptr, c.pre.ld) && fits(c.mul.ptr, c.mul.ld) &&
            fits(c.post.ptr, c.post.ld) && fits(c.post2.ptr, c.post2.ld) && fits(c.preact.ptr, c.preact.ld)))
        p.flags |= VAM_CONVI_STAGED;
    }
    if (c.preact.ptr || (c.flags & VAM_CONV_MUL_GELU_GRAD)) p.flags |= VAM_CONVI_STAGED;   // features of the staged epilogue only
    VAM_REQUIRE(!(c.flags & VAM_CONV_MUL_GELU_GRAD) || (c.mul.ptr && !(c.flags & VAM_CONV_AUX_BF16)),
                "conv[%d]: VAM_CONV_MUL_GELU_GRAD needs an fp32 mul operand (the GELU's pre-activation)", i);
    VAM_REQUIRE(!c.preact.ptr || (!(c.flags & (VAM_CONV_OUT_NCHW | VAM_CONV_W_BF16)) && c.preact.ld % 4 == 0 &&
                                  (((uintptr_t)c.preact.ptr) & 15) == 0 && c.preact.ld >= ((c.flags & VAM_CONV_PS2) ? c.Cq : c.N)),
                "conv[%d]: preact is an fp32 NHWC tensor indexed like the output (16-byte aligned, ld %% 4 == 0, ld >= channels)", i);
    p.preact = const_cast<float*>(c.preact.ptr); p.ld_preact = c.preact.ld;
    p.pre = c.pre.ptr; p.ld_pre = c.pre.ld;
    p.mul = c.mul.ptr; p.ld_mul = c.mul.ld;
    p.post = c.post.ptr; p.ld_post = c.post.ld;
    p.post2 = c.post2.ptr; p.ld_post2 = c.post2.ld;
    for (int sg = 0; sg < VAM_MAX_SEG; ++sg) p.in_amax[sg] = sg < c.n_seg ? c.in_amax[sg] : nullptr;
    p.out_amax = c.out_amax;
    if (P > max_p) max_p = P;
    if (c.N > max_n) max_n = c.N;
    flops += 2.0 * (double)P * c.N * cin * c.kh * c.kw;
    bytes += 4.0 * ((double)c.B * c.H * c.W * cin + (double)P * c.N + (double)c.kh * c.kw * cin * c.N);
  }
  // ---- tile choice: one configuration per launch, by a cost score fitted to a sweep of all
  // (BM,BN,BK) over the model's layer shapes on MI355X (scratch/tune.py; DESIGN.md "Tile choice"):
  //   score = padded work / real work  x  block-count penalty (fewer than ~4 blocks per CU leaves
  //   barriers and load latency exposed)  x  tile-shape factor  x  thin-K factor (1x1 layers are
  //   load-bound: narrow tiles, more blocks in flight).
  struct Cand { int bm, bn; double shape; };
  static const Cand cands[14] = {{128, 192, 1.0}, {128, 224, 1.04}, {128, 160, 1.04}, {128, 128, 1.06}, {128, 96, 1.04},
                                 {128, 64, 1.10}, {128, 32, 1.35}, {64, 192, 1.05}, {64, 128, 1.04}, {64, 64, 1.05},
                                 {64, 96, 1.12}, {64, 160, 1.35}, {64, 224, 1.6}, {64, 32, 1.5}};
  int ktot_max = 0;
  for (int i = 0; i < nprob; ++i) {
    int kt = ga.p[i].Cin * ga.p[i].kh * ga.p[i].kw;
    if (kt > ktot_max) ktot_max = kt;
  }
  // bf16x3 mode: nine configurations (the wave tile needs 12 operand registers per 32 rows / columns and K step), own fit,
  // plus the wave-specialised 128x224 tile (below)
  static const Cand cands1[10] = {{128, 192, 1.0}, {128, 128, 0.974}, {128, 96, 1.021}, {128, 64, 1.035}, {128, 32, 1.35},
                                  {64, 192, 1.05}, {64, 128, 1.019}, {64, 64, 1.05}, {64, 32, 1.363}, {128, 224, 1.0}};
  const bool m1 = conv_mode() == 1 || conv_mode() == 3 || w16;
  const Cand* cand = m1 ? cands1 : cands;
  const int n_cand = m1 ? 10 : 14;
  // deep-K launches of the split-operand mode may use the wave-specialised 128x224 tile (4x1 consumer waves of 32 x 224:
  // the layers with 224 output channels without padding them to 256); it exists as a specialised block only
  int min_chunks32 = 1 << 30;
  for (int i = 0; i < nprob; ++i) {
    const int nc = ga.p[i].kh * ga.p[i].kw * ((ga.p[i].Cin + 31) / 32);
    if (nc < min_chunks32) min_chunks32 = nc;
  }
  static int spec_env0 = -2;
  if (spec_env0 == -2) {
    const char* e = getenv("VAMPIC_SPEC");
    spec_env0 = e ? (e[0] == '1' ? 1 : 0) : -1;
  }
  const bool wide224_ok = (conv_mode() == 1 || conv_mode() == 3) && !w16 && spec_env0 != 0 && min_chunks32 >= 16;
  const double b512 = m1 ? 1.018 : 1.04, b256 = m1 ? 1.097 : 1.10, k96 = m1 ? 1.084 : 1.1;
  int bm = 128, best_bn = 128;
  double best_score = -1.0;
  for (int c = 0; c < n_cand; ++c) {
    if (m1 && cand[c].bn == 224 && !wide224_ok) continue;
    double padded = 0.0, real = 0.0;
    long blocks = 0;
    for (int i = 0; i < nprob; ++i) {
      long tm_ = cdiv(ga.p[i].P, cand[c].bm), tn_ = cdiv(ga.p[i].Npad, cand[c].bn);
      blocks += tm_ * tn_;
      padded += (double)tm_ * cand[c].bm * tn_ * cand[c].bn;
      real += (double)ga.p[i].P * ga.p[i].N;
    }
    double bp = blocks >= 1024 ? 1.0 : blocks >= 512 ? b512 : blocks >= 256 ? b256 : b256 * 256.0 / (double)blocks;
    // the widest tiles run two blocks per CU: below 512 blocks part of the chip holds a single
    // 4-wave block per CU and the launch runs at that block's pace
    if (cand[c].bm == 128 && cand[c].bn >= 160 && blocks < 512) bp *= 1.18;
    double kp = 1.0;
    if (ktot_max <= 256 && cand[c].bn > 96) kp = 1.3;
    if (ktot_max <= 256 && cand[c].bn == 96) kp = k96;
    double sc = padded / real * bp * cand[c].shape * kp;
    if (best_score < 0 || sc < best_score) { best_score = sc; bm = cand[c].bm; best_bn = cand[c].bn; }
  }
  if (g_force[0] == 64 || g_force[0] == 128) bm = g_force[0];
  if (g_force[1] > 0) best_bn = g_force[1];
  // K step 32 halves the barriers per FLOP; the sweep prefers 16 for the 128x192 tile (LDS for
  // three resident blocks per CU) and for thin-K layers.
  // (BN=224 and the two-wave 64x160 tile would spill at BK=32 with two register stages.)
  if (bk == 32 && (g_force[2] == 16 || best_bn == 224 || (bm == 64 && best_bn == 160) ||
                   (g_force[2] != 32 && ((bm == 128 && (best_bn == 192 || best_bn == 96)) || ktot_max <= 256)))) {
    bk = 16;
    for (int i = 0; i < nprob; ++i) ga.p[i].Kc = ga.p[i].Kc16;
  }
  if (conv_mode() == 1 || conv_mode() == 3 || w16) {
    // bf16x3 path: K step is always 32; configurations whose operand fragments would not fit the register file
    // (seven 32-column groups per wave) fall back to their two-tile neighbours
    bk = 32;
    for (int i = 0; i < nprob; ++i) ga.p[i].Kc = (ga.p[i].Cin + 31) / 32;
    if (best_bn == 224 && !(bm == 128 && wide224_ok)) best_bn = 128;
    if (best_bn == 160) best_bn = (bm == 128) ? 96 : 64;
    if (bm == 64 && best_bn == 96) best_bn = 64;
  }
  g_last[0] = bm; g_last[1] = best_bn; g_last[2] = bk;
  int total = 0;
  for (int i = 0; i < nprob; ++i) {
    ga.tile_start[i] = total;
    ga.p[i].tiles_n = cdiv(ga.p[i].Npad, best_bn);
    total += cdiv(ga.p[i].P, bm) * ga.p[i].tiles_n;
  }
  for (int i = nprob; i <= VAM_MAX_GROUP; ++i) ga.tile_start[i] = total;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(VAM_FAM_CONV, s, flops, bytes);
  if (w16) {          // bf16 storage mode (MODE 2): one MFMA per block, double-buffered LDS
#define VAM_CFG2(BM_, BN_, WGM_, WGN_) \
    if (bm == BM_ && best_bn == BN_) return in16 ? launch_cfg<BM_, BN_, 32, WGM_, WGN_, 2, 1>(ga, total, s) \
                                                 : launch_cfg<BM_, BN_, 32, WGM_, WGN_, 2, 0>(ga, total, s);
    VAM_CFG2(128, 32, 4, 1) VAM_CFG2(128, 64, 2, 2) VAM_CFG2(128, 96, 4, 1) VAM_CFG2(128, 128, 2, 2) VAM_CFG2(128, 192, 2, 2)
    VAM_CFG2(64, 32, 2, 1) VAM_CFG2(64, 64, 2, 2) VAM_CFG2(64, 128, 2, 2) VAM_CFG2(64, 192, 2, 2)
#undef VAM_CFG2
    set_error("vam_conv_group: no bf16 kernel configuration for BM=%d BN=%d", bm, best_bn);
    return VAM_EINVAL;
  }
  if (conv_mode() == 1 || conv_mode() == 3) {
    const bool f16 = conv_mode() == 3;
    // Wave-specialised blocks where they measured faster (interleaved A/B on one box, scratch/ab_spec.sh): the small
    // tiles of the slice chain (64x32 / 64x64 / 64x128: +12...+32 %) and the large tiles of deep-K layers (128x128,
    // 128x192: +3...+7 %); short-K layers (1x1, the 16-channel first layer) and the 4x1-wave 128x96 tile lose 15-30 %
    // (a longer prologue, one block per CU) and keep the one-role kernel.  VAMPIC_SPEC=0 / 1 forces one kind everywhere.
    static int spec_env = -2;
    if (spec_env == -2) {
      const char* e = getenv("VAMPIC_SPEC");
      spec_env = e ? (e[0] == '1' ? 1 : 0) : -1;
    }
    int min_chunks = 1 << 30;
    for (int i = 0; i < nprob; ++i) {
      const int nc = ga.p[i].kh * ga.p[i].kw * ga.p[i].Kc;
      if (nc < min_chunks) min_chunks = nc;
    }
    // (128x128: the one-role block at three blocks per CU — 168 registers, launch bounds — measured 3.5 % faster than the
    // specialised one, which is alone on its CU with 96 KB of LDS: 474 vs 491 us on 8x[320->224])
    const bool spec_tile = (bm == 64 && (best_bn == 32 || best_bn == 64 || best_bn == 128)) || (bm == 128 && best_bn == 192);
    const bool spec = (bm == 128 && best_bn == 224) || (spec_env >= 0 ? (spec_env == 1) : (spec_tile && min_chunks >= 16));
#define VAM_CFG1(BM_, BN_, WGM_, WGN_) \
    if (bm == BM_ && best_bn == BN_) {                                                                      \
      if (f16) return spec ? launch_cfg<BM_, BN_, 32, WGM_, WGN_, 3, 0, 1>(ga, total, s)                    \
                           : launch_cfg<BM_, BN_, 32, WGM_, WGN_, 3, 0, 0>(ga, total, s);                   \
      if (spec) return in_p3 ? launch_cfg<BM_, BN_, 32, WGM_, WGN_, 1, 1, 1>(ga, total, s)                  \
                             : launch_cfg<BM_, BN_, 32, WGM_, WGN_, 1, 0, 1>(ga, total, s);                 \
      return in_p3 ? launch_cfg<BM_, BN_, 32, WGM_, WGN_, 1, 1, 0>(ga, total, s)                            \
                   : launch_cfg<BM_, BN_, 32, WGM_, WGN_, 1, 0, 0>(ga, total, s);                           \
    }
    VAM_CFG1(128, 32, 4, 1) VAM_CFG1(128, 64, 2, 2) VAM_CFG1(128, 96, 4, 1) VAM_CFG1(128, 128, 2, 2) VAM_CFG1(128, 192, 2, 2)
    VAM_CFG1(64, 32, 2, 1) VAM_CFG1(64, 64, 2, 2) VAM_CFG1(64, 128, 2, 2) VAM_CFG1(64, 192, 2, 2)
#undef VAM_CFG1
    if (bm == 128 && best_bn == 224 && f16) return launch_cfg<128, 224, 32, 4, 1, 3, 0, 1>(ga, total, s);
    if (bm == 128 && best_bn == 224)
      return in_p3 ? launch_cfg<128, 224, 32, 4, 1, 1, 1, 1>(ga, total, s) : launch_cfg<128, 224, 32, 4, 1, 1, 0, 1>(ga, total, s);
    set_error("vam_conv_group: no bf16x3 kernel configuration for BM=%d BN=%d", bm, best_bn);
    return VAM_EINVAL;
  }
#define VAM_CFG(BM_, BN_, WGM_, WGN_)                                                        \
  if (bm == BM_ && best_bn == BN_)                                                           \
    return bk == 32 ? launch_cfg<BM_, BN_, 32, WGM_, WGN_, 0>(ga, total, s) : launch_cfg<BM_, BN_, 16, WGM_, WGN_, 0>(ga, total, s);
  VAM_CFG(128, 32, 4, 1) VAM_CFG(128, 64, 2, 2) VAM_CFG(128, 96, 4, 1) VAM_CFG(128, 128, 2, 2)
  VAM_CFG(128, 160, 4, 1) VAM_CFG(128, 192, 2, 2) VAM_CFG(128, 224, 4, 1)
  VAM_CFG(64, 32, 2, 1) VAM_CFG(64, 64, 2, 2) VAM_CFG(64, 96, 2, 1) VAM_CFG(64, 128, 2, 2)
  VAM_CFG(64, 160, 2, 1) VAM_CFG(64, 192, 2, 2) VAM_CFG(64, 224, 2, 1)
#undef VAM_CFG
  set_error("vam_conv_group: no kernel configuration for BM=%d BN=%d", bm, best_bn);
  return VAM_EINVAL;
}

int vam_pack_group(const vam_pack_job* jobs, int n_jobs, void* stream) {
  VAM_REQUIRE(jobs && n_jobs >= 1, "vam_pack_group: no jobs");
  if (conv_mode() != 1) {                      // other arithmetic modes: one by one through the ordinary entry points
    for (int i = 0; i < n_jobs; ++i) {
      const vam_pack_job& q = jobs[i];
      int rc = q.bias ? vam_pack_bias(q.src, (float*)q.dst, q.mode, q.n, stream)
                      : vam_pack_conv_weights(q.src, (float*)q.dst, q.mode, q.phase, q.kh, q.kw, q.cin, q.n, stream);
      if (rc != VAM_OK) return rc;
    }
    return VAM_OK;
  }
  for (int i0 = 0; i0 < n_jobs; i0 += VAM_MAX_PACK_GROUP) {
    const int cnt = n_jobs - i0 < VAM_MAX_PACK_GROUP ? n_jobs - i0 : VAM_MAX_PACK_GROUP;
    PackGroupArgs a;
    long max_total = 0;
    for (int i = 0; i < cnt; ++i) {
      const vam_pack_job& q = jobs[i0 + i];
      PackJobD& j = a.j[i];
      VAM_REQUIRE(q.src && q.dst && q.n > 0, "vam_pack_group: job %d: bad arguments", i0 + i);
      j.src = q.src; j.dst = q.dst; j.mode = q.mode; j.phase = q.phase; j.kh = q.kh; j.kw = q.kw; j.cin = q.cin; j.n = q.n;
      if (q.bias) {
        j.kind = 1; j.total = q.n; j.npad = j.kc32 = j.dual = 0;
      } else {
        VAM_REQUIRE(q.kh > 0 && q.kw > 0 && q.cin > 0 && q.mode >= VAM_PACK_CONV && q.mode <= VAM_PACK_GDN_T, "vam_pack_group: job %d: bad geometry / mode", i0 + i);
        if (q.mode == VAM_PACK_PS2) VAM_REQUIRE(q.n % 4 == 0, "PS2 pack needs N %% 4 == 0");
        if (q.mode == VAM_PACK_DECONV5S2 && q.phase < 0) VAM_REQUIRE(q.n % 4 == 0 && q.kh == 3 && q.kw == 3, "merged deconv pack needs 3x3, N=4*Cout");
        if (q.mode == VAM_PACK_DECONV5S2 && q.phase >= 0)
          VAM_REQUIRE(q.phase < 4 && q.kh == ((q.phase >> 1) ? 2 : 3) && q.kw == ((q.phase & 1) ? 2 : 3), "deconv phase %d needs kh/kw = 3|2", q.phase);
        if (q.mode == VAM_PACK_GDN || q.mode == VAM_PACK_GDN_T) VAM_REQUIRE(q.kh == 1 && q.kw == 1 && q.cin == q.n, "GDN pack is 1x1 and square");
        j.kind = 0;
        j.npad = (q.n + 31) / 32 * 32;
        j.kc32 = (q.cin + 31) / 32;
        j.dual = dual_tap(q.cin, q.kh * q.kw) ? 1 : 0;
        j.total = j.dual ? (long)((q.kh * q.kw + 1) / 2) * j.npad * 32 : (long)q.kh * q.kw * j.kc32 * j.npad * 32;
      }
      if (j.total > max_total) max_total = j.total;
    }
    long bx = cdiv(max_total, 256 * 4);          // ~4 elements per thread of the largest job
    if (bx > 512) bx = 512;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(pack_group_kernel, dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0, (hipStream_t)stream, a);
    if (int rc = check_launch("pack_group_kernel")) return rc;
  }
  return VAM_OK;
}

int vam_absmax(const vam_seg* segs, int n_seg, long n_pix, int32_t* cell, void* stream) {
  VAM_REQUIRE(segs && n_seg >= 1 && n_seg <= VAM_MAX_SEG && n_pix > 0 && cell, "vam_absmax: bad arguments");
  AbsmaxArgs a;
  double work = 0;
  for (int s = 0; s < n_seg; ++s) {
    VAM_REQUIRE(segs[s].ptr && segs[s].C > 0 && segs[s].C % 4 == 0 && segs[s].ld % 4 == 0 && (((uintptr_t)segs[s].ptr) & 15) == 0,
                "vam_absmax: segment %d needs C, ld multiples of 4 and a 16-byte aligned pointer", s);
    a.ptr[s] = segs[s].ptr; a.C[s] = segs[s].C; a.ld[s] = segs[s].ld;
    work += (double)n_pix * segs[s].C / 4;
  }
  a.n_seg = n_seg; a.n_pix = n_pix; a.cell = cell;
  long blocks = (long)(work / 256 / 8) + 1;            // ~8 float4 per thread
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("absmax_kernel");
}

}  // extern "C"
