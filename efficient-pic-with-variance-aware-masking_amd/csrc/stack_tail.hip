// The last two layers of a slice stack — conv3x3(128 -> 64) + GELU, conv3x3(64 -> 32) [+ the LRP tail's epilogue] — as ONE launch
// (reference: the Sequential tails of cc_mean_transforms / cc_scale_transforms / lrp_transforms, models/pic.py:86-121; DESIGN
// section 10, profiles/r04_stack_tail_prototype.txt: 31.1 us against 48.6 us for the two conv_igemm launches on 2 x 32 images).
//
//   * input: the 128-channel activation as bf16x3 planes (ops.View3: [pixel][8-channel group][plane][8 bf16]); weights and bias as
//     conv_igemm_kernel reads them ([tap][32-channel chunk][n][plane 3][group 4][8 bf16], 192 B per (n, chunk)); output fp32 NHWC;
//   * a workgroup (6 waves) owns 4 output rows x 16 columns of one image: layer 4 on the 6 x 16 pixels those need (rows outside
//     the image are zero = the next layer's padding), kept in LDS as planes after bias + GELU + the exact three-term split (what
//     the two-launch path writes with VAM_CONV_OUT_BF3); layer 5 from LDS;
//   * conv_igemm_kernel's canonical K order (32-channel chunk outer, tap, two 16-channel steps) and six-product order, MFMA with
//     A = weights, B = pixels (resunit.hip's orientation): the results are the two launches' bit for bit
//     (tests/test_gpu_ops.py::test_fused_stack_tail_is_bit_identical);
//   * a weight slab = the three taps of one kernel row of one chunk (36 KB), two buffers, register-staged two slabs ahead: 18
//     barrier-separated steps; the input chunk (8 x 18 halo pixels x 192 B, zero columns left and right) single-buffered per chunk.
// Latents 16 columns wide only (256-pixel-wide images); everything else takes the two launches.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace vam {

constexpr int C_IN = 128, C_MID = 64, C_OUT = 32, WW = 16, TR = 4;
constexpr int HALO_W = WW + 2;                      // zero column left and right
constexpr int IN_ROWS = TR + 4, MID_ROWS = TR + 2;  // input rows r0-2 .. r0+TR+1, layer-4 rows r0-1 .. r0+TR
constexpr int IN_PIX = IN_ROWS * HALO_W;            // 144
constexpr int MID_PIX = MID_ROWS * HALO_W;          // 108 (columns 0 and 17 stay zero)
constexpr int NPB4 = (TR + 2) / 2, NPB5 = TR / 2;   // 32-pixel blocks (two image rows) of layer 4 / layer 5
constexpr int NT = 64 * NPB4 * 2;                   // one wave per (pixel block, 32-channel block) of layer 4
constexpr int NIN = (IN_PIX * 12 + NT - 1) / NT, NWR = (3 * C_MID * 12 + NT - 1) / NT;
constexpr int S_IN = IN_PIX * 192;                  // one 32-channel chunk of the input tile: 27,648 B
constexpr int S_MID = MID_PIX * 2 * 192;            // layer-4 output, 2 chunks: 41,472 B
constexpr int S_TAP = C_MID * 192;                  // one tap of one chunk: 12,288 B (layer 5: the first half)
constexpr int S_SLAB = 3 * S_TAP;                   // a slab = the three taps of one kernel row
constexpr int TAIL_LDS = S_IN + S_MID + 2 * S_SLAB; // 142,848 B

struct TailProb {
  const unsigned char* x;                           // [pixel][16 groups][3 planes][8 bf16]: the whole 128-channel tensor
  const unsigned char* w4;
  const float* b4;
  const unsigned char* w5;
  const float* b5;
  float* out;
  const float* post;
  const float* post2;
  int ld_out, ld_post, ld_post2, act;
};
struct TailArgs {
  TailProb p[VAM_MAX_TAIL_GROUP];
  int B, H;
};

__device__ __forceinline__ void split4(const float (&v)[4], uint2& h, uint2& m, uint2& l) {
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

#define MFMA6(acc, w, p)                                                              \
  do {                                                                                \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], p[0], acc, 0, 0, 0);          \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], p[2], acc, 0, 0, 0);          \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], p[1], acc, 0, 0, 0);          \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], p[0], acc, 0, 0, 0);          \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], p[1], acc, 0, 0, 0);          \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], p[0], acc, 0, 0, 0);          \
  } while (0)

// LDS row of 192 B = [plane 3][4 groups x 16 B], the 16-byte unit of group g stored at g ^ ((row >> 1) & 3): the 32 lanes of a
// half-wave read rows r .. r+15 (one image row) of one group — distinct banks for 8 consecutive rows x 2 halves
__device__ __forceinline__ int unit_off(int row, int plane, int g) { return row * 192 + plane * 64 + ((g ^ ((row >> 1) & 3)) << 4); }

__global__ __launch_bounds__(NT) void stack_tail_kernel(const TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sIn = smem;
  unsigned char* sMid = smem + S_IN;
  unsigned char* sW = smem + S_IN + S_MID;
  const int HH = a.H;
  const int tiles = HH / TR;
  int bid = blockIdx.x;
  const int pi = bid / (a.B * tiles);
  bid -= pi * a.B * tiles;
  const int img = bid / tiles, r0 = (bid - img * tiles) * TR;
  const TailProb& P = a.p[pi];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;

  // zero the layer-4 image once (its border columns and out-of-image rows are the padding of layer 5)
  for (int i = tid; i < S_MID / 16; i += NT) reinterpret_cast<uint4*>(sMid)[i] = make_uint4(0, 0, 0, 0);

  // ---- staging roles
  // input chunk: IN_PIX x 12 units of 16 B; unit u -> pixel u / 12, e = u % 12 = (group e / 3, plane e % 3) in the P3 source
  auto load_in = [&](int chunk, u32x4 (&r)[NIN]) {
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int u = tid + NT * i;
      r[i] = u32x4{0, 0, 0, 0};
      if (u < IN_PIX * 12) {
        const int px = u / 12, e = u - px * 12, g = e / 3, pl = e - g * 3;
        const int ry = px / HALO_W, cx = px - ry * HALO_W;
        const int iy = r0 - 2 + ry, ix = cx - 1;
        if ((unsigned)iy < (unsigned)HH && (unsigned)ix < (unsigned)WW)
          r[i] = *reinterpret_cast<const u32x4*>(P.x + ((size_t)((img * HH + iy) * WW + ix) * (C_IN / 8) + chunk * 4 + g) * 48 + pl * 16);
      }
    }
  };
  auto store_in = [&](const u32x4 (&r)[NIN]) {
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int u = tid + NT * i;
      if (u < IN_PIX * 12) {
        const int px = u / 12, e = u - px * 12, g = e / 3, pl = e - g * 3;
        *reinterpret_cast<u32x4*>(sIn + unit_off(px, pl, g)) = r[i];
      }
    }
  };
  // weight slab: rows x 12 units of 16 B, packed row = [plane][group] already
  auto load_w = [&](const unsigned char* w, size_t tstride, int rows, u32x4 (&r)[NWR]) {
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      const int u = tid + NT * i;
      const int t = u / (rows * 12), v = u - t * (rows * 12);
      r[i] = u < 3 * rows * 12 ? *reinterpret_cast<const u32x4*>(w + t * tstride + (size_t)v * 16) : u32x4{0, 0, 0, 0};
    }
  };
  auto store_w = [&](int buf, int rows, const u32x4 (&r)[NWR]) {
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      const int u = tid + NT * i;
      if (u < 3 * rows * 12) {
        const int t = u / (rows * 12), v = u - t * (rows * 12);
        const int row = v / 12, c = v - row * 12;
        *reinterpret_cast<u32x4*>(sW + buf * S_SLAB + t * S_TAP + unit_off(row, c >> 2, c & 3)) = r[i];
      }
    }
  };

  // =============================================================== layer 4: 96 pixels x 64 channels, K = 4 chunks x 9 taps x 2
  const int pb4 = wid % NPB4, nb4 = wid / NPB4;                      // pixel block (32 of the 96), channel block
  const int p4 = pb4 * 32 + l31;                               // layer-4 pixel: row p4 / 16 of MID_ROWS, column p4 % 16
  const int my = p4 >> 4, mx = p4 & 15;
  f32x16 acc4;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc4[r] = 0.f;
  // weight slabs: slab s is requested at step s - 2 (two register sets, alternating), stored to LDS ring slot s & 1 at step
  // s, one barrier before its use: two steps of latency hidden
  u32x4 rin[NIN], rwA[NWR], rwB[NWR];
  // step = (chunk, kernel row): 12 steps; slab `st` is requested at step st - 2 and stored at the top of step st
  auto w4_of = [&](int st_) { const int nc = st_ / 3, ty_ = st_ - nc * 3; return P.w4 + (size_t)((ty_ * 3) * 4 + nc) * C_MID * 192; };
  const size_t ts4 = (size_t)4 * C_MID * 192;
  load_in(0, rin);
  load_w(w4_of(0), ts4, C_MID, rwA);
  load_w(w4_of(1), ts4, C_MID, rwB);
  __syncthreads();                                             // sMid zeroed
  for (int chunk = 0; chunk < 4; ++chunk) {
    store_in(rin);
    if (chunk + 1 < 4) load_in(chunk + 1, rin);
    for (int ty = 0; ty < 3; ++ty) {
      const int st = chunk * 3 + ty;
      if (st & 1) {
        store_w(1, C_MID, rwB);
        if (st + 2 < 12) load_w(w4_of(st + 2), ts4, C_MID, rwB);
      } else {
        store_w(0, C_MID, rwA);
        if (st + 2 < 12) load_w(w4_of(st + 2), ts4, C_MID, rwA);
      }
      __syncthreads();                                         // slab `st` (and, at row 0, the input chunk) is in LDS
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int ipx = (my + ty) * HALO_W + mx + tx;          // input halo pixel of this lane's layer-4 pixel under the tap
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          bf16x8 p[3], w[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) p[pl] = *reinterpret_cast<const bf16x8*>(sIn + unit_off(ipx, pl, 2 * hf + lh));
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            w[pl] = *reinterpret_cast<const bf16x8*>(sW + (st & 1) * S_SLAB + tx * S_TAP + unit_off(nb4 * 32 + l31, pl, 2 * hf + lh));
          MFMA6(acc4, w, p);
        }
      }
      if (ty == 2) __syncthreads();                            // all reads of this input chunk are done before the next store_in
    }
  }
  // layer-4 epilogue: lane = pixel p4, registers 4j .. 4j+3 = channels nb4*32 + 8j + 4lh + {0..3}; GELU; planes into sMid
  {
    const int gy = r0 - 1 + my;                                // image row of this layer-4 pixel
    const bool in = (unsigned)gy < (unsigned)HH;
    const int row = my * HALO_W + mx + 1;                      // sMid pixel (column shifted by the zero column)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 bb = *reinterpret_cast<const float4*>(P.b4 + nb4 * 32 + 8 * j + 4 * lh);
      float v[4] = {acc4[4 * j] + bb.x, acc4[4 * j + 1] + bb.y, acc4[4 * j + 2] + bb.z, acc4[4 * j + 3] + bb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = in ? vam_gelu(v[i]) : 0.f;
      uint2 h, m, l;
      split4(v, h, m, l);
      unsigned char* d = sMid + nb4 * (MID_PIX * 192) + lh * 8;    // chunk nb4 of the 64 channels; group j of the chunk
      *reinterpret_cast<uint2*>(d + unit_off(row, 0, j)) = h;
      *reinterpret_cast<uint2*>(d + unit_off(row, 1, j)) = m;
      *reinterpret_cast<uint2*>(d + unit_off(row, 2, j)) = l;
    }
  }
  // =============================================================== layer 5: 64 pixels x 32 channels, K = 2 chunks x 9 taps x 2
  const int p5 = (wid % NPB5) * 32 + l31;                      // waves 0 .. NPB5-1 compute; the others only stage slabs
  const int oy = p5 >> 4, ox = p5 & 15;
  f32x16 acc5;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc5[r] = 0.f;
  auto w5_of = [&](int st_) { const int nc = st_ / 3, ty_ = st_ - nc * 3; return P.w5 + (size_t)((ty_ * 3) * 2 + nc) * C_OUT * 192; };
  const size_t ts5 = (size_t)2 * C_OUT * 192;
  load_w(w5_of(0), ts5, C_OUT, rwA);
  load_w(w5_of(1), ts5, C_OUT, rwB);
  __syncthreads();                                             // sMid complete; layer-4 slab buffers free
  for (int st = 0; st < 6; ++st) {
    const int chunk = st / 3, ty = st - chunk * 3;
    if (st & 1) {
      store_w(1, C_OUT, rwB);
      if (st + 2 < 6) load_w(w5_of(st + 2), ts5, C_OUT, rwB);
    } else {
      store_w(0, C_OUT, rwA);
      if (st + 2 < 6) load_w(w5_of(st + 2), ts5, C_OUT, rwA);
    }
    __syncthreads();
    if (wid < NPB5) {
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int mpx = (oy + ty) * HALO_W + ox + tx;          // layer-4 halo pixel (rows r0-1.., columns with the zero border)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          bf16x8 p[3], w[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) p[pl] = *reinterpret_cast<const bf16x8*>(sMid + chunk * (MID_PIX * 192) + unit_off(mpx, pl, 2 * hf + lh));
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const bf16x8*>(sW + (st & 1) * S_SLAB + tx * S_TAP + unit_off(l31, pl, 2 * hf + lh));
          MFMA6(acc5, w, p);
        }
      }
    }
  }
  if (wid < NPB5) {
    // conv_igemm_kernel's epilogue order: (acc + bias) -> act -> + post -> + post2
    const size_t opix = (size_t)(img * HH + r0 + oy) * WW + ox;
    float* o = P.out + opix * P.ld_out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 8 * j + 4 * lh;
      const float4 bb = *reinterpret_cast<const float4*>(P.b5 + c);
      float v[4] = {acc5[4 * j] + bb.x, acc5[4 * j + 1] + bb.y, acc5[4 * j + 2] + bb.z, acc5[4 * j + 3] + bb.w};
      if (P.act == VAM_ACT_HALF_TANH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = 0.5f * tanhf(v[i]);
      }
      if (P.post) {
        const float4 t = *reinterpret_cast<const float4*>(P.post + opix * P.ld_post + c);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      }
      if (P.post2) {
        const float4 t = *reinterpret_cast<const float4*>(P.post2 + opix * P.ld_post2 + c);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      }
      *reinterpret_cast<float4*>(o + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

}  // namespace vam

using namespace vam;

extern "C" int vam_stack_tail_group(const vam_stack_tail* probs, int n, void* stream) {
  VAM_REQUIRE(probs && n >= 1, "vam_stack_tail_group: no problems");
  const int B = probs[0].B, H = probs[0].H, W = probs[0].W;
  VAM_REQUIRE(B >= 1 && W == WW && H >= TR && H % TR == 0, "vam_stack_tail_group: built for latents 16 columns wide, rows a multiple of 4 (got %d x %d)", H, W);
  VAM_REQUIRE((long)B * H * W < (1L << 24), "vam_stack_tail_group: too many pixels");
  static bool attr_set = false;
  if (!attr_set) {
    VAM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stack_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS));
    attr_set = true;
  }
  for (int i0 = 0; i0 < n; i0 += VAM_MAX_TAIL_GROUP) {
    const int m = n - i0 < VAM_MAX_TAIL_GROUP ? n - i0 : VAM_MAX_TAIL_GROUP;
    TailArgs a;
    a.B = B;
    a.H = H;
    for (int i = 0; i < VAM_MAX_TAIL_GROUP; ++i) {
      const vam_stack_tail& q = probs[i0 + (i < m ? i : 0)];
      if (i < m) {
        VAM_REQUIRE(q.B == B && q.H == H && q.W == W, "vam_stack_tail_group: problem %d has another extent", i0 + i);
        VAM_REQUIRE(q.x && q.w4 && q.b4 && q.w5 && q.b5 && q.out, "vam_stack_tail_group: problem %d: null operand", i0 + i);
        VAM_REQUIRE(q.x_groups == C_IN / 8 && q.ld_out >= C_OUT && q.ld_out % 4 == 0 && ((uintptr_t)q.out & 15) == 0 && ((uintptr_t)q.x & 15) == 0,
                    "vam_stack_tail_group: problem %d: pitches / alignment", i0 + i);
        VAM_REQUIRE(q.act == VAM_ACT_NONE || q.act == VAM_ACT_HALF_TANH, "vam_stack_tail_group: problem %d: activation %d", i0 + i, q.act);
        VAM_REQUIRE((!q.post.ptr || (q.post.ld >= C_OUT && q.post.ld % 4 == 0 && ((uintptr_t)q.post.ptr & 15) == 0)) &&
                    (!q.post2.ptr || (q.post2.ld >= C_OUT && q.post2.ld % 4 == 0 && ((uintptr_t)q.post2.ptr & 15) == 0)),
                    "vam_stack_tail_group: problem %d: epilogue operand", i0 + i);
      }
      TailProb& t = a.p[i];
      t.x = static_cast<const unsigned char*>(q.x);
      t.w4 = static_cast<const unsigned char*>(q.w4);
      t.b4 = q.b4;
      t.w5 = static_cast<const unsigned char*>(q.w5);
      t.b5 = q.b5;
      t.out = q.out;
      t.post = q.post.ptr;
      t.post2 = q.post2.ptr;
      t.ld_out = q.ld_out;
      t.ld_post = q.post.ld;
      t.ld_post2 = q.post2.ld;
      t.act = q.act;
    }
    const double px = (double)m * B * H * W;
    ProfScope ps(VAM_FAM_CONV, (hipStream_t)stream, 2.0 * px * 9.0 * (C_IN * C_MID + C_MID * C_OUT),
                 4.0 * (px * (C_IN + C_OUT) + 9.0 * m * (C_IN * C_MID + C_MID * C_OUT)));     // (the 64-channel intermediate never leaves LDS)
    hipLaunchKernelGGL(stack_tail_kernel, dim3((unsigned)(m * B * (H / TR))), dim3(NT), TAIL_LDS, (hipStream_t)stream, a);
    if (int rc = check_launch("stack_tail_kernel")) return rc;
  }
  return 0;
}
