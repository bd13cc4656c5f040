// Probe for VERDICT r02 item 3b: is v_mfma_f32_16x16x32_bf16 faster than v_mfma_f32_32x32x16_bf16 for the consumer wave
// of the split-operand convolution (6 products per block, fragments re-read from LDS every 32-channel chunk), on a chip
// whose clock is power-limited while the matrix pipe is busy?  Stand-alone; no global traffic inside the timed loop, so it
// bounds what a re-tiling of conv_igemm / resunit could gain from the instruction alone.
//
// One workgroup = 4 waves in a 2 x 2 arrangement over a 128 x 192 tile (the kernel's best tile): a wave owns 64 x 96.
// Per chunk (k = 32) a wave reads its A rows (64 x 32 x 3 planes) and B rows (96 x 32 x 3 planes) from LDS with
// ds_read_b128 — the same 30 reads for either shape — and issues
//   32x32x16:  2 x 3 blocks x 2 k-steps x 6 products =  72 MFMAs of  8 passes
//   16x16x32:  4 x 6 blocks x 1 k-step  x 6 products = 144 MFMAs of  4 passes        (96 accumulator registers either way)
// LDS rows are 192 B (3 planes x 32 bf16), XOR-swizzled in 16 B units like the kernel's.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shapes mfma_shapes.hip && ./mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROW_B = 192;                 // bytes per LDS row: plane p at p * 64, k (0..31) at 2 B each
constexpr int A_ROWS = 128, B_ROWS = 192;

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* base, int row, int plane, int kblk) {
  // 16 B unit index inside the row: plane * 4 + kblk, XOR-swizzled by the row quad.  conv_igemm uses (row >> 2) & 3; the table
  // {0, 2, 3, 1} is conflict-free for BOTH access patterns: ds_read_b128 is served in lane groups {0-3, 12-15, 20-27} /
  // {4-11, 16-19, 28-31} (+32), which for the 16-row fragment mix two k-blocks (pattern k, k^1, k^1, k over the row quads)
  const int unit = (plane * 4 + kblk) ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3);
  return *reinterpret_cast<const bf16x8*>(base + row * ROW_B + unit * 16);
}

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ img, float* __restrict__ out, int iters) {
  extern __shared__ unsigned char lds[];
  unsigned char* sA = lds;
  unsigned char* sB = lds + A_ROWS * ROW_B;
  for (int i = threadIdx.x; i < (A_ROWS + B_ROWS) * ROW_B / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = img[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float sum = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[2][3];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int r32 = lane & 31, kh = lane >> 5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[2][3], fb[3][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int p = 0; p < 3; ++p) fa[i][p] = lds_frag(sA, wm * 64 + i * 32 + r32, p, ks * 2 + kh);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) fb[j][p] = lds_frag(sB, wn * 96 + j * 32 + r32, p, ks * 2 + kh);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
          }
      }
      asm volatile("" ::: "memory");     // the fragments are re-read every chunk, as in the kernel (the ring slot changes there)
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  } else {
    f32x4 acc[4][6];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const int r16 = lane & 15, kq = lane >> 4;
    for (int it = 0; it < iters; ++it) {
      bf16x8 fa[4][3], fb[6][3];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[i][p] = lds_frag(sA, wm * 64 + i * 16 + r16, p, kq);
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) fb[j][p] = lds_frag(sB, wn * 96 + j * 16 + r16, p, kq);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
        }
      asm volatile("" ::: "memory");
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

static uint16_t bf16_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

int main(int argc, char** argv) {
  const int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 2;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  const bool zeros = argc > 3 && atoi(argv[3]) == 0;      // third argument 0: all-zero operands (no toggling: the clock's upper end)
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t img_bytes = (size_t)(A_ROWS + B_ROWS) * ROW_B;
  // operands as the kernel sees them: hi / mid / lo planes of Gaussian fp32 values (so the planes' magnitudes differ by 2^-8)
  std::vector<uint16_t> h(img_bytes / 2);
  srand(1);
  for (int row = 0; row < A_ROWS + B_ROWS; ++row)
    for (int k = 0; k < 32; ++k) {
      float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
      float v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2), rest = v;
      for (int p = 0; p < 3; ++p) {
        uint16_t b = zeros ? 0 : bf16_bits(rest);
        uint32_t w = (uint32_t)b << 16;
        float back;
        memcpy(&back, &w, 4);
        rest -= back;
        const int rr = row < A_ROWS ? row : row - A_ROWS;          // same swizzle as lds_frag (rows relative to their image)
        const int unit2 = (p * 4 + k / 8) ^ ((0x78 >> (2 * ((rr >> 2) & 3))) & 3);
        h[((size_t)row * ROW_B + unit2 * 16) / 2 + (k % 8)] = b;
      }
    }
  void* img;
  float* out;
  CK(hipMalloc(&img, img_bytes));
  CK(hipMemcpy(img, h.data(), img_bytes, hipMemcpyHostToDevice));
  const int grid = cus * wgs_per_cu;
  CK(hipMalloc(&out, (size_t)grid * 256 * 4));
  const size_t lds_bytes = wgs_per_cu == 1 ? 120 * 1024 : img_bytes;      // 1: force one workgroup per CU
  CK(hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  CK(hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double flop = 2.0 * 128 * 192 * 32 * 6 * (double)iters * grid;      // executed bf16 MFMA flops
  printf("%d CUs, %d workgroup(s) per CU x 4 waves, %d chunks per wave, %.1f GFLOP per launch, %s operands\n", cus, wgs_per_cu, iters, flop / 1e9,
         zeros ? "all-zero" : "Gaussian");
  std::vector<float> sums[2];
  for (int rep = 0; rep < 4; ++rep)
    for (int shape = 0; shape < 2; ++shape) {
      CK(hipEventRecord(e0, 0));
      if (shape == 0) probe<0><<<grid, 256, lds_bytes>>>((const uint4*)img, out, iters);
      else probe<1><<<grid, 256, lds_bytes>>>((const uint4*)img, out, iters);
      CK(hipGetLastError());
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 0) {
        sums[shape].resize(256);
        CK(hipMemcpy(sums[shape].data(), out, 256 * 4, hipMemcpyDeviceToHost));
      }
      printf("rep %d  %s  %8.2f ms  %7.1f TF/s executed  (%5.1f TF/s fp32-equivalent)\n", rep,
             shape == 0 ? "32x32x16" : "16x16x32", ms, flop / ms / 1e9, flop / ms / 1e9 / 6);
    }
  // both shapes computed the same 128 x 192 tile sums (up to summation order): compare the wave totals
  double t0 = 0, t1 = 0;
  for (int i = 0; i < 256; ++i) { t0 += sums[0][i]; t1 += sums[1][i]; }
  printf("tile checksum 32x32x16 %.6e   16x16x32 %.6e   (relative difference %.2e)\n", t0, t1, fabs(t0 - t1) / (fabs(t0) + 1e-30));
  return 0;
}
