// Numerical probe: fp32 GEMM via v_mfma_f32_32x32x2_f32 vs 3-way bf16 split on v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short bf16_rn(float x) {   // round to nearest even
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ inline float bf16_f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// A: [32][K] row-major, B: [32][K] (n-major), C: [32][32]
__global__ void k_f32(const float* A, const float* B, float* C, int K) {
  int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + lh], B[l31 * K + k + lh], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[r];
}
template <int NPROD>
__global__ void k_bf16(const float* A, const float* B, float* C, int K) {
  int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += 16) {
    bf16x8 a[3], b[3];
    for (int e = 0; e < 8; ++e) {
      float x = A[l31 * K + k + lh * 8 + e], y = B[l31 * K + k + lh * 8 + e];
      unsigned short xh = bf16_rn(x); float r1 = x - bf16_f(xh);
      unsigned short xm = (unsigned short)(__float_as_uint(r1) >> 16); float r2 = r1 - bf16_f(xm);
      unsigned short xl = (unsigned short)(__float_as_uint(r2) >> 16);
      unsigned short yh = (unsigned short)(__float_as_uint(y) >> 16); float s1 = y - bf16_f(yh);
      unsigned short ym = (unsigned short)(__float_as_uint(s1) >> 16); float s2 = s1 - bf16_f(ym);
      unsigned short yl = (unsigned short)(__float_as_uint(s2) >> 16);
      unsigned short xs[3] = {xh, xm, xl}, ys[3] = {yh, ym, yl};
      for (int p = 0; p < 3; ++p) {
        a[p][e] = __builtin_bit_cast(__bf16, xs[p]);
        b[p][e] = __builtin_bit_cast(__bf16, ys[p]);
      }
    }
    // smallest terms first
    if (NPROD >= 9) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0);
                      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0);
                      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0); }
    if (NPROD >= 6) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
                      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
                      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0); }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[r];
}
int main() {
  for (int K : {192, 1728, 4800}) {
    std::vector<float> A(32 * K), B(32 * K);
    srand(K);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.3f) * 4.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    std::vector<double> ref(1024);
    std::vector<float> seq(1024);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0; float f = 0.f;
      for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[j * K + k]; f = fmaf(A[i * K + k], B[j * K + k], f); }
      ref[i * 32 + j] = s; seq[i * 32 + j] = f;
    }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> C(1024);
    auto err = [&](const float* c) { double m = 0, rms = 0, sc = 0; for (int i = 0; i < 1024; ++i) { double e = fabs(c[i] - ref[i]); m = fmax(m, e); rms += e * e; sc += ref[i] * ref[i]; } printf("max %.3e rms/rmsref %.3e", m, sqrt(rms / sc)); };
    printf("K=%d  cpu-seq-fma: ", K); err(seq.data()); printf("\n");
    k_f32<<<1, 64>>>(dA, dB, dC, K); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); printf("        f32 mfma   : "); err(C.data()); printf("\n");
    k_bf16<3><<<1, 64>>>(dA, dB, dC, K); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); printf("        bf16x3 (3) : "); err(C.data()); printf("\n");
    k_bf16<6><<<1, 64>>>(dA, dB, dC, K); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); printf("        bf16x3 (6) : "); err(C.data()); printf("\n");
    k_bf16<9><<<1, 64>>>(dA, dB, dC, K); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); printf("        bf16x3 (9) : "); err(C.data()); printf("\n");
  }
  return 0;
}
