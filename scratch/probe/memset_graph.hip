// Probe for the round-2 finding "a hipMemsetAsync NODE in a captured hipGraph, in front of a kernel that accumulates into
// the cleared buffer with float atomics, intermittently left garbage" (DESIGN.md section 5).  Stand-alone: captures
//   [filler kernel] -> hipMemsetAsync(table, 0, 392 floats) -> accumulate kernel (atomicAdd into table) -> [reader kernel]
// on ONE stream (ThreadLocal capture, as libvampic does), dumps the graph with hipGraphDebugDotPrint, replays it N times
// and checks the table after every replay.  Variants: memset node vs zero-fill kernel; table sub-allocated at an unaligned
// offset of a larger buffer (as a torch caching-allocator block would be) vs its own allocation.
//   hipcc --offload-arch=gfx950 -O2 -o memset_graph memset_graph.hip && ./memset_graph
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void filler(float* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.0f;
}
__global__ void zero_fill(float* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.f;
}
// every block adds 1 to every table entry (LDS staging + one global atomic per entry, like win_attn_bwd_kernel)
__global__ void accumulate(float* table, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(table + i, 1.0f);
}
__global__ void reader(const float* table, int n, float* out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = table[i];
}

int run(bool memset_node, bool suballoc, int replays, const char* dot) {
  const int n = 392, blocks = 16384;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *big, *table, *out, *fill;
  const long nf = 8 << 20;
  CK(hipMalloc(&fill, nf * 4));
  CK(hipMemset(fill, 0, nf * 4));
  CK(hipMalloc(&big, 1 << 20));
  table = suballoc ? big + 1027 * 4 : big;          // 16-byte aligned, not 256-byte aligned
  CK(hipMalloc(&out, n * 4));
  // poison, so that an un-cleared element shows
  std::vector<float> poison(n, 1e20f);
  CK(hipMemcpy(table, poison.data(), n * 4, hipMemcpyHostToDevice));
  hipGraph_t g;
  hipGraphExec_t ex;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(filler, dim3(2048), dim3(256), 0, s, fill, nf);
  if (memset_node) CK(hipMemsetAsync(table, 0, n * 4, s));
  else hipLaunchKernelGGL(zero_fill, dim3(2), dim3(256), 0, s, table, n);
  hipLaunchKernelGGL(accumulate, dim3(blocks), dim3(64), 0, s, table, n);
  hipLaunchKernelGGL(reader, dim3(2), dim3(256), 0, s, table, n, out);
  hipLaunchKernelGGL(filler, dim3(2048), dim3(256), 0, s, fill, nf);
  CK(hipStreamEndCapture(s, &g));
  if (dot) CK(hipGraphDebugDotPrint(g, dot, hipGraphDebugDotFlagsVerbose));
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  std::vector<float> h(n);
  int bad = 0;
  for (int r = 0; r < replays; ++r) {
    CK(hipGraphLaunch(ex, s));
    if ((r & 7) == 7 || r == replays - 1) {          // back-to-back launches in flight, then a check
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < n; ++i)
        if (h[i] != (float)blocks) {
          if (bad < 5) printf("  replay %d entry %d: %g (expected %d)\n", r, i, h[i], blocks);
          ++bad;
        }
    }
  }
  printf("%s, %s: %d bad entries over %d replays\n", memset_node ? "memset NODE" : "zero-fill KERNEL", suballoc ? "sub-allocated table" : "own allocation",
         bad, replays);
  CK(hipGraphExecDestroy(ex));
  CK(hipGraphDestroy(g));
  CK(hipFree(big)); CK(hipFree(out)); CK(hipFree(fill));
  CK(hipStreamDestroy(s));
  return bad;
}

int main(int argc, char** argv) {
  const int replays = argc > 1 ? atoi(argv[1]) : 2000;
  int bad = 0;
  bad += run(true, true, replays, "memset_graph_node.dot");
  bad += run(true, false, replays, nullptr);
  bad += run(false, true, replays, "memset_graph_kernel.dot");
  printf("total bad: %d\n", bad);
  return 0;
}
