// Stand-alone probe for the round-3 finding "hipGraphExecDestroy of ~20 finished executable graphs, then capture +
// hipGraphLaunch of a new one, segfaults inside the runtime" (gpurun_out/r3_segv.log: ops.Graph.launch
// -> hipGraphLaunch in tests/test_gpu_bitstream.py after net.update() had dropped the model's plans).
//
// One process = one variant: capture G graphs of ~230 kernel nodes the way libvampic's plans do (ThreadLocal capture on a
// non-blocking stream, a second stream forked and joined through events inside the capture, kernel arguments passed by value
// in a ~1.8 KB struct like conv_igemm_kernel's GroupArgs), launch each a few times, synchronise, hipGraphExecDestroy ALL of
// them, then capture + launch a new one; ROUNDS times.  The bit mask on the command line adds the library's other
// ingredients one at a time:
//   1  hipGraphDestroy(template) right after hipGraphInstantiate (runtime.hip:vam_graph_end) instead of with the exec
//   2  the fork / join events are destroyed right after the capture (torch.cuda.Event objects of engine.Plan.run die there)
//   4  the side stream of a graph is destroyed while its executable graph lives (streams of dropped / deep-copied models)
//   8  graphs are launched on a DIFFERENT stream than the one they were captured on
//  16  hipMalloc / hipFree traffic between destroy and the next capture (torch's allocator releasing cached blocks)
//  32  the destroyed graphs' streams are only synchronised, not the device (as ops.drain_graveyard does)
//   hipcc --offload-arch=gfx950 -O2 -o graph_destroy graph_destroy.hip && ./graph_destroy <mask> [G] [ROUNDS]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(2); } } while (0)

struct BigArgs { float* p; long n; float k; int pad[450]; };      // ~1.8 KB by value, like GroupArgs

__global__ void work(const BigArgs a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) a.p[i] = a.p[i] * a.k + (float)a.pad[7];
}

struct G {
  hipGraph_t tmpl = nullptr;
  hipGraphExec_t ex = nullptr;
  hipStream_t main = nullptr, side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};

static G capture(int mask, float* buf, long n, int nodes) {
  G g;
  CK(hipStreamCreateWithFlags(&g.main, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&g.side, hipStreamNonBlocking));
  CK(hipEventCreateWithFlags(&g.fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&g.join, hipEventDisableTiming));
  BigArgs a{};
  a.p = buf; a.n = n; a.k = 1.0001f;
  CK(hipStreamBeginCapture(g.main, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nodes / 4; ++i) hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, g.main, a);
  CK(hipEventRecord(g.fork, g.main));
  CK(hipStreamWaitEvent(g.side, g.fork, 0));
  BigArgs b = a;
  b.p = buf + n;                                                    // the side branch works on the second half
  for (int i = 0; i < nodes / 4; ++i) {
    hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, g.main, a);
    hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, g.side, b);
  }
  CK(hipEventRecord(g.join, g.side));
  CK(hipStreamWaitEvent(g.main, g.join, 0));
  for (int i = 0; i < nodes / 4; ++i) hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, g.main, a);
  CK(hipStreamEndCapture(g.main, &g.tmpl));
  CK(hipGraphInstantiate(&g.ex, g.tmpl, nullptr, nullptr, 0));
  if (mask & 1) { CK(hipGraphDestroy(g.tmpl)); g.tmpl = nullptr; }
  if (mask & 2) { CK(hipEventDestroy(g.fork)); CK(hipEventDestroy(g.join)); g.fork = g.join = nullptr; }
  return g;
}

int main(int argc, char** argv) {
  const int mask = argc > 1 ? atoi(argv[1]) : 0;
  const int NG = argc > 2 ? atoi(argv[2]) : 24;
  const int rounds = argc > 3 ? atoi(argv[3]) : 4;
  const int nodes = 232;
  const long n = 1 << 20;
  float* buf;
  CK(hipMalloc(&buf, 2 * n * sizeof(float)));
  CK(hipMemset(buf, 0, 2 * n * sizeof(float)));
  hipStream_t other;
  CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
  printf("graph_destroy: mask %d, %d graphs of %d kernel nodes per round, %d rounds\n", mask, NG, nodes, rounds);
  fflush(stdout);
  long launched = 0, destroyed = 0;
  for (int r = 0; r < rounds; ++r) {
    std::vector<G> gs;
    for (int i = 0; i < NG; ++i) gs.push_back(capture(mask, buf, n, nodes));
    for (int rep = 0; rep < 3; ++rep)
      for (auto& g : gs) { CK(hipGraphLaunch(g.ex, (mask & 8) ? other : g.main)); ++launched; }
    if (mask & 32) { for (auto& g : gs) CK(hipStreamSynchronize((mask & 8) ? other : g.main)); }
    else CK(hipDeviceSynchronize());
    if (mask & 4) for (auto& g : gs) { CK(hipStreamDestroy(g.side)); g.side = nullptr; }
    for (auto& g : gs) {                                            // the step that preceded the crash in the library
      CK(hipGraphExecDestroy(g.ex)); ++destroyed;
      if (g.tmpl) CK(hipGraphDestroy(g.tmpl));
    }
    if (mask & 16) {
      std::vector<void*> ps;
      for (int i = 0; i < 64; ++i) { void* q; CK(hipMalloc(&q, (size_t)(1 + i % 7) << 20)); ps.push_back(q); }
      for (void* q : ps) CK(hipFree(q));
    }
    G fresh = capture(mask, buf, n, nodes);                         // "the capture + hipGraphLaunch of the new plan faults"
    for (int rep = 0; rep < 3; ++rep) { CK(hipGraphLaunch(fresh.ex, (mask & 8) ? other : fresh.main)); ++launched; }
    CK(hipDeviceSynchronize());
    printf("round %d: %ld launches, %ld executable graphs destroyed, new graph captured and launched\n", r, launched, destroyed);
    fflush(stdout);
    for (auto& g : gs) {
      if (g.side) CK(hipStreamDestroy(g.side));
      CK(hipStreamDestroy(g.main));
      if (g.fork) { CK(hipEventDestroy(g.fork)); CK(hipEventDestroy(g.join)); }
    }
    // `fresh` is kept alive (a live graph beside the next round's destroyed ones, like the model's current plan)
  }
  float h[4];
  CK(hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost));
  printf("done: mask %d, no fault (buf[0] = %g)\n", mask, h[0]);
  return 0;
}
