// libvampic runtime pieces: error string, device query, HIP-graph capture helpers and
// the HIP-event launch profiler used by bench.py's roofline leg.
#include "common.h"
#include <unordered_map>
#include <cstring>
#include <mutex>

namespace vam {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct ProfRec {
  hipEvent_t a, b;
  int family;
  int cls;
};
struct ProfState {
  bool on = false;
  std::vector<ProfRec> recs;
  double ms[VAM_FAM_COUNT] = {0};
  long launches[VAM_FAM_COUNT] = {0};
  double flops[VAM_FAM_COUNT] = {0};
  double bytes[VAM_FAM_COUNT] = {0};
  // the same sums per caller-defined launch class (vam_prof_set_class): which part of the model a launch belongs to
  int cur_class = 0;
  double c_ms[VAM_PROF_CLASSES] = {0};
  long c_launches[VAM_PROF_CLASSES] = {0};
  double c_flops[VAM_PROF_CLASSES] = {0};
  double c_bytes[VAM_PROF_CLASSES] = {0};
  std::mutex mu;
};
static ProfState g_prof;

ProfScope::ProfScope(int fam, hipStream_t s, double fl, double by)
    : family(fam), stream(s), active(false), slot(0) {
  if (!g_prof.on) return;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  ProfRec r;
  r.family = fam;
  r.cls = g_prof.cur_class;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, s);
  g_prof.recs.push_back(r);
  slot = g_prof.recs.size() - 1;
  g_prof.launches[fam] += 1;
  g_prof.flops[fam] += fl;
  g_prof.bytes[fam] += by;
  if (fam == VAM_FAM_CONV) {
    g_prof.c_launches[r.cls] += 1;
    g_prof.c_flops[r.cls] += fl;
    g_prof.c_bytes[r.cls] += by;
  }
  active = true;
}

ProfScope::~ProfScope() {
  if (!active) return;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  (void)hipEventRecord(g_prof.recs[slot].b, stream);
}

static int prof_drain() {
  for (auto& r : g_prof.recs) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      g_prof.ms[r.family] += ms;
      if (r.family == VAM_FAM_CONV) g_prof.c_ms[r.cls] += ms;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.recs.clear();
  return VAM_OK;
}

}  // namespace vam

using namespace vam;

extern "C" {

const char* vam_last_error(void) { return g_err; }
int vam_version(void) { return 100; }
size_t vam_conv_struct_size(void) { return sizeof(vam_conv); }

int vam_device_info(char* name128, int* cu_count) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    set_error("no HIP device visible");
    return VAM_ENOGPU;
  }
  int dev = 0;
  VAM_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  VAM_CHECK_HIP(hipGetDeviceProperties(&p, dev));
  if (name128) {
    snprintf(name128, 128, "%s (%s)", p.name, p.gcnArchName);
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  return VAM_OK;
}

static std::mutex g_tmpl_mu;
static std::unordered_map<void*, hipGraph_t> g_tmpl;     // VAMPIC_GRAPH_KEEP_TEMPLATE=1: executable graph -> its template

int vam_graph_begin(void* stream) {
  VAM_REQUIRE(!g_prof.on, "vam_graph_begin: disable the event profiler before capturing");
  VAM_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return VAM_OK;
}

int vam_graph_end(void* stream, void** exec_out) {
  hipGraph_t g = nullptr;
  VAM_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t ex = nullptr;
  hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  // The template graph is not needed once the executable graph exists (CUDA / HIP semantics).  VAMPIC_GRAPH_KEEP_TEMPLATE=1
  // keeps it until vam_graph_destroy — one of the ingredients the graph-destroy probes toggle.
  static int keep = -1;
  if (keep < 0) { const char* k = getenv("VAMPIC_GRAPH_KEEP_TEMPLATE"); keep = (k && k[0] == '1') ? 1 : 0; }
  if (e != hipSuccess || !keep) (void)hipGraphDestroy(g);
  if (e != hipSuccess) {
    set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return VAM_EHIP;
  }
  if (keep) {
    std::lock_guard<std::mutex> lk(g_tmpl_mu);
    g_tmpl[(void*)ex] = g;
  }
  *exec_out = (void*)ex;
  return VAM_OK;
}

int vam_graph_launch(void* exec, void* stream) {
  VAM_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream));
  return VAM_OK;
}

int vam_graph_destroy(void* exec) {
  // The CALLER guarantees that no launch of this executable graph is in flight and that the calling thread is not
  // capturing (ops.Graph parks dropped handles and destroys them from the next plan entry point, after synchronising the
  // stream they were last launched on).  Nothing is synchronised here: a device-wide wait from a destructor is illegal
  // inside another plan's stream capture and hid the ordering the caller has to provide anyway.
  if (exec) {
    VAM_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)exec));
    hipGraph_t g = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_tmpl_mu);
      auto it = g_tmpl.find(exec);
      if (it != g_tmpl.end()) { g = it->second; g_tmpl.erase(it); }
    }
    if (g) VAM_CHECK_HIP(hipGraphDestroy(g));
  }
  return VAM_OK;
}

int vam_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.on = on != 0;
  return VAM_OK;
}

int vam_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  prof_drain();
  for (int i = 0; i < VAM_FAM_COUNT; ++i) {
    g_prof.ms[i] = 0;
    g_prof.launches[i] = 0;
    g_prof.flops[i] = 0;
    g_prof.bytes[i] = 0;
  }
  for (int i = 0; i < VAM_PROF_CLASSES; ++i) {
    g_prof.c_ms[i] = 0;
    g_prof.c_launches[i] = 0;
    g_prof.c_flops[i] = 0;
    g_prof.c_bytes[i] = 0;
  }
  return VAM_OK;
}

int vam_prof_set_class(int cls) {
  VAM_REQUIRE(cls >= 0 && cls < VAM_PROF_CLASSES, "vam_prof_set_class: class %d outside 0..%d", cls, VAM_PROF_CLASSES - 1);
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.cur_class = cls;
  return VAM_OK;
}

int vam_prof_read_class(int cls, double* ms, long* launches, double* flops, double* bytes) {
  VAM_REQUIRE(cls >= 0 && cls < VAM_PROF_CLASSES, "vam_prof_read_class: class %d outside 0..%d", cls, VAM_PROF_CLASSES - 1);
  std::lock_guard<std::mutex> lk(g_prof.mu);
  prof_drain();
  if (ms) *ms = g_prof.c_ms[cls];
  if (launches) *launches = g_prof.c_launches[cls];
  if (flops) *flops = g_prof.c_flops[cls];
  if (bytes) *bytes = g_prof.c_bytes[cls];
  return VAM_OK;
}

int vam_prof_read(int family, double* ms, long* launches, double* flops, double* bytes) {
  VAM_REQUIRE(family >= 0 && family < VAM_FAM_COUNT, "vam_prof_read: bad family %d", family);
  std::lock_guard<std::mutex> lk(g_prof.mu);
  prof_drain();
  if (ms) *ms = g_prof.ms[family];
  if (launches) *launches = g_prof.launches[family];
  if (flops) *flops = g_prof.flops[family];
  if (bytes) *bytes = g_prof.bytes[family];
  return VAM_OK;
}

}  // extern "C"
