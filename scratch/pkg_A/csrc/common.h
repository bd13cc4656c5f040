// Shared host-side helpers of libvampic (error reporting, launch bracketing for the
// HIP-event profiler).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../../include/vampic.h"

namespace vam {

void set_error(const char* fmt, ...);

#define VAM_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      vam::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VAM_EHIP;                                                             \
    }                                                                              \
  } while (0)

#define VAM_REQUIRE(cond, ...)                                                     \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      vam::set_error(__VA_ARGS__);                                                 \
      return VAM_EINVAL;                                                           \
    }                                                                              \
  } while (0)

// Event bracketing: when profiling is on, record an event before and after the launch
// on the SAME stream the kernel runs on; vam_prof_read sums the elapsed times.
struct ProfScope {
  ProfScope(int family, hipStream_t s, double flops, double bytes);
  ~ProfScope();
  int family;
  hipStream_t stream;
  bool active;
  size_t slot;
};

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("launch of %s failed: %s", what, hipGetErrorString(e));
    return VAM_EHIP;
  }
  return VAM_OK;
}

inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

}  // namespace vam
