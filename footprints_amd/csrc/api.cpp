// Error plumbing and version for libfootprints_hip (see include/footprints_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "footprints_hip.h"

static thread_local char g_err[512] = "";

int fp_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int fp_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return FP_OK;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

extern "C" int fp_version(void) { return 1; }
extern "C" const char* fp_last_error_string(void) { return g_err; }

// amax sink (include/footprints_hip.h, fp_amax_out_next): consumed -- and cleared -- by the next launch of this thread that can publish
static thread_local uint32_t* g_amax_next = nullptr;
extern "C" int fp_amax_out_next(uint32_t* slot) {
  g_amax_next = slot;
  return FP_OK;
}
unsigned* fp_take_amax_out() {
  unsigned* s = g_amax_next;
  g_amax_next = nullptr;
  return s;
}

// BatchNorm-statistics sink (fp_bn_stats_out_next): consumed -- and cleared -- by this thread's next convolution launch
struct FpBnSink {
  float* part;
  int64_t cap_floats;
  int32_t* nblk_out;
  const float* z;
  const float* mean;
  const float* invstd;
};
static thread_local FpBnSink g_bn_next = {nullptr, 0, nullptr, nullptr, nullptr, nullptr};
extern "C" int fp_bn_stats_out_next(float* part, int64_t capacity_floats, int32_t* nblk_out) {
  g_bn_next = FpBnSink{part, capacity_floats, nblk_out, nullptr, nullptr, nullptr};
  if (nblk_out) *nblk_out = 0;
  return FP_OK;
}
extern "C" int fp_bn_bwd_out_next(float* part, int64_t capacity_floats, int32_t* nblk_out, const float* z, const float* save_mean,
                                  const float* save_invstd) {
  if (nblk_out) *nblk_out = 0;
  if (!part || !z || !save_mean || !save_invstd) {
    g_bn_next = FpBnSink{nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    return fp_set_error(FP_EINVAL, "fp_bn_bwd_out_next: null pointer");
  }
  g_bn_next = FpBnSink{part, capacity_floats, nblk_out, z, save_mean, save_invstd};
  return FP_OK;
}
FpBnSink fp_take_bn_sink() {
  FpBnSink s = g_bn_next;
  g_bn_next = FpBnSink{nullptr, 0, nullptr, nullptr, nullptr, nullptr};
  return s;
}
