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

