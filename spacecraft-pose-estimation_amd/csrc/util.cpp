// Error plumbing of the C ABI: thread-local message, no exceptions across the boundary; and the gate in front of
// the development switches.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

namespace scpose {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* last_error() { return g_err; }

// The tuning / ablation switches (tools_dev/README.md) are read through this function only.  They exist for the
// developer scripts; some of them skip work and give wrong results.  A production process must not be steerable by a
// stray SCPOSE_* variable, so they are ignored unless SCPOSE_DEV=1 is set as well, which is announced on stderr.
const char* dev_env(const char* name) {
  static const bool enabled = [] {
    const char* e = getenv("SCPOSE_DEV");
    const bool on = e && atoi(e) != 0;
    if (on) fprintf(stderr, "[scpose] SCPOSE_DEV=1: development switches (SCPOSE_*) are honoured; results and timings may differ from the production path\n");
    return on;
  }();
  return enabled ? getenv(name) : nullptr;
}

int32_t lds_opt_in(const void* kernel, int bytes, LdsOptIn* memo) {
  int dev = 0;
  SCP_CHECK_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 16 && memo->done[dev]) return SCPOSE_OK;
  SCP_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (dev >= 0 && dev < 16) memo->done[dev] = true;
  return SCPOSE_OK;
}

}  // namespace scpose
