// Library-level entry points of libvc_hip: version, last-error string.
#include <stdarg.h>

#include "vc_common.h"

namespace {
thread_local char g_err[512] = "";
}

void vc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Compute units of the current device (the bound on how many one-per-CU workgroups can be resident at once);
// 256 (MI355X) when no device can be asked, e.g. on a build host.
int vc_device_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      n <= 0) {
    (void)hipGetLastError();
    return 256;
  }
  return n;
}

extern "C" {

int vc_version(void) { return 101; }

const char* vc_last_error(void) { return g_err; }

}  // extern "C"
