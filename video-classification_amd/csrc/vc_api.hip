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

extern "C" {

int vc_version(void) { return 100; }

const char* vc_last_error(void) { return g_err; }

}  // extern "C"
