// Error reporting and version of the C ABI (include/vsseg_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/vsseg_hip.h"

static thread_local char g_err[512] = "";

extern "C" void vsseg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* vsseg_last_error(void) { return g_err; }
extern "C" int vsseg_version(void) { return 1; }
