// Error reporting + version for libcinema_hip.so.
#include "common.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

void cinema_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *cinema_last_error(void) { return g_err; }
extern "C" int cinema_version(void) { return 100; }  // 0.1.0
