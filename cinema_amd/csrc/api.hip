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

int cinema_dev() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return d < CINEMA_MAX_DEVICES ? d : CINEMA_MAX_DEVICES - 1;
}

int cinema_cu_count() {
  static std::atomic<int> ncu[CINEMA_MAX_DEVICES];
  const int d = cinema_dev();
  int n = ncu[d].load(std::memory_order_relaxed);
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    ncu[d].store(n, std::memory_order_relaxed);
  }
  return n;
}
