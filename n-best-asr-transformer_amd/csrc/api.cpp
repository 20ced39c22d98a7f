// libnbest_hip.so: version + thread-local error message plumbing.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/nbest_hip.h"

static thread_local char g_err[512] = "";

void nbest_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int nbest_version(void) { return NBEST_ABI_VERSION; }

extern "C" int nbest_last_error(char* buf, size_t n) {
  if (buf && n) {
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
  }
  return (int)strlen(g_err);
}
