// Error plumbing + ABI version for libdbir_hip.so (see include/dbir.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void dbir_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dbir_last_error(void) { return g_err; }
extern "C" int dbir_abi_version(void) { return 2; }
