// Error reporting of the C ABI (thread-local message, never throws, never allocates on the device).
#include "common.hpp"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void pero_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pero_last_error(void) { return g_err; }
extern "C" int pero_abi_version(void) { return 2; }  // 2: pero_gemm takes a caller-owned workspace (round 3)
