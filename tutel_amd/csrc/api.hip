// api.hip -- library info + error reporting for libtutel_amd.so.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void tutel_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int tutel_amd_abi_version(void) { return TUTEL_AMD_ABI_VERSION; }
extern "C" const char *tutel_amd_target_arch(void) { return "gfx950"; }
extern "C" const char *tutel_amd_last_error(void) { return g_err; }
