// api.hip -- library info + error reporting for libtutel_amd.so.
#include <stdarg.h>
#include <stdlib.h>

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

// ---- tuning knobs (A/B runs and tests; defaults come from the environment once) -----------------
static int g_opt[3] = {-2, -2, -2};  // TUTEL_OPT_GEMM_IMPL, TUTEL_OPT_GEMM_TILE; -2 = not initialised, -1 = automatic
static const char *const g_opt_env[3] = {"TUTEL_AMD_GEMM_IMPL", "TUTEL_AMD_GEMM_BIG", "TUTEL_AMD_GEMM_ABL"};

int tutel_get_option(int key) {
  if (key < 0 || key > 2) return -1;
  if (g_opt[key] == -2) {
    const char *s = getenv(g_opt_env[key]);
    g_opt[key] = s ? atoi(s) : -1;
  }
  return g_opt[key];
}

extern "C" int tutel_amd_set_option(int key, int value) {
  TUTEL_REQUIRE(key >= 0 && key <= 2, "tutel_amd_set_option: unknown key %d", key);
  TUTEL_REQUIRE(value >= -1 && value <= (key == 2 ? 511 : 4), "tutel_amd_set_option: value %d out of range", value);
  g_opt[key] = value;
  return 0;
}
