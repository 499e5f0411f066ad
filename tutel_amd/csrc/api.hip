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
static int g_opt[TUTEL_OPT_COUNT] = {-2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2};  // indexed by TUTEL_OPT_*; -2 = not initialised, -1 = automatic
static const char *const g_opt_env[TUTEL_OPT_COUNT] = {"TUTEL_AMD_GEMM_IMPL", "TUTEL_AMD_GEMM_BIG", "TUTEL_AMD_DECODE", "TUTEL_AMD_EP_STAGE_GRID",
                                                       "TUTEL_AMD_GEMM_PERSIST", "TUTEL_AMD_EP_STREAMS", "TUTEL_AMD_EP_CANARY", "TUTEL_AMD_GEMM_SPLITK",
                                                       "TUTEL_AMD_GEMM_GATHER", "TUTEL_AMD_FUSED_LOCATION", "TUTEL_AMD_GEMM_STORE", "TUTEL_AMD_TIE_RULE", "TUTEL_AMD_FFN_FUSED"};

int tutel_get_option(int key) {
  if (key < 0 || key >= TUTEL_OPT_COUNT) return -1;
  if (g_opt[key] == -2) {
    const char *s = getenv(g_opt_env[key]);
    g_opt[key] = s ? atoi(s) : -1;
  }
  return g_opt[key];
}

extern "C" int tutel_amd_set_option(int key, int value) {
  TUTEL_REQUIRE(key >= 0 && key < TUTEL_OPT_COUNT, "tutel_amd_set_option: unknown key %d", key);
  TUTEL_REQUIRE(value >= -1 && value <= 7, "tutel_amd_set_option: value %d out of range", value);
  g_opt[key] = value;
  return 0;
}

// ---- stage markers: roctx ranges (rocprofv3 --marker-trace); no-ops when libroctx64 is not in the process ----------------
#include <dlfcn.h>
static thread_local int g_stage_hint = -1;  // set by callers that know which stage their next launches belong to
static int (*g_roctx_push)(const char *) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static void roctx_init() {
  static bool done = false;
  if (done) return;
  done = true;
  void *h = dlopen("libroctx64.so", RTLD_NOW | RTLD_NOLOAD);
  if (h == nullptr) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_NOLOAD);
  if (h == nullptr && getenv("TUTEL_AMD_ROCTX") != nullptr) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) return;
  g_roctx_push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
  g_roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
  if (g_roctx_push == nullptr || g_roctx_pop == nullptr) g_roctx_push = nullptr;
}
extern "C" int tutel_amd_range_push(const char *name) {
  roctx_init();
  return g_roctx_push ? g_roctx_push(name) : 0;
}
extern "C" int tutel_amd_range_pop(void) { return g_roctx_push ? g_roctx_pop() : 0; }
static const char *const g_stage_names[TUTEL_STAGE_COUNT] = {
    "tutel_amd.gate_topk", "tutel_amd.compute_location", "tutel_amd.fast_encode", "tutel_amd.expert_fc1", "tutel_amd.expert_fc2",
    "tutel_amd.fast_decode", "tutel_amd.all_to_all_dispatch", "tutel_amd.all_to_all_combine", "tutel_amd.other", "tutel_amd.gate_projection"};
void tutel_stage_range_push(int stage) {
  roctx_init();
  if (g_stage_hint >= 0) stage = g_stage_hint;
  if (g_roctx_push) g_roctx_push(g_stage_names[stage >= 0 && stage < TUTEL_STAGE_COUNT ? stage : TUTEL_STAGE_OTHER]);
}
void tutel_stage_range_pop() {
  if (g_roctx_push) g_roctx_pop();
}

// ---- per-stage timing (measurement only) -------------------------------------------------------------------
// When enabled, every C-ABI entry point that launches a kernel brackets its launch with a pair of HIP events on
// the launch stream (timing events; ~2 us of host time each, nothing on the device between kernels of one stream).
// bench.py turns it on for the timed steps and reads back per-stage totals: the HIP-event durations the roofline
// object is computed from, for every stage of the forward, with no Python between the launches.
#include <mutex>
#include <vector>
// one lock for the measurement state below (records, event pools, marks): launches may come from several host threads
// (one per GPU / stream); the lock is only ever taken while timing is switched on or a report is read
static std::mutex g_meas_mu;
struct StageRec { hipEvent_t a, b; int stage; };
static std::vector<StageRec> g_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_free;
static int g_timing = 0;

void tutel_stage_hint(int stage) { g_stage_hint = stage; }
static thread_local int g_gemm_corun = 0;
void tutel_gemm_corun_hint(int on) { g_gemm_corun = on; }
int tutel_gemm_corun() { return g_gemm_corun; }

int tutel_stage_begin(int stage, hipStream_t st) {
  if (!g_timing) return -1;
  if (g_stage_hint >= 0) stage = g_stage_hint;
  if (g_timing == 2 && stage != TUTEL_STAGE_FC1 && stage != TUTEL_STAGE_FC2) return -1;  // mode 2: the two expert GEMMs only
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return -1;  // never inside a graph capture
  std::lock_guard<std::mutex> lock(g_meas_mu);
  StageRec r;
  r.stage = stage;
  if (!g_free.empty()) {
    r.a = g_free.back().first;
    r.b = g_free.back().second;
    g_free.pop_back();
  } else if (hipEventCreateWithFlags(&r.a, hipEventReleaseToDevice) != hipSuccess ||
             hipEventCreateWithFlags(&r.b, hipEventReleaseToDevice) != hipSuccess) {
    // device-scope release: a default event makes the queue write the L2 back to system scope before it takes its
    // timestamp (5-6 us of idle GPU per record behind a GEMM that just wrote 33 MB, measured in a kernel trace)
    return -1;
  }
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}
void tutel_stage_end(int token, hipStream_t st) {
  if (token < 0) return;
  std::lock_guard<std::mutex> lock(g_meas_mu);
  if (token < (int)g_recs.size()) (void)hipEventRecord(g_recs[token].b, st);
}

// step marks: one event per call; tutel_amd_stage_report returns nothing about them, tutel_amd_marks_report the deltas
static std::vector<hipEvent_t> g_marks;
static std::vector<hipEvent_t> g_marks_free;
extern "C" int tutel_amd_mark(tutel_stream_t stream) {
  std::lock_guard<std::mutex> lock(g_meas_mu);
  hipEvent_t ev;
  if (!g_marks_free.empty()) {
    ev = g_marks_free.back();
    g_marks_free.pop_back();
  } else {
    TUTEL_REQUIRE(hipEventCreateWithFlags(&ev, hipEventReleaseToDevice) == hipSuccess, "tutel_amd_mark: cannot create an event");
  }
  TUTEL_REQUIRE(hipEventRecord(ev, (hipStream_t)stream) == hipSuccess, "tutel_amd_mark: hipEventRecord failed");
  g_marks.push_back(ev);
  return 0;
}
extern "C" int tutel_amd_marks_reserve(int n) {
  std::lock_guard<std::mutex> lock(g_meas_mu);
  while ((int)g_marks_free.size() < n) {
    hipEvent_t ev;
    TUTEL_REQUIRE(hipEventCreateWithFlags(&ev, hipEventReleaseToDevice) == hipSuccess, "tutel_amd_marks_reserve: cannot create an event");
    g_marks_free.push_back(ev);
  }
  return 0;
}
extern "C" int tutel_amd_marks_report(double *delta_us, int n) {
  std::lock_guard<std::mutex> lock(g_meas_mu);
  const int have = (int)g_marks.size();
  int m = 0;
  for (int i = 0; i + 1 < have; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_marks[i + 1]) == hipSuccess && hipEventElapsedTime(&ms, g_marks[i], g_marks[i + 1]) == hipSuccess && m < n && delta_us)
      delta_us[m++] = 1e3 * ms;
  }
  for (auto ev : g_marks) g_marks_free.push_back(ev);
  g_marks.clear();
  return m;
}

extern "C" int tutel_amd_stage_timing(int enable) {
  g_timing = enable < 0 ? 0 : enable > 2 ? 1 : enable;
  return 0;
}

extern "C" int tutel_amd_stage_report(double *total_us, int *counts, int n_stages) {
  TUTEL_REQUIRE(total_us != nullptr && counts != nullptr && n_stages >= TUTEL_STAGE_COUNT, "tutel_amd_stage_report: need arrays of %d entries", TUTEL_STAGE_COUNT);
  for (int i = 0; i < n_stages; ++i) { total_us[i] = 0.0; counts[i] = 0; }
  std::lock_guard<std::mutex> lock(g_meas_mu);
  for (auto &r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.stage >= 0 && r.stage < n_stages) {
      total_us[r.stage] += 1e3 * ms;
      counts[r.stage] += 1;
    }
    g_free.push_back({r.a, r.b});
  }
  g_recs.clear();
  return 0;
}
