// runtime.hip -- error reporting + HIP-event profiling hooks of libevoke_hip.so
#include <stdarg.h>
#include <mutex>
#include <vector>
#include "common.h"

static thread_local char g_err[512] = "";

void evk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int evk_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    evk_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return EVK_ELAUNCH;
  }
  return EVK_OK;
}

struct ProfRec { int fam; hipEvent_t a, b; double flops; };
static std::mutex g_mu;
static bool g_prof = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static thread_local hipEvent_t g_open = nullptr;

static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void evk_prof_begin(int family, hipStream_t s) {
  if (!g_prof) return;
  std::lock_guard<std::mutex> lk(g_mu);
  g_open = get_event();
  if (g_open) (void)hipEventRecord(g_open, s);
}

void evk_prof_end(int family, hipStream_t s, double flops) {
  if (!g_prof || !g_open) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t b = get_event();
  if (!b) return;
  (void)hipEventRecord(b, s);
  g_recs.push_back({family, g_open, b, flops});
  g_open = nullptr;
}

extern "C" {

int evk_version(void) { return 100; }
const char* evk_last_error(void) { return g_err; }

int evk_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof = on != 0;
  return EVK_OK;
}

int evk_prof_collect(double* ms, int64_t* launches, double* flops_gemm) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < EVK_FAM_COUNT; ++i) { ms[i] = 0.0; launches[i] = 0; }
  double fl = 0.0;
  for (auto& r : g_recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
      ms[r.fam] += t;
      launches[r.fam] += 1;
      if (r.fam == EVK_FAM_GEMM) fl += r.flops;
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  if (flops_gemm) *flops_gemm = fl;
  return EVK_OK;
}

}  // extern "C"
