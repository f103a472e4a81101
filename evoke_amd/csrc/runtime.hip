// runtime.hip -- error reporting + HIP-event profiling hooks of libevoke_hip.so
#include <stdarg.h>
#include <stdlib.h>
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

// The ONE place the library reads its environment.  `name` selects a measured alternative to a kernel route's default (or a probe / debug
// print); the variable counts only under EVK_EXPERIMENTAL=1, which the tests that force a route set for their child process.
int evk_tunable(const char* name, int dflt) {
  static const bool on = [] { const char* e = getenv("EVK_EXPERIMENTAL"); return e && atoi(e) != 0; }();
  if (!on) return dflt;
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

int evk_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    evk_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return EVK_ELAUNCH;
  }
  return EVK_OK;
}

struct ProfRec { int fam; hipEvent_t a, b; double flops; int tag[6]; };
static thread_local int g_tag[6] = {0, 0, 0, 0, 0, 0};
void evk_prof_tag(int a, int b, int c, int d, int e, int f) { g_tag[0] = a; g_tag[1] = b; g_tag[2] = c; g_tag[3] = d; g_tag[4] = e; g_tag[5] = f; }
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
  g_recs.push_back({family, g_open, b, flops, {g_tag[0], g_tag[1], g_tag[2], g_tag[3], g_tag[4], g_tag[5]}});
  for (int i = 0; i < 6; ++i) g_tag[i] = 0;
  g_open = nullptr;
}

static std::vector<hipStream_t> g_masked_streams;          // guarded by g_mu
bool evk_stream_is_cu_masked(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (hipStream_t m : g_masked_streams) if (m == s) return true;
  return false;
}

static const unsigned long long* g_seed_epoch = nullptr;
const unsigned long long* evk_seed_epoch_ptr() { return g_seed_epoch; }

extern "C" {

int evk_set_seed_epoch(const uint64_t* epoch_dev) {
  g_seed_epoch = reinterpret_cast<const unsigned long long*>(epoch_dev);
  return EVK_OK;
}

int evk_version(void) { return 101; }

/* A HIP stream whose kernels may only run on the compute units whose bit is set in `mask` (bit i of word i / 32; `words` x 32 bits, at
 * least the device's CU count).  The serving loop gives the ENCODER stream such a mask: a convolution grid otherwise occupies every CU, and
 * the ~10 us kernels of the searches in flight then wait for a workgroup slot behind 50-100 us tiles (measured: 4-5 x their lone
 * duration).  The stream lives until evk_stream_destroy.  *stream_out is a hipStream_t. */
int evk_stream_create_cu_mask(const uint32_t* mask, int32_t words, void** stream_out) {
  EVK_REQUIRE(mask && stream_out && words > 0, "stream_create_cu_mask: bad args");
  hipStream_t s = nullptr;
  const hipError_t err = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
  if (err != hipSuccess) { evk_set_error("stream_create_cu_mask: %s", hipGetErrorString(err)); return EVK_ELAUNCH; }
  *stream_out = s;
  { std::lock_guard<std::mutex> lk(g_mu); g_masked_streams.push_back(s); }
  return EVK_OK;
}

int evk_stream_destroy(void* stream) {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_masked_streams.size(); ++i)
      if (g_masked_streams[i] == reinterpret_cast<hipStream_t>(stream)) { g_masked_streams.erase(g_masked_streams.begin() + i); break; }
  }
  if (stream && hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)) != hipSuccess) { evk_set_error("stream_destroy failed"); return EVK_ELAUNCH; }
  return EVK_OK;
}

int evk_device_cu_count(int32_t device, int32_t* cus) {
  EVK_REQUIRE(cus, "device_cu_count: bad args");
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, device) != hipSuccess) { evk_set_error("device_cu_count: hipGetDeviceProperties failed"); return EVK_ELAUNCH; }
  *cus = pr.multiProcessorCount;
  return EVK_OK;
}
int evk_storage_format(void) {
#ifdef EVK_STORE_F16
  return 16;
#else
  return 0;
#endif
}
const char* evk_last_error(void) { return g_err; }

int evk_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof = on != 0;
  return EVK_OK;
}

static const char* g_dump_path = nullptr;
int evk_prof_dump_to(const char* path) { g_dump_path = path; return EVK_OK; }

int evk_prof_collect(double* ms, int64_t* launches, double* flops_gemm) {
  std::lock_guard<std::mutex> lk(g_mu);
  FILE* df = g_dump_path ? fopen(g_dump_path, "w") : nullptr;
  if (df) fprintf(df, "family,ms,flops,M,N,K,batch,a_mode,b_mode\n");
  for (int i = 0; i < EVK_FAM_COUNT; ++i) { ms[i] = 0.0; launches[i] = 0; }
  double fl = 0.0;
  for (auto& r : g_recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
      ms[r.fam] += t;
      launches[r.fam] += 1;
      if (r.fam == EVK_FAM_GEMM) fl += r.flops;
      if (df) fprintf(df, "%d,%.6f,%.0f,%d,%d,%d,%d,%d,%d\n", r.fam, t, r.flops, r.tag[0], r.tag[1], r.tag[2], r.tag[3], r.tag[4], r.tag[5]);
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  if (df) fclose(df);
  if (flops_gemm) *flops_gemm = fl;
  return EVK_OK;
}

}  // extern "C"
