// common.h -- shared device helpers and host-side launch plumbing for libevoke_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include "../../include/evoke_hip.h"

// 16-bit storage format of activations, weight shadows and gradients.  The DEFAULT build stores IEEE fp16 (libevoke_hip.so:
// 11-bit mantissa, eight times less rounding noise per stored tensor than bf16 -- what the north star's 1e-3 loss parity needs);
// its backward runs under the dynamic loss scale of eltwise.hip (evk_grad_nonfinite / evk_optim_step_dyn /
// evk_loss_scale_update).  -DEVK_STORE_BF16 builds the SAME kernels over bf16 storage (libevoke_hip_bf16.so: fp32's exponent
// range, no loss scaling, 3.5e-3 loss parity).  Every kernel converts through the helpers below and multiplies through
// EVK_MFMA_16x16x32, so the type names keep their historical "bf" spelling in both builds.
#if !defined(EVK_STORE_BF16) && !defined(EVK_STORE_F16)
#define EVK_STORE_F16 1
#endif
typedef unsigned short bf16_t;  // raw bits of one stored value
typedef __attribute__((ext_vector_type(4))) float f32x4;
#ifdef EVK_STORE_F16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
#define EVK_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f2bf(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(bf16_t, h); }   // RNE
__device__ __forceinline__ float lo_bf(uint32_t v) { return bf2f((bf16_t)(v & 0xffffu)); }
__device__ __forceinline__ float hi_bf(uint32_t v) { return bf2f((bf16_t)(v >> 16)); }
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define EVK_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// plain cast -> v_cvt_pk_bf16_f32 (RNE, NaN stays NaN: MI355X_MICROARCH.md "Correctness boundaries")
__device__ __forceinline__ bf16_t f2bf(float f) { __bf16 b = (__bf16)f; return __builtin_bit_cast(bf16_t, b); }
__device__ __forceinline__ float lo_bf(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi_bf(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
#endif
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

// Dropout seeds are kernel arguments, i.e. frozen when a launch sequence is captured in a HIP graph.  Every seeded kernel adds
// the device-resident "seed epoch" (evk_set_seed_epoch: one uint64 the host advances once per training step, by a device op
// that is part of the captured step) so that a replayed graph draws fresh masks while forward and backward of ONE step agree.
const unsigned long long* evk_seed_epoch_ptr();
__device__ __forceinline__ unsigned long long evk_mix_seed(unsigned long long seed, const unsigned long long* epoch) {
  return epoch ? seed + epoch[0] * 0xD6E8FEB86659FD93ULL : seed;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case EVK_ACT_RELU: return fmaxf(x, 0.f);
    case EVK_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    case EVK_ACT_TANH: return tanhf(x);
    case EVK_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
    case EVK_ACT_GELU_NEW: return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
    default: return x;
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
void evk_set_error(const char* fmt, ...);
#define EVK_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      evk_set_error(__VA_ARGS__);             \
      return EVK_EINVAL;                      \
    }                                         \
  } while (0)

// Experiment switches.  Every kernel route has ONE measured default; the EVK_* environment variables read through evk_tunable() select the
// alternatives that were measured against it (DESIGN.md names each).  They are inert unless EVK_EXPERIMENTAL=1 is set as well, so a stray
// variable in a production environment cannot move a launch onto a route the test-suite does not cover (runtime.hip).
int evk_tunable(const char* name, int dflt);

// First-use initialisation that belongs to ONE device (hipFuncSetAttribute of a kernel's dynamic-LDS limit, hipGetSymbolAddress of a
// __device__ block), safe when several host threads enter the library at once (autograd's backward threads, one thread per search in flight,
// the encoder thread): one std::once_flag per device ordinal; init() returns the pointer to cache (any non-null value when there is none).
constexpr int EVK_MAX_DEVICES = 16;
struct EvkDeviceOnce {
  std::once_flag flag[EVK_MAX_DEVICES];
  void* value[EVK_MAX_DEVICES] = {};
  template <class F>
  void* get(F&& init) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= EVK_MAX_DEVICES) dev = 0;
    std::call_once(flag[dev], [&] { value[dev] = init(); });
    return value[dev];
  }
};
#define EVK_DYN_LDS_ONCE(kern, bytes)                                                                                              \
  do {                                                                                                                             \
    static EvkDeviceOnce once_;                                                                                                    \
    once_.get([&]() -> void* {                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));    \
      return reinterpret_cast<void*>(1);                                                                                           \
    });                                                                                                                            \
  } while (0)

// runtime.hip: true for streams made by evk_stream_create_cu_mask (kernels that need their whole grid resident at once must not run there)
bool evk_stream_is_cu_masked(hipStream_t s);

// Capture probe (replay.hip): while a training step is being stream-captured for the step replayer, every launch of the library notes which
// HIP stream created which graph node (hipStreamGetCaptureInfo_v2 right after the launch: the stream's dependency set is the new node), so
// that the replayer's lanes ARE the capture's streams.  One predictable branch per launch when the probe is off.
extern bool g_evk_capture_probe;
void evk_capture_note(hipStream_t s);
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)            \
  do {                                                                                               \
    (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);             \
    if (g_evk_capture_probe) evk_capture_note(streamId);                                             \
  } while (0)

// profiling: an event pair around each launch when enabled (bench.py roofline leg)
void evk_prof_begin(int family, hipStream_t s);
void evk_prof_end(int family, hipStream_t s, double flops);
int evk_check_launch(const char* what);
void evk_prof_tag(int a, int b, int c, int d, int e, int f);   // attaches a shape tag to the next profiled launch

struct ProfScope {
  int fam; hipStream_t s; double flops;
  ProfScope(int f, hipStream_t st, double fl = 0.0) : fam(f), s(st), flops(fl) { evk_prof_begin(fam, s); }
  ~ProfScope() { evk_prof_end(fam, s, flops); }
};

// gemm.hip: C[z][m][n] += sum over `splitk` f32 slabs [z][split][M][N] (z = zo * bi + zi -> C + zo * sCo + zi * sCi + m * ldc + n), N % 4 == 0
int evk_splitk_reduce_launch(const float* slab, float* C, long mn, int M, int N, int splitk, int bi, long ldc, long sCo, long sCi, int batch, hipStream_t s);

// gemm_tn.hip: C[M][N] (+)= A[k][m]^T B[k][n] (both operands K-strided: weight gradients), deep-pipelined 128 x 128 tiles; K-slices leave
// as f32 slabs [batch][nsplit][M][N] (summed by evk_splitk_reduce_launch) or, with one slice and slab == null, are added to C directly
bool evk_gemm_tn_supported(int M, int N, int K, long lda, long ldb, long ldc, long sAo, long sAi, long sBo, long sBi, long sCo, long sCi);
int evk_gemm_tn_launch(const void* A, const void* B, float* C, float* slab, int M, int N, int K, long lda, long ldb, long ldc, int nsplit,
                       int steps_per_split, int batch, int bi, long sAo, long sAi, long sBo, long sBi, long sCo, long sCi, hipStream_t s);

static inline int ilog2_exact(int64_t v) {
  int l = 0;
  while ((int64_t(1) << l) < v) ++l;
  return ((int64_t(1) << l) == v) ? l : -1;
}
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
