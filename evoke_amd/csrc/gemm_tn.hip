// gemm_tn.hip -- C[M][N] (+)= sum_k A[k][m] * B[k][n]: the weight gradient of every pointwise convolution and every linear layer of the path
// (dW = dY^T X; torch autograd of torchvision Bottleneck.conv1 / conv3 as driven by modules/visual_extractor.py:30-43, of the heads
// modules/utils_v0511.py:131-208, of the transformer linears), both operands stored [row k][channels]: K is the STRIDED index of both.
//
// gemm.hip's tile kernel runs this product with ONE workgroup per CU (16 output tiles x 16 K-slices on layer3) and ONE 64-row K step of
// loads in flight: 32 KB per CU against a loaded memory latency of 1-2 us = 28 GB/s per CU, 41 us for 94 MB of operands (MfmaUtil 14 %).
// This kernel keeps the 128 x 128 tile (the HBM-optimal point: larger tiles need more K-slices to fill 256 CUs and pay it back in slab
// traffic) and deepens the pipeline instead: 8 waves (4 x 2, each 32 x 64 = 2 x 4 MFMA tiles), three LDS stages of 64 K-rows, loads of
// steps s + 3 and s + 4 in flight in two named register sets while step s is multiplied (72 KB per CU: the depth at which
// MI355X_MICROARCH.md measures 66-73 GB/s per CU out of L2), ONE barrier per step, branch-free body.  Both operands are staged as they
// lie in memory (16 lanes x 16 B = one 256-byte row piece) at a row pitch of 288 B and the MFMA fragments come from the transposing LDS
// read (ds_read_b64_tr_b16): the eight K-rows a 32-lane half touches fall into eight different 32-byte bank groups without a swizzle.
// K-slices leave as f32 slabs [z][split][M][N] that gemm.hip's split-K reduction sums into C; with one slice the tile is added to C directly.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int NTH = 512;
constexpr int TM = 128, TN = 128, KS = 64;
constexpr int PITCH = 288;                        // 128 channels x 2 B + 32
constexpr int OP_BYTES = KS * PITCH;              // 18432: one operand of one stage
constexpr int STAGE = 2 * OP_BYTES;               // 36864
constexpr int NST = 3;
constexpr int LDS_BYTES = NST * STAGE;            // 110592

__device__ uint4 g_tn_zero16;                     // zero-initialised: what K-rows beyond the slice are loaded from

struct TnP {
  const bf16_t* A; const bf16_t* B; float* C; float* slab;
  int M, N, K;
  long lda, ldb, ldc;
  int tilesM, tilesN, ntiles, nsplit, steps_per_split, group_m;
  long slab_mn;
  int bi;
  long sAo, sAi, sBo, sBi, sCo, sCi;
  const void* zeros;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ bf16x8 frag2(const lds_char* lo_addr) {          // K-rows r .. and r + 16 ..: one 16 x 32 MFMA operand
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo_addr);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo_addr + 16 * PITCH));
  const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(NTH, 2) void gemm_tn_kernel(const TnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fq = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const int wm = wave >> 1, wn = wave & 1;                      // 32 rows of M, 64 columns of N per wave

  // blocks b, b + 8, ... share an XCD: every XCD gets a contiguous run of (z, split, tile) -- the tiles of one K-slice read the same
  // operand rows and find them in that L2; inside a slice the tiles go in super-rows of `group_m` row tiles
  const int bid = blockIdx.x, nwg = gridDim.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tile = wg % p.ntiles;
  wg /= p.ntiles;
  const int split = wg % p.nsplit, z = wg / p.nsplit;
  int tm, tn;
  {
    const int gsz = p.group_m * p.tilesN, g = tile / gsz, r = tile - g * gsz;
    const int rows = min(p.group_m, p.tilesM - g * p.group_m);
    tm = g * p.group_m + r % rows;
    tn = r / rows;
  }
  const int zo = z / p.bi, zi = z - zo * p.bi;
  const int m0 = tm * TM, n0 = tn * TN;
  const int s_begin = split * p.steps_per_split;
  const int nsteps_all = (p.K + KS - 1) / KS;
  const int ns = max(0, min(p.steps_per_split, nsteps_all - s_begin));
  const int k_begin = s_begin * KS;

  // this thread's pieces of a stage: K-rows pr and pr + 32, 16-byte chunk pc (8 channels) of the A tile and of the B tile
  const int pr = tid >> 4, pc = tid & 15;
  const char* const abase = reinterpret_cast<const char*>(p.A + zo * p.sAo + zi * p.sAi) + ((long)(k_begin + pr) * p.lda + m0 + pc * 8) * 2;
  const char* const bbase = reinterpret_cast<const char*>(p.B + zo * p.sBo + zi * p.sBi) + ((long)(k_begin + pr) * p.ldb + n0 + pc * 8) * 2;
  const long astep = (long)KS * p.lda * 2, bstep = (long)KS * p.ldb * 2;
  const long a32 = 32L * p.lda * 2, b32 = 32L * p.ldb * 2;
  const char* const zsrc = reinterpret_cast<const char*>(p.zeros);
  const int krows = p.K - k_begin - pr;           // K-row pr of step s holds data while s * 64 < krows; row pr + 32 while s * 64 + 32 < krows
  const int pdst = pr * PITCH + pc * 16;

  // two named register sets: the loads of steps s + 3 and s + 4 are in flight while step s is multiplied
  uint4 xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
#define EVK_TN_LOAD(a0, a1, b0, b1, step)                                                           \
  {                                                                                                 \
    const int st_ = min((step), ns - 1);                                                            \
    const bool v0_ = st_ * KS < krows, v1_ = st_ * KS + 32 < krows;                                  \
    const char* const ap_ = abase + st_ * astep;                                                    \
    const char* const bp_ = bbase + st_ * bstep;                                                    \
    a0 = *reinterpret_cast<const uint4*>(v0_ ? ap_ : zsrc);                                         \
    a1 = *reinterpret_cast<const uint4*>(v1_ ? ap_ + a32 : zsrc);                                   \
    b0 = *reinterpret_cast<const uint4*>(v0_ ? bp_ : zsrc);                                         \
    b1 = *reinterpret_cast<const uint4*>(v1_ ? bp_ + b32 : zsrc);                                   \
  }
#define EVK_TN_STORE(a0, a1, b0, b1, off)                                                           \
  {                                                                                                 \
    char* const d_ = smem + (off) + pdst;                                                           \
    *reinterpret_cast<uint4*>(d_) = a0;                                                             \
    *reinterpret_cast<uint4*>(d_ + 32 * PITCH) = a1;                                                \
    *reinterpret_cast<uint4*>(d_ + OP_BYTES) = b0;                                                  \
    *reinterpret_cast<uint4*>(d_ + OP_BYTES + 32 * PITCH) = b1;                                     \
  }

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (ns > 0) {
    // prologue: steps 0 .. 3 are requested before the first wait (one memory round trip)
    {
      uint4 ta0, ta1, tb0, tb1, ua0, ua1, ub0, ub1;
      EVK_TN_LOAD(ta0, ta1, tb0, tb1, 0)
      EVK_TN_LOAD(ua0, ua1, ub0, ub1, 1)
      EVK_TN_LOAD(xa0, xa1, xb0, xb1, 2)
      EVK_TN_LOAD(ya0, ya1, yb0, yb1, 3)
      EVK_TN_STORE(ta0, ta1, tb0, tb1, 0)
      EVK_TN_STORE(ua0, ua1, ub0, ub1, STAGE)
    }
    __syncthreads();

    const lds_char* const L = (const lds_char*)(uintptr_t)(uint32_t)(uintptr_t)smem;
    // fragment addresses inside a stage (first 32-row half): A tile rows of M at wm * 64 B ..., B tile columns of N at wn * 128 B ...
    const int fro = (4 * fq + q4) * PITCH + pp * 8;
    const int aoff = fro + wm * 64;                        // + i * 32 (i-th 16-channel tile), + 32 * PITCH for the second half
    const int boff = OP_BYTES + fro + wn * 128;            // + j * 32
    bf16x8 a0[2], b0[4], a1[2], b1[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a0[i] = frag2(L + aoff + i * 32);
#pragma unroll
    for (int j = 0; j < 4; ++j) b0[j] = frag2(L + boff + j * 32);

    int cur = 0, nxt = STAGE, wr = 2 * STAGE;
#define EVK_TN_BODY(SA0, SA1, SB0, SB1, s)                                                          \
    {                                                                                               \
      EVK_TN_STORE(SA0, SA1, SB0, SB1, wr)          /* step s + 2 -> its stage */                    \
      EVK_TN_LOAD(SA0, SA1, SB0, SB1, (s) + 4)      /* this register set is free again */            \
      const lds_char* const Sc = L + cur + 32 * PITCH;                                              \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) a1[i] = frag2(Sc + aoff + i * 32);              \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) b1[j] = frag2(Sc + boff + j * 32);              \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = EVK_MFMA_16x16x32(b0[j], a0[i], acc[i][j], 0, 0, 0); \
      const lds_char* const Sn = L + nxt;           /* after the last step: stale data, never used */ \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) a0[i] = frag2(Sn + aoff + i * 32);              \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) b0[j] = frag2(Sn + boff + j * 32);              \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = EVK_MFMA_16x16x32(b1[j], a1[i], acc[i][j], 0, 0, 0); \
      __syncthreads();                                                                              \
      const int t_ = cur; cur = nxt; nxt = wr; wr = t_;                                             \
    }
    int s = 0;
    for (; s + 1 < ns; s += 2) {
      EVK_TN_BODY(xa0, xa1, xb0, xb1, s)
      EVK_TN_BODY(ya0, ya1, yb0, yb1, s + 1)
    }
    if (s < ns) EVK_TN_BODY(xa0, xa1, xb0, xb1, s)
#undef EVK_TN_BODY
  }
#undef EVK_TN_LOAD
#undef EVK_TN_STORE

  // ---- epilogue: lane holds C[m][n .. n + 3], m = m0 + wm * 32 + i * 16 + (lane & 15), n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4) ----
  const int m = m0 + wm * 32 + (lane & 15);
  const int n = n0 + wn * 64 + fq * 4;
  if (p.slab) {
    float* const sb = p.slab + ((long)z * p.nsplit + split) * p.slab_mn + (long)m * p.N + n;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4*>(sb + (long)i * 16 * p.N + j * 16) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  } else {
    float* const cb = p.C + zo * p.sCo + zi * p.sCi + (long)m * p.ldc + n;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4* const q = reinterpret_cast<float4*>(cb + (long)i * 16 * p.ldc + j * 16);
        float4 t = *q;
        t.x += acc[i][j][0]; t.y += acc[i][j][1]; t.z += acc[i][j][2]; t.w += acc[i][j][3];
        *q = t;
      }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Shape / layout test shared with gemm.hip's router: whole 128 x 128 tiles, 16-byte aligned rows and batch strides.
bool evk_gemm_tn_supported(int M, int N, int K, long lda, long ldb, long ldc, long sAo, long sAi, long sBo, long sBi, long sCo, long sCi) {
  static const int on = evk_tunable("EVK_GEMM_TN", 1);
  if (!on) return false;
  if (M < TM || N < TN || (M % TM) || (N % TN) || K < 1) return false;
  if ((lda % 8) || (ldb % 8) || (ldc % 4) || lda < M || ldb < N || ldc < N) return false;
  if ((sAo % 8) || (sAi % 8) || (sBo % 8) || (sBi % 8) || (sCo % 4) || (sCi % 4)) return false;
  return true;
}

// C (+)= A^T B in `nsplit` K-slices of `steps_per_split` 64-row steps; slab = null (nsplit must be 1: the tile is added to C) or the
// f32 slabs [batch][nsplit][M][N] that the caller sums into C afterwards (evk_splitk_reduce_launch).
int evk_gemm_tn_launch(const void* A, const void* B, float* C, float* slab, int M, int N, int K, long lda, long ldb, long ldc, int nsplit,
                       int steps_per_split, int batch, int bi, long sAo, long sAi, long sBo, long sBi, long sCo, long sCi, hipStream_t s) {
  EVK_REQUIRE(A && B && C && al16(A) && al16(B) && al16(C) && (!slab || al16(slab)), "gemm_tn: null / misaligned operand");
  EVK_REQUIRE(evk_gemm_tn_supported(M, N, K, lda, ldb, ldc, sAo, sAi, sBo, sBi, sCo, sCi), "gemm_tn: unsupported shape M=%d N=%d K=%d", M, N, K);
  EVK_REQUIRE(nsplit >= 1 && steps_per_split >= 1 && (long)nsplit * steps_per_split * KS >= K && (slab || nsplit == 1) && batch >= 1 && bi >= 1,
              "gemm_tn: bad K-slicing (nsplit %d x %d steps for K = %d)", nsplit, steps_per_split, K);
  TnP p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.slab = slab;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.tilesM = M / TM; p.tilesN = N / TN; p.ntiles = p.tilesM * p.tilesN;
  p.nsplit = nsplit; p.steps_per_split = steps_per_split; p.slab_mn = (long)M * N;
  static const int group_m = evk_tunable("EVK_TN_GROUP_M", 4);
  p.group_m = group_m < 1 ? 1 : group_m;
  p.bi = bi; p.sAo = sAo; p.sAi = sAi; p.sBo = sBo; p.sBi = sBi; p.sCo = sCo; p.sCi = sCi;
  const long nwg = (long)p.ntiles * nsplit * batch;
  EVK_REQUIRE(nwg < (1L << 31), "gemm_tn: grid too large");
  static EvkDeviceOnce zeros_once;          // per device: the symbol's address and the kernel attribute belong to one GPU
  p.zeros = zeros_once.get([]() -> void* {
    void* z = nullptr;
    if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_tn_zero16)) != hipSuccess) z = nullptr;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    return z;
  });
  EVK_REQUIRE(p.zeros, "gemm_tn: no address for the zero block");
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)nwg), dim3(NTH), LDS_BYTES, s, p);
  return evk_check_launch("gemm_tn_kernel");
}
