// gemm_strip.hip -- C[M][N] = A[M][K] . B[N][K]^T for the tall products of the trunk's CONTRACTING pointwise convolutions
// (torchvision Bottleneck.conv1 forward, 4 planes -> planes, and the data gradient of conv3 over transposed weights; the reference
// drives them through modules/visual_extractor.py:30-38), 16-bit operands, f32 accumulation on v_mfma_f32_16x16x32.
//
// gemm.hip's 128 x 128 tile fills 32 KB of LDS per 2 * 128 * 128 * 64 flop and runs 2.25 tiles per CU on layer3 (M = 36864, N = 256):
// three rounds of blocks on the unlucky CUs, every one of them bound by its LDS-fill rate.  This kernel gives a workgroup a strip of
// 288 rows x 128 columns (1.4 x the flop per filled byte) -- layer3 is then exactly 128 x 2 = 256 workgroups, one per CU -- and runs
// the pipeline of conv3x3.hip: 8 waves (2 x 4, each 144 rows x 32 columns = 9 x 2 MFMA tiles), three LDS stages per operand (rows of
// 128 B = one 64-deep K step, 16-byte chunk index XOR (row & 7)), register-staged loads issued three steps ahead and written two steps
// ahead, MFMA fragments read half a step ahead of their use, ONE barrier per K step and nothing after it that the next MFMAs wait for.
// The loop body is branch-free: loads beyond the last step re-read it and land in a stage nobody reads any more.
// Epilogues as in conv3x3.hip: per-column sum / sum of squares partials (batch-norm statistics of a forward convolution), or residual +
// ReLU gate + gate statistics (data gradient).  Rows beyond M are loaded from a block of zeros and masked on the way out.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int NTH = 512;
constexpr int WM = 2, WN = 4, MI = 9, NI = 2;
constexpr int TM = 16 * MI * WM;             // 288 rows per workgroup
constexpr int TN = 16 * NI * WN;             // 128 columns per workgroup
constexpr int A_BYTES = TM * 128, B_BYTES = TN * 128;
constexpr int NST = 3;
constexpr int STAGE = A_BYTES + B_BYTES;     // 53248
constexpr int LDS_BYTES = NST * STAGE;       // 159744
constexpr int NPA = (TM * 8 + NTH - 1) / NTH;   // 16-byte A pieces per thread and step (5; the last one covers rows 256..287 only)
static_assert(TM % 8 == 0 && (16 * MI) % 8 == 0, "fragment rows keep their (row & 7) across MFMA tiles");

__device__ uint4 g_zero16;                   // zero-initialised: what rows beyond M are loaded from

struct StP {
  const bf16_t* A; const bf16_t* B; bf16_t* C;
  int M, N, K;
  long lda, ldb, ldc;
  int tilesN;
  float* colstats;                 // [tilesM * WM][2][N] or null
  const bf16_t* resid; long ldr;
  const bf16_t* gate; long ldg;
  float* gatestats;                // [tilesM * WM][2][N] or null (needs gate)
  const void* zeros;
  unsigned kmul;                   // 128; 0 = timing probe (EVK_STRIP_PROBE=1): every load re-reads K step 0 (cache hits, wrong results)
  const float* scale; const float* bias; int relu;     // inference (eval-mode batch norm): C = relu?((A . B^T) * scale[n] + bias[n] (+ resid)); no gate, no statistics
};

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
  return v + __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float row16_sum(float v) {      // sum over the 16 lanes sharing lane >> 4
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  return dpp_add<0x140>(v);
}

__global__ __launch_bounds__(NTH, 2) void gemm_strip_kernel(const StP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int frow = lane & 15, fq = lane >> 4;

  // blocks b, b + 8, ... share an XCD: contiguous runs of tiles per XCD, the column tiles of one row strip adjacent (they read the same
  // A strip: the later ones find it in that L2)
  const int bid = blockIdx.x, nwg = gridDim.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tm = wg / p.tilesN, tn = wg - tm * p.tilesN;
  const int row0 = tm * TM, col0 = tn * TN;
  const int ns = p.K >> 6;

  // this thread's pieces: tile row (tid >> 3) + 64 i, 16-byte chunk tid & 7 of the 64-deep K step
  const int pr = tid >> 3, pc = tid & 7;
  const char* ap[NPA];
  unsigned amsk[NPA];                 // all ones where the piece holds data (its pointer advances with K), zero for the zero block
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int r = pr + 64 * i;
    const bool valid = r < TM && row0 + r < p.M;
    ap[i] = valid ? reinterpret_cast<const char*>(p.A) + ((long)(row0 + r) * p.lda + pc * 8) * 2 : reinterpret_cast<const char*>(p.zeros);
    amsk[i] = valid ? 0xffffffffu : 0u;
  }
  const char* const bp0 = reinterpret_cast<const char*>(p.B) + ((long)(col0 + pr) * p.ldb + pc * 8) * 2;
  const long brow64 = 64L * p.ldb * 2;
  const int pdst = pr * 128 + ((pc ^ (pr & 7)) << 4);        // + i * 8192; B pieces: + A_BYTES
  const bool last_piece_in = pr + 64 * (NPA - 1) < TM;

  // the staged pieces are named registers (an array captured by a lambda ends up in scratch memory, and the store to scratch waits
  // for every load at once)
  static_assert(NPA == 5, "the staging registers are spelled out for five A pieces");
  uint4 ra0, ra1, ra2, ra3, ra4, rb0, rb1;
#define EVK_ST_A(M) M(0, ra0) M(1, ra1) M(2, ra2) M(3, ra3) M(4, ra4)
  auto load = [&](int s) {
    const unsigned koff = (unsigned)min(s, ns - 1) * p.kmul;
#define EVK_ST_LD(i, r) r = *reinterpret_cast<const uint4*>(ap[i] + (koff & amsk[i]));
    EVK_ST_A(EVK_ST_LD)
#undef EVK_ST_LD
    rb0 = *reinterpret_cast<const uint4*>(bp0 + koff);
    rb1 = *reinterpret_cast<const uint4*>(bp0 + brow64 + koff);
  };
  auto store = [&](int stage_off) {
    char* d = smem + stage_off + pdst;
#define EVK_ST_ST(i, r) if (i + 1 < NPA || last_piece_in) *reinterpret_cast<uint4*>(d + i * 8192) = r;
    EVK_ST_A(EVK_ST_ST)
#undef EVK_ST_ST
    *reinterpret_cast<uint4*>(d + A_BYTES) = rb0;
    *reinterpret_cast<uint4*>(d + A_BYTES + 8192) = rb1;
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a stage: MFMA tile im of this wave is 16 rows = 2048 B further, same (row & 7)
  const int arow = wm * (16 * MI) + frow, brow = wn * (16 * NI) + frow;
  const int a0 = arow * 128 + ((fq ^ (arow & 7)) << 4);
  const int b0 = A_BYTES + brow * 128 + ((fq ^ (brow & 7)) << 4);

  // prologue: steps 0, 1 and 2 are requested before the first wait (one memory round trip, not three)
  {
    uint4 t0[NPA + 2], t1[NPA + 2];
    const unsigned k1 = (unsigned)min(1, ns - 1) * p.kmul;
#pragma unroll
    for (int i = 0; i < NPA; ++i) { t0[i] = *reinterpret_cast<const uint4*>(ap[i]); t1[i] = *reinterpret_cast<const uint4*>(ap[i] + (k1 & amsk[i])); }
    t0[NPA] = *reinterpret_cast<const uint4*>(bp0); t0[NPA + 1] = *reinterpret_cast<const uint4*>(bp0 + brow64);
    t1[NPA] = *reinterpret_cast<const uint4*>(bp0 + k1); t1[NPA + 1] = *reinterpret_cast<const uint4*>(bp0 + brow64 + k1);
    load(2);
    char* d = smem + pdst;
#pragma unroll
    for (int i = 0; i < NPA; ++i)
      if (i + 1 < NPA || last_piece_in) { *reinterpret_cast<uint4*>(d + i * 8192) = t0[i]; *reinterpret_cast<uint4*>(d + STAGE + i * 8192) = t1[i]; }
    *reinterpret_cast<uint4*>(d + A_BYTES) = t0[NPA];
    *reinterpret_cast<uint4*>(d + A_BYTES + 8192) = t0[NPA + 1];
    *reinterpret_cast<uint4*>(d + STAGE + A_BYTES) = t1[NPA];
    *reinterpret_cast<uint4*>(d + STAGE + A_BYTES + 8192) = t1[NPA + 1];
  }
  __syncthreads();

  bf16x8 af0[MI], bf0[NI], af1[MI], bf1[NI];
#pragma unroll
  for (int im = 0; im < MI; ++im) af0[im] = *reinterpret_cast<const bf16x8*>(smem + a0 + im * 2048);
#pragma unroll
  for (int in = 0; in < NI; ++in) bf0[in] = *reinterpret_cast<const bf16x8*>(smem + b0 + in * 2048);

  int cur = 0, nxt = STAGE, wr = 2 * STAGE;          // byte offsets of the stages of step s, s + 1, s + 2
  for (int s = 0; s < ns; ++s) {
    store(wr);                                       // registers -> stage of step s + 2, then the loads of step s + 3
    load(s + 3);
    const char* const Sc = smem + cur;
#pragma unroll
    for (int im = 0; im < MI; ++im) af1[im] = *reinterpret_cast<const bf16x8*>(Sc + (a0 ^ 64) + im * 2048);
#pragma unroll
    for (int in = 0; in < NI; ++in) bf1[in] = *reinterpret_cast<const bf16x8*>(Sc + (b0 ^ 64) + in * 2048);
#pragma unroll
    for (int in = 0; in < NI; ++in)
#pragma unroll
      for (int im = 0; im < MI; ++im) acc[in][im] = EVK_MFMA_16x16x32(bf0[in], af0[im], acc[in][im], 0, 0, 0);
    const char* const Sn = smem + nxt;               // after the last step: stale data, never used
#pragma unroll
    for (int im = 0; im < MI; ++im) af0[im] = *reinterpret_cast<const bf16x8*>(Sn + a0 + im * 2048);
#pragma unroll
    for (int in = 0; in < NI; ++in) bf0[in] = *reinterpret_cast<const bf16x8*>(Sn + b0 + in * 2048);
#pragma unroll
    for (int in = 0; in < NI; ++in)
#pragma unroll
      for (int im = 0; im < MI; ++im) acc[in][im] = EVK_MFMA_16x16x32(bf1[in], af1[im], acc[in][im], 0, 0, 0);
    __syncthreads();
    const int t = cur; cur = nxt; nxt = wr; wr = t;
  }

#undef EVK_ST_A
  // ---- epilogue: lane holds C[m][n0 .. n0+3], m = row0 + wm*144 + im*16 + frow, n0 = col0 + wn*32 + in*16 + fq*4 ----
  const int N = p.N;
  bool rowok[MI];
#pragma unroll
  for (int im = 0; im < MI; ++im) {
    rowok[im] = row0 + wm * (16 * MI) + im * 16 + frow < p.M;
    if (!rowok[im]) {
#pragma unroll
      for (int in = 0; in < NI; ++in) acc[in][im] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (p.colstats) {
    float* prow = p.colstats + ((long)(tm * WM + wm)) * 2 * N;
#pragma unroll
    for (int in = 0; in < NI; ++in) {
      float sm[4], sq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int im = 0; im < MI; ++im) { const float v = acc[in][im][j]; a += v; b += v * v; }
        sm[j] = row16_sum(a);
        sq[j] = row16_sum(b);
      }
      const int n0 = col0 + wn * (16 * NI) + in * 16 + fq * 4;
      if (frow == 0) {
        *reinterpret_cast<float4*>(prow + n0) = make_float4(sm[0], sm[1], sm[2], sm[3]);
        *reinterpret_cast<float4*>(prow + N + n0) = make_float4(sq[0], sq[1], sq[2], sq[3]);
      }
    }
  }
  if (p.bias) {
#pragma unroll
    for (int in = 0; in < NI; ++in) {
      const int n0 = col0 + wn * (16 * NI) + in * 16 + fq * 4;
      const float4 bb = *reinterpret_cast<const float4*>(p.bias + n0);
      const float4 sc = *reinterpret_cast<const float4*>(p.scale + n0);
#pragma unroll
      for (int im = 0; im < MI; ++im) {
        if (!rowok[im]) continue;
        const long m = (long)row0 + wm * (16 * MI) + im * 16 + frow;
        // bit for bit the unfused eval forward: this kernel's ROUNDED output -> bn_apply_kernel's fma(x, scale, shift) + identity, ReLU
        const uint32_t r01 = pack2bf(acc[in][im][0], acc[in][im][1]), r23 = pack2bf(acc[in][im][2], acc[in][im][3]);
        float v[4] = {__builtin_fmaf(lo_bf(r01), sc.x, bb.x), __builtin_fmaf(hi_bf(r01), sc.y, bb.y), __builtin_fmaf(lo_bf(r23), sc.z, bb.z), __builtin_fmaf(hi_bf(r23), sc.w, bb.w)};
        if (p.resid) {
          const uint2 t = *reinterpret_cast<const uint2*>(p.resid + m * p.ldr + n0);
          v[0] += lo_bf(t.x); v[1] += hi_bf(t.x); v[2] += lo_bf(t.y); v[3] += hi_bf(t.y);
        }
        if (p.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        *reinterpret_cast<uint2*>(p.C + m * p.ldc + n0) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      }
    }
    return;
  }
  if (!p.gate && !p.resid) {
    // forward convolution: nothing but the rounded product leaves -- one row pointer per MFMA tile row, no per-tile branches
    char* const cb = reinterpret_cast<char*>(p.C) + (((long)row0 + wm * (16 * MI) + frow) * p.ldc + col0 + wn * (16 * NI) + fq * 4) * 2;
    const long rstep = 16L * p.ldc * 2;
#pragma unroll
    for (int im = 0; im < MI; ++im) {
      if (!rowok[im]) continue;
#pragma unroll
      for (int in = 0; in < NI; ++in)
        *reinterpret_cast<uint2*>(cb + im * rstep + in * 32) =
            make_uint2(pack2bf(acc[in][im][0], acc[in][im][1]), pack2bf(acc[in][im][2], acc[in][im][3]));
    }
    return;
  }
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 gs[NI][2], gz[NI][2];
#pragma unroll
  for (int in = 0; in < NI; ++in)
#pragma unroll
    for (int h = 0; h < 2; ++h) { gs[in][h] = f32x2{0.f, 0.f}; gz[in][h] = f32x2{0.f, 0.f}; }
#pragma unroll
  for (int im = 0; im < MI; ++im) {
    if (!rowok[im]) continue;
    const long m = (long)row0 + wm * (16 * MI) + im * 16 + frow;
#pragma unroll
    for (int in = 0; in < NI; ++in) {
      const int n0 = col0 + wn * (16 * NI) + in * 16 + fq * 4;
      float v[4] = {acc[in][im][0], acc[in][im][1], acc[in][im][2], acc[in][im][3]};
      if (p.resid) {
        const uint2 t = *reinterpret_cast<const uint2*>(p.resid + m * p.ldr + n0);
        v[0] += lo_bf(t.x); v[1] += hi_bf(t.x); v[2] += lo_bf(t.y); v[3] += hi_bf(t.y);
      }
      if (p.gate) {
        const uint2 t = *reinterpret_cast<const uint2*>(p.gate + m * p.ldg + n0);
        const float gv[4] = {lo_bf(t.x), hi_bf(t.x), lo_bf(t.y), hi_bf(t.y)};
#pragma unroll
        for (int j = 0; j < 4; ++j) if (!(gv[j] > 0.f)) v[j] = 0.f;
        if (p.gatestats) {
          // explicit two-wide vectors in natural order: see gemm.hip (the SLP-chosen cross-half v_pk_add_f32 form is unsafe on gfx950)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 vv = {v[2 * h], v[2 * h + 1]}, gg = {gv[2 * h], gv[2 * h + 1]};
            gs[in][h] += vv;
            gz[in][h] += vv * gg;
          }
        }
      }
      *reinterpret_cast<uint2*>(p.C + m * p.ldc + n0) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
    }
  }
  if (p.gatestats) {
    float* prow = p.gatestats + ((long)(tm * WM + wm)) * 2 * N;
#pragma unroll
    for (int in = 0; in < NI; ++in) {
      float a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = row16_sum(gs[in][j >> 1][j & 1]); b[j] = row16_sum(gz[in][j >> 1][j & 1]); }
      const int n0 = col0 + wn * (16 * NI) + in * 16 + fq * 4;
      if (frow == 0) {
        *reinterpret_cast<float4*>(prow + n0) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(prow + N + n0) = make_float4(b[0], b[1], b[2], b[3]);
      }
    }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the device's zero block + the kernel's dynamic-LDS limit, once per device (both belong to ONE GPU) and safe against concurrent first calls
inline const void* strip_zero_block() {
  static EvkDeviceOnce once;
  return once.get([]() -> void* {
    void* z = nullptr;
    if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_zero16)) != hipSuccess) z = nullptr;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_strip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    return z;
  });
}

}  // namespace

extern "C" {

int evk_gemm_strip_supported(int64_t M, int32_t N, int32_t K) {
  return M > 0 && M < (1L << 31) && N >= TN && N % TN == 0 && K >= 64 && K % 64 == 0 ? 1 : 0;
}

int64_t evk_gemm_strip_part_bytes(int64_t M, int32_t N) { return cdiv(M, TM) * WM * 2 * N * (int64_t)sizeof(float); }

int evk_gemm_strip(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int32_t N, int32_t K,
                   const void* resid, int64_t ldr, const void* gate, int64_t ldg, float* colstats, float* gatestats, int64_t part_bytes,
                   int32_t* nblk, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(A && B && C, "gemm_strip: null operand");
  EVK_REQUIRE(evk_gemm_strip_supported(M, N, K), "gemm_strip: unsupported shape M=%ld N=%d K=%d (N %% 128, K %% 64)", (long)M, N, K);
  EVK_REQUIRE(al16(A) && al16(B) && al16(C) && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && lda >= K && ldb >= K && ldc >= N &&
              (!resid || (al16(resid) && ldr % 4 == 0 && ldr >= N)) && (!gate || (al16(gate) && ldg % 4 == 0 && ldg >= N)),
              "gemm_strip: alignment / leading dimensions");
  EVK_REQUIRE(!(colstats && gatestats) && (!gatestats || gate), "gemm_strip: one statistics epilogue at a time; gate statistics need a gate");
  StP p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = (bf16_t*)C;
  p.M = (int)M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.tilesN = N / TN;
  const int tilesM = (int)cdiv(M, TM);
  p.resid = (const bf16_t*)resid; p.ldr = ldr; p.gate = (const bf16_t*)gate; p.ldg = ldg;
  p.colstats = colstats; p.gatestats = gatestats;
  if (colstats || gatestats) {
    EVK_REQUIRE(nblk && part_bytes >= evk_gemm_strip_part_bytes(M, N), "gemm_strip: statistics buffer too small");
    *nblk = tilesM * WM;
  }
  p.zeros = strip_zero_block();
  EVK_REQUIRE(p.zeros, "gemm_strip: no address for the zero block");
  static const int probe = evk_tunable("EVK_STRIP_PROBE", 0);
  p.kmul = probe ? 0u : 128u;
  evk_prof_tag((int)M, N, K, 1, EVK_A_PLAIN, EVK_B_PLAIN);
  ProfScope ps(EVK_FAM_GEMM, s, 2.0 * M * (double)N * K);
  hipLaunchKernelGGL(gemm_strip_kernel, dim3(tilesM * p.tilesN), dim3(NTH), LDS_BYTES, s, p);
  return evk_check_launch("gemm_strip_kernel");
}

/* inference form (eval-mode batch norm as per-column scale / shift): C = relu?((A . B^T) * scale[n] + bias[n] (+ resid)) */
int evk_gemm_strip_affine(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int32_t N, int32_t K,
                          const float* scale, const float* bias, const void* resid, int64_t ldr, int32_t relu, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(A && B && C && scale && bias, "gemm_strip_affine: null operand");
  EVK_REQUIRE(evk_gemm_strip_supported(M, N, K), "gemm_strip_affine: unsupported shape M=%ld N=%d K=%d (N %% 128, K %% 64)", (long)M, N, K);
  EVK_REQUIRE(al16(A) && al16(B) && al16(C) && al16(scale) && al16(bias) && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && lda >= K && ldb >= K && ldc >= N &&
              (!resid || (al16(resid) && ldr % 4 == 0 && ldr >= N)), "gemm_strip_affine: alignment / leading dimensions");
  StP p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = (bf16_t*)C;
  p.M = (int)M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.tilesN = N / TN;
  const int tilesM = (int)cdiv(M, TM);
  p.resid = (const bf16_t*)resid; p.ldr = ldr;
  p.scale = scale; p.bias = bias; p.relu = relu;
  p.zeros = strip_zero_block();
  EVK_REQUIRE(p.zeros, "gemm_strip: no address for the zero block");
  p.kmul = 128u;
  evk_prof_tag((int)M, N, K, 1, EVK_A_PLAIN, EVK_B_PLAIN);
  ProfScope ps(EVK_FAM_GEMM, s, 2.0 * M * (double)N * K);
  hipLaunchKernelGGL(gemm_strip_kernel, dim3(tilesM * p.tilesN), dim3(NTH), LDS_BYTES, s, p);
  return evk_check_launch("gemm_strip_kernel");
}

// routing used by conv.hip for pointwise convolutions: the strip kernel pays when its one-block-per-CU grid fills the chip and the
// K loop is long enough to amortise its prologue (EVK_GEMM_STRIP=0 disables, EVK_GEMM_STRIP_MIN_BLOCKS moves the threshold)
int evk_gemm_strip_routes(int64_t M, int32_t N, int32_t K, int64_t part_bytes, int32_t want_stats) {
  static const int on = evk_tunable("EVK_GEMM_STRIP", 1);
  static const int min_blocks = evk_tunable("EVK_GEMM_STRIP_MIN_BLOCKS", 200);
  if (!on || !evk_gemm_strip_supported(M, N, K) || K < 256) return 0;
  if (cdiv(M, TM) * (N / TN) < min_blocks) return 0;
  if (want_stats && part_bytes < evk_gemm_strip_part_bytes(M, N)) return 0;
  return 1;
}

}  // extern "C"
