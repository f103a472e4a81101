// rm_f32.hip -- the relational memory of the DECODE step in f32 (RelationalMemory.forward_step, modules/encoder_decoder.py:274-291, as
// CaptionModel.beam_search drives it: one token per hypothesis per call).
//
// Why f32: the memory is a recurrence over every generated position, and on the weights the parity fixtures use it is an EXPANDING one -- a
// relative perturbation of 1e-6 of its weights moves the log-probabilities of position 99 by 3e-3, 16-bit operands by 0.3-0.5
// (tests/golden/make_beam_trace.py::drift16) -- so a 16-bit recurrence cannot follow the reference's beam search past the first few
// near-ties.  Everything that feeds back into the memory therefore stays in f32 here: the token embedding, the memory itself, the seven
// weight matrices (the f32 masters, no 16-bit shadows), every intermediate.  Only the copy of the new memory that the decoder's conditional
// layer norms read is rounded to the 16-bit storage format (that path does not feed back).
//
// The six products (768 x {512 .. 2048} x 512 for 256 hypotheses: 3.8 GFLOP per token) run on the f32-input MFMA (v_mfma_f32_16x16x4_f32, 1/16 of
// the 16-bit rate: 27 us per token at the chip's f32 matrix peak) in five launches: {token projections | q, k, v of the memory | U tanh(m)} as
// one grid, the slot attention, Wo (+ m), W0, and W2 with the gate in its epilogue (gemm_f32_kernel below).
#include "common.h"

namespace {

constexpr int S3 = 3, KEYS4 = 4, HEADS8 = 8, DH64 = 64, D512 = 512;
constexpr int KP = 512;

// gate epilogue of the LAST product of a token (h2 = relu(h1 W2^T + b2)): next = sigmoid(ig) * tanh(nm1 + h2) + sigmoid(fg) * m with
// gates = W x (xp columns 1024..2047 of the token, shared by the 3 slots) + U tanh(m) (gu); the memory is updated in place (every element is
// read and written by the one lane that owns it) and the 16-bit copies the decoder / the 16-bit BPTT read are written alongside
struct GateE {
  const float* xp; long x_bstride; const float* gu; const float* nm1; float* m;
  bf16_t* out16; long out_bstride;                 // memory row for the conditional layer norms: out16[b * out_bstride + slot * 512 + c]
  bf16_t* m16_next; bf16_t* tm16_next; bf16_t* si; bf16_t* sf; bf16_t* tnm;      // training saves ([R][512] each) or null
  float* tm32;                                     // tanh(new memory) in f32 ([R][512]) or null
};
struct GemmF { const float* A; const float* W; const float* bias; const float* resid; float* C; int M, N; long lda, ldc, ldr; int act, a_tanh; bf16_t* C16; long ldc16;
               int nbx; };                          // nbx = N / 32: column blocks of this problem
struct GemmF3 { GemmF g[3]; int nprob; int b1, b2; GateE gate; int has_gate; };     // blocks [0, b1) -> g[0], [b1, b2) -> g[1], [b2, ..) -> g[2]

// 32 x 64 output tile per workgroup (4 waves as 2 x 2, each 16 rows x 32 columns = two 16 x 16 accumulators), K = 512 in eight chunks of 64
// through a double-buffered LDS image (26 KB per stage: several workgroups share a CU and overlap each other's loads), the next chunk's
// global loads in flight while the current one is multiplied.  A lane reads 4 consecutive k of its row with one ds_read_b128 and feeds them to
// 4 successive v_mfma_f32_16x16x4_f32 (within a 16-deep group the k order is a permutation common to both operands).
// Round 3's kernel gave every workgroup a 32 x 32 tile and BOTH whole 32 x 512 panels (128 KB of LDS, one workgroup per CU, no overlap):
// ~10 us per round of 256 workgroups for 2 us of MFMA, 160 us of a 580 us decode step.
constexpr int FTM = 32, FTN = 64, FKC = 64, FLD = FKC + 4;          // LDS row pitch 68 floats: 16 lanes x 16 B land in 16 different bank groups
constexpr int FSTAGE = (FTM + FTN) * FLD;                            // floats per stage (6528)

// SPLIT (fp16-storage build): every f32 operand element x goes into LDS as TWO fp16 values, hi = fp16(x) and lo = fp16(x - hi) (x = hi + lo to
// 2^-22 |x|), and the product is accumulated in f32 from three 16-bit MFMAs per 32-deep step -- hi.hi + hi.lo + lo.hi; the lo.lo term is below
// f32's own rounding -- on v_mfma_f32_16x16x32_f16 (16 x the f32 MFMA's rate: 6 instructions of 16 cycles replace 16 of 32).  Same bytes from
// memory (the f32 masters), same LDS footprint (2 + 2 bytes per element), the split done once per element on its way into LDS.  The error of
// MEASURED: the products get 2.5 x cheaper but the launches are latency bound -- decode 119.9 -> 121.5 k tokens/s, train step 48.85 -> 48.65 ms
// (two A/B pairs on one box) -- while the 60-token recurrence of tests/test_hip_ops.py ends 4.9 x the f32 oracle's distance from float64
// (f32 MFMA: 0.5 x; operands to 2^-22 instead of 2^-24).  Model-level parity is unchanged (log-probabilities 2-3e-3 from the reference in
// both modes: 16-bit noise elsewhere dominates), but a 1 % gain does not buy a 9 x coarser recurrence: OPT-IN (EVK_RM_SPLIT16=1).
constexpr int HLD = FKC + 8;                                         // fp16 row pitch (144 bytes: 16-byte reads of 16 rows hit 16 bank groups)
constexpr int HIMG = (FTM + FTN) * HLD;                              // halfs per image (hi or lo) of one stage
__device__ __forceinline__ void split4(const float4 x, uint2& h, uint2& l) {
#ifdef EVK_STORE_F16
  const float v[4] = {x.x, x.y, x.z, x.w};
  unsigned short hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 a = (_Float16)v[i];
    const _Float16 b = (_Float16)(v[i] - (float)a);
    hh[i] = __builtin_bit_cast(unsigned short, a);
    ll[i] = __builtin_bit_cast(unsigned short, b);
  }
  h = make_uint2((uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16));
  l = make_uint2((uint32_t)ll[0] | ((uint32_t)ll[1] << 16), (uint32_t)ll[2] | ((uint32_t)ll[3] << 16));
#else
  h = l = make_uint2(0, 0);
#endif
}

// 1: split products (opt-in, fp16-storage build only), 0: f32 MFMA (default).  EVK_RM_SPLIT16 sets the initial value, evk_rm_f32_split16() changes it.
int g_split16 = -1;
inline int rm_split16() {
#ifdef EVK_STORE_F16
  if (g_split16 < 0) g_split16 = evk_tunable("EVK_RM_SPLIT16", 0) != 0;
  return g_split16;
#else
  return 0;                                   // (bf16 has 8 significand bits: three terms would not reach f32)
#endif
}

// one output element of a product: activation, residual, f32 / 16-bit stores and, on the last product of a token, the gate (see GateE)
__device__ __forceinline__ void f32_epilogue(const GemmF3& pp, const GemmF& p, int m, int col, float v) {
  if (m >= p.M) return;
  if (p.act == EVK_ACT_RELU) v = fmaxf(v, 0.f);
  if (p.resid) v += p.resid[(long)m * p.ldr + col];
  if (p.C) p.C[(long)m * p.ldc + col] = v;
  if (p.C16) p.C16[(long)m * p.ldc16 + col] = f2bf(v);          // what the 16-bit backward of the training recurrence reads
  if (pp.has_gate) {
    const GateE& g = pp.gate;                                  // v = h2[m][col]
    const long b = m / S3, e = (long)m * D512 + col;
    const int sl = m - (int)b * S3;
    const float ig = g.xp[b * g.x_bstride + 1024 + col] + g.gu[(long)m * 1024 + col];
    const float fg = g.xp[b * g.x_bstride + 1536 + col] + g.gu[(long)m * 1024 + 512 + col];
    const float si = 1.f / (1.f + expf(-ig)), sf = 1.f / (1.f + expf(-fg));
    const float t = tanhf(g.nm1[e] + v);
    float nx = si * t + sf * g.m[e];
    // the 16-bit copy is the f32 STATE rounded once more: without this fence the compiler forms it with a mixed-precision fma
    // (v_fma_mixlo_f16: one rounding of the exact sum), which differs from round16(round32(.)) at ties
    asm volatile("" : "+v"(nx));
    g.m[e] = nx;
    const bf16_t n16 = f2bf(nx);
    g.out16[b * g.out_bstride + sl * D512 + col] = n16;
    if (g.tm32 || g.m16_next) {
      const float tn = tanhf(nx);
      if (g.tm32) g.tm32[e] = tn;                              // tanh(memory) in f32: the A operand of the next token's U product
      if (g.m16_next) {
        g.m16_next[e] = n16; g.tm16_next[e] = f2bf(tn);
        g.si[e] = f2bf(si); g.sf[e] = f2bf(sf); g.tnm[e] = f2bf(t);
      }
    }
  }
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmF3 pp) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int which = bid < pp.b1 ? 0 : (bid < pp.b2 ? 1 : 2);
  bid -= which == 0 ? 0 : (which == 1 ? pp.b1 : pp.b2);
  const GemmF& p = pp.g[which];
  const int m0 = (bid / p.nbx) * FTM, n0 = (bid % p.nbx) * FTN;
  // staging: thread -> row pr (+16, ...), float4 pc of the 64-float chunk row
  const int pr = tid >> 4, pc = tid & 15;
  const bool v0 = m0 + pr < p.M, v1 = m0 + pr + 16 < p.M;
  const float* const a0p = p.A + (long)(m0 + pr) * p.lda + pc * 4;
  const float* const a1p = a0p + 16 * p.lda;
  const float* const bp = p.W + (long)(n0 + pr) * KP + pc * 4;
  // three register sets: the global loads of chunks c + 2 and c + 3 are in flight while chunk c is multiplied (the f32 masters stream from the
  // Infinity Cache: ~2 us of latency against ~0.5 us of MFMA per chunk; with one chunk of look-ahead every iteration waited for its loads)
  // (named registers and macros: an array captured by a lambda ends up in scratch memory, and a store to scratch waits for every load)
  float4 r00, r01, r02, r03, r04, r05, r10, r11, r12, r13, r14, r15, r20, r21, r22, r23, r24, r25;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define EVK_F32_GLOAD(A0, A1, B0, B1, B2, B3, c)                                     \
  {                                                                                  \
    const int k_ = (c) * FKC;                                                        \
    A0 = v0 ? *reinterpret_cast<const float4*>(a0p + k_) : zero4;                    \
    A1 = v1 ? *reinterpret_cast<const float4*>(a1p + k_) : zero4;                    \
    B0 = *reinterpret_cast<const float4*>(bp + k_);                                  \
    B1 = *reinterpret_cast<const float4*>(bp + 16 * KP + k_);                        \
    B2 = *reinterpret_cast<const float4*>(bp + 32 * KP + k_);                        \
    B3 = *reinterpret_cast<const float4*>(bp + 48 * KP + k_);                        \
  }
#define EVK_F32_SSTORE(A0, A1, B0, B1, B2, B3, stage)                                \
  {                                                                                  \
    float* const As_ = sm + (stage) * FSTAGE;                                        \
    float* const Bs_ = As_ + FTM * FLD;                                              \
    float4 x0_ = A0, x1_ = A1;                                                       \
    if (p.a_tanh) {                                                                  \
      x0_ = make_float4(tanhf(x0_.x), tanhf(x0_.y), tanhf(x0_.z), tanhf(x0_.w));     \
      x1_ = make_float4(tanhf(x1_.x), tanhf(x1_.y), tanhf(x1_.z), tanhf(x1_.w));     \
    }                                                                                \
    if (SPLIT) {                                                                     \
      unsigned short* const H_ = reinterpret_cast<unsigned short*>(sm) + (stage) * 2 * HIMG;   \
      unsigned short* const L_ = H_ + HIMG;                                          \
      const float4 xs_[6] = {x0_, x1_, B0, B1, B2, B3};                              \
      _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                             \
        const int row_ = (i_ < 2 ? pr + 16 * i_ : FTM + pr + 16 * (i_ - 2));         \
        uint2 h_, l_;                                                                \
        split4(xs_[i_], h_, l_);                                                     \
        *reinterpret_cast<uint2*>(H_ + row_ * HLD + pc * 4) = h_;                    \
        *reinterpret_cast<uint2*>(L_ + row_ * HLD + pc * 4) = l_;                    \
      }                                                                              \
    } else {                                                                         \
      *reinterpret_cast<float4*>(As_ + pr * FLD + pc * 4) = x0_;                     \
      *reinterpret_cast<float4*>(As_ + (pr + 16) * FLD + pc * 4) = x1_;              \
      *reinterpret_cast<float4*>(Bs_ + pr * FLD + pc * 4) = B0;                      \
      *reinterpret_cast<float4*>(Bs_ + (pr + 16) * FLD + pc * 4) = B1;               \
      *reinterpret_cast<float4*>(Bs_ + (pr + 32) * FLD + pc * 4) = B2;               \
      *reinterpret_cast<float4*>(Bs_ + (pr + 48) * FLD + pc * 4) = B3;               \
    }                                                                                \
  }
#define EVK_F32_SET0 r00, r01, r02, r03, r04, r05
#define EVK_F32_SET1 r10, r11, r12, r13, r14, r15
#define EVK_F32_SET2 r20, r21, r22, r23, r24, r25
#define EVK_F32_GL(SET, c) EVK_F32_GLOAD_X(SET, c)
#define EVK_F32_GLOAD_X(A0, A1, B0, B1, B2, B3, c) EVK_F32_GLOAD(A0, A1, B0, B1, B2, B3, c)
#define EVK_F32_SS(SET, st) EVK_F32_SSTORE_X(SET, st)
#define EVK_F32_SSTORE_X(A0, A1, B0, B1, B2, B3, st) EVK_F32_SSTORE(A0, A1, B0, B1, B2, B3, st)
  const int wm = wave >> 1, wn = wave & 1;
  const int aoff = (wm * 16 + (lane & 15)) * FLD + 4 * (lane >> 4);
  const int boff = FTM * FLD + (wn * 32 + (lane & 15)) * FLD + 4 * (lane >> 4);
  const int haoff = (wm * 16 + (lane & 15)) * HLD + 8 * (lane >> 4);                       // (SPLIT) 8 consecutive k of the lane's row
  const int hboff = (FTM + wn * 32 + (lane & 15)) * HLD + 8 * (lane >> 4);
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  constexpr int NCH = KP / FKC;
  static_assert(NCH == 8, "the chunk loop below is written out for eight chunks");
  EVK_F32_GL(EVK_F32_SET0, 0)
  EVK_F32_GL(EVK_F32_SET1, 1)
  EVK_F32_GL(EVK_F32_SET2, 2)
  EVK_F32_SS(EVK_F32_SET0, 0)
  __syncthreads();
#define EVK_F32_MAC(c)                                                                              \
  if (SPLIT) {                                                                                      \
    const unsigned short* const H_ = reinterpret_cast<const unsigned short*>(sm) + ((c) & 1) * 2 * HIMG;   \
    const unsigned short* const L_ = H_ + HIMG;                                                     \
    _Pragma("unroll") for (int q = 0; q < FKC / 32; ++q) {                                          \
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(H_ + haoff + 32 * q), al = *reinterpret_cast<const bf16x8*>(L_ + haoff + 32 * q);        \
      const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(H_ + hboff + 32 * q), bl0 = *reinterpret_cast<const bf16x8*>(L_ + hboff + 32 * q);      \
      const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(H_ + hboff + 16 * HLD + 32 * q), bl1 = *reinterpret_cast<const bf16x8*>(L_ + hboff + 16 * HLD + 32 * q);  \
      acc[0] = EVK_MFMA_16x16x32(ah, bh0, acc[0], 0, 0, 0);                                         \
      acc[1] = EVK_MFMA_16x16x32(ah, bh1, acc[1], 0, 0, 0);                                         \
      acc[0] = EVK_MFMA_16x16x32(ah, bl0, acc[0], 0, 0, 0);                                         \
      acc[1] = EVK_MFMA_16x16x32(ah, bl1, acc[1], 0, 0, 0);                                         \
      acc[0] = EVK_MFMA_16x16x32(al, bh0, acc[0], 0, 0, 0);                                         \
      acc[1] = EVK_MFMA_16x16x32(al, bh1, acc[1], 0, 0, 0);                                         \
    }                                                                                               \
    __syncthreads();                                                                                \
  } else {                                                                                          \
    const float* const S = sm + ((c) & 1) * FSTAGE;                                                 \
    _Pragma("unroll") for (int q = 0; q < FKC / 16; ++q) {                                          \
      const float4 a4 = *reinterpret_cast<const float4*>(S + aoff + 16 * q);                        \
      const float4 b40 = *reinterpret_cast<const float4*>(S + boff + 16 * q);                       \
      const float4 b41 = *reinterpret_cast<const float4*>(S + boff + 16 * FLD + 16 * q);            \
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b40.x, acc[0], 0, 0, 0);                  \
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b41.x, acc[1], 0, 0, 0);                  \
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b40.y, acc[0], 0, 0, 0);                  \
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b41.y, acc[1], 0, 0, 0);                  \
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b40.z, acc[0], 0, 0, 0);                  \
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b41.z, acc[1], 0, 0, 0);                  \
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b40.w, acc[0], 0, 0, 0);                  \
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b41.w, acc[1], 0, 0, 0);                  \
    }                                                                                               \
    __syncthreads();                                                                                \
  }
  // iteration c: chunk c + 1 goes to the other LDS stage (its readers passed the previous barrier), chunk c + 3 is requested into the
  // register set chunk c left one iteration ago, chunk c is multiplied
  EVK_F32_SS(EVK_F32_SET1, 1) EVK_F32_GL(EVK_F32_SET0, 3) EVK_F32_MAC(0)
  EVK_F32_SS(EVK_F32_SET2, 0) EVK_F32_GL(EVK_F32_SET1, 4) EVK_F32_MAC(1)
  EVK_F32_SS(EVK_F32_SET0, 1) EVK_F32_GL(EVK_F32_SET2, 5) EVK_F32_MAC(2)
  EVK_F32_SS(EVK_F32_SET1, 0) EVK_F32_GL(EVK_F32_SET0, 6) EVK_F32_MAC(3)
  EVK_F32_SS(EVK_F32_SET2, 1) EVK_F32_GL(EVK_F32_SET1, 7) EVK_F32_MAC(4)
  EVK_F32_SS(EVK_F32_SET0, 0) EVK_F32_MAC(5)
  EVK_F32_SS(EVK_F32_SET1, 1) EVK_F32_MAC(6)
  EVK_F32_MAC(7)
#undef EVK_F32_MAC
#undef EVK_F32_GL
#undef EVK_F32_SS
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int col = n0 + wn * 32 + jt * 16 + (lane & 15);
    const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) f32_epilogue(pp, p, m0 + wm * 16 + 4 * (lane >> 4) + j, col, acc[jt][j] + bias);
  }
}

inline GemmF mkf(const float* A, long lda, const float* W, const float* bias, const float* resid, long ldr, float* C, long ldc, int M, int N, int act,
                 int a_tanh, bf16_t* C16 = nullptr, long ldc16 = 0) {
  return GemmF{A, W, bias, resid, C, M, N, lda, ldc, ldr, act, a_tanh, C16, ldc16, N / FTN};
}

// up to three independent products in ONE launch (their workgroups are simply concatenated), optionally with the gate epilogue on the first
int gemm_f32_multi(const GemmF* g, int nprob, const GateE* gate, hipStream_t s) {
  EVK_REQUIRE(nprob >= 1 && nprob <= 3, "rm f32 gemm: 1 .. 3 problems per launch");
  constexpr int LDS = (2 * FSTAGE * 4 > 2 * 2 * HIMG * 2) ? 2 * FSTAGE * 4 : 2 * 2 * HIMG * 2;
  const bool split = rm_split16() != 0;
  EVK_DYN_LDS_ONCE(gemm_f32_kernel<false>, LDS);
  EVK_DYN_LDS_ONCE(gemm_f32_kernel<true>, LDS);
  GemmF3 pp{};
  int blocks[3] = {0, 0, 0};
  for (int i = 0; i < nprob; ++i) {
    EVK_REQUIRE(g[i].N % FTN == 0 && g[i].M > 0, "rm f32 gemm: N must be a multiple of 64");
    pp.g[i] = g[i];
    blocks[i] = (g[i].N / FTN) * ((g[i].M + FTM - 1) / FTM);
  }
  pp.nprob = nprob;
  pp.b1 = blocks[0]; pp.b2 = blocks[0] + blocks[1];
  pp.has_gate = gate ? 1 : 0;
  if (gate) { EVK_REQUIRE(nprob == 1 && g[0].N == D512, "rm f32 gemm: the gate epilogue rides on the single 512-column product"); pp.gate = *gate; }
  if (split) hipLaunchKernelGGL(gemm_f32_kernel<true>, dim3(blocks[0] + blocks[1] + blocks[2]), dim3(256), LDS, s, pp);
  else hipLaunchKernelGGL(gemm_f32_kernel<false>, dim3(blocks[0] + blocks[1] + blocks[2]), dim3(256), LDS, s, pp);
  return evk_check_launch("rm f32 gemm");
}

int gemm_f32(const float* A, long lda, const float* W, const float* bias, const float* resid, long ldr, float* C, long ldc, int M, int N, int act,
             int a_tanh, hipStream_t s, bf16_t* C16 = nullptr, long ldc16 = 0) {
  const GemmF g = mkf(A, lda, W, bias, resid, ldr, C, ldc, M, N, act, a_tanh, C16, ldc16);
  return gemm_f32_multi(&g, 1, nullptr, s);
}

__device__ __forceinline__ float half_sum32(float v) {      // sum over the 32 lanes that share a head
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one workgroup per hypothesis; thread -> head h = tid >> 5, dims 2l, 2l + 1: softmax(q k^T / 8) v over the 3 memory slots + the token
__global__ __launch_bounds__(256) void rm_attn_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ xp, float* __restrict__ a) {
  const int b = blockIdx.x, h = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int c = h * DH64 + 2 * l;
  float q[S3][2], k[KEYS4][2], v[KEYS4][2];
#pragma unroll
  for (int i = 0; i < S3; ++i) {
    const float* r = qkv + (long)(b * S3 + i) * 1536 + c;
    q[i][0] = r[0]; q[i][1] = r[1]; k[i][0] = r[512]; k[i][1] = r[513]; v[i][0] = r[1024]; v[i][1] = r[1025];
  }
  const float* x = xp + (long)b * 2048 + c;
  k[3][0] = x[0]; k[3][1] = x[1]; v[3][0] = x[512]; v[3][1] = x[513];
#pragma unroll
  for (int i = 0; i < S3; ++i) {
    float s[KEYS4], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) { s[j] = half_sum32(q[i][0] * k[j][0] + q[i][1] * k[j][1]) * 0.125f; mx = fmaxf(mx, s[j]); }
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
    float o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) { const float w = s[j] / sum; o0 += w * v[j][0]; o1 += w * v[j][1]; }
    float* o = a + (long)(b * S3 + i) * D512 + c;
    o[0] = o0; o[1] = o1;
  }
}

// next = sigmoid(ig) * tanh(nm1 + h2) + sigmoid(fg) * m, gates = W x (xp columns 1024..2047, shared by the slots) + U tanh(m); in place
__global__ __launch_bounds__(256) void rm_gate_f32_kernel(const float* __restrict__ xp, const float* __restrict__ gu, const float* __restrict__ nm1,
                                                          const float* __restrict__ h2, float* __restrict__ m, bf16_t* __restrict__ out16, int B) {
  const long total = (long)B * S3 * D512;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % D512);
    const long bs = i / D512;
    const long b = bs / S3;
    const float ig = xp[b * 2048 + 1024 + c] + gu[bs * 1024 + c];
    const float fg = xp[b * 2048 + 1536 + c] + gu[bs * 1024 + 512 + c];
    const float si = 1.f / (1.f + expf(-ig)), sf = 1.f / (1.f + expf(-fg));
    const float nx = si * tanhf(nm1[i] + h2[i]) + sf * m[i];
    m[i] = nx;
    out16[i] = f2bf(nx);
  }
}

// ---- the TRAINING recurrence in f32 (RelationalMemory.forward, modules/encoder_decoder.py:293-300): same arithmetic as above plus what
// the 16-bit BPTT of rm.hip reads back -- attention probabilities, dropout (the stateless hash of rm.hip: forward and backward draw the
// same mask), 16-bit copies of every saved intermediate.
__device__ __forceinline__ uint32_t hash32f(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}
__device__ __forceinline__ float keep_scale_f(uint64_t seed, uint64_t idx, float p) {          // == rm.hip keep_scale
  if (p <= 0.f) return 1.f;
  return (hash32f(seed * 0x9E3779B97F4A7C15ULL + idx) >> 8) * (1.f / 16777216.f) >= p ? 1.f / (1.f - p) : 0.f;
}

struct AttT { const float* qkv; const float* xp; long x_bstride; float* a; bf16_t* a16; float* P; float p_drop; unsigned long long seed; const unsigned long long* epoch; };
__global__ __launch_bounds__(256) void rm_attn_f32_train_kernel(const AttT p) {
  const int b = blockIdx.x, h = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int c = h * DH64 + 2 * l;
  float q[S3][2], k[KEYS4][2], v[KEYS4][2];
#pragma unroll
  for (int i = 0; i < S3; ++i) {
    const float* r = p.qkv + (long)(b * S3 + i) * 1536 + c;
    q[i][0] = r[0]; q[i][1] = r[1]; k[i][0] = r[512]; k[i][1] = r[513]; v[i][0] = r[1024]; v[i][1] = r[1025];
  }
  const float* x = p.xp + (long)b * p.x_bstride + c;
  k[3][0] = x[0]; k[3][1] = x[1]; v[3][0] = x[512]; v[3][1] = x[513];
  const unsigned long long seed = evk_mix_seed(p.seed, p.epoch);
#pragma unroll
  for (int i = 0; i < S3; ++i) {
    float s[KEYS4], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) { s[j] = half_sum32(q[i][0] * k[j][0] + q[i][1] * k[j][1]) * 0.125f; mx = fmaxf(mx, s[j]); }
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
    float o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) {
      const float pr = s[j] / sum;
      if (l == 0 && p.P) p.P[((long)(b * HEADS8 + h) * S3 + i) * KEYS4 + j] = pr;
      const float w = pr * keep_scale_f(seed, ((uint64_t)(b * HEADS8 + h) * S3 + i) * KEYS4 + j, p.p_drop);
      o0 += w * v[j][0]; o1 += w * v[j][1];
    }
    const long o = (long)(b * S3 + i) * D512 + c;
    p.a[o] = o0; p.a[o + 1] = o1;
    *reinterpret_cast<uint32_t*>(p.a16 + o) = pack2bf(o0, o1);
  }
}

struct GateT {
  const float* xp; long x_bstride; const float* gu; const float* nm1; const float* h2; float* m;
  bf16_t* m16_next; bf16_t* tm16_next; bf16_t* out; long out_bstride; bf16_t* si; bf16_t* sf; bf16_t* tnm; int B;
};
__global__ __launch_bounds__(256) void rm_gate_f32_train_kernel(const GateT p) {
  const long total = (long)p.B * S3 * D512;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % D512);
    const long bs = i / D512;
    const long b = bs / S3;
    const int sl = (int)(bs - b * S3);
    const float ig = p.xp[b * p.x_bstride + 1024 + c] + p.gu[bs * 1024 + c];
    const float fg = p.xp[b * p.x_bstride + 1536 + c] + p.gu[bs * 1024 + 512 + c];
    const float si = 1.f / (1.f + expf(-ig)), sf = 1.f / (1.f + expf(-fg));
    const float t = tanhf(p.nm1[i] + p.h2[i]);
    const float nx = si * t + sf * p.m[i];
    p.m[i] = nx;                                        // the carried state stays in f32
    const bf16_t n16 = f2bf(nx);
    p.m16_next[i] = n16;
    p.tm16_next[i] = f2bf(tanhf(nx));
    p.out[b * p.out_bstride + sl * D512 + c] = n16;
    p.si[i] = f2bf(si); p.sf[i] = f2bf(sf); p.tnm[i] = f2bf(t);
  }
}

}  // namespace

// launchers used by rm.hip's evk_rm_forward_f32 (which owns the workspace layout the 16-bit backward reads)
int rmf32_gemm(const float* A, long lda, const float* W, const float* bias, const float* resid, long ldr, float* C, long ldc, int M, int N, int act,
               int a_tanh, bf16_t* C16, long ldc16, hipStream_t s) {
  return gemm_f32(A, lda, W, bias, resid, ldr, C, ldc, M, N, act, a_tanh, s, C16, ldc16);
}
int rmf32_attn_train(const float* qkv, const float* xp, long x_bstride, float* a, bf16_t* a16, float* P, float p_drop, unsigned long long seed, int B,
                     hipStream_t s) {
  AttT p{qkv, xp, x_bstride, a, a16, P, p_drop, seed, evk_seed_epoch_ptr()};
  hipLaunchKernelGGL(rm_attn_f32_train_kernel, dim3(B), dim3(256), 0, s, p);
  return evk_check_launch("rm f32 attention (train)");
}
// q | k | v of the memory and U tanh(m) as one launch (training recurrence: the token's x projections are hoisted out of the loop)
int rmf32_qkv_gu(const float* mem, const float* tmem, const float* Wqkv, const float* bqkv, float* qkv, bf16_t* qkv16, const float* U, const float* bU,
                 float* gu, int R, hipStream_t s) {
  // tmem = tanh(mem) in f32 when the previous token's gate epilogue left it (GateE::tm32); null: the product applies tanh to its A operand itself
  const GemmF g2[2] = {mkf(mem, D512, Wqkv, bqkv, nullptr, 0, qkv, 1536, R, 1536, EVK_ACT_NONE, 0, qkv16, 1536),
                       mkf(tmem ? tmem : mem, D512, U, bU, nullptr, 0, gu, 1024, R, 1024, EVK_ACT_NONE, tmem ? 0 : 1)};
  return gemm_f32_multi(g2, 2, nullptr, s);
}
// h2 = relu(h1 W2^T + b2) (16-bit copy saved) with the gate epilogue: memory in place + every 16-bit save of the training recurrence
int rmf32_w2_gate_train(const float* h1, const float* W2, const float* b2, bf16_t* h2_16, const float* xp, long x_bstride, const float* gu, const float* nm1,
                        float* m, bf16_t* m16_next, bf16_t* tm16_next, bf16_t* out, long out_bstride, bf16_t* si, bf16_t* sf, bf16_t* tnm, float* tm32,
                        int R, hipStream_t s) {
  const GemmF g1 = mkf(h1, D512, W2, b2, nullptr, 0, nullptr, D512, R, D512, EVK_ACT_RELU, 0, h2_16, D512);
  const GateE ge{xp, x_bstride, gu, nm1, m, out, out_bstride, m16_next, tm16_next, si, sf, tnm, tm32};
  return gemm_f32_multi(&g1, 1, &ge, s);
}
int rmf32_gate_train(const float* xp, long x_bstride, const float* gu, const float* nm1, const float* h2, float* m, bf16_t* m16_next, bf16_t* tm16_next,
                     bf16_t* out, long out_bstride, bf16_t* si, bf16_t* sf, bf16_t* tnm, int B, hipStream_t s) {
  GateT p{xp, x_bstride, gu, nm1, h2, m, m16_next, tm16_next, out, out_bstride, si, sf, tnm, B};
  const long total = (long)B * S3 * D512;
  hipLaunchKernelGGL(rm_gate_f32_train_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
  return evk_check_launch("rm f32 gate (train)");
}

extern "C" {

/* how the f32 relational-memory products are formed: 1 = three fp16 MFMAs over hi / lo halves of the f32 operands (fp16-storage build only),
 * 0 = v_mfma_f32_16x16x4_f32; mode < 0 only queries.  Returns the mode in force. */
int evk_rm_f32_split16(int32_t mode) {
  (void)rm_split16();
#ifdef EVK_STORE_F16
  if (mode >= 0) g_split16 = mode != 0;
#endif
  return rm_split16();
}


int64_t evk_rm_decode_f32_ws_bytes(int32_t B) {
  const long R = (long)B * S3;
  return ((long)B * 2048 + R * 1536 + 4 * R * D512 + R * 1024) * 4 + 4096;
}

int evk_rm_decode_step_f32(const float* x, const float* Wx, const float* bx, float* mem, const float* Wqkv, const float* bqkv, const float* Wo,
                           const float* bo, const float* W0, const float* b0, const float* W2, const float* b2, const float* U, const float* bU,
                           void* out16, void* ws, int64_t ws_bytes, int32_t B, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && Wx && bx && mem && Wqkv && bqkv && Wo && bo && W0 && b0 && W2 && b2 && U && bU && out16 && ws && B > 0, "rm_decode_step_f32: null/empty argument");
  EVK_REQUIRE(ws_bytes >= evk_rm_decode_f32_ws_bytes(B), "rm_decode_step_f32: workspace too small");
  EVK_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(mem) | reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(Wx) |
                reinterpret_cast<uintptr_t>(Wqkv) | reinterpret_cast<uintptr_t>(Wo) | reinterpret_cast<uintptr_t>(W0) | reinterpret_cast<uintptr_t>(W2) |
                reinterpret_cast<uintptr_t>(U)) & 15) == 0, "rm_decode_step_f32: 16-byte aligned buffers");
  const int R = B * S3;
  float* p = reinterpret_cast<float*>(ws);
  auto take = [&](long n) { float* r = p; p += (n + 63) / 64 * 64; return r; };
  float* xp = take((long)B * 2048);
  float* qkv = take((long)R * 1536);
  float* a = take((long)R * D512); float* nm1 = take((long)R * D512); float* h1 = take((long)R * D512); float* h2 = take((long)R * D512);
  float* gu = take((long)R * 1024);
  ProfScope ps(EVK_FAM_GEMM, s);
  // 5 launches: {x projections | q, k, v of the memory | U tanh(m)} as ONE launch of three independent products, slot attention, Wo (+ m), W0,
  // W2 with the gate in its epilogue (memory updated in place)
  {
    const GemmF g3[3] = {mkf(x, D512, Wx, bx, nullptr, 0, xp, 2048, B, 2048, EVK_ACT_NONE, 0),             // keys | values | gates of the token
                         mkf(mem, D512, Wqkv, bqkv, nullptr, 0, qkv, 1536, R, 1536, EVK_ACT_NONE, 0),
                         mkf(mem, D512, U, bU, nullptr, 0, gu, 1024, R, 1024, EVK_ACT_NONE, 1)};              // U tanh(m)
    if (int e = gemm_f32_multi(g3, 3, nullptr, s)) return e;
  }
  hipLaunchKernelGGL(rm_attn_f32_kernel, dim3(B), dim3(256), 0, s, qkv, xp, a);
  if (int e = gemm_f32(a, D512, Wo, bo, mem, D512, nm1, D512, R, D512, EVK_ACT_NONE, 0, s)) return e;
  if (int e = gemm_f32(nm1, D512, W0, b0, nullptr, 0, h1, D512, R, D512, EVK_ACT_RELU, 0, s)) return e;
  {
    const GemmF g1 = mkf(h1, D512, W2, b2, nullptr, 0, nullptr, D512, R, D512, EVK_ACT_RELU, 0);
    const GateE ge{xp, 2048L, gu, nm1, mem, (bf16_t*)out16, (long)S3 * D512, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (int e = gemm_f32_multi(&g1, 1, &ge, s)) return e;
  }
  (void)h2;
  return evk_check_launch("rm_decode_step_f32");
}

}  // extern "C"
