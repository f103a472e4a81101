// gemm.hip -- the MFMA GEMM / implicit-GEMM kernel family of libevoke_hip.so (gfx950).
//
// One templated kernel computes C[m][n] = act(alpha * sum_k A(m,k) * B(n,k) + bias[n]) + resid[m][n]
// for bf16 operands with f32 accumulation on v_mfma_f32_16x16x32_bf16.  The operand "loaders" differ
// by addressing mode so the same main loop serves nn.Linear forward / dX / dW, batched attention
// products, and the NHWC convolutions of the ResNet-101 trunk (forward = implicit im2col gather,
// data-gradient = gather from dY, weight-gradient = K-strided gather with split-K + f32 atomics).
//
// Tile: (64*WM) x (64*WN) x 64, 4 waves (256 threads), each wave a 64x64 sub-tile = 4x4 MFMA tiles.
// LDS image per operand: [row][64 k] bf16 = 128-byte rows, 16-byte chunk index XOR (row & 7): the
// ds_read_b128 fragment reads are bank-conflict free (cdna_hip_programming.md T2).  Register-staged
// double buffering: global loads of K-tile t+1 are issued before the MFMAs of tile t and written to the
// other LDS buffer after them (one barrier per K-tile).
// MFMA orientation is swapped (weights tile as the "A" operand) so each lane ends up with 4 consecutive
// n of one row m -> 8/16-byte epilogue stores.
#include <type_traits>
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int BK = 64;
constexpr int NTHR = 256;

struct GemmP {
  const bf16_t* A; const bf16_t* B; void* C; const float* bias; const void* resid;
  int M, N, K;
  long lda, ldb, ldc, ldr;
  int bi;
  long sAo, sAi, sBo, sBi, sCo, sCi, sRo, sRi, sbias;
  float alpha; int act, c_f32, r_f32, accumulate, vec_ok;
  int ksteps_per_split, tilesN, kslice_xcd;
  int group_m;        // tile order inside an XCD's run: super-rows of group_m M-tiles, N fastest across, M fastest inside (0/1 = plain M-major)
  int fast_loads;                   // interior tiles take the select-free loader path (EVK_FAST_LOADS=0 disables)
  int lds_store;                    // bf16 output rows leave through LDS in 16-byte pieces (needs N % 8 == 0, ldc % 8 == 0, aligned C)
  const bf16_t* gate; long ldg;     // optional ReLU gate: C = (gate > 0) ? C : 0, applied last (bf16 [M][ldg], batch 1)
  float* colstats;                  // per 64-row block partial column sums / sums of squares [row block][2][N] (or null)
  float* gatestats;                 // per 64-row block partial column sums of the GATED output g and of g * gate [row block][2][N] (or null)
  float* slab; long slab_mn;        // split-K partial slabs [z][split][M][N] f32 (accumulate mode with splitk > 1)
  int b_klog, b_kmask; long b_tapstride;
  // gather geometry
  int Hi, Wi, Ho, Wo, KH, KW, sh, sw, ph, pw, lgs;
  // optional LayerNorm of the A rows on their way into LDS (gemm_skinny_kernel<512,..>, K == 512): ln_g != null enables
  const float* ln_g; const float* ln_b; const bf16_t* ln_dg; const bf16_t* ln_db; long ln_ld; float ln_eps; int ln_mode;
  int lgC, Cg;       // channels of the gathered tensor (power of two)
  int rows_per_img, row_w;  // decomposition of the GEMM row (A_CONV: Ho*Wo, Wo; A_DGRAD: Hi*Wi, Wi)
  long sN, sH, sW;
};

// ------------------------------------------------------------------------------------------------
// K-contiguous loaders (PLAIN / CONV / DGRAD): thread -> row (tid>>3)+32i, 16-byte chunk kc = tid&7
// ------------------------------------------------------------------------------------------------
template <int ROWS, int MODE, int NT = NTHR>
struct RowLoader {
  static constexpr int RPP = NT / 8;            // rows covered by one pass of the block
  static constexpr int NI = ROWS / RPP;
  static constexpr int NREG = NI * 4;
  static constexpr bool HAS_FULL = MODE == EVK_A_PLAIN;
  const bf16_t* base;
  long off[NI];
  int y0[NI], x0[NI];
  bool ok[NI];
  // interior-tile fast path (PLAIN): one uniform tile pointer + a 32-bit per-lane byte offset per row, so the K advance is
  // scalar arithmetic and neither the loads nor the LDS stores carry validity selects
  const char* tbase;
  unsigned voff[NI];
  bool full;

  __device__ __forceinline__ void init(const GemmP& p, const bf16_t* b, long ld, int row0, int nrows, int tid) {
    base = b;
    if constexpr (HAS_FULL) {
      tbase = reinterpret_cast<const char*>(b + (long)row0 * ld);
      full = row0 + ROWS <= nrows;
#pragma unroll
      for (int i = 0; i < NI; ++i) voff[i] = (unsigned)(((tid >> 3) + RPP * i) * ld * 2 + (tid & 7) * 16);
    } else {
      tbase = nullptr; full = false;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = row0 + (tid >> 3) + RPP * i;
      ok[i] = r < nrows;
      const int rr = ok[i] ? r : 0;
      if constexpr (MODE == EVK_A_PLAIN) {
        off[i] = (long)rr * ld;
        y0[i] = x0[i] = 0;
      } else {
        const int n = rr / p.rows_per_img;
        const int rem = rr - n * p.rows_per_img;
        const int yy = rem / p.row_w;
        const int xx = rem - yy * p.row_w;
        if constexpr (MODE == EVK_A_CONV) {
          off[i] = (long)n * p.sN;
          y0[i] = yy * p.sh - p.ph;
          x0[i] = xx * p.sw - p.pw;
        } else {  // DGRAD: gather from dY [N][Ho][Wo][Cg]
          off[i] = (long)n * p.Ho * p.Wo * p.Cg;
          y0[i] = yy + p.ph;
          x0[i] = xx + p.pw;
        }
      }
    }
  }

  // Loads are UNCONDITIONAL (invalid chunks read the operand's first bytes) and the validity mask is applied when the
  // registers are written to LDS: a per-chunk `valid ? load : 0` makes hipcc branch around every load and drain vmcnt
  // in the middle of the sequence (cdna_hip_programming.md section 5, trap (c)).
  __device__ __forceinline__ unsigned load(const GemmP& p, int k0, int kend, int tid, uint4 (&v)[NI]) const {
    const int k = k0 + (tid & 7) * 8;
    const bool kin = k < kend;
    unsigned mask = 0;
    if constexpr (MODE == EVK_A_PLAIN) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const bool valid = ok[i] && kin;
        v[i] = *reinterpret_cast<const uint4*>(base + (valid ? off[i] + k : 0));
        mask |= (valid ? 1u : 0u) << i;
      }
    } else {
      const int tap = k >> p.lgC;
      const int c = k & (p.Cg - 1);
      const int kh = tap / p.KW;
      const int kw = tap - kh * p.KW;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        bool valid = ok[i] && kin;
        long a;
        if constexpr (MODE == EVK_A_CONV) {
          const int ih = y0[i] + kh, iw = x0[i] + kw;
          valid = valid && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
          a = off[i] + (long)ih * p.sH + (long)iw * p.sW + c;
        } else {
          const int th = y0[i] - kh, tw = x0[i] - kw;
          const int m = (1 << p.lgs) - 1;
          valid = valid && th >= 0 && tw >= 0 && !(th & m) && !(tw & m);
          const int oh = th >> p.lgs, ow = tw >> p.lgs;
          valid = valid && oh < p.Ho && ow < p.Wo;
          a = off[i] + ((long)oh * p.Wo + ow) * p.Cg + c;
        }
        v[i] = *reinterpret_cast<const uint4*>(base + (valid ? a : 0));
        mask |= (valid ? 1u : 0u) << i;
      }
    }
    return mask;
  }

  __device__ __forceinline__ void store(char* lds, int tid, const uint4 (&v)[NI], unsigned mask) const {
    const int rl = tid >> 3;
    char* d = lds + rl * 128 + (((tid & 7) ^ (rl & 7)) << 4);
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<uint4*>(d + i * (RPP * 128)) = ((mask >> i) & 1u) ? v[i] : make_uint4(0, 0, 0, 0);
  }

  __device__ __forceinline__ void load_full(const GemmP&, int k0, uint4 (&v)[NI]) const {
    const char* sb = tbase + (long)k0 * 2;
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const uint4*>(sb + voff[i]);
  }
  __device__ __forceinline__ void store_full(char* lds, int tid, const uint4 (&v)[NI]) const {
    const int rl = tid >> 3;
    char* d = lds + rl * 128 + (((tid & 7) ^ (rl & 7)) << 4);
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<uint4*>(d + i * (RPP * 128)) = v[i];
  }
};

// ------------------------------------------------------------------------------------------------
// K-strided loaders (KSTR / WGATHER): the operand is contiguous along its row (m or n) index, strided along k
// ("transposed").  The tile is staged AS IT LIES IN MEMORY -- LDS image [64 k][ROWS] with 16-byte global loads
// along the contiguous dim -- and the MFMA fragments (8 consecutive k per lane) are produced by the gfx950
// transposing LDS read ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per 16-lane group a 4(k) x 16(row) block
// comes back column-major, two reads give the 8 k of a 16x16x32 fragment.  Bank conflicts: the 8 k-rows touched by
// one 32-lane half are spread over the eight 32-byte slots of the 256-byte bank line by XOR-ing the 32-byte chunk
// index with s(k) = (k & 3) | ((k >> 3) & 1) << 2.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ s16x4 lds_tr_read(const char* generic_lds_ptr) {
  lds_s16x4* p = (lds_s16x4*)(__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)generic_lds_ptr;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
}
__device__ __forceinline__ int kswz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

template <int ROWS, int MODE, int NT = NTHR>  // MODE: 0 = plain/2-level K-strided, 1 = conv weight-gradient gather
struct KstrLoader {
  static constexpr int CPR = ROWS / 8;          // 16-byte chunks per k-row
  static constexpr int KSTEP = NT / CPR;        // k-rows covered by one pass of the block
  static constexpr int NI = BK / KSTEP;
  static_assert(KSTEP >= 1 && KSTEP <= BK && BK % KSTEP == 0, "K-strided loader: bad rows / threads combination");
  static constexpr int ROWB = ROWS * 2;
  static constexpr int NCH = ROWS / 16;         // 32-byte chunks per k-row
  static constexpr bool HAS_FULL = MODE == 0;
  const bf16_t* ptr;
  const bf16_t* safe;     // always-readable address for masked-off chunks
  long ld, tapstride;
  bool rok;
  int kh, kw, klog, kmask;
  float inv_rpi, inv_rw;
  // interior-tile fast path (MODE 0, K-steps that do not straddle a tap): uniform pointer + 32-bit lane offsets
  const char* tbase;
  unsigned voff[NI];
  bool full;

  __device__ __forceinline__ void init(const GemmP& p, const bf16_t* b, long ld_, int row0, int nrows, int tid, int tap,
                                       bool two_level) {
    const int r0 = row0 + (tid % CPR) * 8;
    rok = r0 < nrows;
    ptr = b + (rok ? r0 : 0);
    safe = b;
    ld = ld_;
    kh = tap / p.KW;
    kw = tap - kh * p.KW;
    klog = two_level ? p.b_klog : 30;
    kmask = two_level ? p.b_kmask : 0x7fffffff;
    tapstride = two_level ? p.b_tapstride : 0;
    inv_rpi = 1.f / (float)p.rows_per_img;
    inv_rw = 1.f / (float)p.row_w;
    if constexpr (HAS_FULL) {
      tbase = reinterpret_cast<const char*>(b + row0);
      full = row0 + ROWS <= nrows && klog >= 6;          // klog >= 6: the 64 k of a step share one tap
#pragma unroll
      for (int i = 0; i < NI; ++i) voff[i] = (unsigned)(((tid % CPR) * 8 + (long)(tid / CPR + i * KSTEP) * ld) * 2);
    } else {
      tbase = nullptr; full = false;
    }
  }

  __device__ __forceinline__ unsigned load(const GemmP& p, int k0, int kend, int tid, uint4 (&v)[NI]) const {
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = k0 + tid / CPR + i * KSTEP;
      bool valid = rok && k < kend;
      long a;
      if constexpr (MODE == 0) {
        a = (long)(k & kmask) * ld + (long)(k >> klog) * tapstride;
      } else {
        // pixel k -> (n, oh, ow) by float reciprocal + fix-up (k < 2^24)
        int n = (int)((float)k * inv_rpi);
        int rem = k - n * p.rows_per_img;
        if (rem < 0) { --n; rem += p.rows_per_img; } else if (rem >= p.rows_per_img) { ++n; rem -= p.rows_per_img; }
        int oh = (int)((float)rem * inv_rw);
        int ow = rem - oh * p.row_w;
        if (ow < 0) { --oh; ow += p.row_w; } else if (ow >= p.row_w) { ++oh; ow -= p.row_w; }
        const int ih = oh * p.sh - p.ph + kh, iw = ow * p.sw - p.pw + kw;
        valid = valid && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
        a = (long)n * p.sN + (long)ih * p.sH + (long)iw * p.sW;
      }
      v[i] = *reinterpret_cast<const uint4*>(valid ? ptr + a : safe);
      mask |= (valid ? 1u : 0u) << i;
    }
    return mask;
  }

  __device__ __forceinline__ void store(char* lds, int tid, const uint4 (&v)[NI], unsigned mask) const {
    const int m8 = tid % CPR;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int kr = tid / CPR + i * KSTEP;
      const int ch = (m8 >> 1) ^ (kswz(kr) & (NCH - 1));
      *reinterpret_cast<uint4*>(lds + kr * ROWB + (ch << 5) + ((m8 & 1) << 4)) = ((mask >> i) & 1u) ? v[i] : make_uint4(0, 0, 0, 0);
    }
  }

  __device__ __forceinline__ void load_full(const GemmP&, int k0, uint4 (&v)[NI]) const {
    const char* sb = tbase + ((long)(k0 & kmask) * ld + (long)(k0 >> klog) * tapstride) * 2;
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const uint4*>(sb + voff[i]);
  }
  __device__ __forceinline__ void store_full(char* lds, int tid, const uint4 (&v)[NI]) const {
    const int m8 = tid % CPR;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int kr = tid / CPR + i * KSTEP;
      const int ch = (m8 >> 1) ^ (kswz(kr) & (NCH - 1));
      *reinterpret_cast<uint4*>(lds + kr * ROWB + (ch << 5) + ((m8 & 1) << 4)) = v[i];
    }
  }

  // fragment of the 16 rows [w64 + 16 i, +16) for MFMA k-step ks: lane (frow, fq) gets k = 32 ks + 8 fq + 0..7
  static __device__ __forceinline__ bf16x8 frag(const char* tile, int w64, int i, int ks, int frow, int fq) {
    const int q = frow >> 2, pp = frow & 3;
    const int k1 = ks * 32 + 8 * fq + q;
    const int ch = (((w64 >> 4) + i) ^ (kswz(k1) & (NCH - 1)));
    const char* a = tile + k1 * ROWB + (ch << 5) + (pp << 3);
    const s16x4 lo = lds_tr_read(a);
    const s16x4 hi = lds_tr_read(a + 4 * ROWB);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
  }
};

template <int ROWS, int MODE, bool IS_A, int NT> struct LoaderSel;
template <int ROWS, int NT> struct LoaderSel<ROWS, EVK_A_PLAIN, true, NT> { using T = RowLoader<ROWS, EVK_A_PLAIN, NT>; static constexpr bool KS = false; };
template <int ROWS, int NT> struct LoaderSel<ROWS, EVK_A_CONV, true, NT> { using T = RowLoader<ROWS, EVK_A_CONV, NT>; static constexpr bool KS = false; };
template <int ROWS, int NT> struct LoaderSel<ROWS, EVK_A_DGRAD, true, NT> { using T = RowLoader<ROWS, EVK_A_DGRAD, NT>; static constexpr bool KS = false; };
template <int ROWS, int NT> struct LoaderSel<ROWS, EVK_A_KSTR, true, NT> { using T = KstrLoader<ROWS, 0, NT>; static constexpr bool KS = true; };
template <int ROWS, int NT> struct LoaderSel<ROWS, EVK_B_PLAIN, false, NT> { using T = RowLoader<ROWS, EVK_A_PLAIN, NT>; static constexpr bool KS = false; };
template <int ROWS, int NT> struct LoaderSel<ROWS, EVK_B_KSTR, false, NT> { using T = KstrLoader<ROWS, 0, NT>; static constexpr bool KS = true; };
template <int ROWS, int NT> struct LoaderSel<ROWS, EVK_B_WGATHER, false, NT> { using T = KstrLoader<ROWS, 1, NT>; static constexpr bool KS = true; };

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4); every lane of the row ends up with the total
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
  return v + __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);    // row_half_mirror
  return dpp_add<0x140>(v); // row_mirror
}

// ---- epilogue shared by the GEMM kernels: lane holds C[m][n0..n0+3], m = ..+(lane&15), n0 = ..+(lane>>4)*4 ----
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x4 (&acc)[4][4], int tm, int tn, int wm, int wn, int frow, int fq,
                                              int zo, int zi, int by, int bz, char* smem) {
  // bf16 outputs can leave through LDS: the MFMA layout gives every lane 4 consecutive n of one row (8-byte stores, 16 rows x
  // 32 B per wave instruction); staged through a wave-private 64 x 64 tile the wave stores 8 whole 128-byte rows at a time
  const bool stage = p.lds_store && !p.accumulate && !p.c_f32;
  char* stg = smem + (wm * (TN / 64) + wn) * 8192;
  if (stage) __syncthreads();          // every wave is done reading the operand tiles
  if (p.colstats) {
    // Batch-norm statistics of a convolution output, taken from the f32 accumulators: every wave reduces its 64 rows
    // (4 in-lane sub-tiles, then the 16 lanes of a DPP row) and writes one partial row [2][N]; rows beyond M are zero
    // because their A rows were zero-filled.  bn.hip's final stage sums the partials.
    float* prow = p.colstats + ((long)(tm * (TM / 64) + wm)) * 2 * p.N;
#pragma unroll
    for (int in = 0; in < 4; ++in) {
      float sm[4], sq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int im = 0; im < 4; ++im) { const float v = acc[in][im][j] * p.alpha; a += v; b += v * v; }
        sm[j] = row16_sum(a);
        sq[j] = row16_sum(b);
      }
      const int n0 = tn * TN + wn * 64 + in * 16 + fq * 4;
      if (frow == 0 && n0 < p.N) {
        if ((p.N & 3) == 0) {
          *reinterpret_cast<float4*>(prow + n0) = make_float4(sm[0], sm[1], sm[2], sm[3]);
          *reinterpret_cast<float4*>(prow + p.N + n0) = make_float4(sq[0], sq[1], sq[2], sq[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (n0 + j < p.N) { prow[n0 + j] = sm[j]; prow[p.N + n0 + j] = sq[j]; }
        }
      }
    }
  }
  // batch-norm backward sums of the NEXT layer down, taken from this data-gradient GEMM's epilogue: the output is the gradient
  // g w.r.t. z = relu(gamma * xhat + beta) of that layer, gated by z > 0, and where the gate is open xhat = (z - beta) / gamma --
  // so sum(g) and sum(g * z) per column are all its backward needs (bn.hip: bn_bwd_sums_from_gate_kernel), at no extra traffic:
  // the gate tile is loaded anyway.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 gs[4][2], gz[4][2];
#pragma unroll
  for (int in = 0; in < 4; ++in)
#pragma unroll
    for (int h = 0; h < 2; ++h) { gs[in][h] = f32x2{0.f, 0.f}; gz[in][h] = f32x2{0.f, 0.f}; }
  char* Cb = reinterpret_cast<char*>(p.C) + (zo * p.sCo + zi * p.sCi) * (p.c_f32 ? 4 : 2);
  const char* Rb = p.resid ? reinterpret_cast<const char*>(p.resid) + (zo * p.sRo + zi * p.sRi) * (p.r_f32 ? 4 : 2) : nullptr;
#pragma unroll
  for (int im = 0; im < 4; ++im) {
    const int m = tm * TM + wm * 64 + im * 16 + frow;
    if (m >= p.M) continue;
#pragma unroll
    for (int in = 0; in < 4; ++in) {
      const int n0 = tn * TN + wn * 64 + in * 16 + fq * 4;
      if (n0 >= p.N) continue;
      float v[4] = {acc[in][im][0] * p.alpha, acc[in][im][1] * p.alpha, acc[in][im][2] * p.alpha, acc[in][im][3] * p.alpha};
      const bool full = p.vec_ok && (n0 + 3 < p.N);
      if (p.accumulate) {
        if (p.slab) {          // split-K partial: plain row-contiguous stores, summed into C by splitk_reduce_kernel
          float* c = p.slab + ((long)bz * gridDim.y + by) * p.slab_mn + (long)m * p.N + n0;
          if ((p.N & 3) == 0 && n0 + 3 < p.N) *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n0 + j < p.N) c[j] = v[j];
          }
        } else if (gridDim.y == 1) {   // this block owns the tile: plain read-modify-write
          float* c = reinterpret_cast<float*>(Cb) + (long)m * p.ldc + n0;
          if (full) { float4 t = *reinterpret_cast<float4*>(c); t.x += v[0]; t.y += v[1]; t.z += v[2]; t.w += v[3]; *reinterpret_cast<float4*>(c) = t; }
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n0 + j < p.N) c[j] += v[j];
          }
        } else {               // no workspace given: f32 atomics (slow access shape, kept as a fallback)
          float* c = reinterpret_cast<float*>(Cb) + (long)m * p.ldc + n0;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n0 + j < p.N) unsafeAtomicAdd(c + j, v[j]);
        }
        continue;
      }
      if (p.bias) {
        const float* bias = p.bias + zi * p.sbias;       // per inner-batch bias rows (sbias = 0: shared)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n0 + j < p.N) v[j] += bias[n0 + j];
      }
      if (p.act != EVK_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], p.act);
      }
      if (Rb) {
        if (p.r_f32) {
          const float* r = reinterpret_cast<const float*>(Rb) + (long)m * p.ldr + n0;
          if (full) { const float4 t = *reinterpret_cast<const float4*>(r); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n0 + j < p.N) v[j] += r[j];
          }
        } else {
          const bf16_t* r = reinterpret_cast<const bf16_t*>(Rb) + (long)m * p.ldr + n0;
          if (full) { const uint2 t = *reinterpret_cast<const uint2*>(r); v[0] += lo_bf(t.x); v[1] += hi_bf(t.x); v[2] += lo_bf(t.y); v[3] += hi_bf(t.y); }
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n0 + j < p.N) v[j] += bf2f(r[j]);
          }
        }
      }
      if (p.gate) {        // gradient of a ReLU whose output is `gate`: zero where the forward activation was clipped
        const bf16_t* gt = p.gate + (long)m * p.ldg + n0;
        float gv[4] = {0.f, 0.f, 0.f, 0.f};
        if (full && (p.ldg & 3) == 0) {
          const uint2 t = *reinterpret_cast<const uint2*>(gt);
          gv[0] = lo_bf(t.x); gv[1] = hi_bf(t.x); gv[2] = lo_bf(t.y); gv[3] = hi_bf(t.y);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (n0 + j < p.N) gv[j] = bf2f(gt[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (!(gv[j] > 0.f)) v[j] = 0.f;
        if (p.gatestats) {
          // Explicit two-wide vectors in natural order (low = even column).  Left to the SLP vectoriser these scalar accumulations were
          // paired in SWAPPED order -- v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0], the low result reading the HIGH half of the other
          // operand -- and on gfx950 that form returned a stale high half for lanes 48-63 about once in 10^6 results: one partial entry
          // off by ~1 every few launches, same inputs (tools/gatestats_determinism.py; found by tests/test_model_gpu.py::
          // test_full_size_step_properties).  tests/test_abi.py checks the built code objects for that instruction form.
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 vv = {v[2 * h], v[2 * h + 1]}, gg = {gv[2 * h], gv[2 * h + 1]};
            gs[in][h] += vv;
            gz[in][h] += vv * gg;
          }
        }
      }
      if (p.c_f32) {
        float* c = reinterpret_cast<float*>(Cb) + (long)m * p.ldc + n0;
        if (full) *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (n0 + j < p.N) c[j] = v[j];
        }
      } else if (stage) {
        const int row = im * 16 + frow, q8 = in * 4 + fq;
        *reinterpret_cast<uint2*>(stg + row * 128 + ((((q8 >> 1) ^ (row & 7)) << 4) | ((q8 & 1) << 3))) =
            make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      } else {
        bf16_t* c = reinterpret_cast<bf16_t*>(Cb) + (long)m * p.ldc + n0;
        if (full) *reinterpret_cast<uint2*>(c) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (n0 + j < p.N) c[j] = f2bf(v[j]);
        }
      }
    }
  }
  if (p.gatestats) {
    float* prow = p.gatestats + ((long)(tm * (TM / 64) + wm)) * 2 * p.N;
#pragma unroll
    for (int in = 0; in < 4; ++in) {
      float a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = row16_sum(gs[in][j >> 1][j & 1]); b[j] = row16_sum(gz[in][j >> 1][j & 1]); }
      const int n0 = tn * TN + wn * 64 + in * 16 + fq * 4;
      if (frow == 0 && n0 < p.N) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n0 + j < p.N) { prow[n0 + j] = a[j]; prow[p.N + n0 + j] = b[j]; }
      }
    }
  }
  if (stage) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int lane = frow + 16 * fq;
    const int ch = lane & 7;
    const int ncol = tn * TN + wn * 64 + ch * 8;
    bf16_t* cb = reinterpret_cast<bf16_t*>(Cb);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3);
      const int m = tm * TM + wm * 64 + row;
      if (m < p.M && ncol < p.N)
        *reinterpret_cast<uint4*>(cb + (long)m * p.ldc + ncol) = *reinterpret_cast<const uint4*>(stg + row * 128 + ((ch ^ (row & 7)) << 4));
    }
  }
}

template <int WM, int WN, int AMODE, int BMODE, bool SB>
__global__ __launch_bounds__(64 * WM * WN, WM * WN > 4 ? 1 : ((SB && WM == 2) ? 3 : 2)) void gemm_kernel(const GemmP p) {
  constexpr int TM = 64 * WM, TN = 64 * WN, NT = 64 * WM * WN;
  constexpr int TILE_BYTES = (TM + TN) * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;

  // XCD-aware bijective remap (cdna_hip_programming.md T1): blocks b, b+8, ... share an XCD/L2, give
  // each XCD a contiguous run of tiles so neighbouring tiles re-use the same operand panels in L2.
  int wg, by = blockIdx.y, bz = blockIdx.z;
  if (p.kslice_xcd && gridDim.y > 1 && (gridDim.y & 7) == 0) {
    // split-K launch (weight gradients: small M x N, long K): every tile / tap of one K-slice reads the same operand
    // slice, so give each XCD whole K-slices -- slice s lives on XCD s % 8 and is fetched into that L2 once.
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int per = gridDim.x * gridDim.z;               // blocks per K-slice
    const int xcd = lin & 7, j = lin >> 3;
    const int sl = j / per, w = j - sl * per;
    by = xcd + 8 * sl;
    bz = w / gridDim.x;
    wg = w - bz * gridDim.x;
  } else {
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // Tile order.  The blocks resident on one XCD at a time (2-3 per CU x 32 CUs) are consecutive wg: M-major order makes that wave one
  // M-tile row x many N-tiles, so with N wide every B panel is live in one block only and is fetched again for each M row the XCD
  // owns (measured: 4640 x 16384 x 2048 fetched 1.33 GB for 86 MB of operands).  Super-rows of group_m M-tiles swept N-first make
  // the wave group_m x (wave / group_m) tiles: each live A panel serves wave/group_m blocks and each B panel group_m of them.
  int tm, tn;
  if (p.group_m > 1 && p.tilesN > 1) {
    const int tilesM = gridDim.x / p.tilesN;
    const int gsz = p.group_m * p.tilesN;
    const int gi = wg / gsz, first = gi * p.group_m;
    const int rows = min(tilesM - first, p.group_m);
    const int r = wg - gi * gsz;
    tn = r / rows;
    tm = first + (r - tn * rows);
  } else {
    tm = wg / p.tilesN; tn = wg - tm * p.tilesN;
  }
  const int zo = bz / p.bi, zi = bz - zo * p.bi;
  const int k_begin = by * p.ksteps_per_split * BK;
  const int k_end = min(p.K, k_begin + p.ksteps_per_split * BK);
  if (k_begin >= k_end) return;

  using LA = typename LoaderSel<TM, AMODE, true, NT>::T;
  using LB = typename LoaderSel<TN, BMODE, false, NT>::T;
  constexpr bool AKS = LoaderSel<TM, AMODE, true, NT>::KS;
  constexpr bool BKS = LoaderSel<TN, BMODE, false, NT>::KS;
  LA la;
  LB lb;
  const bf16_t* Ab = p.A + zo * p.sAo + zi * p.sAi;
  const bf16_t* Bb = p.B + zo * p.sBo + zi * p.sBi;
  if constexpr (AKS) la.init(p, Ab, p.lda, tm * TM, p.M, tid, 0, false);
  else la.init(p, Ab, p.lda, tm * TM, p.M, tid);
  if constexpr (BKS) lb.init(p, Bb, p.ldb, tn * TN, p.N, tid, zi, true);
  else lb.init(p, Bb, p.ldb, tn * TN, p.N, tid);

  uint4 ra[LA::NI];
  uint4 rb[LB::NI];

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int frow = lane & 15, fq = lane >> 4;

  auto compute = [&](const char* As, const char* Bs) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (AKS) af[i] = LA::frag(As, wm * 64, i, ks, frow, fq);
        else {
          const int row = wm * 64 + i * 16 + frow;
          af[i] = *reinterpret_cast<const bf16x8*>(As + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (BKS) bfr[i] = LB::frag(Bs, wn * 64, i, ks, frow, fq);
        else {
          const int row = wn * 64 + i * 16 + frow;
          bfr[i] = *reinterpret_cast<const bf16x8*>(Bs + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
        }
      }
#pragma unroll
      for (int in = 0; in < 4; ++in)
#pragma unroll
        for (int im = 0; im < 4; ++im)
          acc[in][im] = EVK_MFMA_16x16x32(bfr[in], af[im], acc[in][im], 0, 0, 0);
    }
  };

  // FA / FB: the operand's tile is interior (every row valid, the K range a whole number of steps) -> loads and LDS stores
  // without validity selects and with scalar K advance (measured: the masked path spends ~70 of its ~118 VALU instructions
  // per K-step on selects and 64-bit address arithmetic, and these kernels are issue-bound at 3 waves per SIMD)
  auto mainloop = [&](auto FA, auto FB) {
    constexpr bool fa = decltype(FA)::value, fb = decltype(FB)::value;
    unsigned ma = 0, mb = 0;
    auto loadA = [&](int k0) { if constexpr (fa) la.load_full(p, k0, ra); else ma = la.load(p, k0, k_end, tid, ra); };
    auto loadB = [&](int k0) { if constexpr (fb) lb.load_full(p, k0, rb); else mb = lb.load(p, k0, k_end, tid, rb); };
    auto storeA = [&](char* d) { if constexpr (fa) la.store_full(d, tid, ra); else la.store(d, tid, ra, ma); };
    auto storeB = [&](char* d) { if constexpr (fb) lb.store_full(d, tid, rb); else lb.store(d, tid, rb, mb); };
    loadA(k_begin);
    loadB(k_begin);
    if constexpr (SB) {
      // single LDS buffer (32-40 KB/block -> 4 blocks per CU): the next tile's global loads are in flight in registers
      // while this tile is multiplied; two barriers per K-step
      for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        __syncthreads();
        storeA(smem);
        storeB(smem + TM * 128);
        __syncthreads();
        if (k0 + BK < k_end) {
          loadA(k0 + BK);
          loadB(k0 + BK);
        }
        compute(smem, smem + TM * 128);
      }
    } else {
      storeA(smem);
      storeB(smem + TM * 128);
      __syncthreads();
      int buf = 0;
      for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = (k0 + BK) < k_end;
        if (more) {
          loadA(k0 + BK);
          loadB(k0 + BK);
        }
        const char* As = smem + buf * TILE_BYTES;
        compute(As, As + TM * 128);
        if (more) {
          char* nxt = smem + (buf ^ 1) * TILE_BYTES;
          storeA(nxt);
          storeB(nxt + TM * 128);
        }
        __syncthreads();
        buf ^= 1;
      }
    }
  };
  const bool kfull = p.fast_loads && ((k_end - k_begin) % BK) == 0;
  const bool fullA = LA::HAS_FULL && kfull && la.full, fullB = LB::HAS_FULL && kfull && lb.full;
  using T1 = std::integral_constant<bool, true>;
  using T0 = std::integral_constant<bool, false>;
  if constexpr (LA::HAS_FULL && LB::HAS_FULL) {
    if (fullA && fullB) mainloop(T1{}, T1{});
    else mainloop(T0{}, T0{});
  } else if constexpr (LB::HAS_FULL) {
    if (fullB) mainloop(T0{}, T1{});
    else mainloop(T0{}, T0{});
  } else {
    mainloop(T0{}, T0{});
  }

  gemm_epilogue<TM, TN>(p, acc, tm, tn, wm, wn, frow, fq, zo, zi, by, bz, smem);
}


// ------------------------------------------------------------------------------------------------
// latency-optimised kernel for small problems (the relational-memory recurrence, decode steps):
// C[M][N] = act(alpha*A[M][K].B[N][K]^T + bias) + resid with A_PLAIN / B_PLAIN, K % 256 == 0.
// Tile 128(M) x 32(N) per block so a 96 x 512 problem spreads over 16 CUs, and K is consumed in 256-deep chunks with
// ALL loads of a chunk in flight at once (80 KB of LDS per chunk): a K = 512 product pays 2 global-load latencies
// instead of the 8 of the throughput kernel's 64-deep pipeline.
// ------------------------------------------------------------------------------------------------
constexpr int SK_KC = 256;                   // the K granularity the launcher requires
template <int KC>
__device__ __forceinline__ int sk_off(int row, int chunk) { return row * (KC * 2) + ((chunk ^ (row & 15)) << 4); }

// KC = K-chunk depth (all loads of a chunk in flight at once), TN = output columns per block (16 or 32), TM = rows per block
// (128, or 64: these launches are bound by the LDS-fill bytes of ONE CU -- A panel TM x KC + B panel TN x KC at ~25 GB/s per CU,
// gemm.hip use_big_tile() -- so for <= 256 rows half the panel on twice the CUs is faster)
template <int KC, int TN, int TM>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const GemmP p) {
  constexpr int CPR = KC / 8;                    // 16-byte chunks per row
  constexpr int NA = TM * CPR / 256, NB = TN * CPR / 256, NT = TN / 16, MT = TM / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                               // [TM][KC] bf16, chunk index XOR (row & 15)
  char* Bs = smem + TM * KC * 2;                 // [TN][KC]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  uint4 ra[NA], rb[NB];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + 256 * i, row = c / CPR, ch = c % CPR;
      const int m = m0 + row;
      ra[i] = m < p.M ? *reinterpret_cast<const uint4*>(p.A + (long)m * p.lda + k0 + ch * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = tid + 256 * i, row = c / CPR, ch = c % CPR;
      const int n = n0 + row;
      rb[i] = n < p.N ? *reinterpret_cast<const uint4*>(p.B + (long)n * p.ldb + k0 + ch * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) { const int c = tid + 256 * i; *reinterpret_cast<uint4*>(As + sk_off<KC>(c / CPR, c % CPR)) = ra[i]; }
#pragma unroll
    for (int i = 0; i < NB; ++i) { const int c = tid + 256 * i; *reinterpret_cast<uint4*>(Bs + sk_off<KC>(c / CPR, c % CPR)) = rb[i]; }
  };
  f32x4 acc[NT][MT];    // [n tile][m tile]; wave owns rows wave*(TM/4) .. +TM/4
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  load(0);
  if constexpr (KC == 512) {
    // LayerNorm prologue (K == 512 == the normalised width): thread (wave, lane) holds chunk `lane` of rows wave + 4 i -- the element
    // mapping of norm.hip's ln_fwd_kernel, whose arithmetic is repeated here term for term (sum of the lane's 8 values, wave_sum,
    // squared deviations, wave_sum, (v - mu) * r * g + b) so that the 16-bit rows entering LDS are the ones that kernel would have
    // written to memory; the norm launch in front of the GEMM and its round trip disappear (decode step: 10 per generated token).
    if (p.ln_g) {
      float g8[8], b8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { g8[j] = p.ln_g[lane * 8 + j]; b8[j] = p.ln_b[lane * 8 + j]; }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int m = m0 + wave + 4 * i;
        if (m >= p.M) continue;                        // wave-uniform
        const uint32_t w4[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
        float v[8], g[8], b[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = lo_bf(w4[j]); v[2 * j + 1] = hi_bf(w4[j]); }
#pragma unroll
        for (int j = 0; j < 8; ++j) { g[j] = g8[j]; b[j] = b8[j]; }
        if (p.ln_dg) {
          const uint4 dg = *reinterpret_cast<const uint4*>(p.ln_dg + (long)m * p.ln_ld + lane * 8);
          const uint4 db = *reinterpret_cast<const uint4*>(p.ln_db + (long)m * p.ln_ld + lane * 8);
          const uint32_t dgw[4] = {dg.x, dg.y, dg.z, dg.w}, dbw[4] = {db.x, db.y, db.z, db.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) { g[2 * j] += lo_bf(dgw[j]); g[2 * j + 1] += hi_bf(dgw[j]); b[2 * j] += lo_bf(dbw[j]); b[2 * j + 1] += hi_bf(dbw[j]); }
        }
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sm += v[j];
        const float mu = wave_sum(sm) / 512;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[j] - mu; q += d * d; }
        q = wave_sum(q);
        const float r = p.ln_mode == 0 ? rsqrtf(q / 512 + p.ln_eps) : 1.f / (sqrtf(q / (512 - 1)) + p.ln_eps);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[j] - mu) * r * g[j] + b[j];
        ra[i] = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
      }
    }
  }
  for (int k0 = 0; k0 < p.K; k0 += KC) {
    store();
    __syncthreads();
    if (k0 + KC < p.K) load(k0 + KC);
#pragma unroll
    for (int ks = 0; ks < KC / 32; ++ks) {
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + sk_off<KC>(wave * (TM / 4) + i * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < NT; ++i) bfr[i] = *reinterpret_cast<const bf16x8*>(Bs + sk_off<KC>(i * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int in = 0; in < NT; ++in)
#pragma unroll
        for (int im = 0; im < MT; ++im)
          acc[in][im] = EVK_MFMA_16x16x32(bfr[in], af[im], acc[in][im], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int im = 0; im < MT; ++im) {
    const int m = m0 + wave * (TM / 4) + im * 16 + frow;
    if (m >= p.M) continue;
#pragma unroll
    for (int in = 0; in < NT; ++in) {
      const int nn = n0 + in * 16 + fq * 4;
      if (nn >= p.N) continue;
      float v[4] = {acc[in][im][0] * p.alpha, acc[in][im][1] * p.alpha, acc[in][im][2] * p.alpha, acc[in][im][3] * p.alpha};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (nn + j < p.N) {
          if (p.bias) v[j] += p.bias[nn + j];
          v[j] = act_apply(v[j], p.act);
          if (p.resid) v[j] += p.r_f32 ? reinterpret_cast<const float*>(p.resid)[(long)m * p.ldr + nn + j]
                                       : bf2f(reinterpret_cast<const bf16_t*>(p.resid)[(long)m * p.ldr + nn + j]);
          if (p.gate && !(bf2f(p.gate[(long)m * p.ldg + nn + j]) > 0.f)) v[j] = 0.f;          // relu_gate, after resid (as the tile kernel)
          if (p.c_f32) reinterpret_cast<float*>(p.C)[(long)m * p.ldc + nn + j] = v[j];
          else reinterpret_cast<bf16_t*>(p.C)[(long)m * p.ldc + nn + j] = f2bf(v[j]);
        }
      }
    }
  }
}

template <int KC, int TN, int TM>
int launch_skinny_cfg(const GemmP& p, hipStream_t s) {
  constexpr int LDS = (TM + TN) * KC * 2;
  auto kern = gemm_skinny_kernel<KC, TN, TM>;
  EVK_DYN_LDS_ONCE(kern, LDS);
  dim3 grid((unsigned)cdiv(p.N, TN), (unsigned)cdiv(p.M, TM), 1);
  hipLaunchKernelGGL(kern, grid, dim3(NTHR), LDS, s, p);
  return evk_check_launch("gemm_skinny_kernel");
}

int launch_skinny(const GemmP& p, hipStream_t s) {
  // <= 256 rows (relational memory, decode step): the whole K = 512 panel in flight at once and 16-column blocks (twice the
  // blocks) -- one memory round trip per 512 of K instead of two.  Larger M: 2x the LDS per block would halve the residency.
  static const int deep = evk_tunable("EVK_SKINNY_DEEP", 1);
  static const int half = evk_tunable("EVK_SKINNY_TM64", 1);
  // (<= 1024 rows: the decode step's 768-row relational-memory products, 0.439 -> 0.419 ms per token)
  static const int deep_rows = evk_tunable("EVK_SKINNY_DEEP_ROWS", 1024);
  if (deep && p.K % 512 == 0 && p.M <= deep_rows) return half ? launch_skinny_cfg<512, 16, 64>(p, s) : launch_skinny_cfg<512, 16, 128>(p, s);
  return launch_skinny_cfg<256, 32, 128>(p, s);
}

// ------------------------------------------------------------------------------------------------
// gemm_small_kernel: weight gradients with M <= 64 output rows (the 64-channel convolutions of layer1 and the stem): a
// 64 x 64 output tile per block, both operands K-strided.  The throughput kernel's smallest tile (256 x 64 / 128 x 128)
// wastes 75 % of its MFMAs and of its block slots on such problems.  Here the four waves split every K-step instead of
// the tile: wave = (half, ksel) multiplies MFMA k-step `ksel` of the 64-deep tile into the 32 output columns `half`, and
// the two k-step partials leave as two split-K slabs that splitk_reduce_kernel sums anyway.
// ------------------------------------------------------------------------------------------------
template <int BMODE>
__global__ __launch_bounds__(NTHR) void gemm_small_kernel(const GemmP p) {
  constexpr int TILE_BYTES = (64 + 64) * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  int wg = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if ((gridDim.y & 7) == 0) {
    // every tile and every TAP (batch index) of one K-slice reads the same pixel rows of dY and X: slice s lives on XCD s % 8, so the
    // nine taps of a 3x3 weight gradient fetch their slice into that L2 once (measured before: 1.14 GB fetched for 150 MB of operands)
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int per = gridDim.x * gridDim.z;               // blocks per K-slice
    const int xcd = lin & 7, j = lin >> 3;
    const int sl = j / per, w = j - sl * per;
    by = xcd + 8 * sl;
    bz = w / gridDim.x;
    wg = w - bz * gridDim.x;
  }
  const int tm = wg / p.tilesN, tn = wg - tm * p.tilesN;
  const int zo = bz / p.bi, zi = bz - zo * p.bi;
  const int k_begin = by * p.ksteps_per_split * BK;
  const int k_end = min(p.K, k_begin + p.ksteps_per_split * BK);
  using LA = KstrLoader<64, 0>;
  using LB = KstrLoader<64, BMODE == EVK_B_WGATHER ? 1 : 0>;
  LA la;
  LB lb;
  la.init(p, p.A + zo * p.sAo + zi * p.sAi, p.lda, tm * 64, p.M, tid, 0, false);
  lb.init(p, p.B + zo * p.sBo + zi * p.sBi, p.ldb, tn * 64, p.N, tid, zi, true);
  uint4 ra[LA::NI], rb[LB::NI];
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane = tid & 63, wave = tid >> 6;
  const int ksel = wave & 1, half = wave >> 1;
  const int frow = lane & 15, fq = lane >> 4;

  unsigned ma = la.load(p, k_begin, k_end, tid, ra);
  unsigned mb = lb.load(p, k_begin, k_end, tid, rb);
  la.store(smem, tid, ra, ma);
  lb.store(smem + 64 * 128, tid, rb, mb);
  __syncthreads();
  int buf = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    const bool more = (k0 + BK) < k_end;
    if (more) {
      ma = la.load(p, k0 + BK, k_end, tid, ra);
      mb = lb.load(p, k0 + BK, k_end, tid, rb);
    }
    const char* As = smem + buf * TILE_BYTES;
    const char* Bs = As + 64 * 128;
    bf16x8 af[4], bfr[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = LA::frag(As, 0, i, ksel, frow, fq);
#pragma unroll
    for (int j = 0; j < 2; ++j) bfr[j] = LB::frag(Bs, 0, half * 2 + j, ksel, frow, fq);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int im = 0; im < 4; ++im) acc[j][im] = EVK_MFMA_16x16x32(bfr[j], af[im], acc[j][im], 0, 0, 0);
    if (more) {
      char* nxt = smem + (buf ^ 1) * TILE_BYTES;
      la.store(nxt, tid, ra, ma);
      lb.store(nxt + 64 * 128, tid, rb, mb);
    }
    __syncthreads();
    buf ^= 1;
  }
  float* slab = p.slab + (((long)bz * gridDim.y + by) * 2 + ksel) * p.slab_mn;
#pragma unroll
  for (int im = 0; im < 4; ++im) {
    const int m = tm * 64 + im * 16 + frow;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n0 = tn * 64 + (half * 2 + j) * 16 + fq * 4;
      if (n0 >= p.N) continue;                       // N % 4 == 0 on this path
      *reinterpret_cast<float4*>(slab + (long)m * p.N + n0) =
          make_float4(acc[j][im][0] * p.alpha, acc[j][im][1] * p.alpha, acc[j][im][2] * p.alpha, acc[j][im][3] * p.alpha);
    }
  }
}

constexpr int SMALL_TARGET_BLOCKS = 768;

inline int small_splitk(int M, int N, int K, int batch) {
  const long tiles = cdiv(M, 64) * cdiv(N, 64) * batch;
  const int ksteps = (int)cdiv(K, BK);
  long sk = cdiv(SMALL_TARGET_BLOCKS, tiles);
  if (sk > ksteps) sk = ksteps;
  if (sk < 1) sk = 1;
  if (sk >= 8) {            // whole K-slices per XCD (gemm_small_kernel's block remap): a multiple of 8 non-empty slices
    for (long s8 = (sk + 4) / 8 * 8; s8 >= 8; s8 -= 8) {
      const int per = (int)cdiv(ksteps, s8);
      if (cdiv(ksteps, per) == s8) return (int)s8;
    }
  }
  const int per = (int)cdiv(ksteps, sk);
  return (int)cdiv(ksteps, per);
}

inline bool small_eligible(int M, int N, int a_mode, int accumulate) { return accumulate && a_mode == EVK_A_KSTR && M <= 64 && (N % 4) == 0; }

// C[z][m][n] += sum_split slab[z][split][m][n].  Block = 16 float4 columns x 16 split lanes.
struct SkrP { const float* slab; float* C; long mn; int M, N, splitk, bi; long ldc, sCo, sCi; };
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const SkrP p) {
  __shared__ float4 red[16][17];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long q = (long)blockIdx.x * 16 + e;          // float4 index within the M x N slab (N % 4 == 0)
  const long nq = p.mn >> 2;
  const int z = blockIdx.y;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < nq) {
    const float4* base = reinterpret_cast<const float4*>(p.slab + (long)z * p.splitk * p.mn) + q;
    int sp = sl;
    for (; sp + 48 < p.splitk; sp += 64) {
      const float4 t0 = base[(long)sp * nq], t1 = base[(long)(sp + 16) * nq], t2 = base[(long)(sp + 32) * nq], t3 = base[(long)(sp + 48) * nq];
      a.x += (t0.x + t1.x) + (t2.x + t3.x); a.y += (t0.y + t1.y) + (t2.y + t3.y);
      a.z += (t0.z + t1.z) + (t2.z + t3.z); a.w += (t0.w + t1.w) + (t2.w + t3.w);
    }
    for (; sp < p.splitk; sp += 16) { const float4 t = base[(long)sp * nq]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
  }
  red[sl][e] = a;
  __syncthreads();
  if (sl == 0 && q < nq) {
    float4 s4 = red[0][e];
#pragma unroll
    for (int i = 1; i < 16; ++i) { const float4 t = red[i][e]; s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w; }
    const long idx = q << 2;
    const int m = (int)(idx / p.N), n = (int)(idx - (long)m * p.N);
    const int zo = z / p.bi, zi = z - zo * p.bi;
    float* c = p.C + zo * p.sCo + zi * p.sCi + (long)m * p.ldc + n;
    c[0] += s4.x; c[1] += s4.y; c[2] += s4.z; c[3] += s4.w;
  }
}

// The same sum for large outputs: one float4 column per thread, 256 consecutive columns per block (whole 4 KB runs of every slab),
// four slabs in flight per thread.  Order of the additions: ((s0 + s1) + (s2 + s3)) per group of four slabs, groups in order.
__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(const SkrP p) {
  const long nq = p.mn >> 2;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const int z = blockIdx.y;
  const float4* base = reinterpret_cast<const float4*>(p.slab + (long)z * p.splitk * p.mn) + q;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int sp = 0;
  for (; sp + 3 < p.splitk; sp += 4) {
    const float4 t0 = base[(long)sp * nq], t1 = base[(long)(sp + 1) * nq], t2 = base[(long)(sp + 2) * nq], t3 = base[(long)(sp + 3) * nq];
    a.x += (t0.x + t1.x) + (t2.x + t3.x); a.y += (t0.y + t1.y) + (t2.y + t3.y);
    a.z += (t0.z + t1.z) + (t2.z + t3.z); a.w += (t0.w + t1.w) + (t2.w + t3.w);
  }
  for (; sp < p.splitk; ++sp) { const float4 t = base[(long)sp * nq]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
  const long idx = q << 2;
  const int m = (int)(idx / p.N), n = (int)(idx - (long)m * p.N);
  const int zo = z / p.bi, zi = z - zo * p.bi;
  float* c = p.C + zo * p.sCo + zi * p.sCi + (long)m * p.ldc + n;
  if ((reinterpret_cast<uintptr_t>(c) & 15) == 0) {
    float4 t = *reinterpret_cast<float4*>(c);
    t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
    *reinterpret_cast<float4*>(c) = t;
  } else {
    c[0] += a.x; c[1] += a.y; c[2] += a.z; c[3] += a.w;
  }
}

// one launch of the reduction: the wide kernel once the output gives every CU a few blocks of whole 4 KB runs (EVK_REDUCE_WIDE=0/1 forces)
inline int launch_splitk_reduce(const SkrP& r, int batch, hipStream_t s) {
  static const int mode = evk_tunable("EVK_REDUCE_WIDE", -1);
  static const bool probe_skip = evk_tunable("EVK_PROBE_SKIP_SPLITK_REDUCE", 0) != 0;          // timing probe: wrong weight gradients
  if (probe_skip) return EVK_OK;
  const long nq = r.mn >> 2;
  const bool wide = mode >= 0 ? mode == 1 : cdiv(nq, 256) * batch >= 192;
  if (wide) hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3((int)cdiv(nq, 256), batch), dim3(256), 0, s, r);
  else hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)cdiv(nq, 16), batch), dim3(256), 0, s, r);
  return evk_check_launch("splitk_reduce");
}

constexpr long SLAB_MAX_BYTES = 192L << 20;

// LDS staging per operand mode (measured on MI355X, FineTune 384^2 bs 32, profiles/r01_*): a single 32-40 KB buffer
// (4 blocks/CU, two barriers per K-step) wins for the K-contiguous A loaders (fwd GEMMs +10 %, conv fwd +23 %, dX +15 %);
// the double buffer (2 blocks/CU, one barrier) wins for the gather-from-dY and K-strided A loaders (dgrad, wgrad: 5-10 %).
// EVK_GEMM_SB=0/1 forces one of them for experiments.
inline int single_buf_override() {
  static int v = -2;
  if (v == -2) { const int t = evk_tunable("EVK_GEMM_SB", -1); v = t < 0 ? -1 : (t ? 1 : 0); }
  return v;
}

template <int WM, int WN, int AMODE, int BMODE, bool SB>
int launch_cfg_sb(const GemmP& p, dim3 grid, hipStream_t s) {
  constexpr int TILE_LDS = (SB ? 1 : 2) * (64 * WM + 64 * WN) * BK * 2;
  constexpr int STAGE_LDS = WM * WN * 8192;            // the epilogue's per-wave 64 x 64 bf16 staging tiles
  constexpr int LDS = TILE_LDS > STAGE_LDS ? TILE_LDS : STAGE_LDS;
  auto kern = gemm_kernel<WM, WN, AMODE, BMODE, SB>;
  EVK_DYN_LDS_ONCE(kern, LDS);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), LDS, s, p);
  return evk_check_launch("gemm_kernel");
}

template <int WM, int WN, int AMODE, int BMODE>
int launch_cfg(const GemmP& p, dim3 grid, hipStream_t s) {
  const int ov = single_buf_override();
  const bool sb = ov >= 0 ? ov == 1 : (AMODE == EVK_A_PLAIN || AMODE == EVK_A_CONV);
  return sb ? launch_cfg_sb<WM, WN, AMODE, BMODE, true>(p, grid, s) : launch_cfg_sb<WM, WN, AMODE, BMODE, false>(p, grid, s);
}

inline long skinny_tiles() { static long v = evk_tunable("EVK_SKINNY_TILES", 113); return v; }
inline long skinny_rows() { static long v = evk_tunable("EVK_SKINNY_ROWS", 4096); return v; }

inline long split_target() {
  static long v = 0;
  if (!v) { v = evk_tunable("EVK_SPLIT_TARGET", 256); if (v < 1) v = 256; }
  return v;
}

// 256 x 256 tile, 16 waves, one block per CU, double-buffered LDS (A_PLAIN x B_PLAIN only).  Measured with rocprofv3
// (TCP_PENDING_STALL_CYCLES ~47 % of the kernel, TCP_TCC_READ_REQ_LATENCY 500-800 cycles per request): a CU's vector L1
// sustains ~25-30 GB/s of fills at the loaded latency, ~6.4 TB/s chip-wide whether the lines come from HBM or from L2, so
// the tile kernels are bound by LDS-FILL bytes, A and B re-reads included: 128 x 128 tiles cap near 800 TFLOP/s (measured
// 848 on 8192^3).  A 256 x 256 tile halves the fill bytes per flop (1056 TFLOP/s on 8192^3) but runs ONE lock-stepped block
// per CU, which loses on problems of few tiles -> only when the big tiles alone give every CU four rounds of blocks, and not for
// the convolutions with fused statistics (neutral within noise on the trunk's 1x1 shapes; their partial-row layout stays).
inline bool use_big_tile(long M, long N, long nz) {
  static const int mode = evk_tunable("EVK_TILE256", -1);
  if (mode == 0 || M < 256 || N < 256) return false;
  if (mode == 1) return true;
  return cdiv(M, 256) * cdiv(N, 256) * nz >= 1024;
}
// split-K choice shared by the launcher and evk_gemm_workspace_bytes
inline int choose_splitk(int M, int N, int K, int batch, int splitk_req) {
  const bool narrow = N <= 64;
  const int TM = narrow ? 256 : 128, TN = narrow ? 64 : 128;
  const long tiles = cdiv(M, TM) * cdiv(N, TN) * batch;
  const int ksteps = (int)cdiv(K, BK);
  long splitk = splitk_req > 0 ? splitk_req : cdiv(split_target(), tiles);
  if (splitk > 512) splitk = 512;
  const long per_split = (long)M * N * batch * 4;
  if (splitk * per_split > SLAB_MAX_BYTES) splitk = SLAB_MAX_BYTES / per_split;
  if (splitk > ksteps) splitk = ksteps;
  if (splitk < 1) splitk = 1;
  if (splitk >= 8 && ksteps >= 8) {      // whole K-slices per XCD (see gemm_kernel): a multiple of 8 non-empty slices
    for (long s8 = (splitk + 4) / 8 * 8; s8 >= 8; s8 -= 8) {
      if (s8 * per_split > SLAB_MAX_BYTES) continue;
      const int per = (int)cdiv(ksteps, s8);
      if (cdiv(ksteps, per) == s8) return (int)s8;
    }
  }
  const int per = (int)cdiv(ksteps, splitk);
  return (int)cdiv(ksteps, per);
}

template <int AMODE, int BMODE>
int launch_modes(GemmP& p, int batch, int splitk_req, void* ws, long ws_bytes, const evk_gemm* d, hipStream_t s) {
  if constexpr (AMODE == EVK_A_KSTR) {
    if (small_eligible(p.M, p.N, AMODE, p.accumulate) && ws) {
      const int sk = small_splitk(p.M, p.N, p.K, batch);
      p.slab_mn = (long)p.M * p.N;
      if (ws_bytes >= 2L * sk * batch * p.slab_mn * 4) {
        p.slab = reinterpret_cast<float*>(ws);
        p.tilesN = (int)cdiv(p.N, 64);
        p.ksteps_per_split = (int)cdiv(cdiv(p.K, BK), sk);
        constexpr int LDS = 2 * (64 + 64) * BK * 2;
        hipLaunchKernelGGL(gemm_small_kernel<BMODE>, dim3((int)cdiv(p.M, 64) * p.tilesN, sk, batch), dim3(NTHR), LDS, s, p);
        int rc = evk_check_launch("gemm_small_kernel");
        if (rc == EVK_OK) {
          SkrP r{p.slab, reinterpret_cast<float*>(p.C), p.slab_mn, p.M, p.N, 2 * sk, p.bi, p.ldc, p.sCo, p.sCi};
          rc = launch_splitk_reduce(r, batch, s);
        }
        return rc;
      }
    }
  }
  const bool narrow = p.N <= 64;
  const bool big = AMODE == EVK_A_PLAIN && BMODE == EVK_B_PLAIN && !p.accumulate && !p.colstats && use_big_tile(p.M, p.N, batch);
  const int TM = narrow || big ? 256 : 128, TN = narrow ? 64 : (big ? 256 : 128);
  const int tilesM = (int)cdiv(p.M, TM);
  p.tilesN = (int)cdiv(p.N, TN);
  const int ksteps = (int)cdiv(p.K, BK);
  int splitk = 1;
  p.slab = nullptr;
  if (p.accumulate) {
    splitk = choose_splitk(p.M, p.N, p.K, batch, splitk_req);
    p.slab_mn = (long)p.M * p.N;
    if (splitk > 1 && ws && (p.N % 4 == 0) && ws_bytes >= (long)splitk * batch * p.slab_mn * 4) p.slab = reinterpret_cast<float*>(ws);
  }
  p.ksteps_per_split = (int)cdiv(ksteps, splitk);
  splitk = (int)cdiv(ksteps, p.ksteps_per_split);
  static const int kslice = evk_tunable("EVK_KSLICE_XCD", 1);
  p.kslice_xcd = kslice;
  // measured (tools/gemm_bench.py --cold, 4640 x 16384 x 2048): 256-wide tiles (32 resident per XCD) 745 -> 785 TFLOP/s with 4-row groups,
  // the NN data gradient on 128-wide tiles (64-96 resident) 780 -> 850 with 8; EVK_GROUP_M forces a value (0 = plain M-major order)
  static const int group_m = evk_tunable("EVK_GROUP_M", -1);
  p.group_m = group_m >= 0 ? group_m : (big ? 4 : 8);
  dim3 grid(tilesM * p.tilesN, splitk, batch);
  int rc;
  if constexpr (AMODE == EVK_A_KSTR && BMODE == EVK_B_KSTR) {
    // weight gradients of the pointwise convolutions and the linear layers: the deep-pipelined kernel of gemm_tn.hip (same tile, same
    // K-slicing and slab layout as the launch below, 72 KB of loads in flight per CU instead of 32)
    if (p.accumulate && p.c_f32 && p.alpha == 1.f && !narrow && d->b_klog <= 0 && d->b_tapstride == 0 && (p.slab || splitk == 1) &&
        evk_gemm_tn_supported(p.M, p.N, p.K, p.lda, p.ldb, p.ldc, p.sAo, p.sAi, p.sBo, p.sBi, p.sCo, p.sCi)) {
      rc = evk_gemm_tn_launch(p.A, p.B, reinterpret_cast<float*>(p.C), p.slab, p.M, p.N, p.K, p.lda, p.ldb, p.ldc, splitk, p.ksteps_per_split, batch,
                              p.bi, p.sAo, p.sAi, p.sBo, p.sBi, p.sCo, p.sCi, s);
      if (rc == EVK_OK && p.slab) {
        SkrP r{p.slab, reinterpret_cast<float*>(p.C), p.slab_mn, p.M, p.N, splitk, p.bi, p.ldc, p.sCo, p.sCi};
        rc = launch_splitk_reduce(r, batch, s);
      }
      return rc;
    }
  }
  if constexpr (AMODE == EVK_A_PLAIN && BMODE == EVK_B_PLAIN) {
    rc = narrow ? launch_cfg<4, 1, AMODE, BMODE>(p, grid, s)
                : (big ? launch_cfg_sb<4, 4, AMODE, BMODE, false>(p, grid, s) : launch_cfg<2, 2, AMODE, BMODE>(p, grid, s));
  } else {
    rc = narrow ? launch_cfg<4, 1, AMODE, BMODE>(p, grid, s) : launch_cfg<2, 2, AMODE, BMODE>(p, grid, s);
  }
  if (rc == EVK_OK && p.slab) {
    SkrP r{p.slab, reinterpret_cast<float*>(p.C), p.slab_mn, p.M, p.N, splitk, p.bi, p.ldc, p.sCo, p.sCi};
    rc = launch_splitk_reduce(r, batch, s);
  }
  return rc;
}

inline bool al(const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

int evk_splitk_reduce_launch(const float* slab, float* C, long mn, int M, int N, int splitk, int bi, long ldc, long sCo, long sCi, int batch, hipStream_t s) {
  SkrP r{slab, C, mn, M, N, splitk, bi, ldc, sCo, sCi};
  return launch_splitk_reduce(r, batch, s);
}

// y[M][N] = act(LayerNorm(x)[M][512] . W[N][512]^T + bias) (+ resid): the norm happens in the GEMM's operand load (decode step)
extern "C" int evk_linear_ln(const void* x, const float* gamma, const float* beta, const void* dgam, const void* dbet, int64_t ld_delta,
                             float eps, int32_t mode, const void* w, const float* bias, const void* resid, int64_t ldr, void* y, int32_t y_dtype,
                             int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && gamma && beta && w && y && M > 0 && N > 0, "linear_ln: null operand");
  EVK_REQUIRE(K == 512 && M <= 4096, "linear_ln: built for K = 512 (the decoder width) and M <= 4096 (got K=%d M=%d)", K, M);
  EVK_REQUIRE((dgam == nullptr) == (dbet == nullptr) && (!dgam || ld_delta % 8 == 0) && (mode == 0 || mode == 1), "linear_ln: bad deltas / mode");
  EVK_REQUIRE(ldc >= N && (!resid || ldr >= N) && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 &&
              (!dgam || ((reinterpret_cast<uintptr_t>(dgam) | reinterpret_cast<uintptr_t>(dbet)) & 15) == 0), "linear_ln: alignment / leading dimensions");
  GemmP p{};
  p.A = (const bf16_t*)x; p.B = (const bf16_t*)w; p.C = y; p.bias = bias; p.resid = resid;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = ldc; p.ldr = ldr; p.bi = 1;
  p.alpha = 1.f; p.act = act; p.c_f32 = y_dtype == EVK_F32; p.r_f32 = 0;
  p.ln_g = gamma; p.ln_b = beta; p.ln_dg = (const bf16_t*)dgam; p.ln_db = (const bf16_t*)dbet; p.ln_ld = ld_delta; p.ln_eps = eps; p.ln_mode = mode;
  evk_prof_tag(M, N, K, 1, 0, 0);
  ProfScope ps(EVK_FAM_GEMM, s, 2.0 * M * (double)N * K);
  return launch_skinny_cfg<512, 16, 64>(p, s);
}

extern "C" int64_t evk_gemm_workspace_bytes(const evk_gemm* d) {
  if (!d || !d->accumulate || d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->N % 4)) return 0;
  const int batch = d->batch_outer * d->batch_inner;
  if (small_eligible(d->M, d->N, d->a_mode, d->accumulate))
    return 2LL * small_splitk(d->M, d->N, d->K, batch) * batch * d->M * d->N * 4;
  const int sk = choose_splitk(d->M, d->N, d->K, batch, d->splitk);
  return sk > 1 ? (int64_t)sk * batch * d->M * d->N * 4 : 0;
}

extern "C" int evk_gemm_launch(const evk_gemm* d, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(d && d->A && d->B && d->C, "evk_gemm: null operand");
  EVK_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "evk_gemm: bad dims M=%d N=%d K=%d", d->M, d->N, d->K);
  EVK_REQUIRE(d->batch_outer >= 1 && d->batch_inner >= 1 && (long)d->batch_outer * d->batch_inner <= 65535,
              "evk_gemm: bad batch %d x %d", d->batch_outer, d->batch_inner);
  GemmP p{};
  p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = d->C; p.bias = d->bias; p.resid = d->resid;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr;
  p.bi = d->batch_inner;
  p.sAo = d->sAo; p.sAi = d->sAi; p.sBo = d->sBo; p.sBi = d->sBi; p.sCo = d->sCo; p.sCi = d->sCi; p.sRo = d->sRo; p.sRi = d->sRi;
  p.sbias = d->bias_stride_inner;
  p.alpha = d->alpha; p.act = d->act; p.c_f32 = d->c_dtype == EVK_F32; p.r_f32 = d->r_dtype == EVK_F32;
  p.accumulate = d->accumulate;
  p.colstats = reinterpret_cast<float*>(d->colstats);
  p.gatestats = reinterpret_cast<float*>(d->gatestats);
  {
    // measured (cold caches, tools/gemm_bench.py --cold): +20 % on the write-dominated K = 64 convolutions of layer1, neutral
    // elsewhere -> on for short-K problems; EVK_LDS_STORE=0/1 forces it off / on for every eligible launch
    static const int lds_store = evk_tunable("EVK_LDS_STORE", -1);
    p.lds_store = (lds_store < 0 ? d->K <= 128 : lds_store != 0) && !p.c_f32 && !p.accumulate && (d->N % 8 == 0) && (d->ldc % 8 == 0) && al(d->C, 16) && (d->sCo % 8 == 0) && (d->sCi % 8 == 0);
  }
  {
    static const int fast_loads = evk_tunable("EVK_FAST_LOADS", 1);
    p.fast_loads = fast_loads;
  }
  p.gate = reinterpret_cast<const bf16_t*>(d->relu_gate); p.ldg = d->ldg;
  EVK_REQUIRE(!p.gate || (!p.accumulate && d->batch_outer * d->batch_inner == 1 && d->ldg >= d->N), "evk_gemm: relu_gate needs batch 1, no accumulate, ldg >= N");
  EVK_REQUIRE(!p.colstats || (!p.accumulate && d->batch_outer * d->batch_inner == 1), "evk_gemm: colstats needs batch 1 and no accumulate");
  EVK_REQUIRE(!p.gatestats || p.gate, "evk_gemm: gatestats needs relu_gate");
  EVK_REQUIRE(!p.accumulate || (p.c_f32 && !d->bias && !d->resid && d->act == EVK_ACT_NONE),
              "evk_gemm: accumulate needs f32 C and no bias/resid/act");
  const int esz = p.c_f32 ? 4 : 2;
  p.vec_ok = (d->ldc % 4 == 0) && al(d->C, 4 * esz) && (d->sCo % 4 == 0) && (d->sCi % 4 == 0) &&
             (!d->resid || ((d->ldr % 4 == 0) && al(d->resid, p.r_f32 ? 16 : 8) && (d->sRo % 4 == 0) && (d->sRi % 4 == 0)));
  const evk_conv_geom& g = d->g;
  const bool gather = d->a_mode == EVK_A_CONV || d->a_mode == EVK_A_DGRAD || d->b_mode == EVK_B_WGATHER;
  if (gather) {
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ho = g.Ho; p.Wo = g.Wo; p.KH = g.KH; p.KW = g.KW;
    p.sh = g.stride_h; p.sw = g.stride_w; p.ph = g.pad_h; p.pw = g.pad_w;
    p.sN = g.sN; p.sH = g.sH; p.sW = g.sW;
    EVK_REQUIRE(g.KH >= 1 && g.KW >= 1 && g.stride_h >= 1 && g.stride_w >= 1, "evk_gemm: bad conv geometry");
    if (d->a_mode == EVK_A_DGRAD) {
      p.Cg = g.Co; p.rows_per_img = g.Hi * g.Wi; p.row_w = g.Wi;
      EVK_REQUIRE(g.stride_h == g.stride_w && (g.stride_h == 1 || g.stride_h == 2), "dgrad: stride must be 1 or 2");
      p.lgs = g.stride_h == 2 ? 1 : 0;
    } else {
      p.Cg = g.Ci; p.rows_per_img = g.Ho * g.Wo; p.row_w = g.Wo;
    }
    p.lgC = ilog2_exact(p.Cg);
    EVK_REQUIRE(p.lgC >= 3, "evk_gemm: gathered channel count %d must be a power of two >= 8", p.Cg);
    EVK_REQUIRE(g.sW % 2 == 0 && g.sH % 2 == 0 && g.sN % 2 == 0, "evk_gemm: gather strides must be even");
  } else {
    p.KW = 1; p.KH = 1;
  }
  // alignment rules of the loaders
  EVK_REQUIRE(al(d->A, 16) && al(d->B, 16), "evk_gemm: operands must be 16-byte aligned");
  if (d->a_mode == EVK_A_PLAIN) EVK_REQUIRE(d->K % 8 == 0 && d->lda % 8 == 0 && d->sAo % 8 == 0 && d->sAi % 8 == 0, "A_PLAIN: K, lda, batch strides must be multiples of 8 (K=%d lda=%ld)", d->K, (long)d->lda);
  if (d->a_mode == EVK_A_CONV || d->a_mode == EVK_A_DGRAD) EVK_REQUIRE(d->K % 8 == 0, "A gather: K %% 8");
  if (d->a_mode == EVK_A_CONV) EVK_REQUIRE(g.sW % 8 == 0 && g.sH % 8 == 0 && g.sN % 8 == 0, "A_CONV: strides %% 8");
  if (d->a_mode == EVK_A_KSTR) EVK_REQUIRE(d->lda % 8 == 0 && d->sAo % 8 == 0 && d->sAi % 8 == 0 && d->lda >= (d->M + 7) / 8 * 8, "A_KSTR: lda must be a multiple of 8 and >= pad8(M) (M=%d lda=%ld)", d->M, (long)d->lda);
  if (d->b_mode == EVK_B_PLAIN) EVK_REQUIRE(d->K % 8 == 0 && d->ldb % 8 == 0 && d->sBo % 8 == 0 && d->sBi % 8 == 0, "B_PLAIN: K, ldb, batch strides must be multiples of 8 (K=%d ldb=%ld)", d->K, (long)d->ldb);
  if (d->b_mode == EVK_B_KSTR) {
    EVK_REQUIRE(d->ldb % 8 == 0 && d->sBo % 8 == 0 && d->sBi % 8 == 0 && d->b_tapstride % 8 == 0 && (d->b_klog > 0 || d->ldb >= (d->N + 7) / 8 * 8), "B_KSTR: ldb must be a multiple of 8 and >= pad8(N) (N=%d ldb=%ld)", d->N, (long)d->ldb);
    p.b_klog = d->b_klog > 0 ? d->b_klog : 30;
    p.b_kmask = d->b_klog > 0 ? ((1 << d->b_klog) - 1) : 0x7fffffff;
    p.b_tapstride = d->b_tapstride;
  }
  if (d->b_mode != EVK_B_KSTR) { p.b_klog = 30; p.b_kmask = 0x7fffffff; p.b_tapstride = 0; }
  if (d->b_mode == EVK_B_WGATHER) EVK_REQUIRE(d->N % 8 == 0 && g.sW % 8 == 0 && g.sH % 8 == 0 && g.sN % 8 == 0 && (long)d->K < (1L << 24), "B_WGATHER: N %% 8, strides %% 8, K < 2^24");

  const int batch = d->batch_outer * d->batch_inner;
  const double flops = 2.0 * d->M * (double)d->N * d->K * batch;
  evk_prof_tag(d->M, d->N, d->K, batch, d->a_mode, d->b_mode);
  ProfScope ps(EVK_FAM_GEMM, s, flops);
  const int am = d->a_mode, bm = d->b_mode;
  if (am == EVK_A_PLAIN && bm == EVK_B_PLAIN && batch == 1 && !p.accumulate && d->K % SK_KC == 0 &&
      cdiv(d->M, 128) * cdiv(d->N, 128) < skinny_tiles() && d->M <= skinny_rows() && !d->colstats && !(d->relu_gate && d->gatestats))
    return launch_skinny(p, s);
  if (am == EVK_A_PLAIN && bm == EVK_B_PLAIN) return launch_modes<EVK_A_PLAIN, EVK_B_PLAIN>(p, batch, d->splitk, d->workspace, d->workspace_bytes, d, s);
  if (am == EVK_A_CONV && bm == EVK_B_PLAIN) return launch_modes<EVK_A_CONV, EVK_B_PLAIN>(p, batch, d->splitk, d->workspace, d->workspace_bytes, d, s);
  if (am == EVK_A_PLAIN && bm == EVK_B_KSTR) return launch_modes<EVK_A_PLAIN, EVK_B_KSTR>(p, batch, d->splitk, d->workspace, d->workspace_bytes, d, s);
  if (am == EVK_A_DGRAD && bm == EVK_B_KSTR) return launch_modes<EVK_A_DGRAD, EVK_B_KSTR>(p, batch, d->splitk, d->workspace, d->workspace_bytes, d, s);
  if (am == EVK_A_KSTR && bm == EVK_B_KSTR) return launch_modes<EVK_A_KSTR, EVK_B_KSTR>(p, batch, d->splitk, d->workspace, d->workspace_bytes, d, s);
  if (am == EVK_A_KSTR && bm == EVK_B_WGATHER) return launch_modes<EVK_A_KSTR, EVK_B_WGATHER>(p, batch, d->splitk, d->workspace, d->workspace_bytes, d, s);
  evk_set_error("evk_gemm: unsupported mode pair a=%d b=%d", am, bm);
  return EVK_EUNSUPPORTED;
}
