// rm.hip -- the relational memory of the R2Gen decoder (modules/encoder_decoder.py:246-300) as a native runner.
//
// RelationalMemory.forward is a serial recurrence over the L report tokens; the reference walks it in a Python loop
// of ~25 tiny torch kernels per token (and the naive engine did the same with ~15 autograd nodes per token, i.e.
// ~4.5k launches + Python overhead per training step).  Here the host loop lives in C++ and every token is
//     fwd: 5 GEMM launches (m.[Wq;Wk;Wv], a.Wo + m, relu(.W0), relu(.W2), tanh(m).U) + 1 fused attention kernel
//          (3 slots x 4 keys x 8 heads per sample, softmax + dropout + PV) + 1 fused gate kernel
//     bwd: 5 data-gradient GEMMs + one gate kernel (which also forms the carried gradient of the token before and the first ReLU mask) + the attention kernel,
// while everything that does not depend on the recurrence is hoisted out of the loop by the caller (x_t.Wk, x_t.Wv and
// W(x_t) for all t are three ordinary batched GEMMs) and all weight gradients are ONE K = L*B*slots GEMM per weight
// after the loop (the per-token operands are saved contiguously in the workspace).
#include <stdlib.h>
#include "common.h"

// csrc/rm_f32.hip
int rmf32_gemm(const float* A, long lda, const float* W, const float* bias, const float* resid, long ldr, float* C, long ldc, int M, int N, int act,
               int a_tanh, bf16_t* C16, long ldc16, hipStream_t s);
int rmf32_attn_train(const float* qkv, const float* xp, long x_bstride, float* a, bf16_t* a16, float* P, float p_drop, unsigned long long seed, int B,
                     hipStream_t s);
int rmf32_gate_train(const float* xp, long x_bstride, const float* gu, const float* nm1, const float* h2, float* m, bf16_t* m16_next, bf16_t* tm16_next,
                     bf16_t* out, long out_bstride, bf16_t* si, bf16_t* sf, bf16_t* tnm, int B, hipStream_t s);
int rmf32_qkv_gu(const float* mem, const float* tmem, const float* Wqkv, const float* bqkv, float* qkv, bf16_t* qkv16, const float* U, const float* bU,
                 float* gu, int R, hipStream_t s);
int rmf32_w2_gate_train(const float* h1, const float* W2, const float* b2, bf16_t* h2_16, const float* xp, long x_bstride, const float* gu, const float* nm1,
                        float* m, bf16_t* m16_next, bf16_t* tm16_next, bf16_t* out, long out_bstride, bf16_t* si, bf16_t* sf, bf16_t* tnm, float* tm32,
                        int R, hipStream_t s);

namespace {

__device__ __forceinline__ uint32_t hash32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}
__device__ __forceinline__ float keep_scale(uint64_t seed, uint64_t idx, float p) {
  if (p <= 0.f) return 1.f;
  return (hash32(seed * 0x9E3779B97F4A7C15ULL + idx) >> 8) * (1.f / 16777216.f) >= p ? 1.f / (1.f - p) : 0.f;
}
__device__ __forceinline__ float half_sum(float v) {   // sum over the 32 lanes of a half-wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int S_ = 3, KEYS = 4, HEADS = 8, DH = 64, D_ = 512;   // rm_num_slots, slots+1, rm_num_heads, d/heads, rm_d_model

// one block per sample; thread -> head h = tid>>5, dims 2l, 2l+1 (l = tid&31)
struct AttP {
  const bf16_t* qkv; const bf16_t* xk; const bf16_t* xv; long x_bstride;  // qkv [B*3][1536]; xk/xv row of sample b at b*x_bstride
  float* P; bf16_t* a; float p_drop; unsigned long long seed;
  // backward
  const bf16_t* da; bf16_t* dqkv; bf16_t* dxk; bf16_t* dxv;
  const unsigned long long* epoch;
};

__device__ __forceinline__ void load_qkv(const AttP& p, int b, int h, int l, float (&q)[S_][2], float (&k)[KEYS][2], float (&v)[KEYS][2]) {
  const int c = h * DH + 2 * l;
#pragma unroll
  for (int i = 0; i < S_; ++i) {
    const bf16_t* r = p.qkv + (long)(b * S_ + i) * 1536 + c;
    const uint32_t a = *reinterpret_cast<const uint32_t*>(r), kk = *reinterpret_cast<const uint32_t*>(r + 512), vv = *reinterpret_cast<const uint32_t*>(r + 1024);
    q[i][0] = lo_bf(a); q[i][1] = hi_bf(a); k[i][0] = lo_bf(kk); k[i][1] = hi_bf(kk); v[i][0] = lo_bf(vv); v[i][1] = hi_bf(vv);
  }
  const uint32_t kk = *reinterpret_cast<const uint32_t*>(p.xk + (long)b * p.x_bstride + c);
  const uint32_t vv = *reinterpret_cast<const uint32_t*>(p.xv + (long)b * p.x_bstride + c);
  k[3][0] = lo_bf(kk); k[3][1] = hi_bf(kk); v[3][0] = lo_bf(vv); v[3][1] = hi_bf(vv);
}

__global__ __launch_bounds__(256) void rm_attn_fwd_kernel(const AttP p) {
  const int b = blockIdx.x, h = threadIdx.x >> 5, l = threadIdx.x & 31;
  float q[S_][2], k[KEYS][2], v[KEYS][2];
  load_qkv(p, b, h, l, q, k, v);
  float pr[S_][KEYS];
#pragma unroll
  for (int i = 0; i < S_; ++i) {
    float s[KEYS], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS; ++j) { s[j] = half_sum(q[i][0] * k[j][0] + q[i][1] * k[j][1]) * 0.125f; mx = fmaxf(mx, s[j]); }
#pragma unroll
    for (int j = 0; j < KEYS; ++j) { s[j] = __expf(s[j] - mx); sum += s[j]; }
#pragma unroll
    for (int j = 0; j < KEYS; ++j) pr[i][j] = s[j] / sum;
  }
  if (l == 0 && p.P) {
#pragma unroll
    for (int i = 0; i < S_; ++i)
#pragma unroll
      for (int j = 0; j < KEYS; ++j) p.P[((long)(b * HEADS + h) * S_ + i) * KEYS + j] = pr[i][j];
  }
#pragma unroll
  for (int i = 0; i < S_; ++i) {
    float o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS; ++j) {
      const float w = pr[i][j] * keep_scale(evk_mix_seed(p.seed, p.epoch), ((uint64_t)(b * HEADS + h) * S_ + i) * KEYS + j, p.p_drop);
      o0 += w * v[j][0]; o1 += w * v[j][1];
    }
    *reinterpret_cast<uint32_t*>(p.a + (long)(b * S_ + i) * D_ + h * DH + 2 * l) = pack2bf(o0, o1);
  }
}

__global__ __launch_bounds__(256) void rm_attn_bwd_kernel(const AttP p) {
  const int b = blockIdx.x, h = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int c = h * DH + 2 * l;
  float q[S_][2], k[KEYS][2], v[KEYS][2];
  load_qkv(p, b, h, l, q, k, v);
  float da[S_][2];
#pragma unroll
  for (int i = 0; i < S_; ++i) { const uint32_t t = *reinterpret_cast<const uint32_t*>(p.da + (long)(b * S_ + i) * D_ + c); da[i][0] = lo_bf(t); da[i][1] = hi_bf(t); }
  float dq[S_][2] = {}, dk[KEYS][2] = {}, dv[KEYS][2] = {};
#pragma unroll
  for (int i = 0; i < S_; ++i) {
    float pr[KEYS], ks[KEYS], dp[KEYS], dot = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS; ++j) {
      pr[j] = p.P[((long)(b * HEADS + h) * S_ + i) * KEYS + j];
      ks[j] = keep_scale(evk_mix_seed(p.seed, p.epoch), ((uint64_t)(b * HEADS + h) * S_ + i) * KEYS + j, p.p_drop);
      dp[j] = half_sum(da[i][0] * v[j][0] + da[i][1] * v[j][1]) * ks[j];      // d/dP (through the dropout scale)
      dv[j][0] += pr[j] * ks[j] * da[i][0]; dv[j][1] += pr[j] * ks[j] * da[i][1];
      dot += dp[j] * pr[j];
    }
#pragma unroll
    for (int j = 0; j < KEYS; ++j) {
      const float ds = pr[j] * (dp[j] - dot) * 0.125f;
      dq[i][0] += ds * k[j][0]; dq[i][1] += ds * k[j][1];
      dk[j][0] += ds * q[i][0]; dk[j][1] += ds * q[i][1];
    }
  }
#pragma unroll
  for (int i = 0; i < S_; ++i) {
    bf16_t* r = p.dqkv + (long)(b * S_ + i) * 1536 + c;
    *reinterpret_cast<uint32_t*>(r) = pack2bf(dq[i][0], dq[i][1]);
    *reinterpret_cast<uint32_t*>(r + 512) = pack2bf(dk[i][0], dk[i][1]);
    *reinterpret_cast<uint32_t*>(r + 1024) = pack2bf(dv[i][0], dv[i][1]);
  }
  *reinterpret_cast<uint32_t*>(p.dxk + (long)b * p.x_bstride + c) = pack2bf(dk[3][0], dk[3][1]);
  *reinterpret_cast<uint32_t*>(p.dxv + (long)b * p.x_bstride + c) = pack2bf(dv[3][0], dv[3][1]);
}

// gates = gw[b][2d] (broadcast over slots) + gu[b][s][2d];  nm2 = nm1 + h2;  next = sig(ig)*tanh(nm2) + sig(fg)*m
struct GateP {
  const bf16_t* gw; long gw_bstride; const bf16_t* gu; const bf16_t* nm1; const bf16_t* h2; const bf16_t* m;
  bf16_t* m_next; bf16_t* tm_next; bf16_t* out; long out_bstride; bf16_t* si; bf16_t* sf; bf16_t* tnm; int B;
};
__global__ __launch_bounds__(256) void rm_gate_fwd2_kernel(const GateP p) {
  const long total = (long)p.B * S_ * D_;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % D_);
    const long bs = i / D_;
    const long b = bs / S_;
    const int s = (int)(bs - b * S_);
    const float ig = bf2f(p.gw[b * p.gw_bstride + c]) + bf2f(p.gu[bs * 2 * D_ + c]);
    const float fg = bf2f(p.gw[b * p.gw_bstride + D_ + c]) + bf2f(p.gu[bs * 2 * D_ + D_ + c]);
    const float si = 1.f / (1.f + __expf(-ig)), sf = 1.f / (1.f + __expf(-fg));
    const float t = tanhf(bf2f(p.nm1[i]) + bf2f(p.h2[i]));
    const bf16_t nx = f2bf(si * t + sf * bf2f(p.m[i]));
    p.m_next[i] = nx;
    p.tm_next[i] = f2bf(tanhf(bf2f(nx)));
    p.out[b * p.out_bstride + s * D_ + c] = nx;
    if (p.si) { p.si[i] = f2bf(si); p.sf[i] = f2bf(sf); p.tnm[i] = f2bf(t); }
  }
}

// dnext = dout[:, t] (+ dm_carry); dnm2 = dnext*si*(1-t^2); dmd = dnext*sf; dgates = {dnext*t*si*(1-si), dnext*m*sf*(1-sf)};
// dgw[b][:] = sum_s dgates[b][s][:]
struct GateBP {
  const bf16_t* dout; long dout_bstride; const bf16_t* dcarry; const bf16_t* si; const bf16_t* sf; const bf16_t* tnm; const bf16_t* m;
  bf16_t* dnm2; bf16_t* dmd; bf16_t* dgates; bf16_t* dgw; long dgw_bstride; int B;
  const bf16_t* h2; bf16_t* dh2;          // dh2 = dnm2 * (h2 > 0): the first ReLU mask of the token rides on this launch (or null / null)
  // the carried memory gradient of the token processed BEFORE this one (t + 1), formed here instead of by a launch of its own (rm_combine_kernel):
  // carry = round16(dmp + dmd + dtm * (1 - tm^2)) with that token's dmp / dmd / dtm / tm; null: dcarry is read (or there is none)
  const bf16_t* c_dmp; const bf16_t* c_dmd; const bf16_t* c_dtm; const bf16_t* c_tm;
};
__global__ __launch_bounds__(256) void rm_gate_bwd2_kernel(const GateBP p) {
  const long total = (long)p.B * D_;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % D_);
    const long b = i / D_;
    float gi = 0.f, gf = 0.f;
#pragma unroll
    for (int s = 0; s < S_; ++s) {
      const long e = (b * S_ + s) * D_ + c;
      float g = bf2f(p.dout[b * p.dout_bstride + s * D_ + c]);
      if (p.c_dmp) {          // (c_dmd may be the buffer this thread rewrites below: read first)
        const float t1 = bf2f(p.c_tm[e]);
        g += bf2f(f2bf(bf2f(p.c_dmp[e]) + bf2f(p.c_dmd[e]) + bf2f(p.c_dtm[e]) * (1.f - t1 * t1)));
      } else if (p.dcarry) g += bf2f(p.dcarry[e]);
      const float si = bf2f(p.si[e]), sf = bf2f(p.sf[e]), t = bf2f(p.tnm[e]);
      const bf16_t dn = f2bf(g * si * (1.f - t * t));
      p.dnm2[e] = dn;
      if (p.dh2) p.dh2[e] = bf2f(p.h2[e]) > 0.f ? dn : (bf16_t)0;
      p.dmd[e] = f2bf(g * sf);
      const float di = g * t * si * (1.f - si), df = g * bf2f(p.m[e]) * sf * (1.f - sf);
      p.dgates[(b * S_ + s) * 2 * D_ + c] = f2bf(di);
      p.dgates[(b * S_ + s) * 2 * D_ + D_ + c] = f2bf(df);
      gi += di; gf += df;
    }
    p.dgw[b * p.dgw_bstride + c] = f2bf(gi);
    p.dgw[b * p.dgw_bstride + D_ + c] = f2bf(gf);
  }
}

// y = dy * (ref > 0)
__global__ __launch_bounds__(256) void relu_mask_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ ref, bf16_t* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = bf2f(ref[i]) > 0.f ? dy[i] : (bf16_t)0;
}
// Measured on MI355X (FineTune 384^2, 32 samples, L = 100; profiles/r02_c_rm_persistent_kernel_stats.csv): persistent forward
// 15.1 ms + backward 12.5 ms against 5.5 + 5.8 ms of kernel time for the per-token launch sequence -- every workgroup has to
// pull the 4 MB of weights through ITS OWN vector L1 once per token, and one CU's L1 miss queue sustains ~27 GB/s (the same
// fill bound the tile GEMMs sit on): 4 MB / 27 GB/s = 150 us per token.  The launch sequence spreads each
// GEMM's weights over 32 CUs instead.  So the persistent kernels are OPT-IN (EVK_RM_PERSIST=1 / evk_rm_set_persistent(1)): they
// cut 2000 launches (~10 ms of host time per step) to 2 and are the right trade when the host is the bottleneck.
int g_rm_persist = -1;           // -1: EVK_RM_PERSIST (default off); 0 / 1: evk_rm_set_persistent
inline bool rm_use_persistent(int L) {
  if (g_rm_persist < 0) g_rm_persist = evk_tunable("EVK_RM_PERSIST", 0) != 0;
  return g_rm_persist != 0 && L >= 8;
}

inline int ew_blocks(long work) { long b = cdiv(work, 256); return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b)); }

int gemm(const void* A, const void* B, void* C, int M, int N, int K, int b_mode, long ldb, const float* bias, const void* resid, int act,
         evk_stream_t st, const void* relu_gate = nullptr) {
  evk_gemm d{};
  d.A = A; d.B = B; d.C = C; d.bias = bias; d.resid = resid;
  d.relu_gate = relu_gate; d.ldg = N;
  d.M = M; d.N = N; d.K = K; d.a_mode = EVK_A_PLAIN; d.b_mode = b_mode;
  d.lda = K; d.ldb = ldb; d.ldc = N; d.ldr = N;
  d.batch_outer = d.batch_inner = 1; d.alpha = 1.f; d.act = act; d.c_dtype = EVK_BF16; d.r_dtype = EVK_BF16;
  return evk_gemm_launch(&d, st);
}

// C[N][K] (f32) += dY[rows][N]^T . X[rows][K]
int wgrad(const void* dY, const void* X, float* dW, long rows, int N, int K, void* ws, long ws_bytes, evk_stream_t st) {
  evk_gemm d{};
  d.A = dY; d.B = X; d.C = dW;
  d.M = N; d.N = K; d.K = (int)rows; d.a_mode = EVK_A_KSTR; d.b_mode = EVK_B_KSTR;
  d.lda = N; d.ldb = K; d.ldc = K; d.batch_outer = d.batch_inner = 1; d.alpha = 1.f; d.c_dtype = EVK_F32; d.accumulate = 1;
  d.workspace = ws; d.workspace_bytes = ws_bytes;
  return evk_gemm_launch(&d, st);
}

// workspace layout (bf16 elements unless noted); R = B*S rows
struct Ws {
  long R, per;      // per-token stride of the [L][R][*] arrays
  bf16_t *m, *tm, *qkv, *a, *nm1, *h1, *h2, *si, *sf, *tnm, *gu, *dqkv, *dnm2s, *dh2s, *dh1s, *dgs;
  float* P;
  bf16_t *t_dmd, *t_dtm, *t_t1, *t_dnm1, *t_da, *t_dmp, *t_carry0, *t_carry1;
  char* slab; long slab_bytes;
};

}  // namespace

extern "C" {

/* 1: walks of >= 8 tokens run as ONE persistent launch per direction; 0 (default): always the per-token launch path */
int evk_rm_set_persistent(int32_t on) { g_rm_persist = on ? 1 : 0; return EVK_OK; }

/* bytes of workspace for evk_rm_forward(save=1) + evk_rm_backward */
int64_t evk_rm_ws_bytes(int32_t B, int32_t L) {
  const long R = (long)B * S_;
  long e = 0;
  e += (long)(L + 1) * R * D_ * 2;          // m, tm  ([L+1])
  e += (long)L * R * 1536 * 2;              // qkv, dqkv
  e += (long)L * R * D_ * 10;               // a, nm1, h1, h2, si, sf, tnm, dnm2s(=dnm1 stack), dh2s, dh1s
  e += (long)L * R * 2 * D_ * 2;            // gu (unused after fwd but kept simple), dgs
  long bytes = e * 2 + (long)L * B * HEADS * S_ * KEYS * 4;
  bytes += 8L * R * 1536 * 2;               // per-token temporaries
  bytes += 96L << 20;                       // split-K slabs of the weight-gradient GEMMs
  return bytes + 65536;
}

}  // extern "C"

namespace {

Ws carve(void* ws, int B, int L) {
  Ws w{};
  w.R = (long)B * S_;
  char* p = reinterpret_cast<char*>(ws);
  auto take = [&](long elems) { bf16_t* r = reinterpret_cast<bf16_t*>(p); p += ((elems * 2 + 255) / 256) * 256; return r; };
  const long R = w.R;
  w.m = take((long)(L + 1) * R * D_); w.tm = take((long)(L + 1) * R * D_);
  w.qkv = take((long)L * R * 1536); w.dqkv = take((long)L * R * 1536);
  w.a = take((long)L * R * D_); w.nm1 = take((long)L * R * D_); w.h1 = take((long)L * R * D_); w.h2 = take((long)L * R * D_);
  w.si = take((long)L * R * D_); w.sf = take((long)L * R * D_); w.tnm = take((long)L * R * D_);
  w.dnm2s = take((long)L * R * D_); w.dh2s = take((long)L * R * D_); w.dh1s = take((long)L * R * D_);
  w.gu = take((long)L * R * 2 * D_); w.dgs = take((long)L * R * 2 * D_);
  w.P = reinterpret_cast<float*>(take((long)L * B * HEADS * S_ * KEYS * 2));
  w.t_dmd = take(R * D_); w.t_dtm = take(R * D_); w.t_t1 = take(R * D_); w.t_dnm1 = take(R * D_); w.t_da = take(R * D_);
  w.t_dmp = take(R * D_); w.t_carry0 = take(R * D_); w.t_carry1 = take(R * D_);
  w.slab = p; w.slab_bytes = 96L << 20;
  return w;
}


// ====================================================================================================
// Persistent recurrence kernels: ONE launch walks all L tokens (forward) / all L tokens backwards (BPTT).
//
// One workgroup per SAMPLE (256 threads = 4 waves); the 3 x 512 memory of the sample lives in LDS for the whole walk, every
// intermediate of a token goes LDS -> LDS (and, 16-bit rounded at the same points as the per-token launch path above, to the
// workspace the backward / the weight-gradient GEMMs read).  No workgroup ever waits for another one (no grid barrier, no
// residency requirement), so the kernel cannot dead-lock whatever else shares the GPU.
// The seven weight matrices (4 MB of 16-bit operands) do not fit one CU's LDS + registers: each workgroup STREAMS them from
// L2 / Infinity Cache once per token straight into MFMA A-operand registers (row-major [out][in] rows are K-contiguous = the
// A fragment of v_mfma_f32_16x16x32: lane -> weight row l&15, 8 consecutive k at 8*(l>>4); 16-byte loads, two row blocks in
// flight per wave), the 3 activation rows are the B operand (columns 3..15 of the tile are don't-care).  All workgroups walk
// the same weights in the same order at the same pace, so a line fetched by one is an L2 hit for the other samples of its XCD.
// Bound: weight bytes per token per CU (4 MB at the CU's fill rate), not MFMA (4096 MFMAs per token per CU ~ 7 us).
// ====================================================================================================
constexpr int LS = 520;          // LDS row stride (16-bit elements) of a 512-wide activation row: rows land 16 B apart mod 256 B

struct PersistW {                // forward weights (row-major [out][in]) / backward: the TRANSPOSED weights
  const bf16_t *Wqkv, *Wo, *W0, *W2, *U;
  const float *bqkv, *bo, *b0, *b2, *bU;
};
struct PersistP {
  PersistW w;
  const bf16_t *xk, *xv, *gw, *m0;      // (B, L, 512) x2, (B, L, 1024), (B, 3, 512)
  bf16_t *out, *m_last;                 // (B, L, 1536), (B, 3, 512) or null
  Ws ws; int B, L; float p_drop; unsigned long long seed; const unsigned long long* epoch;
  // backward only
  const bf16_t* dout; bf16_t *dxk, *dxv, *dgw;
};

__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

// out rows [16*rb, 16*rb+16) for the wave's row blocks rb = wave, wave+4, ... < nblocks of W[N][512*KCH] times the 3
// activation rows in `in` (LDS, row stride `lsi`).  Work unit = (row block, 512-wide K chunk); the A fragments of unit u+1 are
// in flight while unit u multiplies (two register sets, ping-pong).  KCH == 1: the B fragments are loaded once; otherwise they
// are re-read from LDS per unit (the accumulators of all row blocks of a wave would not fit beside two A sets).
// epi(rb, acc): lane holds D[row 4*(l>>4)+j][col l&15], useful for col < 3.
template <int KCH, class Epi>
__device__ __forceinline__ void mvgen(const bf16_t* __restrict__ W, int nblocks, const bf16_t* in, int lsi, int wave, int lane, Epi epi) {
  constexpr int ktot = 512 * KCH;
  const int col = lane & 15, kq = lane >> 4;
  const bf16_t* brow = in + (col % 3) * lsi + kq * 8;
  const bf16_t* wl = W + (long)col * ktot + kq * 8;
  const int nbw = nblocks > wave ? (nblocks - wave + 3) / 4 : 0;
  const int units = nbw * KCH;
  if (units == 0) return;
  bf16x8 b[16], a0[16], a1[16];
  if (KCH == 1) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) b[ks] = ld_frag(brow + ks * 32);
  }
  auto load = [&](bf16x8 (&a)[16], int u) {
    const int rb = wave + 4 * (u / KCH), c = u % KCH;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) a[ks] = ld_frag(wl + (long)rb * 16 * ktot + c * 512 + ks * 32);
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  auto mul = [&](bf16x8 (&a)[16], int u) {
    const int rb = wave + 4 * (u / KCH), c = u % KCH;
    if (KCH > 1) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) b[ks] = ld_frag(brow + c * 512 + ks * 32);
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc = EVK_MFMA_16x16x32(a[ks], b[ks], acc, 0, 0, 0);
    if (c == KCH - 1) { epi(rb, acc); acc = f32x4{0.f, 0.f, 0.f, 0.f}; }
  };
  load(a0, 0);
  for (int u = 0; u < units; u += 2) {
    if (u + 1 < units) load(a1, u + 1);
    mul(a0, u);
    if (u + 1 >= units) break;
    if (u + 2 < units) load(a0, u + 2);
    mul(a1, u + 1);
  }
}

__device__ __forceinline__ void st4(bf16_t* p, float v0, float v1, float v2, float v3) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
}

__global__ __launch_bounds__(256) void rm_persist_fwd_kernel(const PersistP p) {
  __shared__ __attribute__((aligned(16))) bf16_t sm[3 * LS], stm[3 * LS], sqkv[3 * (1536 + 8)], sgu[3 * (1024 + 8)], sa[3 * LS], snm1[3 * LS], sh1[3 * LS],
      sh2[3 * LS];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, kq = lane >> 4;
  const Ws& w = p.ws;
  const long R = w.R, RD = R * D_;
  const long row0 = (long)b * S_;                       // first of the sample's 3 rows in the [R][*] workspace arrays
  // m_0, tanh(m_0)
  for (int i = tid; i < S_ * D_; i += 256) {
    const int s = i / D_, c = i - s * D_;
    const bf16_t v = p.m0[(row0 + s) * D_ + c];
    const bf16_t tv = f2bf(tanhf(bf2f(v)));
    sm[s * LS + c] = v; stm[s * LS + c] = tv;
    w.m[(row0 + s) * D_ + c] = v; w.tm[(row0 + s) * D_ + c] = tv;
  }
  __syncthreads();
  for (int t = 0; t < p.L; ++t) {
    // ---- qkv = m.Wqkv^T + bqkv (96 row blocks), gu = tanh(m).U^T + bU (64 row blocks)
    bf16_t* qkv_g = w.qkv + ((long)t * R + row0) * 1536;
    mvgen<1>(p.w.Wqkv, 96, sm, LS, wave, lane, [&](int rb, const f32x4& acc) {
      if (col < 3) {
        const int o = rb * 16 + kq * 4;
        const float4 bb = *reinterpret_cast<const float4*>(p.w.bqkv + o);
        st4(sqkv + col * 1544 + o, acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w);
        st4(qkv_g + (long)col * 1536 + o, acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w);
      }
    });
    mvgen<1>(p.w.U, 64, stm, LS, wave, lane, [&](int rb, const f32x4& acc) {
      if (col < 3) {
        const int o = rb * 16 + kq * 4;
        const float4 bb = *reinterpret_cast<const float4*>(p.w.bU + o);
        st4(sgu + col * 1032 + o, acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w);
      }
    });
    __syncthreads();
    // ---- attention of the 3 slots over [3 slots; x_t]: thread -> head h = tid>>5, dims 2l, 2l+1
    {
      const int h = tid >> 5, l = tid & 31, c = h * DH + 2 * l;
      float q[S_][2], k[KEYS][2], v[KEYS][2];
#pragma unroll
      for (int i = 0; i < S_; ++i) {
        const uint32_t a = *reinterpret_cast<const uint32_t*>(sqkv + i * 1544 + c), kk = *reinterpret_cast<const uint32_t*>(sqkv + i * 1544 + 512 + c),
                       vv = *reinterpret_cast<const uint32_t*>(sqkv + i * 1544 + 1024 + c);
        q[i][0] = lo_bf(a); q[i][1] = hi_bf(a); k[i][0] = lo_bf(kk); k[i][1] = hi_bf(kk); v[i][0] = lo_bf(vv); v[i][1] = hi_bf(vv);
      }
      const uint32_t kk = *reinterpret_cast<const uint32_t*>(p.xk + ((long)b * p.L + t) * D_ + c);
      const uint32_t vv = *reinterpret_cast<const uint32_t*>(p.xv + ((long)b * p.L + t) * D_ + c);
      k[3][0] = lo_bf(kk); k[3][1] = hi_bf(kk); v[3][0] = lo_bf(vv); v[3][1] = hi_bf(vv);
      const unsigned long long seed = evk_mix_seed(p.seed + 0x51ED27ULL * (uint64_t)(t + 1), p.epoch);
      float* Pt = w.P + (long)t * p.B * HEADS * S_ * KEYS;
#pragma unroll
      for (int i = 0; i < S_; ++i) {
        float sc[KEYS], mx = -INFINITY, sum = 0.f;
#pragma unroll
        for (int j = 0; j < KEYS; ++j) { sc[j] = half_sum(q[i][0] * k[j][0] + q[i][1] * k[j][1]) * 0.125f; mx = fmaxf(mx, sc[j]); }
#pragma unroll
        for (int j = 0; j < KEYS; ++j) { sc[j] = __expf(sc[j] - mx); sum += sc[j]; }
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int j = 0; j < KEYS; ++j) {
          const float pr = sc[j] / sum;
          if (l == 0) Pt[((long)(b * HEADS + h) * S_ + i) * KEYS + j] = pr;
          const float wgt = pr * keep_scale(seed, ((uint64_t)(b * HEADS + h) * S_ + i) * KEYS + j, p.p_drop);
          o0 += wgt * v[j][0]; o1 += wgt * v[j][1];
        }
        const uint32_t pk = pack2bf(o0, o1);
        *reinterpret_cast<uint32_t*>(sa + i * LS + c) = pk;
        *reinterpret_cast<uint32_t*>(w.a + (long)t * RD + (row0 + i) * D_ + c) = pk;
      }
    }
    __syncthreads();
    // ---- nm1 = a.Wo^T + bo + m
    mvgen<1>(p.w.Wo, 32, sa, LS, wave, lane, [&](int rb, const f32x4& acc) {
      if (col < 3) {
        const int o = rb * 16 + kq * 4;
        const float4 bb = *reinterpret_cast<const float4*>(p.w.bo + o);
        const uint2 mm = *reinterpret_cast<const uint2*>(sm + col * LS + o);
        const float r0 = acc[0] + bb.x + lo_bf(mm.x), r1 = acc[1] + bb.y + hi_bf(mm.x), r2 = acc[2] + bb.z + lo_bf(mm.y), r3 = acc[3] + bb.w + hi_bf(mm.y);
        st4(snm1 + col * LS + o, r0, r1, r2, r3);
        st4(w.nm1 + (long)t * RD + (row0 + col) * D_ + o, r0, r1, r2, r3);
      }
    });
    __syncthreads();
    // ---- h1 = relu(nm1.W0^T + b0) ; h2 = relu(h1.W2^T + b2)
    mvgen<1>(p.w.W0, 32, snm1, LS, wave, lane, [&](int rb, const f32x4& acc) {
      if (col < 3) {
        const int o = rb * 16 + kq * 4;
        const float4 bb = *reinterpret_cast<const float4*>(p.w.b0 + o);
        const float r0 = fmaxf(acc[0] + bb.x, 0.f), r1 = fmaxf(acc[1] + bb.y, 0.f), r2 = fmaxf(acc[2] + bb.z, 0.f), r3 = fmaxf(acc[3] + bb.w, 0.f);
        st4(sh1 + col * LS + o, r0, r1, r2, r3);
        st4(w.h1 + (long)t * RD + (row0 + col) * D_ + o, r0, r1, r2, r3);
      }
    });
    __syncthreads();
    mvgen<1>(p.w.W2, 32, sh1, LS, wave, lane, [&](int rb, const f32x4& acc) {
      if (col < 3) {
        const int o = rb * 16 + kq * 4;
        const float4 bb = *reinterpret_cast<const float4*>(p.w.b2 + o);
        const float r0 = fmaxf(acc[0] + bb.x, 0.f), r1 = fmaxf(acc[1] + bb.y, 0.f), r2 = fmaxf(acc[2] + bb.z, 0.f), r3 = fmaxf(acc[3] + bb.w, 0.f);
        st4(sh2 + col * LS + o, r0, r1, r2, r3);
        st4(w.h2 + (long)t * RD + (row0 + col) * D_ + o, r0, r1, r2, r3);
      }
    });
    __syncthreads();
    // ---- gates
    const bf16_t* gwt = p.gw + ((long)b * p.L + t) * 2 * D_;
    for (int i = tid; i < S_ * D_; i += 256) {
      const int s = i / D_, c = i - s * D_;
      const float ig = bf2f(gwt[c]) + bf2f(sgu[s * 1032 + c]);
      const float fg = bf2f(gwt[D_ + c]) + bf2f(sgu[s * 1032 + D_ + c]);
      const float si = 1.f / (1.f + __expf(-ig)), sf = 1.f / (1.f + __expf(-fg));
      const float tn = tanhf(bf2f(snm1[s * LS + c]) + bf2f(sh2[s * LS + c]));
      const bf16_t nx = f2bf(si * tn + sf * bf2f(sm[s * LS + c]));
      const bf16_t tnx = f2bf(tanhf(bf2f(nx)));
      const long e = (row0 + s) * D_ + c;
      w.si[(long)t * RD + e] = f2bf(si); w.sf[(long)t * RD + e] = f2bf(sf); w.tnm[(long)t * RD + e] = f2bf(tn);
      w.m[(long)(t + 1) * RD + e] = nx; w.tm[(long)(t + 1) * RD + e] = tnx;
      p.out[((long)b * p.L + t) * S_ * D_ + s * D_ + c] = nx;
      sm[s * LS + c] = nx; stm[s * LS + c] = tnx;
    }
    __syncthreads();
  }
  if (p.m_last)
    for (int i = tid; i < S_ * D_; i += 256) { const int s = i / D_, c = i - s * D_; p.m_last[(row0 + s) * D_ + c] = sm[s * LS + c]; }
}

// BPTT: p.w holds the TRANSPOSED weights (Wqkv -> [512][1536], U -> [512][1024], Wo / W0 / W2 -> [512][512]).
__global__ __launch_bounds__(256) void rm_persist_bwd_kernel(const PersistP p) {
  __shared__ __attribute__((aligned(16))) bf16_t sdg[3 * (1024 + 8)], sdqkv[3 * (1536 + 8)], sA[3 * LS], sB[3 * LS], sdnm2[3 * LS], sdnm1[3 * LS];
  __shared__ float scarry[3 * D_], sdmd[3 * D_], sdtm[3 * D_];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, kq = lane >> 4;
  const Ws& w = p.ws;
  const long R = w.R, RD = R * D_;
  const long row0 = (long)b * S_;
  for (int i = tid; i < S_ * D_; i += 256) scarry[i] = 0.f;
  __syncthreads();
  for (int t = p.L - 1; t >= 0; --t) {
    // ---- gate backward (elementwise): dnext = dout + carry
    for (int i = tid; i < D_; i += 256) {
      float gi = 0.f, gf = 0.f;
#pragma unroll
      for (int s = 0; s < S_; ++s) {
        const long e = (row0 + s) * D_ + i;
        float g = bf2f(p.dout[((long)b * p.L + t) * S_ * D_ + s * D_ + i]);
        if (t != p.L - 1) g += bf2f(f2bf(scarry[s * D_ + i]));          // the launch path hands the carry over as a 16-bit tensor
        const float si = bf2f(w.si[(long)t * RD + e]), sf = bf2f(w.sf[(long)t * RD + e]), tn = bf2f(w.tnm[(long)t * RD + e]);
        const bf16_t dnm2 = f2bf(g * si * (1.f - tn * tn));
        sdnm2[s * LS + i] = dnm2;
        sdmd[s * D_ + i] = bf2f(f2bf(g * sf));
        const float di = g * tn * si * (1.f - si), df = g * bf2f(w.m[(long)t * RD + e]) * sf * (1.f - sf);
        const bf16_t hdi = f2bf(di), hdf = f2bf(df);
        sdg[s * 1032 + i] = hdi; sdg[s * 1032 + D_ + i] = hdf;
        w.dgs[((long)t * R + row0 + s) * 2 * D_ + i] = hdi;
        w.dgs[((long)t * R + row0 + s) * 2 * D_ + D_ + i] = hdf;
        gi += di; gf += df;
        // dh2 = dnm2 * (h2 > 0)
        const bf16_t dh2 = bf2f(w.h2[(long)t * RD + e]) > 0.f ? dnm2 : (bf16_t)0;
        sA[s * LS + i] = dh2;
        w.dh2s[(long)t * RD + e] = dh2;
      }
      p.dgw[((long)b * p.L + t) * 2 * D_ + i] = f2bf(gi);
      p.dgw[((long)b * p.L + t) * 2 * D_ + D_ + i] = f2bf(gf);
    }
    __syncthreads();
    // ---- dtm = dgates . U  (K = 1024)   and   t1 = dh2 . W2 -> dh1 = t1 * (h1 > 0)
    mvgen<2>(p.w.U, 32, sdg, 1032, wave, lane, [&](int rb, const f32x4& acc) {
      if (col < 3) {
        const int o = rb * 16 + kq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) sdtm[col * D_ + o + j] = bf2f(f2bf(acc[j]));
      }
    });
    mvgen<1>(p.w.W2, 32, sA, LS, wave, lane, [&](int rb, const f32x4& acc) {
      if (col < 3) {
        const int o = rb * 16 + kq * 4;
        const uint2 hh = *reinterpret_cast<const uint2*>(w.h1 + (long)t * RD + (row0 + col) * D_ + o);
        // the launch path rounds t1 to 16 bits before the mask
        const float r0 = lo_bf(hh.x) > 0.f ? bf2f(f2bf(acc[0])) : 0.f, r1 = hi_bf(hh.x) > 0.f ? bf2f(f2bf(acc[1])) : 0.f;
        const float r2 = lo_bf(hh.y) > 0.f ? bf2f(f2bf(acc[2])) : 0.f, r3 = hi_bf(hh.y) > 0.f ? bf2f(f2bf(acc[3])) : 0.f;
        st4(sB + col * LS + o, r0, r1, r2, r3);
        st4(w.dh1s + (long)t * RD + (row0 + col) * D_ + o, r0, r1, r2, r3);
      }
    });
    __syncthreads();
    // ---- dnm1 = dh1 . W0 + dnm2
    mvgen<1>(p.w.W0, 32, sB, LS, wave, lane, [&](int rb, const f32x4& acc) {
      if (col < 3) {
        const int o = rb * 16 + kq * 4;
        const uint2 dd = *reinterpret_cast<const uint2*>(sdnm2 + col * LS + o);
        const float r0 = acc[0] + lo_bf(dd.x), r1 = acc[1] + hi_bf(dd.x), r2 = acc[2] + lo_bf(dd.y), r3 = acc[3] + hi_bf(dd.y);
        st4(sdnm1 + col * LS + o, r0, r1, r2, r3);
        st4(w.dnm2s + (long)t * RD + (row0 + col) * D_ + o, r0, r1, r2, r3);
      }
    });
    __syncthreads();
    // ---- da = dnm1 . Wo
    mvgen<1>(p.w.Wo, 32, sdnm1, LS, wave, lane, [&](int rb, const f32x4& acc) {
      if (col < 3) st4(sA + col * LS + rb * 16 + kq * 4, acc[0], acc[1], acc[2], acc[3]);
    });
    __syncthreads();
    // ---- attention backward
    {
      const int h = tid >> 5, l = tid & 31, c = h * DH + 2 * l;
      const bf16_t* qkv_g = w.qkv + ((long)t * R + row0) * 1536;
      float q[S_][2], k[KEYS][2], v[KEYS][2], da[S_][2];
#pragma unroll
      for (int i = 0; i < S_; ++i) {
        const bf16_t* r = qkv_g + (long)i * 1536 + c;
        const uint32_t a = *reinterpret_cast<const uint32_t*>(r), kk = *reinterpret_cast<const uint32_t*>(r + 512), vv = *reinterpret_cast<const uint32_t*>(r + 1024);
        q[i][0] = lo_bf(a); q[i][1] = hi_bf(a); k[i][0] = lo_bf(kk); k[i][1] = hi_bf(kk); v[i][0] = lo_bf(vv); v[i][1] = hi_bf(vv);
        const uint32_t dd = *reinterpret_cast<const uint32_t*>(sA + i * LS + c);
        da[i][0] = lo_bf(dd); da[i][1] = hi_bf(dd);
      }
      const uint32_t kk = *reinterpret_cast<const uint32_t*>(p.xk + ((long)b * p.L + t) * D_ + c);
      const uint32_t vv = *reinterpret_cast<const uint32_t*>(p.xv + ((long)b * p.L + t) * D_ + c);
      k[3][0] = lo_bf(kk); k[3][1] = hi_bf(kk); v[3][0] = lo_bf(vv); v[3][1] = hi_bf(vv);
      const unsigned long long seed = evk_mix_seed(p.seed + 0x51ED27ULL * (uint64_t)(t + 1), p.epoch);
      const float* Pt = w.P + (long)t * p.B * HEADS * S_ * KEYS;
      float dq[S_][2] = {}, dk[KEYS][2] = {}, dv[KEYS][2] = {};
#pragma unroll
      for (int i = 0; i < S_; ++i) {
        float pr[KEYS], dp[KEYS], dot = 0.f;
#pragma unroll
        for (int j = 0; j < KEYS; ++j) {
          pr[j] = Pt[((long)(b * HEADS + h) * S_ + i) * KEYS + j];
          const float ks = keep_scale(seed, ((uint64_t)(b * HEADS + h) * S_ + i) * KEYS + j, p.p_drop);
          dp[j] = half_sum(da[i][0] * v[j][0] + da[i][1] * v[j][1]) * ks;
          dv[j][0] += pr[j] * ks * da[i][0]; dv[j][1] += pr[j] * ks * da[i][1];
          dot += dp[j] * pr[j];
        }
#pragma unroll
        for (int j = 0; j < KEYS; ++j) {
          const float ds = pr[j] * (dp[j] - dot) * 0.125f;
          dq[i][0] += ds * k[j][0]; dq[i][1] += ds * k[j][1];
          dk[j][0] += ds * q[i][0]; dk[j][1] += ds * q[i][1];
        }
      }
      bf16_t* dq_g = w.dqkv + ((long)t * R + row0) * 1536;
#pragma unroll
      for (int i = 0; i < S_; ++i) {
        const uint32_t a = pack2bf(dq[i][0], dq[i][1]), kk2 = pack2bf(dk[i][0], dk[i][1]), vv2 = pack2bf(dv[i][0], dv[i][1]);
        *reinterpret_cast<uint32_t*>(sdqkv + i * 1544 + c) = a;
        *reinterpret_cast<uint32_t*>(sdqkv + i * 1544 + 512 + c) = kk2;
        *reinterpret_cast<uint32_t*>(sdqkv + i * 1544 + 1024 + c) = vv2;
        *reinterpret_cast<uint32_t*>(dq_g + (long)i * 1536 + c) = a;
        *reinterpret_cast<uint32_t*>(dq_g + (long)i * 1536 + 512 + c) = kk2;
        *reinterpret_cast<uint32_t*>(dq_g + (long)i * 1536 + 1024 + c) = vv2;
      }
      *reinterpret_cast<uint32_t*>(p.dxk + ((long)b * p.L + t) * D_ + c) = pack2bf(dk[3][0], dk[3][1]);
      *reinterpret_cast<uint32_t*>(p.dxv + ((long)b * p.L + t) * D_ + c) = pack2bf(dv[3][0], dv[3][1]);
    }
    __syncthreads();
    // ---- carry = (dqkv . Wqkv + dnm1) + dmd + dtm * (1 - tm^2)
    mvgen<3>(p.w.Wqkv, 32, sdqkv, 1544, wave, lane, [&](int rb, const f32x4& acc) {
      if (col < 3) {
        const int o = rb * 16 + kq * 4;
        const uint2 dd = *reinterpret_cast<const uint2*>(sdnm1 + col * LS + o);
        const uint2 tt = *reinterpret_cast<const uint2*>(w.tm + (long)t * RD + (row0 + col) * D_ + o);
        const float dn[4] = {lo_bf(dd.x), hi_bf(dd.x), lo_bf(dd.y), hi_bf(dd.y)};
        const float tm[4] = {lo_bf(tt.x), hi_bf(tt.x), lo_bf(tt.y), hi_bf(tt.y)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float dmp = bf2f(f2bf(acc[j] + dn[j]));
          scarry[col * D_ + o + j] = dmp + sdmd[col * D_ + o + j] + sdtm[col * D_ + o + j] * (1.f - tm[j] * tm[j]);
        }
      }
    });
    __syncthreads();
  }
}

}  // namespace

extern "C" {

/* Forward of RelationalMemory.forward (encoder_decoder.py:293-300) for L tokens.
 *   xk, xv : (B, L, 512) bf16  = x_t.Wk^T + bk, x_t.Wv^T + bv for every token (hoisted out of the recurrence)
 *   gw     : (B, L, 1024) bf16 = W(x_t)
 *   m0     : (B, 3, 512) bf16 initial memory (RelationalMemory.init_memory or the carried decode state)
 *   Wqkv [1536][512] = [linears.0; linears.1; linears.2] bf16, bqkv f32[1536]; Wo, W0 (mlp.0), W2 (mlp.2) [512][512]; U [1024][512]
 *   out    : (B, L, 1536) bf16 memories; m_last (B,3,512) bf16 (may alias nothing) ; ws: evk_rm_ws_bytes (save != 0) */
int evk_rm_forward(const void* xk, const void* xv, const void* gw, const void* m0, const void* Wqkv, const float* bqkv, const void* Wo,
                   const float* bo, const void* W0, const float* b0, const void* W2, const float* b2, const void* U, const float* bU,
                   void* out, void* m_last, void* ws, int64_t ws_bytes, int32_t B, int32_t L, float p_drop, uint64_t seed,
                   evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(xk && xv && gw && m0 && Wqkv && bqkv && Wo && bo && W0 && b0 && W2 && b2 && U && bU && out && ws && B > 0 && L > 0,
              "rm_forward: null/empty argument");
  EVK_REQUIRE(ws_bytes >= evk_rm_ws_bytes(B, L), "rm_forward: workspace too small");
  Ws w = carve(ws, B, L);
  const long R = w.R, RD = R * D_;
  if (rm_use_persistent(L)) {
    // training-length walks: one persistent launch (one workgroup per sample); decode (L = 1, hundreds of hypotheses) keeps
    // the per-token launch path below, where a weight tile is shared by all rows
    PersistP pp{};
    pp.w = PersistW{(const bf16_t*)Wqkv, (const bf16_t*)Wo, (const bf16_t*)W0, (const bf16_t*)W2, (const bf16_t*)U, bqkv, bo, b0, b2, bU};
    pp.xk = (const bf16_t*)xk; pp.xv = (const bf16_t*)xv; pp.gw = (const bf16_t*)gw; pp.m0 = (const bf16_t*)m0;
    pp.out = (bf16_t*)out; pp.m_last = (bf16_t*)m_last; pp.ws = w; pp.B = B; pp.L = L; pp.p_drop = p_drop; pp.seed = seed; pp.epoch = evk_seed_epoch_ptr();
    ProfScope ps(EVK_FAM_GEMM, s, 2.0 * 3 * 4096.0 * 512.0 * B * L);
    hipLaunchKernelGGL(rm_persist_fwd_kernel, dim3(B), dim3(256), 0, s, pp);
    return evk_check_launch("rm_forward (persistent)");
  }
  // copies as kernels (evk_cast 16-bit -> 16-bit), not hipMemcpyAsync: the memcpy nodes a stream capture records for them cannot be read back
  // on ROCm 7.2 (hipGraphMemcpyNodeGetParams returns garbage), which made the step replayer refuse every plan that contains this call
  if (int e = evk_cast(m0, EVK_BF16, w.m, EVK_BF16, RD, stream)) return e;
  if (int e = evk_act_fwd(w.m, w.tm, RD, EVK_ACT_TANH, stream)) return e;   // tm[0] = tanh(m0)
  for (int t = 0; t < L; ++t) {
    const bf16_t* m = w.m + t * RD;
    const bf16_t* tm = w.tm + t * RD;
    bf16_t* qkv = w.qkv + (long)t * R * 1536;
    if (int e = gemm(m, Wqkv, qkv, (int)R, 1536, D_, EVK_B_PLAIN, D_, bqkv, nullptr, EVK_ACT_NONE, stream)) return e;
    AttP ap{qkv, (const bf16_t*)xk + (long)t * D_, (const bf16_t*)xv + (long)t * D_, (long)L * D_, w.P + (long)t * B * HEADS * S_ * KEYS,
            w.a + t * RD, p_drop, (unsigned long long)(seed + 0x51ED27ULL * (uint64_t)(t + 1)), nullptr, nullptr, nullptr, nullptr, evk_seed_epoch_ptr()};
    {
      ProfScope ps(EVK_FAM_NORM, s);
      hipLaunchKernelGGL(rm_attn_fwd_kernel, dim3(B), dim3(256), 0, s, ap);
    }
    if (int e = gemm(w.a + t * RD, Wo, w.nm1 + t * RD, (int)R, D_, D_, EVK_B_PLAIN, D_, bo, m, EVK_ACT_NONE, stream)) return e;
    if (int e = gemm(w.nm1 + t * RD, W0, w.h1 + t * RD, (int)R, D_, D_, EVK_B_PLAIN, D_, b0, nullptr, EVK_ACT_RELU, stream)) return e;
    if (int e = gemm(w.h1 + t * RD, W2, w.h2 + t * RD, (int)R, D_, D_, EVK_B_PLAIN, D_, b2, nullptr, EVK_ACT_RELU, stream)) return e;
    bf16_t* gu = w.gu + (long)t * R * 2 * D_;
    if (int e = gemm(tm, U, gu, (int)R, 2 * D_, D_, EVK_B_PLAIN, D_, bU, nullptr, EVK_ACT_NONE, stream)) return e;
    GateP gp{(const bf16_t*)gw + (long)t * 2 * D_, (long)L * 2 * D_, gu, w.nm1 + t * RD, w.h2 + t * RD, m, w.m + (t + 1) * RD, w.tm + (t + 1) * RD,
             (bf16_t*)out + (long)t * S_ * D_, (long)L * S_ * D_, w.si + t * RD, w.sf + t * RD, w.tnm + t * RD, B};
    {
      ProfScope ps(EVK_FAM_ELTWISE, s);
      hipLaunchKernelGGL(rm_gate_fwd2_kernel, dim3(ew_blocks(RD)), dim3(256), 0, s, gp);
    }
  }
  if (m_last)
    if (int e = evk_cast(w.m + (long)L * RD, EVK_BF16, m_last, EVK_BF16, RD, stream)) return e;
  return evk_check_launch("rm_forward");
}

/* evk_rm_forward with the RECURRENCE IN F32 (the default of the training / teacher-forced forward since round 4).  On the weights the parity
 * fixtures use the memory is an expanding recurrence: with 16-bit operands and a 16-bit carried state the teacher-forced log-probabilities of
 * the training pass drift from the reference's by 0.2-0.3 nats at positions 60-100 of a 100-token report (tests/test_model_gpu.py::
 * test_training_forward_log_probabilities_per_position), exactly as generation did.  Here everything that feeds back into the memory is
 * f32 -- the token embeddings x32 (B, L, 512), their three projections (one f32 product with the stacked MASTER weights Wx32 = [attn.linears.1;
 * attn.linears.2; W], 2048 x 512), the memory, the six f32 master matrices, every intermediate (csrc/rm_f32.hip: v_mfma_f32_16x16x4_f32) -- and
 * 16-bit copies of the intermediates land in the SAME workspace layout as evk_rm_forward's, so evk_rm_backward (16-bit BPTT) runs unchanged.
 *   m0 (B, 3, 512) 16-bit initial memory; out (B, L, 1536) 16-bit memories for the conditional layer norms; ws: evk_rm_ws_bytes;
 *   ws32: evk_rm_f32_ws_bytes(B, L) bytes of f32 scratch. */
int64_t evk_rm_f32_ws_bytes(int32_t B, int32_t L) {
  const long R = (long)B * S_;
  return ((long)B * L * 2048 + 2 * R * D_ + R * 1536 + 4 * R * D_ + R * 2 * D_ + 1024) * 4;
}

int evk_rm_forward_f32(const float* x32, const float* Wx32, const float* bx, const void* m0, const float* Wqkv32, const float* bqkv, const float* Wo32,
                       const float* bo, const float* W032, const float* b0, const float* W232, const float* b2, const float* U32, const float* bU,
                       void* out, void* m_last, void* ws, int64_t ws_bytes, void* ws32, int64_t ws32_bytes, int32_t B, int32_t L, float p_drop,
                       uint64_t seed, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x32 && Wx32 && bx && m0 && Wqkv32 && bqkv && Wo32 && bo && W032 && b0 && W232 && b2 && U32 && bU && out && ws && ws32 && B > 0 && L > 0,
              "rm_forward_f32: null/empty argument");
  EVK_REQUIRE(ws_bytes >= evk_rm_ws_bytes(B, L) && ws32_bytes >= evk_rm_f32_ws_bytes(B, L), "rm_forward_f32: workspace too small");
  EVK_REQUIRE(((reinterpret_cast<uintptr_t>(x32) | reinterpret_cast<uintptr_t>(Wx32) | reinterpret_cast<uintptr_t>(Wqkv32) | reinterpret_cast<uintptr_t>(Wo32) |
                reinterpret_cast<uintptr_t>(W032) | reinterpret_cast<uintptr_t>(W232) | reinterpret_cast<uintptr_t>(U32) | reinterpret_cast<uintptr_t>(ws32)) & 15) == 0,
              "rm_forward_f32: 16-byte aligned buffers");
  Ws w = carve(ws, B, L);
  const long R = w.R, RD = R * D_;
  float* fp = reinterpret_cast<float*>(ws32);
  auto take = [&](long n) { float* r = fp; fp += (n + 63) / 64 * 64; return r; };
  float* xp = take((long)B * L * 2048);            // keys | values | gates of every token, f32
  float* mem = take(RD);
  float* tmem = take(RD);                          // tanh(memory), left by the gate epilogue of the previous token
  float* qkv = take(R * 1536);
  float* a = take(RD); float* nm1 = take(RD); float* h1 = take(RD); float* h2 = take(RD);
  float* gu = take(R * 2 * D_);
  ProfScope ps(EVK_FAM_GEMM, s);
  if (int e = rmf32_gemm(x32, D_, Wx32, bx, nullptr, 0, xp, 2048, B * L, 2048, EVK_ACT_NONE, 0, nullptr, 0, s)) return e;
  if (int e = evk_cast(m0, EVK_BF16, mem, EVK_F32, RD, stream)) return e;
  if (int e = evk_cast(m0, EVK_BF16, w.m, EVK_BF16, RD, stream)) return e;
  if (int e = evk_act_fwd(w.m, w.tm, RD, EVK_ACT_TANH, stream)) return e;
  const long xrow = (long)L * 2048;                 // sample stride of xp
  // timing probe (EVK_EXPERIMENTAL=1 EVK_PROBE_SKIP_RM_FWD=1: garbage memories): the step without the recurrence's forward chain
  static const bool probe_skip_fwd = evk_tunable("EVK_PROBE_SKIP_RM_FWD", 0) != 0;
  for (int t = 0; t < L && !probe_skip_fwd; ++t) {
    const float* xt = xp + (long)t * 2048;
    // 5 launches per token: {q | k | v of the memory, U tanh(m)} as one, slot attention, Wo (+ m), W0, W2 with the gate in its epilogue
    if (int e = rmf32_qkv_gu(mem, t ? tmem : nullptr, Wqkv32, bqkv, qkv, w.qkv + (long)t * R * 1536, U32, bU, gu, (int)R, s)) return e;
    if (int e = rmf32_attn_train(qkv, xt, xrow, a, w.a + t * RD, w.P + (long)t * B * HEADS * S_ * KEYS, p_drop,
                                 (unsigned long long)(seed + 0x51ED27ULL * (uint64_t)(t + 1)), B, s)) return e;
    if (int e = rmf32_gemm(a, D_, Wo32, bo, mem, D_, nm1, D_, (int)R, D_, EVK_ACT_NONE, 0, w.nm1 + t * RD, D_, s)) return e;
    if (int e = rmf32_gemm(nm1, D_, W032, b0, nullptr, 0, h1, D_, (int)R, D_, EVK_ACT_RELU, 0, w.h1 + t * RD, D_, s)) return e;
    if (int e = rmf32_w2_gate_train(h1, W232, b2, w.h2 + t * RD, xt, xrow, gu, nm1, mem, w.m + (t + 1) * RD, w.tm + (t + 1) * RD,
                                    (bf16_t*)out + (long)t * S_ * D_, (long)L * S_ * D_, w.si + t * RD, w.sf + t * RD, w.tnm + t * RD, tmem, (int)R, s)) return e;
  }
  (void)h2;
  if (m_last)
    if (int e = evk_cast(w.m + (long)L * RD, EVK_BF16, m_last, EVK_BF16, RD, stream)) return e;
  return evk_check_launch("rm_forward_f32");
}

/* ONE generated token of the relational memory for R hypotheses (RelationalMemory.forward_step, modules/encoder_decoder.py:274-291, as
 * driven by CaptionModel.beam_search through EncoderDecoder.core): 8 launches, the state updated IN PLACE.
 *   x [B][512]          the token embeddings of the step (16-bit)
 *   Wx [2048][512], bx  = [attn.linears.1; attn.linears.2; W] stacked: the three projections of x_t in one product
 *   mem [B][3][512]     the memory, in / out;  tmem [B][3][512] = tanh(mem), in / out (kept beside it so that the gate product
 *                       U.tanh(m) needs neither a copy nor an activation pass; the caller re-orders both with the hypotheses)
 *   out [B][1536]       the memory row the conditional layer norms of the decoder read (= the new memory)
 *   ws                  >= evk_rm_decode_ws_bytes(B)                                                                         */
int64_t evk_rm_decode_ws_bytes(int32_t B) {
  const long R = (long)B * S_;
  return ((long)B * 2048 + R * 1536 + 4 * R * D_ + R * 2 * D_) * 2 + (long)B * HEADS * S_ * KEYS * 4 + 4096;
}

int evk_rm_decode_step(const void* x, const void* Wx, const float* bx, void* mem, void* tmem, const void* Wqkv, const float* bqkv, const void* Wo,
                       const float* bo, const void* W0, const float* b0, const void* W2, const float* b2, const void* U, const float* bU, void* out,
                       void* ws, int64_t ws_bytes, int32_t B, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && Wx && bx && mem && tmem && Wqkv && bqkv && Wo && bo && W0 && b0 && W2 && b2 && U && bU && out && ws && B > 0,
              "rm_decode_step: null/empty argument");
  EVK_REQUIRE(ws_bytes >= evk_rm_decode_ws_bytes(B), "rm_decode_step: workspace too small");
  const long R = (long)B * S_, RD = R * D_;
  char* p = reinterpret_cast<char*>(ws);
  auto take = [&](long elems) { bf16_t* r = reinterpret_cast<bf16_t*>(p); p += ((elems * 2 + 255) / 256) * 256; return r; };
  bf16_t* xp = take((long)B * 2048);
  bf16_t* qkv = take(R * 1536);
  bf16_t* a = take(RD); bf16_t* nm1 = take(RD); bf16_t* h1 = take(RD); bf16_t* h2 = take(RD);
  bf16_t* gu = take(R * 2 * D_);
  float* P = reinterpret_cast<float*>(p);
  bf16_t* m = (bf16_t*)mem;
  bf16_t* tm = (bf16_t*)tmem;
  if (int e = gemm(x, Wx, xp, B, 2048, D_, EVK_B_PLAIN, D_, bx, nullptr, EVK_ACT_NONE, stream)) return e;        // xk | xv | gates(x)
  if (int e = gemm(m, Wqkv, qkv, (int)R, 1536, D_, EVK_B_PLAIN, D_, bqkv, nullptr, EVK_ACT_NONE, stream)) return e;
  AttP ap{qkv, xp, xp + D_, 2048L, P, a, 0.f, 0ULL, nullptr, nullptr, nullptr, nullptr, evk_seed_epoch_ptr()};
  {
    ProfScope ps(EVK_FAM_NORM, s);
    hipLaunchKernelGGL(rm_attn_fwd_kernel, dim3(B), dim3(256), 0, s, ap);
  }
  if (int e = gemm(a, Wo, nm1, (int)R, D_, D_, EVK_B_PLAIN, D_, bo, m, EVK_ACT_NONE, stream)) return e;
  if (int e = gemm(nm1, W0, h1, (int)R, D_, D_, EVK_B_PLAIN, D_, b0, nullptr, EVK_ACT_RELU, stream)) return e;
  if (int e = gemm(h1, W2, h2, (int)R, D_, D_, EVK_B_PLAIN, D_, b2, nullptr, EVK_ACT_RELU, stream)) return e;
  if (int e = gemm(tm, U, gu, (int)R, 2 * D_, D_, EVK_B_PLAIN, D_, bU, nullptr, EVK_ACT_NONE, stream)) return e;
  // every element of m / tm is read and rewritten by the same thread: in place
  GateP gp{xp + 2 * D_, 2048L, gu, nm1, h2, m, m, tm, (bf16_t*)out, (long)S_ * D_, nullptr, nullptr, nullptr, B};
  {
    ProfScope ps(EVK_FAM_ELTWISE, s);
    hipLaunchKernelGGL(rm_gate_fwd2_kernel, dim3(ew_blocks(RD)), dim3(256), 0, s, gp);
  }
  return evk_check_launch("rm_decode_step");
}

/* Backward through the recurrence (BPTT) with the workspace evk_rm_forward filled.
 *   Wqkvt [512][1536], Wot, W0t, W2t [512][512], Ut [512][1024]: the TRANSPOSED bf16 weights (so every data-gradient
 *   product is the K-contiguous form the latency-optimised small-GEMM kernel takes)
 *   dout (B, L, 1536) bf16 -> dxk, dxv (B, L, 512), dgw (B, L, 1024) bf16 (written), and the f32 gradients of
 *   Wqkv [1536][512], bqkv[1536], Wo, bo, W0, b0, W2, b2, U [1024][512], bU[1024] are ACCUMULATED (+=).           */
int evk_rm_backward(const void* dout, const void* xk, const void* xv, const void* Wqkvt, const void* Wot, const void* W0t, const void* W2t,
                    const void* Ut, void* dxk, void* dxv, void* dgw, float* dWqkv, float* dbqkv, float* dWo, float* dbo, float* dW0,
                    float* db0, float* dW2, float* db2, float* dU, float* dbU, void* ws, int64_t ws_bytes, int32_t B, int32_t L,
                    float p_drop, uint64_t seed, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dout && xk && xv && Wqkvt && Wot && W0t && W2t && Ut && dxk && dxv && dgw && dWqkv && dbqkv && dWo && dbo && dW0 && db0 && dW2 &&
              db2 && dU && dbU && ws && B > 0 && L > 0, "rm_backward: null/empty argument");
  EVK_REQUIRE(ws_bytes >= evk_rm_ws_bytes(B, L), "rm_backward: workspace too small");
  Ws w = carve(ws, B, L);
  const long R = w.R, RD = R * D_;
  // timing probe (EVK_EXPERIMENTAL=1 EVK_PROBE_SKIP_RM_BWD=1: wrong gradients): the step without the recurrence's backward chain
  static const bool probe_skip = evk_tunable("EVK_PROBE_SKIP_RM_BWD", 0) != 0;
  if (probe_skip) return EVK_OK;
  const bool persistent = rm_use_persistent(L);
  if (persistent) {
    PersistP pp{};
    pp.w = PersistW{(const bf16_t*)Wqkvt, (const bf16_t*)Wot, (const bf16_t*)W0t, (const bf16_t*)W2t, (const bf16_t*)Ut, nullptr, nullptr, nullptr, nullptr, nullptr};
    pp.xk = (const bf16_t*)xk; pp.xv = (const bf16_t*)xv; pp.ws = w; pp.B = B; pp.L = L; pp.p_drop = p_drop; pp.seed = seed; pp.epoch = evk_seed_epoch_ptr();
    pp.dout = (const bf16_t*)dout; pp.dxk = (bf16_t*)dxk; pp.dxv = (bf16_t*)dxv; pp.dgw = (bf16_t*)dgw;
    ProfScope ps(EVK_FAM_GEMM, s, 2.0 * 3 * 4096.0 * 512.0 * B * L);
    hipLaunchKernelGGL(rm_persist_bwd_kernel, dim3(B), dim3(256), 0, s, pp);
    if (int e = evk_check_launch("rm_backward (persistent)")) return e;
  }
  for (int t = L - 1; t >= 0 && !persistent; --t) {
    bf16_t* dgs = w.dgs + (long)t * R * 2 * D_;
    bf16_t* dnm2 = w.dnm2s + t * RD;            // becomes dnm1 (stack) below
    bf16_t* dh2 = w.dh2s + t * RD;
    bf16_t* dh1 = w.dh1s + t * RD;
    // (t < L - 1: the carry of token t + 1 is formed from that token's dmp / dmd / dtm, still in the temporaries, and its tanh(m))
    const bool cf = t < L - 1;
    GateBP gb{(const bf16_t*)dout + (long)t * S_ * D_, (long)L * S_ * D_, nullptr, w.si + t * RD, w.sf + t * RD, w.tnm + t * RD, w.m + t * RD,
              w.t_dnm1, w.t_dmd, dgs, (bf16_t*)dgw + (long)t * 2 * D_, (long)L * 2 * D_, B, w.h2 + t * RD, dh2,
              cf ? w.t_dmp : nullptr, cf ? w.t_dmd : nullptr, cf ? w.t_dtm : nullptr, cf ? w.tm + (t + 1) * RD : nullptr};
    {
      ProfScope ps(EVK_FAM_ELTWISE, s);
      hipLaunchKernelGGL(rm_gate_bwd2_kernel, dim3(ew_blocks((long)B * D_)), dim3(256), 0, s, gb);
    }
    // dtm = dgates . U
    if (int e = gemm(dgs, Ut, w.t_dtm, (int)R, D_, 2 * D_, EVK_B_PLAIN, 2 * D_, nullptr, nullptr, EVK_ACT_NONE, stream)) return e;
    // dh2 = dnm2 * (h2 > 0) (written by the gate kernel);  dh1 = (dh2 . W2) * (h1 > 0) (ReLU mask in the product's epilogue);
    // dnm1 = dh1 . W0 + dnm2        -- (10 launches per token with the two masks and the carry as launches of their own)
    if (int e = gemm(dh2, W2t, dh1, (int)R, D_, D_, EVK_B_PLAIN, D_, nullptr, nullptr, EVK_ACT_NONE, stream, w.h1 + t * RD)) return e;
    if (int e = gemm(dh1, W0t, dnm2, (int)R, D_, D_, EVK_B_PLAIN, D_, nullptr, w.t_dnm1, EVK_ACT_NONE, stream)) return e;   // dnm2 now holds dnm1
    // da = dnm1 . Wo ; attention backward
    if (int e = gemm(dnm2, Wot, w.t_da, (int)R, D_, D_, EVK_B_PLAIN, D_, nullptr, nullptr, EVK_ACT_NONE, stream)) return e;
    bf16_t* dqkv = w.dqkv + (long)t * R * 1536;
    AttP ap{w.qkv + (long)t * R * 1536, (const bf16_t*)xk + (long)t * D_, (const bf16_t*)xv + (long)t * D_, (long)L * D_,
            w.P + (long)t * B * HEADS * S_ * KEYS, nullptr, p_drop, (unsigned long long)(seed + 0x51ED27ULL * (uint64_t)(t + 1)), w.t_da, dqkv,
            (bf16_t*)dxk + (long)t * D_, (bf16_t*)dxv + (long)t * D_, evk_seed_epoch_ptr()};
    {
      ProfScope ps(EVK_FAM_NORM, s);
      hipLaunchKernelGGL(rm_attn_bwd_kernel, dim3(B), dim3(256), 0, s, ap);
    }
    // dm(from projections) = dqkv . Wqkv + dnm1
    // (its sum with dmd and dtm * (1 - tm^2) -- the gradient carried to token t - 1 -- is formed by that token's gate kernel: 7 launches per token)
    if (int e = gemm(dqkv, Wqkvt, w.t_dmp, (int)R, D_, 1536, EVK_B_PLAIN, 1536, nullptr, dnm2, EVK_ACT_NONE, stream)) return e;
  }
  // weight / bias gradients: one GEMM (K = L*R rows) + one column sum per parameter
  const long rows = (long)L * R;
  if (int e = wgrad(w.dqkv, w.m, dWqkv, rows, 1536, D_, w.slab, w.slab_bytes, stream)) return e;
  if (int e = evk_colsum(w.dqkv, dbqkv, rows, 1536, 1536, stream)) return e;
  if (int e = wgrad(w.dnm2s, w.a, dWo, rows, D_, D_, w.slab, w.slab_bytes, stream)) return e;
  if (int e = evk_colsum(w.dnm2s, dbo, rows, D_, D_, stream)) return e;
  if (int e = wgrad(w.dh1s, w.nm1, dW0, rows, D_, D_, w.slab, w.slab_bytes, stream)) return e;
  if (int e = evk_colsum(w.dh1s, db0, rows, D_, D_, stream)) return e;
  if (int e = wgrad(w.dh2s, w.h1, dW2, rows, D_, D_, w.slab, w.slab_bytes, stream)) return e;
  if (int e = evk_colsum(w.dh2s, db2, rows, D_, D_, stream)) return e;
  if (int e = wgrad(w.dgs, w.tm, dU, rows, 2 * D_, D_, w.slab, w.slab_bytes, stream)) return e;
  if (int e = evk_colsum(w.dgs, dbU, rows, 2 * D_, 2 * D_, stream)) return e;
  return evk_check_launch("rm_backward");
}

}  // extern "C"
