// rm.hip -- the relational memory of the R2Gen decoder (modules/encoder_decoder.py:246-300) as a native runner.
//
// RelationalMemory.forward is a serial recurrence over the L report tokens; the reference walks it in a Python loop
// of ~25 tiny torch kernels per token (and the naive engine did the same with ~15 autograd nodes per token, i.e.
// ~4.5k launches + Python overhead per training step).  Here the host loop lives in C++ and every token is
//     fwd: 5 GEMM launches (m.[Wq;Wk;Wv], a.Wo + m, relu(.W0), relu(.W2), tanh(m).U) + 1 fused attention kernel
//          (3 slots x 4 keys x 8 heads per sample, softmax + dropout + PV) + 1 fused gate kernel
//     bwd: 5 data-gradient GEMMs + gate / relu-mask / attention / combine kernels,
// while everything that does not depend on the recurrence is hoisted out of the loop by the caller (x_t.Wk, x_t.Wv and
// W(x_t) for all t are three ordinary batched GEMMs) and all weight gradients are ONE K = L*B*slots GEMM per weight
// after the loop (the per-token operands are saved contiguously in the workspace).
#include "common.h"

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
      if (p.dcarry) g += bf2f(p.dcarry[e]);
      const float si = bf2f(p.si[e]), sf = bf2f(p.sf[e]), t = bf2f(p.tnm[e]);
      p.dnm2[e] = f2bf(g * si * (1.f - t * t));
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
// dm = a + dmd + dtm * (1 - tm^2)
__global__ __launch_bounds__(256) void rm_combine_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ dmd, const bf16_t* __restrict__ dtm,
                                                         const bf16_t* __restrict__ tm, bf16_t* __restrict__ dm, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float t = bf2f(tm[i]);
    dm[i] = f2bf(bf2f(a[i]) + bf2f(dmd[i]) + bf2f(dtm[i]) * (1.f - t * t));
  }
}

inline int ew_blocks(long work) { long b = cdiv(work, 256); return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b)); }

int gemm(const void* A, const void* B, void* C, int M, int N, int K, int b_mode, long ldb, const float* bias, const void* resid, int act,
         evk_stream_t st) {
  evk_gemm d{};
  d.A = A; d.B = B; d.C = C; d.bias = bias; d.resid = resid;
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
  if (hipMemcpyAsync(w.m, m0, RD * 2, hipMemcpyDeviceToDevice, s) != hipSuccess) { evk_set_error("rm_forward: memcpy failed"); return EVK_ELAUNCH; }
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
  if (m_last && hipMemcpyAsync(m_last, w.m + (long)L * RD, RD * 2, hipMemcpyDeviceToDevice, s) != hipSuccess) {
    evk_set_error("rm_forward: memcpy failed");
    return EVK_ELAUNCH;
  }
  return evk_check_launch("rm_forward");
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
  const bf16_t* carry = nullptr;
  for (int t = L - 1; t >= 0; --t) {
    bf16_t* dgs = w.dgs + (long)t * R * 2 * D_;
    bf16_t* dnm2 = w.dnm2s + t * RD;            // becomes dnm1 (stack) below
    GateBP gb{(const bf16_t*)dout + (long)t * S_ * D_, (long)L * S_ * D_, carry, w.si + t * RD, w.sf + t * RD, w.tnm + t * RD, w.m + t * RD,
              w.t_dnm1, w.t_dmd, dgs, (bf16_t*)dgw + (long)t * 2 * D_, (long)L * 2 * D_, B};
    {
      ProfScope ps(EVK_FAM_ELTWISE, s);
      hipLaunchKernelGGL(rm_gate_bwd2_kernel, dim3(ew_blocks((long)B * D_)), dim3(256), 0, s, gb);
    }
    // dtm = dgates . U
    if (int e = gemm(dgs, Ut, w.t_dtm, (int)R, D_, 2 * D_, EVK_B_PLAIN, 2 * D_, nullptr, nullptr, EVK_ACT_NONE, stream)) return e;
    // dh2 = dnm2 * (h2 > 0);  t1 = dh2 . W2;  dh1 = t1 * (h1 > 0);  dnm1 = dh1 . W0 + dnm2
    bf16_t* dh2 = w.dh2s + t * RD;
    bf16_t* dh1 = w.dh1s + t * RD;
    {
      ProfScope ps(EVK_FAM_ELTWISE, s);
      hipLaunchKernelGGL(relu_mask_kernel, dim3(ew_blocks(RD)), dim3(256), 0, s, w.t_dnm1, w.h2 + t * RD, dh2, RD);
    }
    if (int e = gemm(dh2, W2t, w.t_t1, (int)R, D_, D_, EVK_B_PLAIN, D_, nullptr, nullptr, EVK_ACT_NONE, stream)) return e;
    {
      ProfScope ps(EVK_FAM_ELTWISE, s);
      hipLaunchKernelGGL(relu_mask_kernel, dim3(ew_blocks(RD)), dim3(256), 0, s, w.t_t1, w.h1 + t * RD, dh1, RD);
    }
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
    if (int e = gemm(dqkv, Wqkvt, w.t_dmp, (int)R, D_, 1536, EVK_B_PLAIN, 1536, nullptr, dnm2, EVK_ACT_NONE, stream)) return e;
    bf16_t* nc = (t & 1) ? w.t_carry1 : w.t_carry0;
    {
      ProfScope ps(EVK_FAM_ELTWISE, s);
      hipLaunchKernelGGL(rm_combine_kernel, dim3(ew_blocks(RD)), dim3(256), 0, s, w.t_dmp, w.t_dmd, w.t_dtm, w.tm + t * RD, nc, RD);
    }
    carry = nc;
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
