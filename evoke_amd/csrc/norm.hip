// norm.hip -- row-wise kernels (one 64-lane wavefront per row, shuffles for the reductions):
//   * LayerNorm family: torch.nn.LayerNorm (bert_model.py:352-363,430-441; ...v0623_large_res.py:34-35),
//     R2Gen LayerNorm (encoder_decoder.py:93-103: unbiased std, eps added to the std) and
//     ConditionalLayerNorm (encoder_decoder.py:166-179: per-row gamma/beta deltas), forward + backward
//   * masked softmax forward/backward for the attention products (encoder_decoder.py:20-28,
//     bert_model.py:322-337, utils_v0511.py:267-273) with the reference's dropout on the probabilities
//   * log_softmax + masked NLL (encoder_decoder.py:393, loss.py:9-16) forward/backward, plain log_softmax
//   * L2 row normalisation and soft-label cross-entropy for the contrastive losses (...v0623...:262-351)
// HBM-bound: each row is read once with 16-byte loads, kept in registers, written once.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int WPB = 4;      // waves (rows) per 256-thread block
constexpr int MAXC = 4;     // 16-byte chunks per lane -> D <= 64*8*4 = 2048 (bf16) ; f32 rows use 2 loads per chunk

__device__ __forceinline__ void load8(const void* base, int is_f32, long elem, float (&v)[8]) {
  if (is_f32) {
    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
    const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const uint4 a = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(base) + elem);
    v[0] = lo_bf(a.x); v[1] = hi_bf(a.x); v[2] = lo_bf(a.y); v[3] = hi_bf(a.y);
    v[4] = lo_bf(a.z); v[5] = hi_bf(a.z); v[6] = lo_bf(a.w); v[7] = hi_bf(a.w);
  }
}
__device__ __forceinline__ void store8(void* base, int is_f32, long elem, const float (&v)[8]) {
  if (is_f32) {
    float* p = reinterpret_cast<float*>(base) + elem;
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(base) + elem) =
        make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
  }
}

struct LnP {
  const void* x; void* y; const float* gamma; const float* beta; const void* dgam; const void* dbet;  // deltas: [rows][D]
  float* mean; float* rstd;
  long rows; int D; int mode; float eps; int x_f32, y_f32, d_f32;
};

// mode 0: y = (x-mu)*rsqrt(var_biased+eps)*g + b ; mode 1: y = g*(x-mu)/(std_unbiased+eps) + b
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnP p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int nch = p.D >> 3;
  float v[MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      load8(p.x, p.x_f32, row * p.D + ch * 8, v[c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[c][j];
    }
  }
  const float mu = wave_sum(s) / p.D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    if (lane + 64 * c < nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mu; q += d * d; }
    }
  }
  q = wave_sum(q);
  const float r = p.mode == 0 ? rsqrtf(q / p.D + p.eps) : 1.f / (sqrtf(q / (p.D - 1)) + p.eps);
  if (lane == 0 && p.mean) { p.mean[row] = mu; p.rstd[row] = r; }
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      float g[8], b[8], o[8];
      load8(p.gamma, 1, ch * 8, g);
      load8(p.beta, 1, ch * 8, b);
      if (p.dgam) {
        float dg[8], db[8];
        load8(p.dgam, p.d_f32, row * p.D + ch * 8, dg);
        load8(p.dbet, p.d_f32, row * p.D + ch * 8, db);
#pragma unroll
        for (int j = 0; j < 8; ++j) { g[j] += dg[j]; b[j] += db[j]; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mu) * r * g[j] + b[j];
      store8(p.y, p.y_f32, row * p.D + ch * 8, o);
    }
  }
}

struct LnBP {
  const void* dy; const void* x; const float* gamma; const void* dgam; const float* mean; const float* rstd;
  void* dx; float* dgamma; float* dbeta;   // f32 [D], accumulated (+=) with atomics
  void* ddgam; void* ddbet;                // CLN: per-row grads of the deltas [rows][D] (same dtype as dgam)
  long rows; int D; int mode; float eps; int dy_f32, x_f32, dx_f32, d_f32;
};

__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBP p) {
  __shared__ float red[2][WPB][64 * 8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nch = p.D >> 3;
  float ag[MAXC][8], ab[MAXC][8];
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) ag[c][j] = ab[c][j] = 0.f;
  for (long row = (long)blockIdx.x * WPB + w; row < p.rows; row += (long)gridDim.x * WPB) {
    const float mu = p.mean[row], r = p.rstd[row];
    float g[MAXC][8], xh[MAXC][8];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float dy[8], xv[8], gm[8];
        load8(p.dy, p.dy_f32, row * p.D + ch * 8, dy);
        load8(p.x, p.x_f32, row * p.D + ch * 8, xv);
        load8(p.gamma, 1, ch * 8, gm);
        if (p.dgam) {
          float dg[8];
          load8(p.dgam, p.d_f32, row * p.D + ch * 8, dg);
#pragma unroll
          for (int j = 0; j < 8; ++j) gm[j] += dg[j];
        }
        float t1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[c][j] = (xv[j] - mu) * r;
          g[c][j] = dy[j] * gm[j];
          sg += g[c][j];
          sgx += g[c][j] * xh[c][j];
          t1[j] = dy[j] * xh[c][j];
          ag[c][j] += t1[j];
          ab[c][j] += dy[j];
        }
        if (p.ddgam) {
          store8(p.ddgam, p.d_f32, row * p.D + ch * 8, t1);
          store8(p.ddbet, p.d_f32, row * p.D + ch * 8, dy);
        }
      }
    }
    sg = wave_sum(sg);
    sgx = wave_sum(sgx);
    const float mg = sg / p.D;
    // mode 0: dx = r*(g - mean(g) - xh*mean(g*xh));  mode 1: dx = r*(g - mean(g)) - xh*sum(g*xh)/((D-1)*sigma)
    const float k2 = p.mode == 0 ? r * sgx / p.D : sgx / ((p.D - 1) * (1.f / r - p.eps));
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = r * (g[c][j] - mg) - xh[c][j] * k2;
        store8(p.dx, p.dx_f32, row * p.D + ch * 8, o);
      }
    }
  }
  if (!p.dgamma) return;
  // block reduction of the per-column partials, then one atomic per column per block
  for (int c = 0; c < MAXC; ++c) {
    if (64 * c >= nch) break;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][w][lane * 8 + j] = ag[c][j]; red[1][w][lane * 8 + j] = ab[c][j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
      const int col = 512 * c + i;
      if (col < p.D) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int ww = 0; ww < WPB; ++ww) { a += red[0][ww][i]; b += red[1][ww][i]; }
        unsafeAtomicAdd(p.dgamma + col, a);
        unsafeAtomicAdd(p.dbeta + col, b);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// softmax over the last dim of scores[z][q][ld] (cols valid: S) with key / full masks and dropout
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}
__device__ __forceinline__ bool keep_elem(uint64_t seed, uint64_t idx, float p) {
  return (hash32(seed * 0x9E3779B97F4A7C15ULL + idx) >> 8) * (1.f / 16777216.f) >= p;
}

struct SmP {
  const float* s; void* p_out; void* pd_out;   // P (pre-dropout) and P' (post-dropout, may alias when p_drop == 0)
  const unsigned char* mask;                   // 1 = keep; index = zo*mBo + q*mQ + col (mQ = 0 for key masks)
  long rows; int Tq, S, ld_in, ld_out, heads; long mBo; int mQ; int causal;
  float p_drop; unsigned long long seed; int p_f32; const unsigned long long* epoch;
};
constexpr int SMC = 16;  // columns per lane -> S <= 1024

__global__ __launch_bounds__(256) void softmax_fwd_kernel(const SmP p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int q = (int)(row % p.Tq);
  const long zo = row / p.Tq / p.heads;
  const float* in = p.s + row * p.ld_in;
  const unsigned char* mk = p.mask ? p.mask + zo * p.mBo + (long)q * p.mQ : nullptr;
  float v[SMC];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < SMC; ++c) {
    const int col = lane + 64 * c;
    float t = -INFINITY;
    if (col < p.S) {
      bool on = !(mk && !mk[col]);
      if (p.causal && col > q) on = false;
      if (on) t = in[col];
    }
    v[c] = t;
    mx = fmaxf(mx, t);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < SMC; ++c) { v[c] = (v[c] == -INFINITY) ? 0.f : __expf(v[c] - mx); sum += v[c]; }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  const float sc = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
#pragma unroll
  for (int c = 0; c < SMC; ++c) {
    const int col = lane + 64 * c;
    if (col < p.ld_out) {
      const float pr = col < p.S ? v[c] * inv : 0.f;
      const long o = row * p.ld_out + col;
      if (p.p_f32) reinterpret_cast<float*>(p.p_out)[o] = pr; else reinterpret_cast<bf16_t*>(p.p_out)[o] = f2bf(pr);
      if (p.p_drop > 0.f) {
        const float pd = keep_elem(evk_mix_seed(p.seed, p.epoch), (uint64_t)o, p.p_drop) ? pr * sc : 0.f;
        if (p.p_f32) reinterpret_cast<float*>(p.pd_out)[o] = pd; else reinterpret_cast<bf16_t*>(p.pd_out)[o] = f2bf(pd);
      }
    }
  }
}

struct SmBP {
  const void* dp; const void* pr; void* ds;  // dP' (f32 or bf16), P (pre-dropout), dS out (same dtype as P)
  long rows; int S, ld_dp, ld, dp_f32; float p_drop; unsigned long long seed; float alpha; int p_f32; const unsigned long long* epoch;
};
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const SmBP p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const float sc = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
  float d[SMC], pv[SMC];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < SMC; ++c) {
    const int col = lane + 64 * c;
    d[c] = 0.f; pv[c] = 0.f;
    if (col < p.S) {
      float g = p.dp_f32 ? reinterpret_cast<const float*>(p.dp)[row * p.ld_dp + col]
                         : bf2f(reinterpret_cast<const bf16_t*>(p.dp)[row * p.ld_dp + col]);
      if (p.p_drop > 0.f) g = keep_elem(evk_mix_seed(p.seed, p.epoch), (uint64_t)row * p.ld + col, p.p_drop) ? g * sc : 0.f;
      pv[c] = p.p_f32 ? reinterpret_cast<const float*>(p.pr)[row * p.ld + col] : bf2f(reinterpret_cast<const bf16_t*>(p.pr)[row * p.ld + col]);
      d[c] = g;
      dot += g * pv[c];
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int c = 0; c < SMC; ++c) {
    const int col = lane + 64 * c;
    if (col < p.ld) {
      const float o = col < p.S ? pv[c] * (d[c] - dot) * p.alpha : 0.f;
      if (p.p_f32) reinterpret_cast<float*>(p.ds)[row * p.ld + col] = o; else reinterpret_cast<bf16_t*>(p.ds)[row * p.ld + col] = f2bf(o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// log_softmax (+ masked NLL) over logits[rows][ld] f32, V valid columns
// ------------------------------------------------------------------------------------------------
struct LsmP {
  const float* logits; float* logp; const long long* target; const float* wmask; float* acc2;  // acc2 = {sum(-logp*w), sum(w)}
  float* lse; long rows; int V, ld, ld_out;
  float* rownll;   // optional [rows]: -logp[target] * w per row instead of the atomic accumulation (summed by the caller: order-independent bits)
};
// REG = true (V <= 24 * 64): the row is read ONCE into registers (one memory round trip; the three passes below re-read it from L2 each --
// the decode step calls this once per generated token on 256 rows x 1445 and is latency bound)
template <bool REG>
__global__ __launch_bounds__(256) void logsoftmax_kernel(const LsmP p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const float* in = p.logits + row * p.ld;
  float mx = -INFINITY;
  float xr[REG ? 24 : 1];
  if constexpr (REG) {
#pragma unroll
    for (int i = 0; i < 24; ++i) { const int c = lane + 64 * i; xr[i] = c < p.V ? in[c] : -INFINITY; mx = fmaxf(mx, xr[i]); }
  } else {
    for (int c = lane; c < p.V; c += 64) mx = fmaxf(mx, in[c]);
  }
  mx = wave_max(mx);
  float s = 0.f;
  if constexpr (REG) {
#pragma unroll
    for (int i = 0; i < 24; ++i) s += (lane + 64 * i < p.V) ? __expf(xr[i] - mx) : 0.f;
  } else {
    for (int c = lane; c < p.V; c += 64) s += __expf(in[c] - mx);
  }
  s = wave_sum(s);
  const float lse = mx + __logf(s);
  if (p.lse && lane == 0) p.lse[row] = lse;
  if (p.logp) {
    float* o = p.logp + row * p.ld_out;
    if constexpr (REG) {
#pragma unroll
      for (int i = 0; i < 24; ++i) { const int c = lane + 64 * i; if (c < p.V) o[c] = xr[i] - lse; }
    } else {
      for (int c = lane; c < p.V; c += 64) o[c] = in[c] - lse;
    }
  }
  if (p.target && lane == 0) {
    const float w = p.wmask[row];
    if (p.rownll) p.rownll[row] = w != 0.f ? -(in[p.target[row]] - lse) * w : 0.f;
    else if (w != 0.f) {
      unsafeAtomicAdd(p.acc2, -(in[p.target[row]] - lse) * w);
      unsafeAtomicAdd(p.acc2 + 1, w);
    }
  }
}

struct NllBP { const float* logits; const float* lse; const long long* target; const float* wmask; const float* gscale;
               bf16_t* dlogits; long rows; int V, ld, ld_out; };
// dlogits = (softmax - onehot) * w[row] * gscale[0]      (gscale = dLoss / sum(w), computed on device by the caller)
__global__ __launch_bounds__(256) void nll_bwd_kernel(const NllBP p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const float* in = p.logits + row * p.ld;
  const float k = p.wmask[row] * p.gscale[0];
  const float lse = p.lse[row];
  const int t = (int)p.target[row];
  bf16_t* o = p.dlogits + row * p.ld_out;
  for (int c = lane; c < p.ld_out; c += 64) {
    float g = 0.f;
    if (c < p.V && k != 0.f) g = (__expf(in[c] - lse) - (c == t ? 1.f : 0.f)) * k;
    o[c] = f2bf(g);
  }
}

// ------------------------------------------------------------------------------------------------
// L2 row normalisation (F.normalize, eps 1e-12) fwd/bwd, f32 in/out, D arbitrary
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* x, float* y, float* nrm, long rows, int D) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) { const float v = x[row * D + c]; s += v * v; }
  const float n = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  if (lane == 0) nrm[row] = n;
  for (int c = lane; c < D; c += 64) y[row * D + c] = x[row * D + c] / n;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* dy, const float* y, const float* nrm, float* dx, long rows, int D) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += dy[row * D + c] * y[row * D + c];
  s = wave_sum(s);
  const float n = nrm[row];
  for (int c = lane; c < D; c += 64) dx[row * D + c] = (dy[row * D + c] - y[row * D + c] * s) / n;
}

// soft-label cross entropy over rows: loss += sum_rows( -sum_c t[c] * log_softmax(z)[c] ) * scale ; dz = (softmax*sum(t) - t)*gs
struct SceP { const float* z; const float* t; float* loss; float* dz; const float* gscale; long rows; int Cn; float scale; int diag_mask; };
__global__ __launch_bounds__(256) void softce_kernel(const SceP p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const float* z = p.z + row * p.Cn;
  const float* t = p.t + row * p.Cn;
  float mx = -INFINITY;
  for (int c = lane; c < p.Cn; c += 64) { const float v = (p.diag_mask && c == row) ? -1e9f : z[c]; mx = fmaxf(mx, v); }
  mx = wave_max(mx);
  float s = 0.f, ts = 0.f, tz = 0.f;
  for (int c = lane; c < p.Cn; c += 64) {
    const float v = (p.diag_mask && c == row) ? -1e9f : z[c];
    s += __expf(v - mx); ts += t[c]; tz += t[c] * v;
  }
  s = wave_sum(s); ts = wave_sum(ts); tz = wave_sum(tz);
  const float lse = mx + __logf(s);
  if (p.loss && lane == 0) unsafeAtomicAdd(p.loss, (lse * ts - tz) * p.scale);
  if (p.dz) {
    const float gs = p.gscale[0] * p.scale;
    for (int c = lane; c < p.Cn; c += 64) {
      const bool dm = p.diag_mask && c == row;
      const float v = dm ? -1e9f : z[c];
      p.dz[row * p.Cn + c] = dm ? 0.f : (__expf(v - lse) * ts - t[c]) * gs;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// beam step: per row, the k (<= 8) largest of x[row][0..n) in descending order; ties -> lowest index first
// (replaces the flat torch.sort over beam*(V+1) candidates, caption_model.py:70-74)
// ------------------------------------------------------------------------------------------------
constexpr int TOPK_MAX = 8;
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, float* __restrict__ vals,
                                                        long long* __restrict__ idx, int n, int k) {
  __shared__ float sv[256 * TOPK_MAX];
  __shared__ int si[256 * TOPK_MAX];
  const float* in = x + (long)blockIdx.x * n;
  float bv[TOPK_MAX]; int bi[TOPK_MAX];
#pragma unroll
  for (int j = 0; j < TOPK_MAX; ++j) { bv[j] = -INFINITY; bi[j] = 0x7fffffff; }
  for (int c = threadIdx.x; c < n; c += 256) {
    float v = in[c]; int id = c;
#pragma unroll
    for (int j = 0; j < TOPK_MAX; ++j) {       // insertion into the sorted local list
      if (j < k && (v > bv[j] || (v == bv[j] && id < bi[j]))) { const float tv = bv[j]; const int ti = bi[j]; bv[j] = v; bi[j] = id; v = tv; id = ti; }
    }
  }
#pragma unroll
  for (int j = 0; j < TOPK_MAX; ++j) { sv[threadIdx.x * TOPK_MAX + j] = bv[j]; si[threadIdx.x * TOPK_MAX + j] = bi[j]; }
  __syncthreads();
  if (threadIdx.x < 64) {        // one wave selects the k winners, one per round
    for (int r = 0; r < k; ++r) {
      float best = -INFINITY; int bid = 0x7fffffff, slot = -1;
      for (int c = threadIdx.x; c < 256 * TOPK_MAX; c += 64) {
        const float v = sv[c]; const int id = si[c];
        if (id != 0x7fffffff && (v > best || (v == best && id < bid))) { best = v; bid = id; slot = c; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bid, o, 64); const int os = __shfl_xor(slot, o, 64);
        if (oi != 0x7fffffff && (ov > best || (ov == best && oi < bid))) { best = ov; bid = oi; slot = os; }
      }
      if (threadIdx.x == 0) { vals[(long)blockIdx.x * k + r] = best; idx[(long)blockIdx.x * k + r] = bid; if (slot >= 0) si[slot] = 0x7fffffff; }
      __syncthreads();
    }
  }
}

inline int row_blocks(long rows) { return (int)cdiv(rows, WPB); }

// ------------------------------------------------------------------------------------------------
// single-query attention of the incremental decode step: out[r][h] = softmax(scale * q[r][h] . K[r][:, h]) V[r][:, h]
// for one new token per hypothesis (T = 1).  One wavefront per (row, head), head size 64: lanes own key positions for the
// scores (16-byte loads of a 128-byte key row against q broadcast from LDS), wave-shuffle softmax in f32, then lanes own
// the 64 output features and walk the keys with the probabilities broadcast by shuffle.  Replaces three launches (batched
// QK^T GEMM with M = 1, softmax, batched PV GEMM) whose tiles were 99 % padding.
// ------------------------------------------------------------------------------------------------
struct DecAttP { const bf16_t* q; const bf16_t* k; const bf16_t* v; const unsigned char* mask; bf16_t* out; int R, S, H, kv_div; float scale;
                 const int* rowmap;      // optional [R][S]: cache row that holds position s of hypothesis r (beam-search cache indirection)
                 const long long* last_pos;      // optional device scalar: only positions <= *last_pos exist (neither read nor attended)
                 const bf16_t* knew; const bf16_t* vnew; long ldq; };   // optional: this step's K / V rows [R][ldq] (fused qkv output); the kernel
                                                 // attends to them as position *last_pos and appends them to row r of the caches

#ifndef EVK_DEC_ATTN_UNR
#define EVK_DEC_ATTN_UNR 10
#endif
__global__ __launch_bounds__(256) void decode_attn_kernel(const DecAttP p) {
  // lane = (g, c): g = lane >> 3 picks one of 8 key rows per pass, c = lane & 7 one 16-byte chunk of the 128-byte head row, so
  // every load instruction of the wave reads 8 whole rows and all passes of a batch are independent (no load waits on a shuffle).
  // The KEY and the VALUE rows of UNR passes (80 keys) are requested together and folded into a running (max, sum, output) triple --
  // an online softmax --, so the 144 cross-attention keys cost two memory round trips and a self-attention over <= 80 positions one.
  // (Round 3 walked the keys in batches of 48, stored the scores in LDS, normalised, then walked the values: six dependent round trips,
  // most of the kernel's 13 us on a step that is latency bound.)
  __shared__ int sr[4][256];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + wv;
  if (w >= p.R * p.H) return;
  int r, h;
  if (p.kv_div > 1 && (p.kv_div & 3) == 0) {
    // hypotheses that share a K / V row (the beams of a sample in the cross attention): the four waves of a workgroup take four of them for
    // ONE head -- they request the same key / value lines at the same time, which the CU's L1 serves once (measured before: 75 MB through
    // L2 per launch at beam 4, 6 TB/s: bandwidth bound on re-reads)
    const int grp = blockIdx.x;                        // (row quad, head)
    h = grp % p.H;
    r = (grp / p.H) * 4 + wv;
  } else {
    r = w / p.H; h = w - r * p.H;
  }
  const int g = lane >> 3, c = lane & 7;
  const long HD = (long)p.H * 64;
  float qr[8];
  {
    const uint4 u = *reinterpret_cast<const uint4*>(p.q + (long)r * (p.ldq ? p.ldq : HD) + h * 64 + c * 8);
    qr[0] = lo_bf(u.x); qr[1] = hi_bf(u.x); qr[2] = lo_bf(u.y); qr[3] = hi_bf(u.y);
    qr[4] = lo_bf(u.z); qr[5] = hi_bf(u.z); qr[6] = lo_bf(u.w); qr[7] = hi_bf(u.w);
  }
  const long rk = r / p.kv_div;          // kv_div consecutive query rows (the beams of one sample) share one K / V / mask row
  const bf16_t* kb = p.k + h * 64 + c * 8;
  const bf16_t* vb = p.v + h * 64 + c * 8;
  const unsigned char* mk = p.mask ? p.mask + rk * p.S : nullptr;
  const int Seff = p.last_pos ? min(p.S, (int)*p.last_pos + 1) : p.S;
  const int passes = (Seff + 7) >> 3;
  // fused append: the new key / value of this hypothesis are read from the projection's output for position Seff - 1 and copied into
  // the caches for the later steps (nothing in this launch reads the copies)
  const int snew = p.knew ? Seff - 1 : -1;
  const bf16_t* kn = p.knew ? p.knew + (long)r * p.ldq + h * 64 + c * 8 : nullptr;
  const bf16_t* vn = p.vnew ? p.vnew + (long)r * p.ldq + h * 64 + c * 8 : nullptr;
  if (p.knew && g == 0) {
    const long dst = ((long)r * p.S + snew) * HD + h * 64 + c * 8;
    *reinterpret_cast<uint4*>(const_cast<bf16_t*>(p.k) + dst) = *reinterpret_cast<const uint4*>(kn);
    *reinterpret_cast<uint4*>(const_cast<bf16_t*>(p.v) + dst) = *reinterpret_cast<const uint4*>(vn);
  }
  // cache row of every position: r / kv_div, or -- beam search without moving the caches -- the row of the ancestor that
  // wrote position s (rowmap); staged in LDS so that no K / V load waits on an index load
  for (int s = lane; s < Seff; s += 64) sr[wv][s] = p.rowmap ? p.rowmap[(long)r * p.S + s] : (int)rk;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr int UNR = EVK_DEC_ATTN_UNR;
  float mrun = -INFINITY, lrun = 0.f;               // running max (wave-uniform) and this lane group's share of the running sum
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int it0 = 0; it0 < passes; it0 += UNR) {
    uint4 ku[UNR], vu[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
      const int s = (it0 + j) * 8 + g;
      const bool ok = s < Seff;
      const long off = ok ? ((long)sr[wv][s] * p.S + s) * HD : 0;
      ku[j] = ok ? *reinterpret_cast<const uint4*>(s == snew ? kn : kb + off) : make_uint4(0, 0, 0, 0);
      vu[j] = ok ? *reinterpret_cast<const uint4*>(s == snew ? vn : vb + off) : make_uint4(0, 0, 0, 0);
    }
    float sc[UNR];
    float bmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
      const int s = (it0 + j) * 8 + g;
      const uint4 u = ku[j];
      float a = lo_bf(u.x) * qr[0] + hi_bf(u.x) * qr[1] + lo_bf(u.y) * qr[2] + hi_bf(u.y) * qr[3] + lo_bf(u.z) * qr[4] + hi_bf(u.z) * qr[5] +
                lo_bf(u.w) * qr[6] + hi_bf(u.w) * qr[7];
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      a += __shfl_xor(a, 4, 64);
      sc[j] = (s < Seff && !(mk && !mk[s])) ? a * p.scale : -INFINITY;
      bmax = fmaxf(bmax, sc[j]);
    }
    bmax = fmaxf(bmax, __shfl_xor(bmax, 8, 64));
    bmax = fmaxf(bmax, __shfl_xor(bmax, 16, 64));
    bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
    const float mnew = fmaxf(mrun, bmax);
    const float resc = (mrun == -INFINITY) ? 0.f : __expf(mrun - mnew);          // (mnew == -inf only while every key so far is masked)
    lrun *= resc;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] *= resc;
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
      const float e = (sc[j] == -INFINITY || mnew == -INFINITY) ? 0.f : __expf(sc[j] - mnew);
      lrun += e;
      const uint4 u = vu[j];
      o[0] += e * lo_bf(u.x); o[1] += e * hi_bf(u.x); o[2] += e * lo_bf(u.y); o[3] += e * hi_bf(u.y);
      o[4] += e * lo_bf(u.z); o[5] += e * hi_bf(u.z); o[6] += e * lo_bf(u.w); o[7] += e * hi_bf(u.w);
    }
    mrun = mnew;
  }
  // every lane of a key group (8 lanes) added the same e to lrun; the groups hold disjoint keys
  lrun += __shfl_xor(lrun, 8, 64);
  lrun += __shfl_xor(lrun, 16, 64);
  lrun += __shfl_xor(lrun, 32, 64);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o[j] += __shfl_xor(o[j], 8, 64);
    o[j] += __shfl_xor(o[j], 16, 64);
    o[j] += __shfl_xor(o[j], 32, 64);
  }
  const float inv = lrun > 0.f ? 1.f / lrun : 0.f;
  if (g == 0) {
    *reinterpret_cast<uint4*>(p.out + (long)r * HD + h * 64 + c * 8) =
        make_uint4(pack2bf(o[0] * inv, o[1] * inv), pack2bf(o[2] * inv, o[3] * inv), pack2bf(o[4] * inv, o[5] * inv), pack2bf(o[6] * inv, o[7] * inv));
  }
}

// (A variant that requests every K and V row of a (row, head) pair in one batch and serves the beams of a sample from one copy of the rows was
// measured in round 4 and removed: 20.1 us per cross-attention launch against 13.0 for the kernel above, 13.9 against 13.0 for the self
// attention -- at 256 VGPRs it runs one wave per SIMD, and a lone wave pays the full latency of every cross-lane reduction of its 4 x 20
// passes, which costs more than the five round trips it saves.)
int launch_decode_attn(const DecAttP& p, hipStream_t s) {
  hipLaunchKernelGGL(decode_attn_kernel, dim3((int)cdiv((long)p.R * p.H, 4)), dim3(256), 0, s, p);
  return evk_check_launch("decode_attention");
}

}  // namespace

extern "C" {

int evk_layernorm_fwd(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta,
                      const void* dgam, const void* dbet, int d_dtype, float* mean, float* rstd,
                      int64_t rows, int32_t D, int32_t mode, float eps, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && y && gamma && beta && rows > 0, "layernorm_fwd: null/empty");
  EVK_REQUIRE(D % 8 == 0 && D >= 8 && D <= 64 * 8 * MAXC, "layernorm_fwd: D=%d must be a multiple of 8 and <= 2048", D);
  EVK_REQUIRE((dgam == nullptr) == (dbet == nullptr) && (mode == 0 || mode == 1), "layernorm_fwd: bad mode/deltas");
  LnP p{x, y, gamma, beta, dgam, dbet, mean, rstd, rows, D, mode, eps, x_dtype == EVK_F32, y_dtype == EVK_F32, d_dtype == EVK_F32};
  ProfScope ps(EVK_FAM_NORM, s);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(row_blocks(rows)), dim3(256), 0, s, p);
  return evk_check_launch("ln_fwd");
}

int evk_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const void* dgam, int d_dtype,
                      const float* mean, const float* rstd, void* dx, int dx_dtype, float* dgamma, float* dbeta,
                      void* ddgam, void* ddbet, int64_t rows, int32_t D, int32_t mode, float eps, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dy && x && gamma && mean && rstd && dx && rows > 0, "layernorm_bwd: null/empty");
  EVK_REQUIRE(D % 8 == 0 && D >= 8 && D <= 64 * 8 * MAXC, "layernorm_bwd: D=%d unsupported", D);
  EVK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr) && (ddgam == nullptr) == (ddbet == nullptr), "layernorm_bwd: grads must come in pairs");
  LnBP p{dy, x, gamma, dgam, mean, rstd, dx, dgamma, dbeta, ddgam, ddbet, rows, D, mode, eps,
         dy_dtype == EVK_F32, x_dtype == EVK_F32, dx_dtype == EVK_F32, d_dtype == EVK_F32};
  int blocks = row_blocks(rows);
  if (blocks > 512) blocks = 512;
  ProfScope ps(EVK_FAM_NORM, s);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), 0, s, p);
  return evk_check_launch("ln_bwd");
}

int evk_softmax_fwd(const float* scores, void* p_out, void* pdrop_out, int p_dtype, const unsigned char* mask, int64_t mask_batch_stride,
                    int32_t mask_q_stride, int32_t causal, int64_t batch, int32_t heads, int32_t Tq, int32_t S,
                    int32_t ld_in, int32_t ld_out, float p_drop, uint64_t seed, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(scores && p_out && batch > 0 && heads > 0 && Tq > 0 && S > 0, "softmax_fwd: null/empty");
  EVK_REQUIRE(S <= ld_in && S <= ld_out && ld_out <= 64 * SMC, "softmax_fwd: S=%d ld_out=%d (max %d)", S, ld_out, 64 * SMC);
  EVK_REQUIRE(p_drop == 0.f || pdrop_out, "softmax_fwd: dropout needs a second output");
  SmP p{scores, p_out, pdrop_out ? pdrop_out : p_out, mask, batch * heads * Tq, Tq, S, ld_in, ld_out, heads, mask_batch_stride,
        mask_q_stride, causal, p_drop, seed, p_dtype == EVK_F32, evk_seed_epoch_ptr()};
  ProfScope ps(EVK_FAM_NORM, s);
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(row_blocks(p.rows)), dim3(256), 0, s, p);
  return evk_check_launch("softmax_fwd");
}

int evk_softmax_bwd(const void* dp, int dp_dtype, int32_t ld_dp, const void* probs, void* ds, int p_dtype, int64_t rows, int32_t S, int32_t ld,
                    float alpha, float p_drop, uint64_t seed, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dp && probs && ds && rows > 0 && S > 0 && S <= ld && ld <= 64 * SMC && S <= ld_dp, "softmax_bwd: bad args");
  SmBP p{dp, probs, ds, rows, S, ld_dp, ld, dp_dtype == EVK_F32, p_drop, seed, alpha, p_dtype == EVK_F32, evk_seed_epoch_ptr()};
  ProfScope ps(EVK_FAM_NORM, s);
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(row_blocks(rows)), dim3(256), 0, s, p);
  return evk_check_launch("softmax_bwd");
}

int evk_log_softmax_nll_fwd(const float* logits, float* logp, float* lse, const int64_t* target, const float* wmask, float* acc2,
                            int64_t rows, int32_t V, int32_t ld, int32_t ld_out, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(logits && rows > 0 && V > 0 && V <= ld, "log_softmax: bad args");
  EVK_REQUIRE(!target || (wmask && acc2), "log_softmax: NLL needs wmask and acc2");
  LsmP p{logits, logp, (const long long*)target, wmask, acc2, lse, rows, V, ld, ld_out, nullptr};
  ProfScope ps(EVK_FAM_NORM, s);
  if (V <= 24 * 64) hipLaunchKernelGGL(logsoftmax_kernel<true>, dim3(row_blocks(rows)), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(logsoftmax_kernel<false>, dim3(row_blocks(rows)), dim3(256), 0, s, p);
  return evk_check_launch("log_softmax");
}

int evk_log_softmax_nll_rows(const float* logits, float* lse, const int64_t* target, const float* wmask, float* row_nll, int64_t rows,
                             int32_t V, int32_t ld, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(logits && target && wmask && row_nll && rows > 0 && V > 0 && V <= ld, "log_softmax_nll_rows: bad args");
  LsmP p{logits, nullptr, (const long long*)target, wmask, nullptr, lse, rows, V, ld, ld, row_nll};
  ProfScope ps(EVK_FAM_NORM, s);
  if (V <= 24 * 64) hipLaunchKernelGGL(logsoftmax_kernel<true>, dim3(row_blocks(rows)), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(logsoftmax_kernel<false>, dim3(row_blocks(rows)), dim3(256), 0, s, p);
  return evk_check_launch("log_softmax_nll_rows");
}

int evk_nll_bwd(const float* logits, const float* lse, const int64_t* target, const float* wmask, const float* gscale,
                void* dlogits, int64_t rows, int32_t V, int32_t ld, int32_t ld_out, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(logits && lse && target && wmask && gscale && dlogits && rows > 0 && V <= ld && V <= ld_out, "nll_bwd: bad args");
  NllBP p{logits, lse, (const long long*)target, wmask, gscale, (bf16_t*)dlogits, rows, V, ld, ld_out};
  ProfScope ps(EVK_FAM_NORM, s);
  hipLaunchKernelGGL(nll_bwd_kernel, dim3(row_blocks(rows)), dim3(256), 0, s, p);
  return evk_check_launch("nll_bwd");
}

int evk_l2norm_fwd(const float* x, float* y, float* nrm, int64_t rows, int32_t D, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && y && nrm && rows > 0 && D > 0, "l2norm_fwd: bad args");
  ProfScope ps(EVK_FAM_NORM, s);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(row_blocks(rows)), dim3(256), 0, s, x, y, nrm, (long)rows, D);
  return evk_check_launch("l2norm_fwd");
}

int evk_l2norm_bwd(const float* dy, const float* y, const float* nrm, float* dx, int64_t rows, int32_t D, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dy && y && nrm && dx && rows > 0 && D > 0, "l2norm_bwd: bad args");
  ProfScope ps(EVK_FAM_NORM, s);
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(row_blocks(rows)), dim3(256), 0, s, dy, y, nrm, dx, (long)rows, D);
  return evk_check_launch("l2norm_bwd");
}

int evk_topk_rows(const float* x, float* vals, int64_t* idx, int64_t rows, int32_t n, int32_t k, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && vals && idx && rows > 0 && n > 0 && k >= 1 && k <= TOPK_MAX && k <= n, "topk_rows: bad args (k <= %d)", TOPK_MAX);
  ProfScope ps(EVK_FAM_NORM, s);
  hipLaunchKernelGGL(topk_rows_kernel, dim3((int)rows), dim3(256), 0, s, x, vals, (long long*)idx, n, k);
  return evk_check_launch("topk_rows");
}

int evk_softce(const float* z, const float* t, float* loss_acc, float* dz, const float* gscale, int64_t rows, int32_t Cn,
               float scale, int32_t diag_mask, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(z && t && rows > 0 && Cn > 0 && (loss_acc || dz) && (!dz || gscale), "softce: bad args");
  SceP p{z, t, loss_acc, dz, gscale, rows, Cn, scale, diag_mask};
  ProfScope ps(EVK_FAM_NORM, s);
  hipLaunchKernelGGL(softce_kernel, dim3(row_blocks(rows)), dim3(256), 0, s, p);
  return evk_check_launch("softce");
}

int evk_decode_attention(const void* q, const void* k, const void* v, const unsigned char* mask, void* out, int32_t R, int32_t S,
                         int32_t heads, int32_t head_dim, int32_t kv_div, float scale, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(q && k && v && out && R > 0 && S > 0 && heads > 0, "decode_attention: null/empty");
  EVK_REQUIRE(head_dim == 64 && S <= 256, "decode_attention: head_dim must be 64 and S <= 256 (got %d, %d)", head_dim, S);
  EVK_REQUIRE(kv_div >= 1 && R % kv_div == 0, "decode_attention: R=%d must be a multiple of kv_div=%d", R, kv_div);
  DecAttP p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, mask, (bf16_t*)out, R, S, heads, kv_div, scale, nullptr, nullptr, nullptr, nullptr, 0};
  ProfScope ps(EVK_FAM_NORM, s);
  return launch_decode_attn(p, s);
}

int evk_decode_attention_indirect(const void* q, const void* k, const void* v, const unsigned char* mask, const int32_t* rowmap,
                                  const int64_t* last_pos, void* out, int32_t R, int32_t S, int32_t heads, int32_t head_dim, float scale,
                                  evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(q && k && v && out && rowmap && R > 0 && S > 0 && heads > 0, "decode_attention_indirect: null/empty");
  EVK_REQUIRE(head_dim == 64 && S <= 256, "decode_attention_indirect: head_dim must be 64 and S <= 256 (got %d, %d)", head_dim, S);
  DecAttP p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, mask, (bf16_t*)out, R, S, heads, 1, scale, rowmap,
            reinterpret_cast<const long long*>(last_pos), nullptr, nullptr, 0};
  ProfScope ps(EVK_FAM_NORM, s);
  return launch_decode_attn(p, s);
}

int evk_decode_attention_qkv(const void* qkv, int64_t ldq, void* k_cache, void* v_cache, const int32_t* rowmap, const int64_t* last_pos, void* out,
                             int32_t R, int32_t S, int32_t heads, int32_t head_dim, float scale, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(qkv && k_cache && v_cache && out && rowmap && last_pos && R > 0 && S > 0 && heads > 0, "decode_attention_qkv: null/empty");
  EVK_REQUIRE(head_dim == 64 && S <= 256, "decode_attention_qkv: head_dim must be 64 and S <= 256 (got %d, %d)", head_dim, S);
  const int64_t HD = (int64_t)heads * 64;
  EVK_REQUIRE(ldq >= 3 * HD && ldq % 8 == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0, "decode_attention_qkv: qkv rows must hold q | k | v and be 16-byte aligned");
  const bf16_t* base = (const bf16_t*)qkv;
  DecAttP p{base, (const bf16_t*)k_cache, (const bf16_t*)v_cache, nullptr, (bf16_t*)out, R, S, heads, 1, scale, rowmap,
            reinterpret_cast<const long long*>(last_pos), base + HD, base + 2 * HD, (long)ldq};
  ProfScope ps(EVK_FAM_NORM, s);
  return launch_decode_attn(p, s);
}

}  // extern "C"
