// eltwise.hip -- memory-bound elementwise / gather / scatter kernels:
//   casts (f32 master weights -> bf16 GEMM operands), activation forward/backward (F.relu / gelu / tanh / sigmoid),
//   dropout (nn.Dropout, stateless hash RNG so the backward recomputes the mask), embedding gather + scatter-add
//   (nn.Embedding: encoder_decoder.py:217-224, HF BertEmbeddings), bias-gradient column sums, the relational-memory
//   gate (encoder_decoder.py:282-289), fused value-clip + RAdam / Adam(amsgrad) (optimizers.py:17-53,
//   trainer_v0401.py:262,434).
#include "common.h"

namespace {

inline int ew_blocks(long work) { long b = cdiv(work, 256); return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

__device__ __forceinline__ uint32_t hash32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}

__device__ __forceinline__ float ldx(const void* p, int f32, long i) {
  return f32 ? reinterpret_cast<const float*>(p)[i] : bf2f(reinterpret_cast<const bf16_t*>(p)[i]);
}
__device__ __forceinline__ void stx(void* p, int f32, long i, float v) {
  if (f32) reinterpret_cast<float*>(p)[i] = v; else reinterpret_cast<bf16_t*>(p)[i] = f2bf(v);
}

__global__ __launch_bounds__(256) void cast_kernel(const void* __restrict__ src, int s32, void* __restrict__ dst, int d32, long n) {
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float v[4];
    if (s32) { const float4 t = reinterpret_cast<const float4*>(src)[i]; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else { const uint2 t = reinterpret_cast<const uint2*>(src)[i]; v[0] = lo_bf(t.x); v[1] = hi_bf(t.x); v[2] = lo_bf(t.y); v[3] = hi_bf(t.y); }
    if (d32) reinterpret_cast<float4*>(dst)[i] = make_float4(v[0], v[1], v[2], v[3]);
    else reinterpret_cast<uint2*>(dst)[i] = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const long i = (n4 << 2) + threadIdx.x; stx(dst, d32, i, ldx(src, s32, i)); }
}

// y = act(x)    /   dx = dy * act'(ref)   (ref = output for relu/tanh/sigmoid, pre-activation for gelu)
__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = f2bf(act_apply(bf2f(x[i]), act));
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ ref,
                                                      bf16_t* __restrict__ dx, long n, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = bf2f(dy[i]), r = bf2f(ref[i]);
    float d;
    switch (act) {
      case EVK_ACT_RELU: d = r > 0.f ? 1.f : 0.f; break;
      case EVK_ACT_TANH: d = 1.f - r * r; break;
      case EVK_ACT_SIGMOID: d = r * (1.f - r); break;
      case EVK_ACT_GELU: d = 0.5f * (1.f + erff(r * 0.70710678118654752f)) + r * 0.3989422804014327f * __expf(-0.5f * r * r); break;
      case EVK_ACT_GELU_NEW: {
        const float u = 0.7978845608028654f * (r + 0.044715f * r * r * r), th = tanhf(u);
        d = 0.5f * (1.f + th) + 0.5f * r * (1.f - th * th) * 0.7978845608028654f * (1.f + 3.f * 0.044715f * r * r);
      } break;
      default: d = 1.f;
    }
    dx[i] = f2bf(g * d);
  }
}

__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ resid,
                                                      bf16_t* __restrict__ y, long n, float p, unsigned long long seed0,
                                                      const unsigned long long* __restrict__ epoch) {
  const unsigned long long seed = evk_mix_seed(seed0, epoch);
  const float sc = 1.f / (1.f - p);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const bool keep = (hash32(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)i) >> 8) * (1.f / 16777216.f) >= p;
    float v = keep ? bf2f(x[i]) * sc : 0.f;
    if (resid) v += bf2f(resid[i]);
    y[i] = f2bf(v);
  }
}

// out[r][:] = table[ids[r]][:] * scale + (pos ? pos[r % L][:] : 0) + (extra ? extra[:] : 0)
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const float* __restrict__ table, const long long* __restrict__ ids,
                                                            const float* __restrict__ pos, const float* __restrict__ extra,
                                                            void* __restrict__ out, int out_f32, long rows, int D, int L, float scale, long long table_rows,
                                                            const long long* __restrict__ pos0_dev, float* __restrict__ out2) {
  const long total = rows * D;
  const long pos0 = pos0_dev ? pos0_dev[0] : 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    const int c = (int)(i - r * D);
    const long long id = ids[r];
    float v = (id >= 0 && id < table_rows) ? table[id * D + c] * scale : 0.f;        // an id outside the table must not become a wild read
    if (pos) v += pos[((r % L) + pos0) * D + c];
    if (extra) v += extra[c];
    stx(out, out_f32, i, v);
    if (out2) out2[i] = v;
  }
}
// dtable[ids[r]][:] += dout[r][:] * scale  (rows with ids == padding_idx are skipped)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const void* __restrict__ dout, int d_f32, const long long* __restrict__ ids,
                                                            float* __restrict__ dtable, long rows, int D, float scale, long long padding_idx, long long table_rows) {
  const long total = rows * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    const long long id = ids[r];
    if (id == padding_idx || id < 0 || id >= table_rows) continue;     // (an id outside the table must not become a wild atomic)
    unsafeAtomicAdd(dtable + id * D + (i - r * D), ldx(dout, d_f32, i) * scale);
  }
}

// out[c] += sum_r x[r*ld + c], c < N  (bias gradients). block = 128 column pairs x 2 row lanes
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, long M, int N, long ld,
                                                     long rows_per_block) {
  const int cp = blockIdx.x * 128 + (threadIdx.x & 127);
  const int c = cp * 2;
  const long r0 = blockIdx.y * rows_per_block + (threadIdx.x >> 7);
  const long r1 = min(M, (long)(blockIdx.y + 1) * rows_per_block);
  float a = 0.f, b = 0.f;
  if (c < N) {
    if (c + 1 < N || ld > N) {
      for (long r = r0; r < r1; r += 2) {
        const uint32_t t = *reinterpret_cast<const uint32_t*>(x + r * ld + c);
        a += lo_bf(t); b += hi_bf(t);
      }
    } else {
      for (long r = r0; r < r1; r += 2) a += bf2f(x[r * ld + c]);
    }
  }
  __shared__ float red[2][256];
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x < 128 && c < N) {
    unsafeAtomicAdd(out + c, red[0][threadIdx.x] + red[0][threadIdx.x + 128]);
    if (c + 1 < N) unsafeAtomicAdd(out + c + 1, red[1][threadIdx.x] + red[1][threadIdx.x + 128]);
  }
}

// same sums with 16-byte loads: thread = 8 columns x one of 8 row lanes, four rows in flight per thread (ld % 8 == 0, x 16-byte aligned)
__global__ __launch_bounds__(256) void colsum8_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, long M, int N, long ld,
                                                      long rows_per_block) {
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cg) * 8;
  const long r0 = blockIdx.y * rows_per_block;
  const long r1 = min(M, r0 + rows_per_block);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < N) {
    const bf16_t* px = x + c;
    long r = r0 + rl;
    for (; r + 24 < r1; r += 32) {
      uint4 u[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) u[q] = *reinterpret_cast<const uint4*>(px + (r + 8 * q) * ld);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[0] += lo_bf(u[q].x); a[1] += hi_bf(u[q].x); a[2] += lo_bf(u[q].y); a[3] += hi_bf(u[q].y);
        a[4] += lo_bf(u[q].z); a[5] += hi_bf(u[q].z); a[6] += lo_bf(u[q].w); a[7] += hi_bf(u[q].w);
      }
    }
    for (; r < r1; r += 8) {
      const uint4 u = *reinterpret_cast<const uint4*>(px + r * ld);
      a[0] += lo_bf(u.x); a[1] += hi_bf(u.x); a[2] += lo_bf(u.y); a[3] += hi_bf(u.y);
      a[4] += lo_bf(u.z); a[5] += hi_bf(u.z); a[6] += lo_bf(u.w); a[7] += hi_bf(u.w);
    }
  }
  __shared__ float red[8][32][9];
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cg][j] = a[j];
  __syncthreads();
  const int j = threadIdx.x & 7, g2 = threadIdx.x >> 3;       // 32 column groups x 8 columns
  const int cc = (blockIdx.x * 32 + g2) * 8 + j;
  if (cc < N) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][g2][j];
    unsafeAtomicAdd(out + cc, t);
  }
}

// fused step over a flat parameter buffer: g = clamp(g, -clip, clip) ; RAdam (torch.optim.RAdam, decoupled=False)
// or Adam with amsgrad + L2 weight decay (optimizers.py:19-21 "AdamW" == optim.Adam(amsgrad=True)); also refreshes
// the bf16 shadow used as GEMM operand.  Hyper-parameters that depend on the step count are precomputed on the host.
struct OptP {
  float* p; const float* g; float* m; float* v; float* vmax; bf16_t* shadow; long n;
  float lr, beta1, beta2, eps, wd, clip, bc1, bc2_sqrt, rect; int kind; int use_rect;
  float gscale;      // gradients arrive multiplied by 1/gscale (static loss scale of the fp16-storage build); 1 = off
};
__global__ __launch_bounds__(256) void optim_kernel(const OptP o) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < o.n; i += (long)gridDim.x * blockDim.x) {
    float g = o.g[i];
    if (o.gscale != 1.f) {
      g *= o.gscale;
      if (!isfinite(g)) continue;      // an overflowed fp16 gradient: leave this element's state untouched for the step
    }
    if (o.clip > 0.f) g = fminf(fmaxf(g, -o.clip), o.clip);
    float w = o.p[i];
    if (o.wd != 0.f) g += o.wd * w;
    const float m = o.beta1 * o.m[i] + (1.f - o.beta1) * g;
    const float v = o.beta2 * o.v[i] + (1.f - o.beta2) * g * g;
    o.m[i] = m; o.v[i] = v;
    if (o.kind == 0) {          // RAdam
      const float mh = m / o.bc1;
      if (o.use_rect) w -= o.lr * mh * o.rect * o.bc2_sqrt / (sqrtf(v) + o.eps);
      else w -= o.lr * mh;
    } else {                    // Adam (+amsgrad)
      float vv = v;
      if (o.vmax) { vv = fmaxf(o.vmax[i], v); o.vmax[i] = vv; }
      w -= (o.lr / o.bc1) * m / (sqrtf(vv) / o.bc2_sqrt + o.eps);
    }
    o.p[i] = w;
    if (o.shadow) o.shadow[i] = f2bf(w);
  }
}

// ---- dynamic loss scaling (fp16 storage): device-resident state, no host read-back ------------------------------------
// state[0] = loss scale, [1] = consecutive good steps, [2] = non-finite gradient seen this step, [3] = skipped steps (total)
__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const float* __restrict__ g, long n, float* __restrict__ state) {
  const long n4 = n >> 2;
  bool bad = false;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const uint4 t = reinterpret_cast<const uint4*>(g)[i];          // exponent all ones <=> inf / NaN
    bad |= ((t.x & 0x7f800000u) == 0x7f800000u) | ((t.y & 0x7f800000u) == 0x7f800000u) | ((t.z & 0x7f800000u) == 0x7f800000u) |
           ((t.w & 0x7f800000u) == 0x7f800000u);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= (__float_as_uint(g[(n4 << 2) + threadIdx.x]) & 0x7f800000u) == 0x7f800000u;
  if (__any(bad) && (threadIdx.x & 63) == 0) state[2] = 1.f;
}

// forward-overflow guard of the fp16-storage build: count[0] += number of 8-element groups of a 16-bit activation tensor that hold an inf / NaN
// (exponent all ones: 0x7c00 in fp16, 0x7f80 in bf16).  The loss scale protects the backward only; a forward activation beyond +-65504
// (e.g. an eval-mode network whose BN running statistics are untrained) is not repairable and must be reported.
__global__ __launch_bounds__(256) void act_nonfinite_kernel(const bf16_t* __restrict__ x, long n, int* __restrict__ count) {
#ifdef EVK_STORE_BF16
  const unsigned m = 0x7f807f80u;
#else
  const unsigned m = 0x7c007c00u;
#endif
  const unsigned lo = m & 0xffffu, hi = m & 0xffff0000u;
  const long n8 = n >> 3;
  bool bad = false;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const uint4 t = reinterpret_cast<const uint4*>(x)[i];
    bad |= ((t.x & lo) == lo) | ((t.x & hi) == hi) | ((t.y & lo) == lo) | ((t.y & hi) == hi) | ((t.z & lo) == lo) | ((t.z & hi) == hi) |
           ((t.w & lo) == lo) | ((t.w & hi) == hi);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) bad |= ((unsigned)reinterpret_cast<const unsigned short*>(x)[(n8 << 3) + threadIdx.x] & lo) == lo;
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicAdd(count, 1);
}

__global__ void loss_scale_update_kernel(float* __restrict__ state, float growth, float backoff, int interval, float lo, float hi) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (state[2] != 0.f) {
    state[0] = fmaxf(state[0] * backoff, lo);
    state[1] = 0.f;
    state[3] += 1.f;
  } else {
    const float good = state[1] + 1.f;
    if (good >= (float)interval) { state[0] = fminf(state[0] * growth, hi); state[1] = 0.f; }
    else state[1] = good;
  }
  state[2] = 0.f;
}

// ---- gradients of a model that no FusedOptimizer owns (the reference's own step: backward -> clip_grad_value_ -> torch.optim) --------
// one table entry per <= 64 Ki-element piece of a parameter's f32 gradient tensor; one workgroup walks pieces with a grid stride
struct GradChunk { float* p; int n; int pad; };
template <int MODE>        // 0: scan for inf / NaN -> state[2];  1: g *= 1 / scale, or g = 0 on an overflowed step;  2: g *= scale
__global__ __launch_bounds__(256) void grads_multi_kernel(const GradChunk* __restrict__ tab, int n_chunks, float* __restrict__ state) {
  float mul = 1.f;
  bool wipe = false;
  if (MODE == 1) { wipe = state[2] != 0.f; mul = 1.f / state[0]; }
  if (MODE == 2) mul = state[0];
  bool bad = false;
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    float* g = tab[c].p;
    const int n = tab[c].n;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
      const int n4 = n >> 2;
      for (int i = threadIdx.x; i < n4; i += 256) {
        float4 t = reinterpret_cast<float4*>(g)[i];
        if (MODE == 0) {
          bad |= ((__float_as_uint(t.x) & 0x7f800000u) == 0x7f800000u) | ((__float_as_uint(t.y) & 0x7f800000u) == 0x7f800000u) |
                 ((__float_as_uint(t.z) & 0x7f800000u) == 0x7f800000u) | ((__float_as_uint(t.w) & 0x7f800000u) == 0x7f800000u);
        } else {
          t = wipe ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(t.x * mul, t.y * mul, t.z * mul, t.w * mul);
          reinterpret_cast<float4*>(g)[i] = t;
        }
      }
      for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
        if (MODE == 0) bad |= (__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u;
        else g[i] = wipe ? 0.f : g[i] * mul;
      }
    } else {
      for (int i = threadIdx.x; i < n; i += 256) {
        if (MODE == 0) bad |= (__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u;
        else g[i] = wipe ? 0.f : g[i] * mul;
      }
    }
  }
  if (MODE == 0 && __any(bad) && (threadIdx.x & 63) == 0) state[2] = 1.f;
}

__global__ void optim_bump_kernel(int* __restrict__ steps, int count, const float* __restrict__ state) {
  if (state && state[2] != 0.f) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) steps[i] += 1;
}

// optim_kernel with the step count, the loss scale and the skip decision read from device memory
struct OptDyn { OptP o; const int* step; const float* state; float inv_world; int zero_g; };
__global__ __launch_bounds__(256) void optim_dyn_kernel(const OptDyn d) {
  if (d.state && d.state[2] != 0.f) {                  // global overflow skip: no element of any parameter moves
    if (d.zero_g)                                      // ... but the (non-finite) gradients are consumed like on a taken step
      for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < d.o.n; i += (long)gridDim.x * blockDim.x) const_cast<float*>(d.o.g)[i] = 0.f;
    return;
  }
  __shared__ float hp[4];
  __shared__ int rect_on;
  if (threadIdx.x == 0) {
    const double step = (double)(d.step[0] + 1), beta1 = d.o.beta1, beta2 = d.o.beta2;
    const double b1t = pow(beta1, step), b2t = pow(beta2, step);
    hp[0] = (float)(1.0 - b1t);
    hp[1] = (float)sqrt(1.0 - b2t);
    hp[2] = 0.f;
    rect_on = 0;
    if (d.o.kind == 0) {
      const double rho_inf = 2.0 / (1.0 - beta2) - 1.0;
      const double rho_t = rho_inf - 2.0 * step * b2t / (1.0 - b2t);
      rect_on = rho_t > 5.0;
      if (rect_on) hp[2] = (float)sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
    }
    hp[3] = d.inv_world / (d.state ? d.state[0] : 1.f);
  }
  __syncthreads();
  const OptP& o = d.o;
  const float bc1 = hp[0], bc2_sqrt = hp[1], rect = hp[2], gscale = hp[3];
  const int use_rect = rect_on;
  // one element of the update (same arithmetic, element by element, as optim_kernel)
  auto upd = [&](float g, float& w, float& m, float& v, float& vm) {
    g *= gscale;
    if (o.clip > 0.f) g = fminf(fmaxf(g, -o.clip), o.clip);
    if (o.wd != 0.f) g += o.wd * w;
    m = o.beta1 * m + (1.f - o.beta1) * g;
    v = o.beta2 * v + (1.f - o.beta2) * g * g;
    if (o.kind == 0) {
      const float mh = m / bc1;
      if (use_rect) w -= o.lr * mh * rect * bc2_sqrt / (sqrtf(v) + o.eps);
      else w -= o.lr * mh;
    } else {
      float vv = v;
      if (o.vmax) { vv = fmaxf(vm, v); vm = vv; }
      w -= (o.lr / bc1) * m / (sqrtf(vv) / bc2_sqrt + o.eps);
    }
  };
  const long gtid = blockIdx.x * (long)blockDim.x + threadIdx.x, gstride = (long)gridDim.x * blockDim.x;
  // 16-byte lanes over the aligned body.  The buffers of one group share their element offset (optim.py: one flat layout), so a
  // scalar head of up to 3 elements brings all of them to a 16-byte boundary (8 bytes for the 2-byte shadow) at once; views that
  // do not line up that way take the scalar loop throughout.
  const long head = (long)((16 - (reinterpret_cast<uintptr_t>(o.p) & 15)) & 15) >> 2;
  const bool al = ((reinterpret_cast<uintptr_t>(o.p) & 3) == 0) && head <= o.n &&
                  (((reinterpret_cast<uintptr_t>(o.g + head) | reinterpret_cast<uintptr_t>(o.m + head) | reinterpret_cast<uintptr_t>(o.v + head) |
                     (o.vmax ? reinterpret_cast<uintptr_t>(o.vmax + head) : 0)) & 15) == 0) &&
                  (!o.shadow || (reinterpret_cast<uintptr_t>(o.shadow + head) & 7) == 0);
  const long n4 = al ? (o.n - head) >> 2 : 0;
  auto scalar = [&](long i) {
    float w = o.p[i], m = o.m[i], v = o.v[i], vm = o.vmax ? o.vmax[i] : 0.f;
    const float g = o.g[i];
    if (d.zero_g) const_cast<float*>(o.g)[i] = 0.f;     // the step consumes the gradient: no separate zero_grad pass over the buffer
    upd(g, w, m, v, vm);
    o.p[i] = w; o.m[i] = m; o.v[i] = v;
    if (o.vmax) o.vmax[i] = vm;
    if (o.shadow) o.shadow[i] = f2bf(w);
  };
  if (!al) {
    for (long i = gtid; i < o.n; i += gstride) scalar(i);
    return;
  }
  if (gtid < head) scalar(gtid);
  for (long i = head + (n4 << 2) + gtid; i < o.n; i += gstride) scalar(i);
  const OptP q{o.p + head, o.g + head, o.m + head, o.v + head, o.vmax ? o.vmax + head : nullptr, o.shadow ? o.shadow + head : nullptr};
  for (long i = gtid; i < n4; i += gstride) {
    float4 g4 = reinterpret_cast<const float4*>(q.g)[i], w4 = reinterpret_cast<float4*>(q.p)[i];
    float4 m4 = reinterpret_cast<float4*>(q.m)[i], v4 = reinterpret_cast<float4*>(q.v)[i];
    float4 x4 = o.vmax ? reinterpret_cast<float4*>(q.vmax)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.zero_g) reinterpret_cast<float4*>(const_cast<float*>(q.g))[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    upd(g4.x, w4.x, m4.x, v4.x, x4.x); upd(g4.y, w4.y, m4.y, v4.y, x4.y);
    upd(g4.z, w4.z, m4.z, v4.z, x4.z); upd(g4.w, w4.w, m4.w, v4.w, x4.w);
    reinterpret_cast<float4*>(q.p)[i] = w4;
    reinterpret_cast<float4*>(q.m)[i] = m4;
    reinterpret_cast<float4*>(q.v)[i] = v4;
    if (q.vmax) reinterpret_cast<float4*>(q.vmax)[i] = x4;
    if (q.shadow) {
      uint2 pk;
      pk.x = pack2bf(w4.x, w4.y);
      pk.y = pack2bf(w4.z, w4.w);
      reinterpret_cast<uint2*>(q.shadow)[i] = pk;
    }
  }
}

// ---- one optimizer step of a whole parameter GROUP (flat buffers of optim.py) in two launches -------------------------------------
// Every per-step quantity is read from device memory: hyper-parameters (hp[0..5] = lr, beta1, beta2, eps, weight decay, clip value:
// an lr scheduler or load_state_dict changes the buffer, not a captured kernel argument), per-parameter step counts (bias
// corrections / rectification are evaluated PER PARAMETER, torch.optim semantics: a parameter that sat out some steps keeps its own
// count), the loss scale and the overflow verdict.
// coef[i] = {1 - beta1^t, sqrt(1 - beta2^t), rectification (RAdam; < 0: not rectified yet), took part in this step}
__global__ void optim_group_coef_kernel(const float* __restrict__ hp, int kind, int n_params, int* __restrict__ steps,
                                        const unsigned char* __restrict__ touched, float4* __restrict__ coef, const float* __restrict__ state) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_params) return;
  const bool skip = state && state[2] != 0.f;
  if (!touched[i]) { coef[i] = make_float4(1.f, 1.f, -1.f, 0.f); return; }
  if (skip) { coef[i] = make_float4(1.f, 1.f, -1.f, 1.f); return; }          // the main kernel only consumes the gradient
  const double step = (double)(steps[i] + 1), beta1 = hp[1], beta2 = hp[2];
  steps[i] += 1;
  const double b1t = pow(beta1, step), b2t = pow(beta2, step);
  float rect = -1.f;
  if (kind == 0) {
    const double rho_inf = 2.0 / (1.0 - beta2) - 1.0;
    const double rho_t = rho_inf - 2.0 * step * b2t / (1.0 - b2t);
    if (rho_t > 5.0) rect = (float)sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
  }
  coef[i] = make_float4((float)(1.0 - b1t), (float)sqrt(1.0 - b2t), rect, 1.f);
}

struct OptGroup {
  float* p; float* g; float* m; float* v; float* vmax; bf16_t* shadow; long n;
  const float* hp; const long* offs; int n_params; const float4* coef; const float* state; float inv_world; int kind; int zero_g;
};
constexpr int OPT_CHUNK = 8192;        // elements per workgroup pass: 256 threads x 8 lanes of 4
__global__ __launch_bounds__(256) void optim_group_kernel(const OptGroup d) {
  const bool skip = d.state && d.state[2] != 0.f;
  const float lr = d.hp[0], beta1 = d.hp[1], beta2 = d.hp[2], eps = d.hp[3], wd = d.hp[4], clip = d.hp[5];
  const float gscale = d.inv_world / (d.state ? d.state[0] : 1.f);
  __shared__ int range[2];
  const long n_chunks = (d.n + OPT_CHUNK - 1) / OPT_CHUNK;
  for (long c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const long e0 = c * OPT_CHUNK, e1 = min(d.n, e0 + OPT_CHUNK);
    __syncthreads();
    if (threadIdx.x < 2) {           // parameter index of the chunk's first / last element: largest i with offs[i] <= e
      const long e = threadIdx.x == 0 ? e0 : e1 - 1;
      int lo = 0, hi = d.n_params - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (d.offs[mid] <= e) lo = mid; else hi = mid - 1; }
      range[threadIdx.x] = lo;
    }
    __syncthreads();
    const int p_lo = range[0], p_hi = range[1];
    for (long e = e0 + 4 * threadIdx.x; e < e1; e += 4 * 256) {      // parameter starts are multiples of 8 elements: a lane has one owner
      int lo = p_lo, hi = p_hi;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (d.offs[mid] <= e) lo = mid; else hi = mid - 1; }
      const float4 cf = d.coef[lo];
      if (cf.w == 0.f) continue;                                     // parameter without a gradient this step: untouched (torch.optim)
      float4 g4 = *reinterpret_cast<const float4*>(d.g + e);
      if (d.zero_g) *reinterpret_cast<float4*>(d.g + e) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (skip) continue;
      float4 w4 = *reinterpret_cast<float4*>(d.p + e), m4 = *reinterpret_cast<float4*>(d.m + e), v4 = *reinterpret_cast<float4*>(d.v + e);
      float4 x4 = d.vmax ? *reinterpret_cast<float4*>(d.vmax + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float bc1 = cf.x, bc2_sqrt = cf.y, rect = cf.z;
      auto upd = [&](float g, float& w, float& m, float& v, float& vm) {
        g *= gscale;
        if (clip > 0.f) g = fminf(fmaxf(g, -clip), clip);
        if (wd != 0.f) g += wd * w;
        m = beta1 * m + (1.f - beta1) * g;
        v = beta2 * v + (1.f - beta2) * g * g;
        if (d.kind == 0) {
          const float mh = m / bc1;
          if (rect >= 0.f) w -= lr * mh * rect * bc2_sqrt / (sqrtf(v) + eps);
          else w -= lr * mh;
        } else {
          float vv = v;
          if (d.vmax) { vv = fmaxf(vm, v); vm = vv; }
          w -= (lr / bc1) * m / (sqrtf(vv) / bc2_sqrt + eps);
        }
      };
      upd(g4.x, w4.x, m4.x, v4.x, x4.x); upd(g4.y, w4.y, m4.y, v4.y, x4.y);
      upd(g4.z, w4.z, m4.z, v4.z, x4.z); upd(g4.w, w4.w, m4.w, v4.w, x4.w);
      *reinterpret_cast<float4*>(d.p + e) = w4;
      *reinterpret_cast<float4*>(d.m + e) = m4;
      *reinterpret_cast<float4*>(d.v + e) = v4;
      if (d.vmax) *reinterpret_cast<float4*>(d.vmax + e) = x4;
      if (d.shadow) {
        uint2 pk;
        pk.x = pack2bf(w4.x, w4.y);
        pk.y = pack2bf(w4.z, w4.w);
        *reinterpret_cast<uint2*>(d.shadow + e) = pk;
      }
    }
  }
}

}  // namespace

extern "C" {

int evk_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(src && dst && n > 0, "cast: bad args");
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(cast_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, s, src, src_dtype == EVK_F32, dst, dst_dtype == EVK_F32, (long)n);
  return evk_check_launch("cast");
}

int evk_act_fwd(const void* x, void* y, int64_t n, int32_t act, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && y && n > 0, "act_fwd: bad args");
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, (long)n, act);
  return evk_check_launch("act_fwd");
}

int evk_act_bwd(const void* dy, const void* ref, void* dx, int64_t n, int32_t act, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dy && ref && dx && n > 0, "act_bwd: bad args");
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)ref, (bf16_t*)dx, (long)n, act);
  return evk_check_launch("act_bwd");
}

int evk_dropout(const void* x, const void* resid, void* y, int64_t n, float p, uint64_t seed, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "dropout: bad args");
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)resid, (bf16_t*)y, (long)n, p, (unsigned long long)seed, evk_seed_epoch_ptr());
  return evk_check_launch("dropout");
}

int evk_embedding_fwd(const float* table, const int64_t* ids, const float* pos, const float* extra, void* out, int out_dtype,
                      int64_t rows, int32_t D, int32_t L, float scale, int64_t table_rows, const int64_t* pos0_dev, float* out_f32_copy, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(table && ids && out && rows > 0 && D > 0 && L > 0 && table_rows > 0, "embedding_fwd: bad args");
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(ew_blocks(rows * D)), dim3(256), 0, s, table, (const long long*)ids, pos, extra, out,
                     out_dtype == EVK_F32, (long)rows, D, L, scale, (long long)table_rows, (const long long*)pos0_dev, out_f32_copy);
  return evk_check_launch("embedding_fwd");
}

int evk_embedding_bwd(const void* dout, int d_dtype, const int64_t* ids, float* dtable, int64_t rows, int32_t D, float scale,
                      int64_t padding_idx, int64_t table_rows, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dout && ids && dtable && rows > 0 && D > 0 && table_rows > 0, "embedding_bwd: bad args");
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(ew_blocks(rows * D)), dim3(256), 0, s, dout, d_dtype == EVK_F32, (const long long*)ids,
                     dtable, (long)rows, D, scale, (long long)padding_idx, (long long)table_rows);
  return evk_check_launch("embedding_bwd");
}

int evk_colsum(const void* x, float* out, int64_t M, int32_t N, int64_t ld, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && out && M > 0 && N > 0 && ld >= N && ld % 2 == 0, "colsum: bad args (ld must be even)");
  static const bool probe_skip = evk_tunable("EVK_PROBE_SKIP_COLSUM", 0) != 0;          // timing probe: wrong bias gradients
  if (probe_skip) return EVK_OK;
  if (ld % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (N % 8 == 0 || ld >= (N + 7) / 8 * 8)) {
    const int gx = (int)cdiv(N, 256);
    long gy = cdiv(768, gx);
    if (gy > cdiv(M, 32)) gy = cdiv(M, 32);
    if (gy < 1) gy = 1;
    const long rows = cdiv(M, gy);
    gy = cdiv(M, rows);
    ProfScope ps(EVK_FAM_REDUCE, s);
    hipLaunchKernelGGL(colsum8_kernel, dim3(gx, (int)gy), dim3(256), 0, s, (const bf16_t*)x, out, (long)M, N, (long)ld, rows);
    return evk_check_launch("colsum");
  }
  const int bx = (int)cdiv(N, 256);
  long by = cdiv(1024, bx);
  if (by > cdiv(M, 16)) by = cdiv(M, 16);
  if (by < 1) by = 1;
  long rpb = cdiv(M, by);
  rpb += rpb & 1;
  by = cdiv(M, rpb);
  ProfScope ps(EVK_FAM_REDUCE, s);
  hipLaunchKernelGGL(colsum_kernel, dim3(bx, (int)by), dim3(256), 0, s, (const bf16_t*)x, out, (long)M, N, (long)ld, rpb);
  return evk_check_launch("colsum");
}

int evk_optim_step(float* p, const float* g, float* m, float* v, float* vmax, void* shadow, int64_t n, int32_t kind, float lr,
                   float beta1, float beta2, float eps, float weight_decay, float clip, int64_t step, evk_stream_t stream) {
  return evk_optim_step_scaled(p, g, m, v, vmax, shadow, n, kind, lr, beta1, beta2, eps, weight_decay, clip, step, 1.f, stream);
}

int evk_optim_step_scaled(float* p, const float* g, float* m, float* v, float* vmax, void* shadow, int64_t n, int32_t kind, float lr,
                          float beta1, float beta2, float eps, float weight_decay, float clip, int64_t step, float grad_scale,
                          evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(p && g && m && v && n > 0 && step >= 1 && (kind == 0 || kind == 1) && grad_scale > 0.f, "optim_step: bad args");
  OptP o{p, g, m, v, vmax, (bf16_t*)shadow, n, lr, beta1, beta2, eps, weight_decay, clip, 0.f, 0.f, 0.f, kind, 0, grad_scale};
  const double b1t = pow((double)beta1, (double)step), b2t = pow((double)beta2, (double)step);
  o.bc1 = (float)(1.0 - b1t);
  o.bc2_sqrt = (float)sqrt(1.0 - b2t);
  if (kind == 0) {  // torch.optim.RAdam
    const double rho_inf = 2.0 / (1.0 - beta2) - 1.0;
    const double rho_t = rho_inf - 2.0 * step * b2t / (1.0 - b2t);
    o.use_rect = rho_t > 5.0;
    if (o.use_rect) o.rect = (float)sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
  }
  ProfScope ps(EVK_FAM_OPTIM, s);
  hipLaunchKernelGGL(optim_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, o);
  return evk_check_launch("optim_step");
}

int evk_optim_step_dyn(float* p, float* g, float* m, float* v, float* vmax, void* shadow, int64_t n, int32_t kind, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float clip, const int32_t* step_dev,
                       const float* scale_state, float inv_world, int32_t zero_grad, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(p && g && m && v && n > 0 && step_dev && (kind == 0 || kind == 1) && inv_world > 0.f, "optim_step_dyn: bad args");
  OptDyn d{{p, g, m, v, vmax, (bf16_t*)shadow, n, lr, beta1, beta2, eps, weight_decay, clip, 0.f, 0.f, 0.f, kind, 0, 1.f},
           step_dev, scale_state, inv_world, zero_grad};
  ProfScope ps(EVK_FAM_OPTIM, s);
  hipLaunchKernelGGL(optim_dyn_kernel, dim3(std::min(ew_blocks(n / 4 + 1), 2048)), dim3(256), 0, s, d);
  return evk_check_launch("optim_step_dyn");
}

int evk_optim_group_step(float* p, float* g, float* m, float* v, float* vmax, void* shadow, int64_t n, int32_t kind, const float* hp_dev,
                         const int64_t* offsets_dev, int32_t n_params, int32_t* steps_dev, const unsigned char* touched_dev, float* coef_dev,
                         const float* scale_state, float inv_world, int32_t zero_grad, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(p && g && m && v && n > 0 && (n & 7) == 0 && hp_dev && offsets_dev && n_params > 0 && steps_dev && touched_dev && coef_dev &&
                  (kind == 0 || kind == 1) && inv_world > 0.f,
              "optim_group_step: bad args");
  EVK_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                reinterpret_cast<uintptr_t>(vmax) | reinterpret_cast<uintptr_t>(coef_dev)) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(shadow) & 7) == 0,
              "optim_group_step: flat buffers must be 16-byte aligned");
  ProfScope ps(EVK_FAM_OPTIM, s);
  hipLaunchKernelGGL(optim_group_coef_kernel, dim3((unsigned)cdiv(n_params, 256)), dim3(256), 0, s, hp_dev, kind, n_params, steps_dev, touched_dev,
                     reinterpret_cast<float4*>(coef_dev), scale_state);
  OptGroup d{p, g, m, v, vmax, (bf16_t*)shadow, (long)n, hp_dev, (const long*)offsets_dev, n_params, reinterpret_cast<const float4*>(coef_dev),
             scale_state, inv_world, kind, zero_grad};
  const long chunks = cdiv(n, OPT_CHUNK);
  hipLaunchKernelGGL(optim_group_kernel, dim3((unsigned)std::min<long>(chunks, 8192)), dim3(256), 0, s, d);
  return evk_check_launch("optim_group_step");
}

int evk_optim_bump(int32_t* step_dev, int32_t count, const float* scale_state, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(step_dev && count > 0, "optim_bump: bad args");
  ProfScope ps(EVK_FAM_OPTIM, s);
  hipLaunchKernelGGL(optim_bump_kernel, dim3((int)cdiv(count, 256)), dim3(256), 0, s, step_dev, count, scale_state);
  return evk_check_launch("optim_bump");
}

int evk_grad_nonfinite(const float* g, int64_t n, float* scale_state, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(g && n > 0 && scale_state && (reinterpret_cast<uintptr_t>(g) & 15) == 0, "grad_nonfinite: bad args");
  ProfScope ps(EVK_FAM_OPTIM, s);
  hipLaunchKernelGGL(grad_nonfinite_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, s, g, (long)n, scale_state);
  return evk_check_launch("grad_nonfinite");
}

int evk_act_nonfinite(const void* x, int64_t n, int32_t* count, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && n > 0 && count && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "act_nonfinite: bad args (16-byte aligned 16-bit tensor)");
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(act_nonfinite_kernel, dim3(ew_blocks(n / 8 + 1)), dim3(256), 0, s, (const bf16_t*)x, (long)n, count);
  return evk_check_launch("act_nonfinite");
}

int evk_grads_multi(const void* chunk_table, int32_t n_chunks, int32_t mode, float* scale_state, evk_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  EVK_REQUIRE(chunk_table && n_chunks > 0 && scale_state && mode >= 0 && mode <= 2, "grads_multi: bad args");
  ProfScope ps(EVK_FAM_OPTIM, s);
  const GradChunk* tab = (const GradChunk*)chunk_table;
  const dim3 grid((unsigned)(n_chunks < 4096 ? n_chunks : 4096)), block(256);
  if (mode == 0) hipLaunchKernelGGL(grads_multi_kernel<0>, grid, block, 0, s, tab, n_chunks, scale_state);
  else if (mode == 1) hipLaunchKernelGGL(grads_multi_kernel<1>, grid, block, 0, s, tab, n_chunks, scale_state);
  else hipLaunchKernelGGL(grads_multi_kernel<2>, grid, block, 0, s, tab, n_chunks, scale_state);
  return evk_check_launch("grads_multi");
}

int evk_loss_scale_update(float* scale_state, float growth, float backoff, int32_t interval, float min_scale, float max_scale,
                          evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(scale_state && growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval > 0 && min_scale > 0.f && max_scale >= min_scale,
              "loss_scale_update: bad args");
  ProfScope ps(EVK_FAM_OPTIM, s);
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, s, scale_state, growth, backoff, interval, min_scale, max_scale);
  return evk_check_launch("loss_scale_update");
}

}  // extern "C"
