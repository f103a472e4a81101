// bn.hip -- BatchNorm (train-mode batch statistics) + ReLU + residual for the NHWC ResNet-101 trunk and the
// projection heads (torch.nn.BatchNorm2d/1d: visual_extractor.py:30-38 via torchvision; utils_v0511.py:131-208),
// max-pool 3x3 s2 p1 and the patch mean (visual_extractor.py:40-42).  Rows = N*H*W (or B*T), channels contiguous.
// All HBM-bound: 16-byte loads, f32 statistics, per-channel partials reduced in LDS then one atomic per block.
#include "common.h"

namespace {

__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  v[0] = lo_bf(a.x); v[1] = hi_bf(a.x); v[2] = lo_bf(a.y); v[3] = hi_bf(a.y);
  v[4] = lo_bf(a.z); v[5] = hi_bf(a.z); v[6] = lo_bf(a.w); v[7] = hi_bf(a.w);
}
__device__ __forceinline__ void unpack8(const uint4& u, float (&v)[8]) {
  v[0] = lo_bf(u.x); v[1] = hi_bf(u.x); v[2] = lo_bf(u.y); v[3] = hi_bf(u.y);
  v[4] = lo_bf(u.z); v[5] = hi_bf(u.z); v[6] = lo_bf(u.w); v[7] = hi_bf(u.w);
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}

// Column reductions over rows of x[M][C] (C % 8 == 0, C <= 2048): thread -> channel group cg = tid % (C/8),
// row lane rl = tid / (C/8); a block sweeps rows rl, rl+RPB, ... of its slice, 4 rows (independent 16-byte loads) per
// iteration.  Two stages, no atomics (deterministic): per-block partials [nblk][2][C], then colreduce_final sums them.
// kind 0: {sum x, sum x^2}      kind 1 (bn backward): {sum g, sum g*xhat} with g = dz * (z > 0 if relu)
struct RedP {
  const bf16_t* x; const bf16_t* dz; const bf16_t* z; const float* mean; const float* invstd;
  float* part; long M; int C; int kind; int relu; long rows_per_block;
};

__global__ __launch_bounds__(256) void colreduce_kernel(const RedP p) {
  __shared__ float red[2][256 * 8];
  const int G = p.C >> 3;                 // channel groups (<= 256)
  const int tpr = G;                      // threads per row
  const int rpb = 256 / tpr;              // rows in flight per block
  const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const long r0 = blockIdx.x * p.rows_per_block;
  const long r1 = min(p.M, r0 + p.rows_per_block);
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = b[j] = 0.f;
  float mu[8], is[8];
  if (p.kind == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = p.mean[cg * 8 + j]; is[j] = p.invstd[cg * 8 + j]; }
  }
  constexpr int U = 4;
  for (long rb = r0 + rl; rb < r1; rb += (long)rpb * U) {
    uint4 xv[U], gv[U], zv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = rb + (long)u * rpb;
      const bool ok = r < r1;
      const long o = (ok ? r : r0) * p.C + cg * 8;
      xv[u] = *reinterpret_cast<const uint4*>(p.x + o);
      if (p.kind == 1) {
        gv[u] = *reinterpret_cast<const uint4*>(p.dz + o);
        if (p.relu) zv[u] = *reinterpret_cast<const uint4*>(p.z + o);
      }
      if (!ok) { xv[u] = make_uint4(0, 0, 0, 0); gv[u] = make_uint4(0, 0, 0, 0); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t xw[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
      if (p.kind == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = lo_bf(xw[j]), hi = hi_bf(xw[j]);
          a[2 * j] += lo; b[2 * j] += lo * lo; a[2 * j + 1] += hi; b[2 * j + 1] += hi * hi;
        }
      } else {
        const uint32_t gw[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
        const uint32_t zw[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
        const bool valid = (rb + (long)u * rpb) < r1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float g0 = lo_bf(gw[j]), g1 = hi_bf(gw[j]);
          if (p.relu) { g0 = lo_bf(zw[j]) > 0.f ? g0 : 0.f; g1 = hi_bf(zw[j]) > 0.f ? g1 : 0.f; }
          if (!valid) { g0 = 0.f; g1 = 0.f; }
          a[2 * j] += g0; b[2 * j] += g0 * (lo_bf(xw[j]) - mu[2 * j]) * is[2 * j];
          a[2 * j + 1] += g1; b[2 * j + 1] += g1 * (hi_bf(xw[j]) - mu[2 * j + 1]) * is[2 * j + 1];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][threadIdx.x * 8 + j] = a[j]; red[1][threadIdx.x * 8 + j] = b[j]; }
  __syncthreads();
  if (rl == 0) {
    float* o0 = p.part + (long)blockIdx.x * 2 * p.C;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float sa = 0.f, sb = 0.f;
      for (int q = 0; q < rpb; ++q) { sa += red[0][(q * tpr + cg) * 8 + j]; sb += red[1][(q * tpr + cg) * 8 + j]; }
      o0[cg * 8 + j] = sa;
      o0[p.C + cg * 8 + j] = sb;
    }
  }
}

// out0[c] = sum_blk part[blk][0][c], out1[c] = sum_blk part[blk][1][c].  Block = 16 columns x 16 row-lanes: every
// thread sums nblk/16 partials with 8 independent loads in flight, then the 16 row-lanes are combined in LDS.
__global__ __launch_bounds__(256) void colreduce_final_kernel(const float* __restrict__ part, float* __restrict__ out0,
                                                              float* __restrict__ out1, float* acc0, float* acc1, int nblk, int C) {
  __shared__ float red[16][17];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15);
  const int rl = threadIdx.x >> 4;
  const long ld = 2L * C;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  if (col < 2 * C) {
    int b = rl;
    for (; b + 16 * 7 < nblk; b += 16 * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += part[(long)(b + 16 * u) * ld + col];
    }
    for (; b < nblk; b += 16) acc[0] += part[(long)b * ld + col];
  }
  red[rl][threadIdx.x & 15] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (threadIdx.x < 16 && col < 2 * C) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[q][threadIdx.x];
    if (col < C) { out0[col] = s; if (acc0) acc0[col] += s; }
    else { out1[col - C] = s; if (acc1) acc1[col - C] += s; }
  }
}

struct FinP {
  const float* sum; const float* sumsq; const float* gamma; const float* beta; float* rmean; float* rvar;
  float* scale; float* shift; float* mean; float* invstd; int C; float count, momentum, eps; int training;
};
__device__ __forceinline__ void bn_finalize_channel(const FinP& p, int c, float sum, float sumsq);
__global__ void bn_finalize_kernel(const FinP p) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.C) return;
  bn_finalize_channel(p, c, p.training ? p.sum[c] : 0.f, p.training ? p.sumsq[c] : 0.f);
}
__device__ __forceinline__ void bn_finalize_channel(const FinP& p, int c, float sum, float sumsq) {
  float mu, var;
  if (p.training) {
    mu = sum / p.count;
    var = fmaxf(sumsq / p.count - mu * mu, 0.f);
    if (p.rmean) {
      p.rmean[c] = (1.f - p.momentum) * p.rmean[c] + p.momentum * mu;
      p.rvar[c] = (1.f - p.momentum) * p.rvar[c] + p.momentum * var * (p.count / fmaxf(p.count - 1.f, 1.f));
    }
  } else {
    mu = p.rmean[c];
    var = p.rvar[c];
  }
  const float is = rsqrtf(var + p.eps);
  const float g = p.gamma ? p.gamma[c] : 1.f, b = p.beta ? p.beta[c] : 0.f;
  p.scale[c] = g * is;
  p.shift[c] = b - mu * g * is;
  p.mean[c] = mu;
  p.invstd[c] = is;
}

// second reduction stage of the gate statistics a data-gradient GEMM epilogue wrote (gemm.hip: gatestats): 8 channels per block
// rows = 2: (sum g, sum g*z) -> sum_gx through z = gamma*xhat + beta;  rows = 3 (invstd given): (sum g, -, sum g*(x - mean)) -> sum_gx = invstd * row 2
// Ill-conditioned channels: the identity xhat = (z - beta) / gamma is evaluated on the 16-bit STORED z, whose rounding (2^-11 |z| in
// fp16) is divided by gamma -- a relative error of ~2^-11 |beta / gamma| in sum g*xhat, i.e. noise for the near-dead channels an
// ImageNet-pretrained resnet101 has (|gamma| << |beta|).  A block that owns such a channel (|beta| > EVK_GATE_RATIO |gamma|) recomputes
// sum g*xhat exactly from the gated gradient and the layer's raw convolution output, xhat = (y - mean) * invstd, for its 8 channels:
// no extra launch, no cost for well-conditioned layers, one strided pass over two tensors for the rare block that needs it.
#ifdef EVK_STORE_F16
#define EVK_GATE_RATIO 16.f
#else
#define EVK_GATE_RATIO 2.f
#endif
struct GateExact { const bf16_t* dz; const bf16_t* y; const float* mean; const float* istd; long M; };
__global__ __launch_bounds__(256) void bn_bwd_sums_from_gate_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float* __restrict__ sum_g, float* __restrict__ sum_gx,
                                                                    float* dbeta_acc, float* dgamma_acc, int C, int rows, const float* __restrict__ invstd,
                                                                    const GateExact ex) {
  __shared__ float red[16][17];
  const int j = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 8 + (j & 7);
  const int col = (j < 8 ? 0 : (rows - 1) * C) + c;
  const long ld = (long)rows * C;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  if (c < C) {
    int b = rl;
    for (; b + 16 * 7 < nblk; b += 16 * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += part[(long)(b + 16 * u) * ld + col];
    }
    for (; b < nblk; b += 16) acc[0] += part[(long)b * ld + col];
  }
  red[rl][j] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (threadIdx.x < 8 && c < C) {
    float sg = 0.f, sgz = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { sg += red[q][threadIdx.x]; sgz += red[q][threadIdx.x + 8]; }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sgx = invstd ? sgz * invstd[c] : (g != 0.f ? (sgz - b * sg) / g : 0.f);
    const bool ill = !invstd && ex.dz && fabsf(b) > EVK_GATE_RATIO * fabsf(g);
    sum_g[c] = sg;
    if (dbeta_acc) dbeta_acc[c] += sg;
    if (!ill) {
      sum_gx[c] = sgx;
      if (dgamma_acc) dgamma_acc[c] += sgx;
    }
    red[0][threadIdx.x] = ill ? 1.f : 0.f;          // (everything of red[][] has been consumed by these 8 threads' own columns)
  }
  if (invstd || !ex.dz) return;
  __syncthreads();
  bool any = false;
#pragma unroll
  for (int q = 0; q < 8; ++q) any |= red[0][q] != 0.f;
  if (!any) return;
  // exact sums for this block's 8 channels: 256 threads stride over the M rows, 16 bytes of each tensor per row
  const int c8 = blockIdx.x * 8;
  float mu[8], is[8], xa[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { mu[q] = ex.mean[c8 + q]; is[q] = ex.istd[c8 + q]; xa[q] = 0.f; }
  for (long r = threadIdx.x; r < ex.M; r += 256) {
    float gv[8], yv[8];
    unpack8(*reinterpret_cast<const uint4*>(ex.dz + r * C + c8), gv);
    unpack8(*reinterpret_cast<const uint4*>(ex.y + r * C + c8), yv);
#pragma unroll
    for (int q = 0; q < 8; ++q) xa[q] += gv[q] * ((yv[q] - mu[q]) * is[q]);
  }
  __shared__ float ex_red[4][8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float w = wave_sum(xa[q]);
    if ((threadIdx.x & 63) == 0) ex_red[threadIdx.x >> 6][q] = w;
  }
  const bool mine = threadIdx.x < 8 && red[0][threadIdx.x] != 0.f;
  __syncthreads();
  if (mine) {
    const float exact = (ex_red[0][threadIdx.x] + ex_red[1][threadIdx.x]) + (ex_red[2][threadIdx.x] + ex_red[3][threadIdx.x]);
    sum_gx[c8 + threadIdx.x] = exact;
    if (dgamma_acc) dgamma_acc[c8 + threadIdx.x] += exact;
  }
}

// Many partial rows, few channels (layer1: 9216 rows of 64-256 columns against 8-32 finalize workgroups): the second reduction stage spent
// 46-58 us in one dependent row loop per workgroup.  A fully parallel FOLD first adds rows r, r + h, r + 2h, ... into row r, in place
// (row r is written only by the threads that read it; every other row they read lies beyond h and is never written): the stage that
// follows walks h = nblk / 16 rows.
__global__ __launch_bounds__(256) void part_fold_kernel(float* __restrict__ part, int nblk, int h, int ld4) {
  const long total = (long)h * ld4;                         // float4 elements of the folded rows
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld4), c = (int)(i - (long)r * ld4);
    float4* row = reinterpret_cast<float4*>(part) + c;
    float4 a = row[(long)r * ld4];
    for (int q = r + h; q < nblk; q += h) { const float4 t = row[(long)q * ld4]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
    row[(long)r * ld4] = a;
  }
}
constexpr int FOLD_MIN_ROWS = 1024, FOLD_FACTOR = 16;
// folds `part` (nblk rows of ld floats) when it pays; returns the row count the next stage has to walk
static int maybe_fold(float* part, int nblk, int ld, hipStream_t s) {
  static const int on = evk_tunable("EVK_BN_FOLD", 1);
  if (!on || nblk < FOLD_MIN_ROWS || (ld & 3) || (reinterpret_cast<uintptr_t>(part) & 15)) return nblk;
  const int h = (nblk + FOLD_FACTOR - 1) / FOLD_FACTOR;
  const long total = (long)h * (ld / 4);
  hipLaunchKernelGGL(part_fold_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, part, nblk, h, ld / 4);
  return h;
}

// colreduce_final + bn_finalize in ONE launch for the training forward of the trunk (two dependent ~5 us launches on the critical
// path of each of the 104 convolutions otherwise): a block sums the partial rows of 8 channels -- 16 columns: their sums and their
// sums of squares -- with 16 row-lanes, then 8 threads finish the statistics of those channels.
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int nblk, float* __restrict__ sum_out,
                                                                float* __restrict__ sumsq_out, const FinP p) {
  __shared__ float red[16][17];
  const int j = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 8 + (j & 7);
  const int col = (j < 8 ? 0 : p.C) + c;
  const long ld = 2L * p.C;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  if (c < p.C) {
    int b = rl;
    for (; b + 16 * 7 < nblk; b += 16 * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += part[(long)(b + 16 * u) * ld + col];
    }
    for (; b < nblk; b += 16) acc[0] += part[(long)b * ld + col];
  }
  red[rl][j] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (threadIdx.x < 8 && c < p.C) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { s0 += red[q][threadIdx.x]; s1 += red[q][threadIdx.x + 8]; }
    sum_out[c] = s0; sumsq_out[c] = s1;
    bn_finalize_channel(p, c, s0, s1);
  }
}

// y = relu?(x*scale[c] + shift[c] + resid).  The grid stride (gridDim.x * 256) is a multiple of G (G divides 256), so a thread
// keeps the same 8 channels for all its iterations: scale / shift are loaded once; two iterations of loads are in flight.
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const bf16_t* __restrict__ resid,
                                                       bf16_t* __restrict__ y, long total8, int G, int relu) {
  const int cg = threadIdx.x % G;
  float sc[8], sh[8];
  *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(scale + cg * 8);
  *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(scale + cg * 8 + 4);
  *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(shift + cg * 8);
  *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(shift + cg * 8 + 4);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += 2 * stride) {
    const long i2 = i + stride;
    const bool two = i2 < total8;
    uint4 xa = *reinterpret_cast<const uint4*>(x + i * 8), xb = two ? *reinterpret_cast<const uint4*>(x + i2 * 8) : make_uint4(0, 0, 0, 0);
    uint4 ra = make_uint4(0, 0, 0, 0), rb = ra;
    if (resid) {
      ra = *reinterpret_cast<const uint4*>(resid + i * 8);
      if (two) rb = *reinterpret_cast<const uint4*>(resid + i2 * 8);
    }
    float v[8], r[8];
    unpack8(xa, v);
    unpack8(ra, r);
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = v[j] * sc[j] + sh[j] + r[j]; if (relu) v[j] = fmaxf(v[j], 0.f); }
    st8(y + i * 8, v);
    if (two) {
      unpack8(xb, v);
      unpack8(rb, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = v[j] * sc[j] + sh[j] + r[j]; if (relu) v[j] = fmaxf(v[j], 0.f); }
      st8(y + i2 * 8, v);
    }
  }
}

struct BwdP {
  const bf16_t* dz; const bf16_t* z; const bf16_t* x; const float* scale; const float* mean; const float* invstd;
  const float* sum_g; const float* sum_gx; bf16_t* dx; bf16_t* dres; long total8; int G; int relu; float inv_count;
};
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BwdP p) {
  // dx = scale * (g - sum_g/n - xhat * sum_gx/n), xhat = (x - mean) * invstd  ==  A*g + B*x + C per channel; a thread keeps its
  // 8 channels (grid stride is a multiple of G), so the coefficients are formed once instead of 40 scalar loads per iteration
  const int c0 = (threadIdx.x % p.G) * 8;
  float A[8], B[8], Cc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float sc = p.scale[c0 + j], is = p.invstd[c0 + j], mu = p.mean[c0 + j];
    const float sg = p.sum_g[c0 + j] * p.inv_count, sgx = p.sum_gx[c0 + j] * p.inv_count;
    A[j] = sc;
    B[j] = -sc * is * sgx;
    Cc[j] = sc * (mu * is * sgx - sg);
  }
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < p.total8; i += 2 * stride) {
    const long i2 = i + stride;
    const bool two = i2 < p.total8;
    const uint4 ga = *reinterpret_cast<const uint4*>(p.dz + i * 8), xa = *reinterpret_cast<const uint4*>(p.x + i * 8);
    const uint4 gb = two ? *reinterpret_cast<const uint4*>(p.dz + i2 * 8) : make_uint4(0, 0, 0, 0);
    const uint4 xb = two ? *reinterpret_cast<const uint4*>(p.x + i2 * 8) : make_uint4(0, 0, 0, 0);
    uint4 za = make_uint4(0, 0, 0, 0), zb = za;
    if (p.relu) {
      za = *reinterpret_cast<const uint4*>(p.z + i * 8);
      if (two) zb = *reinterpret_cast<const uint4*>(p.z + i2 * 8);
    }
    float g[8], xv[8], zv[8], o[8];
    unpack8(ga, g);
    unpack8(xa, xv);
    if (p.relu) {
      unpack8(za, zv);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = zv[j] > 0.f ? g[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = A[j] * g[j] + B[j] * xv[j] + Cc[j];
    st8(p.dx + i * 8, o);
    if (p.dres) st8(p.dres + i * 8, g);
    if (two) {
      unpack8(gb, g);
      unpack8(xb, xv);
      if (p.relu) {
        unpack8(zb, zv);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = zv[j] > 0.f ? g[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = A[j] * g[j] + B[j] * xv[j] + Cc[j];
      st8(p.dx + i2 * 8, o);
      if (p.dres) st8(p.dres + i2 * 8, g);
    }
  }
}

// max-pool 3x3 stride 2 pad 1, NHWC bf16, 8 channels per thread
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int H, int W, int C) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, G = C >> 3;
  const long total = (long)N * Ho * Wo * G;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % G);
    long t = i / G;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 - 1 + kh;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 - 1 + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        float v[8];
        ld8(x + (((long)n * H + ih) * W + iw) * C + cg * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
      }
    }
    st8(y + i * 8, m);
  }
}

// forward that also records, per output element, which of the 9 window taps (kh * 3 + kw) held the FIRST maximum
__global__ __launch_bounds__(256) void maxpool_fwd_idx_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, unsigned char* __restrict__ idx,
                                                              int N, int H, int W, int C) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, G = C >> 3;
  const long total = (long)N * Ho * Wo * G;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % G);
    long t = i / G;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m[8]; int am[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; am[j] = -1; }
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 - 1 + kh;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 - 1 + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        float v[8];
        ld8(x + (((long)n * H + ih) * W + iw) * C + cg * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (v[j] > m[j] || am[j] < 0) { m[j] = v[j]; am[j] = kh * 3 + kw; }
      }
    }
    st8(y + i * 8, m);
    uint2 pk;
    pk.x = (unsigned)am[0] | ((unsigned)am[1] << 8) | ((unsigned)am[2] << 16) | ((unsigned)am[3] << 24);
    pk.y = (unsigned)am[4] | ((unsigned)am[5] << 8) | ((unsigned)am[6] << 16) | ((unsigned)am[7] << 24);
    *reinterpret_cast<uint2*>(idx + i * 8) = pk;
  }
}

// backward from the recorded taps: an input pixel sits at tap (ih - 2 oh + 1, iw - 2 ow + 1) of each window containing it
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const unsigned char* __restrict__ idx, const bf16_t* __restrict__ dy,
                                                              bf16_t* __restrict__ dx, int N, int H, int W, int C) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, G = C >> 3;
  const long total = (long)N * H * W * G;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % G);
    long t = i / G;
    const int iw = (int)(t % W); t /= W;
    const int ih = (int)(t % H);
    const int n = (int)(t / H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int oh = ih / 2; oh <= (ih + 1) / 2 && oh < Ho; ++oh) {
      for (int ow = iw / 2; ow <= (iw + 1) / 2 && ow < Wo; ++ow) {
        const unsigned tap = (unsigned)((ih - 2 * oh + 1) * 3 + (iw - 2 * ow + 1));
        const long o = (((long)n * Ho + oh) * Wo + ow) * C + cg * 8;
        const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
        float g[8];
        ld8(dy + o, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned a = ((j < 4 ? pk.x : pk.y) >> (8 * (j & 3))) & 0xffu;
          if (a == tap) acc[j] += g[j];
        }
      }
    }
    st8(dx + i * 8, acc);
  }
}

// gather form of the backward: each input pixel sums dy of the (<=4) windows whose FIRST maximum (scan order kh,kw,
// strict >, as torch's max_pool2d indices) is this pixel -- deterministic, no atomics.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                          bf16_t* __restrict__ dx, int N, int H, int W, int C) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, G = C >> 3;
  const long total = (long)N * H * W * G;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % G);
    long t = i / G;
    const int iw = (int)(t % W); t /= W;
    const int ih = (int)(t % H);
    const int n = (int)(t / H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // windows containing the pixel: oh in [ih/2, (ih+1)/2], ow in [iw/2, (iw+1)/2]
    for (int oh = ih / 2; oh <= (ih + 1) / 2 && oh < Ho; ++oh) {
      for (int ow = iw / 2; ow <= (iw + 1) / 2 && ow < Wo; ++ow) {
        float m[8]; int am[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; am[j] = -1; }
        for (int kh = 0; kh < 3; ++kh) {
          const int yh = oh * 2 - 1 + kh;
          if ((unsigned)yh >= (unsigned)H) continue;
          for (int kw = 0; kw < 3; ++kw) {
            const int xw = ow * 2 - 1 + kw;
            if ((unsigned)xw >= (unsigned)W) continue;
            float v[8];
            ld8(x + (((long)n * H + yh) * W + xw) * C + cg * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (v[j] > m[j] || am[j] < 0) { m[j] = v[j]; am[j] = yh * W + xw; }
          }
        }
        float g[8];
        ld8(dy + (((long)n * Ho + oh) * Wo + ow) * C + cg * 8, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (am[j] == ih * W + iw) acc[j] += g[j];
      }
    }
    st8(dx + i * 8, acc);
  }
}

// patch mean: fc[n][c] = mean_p att[n][p][c] (f32 accumulate); backward adds dfc/P to every patch gradient
__global__ __launch_bounds__(256) void patch_mean_fwd_kernel(const bf16_t* __restrict__ att, bf16_t* __restrict__ fc, int N, int P, int C) {
  const int G = C >> 3;
  const long total = (long)N * G;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % G);
    const int n = (int)(i / G);
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    for (int q = 0; q < P; ++q) {
      float v[8];
      ld8(att + ((long)n * P + q) * C + cg * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] /= P;
    st8(fc + (long)n * C + cg * 8, a);
  }
}
__global__ __launch_bounds__(256) void patch_mean_bwd_kernel(const bf16_t* __restrict__ datt_in, const bf16_t* __restrict__ dfc,
                                                             bf16_t* __restrict__ datt, int N, int P, int C) {
  const int G = C >> 3;
  const long total = (long)N * P * G;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % G);
    const int n = (int)(i / ((long)G * P));
    float a[8], b[8];
    ld8(dfc + (long)n * C + cg * 8, b);
    if (datt_in) ld8(datt_in + i * 8, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (datt_in ? a[j] : 0.f) + b[j] / P;
    st8(datt + i * 8, a);
  }
}

inline int ew_blocks(long work) { long b = cdiv(work, 256); return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

constexpr int RED_MAX_BLOCKS = 1024;

int launch_reduce(const RedP& p0, float* out0, float* out1, void* ws, long ws_bytes, hipStream_t s, float* acc0 = nullptr,
                  float* acc1 = nullptr) {
  RedP p = p0;
  const int G = p.C >> 3;
  const int rpb = 256 / G;
  long blocks = cdiv(p.M, (long)rpb * 32);     // >= 32 rows per thread-row (8 iterations of 4 independent loads)
  if (blocks > RED_MAX_BLOCKS) blocks = RED_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  p.rows_per_block = cdiv(p.M, blocks);
  blocks = cdiv(p.M, p.rows_per_block);
  EVK_REQUIRE(ws && ws_bytes >= blocks * 2 * p.C * (long)sizeof(float), "colreduce: workspace too small (%ld bytes needed)",
              blocks * 2 * p.C * (long)sizeof(float));
  p.part = reinterpret_cast<float*>(ws);
  ProfScope ps(EVK_FAM_REDUCE, s);
  hipLaunchKernelGGL(colreduce_kernel, dim3((int)blocks), dim3(256), 0, s, p);
  hipLaunchKernelGGL(colreduce_final_kernel, dim3((int)cdiv(2 * p.C, 16)), dim3(256), 0, s, p.part, out0, out1, acc0, acc1, (int)blocks, p.C);
  return evk_check_launch("colreduce");
}

}  // namespace

extern "C" {

int64_t evk_colreduce_ws_bytes(int32_t C) { return (int64_t)RED_MAX_BLOCKS * 2 * C * (int64_t)sizeof(float); }

// sum[c] = sum_rows x, sumsq[c] = sum_rows x^2   (two-stage, deterministic; ws >= evk_colreduce_ws_bytes(C))
int evk_bn_stats(const void* x, float* sum, float* sumsq, void* ws, int64_t ws_bytes, int64_t M, int32_t C, evk_stream_t stream) {
  EVK_REQUIRE(x && sum && sumsq && M > 0 && C % 8 == 0 && C >= 8 && C <= 2048 && 256 % (C / 8) == 0,
              "bn_stats: bad args (C=%d must be a power-of-two multiple of 8, <= 2048)", C);
  RedP p{(const bf16_t*)x, nullptr, nullptr, nullptr, nullptr, nullptr, M, C, 0, 0, 0};
  return launch_reduce(p, sum, sumsq, ws, ws_bytes, reinterpret_cast<hipStream_t>(stream));
}

// second stage alone: sum / sumsq from the partial rows a convolution epilogue wrote (evk_conv2d_fwd_stats)
int evk_bn_stats_from_partials(const float* part, int32_t nblk, float* sum, float* sumsq, int32_t C, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(part && sum && sumsq && nblk > 0 && C > 0, "bn_stats_from_partials: bad args");
  ProfScope ps(EVK_FAM_REDUCE, s);
  hipLaunchKernelGGL(colreduce_final_kernel, dim3((int)cdiv(2 * C, 16)), dim3(256), 0, s, part, sum, sumsq, (float*)nullptr,
                     (float*)nullptr, (int)nblk, (int)C);
  return evk_check_launch("bn_stats_from_partials");
}

int evk_bn_finalize(const float* sum, const float* sumsq, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float* scale, float* shift, float* mean, float* invstd, int32_t C, float count,
                    float momentum, float eps, int32_t training, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(scale && shift && mean && invstd && C > 0, "bn_finalize: null outputs");
  EVK_REQUIRE(training ? (sum && sumsq && count > 0) : (running_mean && running_var), "bn_finalize: missing statistics");
  FinP p{sum, sumsq, gamma, beta, running_mean, running_var, scale, shift, mean, invstd, C, count, momentum, eps, training};
  ProfScope ps(EVK_FAM_REDUCE, s);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((int)cdiv(C, 256)), dim3(256), 0, s, p);
  return evk_check_launch("bn_finalize");
}

// Inference: the batch norm behind a convolution is a fixed affine map per output channel -- scale[c] = gamma / sqrt(var + eps),
// shift[c] = beta - mean * scale -- which the convolution kernels apply to their f32 accumulators (evk_conv2d_fwd_affine): conv + BN (+ identity)
// (+ ReLU) in one launch with ONE rounding to the storage type, the arithmetic of bn_finalize_channel's eval branch.
__global__ __launch_bounds__(256) void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                                             const float* __restrict__ var, float eps, float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float is = rsqrtf(var[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * is;
  shift[c] = b - mean[c] * g * is;
}

int evk_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, float* scale, float* shift,
                       int32_t C, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(running_mean && running_var && scale && shift && C > 0, "bn_eval_coeffs: bad args");
  ProfScope ps(EVK_FAM_REDUCE, s);
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((int)cdiv(C, 256)), dim3(256), 0, s, gamma, beta, running_mean, running_var, eps, scale, shift, (int)C);
  return evk_check_launch("bn_eval_coeffs");
}

int evk_bn_bwd_sums_from_gate_partials(const float* part, int32_t nblk, const float* gamma, const float* beta, float* sum_g, float* sum_gx,
                                       float* dbeta_acc, float* dgamma_acc, int32_t C, const void* dz, const void* y, const float* mean,
                                       const float* invstd, int64_t M, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(part && sum_g && sum_gx && nblk > 0 && C > 0 && C % 8 == 0, "bn_bwd_sums_from_gate_partials: bad args");
  EVK_REQUIRE(!dz || (y && mean && invstd && M > 0 && ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(y)) & 15) == 0),
              "bn_bwd_sums_from_gate_partials: the exact fallback needs dz, y (16-byte aligned), mean, invstd and M");
  ProfScope ps(EVK_FAM_REDUCE, s);
  const GateExact ex{(const bf16_t*)dz, (const bf16_t*)y, mean, invstd, (long)M};
  nblk = maybe_fold(const_cast<float*>(part), nblk, 2 * C, s);          // (the partial rows are scratch: consumed here, dead afterwards)
  hipLaunchKernelGGL(bn_bwd_sums_from_gate_kernel, dim3(C / 8), dim3(256), 0, s, part, (int)nblk, gamma, beta, sum_g, sum_gx, dbeta_acc, dgamma_acc, (int)C, 2, nullptr, ex);
  return evk_check_launch("bn_bwd_sums_from_gate_partials");
}

int evk_bn_bwd_sums_from_xstat_partials(const float* part, int32_t nblk, const float* invstd, float* sum_g, float* sum_gx, float* dbeta_acc,
                                        float* dgamma_acc, int32_t C, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(part && invstd && sum_g && sum_gx && nblk > 0 && C > 0 && C % 8 == 0, "bn_bwd_sums_from_xstat_partials: bad args");
  ProfScope ps(EVK_FAM_REDUCE, s);
  nblk = maybe_fold(const_cast<float*>(part), nblk, 3 * C, s);
  hipLaunchKernelGGL(bn_bwd_sums_from_gate_kernel, dim3(C / 8), dim3(256), 0, s, part, (int)nblk, nullptr, nullptr, sum_g, sum_gx, dbeta_acc, dgamma_acc, (int)C, 3,
                     invstd, GateExact{nullptr, nullptr, nullptr, nullptr, 0});
  return evk_check_launch("bn_bwd_sums_from_xstat_partials");
}

int evk_bn_stats_finalize_from_partials(const float* part, int32_t nblk, float* sum, float* sumsq, const float* gamma, const float* beta,
                                        float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* invstd,
                                        int32_t C, float count, float momentum, float eps, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(part && sum && sumsq && scale && shift && mean && invstd && nblk > 0 && C > 0 && C % 8 == 0 && count > 0, "bn_stats_finalize_from_partials: bad args");
  FinP p{sum, sumsq, gamma, beta, running_mean, running_var, scale, shift, mean, invstd, C, count, momentum, eps, 1};
  ProfScope ps(EVK_FAM_REDUCE, s);
  nblk = maybe_fold(const_cast<float*>(part), nblk, 2 * C, s);          // (the partial rows are scratch: consumed here, dead afterwards)
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C / 8), dim3(256), 0, s, part, (int)nblk, sum, sumsq, p);
  return evk_check_launch("bn_stats_finalize_from_partials");
}

int evk_bn_apply(const void* x, const float* scale, const float* shift, const void* resid, void* y, int64_t M, int32_t C,
                 int32_t relu, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && scale && shift && y && M > 0 && C % 8 == 0 && 256 % (C / 8) == 0, "bn_apply: bad args (C must be a power-of-two multiple of 8, <= 2048)");
  const long total8 = M * (C / 8);
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks(total8)), dim3(256), 0, s, (const bf16_t*)x, scale, shift,
                     (const bf16_t*)resid, (bf16_t*)y, total8, C / 8, relu);
  return evk_check_launch("bn_apply");
}

// sum_g[c] = sum_rows g, sum_gx[c] = sum_rows g*xhat  with g = dz * (z > 0 if relu)
int evk_bn_bwd_reduce(const void* dz, const void* z, const void* x, const float* mean, const float* invstd, float* sum_g,
                      float* sum_gx, void* ws, int64_t ws_bytes, int64_t M, int32_t C, int32_t relu, evk_stream_t stream) {
  return evk_bn_bwd_reduce_acc(dz, z, x, mean, invstd, sum_g, sum_gx, nullptr, nullptr, ws, ws_bytes, M, C, relu, stream);
}

// same, and dbeta_acc[c] += sum_g[c], dgamma_acc[c] += sum_gx[c] (the affine parameters' gradients) when given
int evk_bn_bwd_reduce_acc(const void* dz, const void* z, const void* x, const float* mean, const float* invstd, float* sum_g,
                          float* sum_gx, float* dbeta_acc, float* dgamma_acc, void* ws, int64_t ws_bytes, int64_t M, int32_t C,
                          int32_t relu, evk_stream_t stream) {
  EVK_REQUIRE(dz && x && mean && invstd && sum_g && sum_gx && (!relu || z) && M > 0 && C % 8 == 0 && C >= 8 && C <= 2048 &&
              256 % (C / 8) == 0, "bn_bwd_reduce: bad args");
  RedP p{(const bf16_t*)x, (const bf16_t*)dz, (const bf16_t*)z, mean, invstd, nullptr, M, C, 1, relu, 0};
  return launch_reduce(p, sum_g, sum_gx, ws, ws_bytes, reinterpret_cast<hipStream_t>(stream), dbeta_acc, dgamma_acc);
}

int evk_bn_bwd_apply(const void* dz, const void* z, const void* x, const float* scale, const float* mean, const float* invstd,
                     const float* sum_g, const float* sum_gx, void* dx, void* dres, int64_t M, int32_t C, int32_t relu,
                     evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dz && x && scale && mean && invstd && sum_g && sum_gx && dx && (!relu || z) && M > 0 && C % 8 == 0 && 256 % (C / 8) == 0,
              "bn_bwd_apply: bad args (C must be a power-of-two multiple of 8, <= 2048)");
  BwdP p{(const bf16_t*)dz, (const bf16_t*)z, (const bf16_t*)x, scale, mean, invstd, sum_g, sum_gx, (bf16_t*)dx, (bf16_t*)dres,
         M * (C / 8), C / 8, relu, 1.f / (float)M};
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(p.total8)), dim3(256), 0, s, p);
  return evk_check_launch("bn_bwd_apply");
}

int evk_maxpool3x3s2_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C % 8 == 0, "maxpool_fwd: bad args");
  const long total = (long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (C / 8);
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, N, H, W, C);
  return evk_check_launch("maxpool_fwd");
}

int evk_maxpool3x3s2_bwd(const void* x, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && C % 8 == 0, "maxpool_bwd: bad args");
  const long total = (long)N * H * W * (C / 8);
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C);
  return evk_check_launch("maxpool_bwd");
}

int evk_maxpool3x3s2_fwd_idx(const void* x, void* y, void* idx, int32_t N, int32_t H, int32_t W, int32_t C, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && y && idx && N > 0 && H > 0 && W > 0 && C % 8 == 0, "maxpool_fwd_idx: bad args");
  const long total = (long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (C / 8);
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(maxpool_fwd_idx_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, (unsigned char*)idx, N, H, W, C);
  return evk_check_launch("maxpool_fwd_idx");
}

int evk_maxpool3x3s2_bwd_idx(const void* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(idx && dy && dx && N > 0 && H > 0 && W > 0 && C % 8 == 0, "maxpool_bwd_idx: bad args");
  const long total = (long)N * H * W * (C / 8);
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, (const unsigned char*)idx, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C);
  return evk_check_launch("maxpool_bwd_idx");
}

int evk_patch_mean_fwd(const void* att, void* fc, int32_t N, int32_t P, int32_t C, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(att && fc && N > 0 && P > 0 && C % 8 == 0, "patch_mean_fwd: bad args");
  ProfScope ps(EVK_FAM_REDUCE, s);
  hipLaunchKernelGGL(patch_mean_fwd_kernel, dim3(ew_blocks((long)N * C / 8)), dim3(256), 0, s, (const bf16_t*)att, (bf16_t*)fc, N, P, C);
  return evk_check_launch("patch_mean_fwd");
}

int evk_patch_mean_bwd(const void* datt_in, const void* dfc, void* datt, int32_t N, int32_t P, int32_t C, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dfc && datt && N > 0 && P > 0 && C % 8 == 0, "patch_mean_bwd: bad args");
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(patch_mean_bwd_kernel, dim3(ew_blocks((long)N * P * C / 8)), dim3(256), 0, s, (const bf16_t*)datt_in,
                     (const bf16_t*)dfc, (bf16_t*)datt, N, P, C);
  return evk_check_launch("patch_mean_bwd");
}

}  // extern "C"
