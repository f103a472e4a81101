// conv.hip -- NHWC bf16 convolutions of the ResNet-101 trunk as implicit GEMMs (visual_extractor.py:30-38
// -> torchvision resnet101 children 0-7) on top of gemm.hip, plus the stem's pack / unpack kernels.
#include <stdlib.h>
#include "common.h"

namespace {

bool is_pointwise(const evk_conv_geom* g) {
  return g->KH == 1 && g->KW == 1 && g->stride_h == 1 && g->stride_w == 1 && g->pad_h == 0 && g->pad_w == 0;
}

int check_geom(const evk_conv_geom* g) {
  EVK_REQUIRE(g && g->N > 0 && g->Hi > 0 && g->Wi > 0 && g->Ci > 0 && g->Ho > 0 && g->Wo > 0 && g->Co > 0, "conv: bad geometry");
  EVK_REQUIRE(g->Ho == (g->Hi + 2 * g->pad_h - g->KH) / g->stride_h + 1 && g->Wo == (g->Wi + 2 * g->pad_w - g->KW) / g->stride_w + 1,
              "conv: Ho/Wo inconsistent with Hi/Wi, kernel, stride, pad");
  return EVK_OK;
}

// images f32 NCHW [N][3][H][W] -> bf16 [N][H+6][W+8][4] with a zero halo (3 left/top, 3 bottom, 5 right) and c=3 zero
__global__ void stem_pack_image_kernel(const float* __restrict__ img, uint2* __restrict__ out, int N, int H, int W) {
  const int Hp = H + 6, Wp = W + 8;
  const long total = (long)N * Hp * Wp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xp = (int)(i % Wp);
    const long t = i / Wp;
    const int yp = (int)(t % Hp);
    const int n = (int)(t / Hp);
    const int x = xp - 3, y = yp - 3;
    uint2 v = make_uint2(0, 0);
    if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
      const long b = ((long)n * 3 * H + y) * W + x;
      const long cs = (long)H * W;
      v.x = pack2bf(img[b], img[b + cs]);
      v.y = pack2bf(img[b + 2 * cs], 0.f);
    }
    out[i] = v;
  }
}

// w f32 OIHW [64][3][7][7] -> bf16 [64][7][8][4] (kw = 7 and c = 3 are zero)
__global__ void stem_pack_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * 7 * 8 * 4) return;
  const int c = i & 3, kw = (i >> 2) & 7, kh = (i >> 5) % 7, co = i / 224;
  float v = 0.f;
  if (c < 3 && kw < 7) v = w[((co * 3 + c) * 7 + kh) * 7 + kw];
  out[i] = f2bf(v);
}

// dw_packed f32 [64][7][8][4] -> += into OIHW f32 grad [64][3][7][7]
__global__ void stem_unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * 3 * 7 * 7) return;
  const int kw = i % 7, kh = (i / 7) % 7, c = (i / 49) % 3, co = i / 147;
  dw[i] += dwp[((co * 7 + kh) * 8 + kw) * 4 + c];
}

// Weights of a convolution as its DATA GRADIENT reads them: wt[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci].  With them the data gradient
// of a stride-1 "same" convolution is itself a forward convolution of dy (dx = conv(dy, wt), pad K-1-p), i.e. the K-contiguous
// A_CONV / B_PLAIN main loop instead of the gather + K-strided one.  One launch transposes every layer of a table: a workgroup moves
// one 32 x 32 (co, ci) tile of one tap through LDS.
struct FlipTab { const bf16_t* w[32]; bf16_t* wt[32]; int Co[32], Ci[32], T[32], KW[32], first[33]; int n; };
__global__ __launch_bounds__(256) void conv_flip_weights_kernel(const FlipTab t) {
  __shared__ bf16_t tile[32][34];
  int l = 0;
  while (l + 1 < t.n && (int)blockIdx.x >= t.first[l + 1]) ++l;
  const int Co = t.Co[l], Ci = t.Ci[l], T = t.T[l];
  int r = blockIdx.x - t.first[l];
  const int tiles_ci = (Ci + 31) / 32, tiles_co = (Co + 31) / 32;
  const int tci = r % tiles_ci; r /= tiles_ci;
  const int tco = r % tiles_co;
  const int tap = r / tiles_co;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const bf16_t* w = t.w[l];
  bf16_t* wt = t.wt[l];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = tco * 32 + ty + 8 * j, ci = tci * 32 + tx;
    tile[ty + 8 * j][tx] = (co < Co && ci < Ci) ? w[((long)co * T + tap) * Ci + ci] : (bf16_t)0;
  }
  __syncthreads();
  const int ftap = T - 1 - tap;            // (KH-1-kh) * KW + (KW-1-kw)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ci = tci * 32 + ty + 8 * j, co = tco * 32 + tx;
    if (ci < Ci && co < Co) wt[((long)ci * T + ftap) * Co + co] = tile[tx][ty + 8 * j];
  }
}

// dst[n][2 y][2 x][:] = src[n][y][x][:], every other pixel of dst zero: the data gradient of a pointwise stride-2 convolution from
// the compact product (one 16-byte piece per thread; C % 8 == 0)
__global__ __launch_bounds__(256) void upsample2_zero_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int Ho, int Wo, int C8, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    long t = i / C8;
    const int x = (int)(t % (2 * Wo)); t /= 2 * Wo;
    const int y = (int)(t % (2 * Ho));
    const long n = t / (2 * Ho);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!((x | y) & 1)) v = src[((n * Ho + (y >> 1)) * Wo + (x >> 1)) * C8 + c];
    dst[i] = v;
  }
}

// ---- data gradient of a 3x3 / stride 2 / pad 1 convolution by OUTPUT PARITY ----------------------------------------------------------
// dx[2i + a][2j + b] only receives the taps kh = a + 1 (mod 2), kw = b + 1 (mod 2): one tap for (even, even), two for the mixed classes,
// four for (odd, odd) -- 2.25 per input pixel instead of the 9 the gathering GEMM multiplies (three quarters of them by zero).  Each class
// is a small stride-1 convolution over dy (rows i, i + 1 / columns j, j + 1) with its own slice of the weights:
//   a = 0: tap t = 0 <-> kh = 1 (dy row i);   a = 1: t = 0 <-> kh = 2 (dy row i), t = 1 <-> kh = 0 (dy row i + 1);  likewise for columns.
// class weights wc[class][ci][t][u][co] = w[co][kh(a, t)][kw(b, u)][ci], classes in the order (0,0) (0,1) (1,0) (1,1) at element offsets
// 0, 1, 3, 5 x Ci*Co.
__device__ __forceinline__ int s2_tap(int par, int t) { return par == 0 ? 1 : (t == 0 ? 2 : 0); }
__global__ __launch_bounds__(256) void conv_s2_class_weights_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wc, int Co, int Ci) {
  const long per = (long)Ci * Co, total = 9 * per;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long q = i / per;                       // 0 | 1,2 | 3,4 | 5..8
    const int cls = q < 1 ? 0 : (q < 3 ? 1 : (q < 5 ? 2 : 3));
    const int a = cls >> 1, b = cls & 1, TA = a + 1, TB = b + 1;
    const long base = (cls == 0 ? 0 : cls == 1 ? 1 : cls == 2 ? 3 : 5) * per;
    long r = i - base;                            // [ci][t][u][co]
    const int co = (int)(r % Co); r /= Co;
    const int u = (int)(r % TB); r /= TB;
    const int t = (int)(r % TA);
    const int ci = (int)(r / TA);
    wc[i] = w[(((long)co * 3 + s2_tap(a, t)) * 3 + s2_tap(b, u)) * Ci + ci];
  }
}

// dx[n][2i + a][2j + b][:] = gate > 0 ? cls[a][b][n][i][j][:] : 0, plus per-block partial sums (sum g, sum g * gate) per channel.
// A thread owns 8 channels (blockDim.x % (C / 8) == 0: the grid stride keeps them), a block a contiguous range of output pixels;
// its pixel lanes are summed through LDS and leave as one partial row [2][C].
struct S2IP { const bf16_t* cls; const bf16_t* gate; bf16_t* dx; float* part; int N, Ho, Wo, C; long px_per_block; };
__global__ __launch_bounds__(256) void conv_s2_interleave_kernel(const S2IP p) {
  __shared__ float red[2][256][8];
  const int G = p.C >> 3, cg = threadIdx.x % G, pl = threadIdx.x / G, PL = 256 / G;
  const long M = (long)p.N * 4 * p.Ho * p.Wo, Mc = (long)p.N * p.Ho * p.Wo;
  const long m0 = blockIdx.x * p.px_per_block, m1 = min(M, m0 + p.px_per_block);
  float sg[8], sz[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sg[j] = 0.f; sz[j] = 0.f; }
  const int W2 = 2 * p.Wo, H2 = 2 * p.Ho;
  for (long m = m0 + pl; m < m1; m += PL) {
    const int x = (int)(m % W2);
    const long t = m / W2;
    const int y = (int)(t % H2);
    const long n = t / H2;
    const int c = ((y & 1) << 1) | (x & 1);
    const uint4 v = *reinterpret_cast<const uint4*>(p.cls + (((long)c * Mc + (n * p.Ho + (y >> 1)) * p.Wo + (x >> 1)) * p.C + cg * 8));
    const uint4 z = *reinterpret_cast<const uint4*>(p.gate + m * p.C + cg * 8);
    const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, zw[4] = {z.x, z.y, z.z, z.w};
    uint32_t ow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float z0 = lo_bf(zw[k]), z1 = hi_bf(zw[k]);
      // the gated value as it is stored (16-bit), so that the statistics describe the tensor the next pass reads
      const uint32_t kept = (z0 > 0.f ? (vw[k] & 0xffffu) : 0u) | (z1 > 0.f ? (vw[k] & 0xffff0000u) : 0u);
      ow[k] = kept;
      const float g0 = lo_bf(kept), g1 = hi_bf(kept);
      sg[2 * k] += g0; sg[2 * k + 1] += g1;
      sz[2 * k] += g0 * z0; sz[2 * k + 1] += g1 * z1;
    }
    *reinterpret_cast<uint4*>(p.dx + m * p.C + cg * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
  if (p.part) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][threadIdx.x][j] = sg[j]; red[1][threadIdx.x][j] = sz[j]; }
    __syncthreads();
    if (pl == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = 0.f, b = 0.f;
        for (int q = 0; q < PL; ++q) { a += red[0][q * G + cg][j]; b += red[1][q * G + cg][j]; }
        p.part[(long)blockIdx.x * 2 * p.C + cg * 8 + j] = a;
        p.part[(long)blockIdx.x * 2 * p.C + p.C + cg * 8 + j] = b;
      }
    }
  }
}

void stem_geom(evk_conv_geom* g, int N, int H, int W) {
  const int Hp = H + 6, Wp = W + 8;
  g->N = N; g->Hi = Hp; g->Wi = W / 2; g->Ci = 32;
  g->Ho = H / 2; g->Wo = W / 2; g->Co = 64;
  g->KH = 7; g->KW = 1; g->stride_h = 2; g->stride_w = 1; g->pad_h = 0; g->pad_w = 0;
  g->sN = (int64_t)Hp * Wp * 4; g->sH = (int64_t)Wp * 4; g->sW = 8;
}

}  // namespace

extern "C" {

int evk_conv2d_fwd(const void* x, const void* w, void* y, const evk_conv_geom* g, evk_stream_t stream) {
  return evk_conv2d_fwd_stats(x, w, y, g, nullptr, 0, nullptr, stream);
}

// rows of per-64-row-block partial statistics the GEMM epilogue writes for an M-row output
static int stats_rows(int M, int N) { return N <= 64 ? (int)((M + 255) / 256) * 4 : (int)((M + 127) / 128) * 2; }

int64_t evk_conv_stats_bytes(int64_t M, int32_t C) { return (int64_t)stats_rows((int)M, C) * 2 * C * (int64_t)sizeof(float); }

// short-K, wide-N pointwise products (Bottleneck conv3 forward, conv1 data gradient) go to the weight-stationary kernel
static bool use_ws(const evk_conv_geom* g, int64_t M, int K, int N) {
  static const int on = evk_tunable("EVK_CONV1X1_WS", 1);
  return on && is_pointwise(g) && N >= 2 * K && evk_conv1x1_ws_supported(M, K, N);
}

int evk_conv2d_fwd_stats(const void* x, const void* w, void* y, const evk_conv_geom* g, float* part, int64_t part_bytes,
                         int32_t* nblk, evk_stream_t stream) {
  if (int e = check_geom(g)) return e;
  const int64_t M = (int64_t)g->N * g->Ho * g->Wo;
  if (use_ws(g, M, g->Ci, g->Co) && (!part || part_bytes >= evk_conv1x1_ws_part_bytes(M, g->Ci, g->Co)))
    return evk_conv1x1_ws_fwd(x, w, y, M, g->Ci, g->Co, part, part_bytes, nblk, stream);
  return evk_conv2d_fwd_stats_tile(x, w, y, g, part, part_bytes, nblk, stream);
}

int evk_conv2d_fwd_stats_tile(const void* x, const void* w, void* y, const evk_conv_geom* g, float* part, int64_t part_bytes,
                              int32_t* nblk, evk_stream_t stream) {
  if (int e = check_geom(g)) return e;
  if (is_pointwise(g) && evk_gemm_strip_routes((int64_t)g->N * g->Ho * g->Wo, g->Co, g->Ci, part ? part_bytes : 0, part != nullptr))   // gemm_strip.hip
    return evk_gemm_strip(x, g->Ci, w, g->Ci, y, g->Co, (int64_t)g->N * g->Ho * g->Wo, g->Co, g->Ci, nullptr, 0, nullptr, 0, part, nullptr,
                          part_bytes, nblk, stream);
  if (evk_conv3x3_halo_routes(g, g->Ci, g->Co, part ? part_bytes : 0, part != nullptr))      // conv3x3.hip: halo tile in LDS
    return evk_conv3x3_halo(x, w, y, g->N, g->Hi, g->Wi, g->Ci, g->Co, nullptr, 0, nullptr, 0, part, nullptr, part_bytes, nblk, stream);
  evk_gemm d{};
  if (part) {
    const int M = g->N * g->Ho * g->Wo;
    EVK_REQUIRE(nblk && part_bytes >= evk_conv_stats_bytes(M, g->Co), "conv fwd: statistics buffer too small");
    *nblk = stats_rows(M, g->Co);
    d.colstats = part;
  }
  d.A = x; d.B = w; d.C = y;
  d.M = g->N * g->Ho * g->Wo; d.N = g->Co; d.K = g->KH * g->KW * g->Ci;
  d.a_mode = is_pointwise(g) ? EVK_A_PLAIN : EVK_A_CONV; d.b_mode = EVK_B_PLAIN;
  d.lda = g->Ci; d.ldb = d.K; d.ldc = g->Co;
  d.batch_outer = d.batch_inner = 1; d.alpha = 1.f; d.c_dtype = EVK_BF16;
  d.g = *g;
  d.g.sN = (int64_t)g->Hi * g->Wi * g->Ci; d.g.sH = (int64_t)g->Wi * g->Ci; d.g.sW = g->Ci;
  return evk_gemm_launch(&d, stream);
}

// Inference form of a convolution of the trunk: y = relu?(conv(x, w) * scale[co] + bias[co] (+ resid)), the eval-mode batch norm (evk_bn_eval_coeffs)
// applied to the f32 accumulators.  Routes: the weight-stationary / strip / halo kernels with their inference epilogues; the few geometries the tile
// GEMM takes (stride-2 convolutions, Co = 64) are refused (EVK_EUNSUPPORTED) -- evk_conv2d_fwd_affine_routes tells, the caller runs conv + bn there.
int evk_conv2d_fwd_affine(const void* x, const void* w, void* y, const evk_conv_geom* g, const float* scale, const float* bias, const void* resid,
                          int32_t relu, evk_stream_t stream) {
  if (int e = check_geom(g)) return e;
  EVK_REQUIRE(x && w && y && scale && bias, "conv fwd affine: null operand");
  const int64_t M = (int64_t)g->N * g->Ho * g->Wo;
  if (use_ws(g, M, g->Ci, g->Co)) return evk_conv1x1_ws_fwd_affine(x, w, y, M, g->Ci, g->Co, scale, bias, resid, relu, stream);
  if (is_pointwise(g) && evk_gemm_strip_routes(M, g->Co, g->Ci, 0, 0))
    return evk_gemm_strip_affine(x, g->Ci, w, g->Ci, y, g->Co, M, g->Co, g->Ci, scale, bias, resid, g->Co, relu, stream);
  if (evk_conv3x3_halo_routes(g, g->Ci, g->Co, 0, 0))
    return evk_conv3x3_halo_affine(x, w, y, g->N, g->Hi, g->Wi, g->Ci, g->Co, scale, bias, resid, g->Co, relu, stream);
  evk_set_error("conv fwd affine: no kernel with the inference epilogue takes this geometry (evk_conv2d_fwd_affine_routes)");
  return EVK_EUNSUPPORTED;
}

// 1 when evk_conv2d_fwd_affine takes this geometry (every route but the tile GEMM)
int evk_conv2d_fwd_affine_routes(const evk_conv_geom* g) {
  if (!g || check_geom(g)) return 0;
  const int64_t M = (int64_t)g->N * g->Ho * g->Wo;
  return use_ws(g, M, g->Ci, g->Co) || (is_pointwise(g) && evk_gemm_strip_routes(M, g->Co, g->Ci, 0, 0)) || evk_conv3x3_halo_routes(g, g->Ci, g->Co, 0, 0) ? 1 : 0;
}

int evk_conv2d_dgrad(const void* dy, const void* w, void* dx, const evk_conv_geom* g, evk_stream_t stream) {
  return evk_conv2d_dgrad_add(dy, w, nullptr, dx, g, stream);
}

int evk_conv2d_dgrad_add(const void* dy, const void* w, const void* resid, void* dx, const evk_conv_geom* g, evk_stream_t stream) {
  return evk_conv2d_dgrad_gated(dy, w, resid, nullptr, dx, g, stream);
}

int evk_conv2d_dgrad_gated(const void* dy, const void* w, const void* resid, const void* gate, void* dx, const evk_conv_geom* g,
                           evk_stream_t stream) {
  return evk_conv2d_dgrad_gated_stats(dy, w, resid, gate, dx, g, nullptr, 0, nullptr, stream);
}

int64_t evk_conv_xstat_bytes(const evk_conv_geom* g) {
  if (!g || check_geom(g)) return 0;
  const int64_t M = (int64_t)g->N * g->Hi * g->Wi;
  return use_ws(g, M, g->Co, g->Ci) ? evk_conv1x1_ws_part_bytes(M, g->Co, g->Ci) / 2 * 3 : 0;
}

int evk_conv2d_dgrad_gated_xstat(const void* dy, const void* w, const void* resid, const void* gate, void* dx, const evk_conv_geom* g,
                                 const void* stat_x, const float* stat_mean, float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream) {
  if (int e = check_geom(g)) return e;
  EVK_REQUIRE(nblk, "conv dgrad xstat: nblk is required");
  *nblk = 0;
  const int64_t need = evk_conv_xstat_bytes(g);
  if (stat_x && gate && part && need > 0 && part_bytes >= need)
    return evk_conv1x1_ws_dgrad_xstat(dy, w, resid, gate, dx, (int64_t)g->N * g->Hi * g->Wi, g->Co, g->Ci, stat_x, stat_mean, part, part_bytes, nblk, stream);
  return evk_conv2d_dgrad_gated(dy, w, resid, gate, dx, g, stream);
}

int evk_conv2d_dgrad_gated_stats(const void* dy, const void* w, const void* resid, const void* gate, void* dx, const evk_conv_geom* g,
                                 float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream) {
  if (int e = check_geom(g)) return e;
  {
    const int64_t M = (int64_t)g->N * g->Hi * g->Wi;
    if (use_ws(g, M, g->Co, g->Ci) && (!part || (gate && part_bytes >= evk_conv1x1_ws_part_bytes(M, g->Co, g->Ci))))
      return evk_conv1x1_ws_dgrad(dy, w, resid, gate, dx, M, g->Co, g->Ci, part, part_bytes, nblk, stream);
  }
  const int T = g->KH * g->KW;
  EVK_REQUIRE(ilog2_exact(g->Co) >= 3, "conv dgrad: Co must be a power of two >= 8");
  evk_gemm d{};
  d.A = dy; d.B = w; d.C = dx;
  d.M = g->N * g->Hi * g->Wi; d.N = g->Ci; d.K = T * g->Co;
  d.a_mode = is_pointwise(g) ? EVK_A_PLAIN : EVK_A_DGRAD; d.b_mode = EVK_B_KSTR;
  d.lda = g->Co; d.ldb = (int64_t)T * g->Ci; d.ldc = g->Ci;
  d.b_klog = ilog2_exact(g->Co); d.b_tapstride = g->Ci;
  d.batch_outer = d.batch_inner = 1; d.alpha = 1.f; d.c_dtype = EVK_BF16;
  if (resid) { d.resid = resid; d.ldr = g->Ci; d.r_dtype = EVK_BF16; }
  if (gate) { d.relu_gate = gate; d.ldg = g->Ci; }
  if (part) {
    EVK_REQUIRE(gate && nblk && part_bytes >= evk_conv_stats_bytes(d.M, g->Ci), "conv dgrad: gate statistics need a gate and a large enough buffer");
    *nblk = stats_rows(d.M, g->Ci);
    d.gatestats = part;
  }
  d.g = *g;
  return evk_gemm_launch(&d, stream);
}

static bool s2_parity_geom(const evk_conv_geom* g) {
  return g && g->KH == 3 && g->KW == 3 && g->stride_h == 2 && g->stride_w == 2 && g->pad_h == 1 && g->pad_w == 1 && g->Hi == 2 * g->Ho &&
         g->Wi == 2 * g->Wo && g->Ci % 8 == 0 && g->Ci <= 2048 && 256 % (g->Ci / 8) == 0 && ilog2_exact(g->Co) >= 3;
}

int evk_conv3x3s2_dgrad_parity_supported(const evk_conv_geom* g) {
  static const int on = evk_tunable("EVK_S2_PARITY", 1);
  return on && s2_parity_geom(g) ? 1 : 0;
}

int64_t evk_conv3x3s2_dgrad_parity_ws_bytes(const evk_conv_geom* g) {
  return s2_parity_geom(g) ? (int64_t)g->N * g->Hi * g->Wi * g->Ci * 2 : 0;          // the four compact class outputs
}

int evk_conv3x3s2_class_weights(const void* w, void* wc, int32_t Co, int32_t Ci, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(w && wc && Co > 0 && Ci > 0, "conv3x3s2_class_weights: bad args");
  const long total = 9L * Co * Ci;
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(conv_s2_class_weights_kernel, dim3((unsigned)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256))), dim3(256), 0, s,
                     (const bf16_t*)w, (bf16_t*)wc, Co, Ci);
  return evk_check_launch("conv_s2_class_weights");
}

int evk_conv3x3s2_dgrad_parity(const void* dy, const void* wc, const void* gate, void* dx, const evk_conv_geom* g, void* ws, int64_t ws_bytes,
                               float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (int e = check_geom(g)) return e;
  EVK_REQUIRE(dy && wc && gate && dx && ws, "conv3x3s2_dgrad_parity: null operand (the ReLU gate is required)");
  EVK_REQUIRE(s2_parity_geom(g), "conv3x3s2_dgrad_parity: 3x3 / stride 2 / pad 1 with even input size, Ci %% 8 == 0 dividing 2048");
  EVK_REQUIRE(ws_bytes >= evk_conv3x3s2_dgrad_parity_ws_bytes(g), "conv3x3s2_dgrad_parity: workspace too small");
  const long Mc = (long)g->N * g->Ho * g->Wo, per = (long)g->Ci * g->Co;
  const long woff[4] = {0, 1, 3, 5};
  for (int cls = 0; cls < 4; ++cls) {
    const int a = cls >> 1, b = cls & 1;
    evk_gemm d{};
    d.A = dy; d.B = (const bf16_t*)wc + woff[cls] * per; d.C = (bf16_t*)ws + (long)cls * Mc * g->Ci;
    d.M = (int)Mc; d.N = g->Ci; d.K = (a + 1) * (b + 1) * g->Co;
    d.a_mode = (a | b) ? EVK_A_CONV : EVK_A_PLAIN; d.b_mode = EVK_B_PLAIN;
    d.lda = g->Co; d.ldb = d.K; d.ldc = g->Ci;
    d.batch_outer = d.batch_inner = 1; d.alpha = 1.f; d.c_dtype = EVK_BF16;
    d.g.N = g->N; d.g.Hi = g->Ho; d.g.Wi = g->Wo; d.g.Ci = g->Co; d.g.Ho = g->Ho; d.g.Wo = g->Wo; d.g.Co = g->Ci;
    d.g.KH = a + 1; d.g.KW = b + 1; d.g.stride_h = d.g.stride_w = 1; d.g.pad_h = d.g.pad_w = 0;
    d.g.sN = (int64_t)g->Ho * g->Wo * g->Co; d.g.sH = (int64_t)g->Wo * g->Co; d.g.sW = g->Co;
    if (int e = evk_gemm_launch(&d, stream)) return e;
  }
  const long M = 4 * Mc;
  long blocks = cdiv(M, 256 / (g->Ci / 8) * 4);          // at least four pixels per thread row, at most 2048 blocks
  if (blocks > 2048) blocks = 2048;
  if (part) {
    EVK_REQUIRE(nblk, "conv3x3s2_dgrad_parity: nblk is required with part");
    const long cap = part_bytes / (2L * g->Ci * (long)sizeof(float));
    EVK_REQUIRE(cap >= 1, "conv3x3s2_dgrad_parity: statistics buffer too small");
    if (blocks > cap) blocks = cap;
    *nblk = (int)blocks;
  }
  S2IP p{(const bf16_t*)ws, (const bf16_t*)gate, (bf16_t*)dx, part, g->N, g->Ho, g->Wo, g->Ci, cdiv(M, blocks)};
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(conv_s2_interleave_kernel, dim3((unsigned)cdiv(M, p.px_per_block)), dim3(256), 0, s, p);
  if (part) *nblk = (int)cdiv(M, p.px_per_block);
  return evk_check_launch("conv_s2_interleave");
}

int evk_upsample2_zero(const void* src, void* dst, int32_t N, int32_t Ho, int32_t Wo, int32_t C, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(src && dst && N > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, "upsample2_zero: bad args (C %% 8)");
  EVK_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0, "upsample2_zero: 16-byte alignment");
  const long total = (long)N * 2 * Ho * 2 * Wo * (C / 8);
  const long blocks = cdiv(total, 256 * 4);
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(upsample2_zero_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 65535 * 4 ? 65535 * 4 : blocks))), dim3(256), 0, s,
                     reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), Ho, Wo, C / 8, total);
  return evk_check_launch("upsample2_zero");
}

int evk_conv_flip_weights(const void* const* w, void* const* wt, const int32_t* Co, const int32_t* Ci, const int32_t* KH, const int32_t* KW,
                          int32_t n_layers, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(w && wt && Co && Ci && KH && KW && n_layers > 0, "conv_flip_weights: bad args");
  for (int base = 0; base < n_layers; base += 32) {
    FlipTab t{};
    t.n = n_layers - base < 32 ? n_layers - base : 32;
    int tot = 0;
    for (int l = 0; l < t.n; ++l) {
      const int i = base + l;
      EVK_REQUIRE(w[i] && wt[i] && Co[i] > 0 && Ci[i] > 0 && KH[i] > 0 && KW[i] > 0, "conv_flip_weights: bad layer %d", i);
      t.w[l] = (const bf16_t*)w[i]; t.wt[l] = (bf16_t*)wt[i]; t.Co[l] = Co[i]; t.Ci[l] = Ci[i]; t.T[l] = KH[i] * KW[i]; t.KW[l] = KW[i];
      t.first[l] = tot;
      tot += t.T[l] * ((Co[i] + 31) / 32) * ((Ci[i] + 31) / 32);
    }
    t.first[t.n] = tot;
    ProfScope ps(EVK_FAM_ELTWISE, s);
    hipLaunchKernelGGL(conv_flip_weights_kernel, dim3(tot), dim3(256), 0, s, t);
    if (int e = evk_check_launch("conv_flip_weights")) return e;
  }
  return EVK_OK;
}

int evk_conv2d_dgrad_flipped_gated_stats(const void* dy, const void* wt, const void* resid, const void* gate, void* dx, const evk_conv_geom* g,
                                         float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream) {
  if (int e = check_geom(g)) return e;
  EVK_REQUIRE(g->stride_h == 1 && g->stride_w == 1 && g->KH == 2 * g->pad_h + 1 && g->KW == 2 * g->pad_w + 1 && g->Co % 8 == 0 && g->Ci % 8 == 0,
              "conv dgrad (flipped weights): stride 1, pad = (K - 1) / 2, channels %% 8");
  const int T = g->KH * g->KW;
  // pointwise: dx[M][Ci] = dy[M][Co] . wt[Ci][Co]^T -- the strip GEMM when it pays
  if (is_pointwise(g) && evk_gemm_strip_routes((int64_t)g->N * g->Hi * g->Wi, g->Ci, g->Co, part ? part_bytes : 0, part != nullptr)) {
    EVK_REQUIRE(!part || gate, "conv dgrad: gate statistics need a gate");
    return evk_gemm_strip(dy, g->Co, wt, g->Co, dx, g->Ci, (int64_t)g->N * g->Hi * g->Wi, g->Ci, g->Co, resid, g->Ci, gate, g->Ci, nullptr, part,
                          part_bytes, nblk, stream);
  }
  // the data gradient is a forward convolution of dy (Co channels) into dx (Ci channels) over wt: same halo kernel
  if (evk_conv3x3_halo_routes(g, g->Co, g->Ci, part ? part_bytes : 0, part != nullptr)) {
    EVK_REQUIRE(!part || gate, "conv dgrad: gate statistics need a gate");
    return evk_conv3x3_halo(dy, wt, dx, g->N, g->Hi, g->Wi, g->Co, g->Ci, resid, g->Ci, gate, g->Ci, nullptr, part, part_bytes, nblk, stream);
  }
  evk_gemm d{};
  d.A = dy; d.B = wt; d.C = dx;
  d.M = g->N * g->Hi * g->Wi; d.N = g->Ci; d.K = T * g->Co;
  d.a_mode = EVK_A_CONV; d.b_mode = EVK_B_PLAIN;
  d.lda = g->Co; d.ldb = d.K; d.ldc = g->Ci;
  d.batch_outer = d.batch_inner = 1; d.alpha = 1.f; d.c_dtype = EVK_BF16;
  if (resid) { d.resid = resid; d.ldr = g->Ci; d.r_dtype = EVK_BF16; }
  if (gate) { d.relu_gate = gate; d.ldg = g->Ci; }
  if (part) {
    EVK_REQUIRE(gate && nblk && part_bytes >= evk_conv_stats_bytes(d.M, g->Ci), "conv dgrad: gate statistics need a gate and a large enough buffer");
    *nblk = stats_rows(d.M, g->Ci);
    d.gatestats = part;
  }
  // the convolution the GEMM gathers: input = dy (N, Ho, Wo, Co), output = dx (N, Hi, Wi, Ci), same kernel size, stride 1, pad K - 1 - p
  d.g = *g;
  d.g.Hi = g->Ho; d.g.Wi = g->Wo; d.g.Ci = g->Co; d.g.Ho = g->Hi; d.g.Wo = g->Wi; d.g.Co = g->Ci;
  d.g.pad_h = g->KH - 1 - g->pad_h; d.g.pad_w = g->KW - 1 - g->pad_w;
  d.g.sN = (int64_t)g->Ho * g->Wo * g->Co; d.g.sH = (int64_t)g->Wo * g->Co; d.g.sW = g->Co;
  return evk_gemm_launch(&d, stream);
}

static int wgrad_desc(evk_gemm& d, const void* dy, const void* x, float* dw, const evk_conv_geom* g) {
  const int T = g->KH * g->KW;
  d.A = dy; d.B = x; d.C = dw;
  d.M = g->Co; d.N = g->Ci; d.K = g->N * g->Ho * g->Wo;
  d.a_mode = EVK_A_KSTR; d.lda = g->Co;
  d.ldc = (int64_t)T * g->Ci; d.sCi = g->Ci;
  d.batch_outer = 1; d.batch_inner = T; d.alpha = 1.f; d.c_dtype = EVK_F32; d.accumulate = 1;
  d.g = *g;
  if (is_pointwise(g)) { d.b_mode = EVK_B_KSTR; d.ldb = g->Ci; }
  else {
    d.b_mode = EVK_B_WGATHER;
    d.g.sN = (int64_t)g->Hi * g->Wi * g->Ci; d.g.sH = (int64_t)g->Wi * g->Ci; d.g.sW = g->Ci;
  }
  return EVK_OK;
}

int64_t evk_conv2d_wgrad_ws_bytes(const evk_conv_geom* g) {
  if (!g) return 0;
  evk_gemm d{};
  wgrad_desc(d, nullptr, nullptr, nullptr, g);
  int64_t nb = evk_gemm_workspace_bytes(&d);
  if (evk_conv3x3_wgrad_halo_routes(g)) {           // conv3x3.hip: enough for either route
    const int64_t nh = evk_conv3x3_wgrad_halo_ws_bytes(g->N, g->Hi, g->Wi, g->Ci, g->Co);
    if (nh > nb) nb = nh;
  }
  return nb;
}

int evk_conv2d_wgrad(const void* dy, const void* x, float* dw, const evk_conv_geom* g, void* ws, int64_t ws_bytes, evk_stream_t stream) {
  if (int e = check_geom(g)) return e;
  if (ws && evk_conv3x3_wgrad_halo_routes(g) && ws_bytes >= evk_conv3x3_wgrad_halo_ws_bytes(g->N, g->Hi, g->Wi, g->Ci, g->Co))
    return evk_conv3x3_wgrad_halo(dy, x, dw, g->N, g->Hi, g->Wi, g->Ci, g->Co, ws, ws_bytes, stream);
  evk_gemm d{};
  wgrad_desc(d, dy, x, dw, g);
  d.workspace = ws; d.workspace_bytes = ws_bytes;
  return evk_gemm_launch(&d, stream);
}

int evk_stem_pack_image(const float* img, void* xpad, int32_t N, int32_t H, int32_t W, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(img && xpad && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "stem_pack_image: bad args");
  const long total = (long)N * (H + 6) * (W + 8);
  const int blocks = (int)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(stem_pack_image_kernel, dim3(blocks), dim3(256), 0, s, img, (uint2*)xpad, N, H, W);
  return evk_check_launch("stem_pack_image");
}

int evk_stem_pack_weight(const float* w, void* wp, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(w && wp, "stem_pack_weight: null");
  hipLaunchKernelGGL(stem_pack_weight_kernel, dim3(56), dim3(256), 0, s, w, (bf16_t*)wp);
  return evk_check_launch("stem_pack_weight");
}

int evk_stem_unpack_wgrad(const float* dwp, float* dw, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dwp && dw, "stem_unpack_wgrad: null");
  hipLaunchKernelGGL(stem_unpack_wgrad_kernel, dim3(37), dim3(256), 0, s, dwp, dw);
  return evk_check_launch("stem_unpack_wgrad");
}

int evk_stem_fwd(const void* xpad, const void* wp, void* y, int32_t N, int32_t H, int32_t W, evk_stream_t stream) {
  return evk_stem_fwd_stats(xpad, wp, y, N, H, W, nullptr, 0, nullptr, stream);
}

int evk_stem_fwd_stats(const void* xpad, const void* wp, void* y, int32_t N, int32_t H, int32_t W, float* part, int64_t part_bytes,
                       int32_t* nblk, evk_stream_t stream) {
  EVK_REQUIRE(H % 2 == 0 && W % 2 == 0, "stem: H and W must be even");
  if (evk_stem_halo_supported(N, H, W) && (!part || part_bytes >= evk_stem_halo_part_bytes(N, H, W)))      // stem.hip: halo tile in LDS
    return evk_stem_halo_fwd(xpad, wp, y, N, H, W, part, part_bytes, nblk, stream);
  evk_gemm d{};
  if (part) {
    const int64_t M = (int64_t)N * (H / 2) * (W / 2);
    EVK_REQUIRE(nblk && part_bytes >= evk_conv_stats_bytes(M, 64), "stem fwd: statistics buffer too small");
    *nblk = stats_rows((int)M, 64);
    d.colstats = part;
  }
  stem_geom(&d.g, N, H, W);
  d.A = xpad; d.B = wp; d.C = y;
  d.M = N * d.g.Ho * d.g.Wo; d.N = 64; d.K = 224;
  d.a_mode = EVK_A_CONV; d.b_mode = EVK_B_PLAIN; d.ldb = 224; d.ldc = 64;
  d.batch_outer = d.batch_inner = 1; d.alpha = 1.f; d.c_dtype = EVK_BF16;
  return evk_gemm_launch(&d, stream);
}

static void stem_wgrad_desc(evk_gemm& d, const void* dy, const void* xpad, float* dwp, int N, int H, int W) {
  stem_geom(&d.g, N, H, W);
  d.A = dy; d.B = xpad; d.C = dwp;
  d.M = 64; d.N = 32; d.K = N * d.g.Ho * d.g.Wo;
  d.a_mode = EVK_A_KSTR; d.lda = 64; d.b_mode = EVK_B_WGATHER;
  d.ldc = 224; d.sCi = 32; d.batch_outer = 1; d.batch_inner = 7;
  d.alpha = 1.f; d.c_dtype = EVK_F32; d.accumulate = 1;
}

int64_t evk_stem_wgrad_ws_bytes(int32_t N, int32_t H, int32_t W) {
  evk_gemm d{};
  stem_wgrad_desc(d, nullptr, nullptr, nullptr, N, H, W);
  int64_t nb = evk_gemm_workspace_bytes(&d);
  const int64_t nh = evk_stem_halo_wgrad_ws_bytes(N, H, W);          // stem.hip: enough for either route
  return nh > nb ? nh : nb;
}

int evk_stem_wgrad(const void* dy, const void* xpad, float* dwp, int32_t N, int32_t H, int32_t W, void* ws, int64_t ws_bytes, evk_stream_t stream) {
  EVK_REQUIRE(H % 2 == 0 && W % 2 == 0, "stem: H and W must be even");
  if (ws && evk_stem_halo_supported(N, H, W) && ws_bytes >= evk_stem_halo_wgrad_ws_bytes(N, H, W))
    return evk_stem_halo_wgrad(dy, xpad, dwp, N, H, W, ws, ws_bytes, stream);
  evk_gemm d{};
  stem_wgrad_desc(d, dy, xpad, dwp, N, H, W);
  d.workspace = ws; d.workspace_bytes = ws_bytes;
  return evk_gemm_launch(&d, stream);
}

}  // extern "C"
