// stem.hip -- the 7x7 / stride 2 / pad 3 stem convolution of the ResNet-101 trunk (torchvision ResNet.conv1 as driven by
// modules/visual_extractor.py:30-38) on the packed image of conv.hip (xpad [N][H+6][W+8][4] 16-bit with a zero halo and a zero fourth
// channel, weights [64][7][8][4] with a zero eighth tap): y[n][oy][ox][co] = sum_{kh < 7} sum_{e < 32} xpad[n][2 oy + kh][2 ox][e] * w[co][kh][e]
// -- per tap row kh the 32 contracted elements are 64 contiguous bytes of the image row (8 pixels x 4 channels).
//
// Through gemm.hip's implicit-GEMM loader this is a 2.36 M x 64 x 224 product whose A tile re-fills every input element ~14 times (the
// 8-pixel windows of neighbouring outputs overlap): 309 us forward, 482 us weight gradient on 64 images of 384^2 for ~65 us of
// compulsory traffic each.  Here a workgroup stages the input halo of a tile of 4 output rows x 64 output columns ONCE (13 rows x 136
// pixels x 8 B = 14 KB, double-buffered), reads the MFMA A fragments straight out of it (16 consecutive outputs = 16 consecutive
// 16-byte steps of one row: conflict-free ds_read_b128, no swizzle), and keeps ALL 7 x 4 weight fragments in registers for the whole
// launch (the wave computes all 64 output channels of its 32 pixels).  Workgroups are persistent (one per CU) and walk the tiles;
// batch-norm statistics accumulate in registers over all the tiles of a wave and leave as one partial row per wave.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int NTH = 512;
constexpr int TRO = 4, TCO = 64;                 // output rows / columns per tile (256 pixels = 16 MFMA row tiles, two per wave)
constexpr int HROWS = 2 * TRO + 5;               // 13 input rows
constexpr int HCOLS = 2 * TCO + 8;               // 136 input pixels per row (134 needed)
constexpr int HPITCH = HCOLS * 8;                // 1088 B
constexpr int HBYTES = HROWS * HPITCH;           // 14144 B per buffer
constexpr int NPC = (HROWS * (HPITCH / 16) + NTH - 1) / NTH;   // 16-byte pieces per thread and tile (2)

struct StemP {
  const bf16_t* xpad; const bf16_t* w; bf16_t* y;
  int N, H, W;                     // image size (H, W multiples of 8 and 128)
  int tiles_x, tiles_y, ntiles;
  float* colstats;                 // [gridDim.x * 8][2][64] or null
};

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
  return v + __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  return dpp_add<0x140>(v);
}

__global__ __launch_bounds__(NTH, 2) void stem_fwd_kernel(const StemP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * HBYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  const int Hp = p.H + 6, Wp = p.W + 8, Ho = p.H / 2, Wo = p.W / 2;

  // all weight fragments of this lane: output channel nt * 16 + frow, tap row kh, elements fq * 8 .. + 8
  bf16x8 bw[7][4];
#pragma unroll
  for (int kh = 0; kh < 7; ++kh)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      bw[kh][nt] = *reinterpret_cast<const bf16x8*>(p.w + ((nt * 16 + frow) * 7 + kh) * 32 + fq * 8);

  // this wave's two MFMA row tiles: tile-local output row wave >> 1, columns (wave & 1) * 32 + 16 i + frow
  const int orow = wave >> 1, ocol = (wave & 1) * 32 + frow;
  const int afrag = (2 * orow) * HPITCH + ocol * 16 + fq * 16;          // + kh * HPITCH, + i * 256

  // halo pieces of this thread: piece j = tid + 512 i -> halo row j / 68, 16-byte column j % 68
  int hoff[NPC], goff[NPC];
  bool pok[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int j = tid + NTH * i;
    const int r = j / (HPITCH / 16), cc = j - r * (HPITCH / 16);
    pok[i] = r < HROWS;
    hoff[i] = r * HPITCH + cc * 16;
    goff[i] = (r * Wp) * 8 + cc * 16;                                   // bytes from the tile's first input pixel
  }
  auto tile_src = [&](int t) {
    const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, n = t / (p.tiles_x * p.tiles_y);
    return reinterpret_cast<const char*>(p.xpad) + (((long)n * Hp + 2 * ty * TRO) * Wp + 2 * tx * TCO) * 8;
  };

  float sm[4][4], sq[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) { sm[nt][j] = 0.f; sq[nt][j] = 0.f; }

  uint4 st[NPC];
  int t = blockIdx.x;
  if (t < p.ntiles) {
    const char* src = tile_src(t);
#pragma unroll
    for (int i = 0; i < NPC; ++i) st[i] = pok[i] ? *reinterpret_cast<const uint4*>(src + goff[i]) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NPC; ++i) if (pok[i]) *reinterpret_cast<uint4*>(smem + hoff[i]) = st[i];
  }
  __syncthreads();
  int buf = 0;
  for (; t < p.ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    const bool more = tn < p.ntiles;
    if (more) {
      const char* src = tile_src(tn);
#pragma unroll
      for (int i = 0; i < NPC; ++i) st[i] = pok[i] ? *reinterpret_cast<const uint4*>(src + goff[i]) : make_uint4(0, 0, 0, 0);
    }
    const char* const Hb = smem + buf * HBYTES + afrag;
    f32x4 acc[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { acc[nt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[nt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Hb + kh * HPITCH);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Hb + kh * HPITCH + 256);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        acc[nt][0] = EVK_MFMA_16x16x32(bw[kh][nt], a0, acc[nt][0], 0, 0, 0);
        acc[nt][1] = EVK_MFMA_16x16x32(bw[kh][nt], a1, acc[nt][1], 0, 0, 0);
      }
    }
    // lane holds y[pixel frow of row tile i][co = nt * 16 + fq * 4 + 0..3]
    {
      const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, n = t / (p.tiles_x * p.tiles_y);
      bf16_t* yb = p.y + ((((long)n * Ho + ty * TRO + orow) * Wo) + tx * TCO + ocol) * 64 + fq * 4;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const f32x4 v = acc[nt][i];
          *reinterpret_cast<uint2*>(yb + i * 16 * 64 + nt * 16) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
#pragma unroll
          for (int j = 0; j < 4; ++j) { sm[nt][j] += v[j]; sq[nt][j] += v[j] * v[j]; }
        }
    }
    if (more) {
      char* d = smem + (buf ^ 1) * HBYTES;
#pragma unroll
      for (int i = 0; i < NPC; ++i) if (pok[i]) *reinterpret_cast<uint4*>(d + hoff[i]) = st[i];
    }
    __syncthreads();
    buf ^= 1;
  }
  if (p.colstats) {
    float* prow = p.colstats + ((long)blockIdx.x * 8 + wave) * 2 * 64;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = row16_sum(sm[nt][j]); b[j] = row16_sum(sq[nt][j]); }
      if (frow == 0) {
        *reinterpret_cast<float4*>(prow + nt * 16 + fq * 4) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(prow + 64 + nt * 16 + fq * 4) = make_float4(b[0], b[1], b[2], b[3]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradient: dw[co][kh][e] += sum over output pixels of dy[px][co] * xpad[2 oy + kh][2 ox][e].  Same tiles; per tile the dy rows
// ([pixel][64 channels], pitch 160 B as in conv3x3.hip's weight gradient) and the input halo are staged once, pixels are the
// contraction index (transposing LDS reads for both operands: consecutive pixels of the halo are 16 B apart and their 64-byte windows
// overlap -- lanes that meet on an address are a broadcast, not a conflict).  Wave kh < 7 owns tap row kh: all 4 x 2 MFMA tiles of
// dw[.][kh][.] (the eighth wave only loads); a workgroup accumulates over all its tiles and leaves ONE slab [kh][workgroup][64][32],
// summed into dw by gemm.hip's split-K reduction.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int DPITCH = 160;
constexpr int DBYTES = TRO * TCO * DPITCH;                  // 40960
constexpr int HB_PAD = (HBYTES + 255) / 256 * 256;          // 14336
constexpr int WSTAGE = HB_PAD + DBYTES;                     // 55296
constexpr int WLDS = 2 * WSTAGE;                            // 110592
constexpr int NPD = TRO * TCO * 8 / NTH;                    // dy pieces per thread and tile (4)

struct StemWP {
  const bf16_t* xpad; const bf16_t* dy; float* slab;
  int N, H, W;
  int tiles_x, tiles_y, ntiles;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ bf16x8 frag2(const lds_char* lo_addr, const lds_char* hi_addr) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo_addr);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)hi_addr);
  const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(NTH, 2) void stem_wgrad_kernel(const StemWP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fq = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const int Hp = p.H + 6, Wp = p.W + 8, Ho = p.H / 2, Wo = p.W / 2;

  // pieces of a tile: halo as in the forward kernel, dy piece j = tid + 512 i -> tile pixel j >> 3, 16-byte chunk j & 7
  int hoff[NPC], goff[NPC];
  bool pok[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int j = tid + NTH * i;
    const int r = j / (HPITCH / 16), cc = j - r * (HPITCH / 16);
    pok[i] = r < HROWS;
    hoff[i] = r * HPITCH + cc * 16;
    goff[i] = (r * Wp) * 8 + cc * 16;
  }
  int doff[NPD], dgoff[NPD];
#pragma unroll
  for (int i = 0; i < NPD; ++i) {
    const int j = tid + NTH * i;
    const int px = j >> 3, ch = j & 7;
    doff[i] = HB_PAD + px * DPITCH + ch * 16;
    dgoff[i] = (((px >> 6) * Wo + (px & 63)) * 64 + ch * 8) * 2;       // bytes from the tile's first output pixel
  }
  auto x_src = [&](int t) {
    const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, n = t / (p.tiles_x * p.tiles_y);
    return reinterpret_cast<const char*>(p.xpad) + (((long)n * Hp + 2 * ty * TRO) * Wp + 2 * tx * TCO) * 8;
  };
  auto dy_src = [&](int t) {
    const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, n = t / (p.tiles_x * p.tiles_y);
    return reinterpret_cast<const char*>(p.dy) + ((((long)n * Ho + ty * TRO) * Wo) + tx * TCO) * 64 * 2;
  };
  uint4 sx0, sx1, sd0, sd1, sd2, sd3;
  static_assert(NPC == 2 && NPD == 4, "staging registers are spelled out");
  auto issue = [&](int t) {
    const char* xs = x_src(t);
    const char* ds = dy_src(t);
    sx0 = pok[0] ? *reinterpret_cast<const uint4*>(xs + goff[0]) : make_uint4(0, 0, 0, 0);
    sx1 = pok[1] ? *reinterpret_cast<const uint4*>(xs + goff[1]) : make_uint4(0, 0, 0, 0);
    sd0 = *reinterpret_cast<const uint4*>(ds + dgoff[0]);
    sd1 = *reinterpret_cast<const uint4*>(ds + dgoff[1]);
    sd2 = *reinterpret_cast<const uint4*>(ds + dgoff[2]);
    sd3 = *reinterpret_cast<const uint4*>(ds + dgoff[3]);
  };
  auto commit = [&](int stage) {
    char* d = smem + stage * WSTAGE;
    if (pok[0]) *reinterpret_cast<uint4*>(d + hoff[0]) = sx0;
    if (pok[1]) *reinterpret_cast<uint4*>(d + hoff[1]) = sx1;
    *reinterpret_cast<uint4*>(d + doff[0]) = sd0;
    *reinterpret_cast<uint4*>(d + doff[1]) = sd1;
    *reinterpret_cast<uint4*>(d + doff[2]) = sd2;
    *reinterpret_cast<uint4*>(d + doff[3]) = sd3;
  };

  // wave kh: acc[ct][et] = dw[co tile ct][kh][element tile et]
  f32x4 acc[4][2];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) { acc[ct][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[ct][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const lds_char* const L = (const lds_char*)(uintptr_t)(uint32_t)(uintptr_t)smem;
  const int kh = wave;
  // within a 32-pixel step logical k = 8 fq + q + 4 hi is pixel 16 hi + 4 fq + q (both operands): the lane's low group starts at pixel 4 fq + q4
  const int pl = 4 * fq + q4;
  const int dlo = HB_PAD + pl * DPITCH + pp * 8;                    // + (kg * 32) * DPITCH, + ct * 32; high group + 16 * DPITCH
  const int xlo = kh * HPITCH + pl * 16 + pp * 8;                   // + (2 row) * HPITCH + (col half) * 512, + et * 32; high group + 256

  int t = blockIdx.x;
  if (t < p.ntiles) { issue(t); commit(0); }
  __syncthreads();
  int buf = 0;
  for (; t < p.ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    const bool more = tn < p.ntiles;
    if (more) issue(tn);
    if (wave < 7) {
      const lds_char* const S = L + buf * WSTAGE;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {                              // tile row kg >> 1, columns (kg & 1) * 32 ..
        const lds_char* const da = S + dlo + kg * 32 * DPITCH;
        const lds_char* const xa = S + xlo + 2 * (kg >> 1) * HPITCH + (kg & 1) * 512;
        const bf16x8 x0 = frag2(xa, xa + 256), x1 = frag2(xa + 32, xa + 32 + 256);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const bf16x8 d = frag2(da + ct * 32, da + ct * 32 + 16 * DPITCH);
          acc[ct][0] = EVK_MFMA_16x16x32(x0, d, acc[ct][0], 0, 0, 0);
          acc[ct][1] = EVK_MFMA_16x16x32(x1, d, acc[ct][1], 0, 0, 0);
        }
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // slab [kh][workgroup][co][e]: lane holds co = ct * 16 + (lane & 15), e = et * 16 + 4 fq + 0..3
  if (wave < 7) {
    float* s0 = p.slab + ((long)kh * gridDim.x + blockIdx.x) * (64 * 32);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int et = 0; et < 2; ++et)
        *reinterpret_cast<float4*>(s0 + (ct * 16 + (lane & 15)) * 32 + et * 16 + fq * 4) = make_float4(acc[ct][et][0], acc[ct][et][1], acc[ct][et][2], acc[ct][et][3]);
  }
}

int stem_grid(int ntiles) {
  static const int g = evk_tunable("EVK_STEM_BLOCKS", 256);
  return ntiles < g ? ntiles : g;
}

}  // namespace

extern "C" {

int evk_stem_halo_supported(int32_t N, int32_t H, int32_t W) {
  static const int on = evk_tunable("EVK_STEM_HALO", 1);
  return on && N > 0 && H > 0 && W > 0 && (H / 2) % TRO == 0 && H % 2 == 0 && (W / 2) % TCO == 0 && W % 2 == 0 ? 1 : 0;
}

int64_t evk_stem_halo_part_bytes(int32_t N, int32_t H, int32_t W) {
  if (!evk_stem_halo_supported(N, H, W)) return 0;
  const int ntiles = N * (H / 2 / TRO) * (W / 2 / TCO);
  return (int64_t)stem_grid(ntiles) * 8 * 2 * 64 * (int64_t)sizeof(float);
}

int evk_stem_halo_fwd(const void* xpad, const void* wp, void* y, int32_t N, int32_t H, int32_t W, float* part, int64_t part_bytes, int32_t* nblk,
                      evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(xpad && wp && y, "stem_halo_fwd: null operand");
  EVK_REQUIRE(evk_stem_halo_supported(N, H, W), "stem_halo_fwd: H / 2 must be a multiple of 4 and W / 2 of 64 (got %dx%d)", H, W);
  EVK_REQUIRE(((reinterpret_cast<uintptr_t>(xpad) | reinterpret_cast<uintptr_t>(wp) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "stem_halo_fwd: 16-byte alignment");
  StemP p{};
  p.xpad = (const bf16_t*)xpad; p.w = (const bf16_t*)wp; p.y = (bf16_t*)y;
  p.N = N; p.H = H; p.W = W;
  p.tiles_x = W / 2 / TCO; p.tiles_y = H / 2 / TRO; p.ntiles = N * p.tiles_x * p.tiles_y;
  const int grid = stem_grid(p.ntiles);
  p.colstats = part;
  if (part) {
    EVK_REQUIRE(nblk && part_bytes >= evk_stem_halo_part_bytes(N, H, W), "stem_halo_fwd: statistics buffer too small");
    *nblk = grid * 8;
  }
  evk_prof_tag(N * (H / 2) * (W / 2), 64, 224, 1, EVK_A_CONV, EVK_B_PLAIN);
  ProfScope ps(EVK_FAM_GEMM, s, 2.0 * N * (H / 2) * (double)(W / 2) * 64 * 224);
  hipLaunchKernelGGL(stem_fwd_kernel, dim3(grid), dim3(NTH), 0, s, p);
  return evk_check_launch("stem_fwd_kernel");
}

int64_t evk_stem_halo_wgrad_ws_bytes(int32_t N, int32_t H, int32_t W) {
  if (!evk_stem_halo_supported(N, H, W)) return 0;
  const int ntiles = N * (H / 2 / TRO) * (W / 2 / TCO);
  return 7LL * stem_grid(ntiles) * 64 * 32 * (int64_t)sizeof(float);
}

int evk_stem_halo_wgrad(const void* dy, const void* xpad, float* dwp, int32_t N, int32_t H, int32_t W, void* ws, int64_t ws_bytes, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dy && xpad && dwp && ws, "stem_halo_wgrad: null operand / workspace");
  EVK_REQUIRE(evk_stem_halo_supported(N, H, W), "stem_halo_wgrad: H / 2 must be a multiple of 4 and W / 2 of 64 (got %dx%d)", H, W);
  EVK_REQUIRE(ws_bytes >= evk_stem_halo_wgrad_ws_bytes(N, H, W), "stem_halo_wgrad: workspace too small");
  EVK_REQUIRE(((reinterpret_cast<uintptr_t>(xpad) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dwp) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0,
              "stem_halo_wgrad: 16-byte alignment");
  StemWP p{};
  p.xpad = (const bf16_t*)xpad; p.dy = (const bf16_t*)dy; p.slab = (float*)ws;
  p.N = N; p.H = H; p.W = W;
  p.tiles_x = W / 2 / TCO; p.tiles_y = H / 2 / TRO; p.ntiles = N * p.tiles_x * p.tiles_y;
  const int grid = stem_grid(p.ntiles);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WLDS);
    attr_done = true;
  }
  evk_prof_tag(64, 32, N * (H / 2) * (W / 2), 7, EVK_A_KSTR, EVK_B_WGATHER);
  ProfScope ps(EVK_FAM_GEMM, s, 2.0 * N * (H / 2) * (double)(W / 2) * 64 * 224);
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(grid), dim3(NTH), WLDS, s, p);
  if (int e = evk_check_launch("stem_wgrad_kernel")) return e;
  // dwp[co][kh][e] += sum over the workgroups' slabs [kh][workgroup][64][32]
  return evk_splitk_reduce_launch(p.slab, dwp, 64L * 32, 64, 32, grid, 7, 224, 0, 32, 7, s);
}

}  // extern "C"
