// conv3x3.hip -- 3x3 stride-1 "same" convolution (torchvision Bottleneck.conv2 as driven by modules/visual_extractor.py:30-38)
// with the INPUT HALO TILE resident in LDS: y[n][oy][ox][co] = sum_{kh,kw,ci} x[n][oy+kh-1][ox+kw-1][ci] * w[co][kh][kw][ci].
//
// Why a second kernel beside the implicit-GEMM path of gemm.hip (A_CONV loader): those tile kernels are bound by the bytes a CU can
// FILL into LDS (~30 GB/s per CU whether the lines come from L2 or HBM), and the im2col gather fills every
// input pixel nine times -- layer3 (64 x 24 x 24, 256 -> 256): 680 MB of fills for 38 MB of operands, 79-88 us.  Here a workgroup
// owns R whole image rows (<= 320 output pixels) x 128 output channels and walks K as (64-channel chunk, tap): the (R+2) x (W+2)
// halo of one chunk is filled ONCE and the nine taps read it at shifted rows, so per chunk the fills are one halo (<= 56 KB) plus nine
// 16 KB weight tiles for 2 * 320 * 128 * 576 flop -- about twice the flop per filled byte of the 128 x 128 x 64 tile -- and
// layer3 is exactly 256 workgroups of equal work, one per CU.
//
// Geometry: the N images are stacked into one "tall" image of N*H rows; a tile is R consecutive tall rows (it may cross image
// boundaries).  In LDS the halo is addressed in PADDED coordinates -- one zero row above every image (padded row P = n*(H+1)+y+1,
// rows with P % (H+1) == 0 are zero) and a zero column left and right -- so the three input rows of any output row are
// consecutive halo rows whether or not the tile crosses an image boundary, and every tap is a constant row offset.
// LDS rows are 128 B (64 channels), 16-byte chunk index XOR (row & 7) as in gemm.hip (conflict-free ds_read_b128).
// Pipeline (register staged, cdna_hip_programming.md T14): at the top of K-step s the loads issued at the top of step s-1 (weight
// tile s+2 into one of three stages, one 16-byte piece per thread of the next chunk's halo) are written to LDS and the loads of step
// s+3 are issued; the MFMA fragments are read half a step ahead of their use (k-step 0 of step s+1 during k-step 1 of step s), so the
// 40 MFMAs that follow the step's one barrier wait for nothing issued after it.  8 waves = 4 (pixels) x 2 (channels), each 80 px x 64 ch = 5 x 4 MFMA tiles.
// Epilogues = those of the tile path: per-channel sum / sum of squares partials (forward: batch-norm statistics), or residual +
// ReLU gate + gate statistics (the data gradient run as a forward convolution over flipped weights, conv.hip).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

unsigned long long* g_stamps = nullptr;      // evk_conv3x3_halo_debug_stamps

constexpr int NTH = 512;
constexpr int WM = 4, WN = 2, MI = 5;
constexpr int TP = 16 * MI * WM;             // 320 output pixels per workgroup
constexpr int NBST = 3;                      // weight-tile stages
// Two configurations: 128 output channels per workgroup (4 MFMA tiles per wave, halo buffers of 432 pixels) and 64 (2 tiles per wave:
// the 64-channel convolutions of layer1, whose 96-pixel rows need the 512-pixel halo the smaller weight stages leave room for).
template <int NI_, int HALO_> struct Cfg {
  static constexpr int NI = NI_, HALO_MAX = HALO_;
  static constexpr int TN = 16 * NI * WN;                       // output channels per workgroup
  static constexpr int NPIECE = (HALO_MAX * 8 + NTH - 1) / NTH; // 16-byte halo pieces per thread and chunk (the last one partly beyond the buffer: masked)
  static constexpr int A_BYTES = HALO_MAX * 128;
  static constexpr int B_BYTES = TN * 128;
  static constexpr int LDS_BYTES = 2 * A_BYTES + NBST * B_BYTES;
  static_assert(NPIECE <= 8, "the halo pieces of the next chunk are loaded one per K-step, before the chunk's last step");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(NI == 4 || NI == 2, "one or two 64-row weight pieces per thread");
};
using Cfg128 = Cfg<4, 432>;                  // 159744 B of LDS
using Cfg64 = Cfg<2, 512>;                   // 155648 B

struct C3P {
  const bf16_t* x; const bf16_t* w; bf16_t* y;
  int N, H, W, C, Co;
  int R, tilesM, tilesN;
  float* colstats;                 // [tilesM * WM][2][Co] or null
  const bf16_t* resid; long ldr;   // optional, added before the gate
  const bf16_t* gate; long ldg;    // optional ReLU gate (post-ReLU forward value of the tensor y belongs to)
  float* gatestats;                // [tilesM * WM][2][Co] or null (needs gate)
  unsigned long long* stamps;      // diagnostic (evk_conv3x3_halo_debug_stamps): per workgroup {memtime x 4, memrealtime x 2} or null
  float inv_w, inv_w2, inv_h, inv_h1;
  int kmul;                        // 1; 0 = timing probe (EVK_C3_PROBE=1): every in-loop load reads the step-0 addresses (cache hits, wrong results)
  const float* scale; const float* bias; int relu;     // inference (eval-mode batch norm): y = relu?(conv * scale[co] + bias[co] (+ resid)); no gate, no statistics
};

// a / b for 0 <= a < 2^22, b > 0, inv = 1 / b (integer division proper costs ~40 instructions per use on this ISA)
__device__ __forceinline__ int fdiv(int a, int b, float inv) {
  int q = (int)((float)a * inv);
  const int r = a - q * b;
  if (r < 0) --q; else if (r >= b) ++q;
  return q;
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
  return v + __builtin_bit_cast(float, t);
}
// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4)
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  return dpp_add<0x140>(v);
}

template <class CF>
__global__ __launch_bounds__(NTH, 2) void conv3x3_halo_kernel(const C3P p) {
  constexpr int NI = CF::NI, TN = CF::TN, HALO_MAX = CF::HALO_MAX, NPIECE = CF::NPIECE, A_BYTES = CF::A_BYTES, B_BYTES = CF::B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Abuf = smem;
  char* const Bst = smem + 2 * A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int frow = lane & 15, fq = lane >> 4;
  if (p.stamps && tid == 0) { p.stamps[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime(); p.stamps[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime(); }

  // blocks b, b + 8, ... share an XCD: give each XCD a contiguous run of tiles, the channel tiles of one pixel tile adjacent
  // (they read the same halo: the second one finds it in that L2)
  const int bid = blockIdx.x, nwg = gridDim.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tm = wg / p.tilesN, tn = wg - tm * p.tilesN;

  const int H = p.H, W = p.W, C = p.C;
  const int W2 = W + 2, H1 = H + 1;
  const int TR = p.N * H;
  const int g0 = tm * p.R;
  const int reff = min(p.R, TR - g0);
  const int npx = reff * W;
  const int n_first = fdiv(g0, H, p.inv_h);
  const int P0 = n_first * H1 + (g0 - n_first * H);           // padded row of tall row g0, minus one
  const int glast = g0 + reff - 1;
  const int n_last = fdiv(glast, H, p.inv_h);
  const int hpx = (n_last * H1 + (glast - n_last * H) + 3 - P0) * W2;   // halo pixels of this tile (<= HALO_MAX by the host's choice of R)

  // the weight tiles of steps 0, 1 and 2 are requested before the index arithmetic below (their addresses need none of it)
  const int co0 = tn * TN;
  const char* const wsrc = reinterpret_cast<const char*>(p.w) + ((long)(co0 + (tid >> 3)) * 9 * C + (tid & 7) * 8) * 2;
  const long wrow64 = (long)64 * 9 * C * 2;
  uint4 rb0, rb1, ra;
  uint4 pb0[2], pb1[2];
  pb0[0] = *reinterpret_cast<const uint4*>(wsrc);
  if constexpr (NI == 4) pb0[1] = *reinterpret_cast<const uint4*>(wsrc + wrow64);
  pb1[0] = *reinterpret_cast<const uint4*>(wsrc + (long)C * 2 * p.kmul);
  if constexpr (NI == 4) pb1[1] = *reinterpret_cast<const uint4*>(wsrc + (long)C * 2 * p.kmul + wrow64);
  rb0 = *reinterpret_cast<const uint4*>(wsrc + (long)C * 4 * p.kmul);
  if constexpr (NI == 4) rb1 = *reinterpret_cast<const uint4*>(wsrc + (long)C * 4 * p.kmul + wrow64);

  // halo pieces of this thread: piece i covers halo pixel (tid >> 3) + 64 i, 16-byte chunk tid & 7 of the 64-channel row
  unsigned aoff[NPIECE];
  unsigned amask = 0, inbuf = 0;                 // amask: the piece holds data (else zeros); inbuf: its halo pixel lies inside the buffer
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) {
    const int hp = (tid >> 3) + 64 * i;
    const int hr = fdiv(hp, W2, p.inv_w2), hx = hp - hr * W2;
    const int P = P0 + hr;
    const int n = fdiv(P, H1, p.inv_h1), yy = P - n * H1;
    const bool valid = hp < hpx && yy != 0 && hx >= 1 && hx <= W && n < p.N;
    aoff[i] = valid ? (unsigned)(((((long)n * H + (yy - 1)) * W + (hx - 1)) * C + (tid & 7) * 8) * 2) : 0u;
    amask |= (valid ? 1u : 0u) << i;
    inbuf |= (hp < HALO_MAX ? 1u : 0u) << i;
  }
  const int adst = (tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 3) & 7)) << 4);     // + i * 8192

  // halo row of each of this lane's MFMA rows (centre tap); rows beyond the tile read pixel 0 and are masked in the epilogue
  int hb[MI];
#pragma unroll
  for (int im = 0; im < MI; ++im) {
    int pp = wm * (16 * MI) + im * 16 + frow;
    if (pp >= npx) pp = 0;
    const int j = fdiv(pp, W, p.inv_w), xx = pp - j * W;
    const int g = g0 + j;
    const int n = fdiv(g, H, p.inv_h), y = g - n * H;
    hb[im] = (n * H1 + y + 1 - P0) * W2 + xx + 1;
  }

  const int bdst = adst;                                                         // same (row, chunk) -> byte mapping; + 8192 for rows 64..127
  const char* const xsrc = reinterpret_cast<const char*>(p.x);

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int baddr[NI];
#pragma unroll
  for (int in = 0; in < NI; ++in) {
    const int row = wn * (16 * NI) + in * 16 + frow;
    baddr[in] = row * 128 + ((fq ^ (row & 7)) << 4);
  }
  const int nchunk = C >> 6;
  const int nsteps = nchunk * 9;

  auto loadB = [&](int c, int t) {
    const char* s = wsrc + (long)(t * C + c * 64) * 2 * p.kmul;
    rb0 = *reinterpret_cast<const uint4*>(s);
    if constexpr (NI == 4) rb1 = *reinterpret_cast<const uint4*>(s + wrow64);
  };
  auto storeB = [&](int stage) {
    char* d = Bst + stage * B_BYTES + bdst;
    *reinterpret_cast<uint4*>(d) = rb0;
    if constexpr (NI == 4) *reinterpret_cast<uint4*>(d + 8192) = rb1;
  };
  auto storeA = [&](char* buf, int i, const uint4& v) {      // only the last piece can lie beyond the buffer
    if ((i + 1) * 64 <= HALO_MAX || ((inbuf >> i) & 1u))
      *reinterpret_cast<uint4*>(buf + adst + i * 8192) = ((amask >> i) & 1u) ? v : make_uint4(0, 0, 0, 0);
  };
  // fragment addresses of tap t inside a halo buffer (k-step 1 = the same address with byte bit 6 flipped: chunk index ^ 4).  The empty
  // asm keeps the compiler from hoisting the 9 x MI addresses of all taps out of the chunk loop (loop-invariant: 45 registers, spills)
  auto tap_addr = [&](int t, int (&a)[MI]) {
    const int toff = (t / 3 - 1) * W2 + (t % 3 - 1);
#pragma unroll
    for (int im = 0; im < MI; ++im) {
      int h = hb[im];
      asm volatile("" : "+v"(h));
      const int hr = h + toff;
      a[im] = hr * 128 + ((fq ^ (hr & 7)) << 4);
    }
  };

  // prologue: the whole halo of chunk 0 and the weight tiles of steps 0 and 1 into LDS (the weight tile of step 2 stays in registers);
  // every load was issued before the first wait
  {
    uint4 v[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) v[i] = *reinterpret_cast<const uint4*>(xsrc + aoff[i]);
    *reinterpret_cast<uint4*>(Bst + bdst) = pb0[0];
    if constexpr (NI == 4) *reinterpret_cast<uint4*>(Bst + bdst + 8192) = pb0[1];
    *reinterpret_cast<uint4*>(Bst + B_BYTES + bdst) = pb1[0];
    if constexpr (NI == 4) *reinterpret_cast<uint4*>(Bst + B_BYTES + bdst + 8192) = pb1[1];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) storeA(Abuf, i, v[i]);
  }
  __syncthreads();

  // K-step s = 9 c + t multiplies tap t of chunk c: weight stage s % 3 = t % 3, halo buffer c & 1.  Per step, in program order:
  //   registers -> LDS (weight tile s + 2, halo piece t - 1 of chunk c + 1: both loaded at the top of step s - 1), the loads of step
  //   s + 3 / the next halo piece, the k-step-1 fragments of step s, MFMAs of k-step 0 (fragments read during step s - 1), the
  //   k-step-0 fragments of step s + 1 (its weight tile was written at the top of step s - 1), MFMAs of k-step 1, barrier.
  // So the MFMAs that follow a barrier never wait for an LDS write or read issued after it.
  bf16x8 af0[MI], bf0[NI], af1[MI], bf1[NI];
  int acur[MI];
  tap_addr(0, acur);
#pragma unroll
  for (int im = 0; im < MI; ++im) af0[im] = *reinterpret_cast<const bf16x8*>(Abuf + acur[im]);
#pragma unroll
  for (int in = 0; in < NI; ++in) bf0[in] = *reinterpret_cast<const bf16x8*>(Bst + baddr[in]);

  // one chunk = nine steps, fully unrolled and branch-free: MORE (another chunk follows) is a compile-time flag, so every
  // "is there a step s + k" test folds (with MORE they all exist; in the last chunk they depend on t alone)
  auto chunk = [&](auto more_tag, int c) {
    constexpr bool MORE = decltype(more_tag)::value;
    const char* const As = Abuf + (c & 1) * A_BYTES;
    char* const Anext = Abuf + ((c + 1) & 1) * A_BYTES;
    const char* const xnext = xsrc + (long)(c + 1) * 128 * p.kmul;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (MORE || t + 2 < 9) storeB((t + 2) % 3);
      if (MORE && t >= 1 && t - 1 < NPIECE) storeA(Anext, t - 1, ra);
      if (MORE || t + 3 < 9) {
        if (t < 6) loadB(c, t + 3); else loadB(c + 1, t - 6);
      }
      if (MORE && t < NPIECE) ra = *reinterpret_cast<const uint4*>(xnext + aoff[t]);

      const char* const Bs = Bst + (t % 3) * B_BYTES;
#pragma unroll
      for (int im = 0; im < MI; ++im) af1[im] = *reinterpret_cast<const bf16x8*>(As + (acur[im] ^ 64));
#pragma unroll
      for (int in = 0; in < NI; ++in) bf1[in] = *reinterpret_cast<const bf16x8*>(Bs + (baddr[in] ^ 64));
#pragma unroll
      for (int in = 0; in < NI; ++in)
#pragma unroll
        for (int im = 0; im < MI; ++im) acc[in][im] = EVK_MFMA_16x16x32(bf0[in], af0[im], acc[in][im], 0, 0, 0);
      if (MORE || t + 1 < 9) {
        const char* const An = t < 8 ? As : Anext;
        const char* const Bn = Bst + ((t + 1) % 3) * B_BYTES;
        tap_addr(t < 8 ? t + 1 : 0, acur);
#pragma unroll
        for (int im = 0; im < MI; ++im) af0[im] = *reinterpret_cast<const bf16x8*>(An + acur[im]);
#pragma unroll
        for (int in = 0; in < NI; ++in) bf0[in] = *reinterpret_cast<const bf16x8*>(Bn + baddr[in]);
      }
#pragma unroll
      for (int in = 0; in < NI; ++in)
#pragma unroll
        for (int im = 0; im < MI; ++im) acc[in][im] = EVK_MFMA_16x16x32(bf1[in], af1[im], acc[in][im], 0, 0, 0);
      __syncthreads();
    }
  };
  if (p.stamps && tid == 0) p.stamps[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();
  for (int c = 0; c + 1 < nchunk; ++c) chunk(std::true_type{}, c);
  chunk(std::false_type{}, nchunk - 1);
  if (p.stamps && tid == 0) p.stamps[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime();

  // ---- epilogue: lane holds y[m][n0 .. n0+3], m = tile pixel wm*80 + im*16 + frow, n0 = co0 + wn*64 + in*16 + fq*4 ----
  const long m0 = (long)g0 * W;
  const int Co = p.Co;
  bool rowok[MI];
#pragma unroll
  for (int im = 0; im < MI; ++im) {
    rowok[im] = wm * (16 * MI) + im * 16 + frow < npx;
    if (!rowok[im]) {
#pragma unroll
      for (int in = 0; in < NI; ++in) acc[in][im] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (p.colstats) {
    float* prow = p.colstats + ((long)(tm * WM + wm)) * 2 * Co;
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
      const int n0 = co0 + wn * (16 * NI) + in * 16 + fq * 4;
      if (frow == 0) {
        *reinterpret_cast<float4*>(prow + n0) = make_float4(sm[0], sm[1], sm[2], sm[3]);
        *reinterpret_cast<float4*>(prow + Co + n0) = make_float4(sq[0], sq[1], sq[2], sq[3]);
      }
    }
  }
  if (p.bias) {
    // inference: eval-mode batch norm as scale / shift (+ identity), ReLU
#pragma unroll
    for (int in = 0; in < NI; ++in) {
      const int n0 = co0 + wn * (16 * NI) + in * 16 + fq * 4;
      const float4 bb = *reinterpret_cast<const float4*>(p.bias + n0);
      const float4 sc = *reinterpret_cast<const float4*>(p.scale + n0);
#pragma unroll
      for (int im = 0; im < MI; ++im) {
        if (!rowok[im]) continue;
        const long m = m0 + wm * (16 * MI) + im * 16 + frow;
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
        *reinterpret_cast<uint2*>(p.y + m * Co + n0) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      }
    }
    return;
  }
  if (!p.gate && !p.resid) {
    // forward convolution: nothing but the rounded result leaves -- one row pointer per MFMA tile row, no per-tile branches
    char* const cb = reinterpret_cast<char*>(p.y) + ((m0 + wm * (16 * MI) + frow) * Co + co0 + wn * (16 * NI) + fq * 4) * 2;
    const long rstep = 16L * Co * 2;
#pragma unroll
    for (int im = 0; im < MI; ++im) {
      if (!rowok[im]) continue;
#pragma unroll
      for (int in = 0; in < NI; ++in)
        *reinterpret_cast<uint2*>(cb + im * rstep + in * 32) =
            make_uint2(pack2bf(acc[in][im][0], acc[in][im][1]), pack2bf(acc[in][im][2], acc[in][im][3]));
    }
    if (p.stamps && tid == 0) { p.stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memtime(); p.stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime(); }
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
    const long m = m0 + wm * (16 * MI) + im * 16 + frow;
#pragma unroll
    for (int in = 0; in < NI; ++in) {
      const int n0 = co0 + wn * (16 * NI) + in * 16 + fq * 4;
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
      *reinterpret_cast<uint2*>(p.y + m * Co + n0) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
    }
  }
  if (p.gatestats) {
    float* prow = p.gatestats + ((long)(tm * WM + wm)) * 2 * Co;
#pragma unroll
    for (int in = 0; in < NI; ++in) {
      float a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = row16_sum(gs[in][j >> 1][j & 1]); b[j] = row16_sum(gz[in][j >> 1][j & 1]); }
      const int n0 = co0 + wn * (16 * NI) + in * 16 + fq * 4;
      if (frow == 0) {
        *reinterpret_cast<float4*>(prow + n0) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(prow + Co + n0) = make_float4(b[0], b[1], b[2], b[3]);
      }
    }
  }
  if (p.stamps && tid == 0) { p.stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memtime(); p.stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime(); }
}

// most halo rows (padded coordinates) any tile of R tall rows needs: the phases g0 % H repeat after at most H tiles
int halo_rows_max(int TR, int H, int R) {
  const int H1 = H + 1;
  const int tiles = (int)cdiv(TR, R), lim = tiles < H ? tiles : H;
  int best = 0;
  for (int k = 0; k < lim; ++k) {
    const int g0 = k * R, gl = (g0 + R < TR ? g0 + R : TR) - 1;
    const int rows = (gl / H) * H1 + gl % H + 3 - ((g0 / H) * H1 + g0 % H);
    if (rows > best) best = rows;
  }
  return best;
}

// rows per tile: the most whole rows whose pixels fit the tile and whose halo fits the buffer; 0 = not tileable
int choose_rows(int N, int H, int W, int halo_max) {
  for (int R = TP / W; R >= 1; --R)
    if (halo_rows_max(N * H, H, R) * (W + 2) <= halo_max) return R;
  return 0;
}
// 128-channel tiles wherever the output has them; the 64-channel configuration for Co % 128 == 64
inline bool wide_tiles(int Co) { return Co % Cfg128::TN == 0; }
inline int halo_max_for(int Co) { return wide_tiles(Co) ? Cfg128::HALO_MAX : Cfg64::HALO_MAX; }



// ================================================================================================================================
// Weight gradient of the same convolution: dw[co][kh][kw][ci] += sum_px dy[px][co] * x[px + (kh-1, kw-1)][ci].
// The tile path (gemm.hip: A_KSTR x B_WGATHER, the tap as batch index) fills both operands once per tap and per 128 x 128 tile: 680 MB
// of LDS fills on layer3, 140-157 us.  Here a workgroup owns a slice of the pixels and one 64 (co) x 64 (ci) block of the filter and
// accumulates ALL NINE taps in registers (9 x 16 MFMA tiles over 8 waves = 72 accumulator registers): per pixel it fills 128 B of dy and
// ~1.35 x 128 B of the x halo for 2 * 9 * 64 * 64 flop -- four times the flop per filled byte -- and the taps read the halo at
// shifted rows.  Pixels are the contraction index, so both operands are staged as they lie in memory ([pixel][64 channels]) and the MFMA
// fragments come from the transposing LDS read (ds_read_b64_tr_b16, cdna_hip_programming.md T10).  LDS rows have a pitch of 160 B
// (5 x 32 B): the eight consecutive pixel rows one 32-lane half of such a read touches fall into eight different 32-byte bank groups
// WITHOUT an XOR swizzle, so the address of a shifted tap is the centre address plus a constant -- the kw shifts are immediate offsets
// of the read, the kh shifts one add each.  (Within a 32-pixel MFMA step logical k = 8 fq + q + 4 hi is pixel 16 hi + 4 fq + q for
// both operands, which makes the eight rows of a half consecutive; any pixel permutation common to both operands is a valid contraction
// order.)  The K-slices leave as f32 slabs [tap][split][co][ci] and gemm.hip's split-K reduction adds them into dw.
// ================================================================================================================================
namespace wgk {

__device__ uint4 g_zero16;        // zero-initialised

constexpr int PITCH = 160;
constexpr int STAGE_ROWS = 464;                       // dy rows (tile pixels padded to 32) + halo rows of one tile
constexpr int STAGE_BYTES = STAGE_ROWS * PITCH;       // 74240
constexpr int LDS_BYTES = 2 * STAGE_BYTES;            // 148480 (wgk::)
constexpr int NPC = (STAGE_ROWS * 8 + NTH - 1) / NTH; // 16-byte pieces per thread and tile (8)

struct W3P {
  const bf16_t* dy; const bf16_t* x; float* slab;
  int N, H, W, Co, Ci;
  int R, tpxp;                // tall rows per tile; tile pixels rounded up to 32
  int rps, nsplit;            // tall rows per K-slice (a multiple of R), slices
  int pairs_ci, npairs;       // 64-channel blocks of ci, of (co, ci)
  float inv_w, inv_w2, inv_h, inv_h1;
  const void* zeros;           // 16 bytes of zeros in device memory: what a piece without data is loaded from
  unsigned long long* stamps;  // diagnostic, as in the forward kernel
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) char lds_char;       // 32-bit LDS pointers: constant offsets fold into the read's offset field
__device__ __forceinline__ bf16x8 frag2(const lds_char* lo_addr, const lds_char* hi_addr) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo_addr);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)hi_addr);
  const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(NTH, 2) void conv3x3_wgrad_halo_kernel(const W3P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fq = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const int wci = wave & 3, wco = (wave >> 2) * 2;        // this wave's 16-channel ci tile and the first of its two co tiles
  if (p.stamps && tid == 0) { p.stamps[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime(); p.stamps[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime(); }

  // the blocks of one K-slice (they read the same pixels) are consecutive on one XCD
  const int bid = blockIdx.x, nwg = gridDim.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int split = wg / p.npairs, pair = wg - split * p.npairs;
  const int cot = pair / p.pairs_ci, cit = pair - cot * p.pairs_ci;

  // R divides H (the host's choice): every tile is R whole rows of ONE image, tiles_per_image = H / R of them per image
  const int H = p.H, W = p.W, W2 = W + 2, TR = p.N * H, R = p.R, TPXP = p.tpxp;
  const int row_begin = split * p.rps;
  const int row_end = min(TR, row_begin + p.rps);
  const int ntiles = row_end > row_begin ? (row_end - row_begin) / R : 0;
  const int full_px = R * W;

  // This thread's pieces of a tile: stage row (tid >> 3) + 64 i, 16-byte chunk tid & 7.  Everything about a piece is known up front:
  // its byte offset from the tile's first pixel, whether it is a dy row or a halo row, whether it can ever hold data; per tile only the
  // top / bottom halo rows switch off when the tile touches the image's edge.  Pieces without data are loaded from a 16-byte block of
  // zeros, so nothing is masked on the way into LDS.
  const int pc = tid & 7;
  int offs[NPC];                     // signed byte offset from the tile origin (dy: row ri; halo: pixel (hr - 1, hx - 1))
  unsigned isdy = 0, smask = 0, topmask = 0, botmask = 0;
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int ri = (tid >> 3) + 64 * i;
    if (ri < TPXP) {
      offs[i] = ri * p.Co * 2;
      isdy |= 1u << i;
      smask |= (ri < full_px ? 1u : 0u) << i;
    } else {
      const int hp = ri - TPXP;
      const int hr = fdiv(hp, W2, p.inv_w2), hx = hp - hr * W2;
      offs[i] = ((hr - 1) * W + (hx - 1)) * p.Ci * 2;
      smask |= ((hr <= R + 1 && hx >= 1 && hx <= W) ? 1u : 0u) << i;
      topmask |= (hr == 0 ? 1u : 0u) << i;
      botmask |= (hr == R + 1 ? 1u : 0u) << i;
    }
  }
  const int ldst = (tid >> 3) * PITCH + pc * 16;       // + i * 64 * PITCH
  const char* const dyb = reinterpret_cast<const char*>(p.dy) + ((long)cot * 64 + pc * 8) * 2;
  const char* const xb = reinterpret_cast<const char*>(p.x) + ((long)cit * 64 + pc * 8) * 2;
  const char* const zsrc = reinterpret_cast<const char*>(p.zeros);

  // the eight staged pieces are eight named register sets (an array captured by the lambdas below ends up in scratch memory, and the
  // store to scratch waits for every load at once)
  static_assert(NPC == 8, "the staging registers are spelled out for eight pieces");
  uint4 st0, st1, st2, st3, st4, st5, st6, st7;
#define EVK_C3_PIECES(M) M(0, st0) M(1, st1) M(2, st2) M(3, st3) M(4, st4) M(5, st5) M(6, st6) M(7, st7)
  int yi = row_begin % H;            // row (inside its image) of the tile whose loads are issued next: advanced by R per tile
  auto issue = [&](int tile) {
    const long m0 = (long)(row_begin + tile * R) * W;
    const unsigned vm = smask & ~(yi == 0 ? topmask : 0u) & ~(yi + R == H ? botmask : 0u);
    const char* const tdy = dyb + m0 * p.Co * 2;
    const char* const tx = xb + m0 * p.Ci * 2;
    auto ld = [&](int i) {
      const char* src = (((isdy >> i) & 1u) ? tdy : tx) + offs[i];
      return *reinterpret_cast<const uint4*>(((vm >> i) & 1u) ? src : zsrc);
    };
#define EVK_C3_LD(i, r) r = ld(i);
    EVK_C3_PIECES(EVK_C3_LD)
#undef EVK_C3_LD
    yi += R;
    if (yi >= H) yi = 0;
  };
  auto commit = [&](int stage) {
    char* d = smem + stage * STAGE_BYTES + ldst;
#define EVK_C3_ST(i, r) \
    if ((i + 1) * 64 <= STAGE_ROWS || (tid >> 3) + 64 * i < STAGE_ROWS) *reinterpret_cast<uint4*>(d + i * 64 * PITCH) = r;
    EVK_C3_PIECES(EVK_C3_ST)
#undef EVK_C3_ST
  };

  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const lds_char* const L = (const lds_char*)(uintptr_t)(uint32_t)(uintptr_t)smem;
  const int doff = (4 * fq + q4) * PITCH + wco * 32 + pp * 8;                  // dy fragment, MFMA step 0, first co tile
  const int hoff = TPXP * PITCH + wci * 32 + pp * 8;                           // halo row 0, this wave's ci tile
  const int rowb = W2 * PITCH;
  // One 32-pixel MFMA step = 2 dy fragments (the wave's two co tiles) + 9 x fragments (its ci tile at the nine taps) -> 18 MFMAs.
  // The fragments of step k + 1 are read while the MFMAs of step k run (two register sets, the loop unrolled by two).
  struct Frags { bf16x8 dy[2]; bf16x8 x[9]; };
  auto load = [&](Frags& f, const lds_char* da, const lds_char* xl, const lds_char* xh) {
    f.dy[0] = frag2(da, da + 16 * PITCH);
    f.dy[1] = frag2(da + 32, da + 16 * PITCH + 32);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const lds_char* const bl = xl + (kh - 1) * rowb;
      const lds_char* const bh = xh + (kh - 1) * rowb;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) f.x[kh * 3 + kw] = frag2(bl + (kw - 1) * PITCH, bh + (kw - 1) * PITCH);
    }
  };
  auto mac = [&](const Frags& f) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc[t][0] = EVK_MFMA_16x16x32(f.x[t], f.dy[0], acc[t][0], 0, 0, 0);
      acc[t][1] = EVK_MFMA_16x16x32(f.x[t], f.dy[1], acc[t][1], 0, 0, 0);
    }
  };
  // (pixel row j, column x) of the lane's two pixel groups at MFMA step 0, and the advance per step (32 pixels)
  int j0[2], x0[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int px = 4 * fq + q4 + 16 * h;
    j0[h] = fdiv(px, W, p.inv_w);
    x0[h] = px - j0[h] * W;
  }
  const int dj = 32 / W, dx = 32 - dj * W;
  const int full_kgs = (full_px + 31) >> 5;
  auto compute = [&](int stage) {
    const lds_char* const S = L + stage * STAGE_BYTES;
    int j[2] = {j0[0], j0[1]}, x[2] = {x0[0], x0[1]};
    const lds_char* da = S + doff;
    auto fetch = [&](Frags& f) {                           // fragments of the next step; advances (j, x) and the dy address
      int row[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        row[h] = (min(j[h], R - 1) + 1) * W2 + x[h] + 1;     // pixels beyond the tile multiply zero dy rows: any staged row will do
        x[h] += dx; j[h] += dj;
        if (x[h] >= W) { x[h] -= W; ++j[h]; }
      }
      load(f, da, S + hoff + row[0] * PITCH, S + hoff + row[1] * PITCH);
      da += 32 * PITCH;
    };
    Frags fa, fb;
    fetch(fa);
    int kg = 0;
    while (true) {
      if (kg + 1 < full_kgs) fetch(fb);
      mac(fa);
      if (++kg >= full_kgs) break;
      if (kg + 1 < full_kgs) fetch(fa);
      mac(fb);
      if (++kg >= full_kgs) break;
    }
  };

  if (ntiles > 0) {
    issue(0);
    commit(0);
    __syncthreads();
    if (p.stamps && tid == 0) p.stamps[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();
    for (int tile = 0; tile < ntiles; ++tile) {
      const bool more = tile + 1 < ntiles;
      if (more) issue(tile + 1);
      compute(tile & 1);
      if (more) commit((tile + 1) & 1);
      __syncthreads();
    }
    if (p.stamps && tid == 0) p.stamps[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime();
  }
#undef EVK_C3_PIECES

  // slab [tap][split][co][ci]: lane holds co = .. + (lane & 15), ci = .. + 4 (lane >> 4) + 0..3
  const long mn = (long)p.Co * p.Ci;
  const int co = cot * 64 + wco * 16 + (lane & 15);
  const int ci = cit * 64 + wci * 16 + fq * 4;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float* s0 = p.slab + ((long)t * p.nsplit + split) * mn + (long)co * p.Ci + ci;
    *reinterpret_cast<float4*>(s0) = make_float4(acc[t][0][0], acc[t][0][1], acc[t][0][2], acc[t][0][3]);
    *reinterpret_cast<float4*>(s0 + 16L * p.Ci) = make_float4(acc[t][1][0], acc[t][1][1], acc[t][1][2], acc[t][1][3]);
  }
  if (p.stamps && tid == 0) { p.stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memtime(); p.stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime(); }
}

struct Plan { int R, tpxp, rps, nsplit, pairs_ci, npairs; };

bool make_plan(int N, int H, int W, int Ci, int Co, Plan& pl) {
  if (N <= 0 || H <= 0 || W <= 0 || Ci < 64 || (Ci % 64) || Co < 64 || (Co % 64) || W + 2 > 0xffff) return false;
  const int TR = N * H;
  if ((long)TR * (H + 1) >= (1L << 22) || (long)TR * W >= (1L << 31)) return false;
  // rows per tile: the largest divisor of H whose pixels (rounded up to 32) and halo fit one stage; tiles never cross an image
  int R = 0;
  for (int r = H; r >= 1; --r) {
    if (H % r) continue;
    const int tp = (r * W + 31) / 32 * 32;
    if (tp + (r + 2) * (W + 2) <= STAGE_ROWS) { R = r; break; }
  }
  if (R < 1 || R * W < 48) return false;                  // tiles of under 48 pixels: the tile path is the better kernel
  pl.R = R;
  pl.tpxp = (R * W + 31) / 32 * 32;
  pl.pairs_ci = Ci / 64;
  pl.npairs = (Ci / 64) * (Co / 64);
  static const int target = evk_tunable("EVK_C3W_BLOCKS", 256);
  int ns = (target + pl.npairs / 2) / pl.npairs;
  const int tiles = (int)cdiv(TR, R);
  if (ns > tiles) ns = tiles;
  if (ns < 1) ns = 1;
  pl.rps = (int)cdiv(cdiv(TR, ns), R) * R;
  pl.nsplit = (int)cdiv(TR, pl.rps);
  return true;
}

}  // namespace wgk

bool halo_enabled() {
  static const int on = evk_tunable("EVK_CONV3X3_HALO", 1);
  return on != 0;
}

template <class CF>
int launch_halo(const C3P& p, hipStream_t s) {
  EVK_DYN_LDS_ONCE(conv3x3_halo_kernel<CF>, CF::LDS_BYTES);
  hipLaunchKernelGGL(conv3x3_halo_kernel<CF>, dim3(p.tilesM * p.tilesN), dim3(NTH), CF::LDS_BYTES, s, p);
  return evk_check_launch("conv3x3_halo_kernel");
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int evk_conv3x3_halo_supported(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Co) {
  if (N <= 0 || H <= 0 || W <= 0 || C < 64 || (C % 64) || Co < Cfg64::TN || (Co % Cfg64::TN)) return 0;
  // the 64-channel configuration stores its eighth halo piece in a chunk's last step, the step that already reads the next chunk's
  // fragments: it is built for single-chunk inputs (C == 64, layer1), where no piece is ever loaded inside the loop
  if (!wide_tiles(Co) && C != 64) return 0;
  if ((long)N * H * W * C * 2 >= (1L << 32)) return 0;          // 32-bit byte offsets into x
  const int R = choose_rows(N, H, W, halo_max_for(Co));
  if (R < 1 || R * W < TP / 2) return 0;                         // tiles under half full: the tile GEMM path is the better kernel
  return 1;
}

int64_t evk_conv3x3_halo_part_bytes(int32_t N, int32_t H, int32_t W, int32_t Co) {
  const int R = choose_rows(N, H, W, halo_max_for(Co));
  if (R < 1) return 0;
  return cdiv((int64_t)N * H, R) * WM * 2 * Co * (int64_t)sizeof(float);
}

int evk_conv3x3_halo(const void* x, const void* w, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Co,
                     const void* resid, int64_t ldr, const void* gate, int64_t ldg, float* colstats, float* gatestats,
                     int64_t part_bytes, int32_t* nblk, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && w && y, "conv3x3_halo: null operand");
  EVK_REQUIRE(evk_conv3x3_halo_supported(N, H, W, C, Co), "conv3x3_halo: unsupported shape N=%d H=%d W=%d C=%d Co=%d (C %% 64, Co %% 64, tileable rows)", N, H, W, C, Co);
  EVK_REQUIRE(al16(x) && al16(w) && al16(y) && (!resid || (al16(resid) && ldr % 4 == 0 && ldr >= Co)) && (!gate || (al16(gate) && ldg % 4 == 0 && ldg >= Co)),
              "conv3x3_halo: operands must be 16-byte aligned, leading dimensions multiples of 4 and >= Co");
  EVK_REQUIRE(!(colstats && gatestats) && (!gatestats || gate), "conv3x3_halo: one statistics epilogue at a time; gate statistics need a gate");
  const bool wide = wide_tiles(Co);
  C3P p{};
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Co = Co;
  p.R = choose_rows(N, H, W, halo_max_for(Co));
  p.tilesM = (int)cdiv((int64_t)N * H, p.R);
  p.tilesN = Co / (wide ? Cfg128::TN : Cfg64::TN);
  p.resid = (const bf16_t*)resid; p.ldr = ldr; p.gate = (const bf16_t*)gate; p.ldg = ldg;
  p.colstats = colstats; p.gatestats = gatestats;
  static const int probe = evk_tunable("EVK_C3_PROBE", 0);
  p.kmul = probe ? 0 : 1;
  p.inv_w = 1.f / W; p.inv_w2 = 1.f / (W + 2); p.inv_h = 1.f / H; p.inv_h1 = 1.f / (H + 1);
  p.stamps = g_stamps;
  if (colstats || gatestats) {
    EVK_REQUIRE(nblk && part_bytes >= evk_conv3x3_halo_part_bytes(N, H, W, Co), "conv3x3_halo: statistics buffer too small");
    *nblk = p.tilesM * WM;
  }
  evk_prof_tag(N * H * W, Co, 9 * C, 1, EVK_A_CONV, EVK_B_PLAIN);
  ProfScope ps(EVK_FAM_GEMM, s, 2.0 * N * H * W * (double)Co * 9 * C);
  return wide ? launch_halo<Cfg128>(p, s) : launch_halo<Cfg64>(p, s);
}

/* inference form (eval-mode batch norm as per-channel scale / shift): y = relu?(conv3x3(x, w) * scale[co] + bias[co] (+ resid)) */
int evk_conv3x3_halo_affine(const void* x, const void* w, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Co, const float* scale,
                            const float* bias, const void* resid, int64_t ldr, int32_t relu, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && w && y && scale && bias, "conv3x3_halo_affine: null operand");
  EVK_REQUIRE(evk_conv3x3_halo_supported(N, H, W, C, Co), "conv3x3_halo_affine: unsupported shape N=%d H=%d W=%d C=%d Co=%d", N, H, W, C, Co);
  EVK_REQUIRE(al16(x) && al16(w) && al16(y) && al16(scale) && al16(bias) && (!resid || (al16(resid) && ldr % 4 == 0 && ldr >= Co)), "conv3x3_halo_affine: alignment / leading dimension");
  const bool wide = wide_tiles(Co);
  C3P p{};
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Co = Co;
  p.R = choose_rows(N, H, W, halo_max_for(Co));
  p.tilesM = (int)cdiv((int64_t)N * H, p.R);
  p.tilesN = Co / (wide ? Cfg128::TN : Cfg64::TN);
  p.resid = (const bf16_t*)resid; p.ldr = ldr;
  p.kmul = 1;
  p.inv_w = 1.f / W; p.inv_w2 = 1.f / (W + 2); p.inv_h = 1.f / H; p.inv_h1 = 1.f / (H + 1);
  p.scale = scale; p.bias = bias; p.relu = relu;
  evk_prof_tag(N * H * W, Co, 9 * C, 1, EVK_A_CONV, EVK_B_PLAIN);
  ProfScope ps(EVK_FAM_GEMM, s, 2.0 * N * H * W * (double)Co * 9 * C);
  return wide ? launch_halo<Cfg128>(p, s) : launch_halo<Cfg64>(p, s);
}

int evk_conv3x3_wgrad_halo_supported(int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co) {
  wgk::Plan pl;
  return wgk::make_plan(N, H, W, Ci, Co, pl) ? 1 : 0;
}

int64_t evk_conv3x3_wgrad_halo_ws_bytes(int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co) {
  wgk::Plan pl;
  if (!wgk::make_plan(N, H, W, Ci, Co, pl)) return 0;
  return 9LL * pl.nsplit * Co * Ci * (int64_t)sizeof(float);
}

int evk_conv3x3_wgrad_halo(const void* dy, const void* x, float* dw, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                           void* ws, int64_t ws_bytes, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dy && x && dw && ws, "conv3x3_wgrad_halo: null operand / workspace");
  wgk::Plan pl;
  EVK_REQUIRE(wgk::make_plan(N, H, W, Ci, Co, pl), "conv3x3_wgrad_halo: unsupported shape N=%d H=%d W=%d Ci=%d Co=%d (channels %% 64, rows that tile)", N, H, W, Ci, Co);
  EVK_REQUIRE(ws_bytes >= evk_conv3x3_wgrad_halo_ws_bytes(N, H, W, Ci, Co), "conv3x3_wgrad_halo: workspace too small");
  EVK_REQUIRE(al16(dy) && al16(x) && al16(dw) && al16(ws), "conv3x3_wgrad_halo: operands must be 16-byte aligned");
  wgk::W3P p{};
  p.dy = (const bf16_t*)dy; p.x = (const bf16_t*)x; p.slab = (float*)ws;
  p.N = N; p.H = H; p.W = W; p.Co = Co; p.Ci = Ci;
  p.R = pl.R; p.tpxp = pl.tpxp; p.rps = pl.rps; p.nsplit = pl.nsplit; p.pairs_ci = pl.pairs_ci; p.npairs = pl.npairs;
  p.inv_w = 1.f / W; p.inv_w2 = 1.f / (W + 2); p.inv_h = 1.f / H; p.inv_h1 = 1.f / (H + 1);
  static EvkDeviceOnce zeros_once;
  p.zeros = zeros_once.get([]() -> void* {
    void* z = nullptr;
    if (hipGetSymbolAddress(&z, HIP_SYMBOL(wgk::g_zero16)) != hipSuccess) z = nullptr;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgk::conv3x3_wgrad_halo_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, wgk::LDS_BYTES);
    return z;
  });
  EVK_REQUIRE(p.zeros, "conv3x3_wgrad_halo: no address for the zero block");
  p.stamps = g_stamps;
  evk_prof_tag(Co, Ci, N * H * W, 9, EVK_A_KSTR, EVK_B_WGATHER);
  ProfScope ps(EVK_FAM_GEMM, s, 2.0 * N * H * W * (double)Co * 9 * Ci);
  hipLaunchKernelGGL(wgk::conv3x3_wgrad_halo_kernel, dim3(pl.npairs * pl.nsplit), dim3(NTH), wgk::LDS_BYTES, s, p);
  if (int e = evk_check_launch("conv3x3_wgrad_halo_kernel")) return e;
  // dw[co][tap][ci] += sum over the K-slices: gemm.hip's split-K reduction with the tap as batch index
  return evk_splitk_reduce_launch(p.slab, dw, (long)Co * Ci, Co, Ci, pl.nsplit, 9, 9L * Ci, 0, Ci, 9, s);
}

int evk_conv3x3_wgrad_halo_routes(const evk_conv_geom* g) {
  static const int on = evk_tunable("EVK_CONV3X3_WGRAD_HALO", 1);
  if (!on || !g) return 0;
  if (g->KH != 3 || g->KW != 3 || g->stride_h != 1 || g->stride_w != 1 || g->pad_h != 1 || g->pad_w != 1) return 0;
  if (g->Hi != g->Ho || g->Wi != g->Wo) return 0;
  return evk_conv3x3_wgrad_halo_supported(g->N, g->Hi, g->Wi, g->Ci, g->Co);
}

// diagnostic: while buf is non-null every evk_conv3x3_halo launch writes, per workgroup, 8 x uint64 {s_memtime at entry, before the
// K loop, after it, at exit; s_memrealtime (100 MHz) at entry and exit; 2 unused} into buf (tools/conv3x3_probe.py)
int evk_conv3x3_halo_debug_stamps(void* buf) { g_stamps = reinterpret_cast<unsigned long long*>(buf); return EVK_OK; }

// routing used by conv.hip: 1 when the halo kernel takes a 3x3 / stride 1 / pad 1 convolution of this geometry with a statistics
// buffer of part_bytes (0 = no statistics requested)
int evk_conv3x3_halo_routes(const evk_conv_geom* g, int32_t C, int32_t Co, int64_t part_bytes, int32_t want_stats) {
  if (!halo_enabled() || !g) return 0;
  if (g->KH != 3 || g->KW != 3 || g->stride_h != 1 || g->stride_w != 1 || g->pad_h != 1 || g->pad_w != 1) return 0;
  if (g->Hi != g->Ho || g->Wi != g->Wo) return 0;
  if (!evk_conv3x3_halo_supported(g->N, g->Hi, g->Wi, C, Co)) return 0;
  if (want_stats && part_bytes < evk_conv3x3_halo_part_bytes(g->N, g->Hi, g->Wi, Co)) return 0;
  return 1;
}

}  // extern "C"
