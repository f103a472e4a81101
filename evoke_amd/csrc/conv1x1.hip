// conv1x1.hip -- weight-stationary kernel for the SHORT-K pointwise convolutions of the ResNet trunk (torchvision Bottleneck
// conv3: planes -> 4*planes, and the data gradient of conv1: the same shape mirrored; modules/visual_extractor.py:27-43).
//
// These products (K = 64 ... 512 input channels, N = 4K output channels, M = 10^4 ... 10^6 pixels) are memory bound -- 205 flop
// per byte of compulsory traffic at K = 256 -- yet the tile GEMM of gemm.hip runs them at 1.6 TB/s of algorithmic traffic: a
// 128 x 128 tile with K = 256 pulls 128 KB of operands through the CU's vector L1 for 32 KB of output, and one CU's L1 fills at
// ~27 GB/s.  Here the WEIGHTS never move: a workgroup owns a slice of 4 x NW output channels, each of its four
// waves keeps its NW x K weight block in registers as MFMA A fragments for the whole launch (128 VGPRs at K = 256, NW = 64), and
// the workgroup walks its share of the pixel tiles: the activation tile (128 pixels x K, the only operand that streams) is staged
// in LDS once and read by all four waves, double buffered against the next tile's global loads.  Fill bytes per output byte drop
// 4x at N = 1024; the output leaves through a wave-private LDS transpose as whole 128-byte rows.
//   D[channel][pixel] = sum_k W[channel][k] * X[pixel][k]:  A operand = W rows (K-contiguous), B operand = pixel rows (K-contiguous,
//   NHWC), so a lane ends up with 4 consecutive channels of one pixel.
// Epilogues: forward = per-channel sum / sum of squares partials for the batch norm that follows (one partial row per pixel tile,
// bn.hip's second stage sums them); data gradient = + skip gradient, ReLU gate, gate statistics (sum g, sum g*z).
#include <stdlib.h>
#include <algorithm>
#include "common.h"

namespace {

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

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ s16x4 lds_tr_read(const bf16_t* generic_lds_ptr) {
  lds_s16x4* p = (lds_s16x4*)(__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)generic_lds_ptr;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
}

// rows of the [K][NS] weight slice the data gradient stages through LDS at a time: the largest power of two (>= 32) that fits the tile
// double buffer, so the staging never sets the LDS footprint
constexpr int ws_kc(int K, int MT, int LW) {
  int kc = 32;
  while (kc * 2 <= K && kc * 2 <= 256 && kc * 2 * LW <= 2 * MT * (K + 8)) kc *= 2;
  return kc;
}

struct WsP {
  const bf16_t* x;      // [M][K] activations (fwd: conv input; dgrad: dy)
  const bf16_t* w;      // fwd: W[N = Co][K = Ci] (K-contiguous); dgrad: W[K = Co][N = Ci] as stored (transposed on the way into registers)
  bf16_t* y;            // [M][N]
  const bf16_t* skip;   // dgrad: optional [M][N] added before the gate
  const bf16_t* gate;   // dgrad: optional [M][N], output zeroed where gate <= 0
  float* part;          // optional [ntiles][2][N]: fwd (sum, sumsq) of y; dgrad (sum g, sum g*gate); [ntiles][3][N] with sx
  const bf16_t* sx;     // dgrad: optional [M][N] second statistic operand: third partial row = sum over pixels of dx * (sx - smean)
  const float* smean;   // [N] per-channel offset of sx (the batch mean of the batch norm whose input sx is), or null
  long M; int N, ntiles, slices, groups;
  const float* scale; const float* bias; int relu;   // forward, inference (eval-mode batch norm in the epilogue): y = relu?((x . w) * scale[n] + bias[n] (+ skip)); no statistics
};

template <int K, int NW, int MT, bool DGRAD, int NWV, int AFF = 0>       // AFF: the inference forward (1: scale / bias / ReLU epilogue, 2: + identity)
__global__ __launch_bounds__(64 * NWV, 1) void conv1x1_ws_kernel(const WsP p) {
  constexpr int KS = K / 32, NT = NW / 16, LDA = K + 8;
  constexpr int NTHR = 64 * NWV;
  constexpr int CH = MT * K / 8 / NTHR;                // 16-byte chunks of an activation tile per thread
  constexpr int OST = NW + 8;                          // row stride of the wave-private output transpose (16-bit elements)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* abuf = reinterpret_cast<bf16_t*>(smem);                          // [2][MT][LDA]
  constexpr int NS = NWV * NW, LW = NS + 8;
  constexpr int KCW = ws_kc(K, MT, NWV * NW + 8);
  constexpr int ABUF = (2 * MT * LDA > (DGRAD ? KCW * LW : 0)) ? 2 * MT * LDA : KCW * LW;  // elements: tile double buffer / weight staging
  bf16_t* obuf = abuf + ABUF;                                              // [NWV waves][16][OST] output transpose
  bf16_t* sbuf = obuf;                                                     // dgrad: the skip rows arrive in the buffer the output leaves through
  bf16_t* gbuf = obuf + NWV * 16 * OST;                                    // (each lane reads its 8 bytes before it overwrites them); gate rows
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  // XCD-aware mapping: the `slices` workgroups that read the same pixel tiles sit on one XCD (blockIdx % 8 labels the XCD)
  int slice, grp;
  {
    const int bid = blockIdx.x;
    if (p.groups % 8 == 0) { const int xcd = bid & 7, idx = bid >> 3; grp = xcd + 8 * (idx / p.slices); slice = idx % p.slices; }
    else { grp = bid / p.slices; slice = bid % p.slices; }
  }
  const int n0 = slice * NWV * NW + wave * NW;         // first output channel of this wave
  // ---- the wave's weights: NT x KS A fragments, resident for the whole launch
  bf16x8 wf[NT][KS];
  if (!DGRAD) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[nt][ks] = *reinterpret_cast<const bf16x8*>(p.w + (long)(n0 + nt * 16 + li) * K + ks * 32 + g * 8);
  } else {
    // data gradient: the weights lie [K = Co][N = Ci] (n contiguous).  The workgroup's [K][NS] slice goes through LDS as it lies and
    // the A fragments (8 consecutive k of one channel) come back through the transposing LDS read (two 4 x 16 blocks per fragment)
    constexpr int KC = KCW;                              // rows of the weight slice staged at a time
    const int q4 = li >> 2, p4 = li & 3;
#pragma unroll
    for (int kc = 0; kc < K; kc += KC) {
      __syncthreads();
      for (int c = tid; c < KC * (NS / 8); c += NTHR) {
        const int k = c / (NS / 8), c8 = c % (NS / 8);
        *reinterpret_cast<uint4*>(abuf + k * LW + c8 * 8) = *reinterpret_cast<const uint4*>(p.w + (long)(kc + k) * p.N + slice * NS + c8 * 8);
      }
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int k2 = 0; k2 < KC / 32; ++k2) {
          const bf16_t* tb = abuf + (k2 * 32 + g * 8 + q4) * LW + wave * NW + nt * 16 + p4 * 4;
          const s16x4 lo = lds_tr_read(tb), hi = lds_tr_read(tb + 4 * LW);
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          wf[nt][kc / 32 + k2] = __builtin_bit_cast(bf16x8, r);
        }
    }
    __syncthreads();
  }

  uint4 pf[CH];
  auto fetch = [&](int t) {                            // global -> registers: tile t (rows beyond M read as zero)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = tid + NTHR * c, r = ch / (K / 8), kc = ch % (K / 8);
      const long row = (long)t * MT + r;
      if (AFF == 2) {     // every load is issued (clamped row: the pixels beyond M are never stored) -- see the identity rows below
        const long rc = row < p.M ? row : p.M - 1;
        pf[c] = *reinterpret_cast<const uint4*>(p.x + rc * K + kc * 8);
      } else {
        pf[c] = row < p.M ? *reinterpret_cast<const uint4*>(p.x + row * K + kc * 8) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto stash = [&](int buf) {                          // registers -> LDS
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = tid + NTHR * c, r = ch / (K / 8), kc = ch % (K / 8);
      *reinterpret_cast<uint4*>(abuf + (buf * MT + r) * LDA + kc * 8) = pf[c];
    }
  };
  constexpr int CPRW = NW / 8;                         // 16-byte chunks per pixel row of the wave's NW channels
  const bool xstat = DGRAD && p.sx && p.part;
  const int prn = xstat ? 3 : 2;                       // partial rows per tile
  float mu[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) mu[k] = (xstat && p.smean) ? p.smean[n0 + (lane % CPRW) * 8 + k] : 0.f;
  int t = grp;
  if (t < p.ntiles) { fetch(t); stash(0); }
  __syncthreads();
  int cur = 0;
  bf16_t* ow = obuf + wave * 16 * OST;
  // statistics accumulate over ALL the pixel tiles this workgroup walks (fixed order: repeatable) and leave as ONE partial row per
  // workgroup group: the second reduction stage (bn.hip) reads <= 256 rows instead of one per 128 pixels (4608 at layer1)
  float s0[NT][4], s1[NT][4], tx[8];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) { s0[nt][j] = 0.f; s1[nt][j] = 0.f; }
#pragma unroll
  for (int k = 0; k < 8; ++k) tx[k] = 0.f;
  constexpr int CPR = NW / 8;                           // 16-byte chunks per pixel row of the wave's NW channels
  constexpr int NCH = (16 * CPR + 63) / 64;             // chunks per lane for a 16-pixel group
  // inference forward with an identity (AFF == 2): its rows (the lane's 16 bytes of every 16-pixel group, in the layout the output leaves in)
  // are requested a whole pixel tile ahead, into one register set per pixel group of the (fully unrolled) tile loop.  Asked for at the point
  // of use, a dependent ~1 us load per group made the kernel 2-5 x slower than the plain forward; a rotating register queue is no better (a
  // register MOVE of a load's destination waits for the load); and a predicated or branch-guarded load anywhere in the loop makes the
  // compiler's wait-count bookkeeping fall back to s_waitcnt vmcnt(0) once per tile (the latency exposed again), so in this variant every
  // load of the loop is issued unconditionally with a clamped address and the weights' loads are drained before the loop.
  constexpr int RD = AFF == 2 ? MT / 16 : 1;
  constexpr int UNR = AFF == 2 ? MT / 16 : 1;
  uint4 rq[RD][NCH];
  auto res_req = [&](uint4 (&dst)[NCH], int tile, int mt) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i, pr = (c / CPR) & 15, cc = c % CPR;
      long prow = (long)tile * MT + mt * 16 + pr;
      prow = prow < p.M ? prow : p.M - 1;
      dst[i] = *reinterpret_cast<const uint4*>(p.skip + prow * p.N + n0 + cc * 8);
    }
  };
  if (AFF == 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the resident weights and the first tile: nothing older than the queue below
    const int t0 = t < p.ntiles ? t : p.ntiles - 1;
#pragma unroll
    for (int d = 0; d < RD; ++d) res_req(rq[d], t0, d);
  }
  // the eval-mode batch norm of the wave's channels (AFF): scale / shift of the 4 channels the lane holds per 16-channel block
  float4 bsc[NT], bsh[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bsc[nt] = AFF ? *reinterpret_cast<const float4*>(p.scale + n0 + nt * 16 + g * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    bsh[nt] = AFF ? *reinterpret_cast<const float4*>(p.bias + n0 + nt * 16 + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (; t < p.ntiles; t += p.groups) {
    const int tn = t + p.groups;
    const int tr = tn < p.ntiles ? tn : t;              // (AFF == 2) the tile whose rows are requested as this one's are consumed
    if (AFF == 2) fetch(tr);
    else if (tn < p.ntiles) fetch(tn);                 // in flight while this tile multiplies
    const bf16_t* at = abuf + cur * MT * LDA;
#pragma unroll UNR
    for (int mt = 0; mt < MT / 16; ++mt) {
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      // data gradient: the skip / gate rows of this pixel group, whole rows, in flight while the group multiplies
      uint4 skv[NCH], gtv[NCH], sxv[NCH];
      if (DGRAD) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c = lane + 64 * i, pr = c / CPR, cc = c % CPR;
          const long prow = (long)t * MT + mt * 16 + pr;
          const bool ok = c < 16 * CPR && prow < p.M;
          skv[i] = (ok && p.skip) ? *reinterpret_cast<const uint4*>(p.skip + prow * p.N + n0 + cc * 8) : make_uint4(0, 0, 0, 0);
          gtv[i] = (ok && p.gate) ? *reinterpret_cast<const uint4*>(p.gate + prow * p.N + n0 + cc * 8) : make_uint4(0, 0, 0, 0);
          sxv[i] = (ok && xstat) ? *reinterpret_cast<const uint4*>(p.sx + prow * p.N + n0 + cc * 8) : make_uint4(0, 0, 0, 0);
        }
      }
      const bf16_t* brow = at + (mt * 16 + li) * LDA + g * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(brow + ks * 32);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = EVK_MFMA_16x16x32(wf[nt][ks], b, acc[nt], 0, 0, 0);
      }
      bf16_t* sw = sbuf + wave * 16 * OST;
      bf16_t* gw = gbuf + wave * 16 * OST;
      if (DGRAD) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c = lane + 64 * i, pr = c / CPR, cc = c % CPR;
          if (c < 16 * CPR) {
            *reinterpret_cast<uint4*>(sw + pr * OST + cc * 8) = skv[i];
            *reinterpret_cast<uint4*>(gw + pr * OST + cc * 8) = gtv[i];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (AFF == 2) {          // the identity rows of this pixel group (requested a tile ago) through the wave's transpose buffer, like the skip rows above
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c = lane + 64 * i, pr = c / CPR, cc = c % CPR;
          if (c < 16 * CPR) *reinterpret_cast<uint4*>(sw + pr * OST + cc * 8) = rq[AFF == 2 ? mt : 0][i];
        }
        res_req(rq[AFF == 2 ? mt : 0], tr, mt);          // this group's rows of the NEXT tile, into the registers just consumed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      // lane holds D[channel n0 + nt*16 + 4g + j][pixel t*MT + mt*16 + li]
      const long pix = (long)t * MT + mt * 16 + li;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float v[4] = {acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]};
        if (DGRAD) {
          if (pix < p.M) {
            if (p.skip) { const uint2 sk = *reinterpret_cast<const uint2*>(sw + li * OST + nt * 16 + g * 4); v[0] += lo_bf(sk.x); v[1] += hi_bf(sk.x); v[2] += lo_bf(sk.y); v[3] += hi_bf(sk.y); }
            if (p.gate) {
              const uint2 gt = *reinterpret_cast<const uint2*>(gw + li * OST + nt * 16 + g * 4);
              const float gv[4] = {lo_bf(gt.x), hi_bf(gt.x), lo_bf(gt.y), hi_bf(gt.y)};
#pragma unroll
              for (int j = 0; j < 4; ++j) { if (!(gv[j] > 0.f)) v[j] = 0.f; s0[nt][j] += v[j]; s1[nt][j] += v[j] * gv[j]; }
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 0.f;
          }
        } else if (AFF) {
          // bit for bit the arithmetic of the unfused eval forward (this kernel's rounded output -> bn_apply_kernel: fma(x, scale, shift) +
          // identity, ReLU, one more rounding): a batch whose size sends a layer down another route then decodes to the same tokens
          const float4 bb = bsh[nt], sc = bsc[nt];
          const uint32_t r01 = pack2bf(v[0], v[1]), r23 = pack2bf(v[2], v[3]);
          v[0] = __builtin_fmaf(lo_bf(r01), sc.x, bb.x); v[1] = __builtin_fmaf(hi_bf(r01), sc.y, bb.y);
          v[2] = __builtin_fmaf(lo_bf(r23), sc.z, bb.z); v[3] = __builtin_fmaf(hi_bf(r23), sc.w, bb.w);
          if (AFF == 2) {
            const uint2 sk = *reinterpret_cast<const uint2*>(sw + li * OST + nt * 16 + g * 4);
            v[0] += lo_bf(sk.x); v[1] += hi_bf(sk.x); v[2] += lo_bf(sk.y); v[3] += hi_bf(sk.y);
          }
          if (p.relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) { s0[nt][j] += v[j]; s1[nt][j] += v[j] * v[j]; }
        }
        *reinterpret_cast<uint2*>(ow + li * OST + nt * 16 + g * 4) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // 16 pixels x NW channels leave as whole rows: NW * 2 bytes per pixel, 16 bytes per lane
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i, pr = c / CPR, cc = c % CPR;
        const long prow = (long)t * MT + mt * 16 + pr;
        if (c < 16 * CPR && prow < p.M) {
          const uint4 o = *reinterpret_cast<const uint4*>(ow + pr * OST + cc * 8);
          *reinterpret_cast<uint4*>(p.y + prow * p.N + n0 + cc * 8) = o;
          if (DGRAD && xstat) {         // the lane holds 8 channels of one pixel of the (rounded) result and of sx
            const uint32_t ov[4] = {o.x, o.y, o.z, o.w}, xv[4] = {sxv[i].x, sxv[i].y, sxv[i].z, sxv[i].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              tx[2 * k] += lo_bf(ov[k]) * (lo_bf(xv[k]) - mu[2 * k]);
              tx[2 * k + 1] += hi_bf(ov[k]) * (hi_bf(xv[k]) - mu[2 * k + 1]);
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (AFF == 2 || tn < p.ntiles) stash(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (p.part) {
    float* prow = p.part + (long)grp * prn * p.N;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = row16_sum(s0[nt][j]); b[j] = row16_sum(s1[nt][j]); }
      if (li == 0) {
        const int c = n0 + nt * 16 + g * 4;
        *reinterpret_cast<float4*>(prow + c) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(prow + p.N + c) = make_float4(b[0], b[1], b[2], b[3]);
      }
    }
    if (DGRAD && xstat) {             // lanes with equal lane % CPRW hold the same 8 channels (one pixel row each): sum over the rows
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int off = CPRW; off < 64; off <<= 1) tx[k] += __shfl_xor(tx[k], off, 64);
      }
      if (lane < CPRW) {
        float* d = prow + 2 * p.N + n0 + lane * 8;
        *reinterpret_cast<float4*>(d) = make_float4(tx[0], tx[1], tx[2], tx[3]);
        *reinterpret_cast<float4*>(d + 4) = make_float4(tx[4], tx[5], tx[6], tx[7]);
      }
    }
  }
}

template <int K, int NW, int MT, bool DGRAD, int NWV, int AFF = 0>
int launch_ws(WsP p, hipStream_t s) {
  const size_t abuf = std::max((size_t)2 * MT * (K + 8), DGRAD ? (size_t)ws_kc(K, MT, NWV * NW + 8) * (NWV * NW + 8) : (size_t)0);
  const size_t lds = abuf * 2 + (size_t)(DGRAD ? 2 : 1) * NWV * 16 * (NW + 8) * 2;
  if (lds > 65536) EVK_DYN_LDS_ONCE((&conv1x1_ws_kernel<K, NW, MT, DGRAD, NWV, AFF>), lds);
  p.ntiles = (int)cdiv(p.M, MT);
  p.slices = p.N / (NWV * NW);
  int groups = 256 / p.slices;
  if (groups < 1) groups = 1;
  if (groups > p.ntiles) groups = p.ntiles;
  if (groups >= 8) groups &= ~7;
  p.groups = groups;
  ProfScope ps(EVK_FAM_GEMM, s, 2.0 * (double)p.M * p.N * K);
  evk_prof_tag((int)p.M, p.N, K, 1, 0, DGRAD ? 1 : 0);
  hipLaunchKernelGGL((conv1x1_ws_kernel<K, NW, MT, DGRAD, NWV, AFF>), dim3(p.slices * groups), dim3(64 * NWV), lds, s, p);
  return evk_check_launch("conv1x1_ws");
}

template <bool DGRAD, int AFF = 0>
int dispatch(const WsP& p, int K, hipStream_t s) {
  static const int v8 = evk_tunable("EVK_WS_WAVES8", 1);
  if (v8) switch (K) {          // 8 waves x 32 channels (2 waves per SIMD hide the LDS / store latency of the per-16-pixel epilogue)
    case 64: return launch_ws<64, 32, 128, DGRAD, 8, AFF>(p, s);
    case 128: return launch_ws<128, 32, 128, DGRAD, 8, AFF>(p, s);
    case 256: return launch_ws<256, 32, 128, DGRAD, 8, AFF>(p, s);
    case 512: return launch_ws<512, 16, 64, DGRAD, 8, AFF>(p, s);
    case 1024: return launch_ws<1024, 16, 32, DGRAD, 8, AFF>(p, s);
    default: break;
  }
  switch (K) {
    case 64: return launch_ws<64, 64, 128, DGRAD, 4, AFF>(p, s);
    case 128: return launch_ws<128, 64, 128, DGRAD, 4, AFF>(p, s);
    case 256: return launch_ws<256, 64, 128, DGRAD, 4, AFF>(p, s);
    case 512: return launch_ws<512, 32, 64, DGRAD, 4, AFF>(p, s);
    case 1024: return launch_ws<1024, 16, 32, DGRAD, 8, AFF>(p, s);
    default: evk_set_error("conv1x1_ws: unsupported K = %d", K); return EVK_EINVAL;
  }
}

}  // namespace

extern "C" {

/* 1 when the weight-stationary kernel takes the product: K in {64, 128, 256, 512, 1024}, N a multiple of the workgroup's channel slice */
int evk_conv1x1_ws_supported(int64_t M, int32_t K, int32_t N) {
  if (M <= 0 || N <= 0) return 0;
  if (K == 64 || K == 128 || K == 256) return N % 256 == 0;
  if (K == 512 || K == 1024) return N % 128 == 0;
  return 0;
}

static int ws_tile_rows(int K) { return K == 1024 ? 32 : (K == 512 ? 64 : 128); }
// workgroup groups of a launch = partial rows it writes (launch_ws: 256 / slices, at most one per pixel tile, whole XCD octets)
static int ws_groups(int64_t M, int K, int N) {
  const int slices = N / (K >= 512 ? 128 : 256);
  int groups = 256 / (slices > 0 ? slices : 1);
  const int ntiles = (int)cdiv(M, ws_tile_rows(K));
  if (groups < 1) groups = 1;
  if (groups > ntiles) groups = ntiles;
  if (groups >= 8) groups &= ~7;
  return groups;
}
int64_t evk_conv1x1_ws_part_bytes(int64_t M, int32_t K, int32_t N) { return cdiv(M, ws_tile_rows(K)) * 2 * (int64_t)N * 4; }

int evk_conv1x1_ws_fwd(const void* x, const void* w, void* y, int64_t M, int32_t K, int32_t N, float* part, int64_t part_bytes, int32_t* nblk,
                       evk_stream_t stream) {
  EVK_REQUIRE(x && w && y && evk_conv1x1_ws_supported(M, K, N), "conv1x1_ws_fwd: unsupported problem M=%ld K=%d N=%d", (long)M, K, N);
  EVK_REQUIRE(!part || (nblk && part_bytes >= evk_conv1x1_ws_part_bytes(M, K, N)), "conv1x1_ws_fwd: statistics buffer too small");
  if (part) *nblk = ws_groups(M, K, N);
  WsP p{(const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, nullptr, nullptr, part, nullptr, nullptr, M, N, 0, 0, 0, nullptr, nullptr, 0};
  return dispatch<false>(p, K, reinterpret_cast<hipStream_t>(stream));
}

/* inference form of the forward (eval-mode batch norm as per-channel scale / shift): y = relu?((x . w^T) * scale[n] + bias[n] (+ resid[m][n])) */
int evk_conv1x1_ws_fwd_affine(const void* x, const void* w, void* y, int64_t M, int32_t K, int32_t N, const float* scale, const float* bias,
                              const void* resid, int32_t relu, evk_stream_t stream) {
  EVK_REQUIRE(x && w && y && scale && bias && evk_conv1x1_ws_supported(M, K, N), "conv1x1_ws_fwd_affine: unsupported problem M=%ld K=%d N=%d", (long)M, K, N);
  EVK_REQUIRE(((reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(scale)) & 15) == 0 && (!resid || (reinterpret_cast<uintptr_t>(resid) & 15) == 0),
              "conv1x1_ws_fwd_affine: 16-byte aligned scale / bias / residual");
  WsP p{(const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, (const bf16_t*)resid, nullptr, nullptr, nullptr, nullptr, M, N, 0, 0, 0, scale, bias, relu};
  return resid ? dispatch<false, 2>(p, K, reinterpret_cast<hipStream_t>(stream)) : dispatch<false, 1>(p, K, reinterpret_cast<hipStream_t>(stream));
}

/* dx[M][N] = gate(dy[M][K] . W[K][N] + skip): W as the forward stores it, [K = Co][N = Ci] */
int evk_conv1x1_ws_dgrad(const void* dy, const void* wt, const void* skip, const void* gate, void* dx, int64_t M, int32_t K, int32_t N,
                         float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream) {
  return evk_conv1x1_ws_dgrad_xstat(dy, wt, skip, gate, dx, M, K, N, nullptr, nullptr, part, part_bytes, nblk, stream);
}

/* same, with a third partial row per pixel tile when stat_x is given: sum over pixels of dx * (stat_x - stat_mean), the second column sum
   of the batch-norm backward whose INPUT is stat_x and whose output gradient is dx (part then holds [nblk][3][N] floats) */
int evk_conv1x1_ws_dgrad_xstat(const void* dy, const void* wt, const void* skip, const void* gate, void* dx, int64_t M, int32_t K, int32_t N,
                               const void* stat_x, const float* stat_mean, float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream) {
  EVK_REQUIRE(dy && wt && dx && evk_conv1x1_ws_supported(M, K, N), "conv1x1_ws_dgrad: unsupported problem M=%ld K=%d N=%d", (long)M, K, N);
  const int64_t need = evk_conv1x1_ws_part_bytes(M, K, N) / 2 * (stat_x ? 3 : 2);
  EVK_REQUIRE(!part || (gate && nblk && part_bytes >= need), "conv1x1_ws_dgrad: gate statistics need a gate and a large enough buffer");
  EVK_REQUIRE(!stat_x || part, "conv1x1_ws_dgrad: stat_x needs the partials buffer");
  if (part) *nblk = ws_groups(M, K, N);
  WsP p{(const bf16_t*)dy, (const bf16_t*)wt, (bf16_t*)dx, (const bf16_t*)skip, (const bf16_t*)gate, part, (const bf16_t*)stat_x, stat_mean, M, N, 0, 0, 0, nullptr, nullptr, 0};
  return dispatch<true>(p, K, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
