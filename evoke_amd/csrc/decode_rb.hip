// decode_rb.hip -- "row-block" kernels of the per-token decode step (modules/encoder_decoder.py:118-131 DecoderLayer.forward as driven by
// CaptionModel.beam_search through EncoderDecoder.core, :396-404): everything between two all-to-all points of a decoder sub-layer in ONE
// launch, for 16 hypotheses per workgroup --
//     v = x + g . W2^T + b2          g = a (attention output)             -- the output projection of self / cross attention + residual
//                                    g = relu(a . W1^T + b1)              -- the whole position-wise feed-forward (:206-214) + residual
//     y = round16(v)                 the residual stream for the next sub-layer
//     n = (gamma + dgamma[r]) * (v - mean) / (std_unbiased + eps) + (beta + dbeta[r])      the NEXT sub-layer's (conditional) layer norm
//                                    (:93-103 LayerNorm, :144-179 ConditionalLayerNorm: memory-conditioned gamma / beta deltas per hypothesis)
// The launch sequence it replaces is GEMM (+bias, +residual) -> LayerNorm (2 launches, or 3 with the feed-forward), each a few microseconds of
// work behind a launch boundary on a step that is bound by its kernel COUNT.  A full output row must sit in one
// workgroup for the norm, so a workgroup owns 16 rows x all 512 columns and STREAMS the 512 x 512 weights (512 KB) straight from L2 into
// MFMA operand registers: the host packs them fragment-major ([n tile][k step][lane][8]), one wave-instruction = 1 KB contiguous, 16
// such loads per lane in flight while the previous 16 are multiplied (128 KB per CU).  R / 16 workgroups (16 for 64 studies x beam 4): the
// other 240 CUs stay free for the second search in flight and the next batch's encoders.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int NTH = 512, D = 512, TMR = 16;
constexpr int HP = D * 2 + 16;                 // LDS pitch of the hidden rows of the feed-forward variant (bytes)

struct RbP {
  const bf16_t* a;            // [R][512] GEMM input rows
  const uint4* w1p; const float* b1;      // packed W1 + bias, or null (no first stage)
  const uint4* w2p; const float* b2;      // packed W2 + bias
  const bf16_t* x; bf16_t* y;             // residual in / out [R][512] (may alias)
  const float* gamma; const float* beta;  // [512] f32, or null: no norm
  const bf16_t* dgam; const bf16_t* dbet; long ld_delta;       // [R][ld_delta] 16-bit deltas or null
  bf16_t* n;                              // [R][512] normed rows out
  float eps; int R;
  unsigned* sync_ctr; float* sync_part;   // split row blocks: arrival counters [2][row blocks], partial row sums [row blocks][4][2][16]
  unsigned* sync_hid;                     // ... and the hidden rows of the split feed-forward [row blocks][16][256 dwords]
  int rbs;
};

// one arrival on a monotone counter + wait for the generation it belongs to (see decode_rowblock_kernel): thread 0 only
template <int NS>
__device__ __forceinline__ void cluster_arrive_wait(unsigned* ctr) {
  const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned target = (old / NS + 1u) * NS;      // the multiple of NS that completes this launch's generation
  while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// acc[j] += W[64 w + 16 j + .., :] . in[m, :] over K = 512: B fragments streamed from the packed weights, A fragments through `afrag(ks)`
template <class AF>
__device__ __forceinline__ void gemm16(const uint4* __restrict__ wp, int wave, int lane, AF afrag, f32x4 (&acc)[4]) {
  // packed index: ((nt * 16 + ks) * 64 + lane), nt = 4 wave + j
  const uint4* const base = wp + ((long)(4 * wave) * 16) * 64 + lane;
  uint4 b0[16], b1[16];
#define EVK_RB_LOAD(dst, q)                                                                         \
  _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) dst[t * 4 + j] = base[((long)j * 16 + (q) * 4 + t) * 64];
#define EVK_RB_MAC(src, q)                                                                          \
  _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                   \
    const bf16x8 af = afrag((q) * 4 + t);                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
      acc[j] = EVK_MFMA_16x16x32(__builtin_bit_cast(bf16x8, src[t * 4 + j]), af, acc[j], 0, 0, 0);  \
  }
  EVK_RB_LOAD(b0, 0)
  EVK_RB_LOAD(b1, 1)
  EVK_RB_MAC(b0, 0)
  EVK_RB_LOAD(b0, 2)
  EVK_RB_MAC(b1, 1)
  EVK_RB_LOAD(b1, 3)
  EVK_RB_MAC(b0, 2)
  EVK_RB_MAC(b1, 3)
#undef EVK_RB_LOAD
#undef EVK_RB_MAC
}

// NS = 4: the 512 output columns of a row block are split over FOUR workgroups (one 16-column MFMA tile per wave, 128 KB of weights per
// workgroup instead of 512 KB: the weights of a decode step -- 60 MB with the conditional-norm and relational-memory matrices -- cycle through
// the 4 MB L2s and arrive from the Infinity Cache at ~34 GB/s per CU, measured 15 us for 512 KB).  The norm needs whole rows, so the four
// exchange their per-row (sum, sum of squares) through device memory: agent-scope atomic stores (they leave through sc1, past the
// non-coherent L2 of another XCD), ONE returning agent-scope add per workgroup on the row block's arrival counter -- the value it returns
// tells the workgroup which multiple of four completes ITS generation, so the counter is never reset and a replayed launch sequence needs no
// per-launch argument --, a poll, agent-scope loads of the three other partial rows.  All 4 x R / 16 workgroups of a launch are resident
// together (64 for 256 hypotheses); a workgroup that waits holds one CU, never a lock.
template <bool FF, int NS>
__global__ __launch_bounds__(NTH, 2) void decode_rowblock_kernel(const RbP p) {
  constexpr int NT = 4 / NS;                             // MFMA column tiles per wave
  __shared__ __attribute__((aligned(16))) unsigned char hs[FF ? TMR * HP : 16];          // hidden rows of the feed-forward variant
  __shared__ float red[2][8][TMR];
  __shared__ float tot[2][TMR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kg = lane >> 4;
  const int rbk = blockIdx.x / NS, sp = blockIdx.x - rbk * NS;
  const int r0 = rbk * TMR;
  const int row = min(r0 + m, p.R - 1);                 // rows beyond R: computed on a copy of the last row, never stored
  const bool live = r0 + m < p.R;
  const int ncol = (NS == 1 ? 64 * wave : 128 * sp + 16 * wave) + 4 * kg;      // + 16 j: the lane's four consecutive output columns of tile j

  // epilogue operands that do not depend on the products: requested first, they arrive under the weight stream
  uint2 xr[NT], dg[NT], db[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    xr[j] = *reinterpret_cast<const uint2*>(p.x + (long)row * D + ncol + 16 * j);
    if (p.dgam) {
      dg[j] = *reinterpret_cast<const uint2*>(p.dgam + (long)row * p.ld_delta + ncol + 16 * j);
      db[j] = *reinterpret_cast<const uint2*>(p.dbet + (long)row * p.ld_delta + ncol + 16 * j);
    }
  }
  // A fragments of the first product: lane -> row m, 8 consecutive k at 32 ks + 8 kg
  const bf16_t* const arow = p.a + (long)row * D + kg * 8;
  auto a_global = [&](int ks) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(arow + ks * 32)); };

  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (NS == 4) {
    // one column tile per wave: its 16 weight fragments (and the 16 activation fragments) are all in flight at once
    uint4 bw[16], aw[16];
    if constexpr (FF) {
      // split feed-forward: this workgroup's 128 hidden columns first, exchanged with the three siblings through device memory
      // (agent-scope dword stores / loads: they pass the non-coherent L2 of another XCD), then its 128 output columns over the whole hidden row
      const uint4* const base1 = p.w1p + ((long)(8 * sp + wave) * 16) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) { bw[ks] = base1[(long)ks * 64]; aw[ks] = *reinterpret_cast<const uint4*>(arow + ks * 32); }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        acc[0] = EVK_MFMA_16x16x32(__builtin_bit_cast(bf16x8, bw[ks]), __builtin_bit_cast(bf16x8, aw[ks]), acc[0], 0, 0, 0);
      const uint4* const base2 = p.w2p + ((long)(8 * sp + wave) * 16) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) bw[ks] = base2[(long)ks * 64];          // the second product's weights travel during the exchange
      const float4 bb = *reinterpret_cast<const float4*>(p.b1 + ncol);
      const uint32_t h01 = pack2bf(fmaxf(acc[0][0] + bb.x, 0.f), fmaxf(acc[0][1] + bb.y, 0.f));
      const uint32_t h23 = pack2bf(fmaxf(acc[0][2] + bb.z, 0.f), fmaxf(acc[0][3] + bb.w, 0.f));
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<uint2*>(hs + m * HP + ncol * 2) = make_uint2(h01, h23);
      unsigned* const xh = p.sync_hid + ((long)rbk * TMR + m) * (D / 2) + ncol / 2;      // hidden row m of the row block, as dwords
      __hip_atomic_store(xh, h01, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(xh + 1, h23, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (tid == 0) cluster_arrive_wait<NS>(p.sync_ctr + p.rbs + rbk);
      __syncthreads();
      // the three other column slices of the 16 hidden rows: 3 x 16 x 64 dwords, 6 per thread
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int e = tid + NTH * i;                       // 0 .. 3071
        const int q = e / (TMR * 64), rem = e - q * (TMR * 64);
        const int rr = rem >> 6, cd = rem & 63;
        const int so = (sp + 1 + q) % NS;                  // sibling slice
        const unsigned val = __hip_atomic_load(p.sync_hid + ((long)rbk * TMR + rr) * (D / 2) + so * 64 + cd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<unsigned*>(hs + rr * HP + (so * 64 + cd) * 4) = val;
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(hs + m * HP + (ks * 32 + kg * 8) * 2));
        acc[0] = EVK_MFMA_16x16x32(__builtin_bit_cast(bf16x8, bw[ks]), af, acc[0], 0, 0, 0);
      }
    } else {
      const uint4* const base = p.w2p + ((long)(8 * sp + wave) * 16) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) { bw[ks] = base[(long)ks * 64]; aw[ks] = *reinterpret_cast<const uint4*>(arow + ks * 32); }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        acc[0] = EVK_MFMA_16x16x32(__builtin_bit_cast(bf16x8, bw[ks]), __builtin_bit_cast(bf16x8, aw[ks]), acc[0], 0, 0, 0);
    }
  } else if constexpr (FF) {
    gemm16(p.w1p, wave, lane, a_global, acc);
    // hidden rows -> LDS (16-bit, as the two-launch path stores them), then they are the A operand of the second product
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 bb = *reinterpret_cast<const float4*>(p.b1 + ncol + 16 * j);
      const float h0 = fmaxf(acc[j][0] + bb.x, 0.f), h1 = fmaxf(acc[j][1] + bb.y, 0.f), h2 = fmaxf(acc[j][2] + bb.z, 0.f), h3 = fmaxf(acc[j][3] + bb.w, 0.f);
      *reinterpret_cast<uint2*>(hs + m * HP + (ncol + 16 * j) * 2) = make_uint2(pack2bf(h0, h1), pack2bf(h2, h3));
      acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    auto a_lds = [&](int ks) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(hs + m * HP + (ks * 32 + kg * 8) * 2)); };
    gemm16(p.w2p, wave, lane, a_lds, acc);
  } else {
    gemm16(p.w2p, wave, lane, a_global, acc);
  }

  // ---- v = product + bias + residual; y = round16(v)
  float v[NT][4];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float4 bb = *reinterpret_cast<const float4*>(p.b2 + ncol + 16 * j);
    v[j][0] = acc[j][0] + bb.x + lo_bf(xr[j].x); v[j][1] = acc[j][1] + bb.y + hi_bf(xr[j].x);
    v[j][2] = acc[j][2] + bb.z + lo_bf(xr[j].y); v[j][3] = acc[j][3] + bb.w + hi_bf(xr[j].y);
    if (live) *reinterpret_cast<uint2*>(p.y + (long)row * D + ncol + 16 * j) = make_uint2(pack2bf(v[j][0], v[j][1]), pack2bf(v[j][2], v[j][3]));
    s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    ss += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
  }
  if (!p.gamma) return;
  // ---- the next (conditional) layer norm over the 512 columns of row m
  float mu, rinv;
  if constexpr (NS == 1) {
    // two passes (mean, then squared deviations) over the 8 waves of the one workgroup that holds the whole row
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (kg == 0) red[0][wave][m] = s;
    __syncthreads();
    mu = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) mu += red[0][w][m];
    mu *= 1.f / D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; q += d * d; }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (kg == 0) red[1][wave][m] = q;
    __syncthreads();
    float qq = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) qq += red[1][w][m];
    rinv = 1.f / (sqrtf(qq / (D - 1)) + p.eps);             // unbiased std, eps added to the std (encoder_decoder.py:100-103)
  } else {
    // (sum, sum of squares) of this workgroup's 128 columns -> device memory -> the three sibling workgroups of the row block
    s += __shfl_xor(s, 16, 64);  s += __shfl_xor(s, 32, 64);
    ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
    if (kg == 0) { red[0][wave][m] = s; red[1][wave][m] = ss; }
    __syncthreads();
    float* const part = p.sync_part + ((long)rbk * NS) * 2 * TMR;            // [NS][2][16]
    if (tid < 2 * TMR) {
      const int which = tid >> 4, r = tid & 15;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[which][w][r];
      tot[which][r] = t;
      __hip_atomic_store(part + (sp * 2 + which) * TMR + r, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                    // (the stores above are issued; the release below orders them)
    if (tid == 0) cluster_arrive_wait<NS>(p.sync_ctr + rbk);
    __syncthreads();
    if (tid < 2 * TMR) {
      const int which = tid >> 4, r = tid & 15;
      float t = tot[which][r];
#pragma unroll
      for (int o = 1; o < NS; ++o) {
        const int q = (sp + o) % NS;
        t += __hip_atomic_load(part + (q * 2 + which) * TMR + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      tot[which][r] = t;
    }
    __syncthreads();
    // (the four workgroups add the partial sums in different orders: s0 + s1 + s2 + s3 is evaluated from each one's own slot onwards;
    // the row statistics may differ in the last bit between column slices of one row -- below the 16-bit rounding of the output)
    mu = tot[0][m] * (1.f / D);
    const float var = fmaxf((tot[1][m] - D * mu * mu) / (D - 1), 0.f);
    rinv = 1.f / (sqrtf(var) + p.eps);
  }
  if (!live) return;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + ncol + 16 * j);
    const float4 b4 = *reinterpret_cast<const float4*>(p.beta + ncol + 16 * j);
    float g[4] = {g4.x, g4.y, g4.z, g4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
    if (p.dgam) {
      g[0] += lo_bf(dg[j].x); g[1] += hi_bf(dg[j].x); g[2] += lo_bf(dg[j].y); g[3] += hi_bf(dg[j].y);
      b[0] += lo_bf(db[j].x); b[1] += hi_bf(db[j].x); b[2] += lo_bf(db[j].y); b[3] += hi_bf(db[j].y);
    }
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mu) * rinv * g[e] + b[e];
    *reinterpret_cast<uint2*>(p.n + (long)row * D + ncol + 16 * j) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
  }
}

// W[512][512] (row-major [out][in], 16-bit) -> fragment-major [n tile 32][k step 16][lane 64][8]: lane = 16 kg + r holds W[16 nt + r][32 ks + 8 kg ..]
__global__ __launch_bounds__(256) void rb_pack_kernel(const bf16_t* __restrict__ w, uint4* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;            // one uint4 (8 elements) per thread: 32 * 16 * 64 = 32768
  if (i >= 32 * 16 * 64) return;
  const int lane = i & 63, ks = (i >> 6) & 15, nt = i >> 10;
  const int r = lane & 15, kg = lane >> 4;
  out[i] = *reinterpret_cast<const uint4*>(w + (long)(nt * 16 + r) * D + ks * 32 + kg * 8);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int evk_decode_rb_pack(const void* w, void* packed, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(w && packed && al16(w) && al16(packed), "decode_rb_pack: null / misaligned");
  hipLaunchKernelGGL(rb_pack_kernel, dim3(128), dim3(256), 0, s, (const bf16_t*)w, (uint4*)packed);
  return evk_check_launch("decode_rb_pack");
}

int64_t evk_decode_rowblock_sync_bytes(int32_t R) {
  const int64_t rbs = (R + TMR - 1) / TMR;
  return rbs * 256 + rbs * 4 * 2 * TMR * 4 + rbs * TMR * D * 2;      // two counters per row block (head of the buffer) + partial sums + hidden rows
}

int evk_decode_rowblock(const void* a, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2, const void* x, void* y,
                        const float* gamma, const float* beta, const void* dgam, const void* dbet, int64_t ld_delta, float eps, void* n, int32_t R,
                        void* sync_ws, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(a && w2_packed && b2 && x && y && R > 0, "decode_rowblock: null/empty argument");
  EVK_REQUIRE((w1_packed == nullptr) == (b1 == nullptr), "decode_rowblock: the first stage needs weights and bias");
  EVK_REQUIRE((gamma == nullptr) == (beta == nullptr) && (gamma == nullptr) == (n == nullptr) && (dgam == nullptr) == (dbet == nullptr) && (!dgam || gamma),
              "decode_rowblock: norm operands come together");
  EVK_REQUIRE(al16(a) && al16(w2_packed) && (!w1_packed || al16(w1_packed)) && al16(x) && al16(y) && (!n || al16(n)) && al16(b2) && (!b1 || al16(b1)) &&
              (!gamma || (al16(gamma) && al16(beta))) && (!dgam || (al16(dgam) && al16(dbet) && ld_delta % 8 == 0 && ld_delta >= D)) &&
              (!sync_ws || al16(sync_ws)), "decode_rowblock: 16-byte aligned operands, delta pitch a multiple of 8");
  RbP p{(const bf16_t*)a, (const uint4*)w1_packed, b1, (const uint4*)w2_packed, b2, (const bf16_t*)x, (bf16_t*)y, gamma, beta, (const bf16_t*)dgam,
        (const bf16_t*)dbet, (long)ld_delta, (bf16_t*)n, eps, R, nullptr, nullptr, nullptr, 0};
  ProfScope ps(EVK_FAM_GEMM, s, 2.0 * R * D * D * (w1_packed ? 2 : 1));
  const int rbs = (R + TMR - 1) / TMR;
  // the projection variant splits a row block over four workgroups when the caller provides the (zero-initialised, persistent) exchange
  // buffer and the whole grid is resident at once (one 512-thread workgroup per CU suffices: 4 x rbs <= 256); EVK_DECODE_RB_SPLIT=0 disables
  static const int split_on = evk_tunable("EVK_DECODE_RB_SPLIT", 1);
  p.rbs = rbs;
  // The four workgroups of a row block wait for each other (cluster_arrive_wait), so ALL 4 x rbs workgroups of the launch must be resident
  // together: the grid is held to what the device admits at once -- compute units x the occupancy the runtime reports for this kernel, queried
  // once per device -- and a stream that may only use a share of the compute units (evk_stream_create_cu_mask) never takes the split variant.
  // Within that bound forward progress needs nothing else: a workgroup that waits holds a slot, never a lock, and workgroups are dispatched in
  // index order, so the siblings of a resident workgroup are the next ones to be placed.
  static EvkDeviceOnce cap_once;
  const long cap = reinterpret_cast<long>(cap_once.get([]() -> void* {
    int dev = 0, cus = 0, per_cu_a = 0, per_cu_b = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_a, decode_rowblock_kernel<false, 4>, NTH, 0) != hipSuccess) per_cu_a = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, decode_rowblock_kernel<true, 4>, NTH, 0) != hipSuccess) per_cu_b = 0;
    (void)hipGetLastError();
    const long c = (long)cus * (per_cu_a < per_cu_b ? per_cu_a : per_cu_b);
    return reinterpret_cast<void*>(c > 0 ? c : 1L);          // (non-null: "queried"; 1 = nothing fits, the split variant is never taken)
  }));
  if (sync_ws && split_on && 4L * rbs <= cap && !evk_stream_is_cu_masked(s)) {
    p.sync_ctr = reinterpret_cast<unsigned*>(sync_ws);                                                        // [2][rbs]
    p.sync_part = reinterpret_cast<float*>(reinterpret_cast<char*>(sync_ws) + (size_t)rbs * 256);
    p.sync_hid = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(sync_ws) + (size_t)rbs * 256 + (size_t)rbs * 4 * 2 * TMR * 4);
    if (w1_packed) hipLaunchKernelGGL((decode_rowblock_kernel<true, 4>), dim3((unsigned)(4 * rbs)), dim3(NTH), 0, s, p);
    else hipLaunchKernelGGL((decode_rowblock_kernel<false, 4>), dim3((unsigned)(4 * rbs)), dim3(NTH), 0, s, p);
  } else if (w1_packed) {
    hipLaunchKernelGGL((decode_rowblock_kernel<true, 1>), dim3((unsigned)rbs), dim3(NTH), 0, s, p);
  } else {
    hipLaunchKernelGGL((decode_rowblock_kernel<false, 1>), dim3((unsigned)rbs), dim3(NTH), 0, s, p);
  }
  return evk_check_launch("decode_rowblock");
}

}  // extern "C"
