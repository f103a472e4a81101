// attn.hip -- fused multi-head attention core, forward and backward (gfx950).
//
// Replaces, per attention site, the three launches  scores = alpha * Q.K^T (f32 in HBM)  ->  masked softmax (+dropout)  ->
// P.V  of the first engine (and the five launches of its backward) by ONE kernel each way:
//   modules/encoder_decoder.py:20-28 `attention` (R2Gen encoder / decoder, d_k = 64, causal + key masks),
//   models/language_encoder/bert_model.py:210-349 BertSelfAttention (text encoder d_k = 64, fusion layers d_k = 256, additive key
//   masks), modules/utils_v0511.py:251-279 ScaledDotProductAttention (multi-view fusion: 8 heads of d_k = 2048, scale 1/sqrt(2048)),
//   HF GPT2Attention of the distilgpt2 backend (models/language_encoder/language_model.py:161-282).
// Sequences on this path are short (T <= 145 queries, S <= 4 x 145 keys) but heads are up to 2048 wide, so the tile that stays
// on chip is the SCORE tile, not an output accumulator: a workgroup owns 32 query rows of one (batch, head) and
//   1. accumulates its 32 x S scores with MFMA over the head dimension -- K tiles of 64 keys x 64 dims staged through LDS and
//      shared by the four waves, Q fragments straight from global memory -- into an f32 LDS image (never written to HBM),
//   2. runs the masked softmax row by row with wavefront shuffles (one wave per row, 64 lanes striding the keys), applies the
//      stateless-hash dropout, writes P (16-bit, the backward's input) and keeps P' = dropout(P) in LDS as the next A operand,
//   3. multiplies P' by V in 64-wide slices of the head dimension -- V tiles of 64 keys x 64 dims staged through LDS as they lie
//      in memory and read as MFMA B fragments with the transposing LDS read ds_read_b64_tr_b16 -- and stores the 32 x 64 output
//      slice through LDS with 16-byte rows.
// The backward kernel is the same skeleton: dP = dO.V^T (phase 1 with dO, V), dS = P * (dropout'(dP) - rowsum(.)) * alpha
// (phase 2; dS goes to HBM in 16 bits for the dK product), dQ = dS.K (phase 3 with K).  dK = dS^T.Q and dV = P'^T.dO are two
// K-strided batched GEMMs of the GEMM family (gemm.hip) over the stored 16-bit P' / dS.
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t hash32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}
__device__ __forceinline__ bool keep_elem(uint64_t seed, uint64_t idx, float p) {
  return (hash32(seed * 0x9E3779B97F4A7C15ULL + idx) >> 8) * (1.f / 16777216.f) >= p;
}

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ s16x4 lds_tr_read(const bf16_t* generic_lds_ptr) {
  lds_s16x4* p = (lds_s16x4*)(__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)generic_lds_ptr;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
}

constexpr int TQ = 32;            // query rows per workgroup
constexpr int TS = 64;            // keys per staged tile
constexpr int TD = 64;            // head dims per staged tile
constexpr int STG = TD + 8;       // LDS row stride of a staged tile (16-bit elements): 144 B rows, 16-byte aligned
constexpr int MAX_SPAD = 640;

struct AttnP {
  const bf16_t* q;   // fwd: Q [B][T][HD]      bwd: dO [B][T][HD]
  const bf16_t* k;   // fwd: K [B][S][HD]      bwd: V
  const bf16_t* v;   // fwd: V [B][S][HD]      bwd: K
  bf16_t* out;       // fwd: O [B][T][HD]      bwd: dQ
  bf16_t* P;         // [B][H][T][Sp] pre-dropout probabilities: fwd writes, bwd reads
  bf16_t* Pd;        // fwd: post-dropout probabilities (null when p_drop == 0); bwd: dS out
  const unsigned char* mask; long mBo; int mQ; int causal;
  int B, H, T, S, Sp, Spad, HD, dh;
  float scale, p_drop; unsigned long long seed; const unsigned long long* epoch;
};

// 64 x 64 tile of X[b][row0 + r][h*dh + c0 + c] (rows clamped to S - 1) -> LDS [64][STG], in two halves: the loads of the NEXT tile are
// issued while the current one is multiplied (a synchronous stage costs a memory round trip per four MFMAs of a wave: the multi-view
// fusion site, 8 heads of 2048 dims, spent 540 us forward / 530 us backward almost entirely waiting on its 2 x 96 tiles)
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ X, long bbase, int row0, int S, int HD, int col0, int tid, uint4& x0, uint4& x1) {
  const int c = (tid & 7) * 8;
  int r0 = row0 + (tid >> 3), r1 = r0 + 32;
  r0 = r0 < S ? r0 : S - 1;
  r1 = r1 < S ? r1 : S - 1;
  x0 = *reinterpret_cast<const uint4*>(X + bbase + (long)r0 * HD + col0 + c);
  x1 = *reinterpret_cast<const uint4*>(X + bbase + (long)r1 * HD + col0 + c);
}
__device__ __forceinline__ void tile_store(bf16_t* lds, int tid, const uint4& x0, const uint4& x1) {
  const int r = tid >> 3, c = (tid & 7) * 8;
  *reinterpret_cast<uint4*>(lds + r * STG + c) = x0;
  *reinterpret_cast<uint4*>(lds + (r + 32) * STG + c) = x1;
}

template <bool BWD>
__global__ __launch_bounds__(256) void attn_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ldS = p.Spad + 4, ldP = p.Spad + 8;
  float* Ssc = reinterpret_cast<float*>(smem);                                   // [TQ][ldS] f32 scores / dP
  bf16_t* Pp = reinterpret_cast<bf16_t*>(smem + (size_t)TQ * ldS * 4);            // [TQ][ldP] 16-bit P' / dS
  bf16_t* stg = Pp + (size_t)TQ * ldP;                                           // [64][STG] staged K / V tile, output slice
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int t0 = blockIdx.x * TQ, h = blockIdx.y, b = blockIdx.z;
  const int mi = wave & 1, nh = wave >> 1;                       // wave -> query tile mi (16 rows), key/dim tiles nh and nh + 2
  const long qbase = (long)b * p.T * p.HD + (long)h * p.dh;      // + t * HD + d
  const long kbase = (long)b * p.S * p.HD + (long)h * p.dh;
  int qrow = t0 + mi * 16 + li;
  qrow = qrow < p.T ? qrow : p.T - 1;
  const bf16_t* qptr = p.q + qbase + (long)qrow * p.HD + g * 8;

  // ---- phase 1: scores (fwd: Q.K^T; bwd: dO.V^T) -> Ssc.  One flat loop over (key tile, dim chunk) with a ring of four tiles in
  // registers: the K tile and the Q fragments of step i + 4 are requested when step i goes to LDS (one step is two barriers around
  // four MFMAs per wave -- far shorter than a memory round trip).
  {
    const int nkc = p.dh / TD, nit = (p.Spad / TS) * nkc;
    uint4 xa0, xa1, xb0, xb1, xc0, xc1, xd0, xd1;
    bf16x8 qa0, qa1, qb0, qb1, qc0, qc1, qd0, qd1;
    int lk = 0, ls = 0, issued = 0;                       // the next tile to request: dim chunk, first key, count
    auto issue = [&](uint4& x0, uint4& x1, bf16x8& q0, bf16x8& q1) {
      if (issued < nit) {
        tile_load(p.k, kbase, ls, p.S, p.HD, lk * TD, tid, x0, x1);
        q0 = *reinterpret_cast<const bf16x8*>(qptr + lk * TD);
        q1 = *reinterpret_cast<const bf16x8*>(qptr + lk * TD + 32);
      }
      ++issued;
      if (++lk == nkc) { lk = 0; ls += TS; }
    };
    issue(xa0, xa1, qa0, qa1); issue(xb0, xb1, qb0, qb1); issue(xc0, xc1, qc0, qc1); issue(xd0, xd1, qd0, qd1);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float sc = BWD ? 1.f : p.scale;
    int kci = 0, s0 = 0;
    auto step = [&](uint4& x0, uint4& x1, bf16x8& q0, bf16x8& q1) {
      __syncthreads();
      tile_store(stg, tid, x0, x1);
      const bf16x8 a0 = q0, a1 = q1;
      issue(x0, x1, q0, q1);                                // this slot's next tenant: step + 4
      __syncthreads();
      const bf16_t* r0 = stg + (nh * 16 + li) * STG + g * 8;
      const bf16_t* r1 = stg + ((nh + 2) * 16 + li) * STG + g * 8;
      acc0 = EVK_MFMA_16x16x32(a0, *reinterpret_cast<const bf16x8*>(r0), acc0, 0, 0, 0);
      acc0 = EVK_MFMA_16x16x32(a1, *reinterpret_cast<const bf16x8*>(r0 + 32), acc0, 0, 0, 0);
      acc1 = EVK_MFMA_16x16x32(a0, *reinterpret_cast<const bf16x8*>(r1), acc1, 0, 0, 0);
      acc1 = EVK_MFMA_16x16x32(a1, *reinterpret_cast<const bf16x8*>(r1 + 32), acc1, 0, 0, 0);
      if (kci == nkc - 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* row = Ssc + (mi * 16 + g * 4 + j) * ldS + s0 + li;
          row[nh * 16] = acc0[j] * sc;
          row[(nh + 2) * 16] = acc1[j] * sc;
        }
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (++kci == nkc) { kci = 0; s0 += TS; }
    };
    for (int it = 0; it < nit; it += 4) {
      step(xa0, xa1, qa0, qa1);
      if (it + 1 < nit) step(xb0, xb1, qb0, qb1);
      if (it + 2 < nit) step(xc0, xc1, qc0, qc1);
      if (it + 3 < nit) step(xd0, xd1, qd0, qd1);
    }
  }
  __syncthreads();

  // ---- phase 2: one wave per row, 64 lanes striding the keys
  const unsigned long long seed = evk_mix_seed(p.seed, p.epoch);
  const float dsc = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
  for (int r = wave * 8; r < wave * 8 + 8; ++r) {
    const int t = t0 + r;
    bf16_t* prow = Pp + r * ldP;
    if (t >= p.T) {
      for (int c = lane; c < p.Spad; c += 64) prow[c] = 0;
      continue;
    }
    const long grow = ((long)(b * p.H + h) * p.T + t) * p.Sp;        // row of the [B][H][T][Sp] images
    const float* srow = Ssc + r * ldS;
    if (!BWD) {
      const unsigned char* mk = p.mask ? p.mask + (long)b * p.mBo + (long)t * p.mQ : nullptr;
      float v[MAX_SPAD / 64];
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < MAX_SPAD / 64; ++i) {
        const int c = lane + 64 * i;
        float x = -INFINITY;
        if (c < p.S && !(mk && !mk[c]) && !(p.causal && c > t)) x = srow[c];
        v[i] = x;
        mx = fmaxf(mx, x);
      }
      mx = wave_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < MAX_SPAD / 64; ++i) { v[i] = v[i] == -INFINITY ? 0.f : __expf(v[i] - mx); sum += v[i]; }
      sum = wave_sum(sum);
      const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
      for (int i = 0; i < MAX_SPAD / 64; ++i) {
        const int c = lane + 64 * i;
        if (c < p.Spad) {
          const float pr = v[i] * inv;
          float pd = pr;
          if (p.p_drop > 0.f) pd = keep_elem(seed, (uint64_t)(grow + c), p.p_drop) ? pr * dsc : 0.f;
          const bf16_t hp = f2bf(pd);
          prow[c] = hp;
          if (c < p.Sp) {
            p.P[grow + c] = f2bf(pr);
            if (p.Pd) p.Pd[grow + c] = hp;
          }
        }
      }
    } else {
      float d[MAX_SPAD / 64], pv[MAX_SPAD / 64];
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < MAX_SPAD / 64; ++i) {
        const int c = lane + 64 * i;
        d[i] = 0.f; pv[i] = 0.f;
        if (c < p.S) {
          float gq = srow[c];
          if (p.p_drop > 0.f) gq = keep_elem(seed, (uint64_t)(grow + c), p.p_drop) ? gq * dsc : 0.f;
          pv[i] = bf2f(p.P[grow + c]);
          d[i] = gq;
          dot += gq * pv[i];
        }
      }
      dot = wave_sum(dot);
#pragma unroll
      for (int i = 0; i < MAX_SPAD / 64; ++i) {
        const int c = lane + 64 * i;
        if (c < p.Spad) {
          const bf16_t hs = f2bf(pv[i] * (d[i] - dot) * p.scale);
          prow[c] = hs;
          if (c < p.Sp) p.Pd[grow + c] = hs;
        }
      }
    }
  }

  // ---- phase 3: out[32 x dh] = Pp[32 x S] . X[S x dh]  (fwd: X = V; bwd: X = K), 64 dims at a time; same prefetch
  const bf16_t* arow = Pp + (mi * 16 + li) * ldP + g * 8;
  const int q4 = li >> 2, p4 = li & 3;
  {
    const int nst = p.Spad / TS, nit = (p.dh / TD) * nst;
    uint4 xa0, xa1, xb0, xb1, xc0, xc1, xd0, xd1;
    int lsi = 0, ld0 = 0, issued = 0;                     // the next tile to request: key tile, first dim, count
    auto issue = [&](uint4& x0, uint4& x1) {
      if (issued < nit) tile_load(p.v, kbase, lsi * TS, p.S, p.HD, ld0, tid, x0, x1);
      ++issued;
      if (++lsi == nst) { lsi = 0; ld0 += TD; }
    };
    issue(xa0, xa1); issue(xb0, xb1); issue(xc0, xc1); issue(xd0, xd1);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int si = 0, d0 = 0;
    auto step = [&](uint4& x0, uint4& x1) {
      __syncthreads();
      tile_store(stg, tid, x0, x1);
      issue(x0, x1);
      __syncthreads();
      const int s0 = si * TS;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + s0 + ks * 32);
        // B[k = 8g + j][n = li] = tile[ks*32 + 8g + j][ncol + li]: two transposing reads of 4 k-rows x 16 columns each
        const bf16_t* tb = stg + (ks * 32 + g * 8 + q4) * STG + p4 * 4;
        const s16x4 lo0 = lds_tr_read(tb + nh * 16), hi0 = lds_tr_read(tb + 4 * STG + nh * 16);
        const s16x4 lo1 = lds_tr_read(tb + (nh + 2) * 16), hi1 = lds_tr_read(tb + 4 * STG + (nh + 2) * 16);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 b0 = {lo0[0], lo0[1], lo0[2], lo0[3], hi0[0], hi0[1], hi0[2], hi0[3]};
        const s16x8 b1 = {lo1[0], lo1[1], lo1[2], lo1[3], hi1[0], hi1[1], hi1[2], hi1[3]};
        acc0 = EVK_MFMA_16x16x32(a, __builtin_bit_cast(bf16x8, b0), acc0, 0, 0, 0);
        acc1 = EVK_MFMA_16x16x32(a, __builtin_bit_cast(bf16x8, b1), acc1, 0, 0, 0);
      }
      if (si == nst - 1) {            // the 32 x 64 output slice of this dim chunk leaves through the staging tile
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16_t* orow = stg + (mi * 16 + g * 4 + j) * STG + li;
          orow[nh * 16] = f2bf(acc0[j]);
          orow[(nh + 2) * 16] = f2bf(acc1[j]);
        }
        __syncthreads();
        {
          const int r = tid >> 3, c = (tid & 7) * 8;
          if (t0 + r < p.T)
            *reinterpret_cast<uint4*>(p.out + qbase + (long)(t0 + r) * p.HD + d0 + c) = *reinterpret_cast<const uint4*>(stg + r * STG + c);
        }
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (++si == nst) { si = 0; d0 += TD; }
    };
    for (int it = 0; it < nit; it += 4) {
      step(xa0, xa1);
      if (it + 1 < nit) step(xb0, xb1);
      if (it + 2 < nit) step(xc0, xc1);
      if (it + 3 < nit) step(xd0, xd1);
    }
  }
}

size_t attn_lds_bytes(int Spad) { return (size_t)TQ * (Spad + 4) * 4 + (size_t)TQ * (Spad + 8) * 2 + (size_t)64 * STG * 2; }

template <bool BWD>
int launch(const AttnP& p, hipStream_t s, const char* what) {
  const size_t lds = attn_lds_bytes(p.Spad);
  static size_t configured = 0;
  if (lds > 65536 && lds > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds_bytes(MAX_SPAD)) != hipSuccess) {
      evk_set_error("%s: hipFuncSetAttribute(max dynamic LDS) failed", what);
      return EVK_ELAUNCH;
    }
    configured = attn_lds_bytes(MAX_SPAD);
  }
  ProfScope ps(EVK_FAM_GEMM, s, 4.0 * p.B * p.H * (double)p.T * p.S * p.dh);
  hipLaunchKernelGGL(attn_kernel<BWD>, dim3((p.T + TQ - 1) / TQ, p.H, p.B), dim3(256), lds, s, p);
  return evk_check_launch(what);
}

int check_shape(int64_t B, int heads, int T, int S, int dh, const char* what) {
  EVK_REQUIRE(B > 0 && heads > 0 && T > 0 && S > 0 && dh > 0, "%s: empty problem", what);
  EVK_REQUIRE(dh % 64 == 0, "%s: head dim %d is not a multiple of 64", what, dh);
  EVK_REQUIRE((S + 63) / 64 * 64 <= MAX_SPAD, "%s: %d keys (max %d)", what, S, MAX_SPAD);
  EVK_REQUIRE(B <= 65535 && heads <= 65535, "%s: grid too large", what);
  return EVK_OK;
}

}  // namespace

extern "C" {

int evk_attention_supported(int32_t S, int32_t dh) { return dh > 0 && dh % 64 == 0 && S > 0 && (S + 63) / 64 * 64 <= MAX_SPAD; }

int evk_attention_fwd(const void* q, const void* k, const void* v, void* out, void* probs, void* probs_dropped, const unsigned char* mask,
                      int64_t mask_batch_stride, int32_t mask_q_stride, int32_t causal, int64_t B, int32_t heads, int32_t T, int32_t S,
                      int32_t dh, float scale, float p_drop, uint64_t seed, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(q && k && v && out && probs, "attention_fwd: null operand");
  if (int e = check_shape(B, heads, T, S, dh, "attention_fwd")) return e;
  EVK_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || probs_dropped), "attention_fwd: dropout needs the second probability image");
  AttnP p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, (bf16_t*)probs, p_drop > 0.f ? (bf16_t*)probs_dropped : nullptr,
          mask, mask_batch_stride, mask_q_stride, causal, (int)B, heads, T, S, (S + 7) / 8 * 8, (S + 63) / 64 * 64, heads * dh, dh, scale, p_drop,
          seed, evk_seed_epoch_ptr()};
  return launch<false>(p, s, "attention_fwd");
}

int evk_attention_bwd(const void* dout, const void* k, const void* v, const void* probs, void* ds, void* dq, int64_t B, int32_t heads,
                      int32_t T, int32_t S, int32_t dh, float scale, float p_drop, uint64_t seed, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(dout && k && v && probs && ds && dq, "attention_bwd: null operand");
  if (int e = check_shape(B, heads, T, S, dh, "attention_bwd")) return e;
  AttnP p{(const bf16_t*)dout, (const bf16_t*)v, (const bf16_t*)k, (bf16_t*)dq, (bf16_t*)const_cast<void*>(probs), (bf16_t*)ds, nullptr, 0, 0, 0,
          (int)B, heads, T, S, (S + 7) / 8 * 8, (S + 63) / 64 * 64, heads * dh, dh, scale, p_drop, seed, evk_seed_epoch_ptr()};
  return launch<true>(p, s, "attention_bwd");
}

}  // extern "C"
