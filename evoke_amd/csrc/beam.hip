// beam.hip -- one kernel per generated token for ALL the beam bookkeeping of CaptionModel.beam_search / beam_step
// (modules/caption_model.py:51-106: candidate scores = running sum + log-prob, flat descending sort over beam x (V+1), top-beam,
// state / sequence reorder; :174-189: beams that emit [EOS] -- or every beam at the last position -- are recorded as finished
// with p = running sum and get -1000 on their running sum; modules/att_model.py:133-135: the best finished beam wins).
// The reference does this with a per-sample Python loop and .item() syncs; the first engine with ~15 torch glue ops per token
// inside the captured decode step (gather / where / max / index_copy kernels).  Here: one workgroup per sample --
//   1. every thread keeps the top-`beam` of its strided share of the beam x (V+1) candidates, `beam` rounds of a block arg-max
//      (ties -> lowest flat index, i.e. torch.sort(descending=True)'s order on distinct scores) pick the winners;
//   2. the sample's old sequences, cache row tables and relational-memory rows are staged in LDS and rewritten IN PLACE in the
//      winners' order (a sample's hypotheses only permute among themselves), the new token goes to position *pos;
//   3. finished-beam tracking and the -1000 penalty, the next step's input tokens.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int KMAX = 8;

struct BeamP {
  const float* logp; int ld;          // [B*beam][ld] f32 log-probs of the step, V1 valid columns
  int V1, beam, max_len, eos, force_end;
  const long long* pos;               // device scalar: position being written
  float* beam_sum;                    // [B][beam] running sums (in / out)
  long long* beam_seq;                // [B][beam][max_len] (in place)
  float* best_p; long long* best_seq; // [B], [B][max_len]
  long long* words;                   // [B*beam] out: tokens fed to the next decoder step
  bf16_t* mem; int mem_row;           // relational-memory state [B*beam][mem_row] (in place) or null
  int* anc; int anc_cols;             // cache row table [B*beam][anc_cols] (in place) or null
  bf16_t* mem2;                       // optional second state row per hypothesis (same geometry as mem)
  long long* pos_out; int* ticket;    // advance: the LAST workgroup to finish writes *pos_out = pos + 1 (every workgroup has read pos by then)
};

__global__ __launch_bounds__(256) void beam_step_kernel(const BeamP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float s_val[4]; __shared__ int s_idx[4];
  __shared__ float win_v[KMAX]; __shared__ int win_i[KMAX];
  __shared__ int s_best;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int beam = p.beam;
  // ---- 1. top-`beam` candidates
  float bv[KMAX]; int bi[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { bv[j] = -INFINITY; bi[j] = 0x7fffffff; }
  // a thread's share of one source beam is requested in one batch (UNR loads in flight), then inserted: the step is latency bound, and
  // one dependent L2 round trip per candidate (23 per thread at beam 4, V = 1444) was most of this kernel's time.  Which thread sees
  // which candidate does not matter: the order (score, then lowest flat index) is total.
  constexpr int UNR = 8;
  for (int src = 0; src < beam; ++src) {
    const float base = p.beam_sum[b * beam + src];
    const float* row = p.logp + (long)(b * beam + src) * p.ld;
    for (int w0 = tid; w0 < p.V1; w0 += 256 * UNR) {
      float lv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) { const int w = w0 + 256 * u; lv[u] = w < p.V1 ? row[w] : 0.f; }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int w = w0 + 256 * u;
        if (w >= p.V1) continue;
        float v = base + lv[u];
        int id = src * p.V1 + w;
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
          if (j < beam && (v > bv[j] || (v == bv[j] && id < bi[j]))) { const float tv = bv[j]; const int ti = bi[j]; bv[j] = v; bi[j] = id; v = tv; id = ti; }
      }
    }
  }
  for (int r = 0; r < beam; ++r) {
    float best = bv[0]; int bid = bi[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bid, o, 64);
      if (ov > best || (ov == best && oi < bid)) { best = ov; bid = oi; }
    }
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = bid; }
    __syncthreads();
    if (tid == 0) {
      float fv = s_val[0]; int fi = s_idx[0];
      for (int w = 1; w < 4; ++w) if (s_val[w] > fv || (s_val[w] == fv && s_idx[w] < fi)) { fv = s_val[w]; fi = s_idx[w]; }
      win_v[r] = fv; win_i[r] = fi;
    }
    __syncthreads();
    if (bi[0] == win_i[r]) {            // the owner pops its head
#pragma unroll
      for (int j = 0; j + 1 < KMAX; ++j) { bv[j] = bv[j + 1]; bi[j] = bi[j + 1]; }
      bv[KMAX - 1] = -INFINITY; bi[KMAX - 1] = 0x7fffffff;
    }
  }
  // ---- 2. stage the sample's rows, rewrite them in the winners' order
  const long pos = p.pos[0];
  long long* seq = p.beam_seq + (long)b * beam * p.max_len;
  long long* l_seq = reinterpret_cast<long long*>(smem);                               // [beam][max_len]
  int* l_anc = reinterpret_cast<int*>(l_seq + (size_t)beam * p.max_len);               // [beam][anc_cols]
  uint32_t* l_mem = reinterpret_cast<uint32_t*>(l_anc + (size_t)beam * (p.anc ? p.anc_cols : 0));   // [beam][mem_row / 2]
  for (int i = tid; i < beam * p.max_len; i += 256) l_seq[i] = seq[i];
  if (p.anc) for (int i = tid; i < beam * p.anc_cols; i += 256) l_anc[i] = p.anc[(long)b * beam * p.anc_cols + i];
  const int mw = p.mem_row / 2;
  if (p.mem) {
    const uint32_t* gm = reinterpret_cast<const uint32_t*>(p.mem + (long)b * beam * p.mem_row);
    for (int i = tid; i < beam * mw; i += 256) l_mem[i] = gm[i];
  }
  __syncthreads();
  for (int i = tid; i < beam * p.max_len; i += 256) {
    const int j = i / p.max_len, c = i - j * p.max_len;
    const int src = win_i[j] / p.V1;
    seq[i] = c == pos ? (long long)(win_i[j] - src * p.V1) : l_seq[src * p.max_len + c];
  }
  if (p.anc)
    for (int i = tid; i < beam * p.anc_cols; i += 256) {
      const int j = i / p.anc_cols, c = i - j * p.anc_cols;
      p.anc[(long)b * beam * p.anc_cols + i] = l_anc[(win_i[j] / p.V1) * p.anc_cols + c];
    }
  if (p.mem) {
    uint32_t* gm = reinterpret_cast<uint32_t*>(p.mem + (long)b * beam * p.mem_row);
    for (int i = tid; i < beam * mw; i += 256) {
      const int j = i / mw, c = i - j * mw;
      gm[i] = l_mem[(win_i[j] / p.V1) * mw + c];
    }
  }
  if (p.mem2) {                         // second state: through the same staging buffer, after the first has left it
    __syncthreads();
    uint32_t* gm = reinterpret_cast<uint32_t*>(p.mem2 + (long)b * beam * p.mem_row);
    for (int i = tid; i < beam * mw; i += 256) l_mem[i] = gm[i];
    __syncthreads();
    for (int i = tid; i < beam * mw; i += 256) {
      const int j = i / mw, c = i - j * mw;
      gm[i] = l_mem[(win_i[j] / p.V1) * mw + c];
    }
  }
  // ---- 3. finished beams (best p so far; the earlier / lower beam index wins ties), the -1000 penalty, next tokens
  if (tid == 0) {
    float pv = -INFINITY; int pi = -1;
    for (int j = 0; j < beam; ++j) {
      const int word = win_i[j] % p.V1;
      const bool end = p.force_end || word == p.eos;
      if (end && win_v[j] > pv) { pv = win_v[j]; pi = j; }
      p.beam_sum[b * beam + j] = win_v[j] - (end ? 1000.f : 0.f);
      p.words[b * beam + j] = word;
    }
    int take = -1;
    if (pi >= 0 && pv > p.best_p[b]) { p.best_p[b] = pv; take = pi; }
    s_best = take;
  }
  __syncthreads();
  if (s_best >= 0) {
    const int j = s_best, src = win_i[j] / p.V1;
    for (int c = tid; c < p.max_len; c += 256)
      p.best_seq[(long)b * p.max_len + c] = c == pos ? (long long)(win_i[j] % p.V1) : l_seq[src * p.max_len + c];
  }
  if (p.pos_out) {
    // the next decoder step writes its keys / values at position pos + 1 of every hypothesis's OWN cache row: seed that column of the row
    // table here (it replaces an index_copy launch), then let the last workgroup advance the position (it replaces an add launch)
    if (p.anc && pos + 1 < p.anc_cols && tid < beam) p.anc[(long)(b * beam + tid) * p.anc_cols + pos + 1] = b * beam + tid;
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      if (atomicAdd(p.ticket, 1) == (int)gridDim.x - 1) { *p.ticket = 0; p.pos_out[0] = pos + 1; }
    }
  }
}

// The same step with ONE memory round trip in front of the selection: the sample's state rows (sequences, cache row table, memory) and
// all beam x (V+1) candidates are requested together -- each wave owns the candidates of one source beam in registers (V + 1 <= 1536) --,
// the top-`beam` of every source row are found by `beam` rounds of a wave arg-max and merged by wave 0 (beam x beam candidates).  The
// kernel above walks the candidates beam by beam (a round trip each), keeps a sorted list per thread and stages the state afterwards
// (two more round trips): 29 us of a ~400 us token step; this one takes a third of that.  Same results bit for bit: the order (score,
// then lowest flat index) is total.
constexpr int CMAX = 24;        // candidates per lane and source row

// arg-max (value, then LOWEST index) over the 64 lanes, result in every lane: four DPP exchanges inside the rows of 16 lanes (a few cycles
// each; __shfl_xor goes through the LDS crossbar, ~100 cycles per dependent step: 12 of them per round were most of this kernel), then the
// four row results are read as scalars
template <int CTRL>
__device__ __forceinline__ void dpp_argmax_step(float& v, int& i) {
  const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
  const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, true);
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
  dpp_argmax_step<0xB1>(v, i);       // quad_perm [1,0,3,2]
  dpp_argmax_step<0x4E>(v, i);       // quad_perm [2,3,0,1]
  dpp_argmax_step<0x141>(v, i);      // row_half_mirror
  dpp_argmax_step<0x140>(v, i);      // row_mirror: every lane of a row holds the row's result
  float bv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  int bi = __builtin_amdgcn_readlane(i, 0);
#pragma unroll
  for (int r = 1; r < 4; ++r) {
    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * r));
    const int oi = __builtin_amdgcn_readlane(i, 16 * r);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  v = bv; i = bi;
}

__global__ __launch_bounds__(256) void beam_step_fast_kernel(const BeamP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float c_val[KMAX * KMAX]; __shared__ int c_idx[KMAX * KMAX];
  __shared__ float win_v[KMAX]; __shared__ int win_i[KMAX];
  __shared__ int s_best;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int beam = p.beam, V1 = p.V1;
  const long pos = p.pos[0];
  // ---- 0. everything this workgroup reads, requested at once
  long long* seq = p.beam_seq + (long)b * beam * p.max_len;
  long long* l_seq = reinterpret_cast<long long*>(smem);                               // [beam][max_len]
  int* l_anc = reinterpret_cast<int*>(l_seq + (size_t)beam * p.max_len);               // [beam][anc_cols]
  uint32_t* l_mem = reinterpret_cast<uint32_t*>(l_anc + (size_t)beam * (p.anc ? p.anc_cols : 0));   // [beam][mem_row / 2]
  const int mw = p.mem_row / 2;
  uint32_t* l_mem2 = l_mem + (size_t)beam * (p.mem ? mw : 0);
  float x[2][CMAX];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int src = wave + 4 * u;
    const float* row = p.logp + (long)(b * beam + min(src, beam - 1)) * p.ld;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) { const int c = lane + 64 * i; x[u][i] = (src < beam && c < V1) ? row[c] : -INFINITY; }
  }
  float base[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) base[u] = p.beam_sum[b * beam + min(wave + 4 * u, beam - 1)];
  // state rows -> LDS in 16-byte pieces, eight per thread in flight (a word-by-word loop of unknown trip count is a chain of dependent
  // round trips: 24 of them for the 24 KB of f32 memory rows were most of this kernel's first version)
  auto stage = [&](void* dst, const void* src, int nbytes) {
    if ((nbytes & 15) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
      const int n4 = nbytes >> 4;
      for (int base = 0; base < n4; base += 256 * 8) {
        uint4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = base + tid + 256 * u; if (i < n4) t[u] = reinterpret_cast<const uint4*>(src)[i]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = base + tid + 256 * u; if (i < n4) reinterpret_cast<uint4*>(dst)[i] = t[u]; }
      }
    } else {
      for (int i = tid; i < (nbytes >> 2); i += 256) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
    }
  };
  stage(l_seq, seq, beam * p.max_len * 8);
  if (p.anc) stage(l_anc, p.anc + (long)b * beam * p.anc_cols, beam * p.anc_cols * 4);
  if (p.mem) stage(l_mem, p.mem + (long)b * beam * p.mem_row, beam * p.mem_row * 2);
  if (p.mem2) stage(l_mem2, p.mem2 + (long)b * beam * p.mem_row, beam * p.mem_row * 2);
  // ---- 1. top-`beam` of every source row (wave = source beam, + 4 for beams 4 .. 7)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int src = wave + 4 * u;
    if (src >= beam) continue;                       // (wave-uniform)
#pragma unroll
    for (int i = 0; i < CMAX; ++i) x[u][i] += base[u];            // -inf stays -inf
    for (int r = 0; r < beam; ++r) {
      float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < CMAX; ++i) if (x[u][i] > bv) { bv = x[u][i]; bi = src * V1 + lane + 64 * i; }      // ascending index: ties keep the lowest
      wave_argmax(bv, bi);
      if (lane == 0) { c_val[src * beam + r] = bv; c_idx[src * beam + r] = bi; }
      const int wc = bi - src * V1;                  // the winner's column: its owner retires it
      if ((wc & 63) == lane) {
#pragma unroll
        for (int i = 0; i < CMAX; ++i) if (i == (wc >> 6)) x[u][i] = -INFINITY;
      }
    }
  }
  __syncthreads();
  // ---- 2. merge the beam x beam survivors (wave 0)
  if (wave == 0) {
    float v = lane < beam * beam ? c_val[lane] : -INFINITY;
    int id = lane < beam * beam ? c_idx[lane] : 0x7fffffff;
    for (int r = 0; r < beam; ++r) {
      float bv = v; int bi = id;
      wave_argmax(bv, bi);
      // No comparable candidate left (every log-probability of the sample NaN or -inf: the forward guard only scans the trunk output): the
      // sentinel index would address LDS rows beyond the sample's beams below -- slot r then keeps ITS OWN hypothesis and emits token 0 ([PAD]),
      // a defined state instead of garbage.
      if (lane == 0) { win_v[r] = bv; win_i[r] = bi == 0x7fffffff ? r * V1 : bi; }
      if (id == bi) { v = -INFINITY; id = 0x7fffffff; }
    }
  }
  __syncthreads();
  // ---- 3. rewrite the sample's rows in the winners' order (a sample's hypotheses only permute among themselves)
  for (int i = tid; i < beam * p.max_len; i += 256) {
    const int j = i / p.max_len, c = i - j * p.max_len;
    const int src = win_i[j] / V1;
    seq[i] = c == pos ? (long long)(win_i[j] - src * V1) : l_seq[src * p.max_len + c];
  }
  if (p.anc)
    for (int i = tid; i < beam * p.anc_cols; i += 256) {
      const int j = i / p.anc_cols, c = i - j * p.anc_cols;
      p.anc[(long)b * beam * p.anc_cols + i] = l_anc[(win_i[j] / V1) * p.anc_cols + c];
    }
  auto permute_rows = [&](void* g, const uint32_t* l) {          // row j <- staged row of its source beam; rows that stay are not rewritten
    if ((mw & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(l)) & 15) == 0) {
      const int w4 = mw >> 2;
      for (int i = tid; i < beam * w4; i += 256) {
        const int j = i / w4, c = i - j * w4;
        const int src = win_i[j] / V1;
        if (src != j) reinterpret_cast<uint4*>(g)[i] = reinterpret_cast<const uint4*>(l)[src * w4 + c];
      }
    } else {
      for (int i = tid; i < beam * mw; i += 256) {
        const int j = i / mw, c = i - j * mw;
        const int src = win_i[j] / V1;
        if (src != j) reinterpret_cast<uint32_t*>(g)[i] = l[src * mw + c];
      }
    }
  };
  if (p.mem) permute_rows(p.mem + (long)b * beam * p.mem_row, l_mem);
  if (p.mem2) permute_rows(p.mem2 + (long)b * beam * p.mem_row, l_mem2);
  // ---- 4. finished beams (best p so far; the earlier / lower beam index wins ties), the -1000 penalty, next tokens
  if (tid == 0) {
    float pv = -INFINITY; int pi = -1;
    for (int j = 0; j < beam; ++j) {
      const int word = win_i[j] % V1;
      const bool end = p.force_end || word == p.eos;
      if (end && win_v[j] > pv) { pv = win_v[j]; pi = j; }
      p.beam_sum[b * beam + j] = win_v[j] - (end ? 1000.f : 0.f);
      p.words[b * beam + j] = word;
    }
    int take = -1;
    if (pi >= 0 && pv > p.best_p[b]) { p.best_p[b] = pv; take = pi; }
    s_best = take;
  }
  __syncthreads();
  if (s_best >= 0) {
    const int j = s_best, src = win_i[j] / V1;
    for (int c = tid; c < p.max_len; c += 256)
      p.best_seq[(long)b * p.max_len + c] = c == pos ? (long long)(win_i[j] % V1) : l_seq[src * p.max_len + c];
  }
  if (p.pos_out) {
    if (p.anc && pos + 1 < p.anc_cols && tid < beam) p.anc[(long)(b * beam + tid) * p.anc_cols + pos + 1] = b * beam + tid;
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      if (atomicAdd(p.ticket, 1) == (int)gridDim.x - 1) { *p.ticket = 0; p.pos_out[0] = pos + 1; }
    }
  }
}

}  // namespace

extern "C" {

int evk_beam_step(const float* logp, int32_t ld, int32_t V1, int32_t beam, int32_t B, int32_t max_len, const int64_t* pos, int32_t eos,
                  int32_t force_end, float* beam_sum, int64_t* beam_seq, float* best_p, int64_t* best_seq, int64_t* words, void* mem,
                  int32_t mem_row, int32_t* anc, int32_t anc_cols, int64_t* pos_advance, int32_t* ticket, void* mem2, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(logp && pos && beam_sum && beam_seq && best_p && best_seq && words && B > 0 && V1 > 0 && ld >= V1 && max_len > 0, "beam_step: bad args");
  EVK_REQUIRE(beam >= 1 && beam <= KMAX && beam <= V1, "beam_step: beam must be in [1, %d]", KMAX);
  EVK_REQUIRE((!mem || (mem_row > 0 && mem_row % 2 == 0)) && (!anc || anc_cols > 0), "beam_step: bad state geometry");
  EVK_REQUIRE(!pos_advance || ticket, "beam_step: advancing the position needs a (zero-initialised) ticket word");
  EVK_REQUIRE(!mem2 || mem, "beam_step: mem2 needs mem");
  static const int fast_on = evk_tunable("EVK_BEAM_STEP_FAST", 1);
  const bool fast = fast_on && V1 <= CMAX * 64;
  const size_t lds = (size_t)beam * max_len * 8 + (anc ? (size_t)beam * anc_cols * 4 : 0) + (mem ? (size_t)beam * mem_row * 2 : 0) +
                     (fast && mem2 ? (size_t)beam * mem_row * 2 : 0);
  EVK_REQUIRE(lds <= 60000, "beam_step: %zu bytes of per-sample state do not fit the staging buffer", lds);
  BeamP p{logp, ld, V1, beam, max_len, eos, force_end, reinterpret_cast<const long long*>(pos), beam_sum, reinterpret_cast<long long*>(beam_seq),
          best_p, reinterpret_cast<long long*>(best_seq), reinterpret_cast<long long*>(words), (bf16_t*)mem, mem_row, anc, anc_cols,
          (bf16_t*)mem2, reinterpret_cast<long long*>(pos_advance), ticket};
  ProfScope ps(EVK_FAM_NORM, s);
  if (fast) hipLaunchKernelGGL(beam_step_fast_kernel, dim3(B), dim3(256), lds, s, p);
  else hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(256), lds, s, p);
  return evk_check_launch("beam_step");
}

}  // extern "C"
