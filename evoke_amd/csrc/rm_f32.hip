// rm_f32.hip -- the relational memory of the DECODE step in f32 (RelationalMemory.forward_step, modules/encoder_decoder.py:274-291, as
// CaptionModel.beam_search drives it: one token per hypothesis per call).
//
// Why f32: the memory is a recurrence over every generated position, and on the weights the parity fixtures use it is an EXPANDING one -- a
// relative perturbation of 1e-6 of its weights moves the log-probabilities of position 99 by 3e-3, 16-bit operands by 0.3-0.5
// (tests/golden/make_beam_trace.py::drift16) -- so a 16-bit recurrence cannot follow the reference's beam search past the first few
// near-ties.  Everything that feeds back into the memory therefore stays in f32 here: the token embedding, the memory itself, the seven
// weight matrices (the f32 masters, no 16-bit shadows), every intermediate.  Only the copy of the new memory that the decoder's conditional
// layer norms read is rounded to the 16-bit storage format (that path does not feed back).
//
// The six products are tiny (768 x 512 x 512 for 256 hypotheses) and latency bound, so the f32-input MFMA (v_mfma_f32_16x16x4_f32, 1/16 of
// the 16-bit rate, bit-for-bit a k-ordered fmaf chain: cdna_hip_programming.md section 3) costs nothing against the 16-bit kernel: a
// workgroup owns a 32 x 32 output tile, requests its whole 32 x 512 A and B panels in ONE round trip (128 KB of LDS), and each of its four
// waves walks the 128 k-steps of one 16 x 16 tile on two alternating accumulators.
#include "common.h"

namespace {

constexpr int S3 = 3, KEYS4 = 4, HEADS8 = 8, DH64 = 64, D512 = 512;
constexpr int TM = 32, TN = 32, KP = 512, LDK = KP + 4;      // + 4 floats: rows 16 bytes apart in the banks

struct GemmF { const float* A; const float* W; const float* bias; const float* resid; float* C; int M, N; long lda, ldc, ldr; int act, a_tanh; bf16_t* C16; long ldc16; };

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* As = sm;                       // [TM][LDK]
  float* Bs = sm + TM * LDK;            // [TN][LDK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  // the two panels: 2 x 32 rows x 128 float4 = 8192 float4, 32 per thread, all in flight at once
  float4 ra[16], rb[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = tid + 256 * i, row = c >> 7, ch = c & 127;
    const int m = m0 + row;
    ra[i] = m < p.M ? *reinterpret_cast<const float4*>(p.A + (long)m * p.lda + ch * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    rb[i] = *reinterpret_cast<const float4*>(p.W + (long)(n0 + row) * KP + ch * 4);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = tid + 256 * i, row = c >> 7, ch = c & 127;
    float4 a = ra[i];
    if (p.a_tanh) a = make_float4(tanhf(a.x), tanhf(a.y), tanhf(a.z), tanhf(a.w));
    *reinterpret_cast<float4*>(As + row * LDK + ch * 4) = a;
    *reinterpret_cast<float4*>(Bs + row * LDK + ch * 4) = rb[i];
  }
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  const float* ap = As + (wm * 16 + (lane & 15)) * LDK + (lane >> 4);
  const float* bp = Bs + (wn * 16 + (lane & 15)) * LDK + (lane >> 4);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int k = 0; k < KP; k += 8) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k], bp[k], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k + 4], bp[k + 4], acc1, 0, 0, 0);
  }
  const int col = n0 + wn * 16 + (lane & 15);
  const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 16 + 4 * (lane >> 4) + j;
    if (m >= p.M) continue;
    float v = (acc0[j] + acc1[j]) + bias;
    if (p.act == EVK_ACT_RELU) v = fmaxf(v, 0.f);
    if (p.resid) v += p.resid[(long)m * p.ldr + col];
    p.C[(long)m * p.ldc + col] = v;
    if (p.C16) p.C16[(long)m * p.ldc16 + col] = f2bf(v);          // what the 16-bit backward of the training recurrence reads
  }
}

int gemm_f32(const float* A, long lda, const float* W, const float* bias, const float* resid, long ldr, float* C, long ldc, int M, int N, int act,
             int a_tanh, hipStream_t s, bf16_t* C16 = nullptr, long ldc16 = 0) {
  EVK_REQUIRE(N % TN == 0 && M > 0, "rm f32 gemm: N must be a multiple of 32");
  static bool attr_done = false;
  constexpr int LDS = 2 * TM * LDK * 4;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  GemmF p{A, W, bias, resid, C, M, N, lda, ldc, ldr, act, a_tanh, C16, ldc16};
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(N / TN, (M + TM - 1) / TM), dim3(256), LDS, s, p);
  return evk_check_launch("rm f32 gemm");
}

__device__ __forceinline__ float half_sum32(float v) {      // sum over the 32 lanes that share a head
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one workgroup per hypothesis; thread -> head h = tid >> 5, dims 2l, 2l + 1: softmax(q k^T / 8) v over the 3 memory slots + the token
__global__ __launch_bounds__(256) void rm_attn_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ xp, float* __restrict__ a) {
  const int b = blockIdx.x, h = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int c = h * DH64 + 2 * l;
  float q[S3][2], k[KEYS4][2], v[KEYS4][2];
#pragma unroll
  for (int i = 0; i < S3; ++i) {
    const float* r = qkv + (long)(b * S3 + i) * 1536 + c;
    q[i][0] = r[0]; q[i][1] = r[1]; k[i][0] = r[512]; k[i][1] = r[513]; v[i][0] = r[1024]; v[i][1] = r[1025];
  }
  const float* x = xp + (long)b * 2048 + c;
  k[3][0] = x[0]; k[3][1] = x[1]; v[3][0] = x[512]; v[3][1] = x[513];
#pragma unroll
  for (int i = 0; i < S3; ++i) {
    float s[KEYS4], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) { s[j] = half_sum32(q[i][0] * k[j][0] + q[i][1] * k[j][1]) * 0.125f; mx = fmaxf(mx, s[j]); }
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
    float o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) { const float w = s[j] / sum; o0 += w * v[j][0]; o1 += w * v[j][1]; }
    float* o = a + (long)(b * S3 + i) * D512 + c;
    o[0] = o0; o[1] = o1;
  }
}

// next = sigmoid(ig) * tanh(nm1 + h2) + sigmoid(fg) * m, gates = W x (xp columns 1024..2047, shared by the slots) + U tanh(m); in place
__global__ __launch_bounds__(256) void rm_gate_f32_kernel(const float* __restrict__ xp, const float* __restrict__ gu, const float* __restrict__ nm1,
                                                          const float* __restrict__ h2, float* __restrict__ m, bf16_t* __restrict__ out16, int B) {
  const long total = (long)B * S3 * D512;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % D512);
    const long bs = i / D512;
    const long b = bs / S3;
    const float ig = xp[b * 2048 + 1024 + c] + gu[bs * 1024 + c];
    const float fg = xp[b * 2048 + 1536 + c] + gu[bs * 1024 + 512 + c];
    const float si = 1.f / (1.f + expf(-ig)), sf = 1.f / (1.f + expf(-fg));
    const float nx = si * tanhf(nm1[i] + h2[i]) + sf * m[i];
    m[i] = nx;
    out16[i] = f2bf(nx);
  }
}

// ---- the TRAINING recurrence in f32 (RelationalMemory.forward, modules/encoder_decoder.py:293-300): same arithmetic as above plus what
// the 16-bit BPTT of rm.hip reads back -- attention probabilities, dropout (the stateless hash of rm.hip: forward and backward draw the
// same mask), 16-bit copies of every saved intermediate.
__device__ __forceinline__ uint32_t hash32f(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}
__device__ __forceinline__ float keep_scale_f(uint64_t seed, uint64_t idx, float p) {          // == rm.hip keep_scale
  if (p <= 0.f) return 1.f;
  return (hash32f(seed * 0x9E3779B97F4A7C15ULL + idx) >> 8) * (1.f / 16777216.f) >= p ? 1.f / (1.f - p) : 0.f;
}

struct AttT { const float* qkv; const float* xp; long x_bstride; float* a; bf16_t* a16; float* P; float p_drop; unsigned long long seed; const unsigned long long* epoch; };
__global__ __launch_bounds__(256) void rm_attn_f32_train_kernel(const AttT p) {
  const int b = blockIdx.x, h = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int c = h * DH64 + 2 * l;
  float q[S3][2], k[KEYS4][2], v[KEYS4][2];
#pragma unroll
  for (int i = 0; i < S3; ++i) {
    const float* r = p.qkv + (long)(b * S3 + i) * 1536 + c;
    q[i][0] = r[0]; q[i][1] = r[1]; k[i][0] = r[512]; k[i][1] = r[513]; v[i][0] = r[1024]; v[i][1] = r[1025];
  }
  const float* x = p.xp + (long)b * p.x_bstride + c;
  k[3][0] = x[0]; k[3][1] = x[1]; v[3][0] = x[512]; v[3][1] = x[513];
  const unsigned long long seed = evk_mix_seed(p.seed, p.epoch);
#pragma unroll
  for (int i = 0; i < S3; ++i) {
    float s[KEYS4], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) { s[j] = half_sum32(q[i][0] * k[j][0] + q[i][1] * k[j][1]) * 0.125f; mx = fmaxf(mx, s[j]); }
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
    float o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int j = 0; j < KEYS4; ++j) {
      const float pr = s[j] / sum;
      if (l == 0 && p.P) p.P[((long)(b * HEADS8 + h) * S3 + i) * KEYS4 + j] = pr;
      const float w = pr * keep_scale_f(seed, ((uint64_t)(b * HEADS8 + h) * S3 + i) * KEYS4 + j, p.p_drop);
      o0 += w * v[j][0]; o1 += w * v[j][1];
    }
    const long o = (long)(b * S3 + i) * D512 + c;
    p.a[o] = o0; p.a[o + 1] = o1;
    *reinterpret_cast<uint32_t*>(p.a16 + o) = pack2bf(o0, o1);
  }
}

struct GateT {
  const float* xp; long x_bstride; const float* gu; const float* nm1; const float* h2; float* m;
  bf16_t* m16_next; bf16_t* tm16_next; bf16_t* out; long out_bstride; bf16_t* si; bf16_t* sf; bf16_t* tnm; int B;
};
__global__ __launch_bounds__(256) void rm_gate_f32_train_kernel(const GateT p) {
  const long total = (long)p.B * S3 * D512;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % D512);
    const long bs = i / D512;
    const long b = bs / S3;
    const int sl = (int)(bs - b * S3);
    const float ig = p.xp[b * p.x_bstride + 1024 + c] + p.gu[bs * 1024 + c];
    const float fg = p.xp[b * p.x_bstride + 1536 + c] + p.gu[bs * 1024 + 512 + c];
    const float si = 1.f / (1.f + expf(-ig)), sf = 1.f / (1.f + expf(-fg));
    const float t = tanhf(p.nm1[i] + p.h2[i]);
    const float nx = si * t + sf * p.m[i];
    p.m[i] = nx;                                        // the carried state stays in f32
    const bf16_t n16 = f2bf(nx);
    p.m16_next[i] = n16;
    p.tm16_next[i] = f2bf(tanhf(nx));
    p.out[b * p.out_bstride + sl * D512 + c] = n16;
    p.si[i] = f2bf(si); p.sf[i] = f2bf(sf); p.tnm[i] = f2bf(t);
  }
}

}  // namespace

// launchers used by rm.hip's evk_rm_forward_f32 (which owns the workspace layout the 16-bit backward reads)
int rmf32_gemm(const float* A, long lda, const float* W, const float* bias, const float* resid, long ldr, float* C, long ldc, int M, int N, int act,
               int a_tanh, bf16_t* C16, long ldc16, hipStream_t s) {
  return gemm_f32(A, lda, W, bias, resid, ldr, C, ldc, M, N, act, a_tanh, s, C16, ldc16);
}
int rmf32_attn_train(const float* qkv, const float* xp, long x_bstride, float* a, bf16_t* a16, float* P, float p_drop, unsigned long long seed, int B,
                     hipStream_t s) {
  AttT p{qkv, xp, x_bstride, a, a16, P, p_drop, seed, evk_seed_epoch_ptr()};
  hipLaunchKernelGGL(rm_attn_f32_train_kernel, dim3(B), dim3(256), 0, s, p);
  return evk_check_launch("rm f32 attention (train)");
}
int rmf32_gate_train(const float* xp, long x_bstride, const float* gu, const float* nm1, const float* h2, float* m, bf16_t* m16_next, bf16_t* tm16_next,
                     bf16_t* out, long out_bstride, bf16_t* si, bf16_t* sf, bf16_t* tnm, int B, hipStream_t s) {
  GateT p{xp, x_bstride, gu, nm1, h2, m, m16_next, tm16_next, out, out_bstride, si, sf, tnm, B};
  const long total = (long)B * S3 * D512;
  hipLaunchKernelGGL(rm_gate_f32_train_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
  return evk_check_launch("rm f32 gate (train)");
}

extern "C" {

int64_t evk_rm_decode_f32_ws_bytes(int32_t B) {
  const long R = (long)B * S3;
  return ((long)B * 2048 + R * 1536 + 4 * R * D512 + R * 1024) * 4 + 4096;
}

int evk_rm_decode_step_f32(const float* x, const float* Wx, const float* bx, float* mem, const float* Wqkv, const float* bqkv, const float* Wo,
                           const float* bo, const float* W0, const float* b0, const float* W2, const float* b2, const float* U, const float* bU,
                           void* out16, void* ws, int64_t ws_bytes, int32_t B, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(x && Wx && bx && mem && Wqkv && bqkv && Wo && bo && W0 && b0 && W2 && b2 && U && bU && out16 && ws && B > 0, "rm_decode_step_f32: null/empty argument");
  EVK_REQUIRE(ws_bytes >= evk_rm_decode_f32_ws_bytes(B), "rm_decode_step_f32: workspace too small");
  EVK_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(mem) | reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(Wx) |
                reinterpret_cast<uintptr_t>(Wqkv) | reinterpret_cast<uintptr_t>(Wo) | reinterpret_cast<uintptr_t>(W0) | reinterpret_cast<uintptr_t>(W2) |
                reinterpret_cast<uintptr_t>(U)) & 15) == 0, "rm_decode_step_f32: 16-byte aligned buffers");
  const int R = B * S3;
  float* p = reinterpret_cast<float*>(ws);
  auto take = [&](long n) { float* r = p; p += (n + 63) / 64 * 64; return r; };
  float* xp = take((long)B * 2048);
  float* qkv = take((long)R * 1536);
  float* a = take((long)R * D512); float* nm1 = take((long)R * D512); float* h1 = take((long)R * D512); float* h2 = take((long)R * D512);
  float* gu = take((long)R * 1024);
  ProfScope ps(EVK_FAM_GEMM, s);
  if (int e = gemm_f32(x, D512, Wx, bx, nullptr, 0, xp, 2048, B, 2048, EVK_ACT_NONE, 0, s)) return e;            // keys | values | gates of the token
  if (int e = gemm_f32(mem, D512, Wqkv, bqkv, nullptr, 0, qkv, 1536, R, 1536, EVK_ACT_NONE, 0, s)) return e;
  hipLaunchKernelGGL(rm_attn_f32_kernel, dim3(B), dim3(256), 0, s, qkv, xp, a);
  if (int e = gemm_f32(a, D512, Wo, bo, mem, D512, nm1, D512, R, D512, EVK_ACT_NONE, 0, s)) return e;
  if (int e = gemm_f32(nm1, D512, W0, b0, nullptr, 0, h1, D512, R, D512, EVK_ACT_RELU, 0, s)) return e;
  if (int e = gemm_f32(h1, D512, W2, b2, nullptr, 0, h2, D512, R, D512, EVK_ACT_RELU, 0, s)) return e;
  if (int e = gemm_f32(mem, D512, U, bU, nullptr, 0, gu, 1024, R, 1024, EVK_ACT_NONE, 1, s)) return e;           // U tanh(m)
  const long total = (long)R * D512;
  hipLaunchKernelGGL(rm_gate_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xp, gu, nm1, h2, mem, (bf16_t*)out16, B);
  return evk_check_launch("rm_decode_step_f32");
}

}  // extern "C"
