/* evoke_hip.h -- C ABI of libevoke_hip.so, the MI355X (gfx950) kernel library behind EVOKE's hot path.
 *
 * Boundary: the reference is pure Python/PyTorch and has no FFI of its own; what
 * calls into this library is the host-side mirror of the reference's model API
 * (evoke_amd/model_pretrain_finetune.py = models/model_pretrain_finetune_v0623_large_res.py:21-395).  Every
 * entry point below replaces the torch op(s) the reference issues at the cited file:line.
 *
 * Conventions
 *   - plain C, raw device pointers + sizes, no torch types; `stream` is a hipStream_t passed as void*.
 *   - all buffers are owned by the caller (PyTorch allocator); the library allocates nothing persistent.
 *   - every call only ENQUEUES work on `stream` (no hidden synchronisation, capturable in a hipGraph).
 *   - return 0 on success, a negative evk_status otherwise; evk_last_error() gives the message
 *     (thread-local).  Never aborts, never throws across the boundary.
 *   - activations / GEMM operands are bf16 (raw uint16 bits), statistics / losses / gradients of
 *     parameters are f32.  "bf16 in, f32 accumulate" on the CDNA4 matrix cores (v_mfma_f32_16x16x32_bf16).
 */
#ifndef EVOKE_HIP_H
#define EVOKE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* evk_stream_t;

enum evk_status { EVK_OK = 0, EVK_EINVAL = -1, EVK_ELAUNCH = -2, EVK_EUNSUPPORTED = -3 };
enum evk_dtype { EVK_F32 = 0, EVK_BF16 = 1 };
enum evk_act { EVK_ACT_NONE = 0, EVK_ACT_RELU = 1, EVK_ACT_GELU = 2, EVK_ACT_TANH = 3, EVK_ACT_SIGMOID = 4,
               EVK_ACT_GELU_NEW = 5 /* HF "gelu_new" (tanh approximation), GPT-2 */ };

/* operand addressing modes of the GEMM family (C[m][n] = sum_k A(m,k) * B(n,k)) */
enum evk_amode {
  EVK_A_PLAIN = 0,  /* A[m*lda + k]                       (K contiguous)                               */
  EVK_A_CONV = 1,   /* implicit im2col of an NHWC tensor  (conv forward; m = output pixel)             */
  EVK_A_DGRAD = 2,  /* gather from dY (NHWC) for conv data-gradient (m = input pixel)                  */
  EVK_A_KSTR = 3    /* A[k*lda + m]                       (K strided: "transposed" operand)            */
};
enum evk_bmode {
  EVK_B_PLAIN = 0,  /* B[n*ldb + k]                                                                     */
  EVK_B_KSTR = 1,   /* B[(k & kmask)*ldb + (k >> klog)*tapstride + n]   (plain [K][N] when klog = 30)   */
  EVK_B_WGATHER = 2 /* B(n,k) = X[pixel k shifted by tap][ci0 + n]      (conv weight-gradient)          */
};

typedef struct evk_conv_geom {
  int32_t N, Hi, Wi, Ci;         /* gathered tensor (input for fwd/wgrad)                               */
  int32_t Ho, Wo, Co;            /* output tensor                                                       */
  int32_t KH, KW, stride_h, stride_w, pad_h, pad_w;
  int64_t sN, sH, sW;            /* element strides of the gathered tensor (NHWC: Hi*Wi*Ci, Wi*Ci, Ci)  */
} evk_conv_geom;

typedef struct evk_gemm {
  const void* A; const void* B; void* C;
  const float* bias;             /* [N] or NULL                                                         */
  const void* resid;             /* added after activation, [M][ldr], or NULL                           */
  int32_t M, N, K;
  int32_t a_mode, b_mode;
  int64_t lda, ldb, ldc, ldr;
  int32_t batch_outer, batch_inner;  /* grid.z = outer*inner; offsets = zo*s?o + zi*s?i                 */
  int64_t sAo, sAi, sBo, sBi, sCo, sCi, sRo, sRi;
  float alpha;                   /* C = act(alpha * A.B + bias) + resid                                 */
  int32_t act;
  int32_t c_dtype, r_dtype;
  int32_t accumulate;            /* 1: C is f32 and receives += (atomics); allows split-K              */
  int32_t splitk;                /* <=0: chosen by the library                                          */
  int32_t b_klog; int64_t b_tapstride;   /* EVK_B_KSTR two-level K (see enum)                           */
  void* workspace; int64_t workspace_bytes; /* accumulate + split-K: partial slabs (evk_gemm_workspace_bytes);
                                            without it split-K falls back to f32 atomics                  */
  int64_t bias_stride_inner;     /* bias row used by inner batch index zi = bias + zi * bias_stride_inner (0: one shared bias)   */
  const void* relu_gate; int64_t ldg; /* optional bf16 [M][ldg]: C = relu_gate > 0 ? C : 0, applied after resid (the gradient of
                                    a ReLU whose forward output is relu_gate); batch 1, no accumulate                   */
  void* colstats;                /* optional f32 [ceil(M/64) (rounded to the tile)][2][N]: per 64-row block column sums and
                                    sums of squares of alpha*A.B, from the f32 accumulators (batch 1, no accumulate)  */
  void* gatestats;               /* optional f32 [ceil(M/64) (rounded to the tile)][2][N], needs relu_gate: per 64-row block column sums
                                    of the gated output g and of g * relu_gate -- the batch-norm backward sums of the layer whose
                                    ReLU output is relu_gate (evk_bn_bwd_sums_from_gate_partials)                          */
  evk_conv_geom g;               /* used by the gather modes                                            */
} evk_gemm;

int evk_version(void);
/* 16-bit storage format this build of the library computes in: 16 = IEEE fp16 (libevoke_hip.so, the default: meets the
 * 1e-3 loss parity, trains under the dynamic loss scale below), 0 = bf16 (libevoke_hip_bf16.so, built from the same sources
 * with -DEVK_STORE_BF16).  "bf16" in the comments of this header means "the library's 16-bit storage format".  The reference
 * has no counterpart: it computes in fp32 (torch CPU / CUDA default dtype). */
int evk_storage_format(void);
const char* evk_last_error(void);
/* Seed epoch of every dropout-drawing kernel (evk_dropout, evk_softmax_fwd/bwd, evk_attention_fwd/bwd, the relational-memory
 * attention): a device uint64 the caller owns and advances once per training step with a device-side op; each kernel adds
 * epoch * odd-constant to its seed argument, so a HIP graph of a whole step draws fresh masks at every replay although its
 * kernel arguments are frozen (nn.Dropout's RNG state in the reference: torch's global generator).  NULL (default) = off. */
int evk_set_seed_epoch(const uint64_t* epoch_dev);

/* ---- profiling hooks used by bench.py: HIP-event timing of every launch of a kernel family ---------- */
enum evk_family { EVK_FAM_GEMM = 0, EVK_FAM_NORM = 1, EVK_FAM_ELTWISE = 2, EVK_FAM_REDUCE = 3, EVK_FAM_OPTIM = 4,
                  EVK_FAM_COUNT = 5 };
int evk_prof_enable(int on);                     /* records a hipEvent pair around every launch when on  */
int evk_prof_collect(double* ms_per_family, int64_t* launches_per_family, double* flops_gemm); /* syncs+resets */
int evk_prof_dump_to(const char* path);          /* next evk_prof_collect also writes one CSV row per launch (shape-tagged GEMMs) */

/* ---- compute-unit partitioned streams (runtime.hip) ------------------------------------------------------------------------
 * The reference serves one batch at a time (modules/tester.py test loop: forward(mode='inference') per batch).  The serving loop here
 * (FineTune.generate_pipelined) overlaps the encoders of the next batch with the searches in flight; evk_stream_create_cu_mask makes
 * the encoder stream leave a share of the CUs to the searches' small dependent kernels.  mask: bit i = CU i may be used. */
int evk_stream_create_cu_mask(const uint32_t* mask, int32_t words, void** stream_out);
int evk_stream_destroy(void* stream);
int evk_device_cu_count(int32_t device, int32_t* cus);

/* ---- step replayer (replay.hip): the launch sequence of a stream-captured training step re-issued from C++ ------------
 * The reference's step is eager PyTorch (modules/trainer_v0401.py:426-435: zero_grad, forward, backward, clip, step).  Here the
 * step is captured ONCE per batch structure with hipStreamBeginCapture (through torch.cuda.graph), and the resulting hipGraph_t
 * is only a recording: evk_replay_build walks its nodes / edges / kernel parameters and evk_replay_run re-issues them with plain
 * hipLaunchKernel calls on a few streams (hipGraphLaunch itself costs 7-12 us of host time per node on ROCm 7.2).  The caller
 * keeps the hipGraph_t alive for the life of the plan (kernel argument blocks are owned by it).
 *   evk_replay_info: out6[0..6] = {nodes, kernels, memcpys, memsets, lanes (streams), cross-lane edges, lane streams replaced by the hardware-queue check (was: nodes replayed as isolated
 *   one-node graphs (copy flavours whose parameters the public query does not return faithfully)}.                        */
void* evk_replay_build(void* hip_graph, int32_t max_lanes);
/* Lanes that ARE the capture's streams: evk_capture_probe(1) before the capture begins makes every launch of the library note which stream
 * created which graph node; evk_replay_build_streams (origin = the stream the capture began on -> lane 0 = the caller's stream at run time)
 * gives every other capture stream a lane of its own; evk_capture_probe(0) afterwards.  Without notes it falls back to evk_replay_build's
 * minimum path cover.  (trainer_v0401.py:426-435 is the step; its eager PyTorch streams have no counterpart to mirror.) */
int evk_capture_probe(int32_t on);
int evk_replay_lane_priority(evk_stream_t captured, int32_t prio);   /* lane of that capture stream: HIP queue priority (-1 above default) */
void* evk_replay_build_streams(void* hip_graph, int32_t max_lanes, evk_stream_t origin);
int evk_replay_info(void* plan, int64_t* out6);
int evk_streams_concurrent(evk_stream_t a, evk_stream_t b, int32_t* concurrent);   /* do kernels on the two streams overlap (1) or share a hardware queue (0)? */
int evk_replay_run(void* plan, evk_stream_t stream);
int evk_replay_run_n(void* plan, evk_stream_t stream, int32_t n);   /* n replays back to back (the ~100 token steps of a beam search,
                                                                       modules/beam_search.py / att_model.py:139-192's loop, from one call) */
int evk_replay_destroy(void* plan);

/* ---- GEMM / implicit-GEMM family (MFMA) -------------------------------------------------------------
 * replaces: nn.Linear / torch.matmul (encoder_decoder.py:20-28,192-214; bert_model.py:262-341;
 * utils_v0511.py:263-278), nn.Conv1d k=1 (utils_v0511.py:135-147), nn.Conv2d of the ResNet-101 trunk
 * (visual_extractor.py:30-38 -> torchvision), and their autograd backward passes.                      */
int evk_gemm_launch(const evk_gemm* desc, evk_stream_t stream);
int64_t evk_gemm_workspace_bytes(const evk_gemm* desc);   /* 0 when no workspace is needed */
/* y[M][N] = act(LN(x)[M][512] . w[N][512]^T + bias) (+ resid, bf16 [M][ldr]): the LayerNorm of evk_layernorm_fwd (same modes, same
 * arithmetic, optional per-row conditional deltas dgam / dbet bf16 [M][ld_delta]) applied to the rows of x as they are loaded by the
 * GEMM, so `sublayer(x) = x + f(norm(x))` (encoder_decoder.py:114-126, 144-179) costs one launch per projection in the per-token decode
 * step instead of norm + projection.  K must be 512 (the decoder width of the released configuration). */
int evk_linear_ln(const void* x, const float* gamma, const float* beta, const void* dgam, const void* dbet, int64_t ld_delta,
                  float eps, int32_t mode, const void* w, const float* bias, const void* resid, int64_t ldr, void* y, int32_t y_dtype,
                  int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act, evk_stream_t stream);

/* NHWC bf16 convolution, weights KRSC bf16 ([Co][KH][KW][Ci]); y = conv(x, w) [+ nothing]: BN is separate.
 * fwd:   y[N,Ho,Wo,Co]      dgrad: dx[N,Hi,Wi,Ci]      wgrad: dw[Co,KH,KW,Ci] (f32, accumulated)       */
int evk_conv2d_fwd(const void* x, const void* w, void* y, const evk_conv_geom* g, evk_stream_t stream);
/* fwd + batch-norm statistics in the epilogue: part receives *nblk rows of [2][Co] partial sums (sum, sum of squares) of
 * the f32 conv result, to be finished by evk_bn_stats_from_partials (replaces a separate read of y for BN's batch stats) */
int64_t evk_conv_stats_bytes(int64_t M, int32_t C);
int evk_conv2d_fwd_stats(const void* x, const void* w, void* y, const evk_conv_geom* g, float* part, int64_t part_bytes,
                         int32_t* nblk, evk_stream_t stream);
/* the tile-GEMM path of evk_conv2d_fwd_stats, bypassing the routing to the weight-stationary kernel (tests / probes compare the two) */
int evk_conv2d_fwd_stats_tile(const void* x, const void* w, void* y, const evk_conv_geom* g, float* part, int64_t part_bytes,
                         int32_t* nblk, evk_stream_t stream);
int evk_conv2d_dgrad(const void* dy, const void* w, void* dx, const evk_conv_geom* g, evk_stream_t stream);
/* dx = dgrad(dy, w) + resid (bf16, shape of dx): the skip-connection gradient of a residual block joins in the epilogue */
int evk_conv2d_dgrad_add(const void* dy, const void* w, const void* resid, void* dx, const evk_conv_geom* g, evk_stream_t stream);
/* dx = relu'(gate) * (dgrad(dy, w) + resid): also applies the ReLU gate of the tensor dx belongs to (gate = that tensor's
 * post-ReLU forward value), so the batch-norm backward that consumes dx needs no mask pass                          */
/* Data gradient of a stride-1 "same" convolution (torchvision Bottleneck.conv2 of every non-strided block, as driven by
 * modules/visual_extractor.py:27-43) as a FORWARD convolution of dy with the weights flipped and transposed,
 * wt[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci] (16-bit, same element count as w): the K-contiguous main loop of evk_conv2d_fwd
 * instead of the gather + K-strided one (layer3: 87 instead of 123 us alone, 167 with the gate epilogue).  evk_conv_flip_weights
 * produces wt for a whole table of layers in one launch (w / wt: arrays of n_layers device pointers in HOST memory);
 * evk_conv2d_dgrad_flipped_gated_stats = evk_conv2d_dgrad_gated_stats with wt in place of w (g = the geometry of the forward conv). */
/* Data gradient of a 3x3 / stride 2 / pad 1 convolution (torchvision Bottleneck.conv2 of the first block of layers 2-4) by OUTPUT PARITY:
 * input pixel (2i + a, 2j + b) only receives the taps kh = a + 1, kw = b + 1 (mod 2) -- 2.25 taps per pixel instead of the 9 the
 * gathering GEMM multiplies.  evk_conv3x3s2_class_weights re-packs w [Co][3][3][Ci] into the four class slices (9 Ci Co elements, once per
 * step); evk_conv3x3s2_dgrad_parity runs the four small stride-1 convolutions over dy into ws (evk_conv3x3s2_dgrad_parity_ws_bytes) and
 * one pass that interleaves them into dx, applies the ReLU gate (required) and leaves *nblk partial rows [2][Ci] of (sum g, sum g * gate)
 * as evk_conv2d_dgrad_gated_stats does.  EVK_S2_PARITY=0 disables the route in evk_trunk_backward. */
int evk_conv3x3s2_dgrad_parity_supported(const evk_conv_geom* g);
int64_t evk_conv3x3s2_dgrad_parity_ws_bytes(const evk_conv_geom* g);
int evk_conv3x3s2_class_weights(const void* w, void* wc, int32_t Co, int32_t Ci, evk_stream_t stream);
int evk_conv3x3s2_dgrad_parity(const void* dy, const void* wc, const void* gate, void* dx, const evk_conv_geom* g, void* ws, int64_t ws_bytes,
                               float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream);
/* dst[N][2 Ho][2 Wo][C] = src[N][Ho][Wo][C] at the even pixels, zero elsewhere (16-bit, C % 8 == 0): the data gradient of a pointwise
 * stride-2 convolution (torchvision Bottleneck.downsample[0] of layers 2-4) from the COMPACT product dy . w -- three quarters of the
 * input pixels receive no gradient, so the product runs on a quarter of the rows and this pass writes the zeros */
int evk_upsample2_zero(const void* src, void* dst, int32_t N, int32_t Ho, int32_t Wo, int32_t C, evk_stream_t stream);
int evk_conv_flip_weights(const void* const* w, void* const* wt, const int32_t* Co, const int32_t* Ci, const int32_t* KH, const int32_t* KW,
                          int32_t n_layers, evk_stream_t stream);
int evk_conv2d_dgrad_flipped_gated_stats(const void* dy, const void* wt, const void* resid, const void* gate, void* dx, const evk_conv_geom* g,
                                         float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream);
/* Strip GEMM for the tall products of the contracting pointwise convolutions (gemm_strip.hip): C[M][N] = A[M][K] . B[N][K]^T, 16-bit
 * row-major operands, f32 accumulation -- torchvision Bottleneck.conv1 forward (4 planes -> planes; colstats = batch-norm partials as
 * evk_conv2d_fwd_stats) and the data gradient of Bottleneck.conv3 over the transposed weights of evk_conv_flip_weights (resid, ReLU
 * gate, gatestats as evk_conv2d_dgrad_gated_stats), both driven by modules/visual_extractor.py:30-38.  A workgroup owns 288 rows x
 * 128 columns (layer3: exactly one workgroup per CU), three LDS stages per operand, one barrier per 64-deep K step.  N % 128 == 0,
 * K % 64 == 0; evk_gemm_strip_routes is what evk_conv2d_fwd_stats / evk_conv2d_dgrad_flipped_gated_stats ask before taking it
 * (>= 200 workgroups, K >= 256; EVK_GEMM_STRIP=0 disables).  The statistics buffer holds evk_gemm_strip_part_bytes. */
int evk_gemm_strip_supported(int64_t M, int32_t N, int32_t K);
int64_t evk_gemm_strip_part_bytes(int64_t M, int32_t N);
int evk_gemm_strip(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int32_t N, int32_t K,
                   const void* resid, int64_t ldr, const void* gate, int64_t ldg, float* colstats, float* gatestats, int64_t part_bytes,
                   int32_t* nblk, evk_stream_t stream);
int evk_gemm_strip_routes(int64_t M, int32_t N, int32_t K, int64_t part_bytes, int32_t want_stats);
/* 3x3 / stride 1 / pad 1 convolution with the input halo tile resident in LDS (conv3x3.hip): torchvision Bottleneck.conv2 of every
 * non-strided block as driven by modules/visual_extractor.py:30-38 -- forward (x [N][H][W][C], w [Co][3][3][C], y [N][H][W][Co], 16-bit
 * NHWC; colstats: *nblk rows of [2][Co] partial (sum, sum of squares) of the f32 result for the batch norm that follows) and, over
 * the flipped / transposed weights of evk_conv_flip_weights, its data gradient (resid added, ReLU gate applied, gatestats: *nblk rows of
 * [2][Co] partial (sum g, sum g * gate)).  A workgroup owns whole image rows (<= 320 pixels) x 128 output channels and fills the halo
 * of a 64-channel chunk once for the nine taps: about half the LDS-fill bytes per flop of the implicit-GEMM tile path, which stays the
 * route for every other geometry (evk_conv3x3_halo_supported: C % 64 == 0, Co % 128 == 0, rows that tile; evk_conv3x3_halo_routes: what
 * evk_conv2d_fwd_stats / evk_conv2d_dgrad_flipped_gated_stats ask before taking it, EVK_CONV3X3_HALO=0 disables).  The statistics
 * buffer holds evk_conv3x3_halo_part_bytes. */
int evk_conv3x3_halo_supported(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Co);
int64_t evk_conv3x3_halo_part_bytes(int32_t N, int32_t H, int32_t W, int32_t Co);
int evk_conv3x3_halo(const void* x, const void* w, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Co,
                     const void* resid, int64_t ldr, const void* gate, int64_t ldg, float* colstats, float* gatestats,
                     int64_t part_bytes, int32_t* nblk, evk_stream_t stream);
/* Weight gradient of the same convolution (torchvision Bottleneck.conv2; the reference gets it from autograd over
 * modules/visual_extractor.py:30-38): dw[Co][3][3][Ci] (f32) += sum over pixels of dy[px][co] * x[px + tap][ci].  A workgroup owns a pixel
 * slice and a 64 x 64 (co, ci) block of the filter and accumulates all nine taps in registers from one dy tile and one x halo tile in LDS
 * (a quarter of the LDS-fill bytes per flop of the per-tap tile path); K-slices leave as f32 slabs in ws (evk_conv3x3_wgrad_halo_ws_bytes)
 * and are added into dw by the split-K reduction.  Ci, Co multiples of 64; evk_conv2d_wgrad routes here (EVK_CONV3X3_WGRAD_HALO=0 disables). */
int evk_conv3x3_wgrad_halo_supported(int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co);
int64_t evk_conv3x3_wgrad_halo_ws_bytes(int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co);
int evk_conv3x3_wgrad_halo(const void* dy, const void* x, float* dw, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                           void* ws, int64_t ws_bytes, evk_stream_t stream);
int evk_conv3x3_wgrad_halo_routes(const evk_conv_geom* g);
/* diagnostic: in-kernel cycle stamps of every later evk_conv3x3_halo launch into buf (8 x uint64 per workgroup), null = off */
int evk_conv3x3_halo_debug_stamps(void* buf);
int evk_conv3x3_halo_routes(const evk_conv_geom* g, int32_t C, int32_t Co, int64_t part_bytes, int32_t want_stats);
/* Weight-stationary kernel for the short-K pointwise convolutions (conv1x1.hip): y[M][N] = x[M][K] . w[N][K]^T with the weights
 * held in registers for the whole launch and only the pixel tiles streaming (torchvision Bottleneck conv3 forward: planes -> 4 planes;
 * the data gradient of conv1 is the same shape with the transposed weights; conv1 forward / conv3 data gradient at K = 512, 1024).
 * K in {64, 128, 256, 512, 1024}; N a multiple of 256 (128 at K >= 512): evk_conv1x1_ws_supported.  part (optional): *nblk rows of [2][N] partial sums, one per pixel tile -- forward: (sum y, sum
 * y^2) for the batch norm that follows (evk_bn_stats_finalize_from_partials); dgrad: (sum g, sum g*gate) of the gated output
 * (evk_bn_bwd_sums_from_gate_partials).  evk_conv2d_fwd_stats / evk_conv2d_dgrad_gated_stats route eligible problems here. */
int evk_conv1x1_ws_supported(int64_t M, int32_t K, int32_t N);
int64_t evk_conv1x1_ws_part_bytes(int64_t M, int32_t K, int32_t N);
int evk_conv1x1_ws_fwd(const void* x, const void* w, void* y, int64_t M, int32_t K, int32_t N, float* part, int64_t part_bytes, int32_t* nblk,
                       evk_stream_t stream);
int evk_conv1x1_ws_dgrad(const void* dy, const void* wt, const void* skip, const void* gate, void* dx, int64_t M, int32_t K, int32_t N,
                         float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream);
/* evk_conv1x1_ws_dgrad with a third partial row per pixel tile: sum over pixels of dx * (stat_x - stat_mean[channel]) -- with stat_x the
 * INPUT of the batch norm whose output gradient dx is (conv3's raw output for bn3 of the previous bottleneck; torchvision Bottleneck:
 * out = relu(bn3(conv3(.)) + identity)) this is sum g * (x - mean), the second column sum of nn.BatchNorm2d's backward, so the block-output
 * gradient is never re-read for it.  part then holds [nblk][3][N] floats (1.5 x evk_conv1x1_ws_part_bytes) */
int evk_conv1x1_ws_dgrad_xstat(const void* dy, const void* wt, const void* skip, const void* gate, void* dx, int64_t M, int32_t K, int32_t N,
                               const void* stat_x, const float* stat_mean, float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream);
/* evk_conv2d_dgrad_gated + the gate statistics of evk_gemm.gatestats: part receives *nblk rows of [2][Ci] (evk_conv_stats_bytes(rows of
 * dx, Ci) bytes) */
int evk_conv2d_dgrad_gated_stats(const void* dy, const void* w, const void* resid, const void* gate, void* dx, const evk_conv_geom* g,
                                 float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream);
/* evk_conv2d_dgrad_gated that also tries to leave the batch-norm backward sums of the layer dx is the output gradient of (stat_x = that
 * layer's conv output, stat_mean = its batch mean): when the problem is one the weight-stationary kernel takes, *nblk > 0 rows of
 * [3][Ci] partials (sum g, sum g*gate, sum g*(stat_x - mean)) are written for evk_bn_bwd_sums_from_xstat_partials; otherwise the plain
 * gated data gradient runs and *nblk = 0 (the caller reduces dx itself: evk_bn_bwd_reduce_acc) */
int64_t evk_conv_xstat_bytes(const evk_conv_geom* g);
int evk_conv2d_dgrad_gated_xstat(const void* dy, const void* w, const void* resid, const void* gate, void* dx, const evk_conv_geom* g,
                                 const void* stat_x, const float* stat_mean, float* part, int64_t part_bytes, int32_t* nblk, evk_stream_t stream);
int evk_conv2d_dgrad_gated(const void* dy, const void* w, const void* resid, const void* gate, void* dx, const evk_conv_geom* g,
                           evk_stream_t stream);
int evk_conv2d_wgrad(const void* dy, const void* x, float* dw, const evk_conv_geom* g, void* ws, int64_t ws_bytes, evk_stream_t stream);
int64_t evk_conv2d_wgrad_ws_bytes(const evk_conv_geom* g);

/* ResNet stem (conv 7x7 s2 p3, 3->64): images f32 NCHW -> zero-padded NHWC4 bf16 staging buffer
 * [N][H+6][W+8][4]; the conv then runs as an implicit GEMM with K = 7 x (8 taps x 4 ch) = 224.        */
int evk_stem_pack_image(const float* img_nchw, void* xpad, int32_t N, int32_t H, int32_t W, evk_stream_t stream);
int evk_stem_pack_weight(const float* w_oihw, void* w_packed, evk_stream_t stream);        /* [64][7][8][4] bf16 */
int evk_stem_unpack_wgrad(const float* dw_packed, float* dw_oihw, evk_stream_t stream);    /* += into OIHW grad  */
int evk_stem_fwd(const void* xpad, const void* w_packed, void* y, int32_t N, int32_t H, int32_t W, evk_stream_t stream);
int evk_stem_fwd_stats(const void* xpad, const void* w_packed, void* y, int32_t N, int32_t H, int32_t W, float* part, int64_t part_bytes,
                       int32_t* nblk, evk_stream_t stream);
/* The stem with the input halo of a 4 x 64 output tile resident in LDS and all weight fragments in registers (stem.hip): same operands
 * and results as evk_stem_fwd_stats, which routes here when H / 2 is a multiple of 4 and W / 2 of 64 (EVK_STEM_HALO=0 disables); the
 * statistics buffer holds evk_stem_halo_part_bytes (one partial row per wave of the persistent grid) */
int evk_stem_halo_supported(int32_t N, int32_t H, int32_t W);
int64_t evk_stem_halo_part_bytes(int32_t N, int32_t H, int32_t W);
int evk_stem_halo_fwd(const void* xpad, const void* w_packed, void* y, int32_t N, int32_t H, int32_t W, float* part, int64_t part_bytes,
                      int32_t* nblk, evk_stream_t stream);
int64_t evk_stem_halo_wgrad_ws_bytes(int32_t N, int32_t H, int32_t W);
int evk_stem_halo_wgrad(const void* dy, const void* xpad, float* dw_packed, int32_t N, int32_t H, int32_t W, void* ws, int64_t ws_bytes,
                        evk_stream_t stream);
int evk_stem_wgrad(const void* dy, const void* xpad, float* dw_packed, int32_t N, int32_t H, int32_t W, void* ws, int64_t ws_bytes, evk_stream_t stream);
int64_t evk_stem_wgrad_ws_bytes(int32_t N, int32_t H, int32_t W);

/* ---- input pipeline: image pre-processing on the GPU (preproc.hip) --------------------------------------------
 * replaces, per image: the torchvision transform stack applied inside collate_fn (modules/dataloaders_v0623.py:22-37,
 * 89-92; modules/dataloaders_v0401.py:25-37): Resize (Pillow antialiased bilinear, two 8-bit passes) -> RandomCrop /
 * CenterCrop -> [RandomHorizontalFlip] -> [RandomRotation, nearest, fill 0] -> ToTensor -> Normalize.  The random
 * parameters are drawn by the caller; `affine` holds Pillow's 16.16 fixed-point inverse map (evoke_amd/pipeline.py).  */
typedef struct evk_preproc_desc {
  const void* src;               /* uint8 pixels [src_h][src_w][channels] in device memory                        */
  int32_t src_h, src_w, channels;/* channels: 3 (RGB) or 1 (grey, replicated like PIL .convert('RGB'))            */
  int32_t resize_h, resize_w;    /* size after transforms.Resize                                                  */
  int32_t crop_top, crop_left, out_size;   /* square crop window inside the resized image                         */
  int32_t flip;                  /* horizontal flip of the crop                                                   */
  int32_t rotate;                /* 0: no rotation; 1: apply `affine` (a0..a5 of Pillow's affine_fixed)           */
  int32_t affine[6];
  float mean[3], std[3];
} evk_preproc_desc;
int64_t evk_preprocess_ws_bytes(const evk_preproc_desc* d);          /* -1 on a bad descriptor */
int evk_preprocess_image(const evk_preproc_desc* d, void* ws, int64_t ws_bytes, float* out, evk_stream_t stream); /* out f32 [3][S][S] */

/* ---- native runner of the ResNet bottleneck trunk (trunk.hip) -------------------------------------------------
 * replaces: the Python module walk over torchvision resnet101 children 0-7 (modules/visual_extractor.py:27-43,
 * patch_feats = self.model(images)) and autograd's backward over it: one call per direction issues every conv /
 * batch-norm / pooling kernel of the trunk from C++.  `layers` lists the conv+BN pairs in torchvision state_dict order:
 * [0] conv1+bn1 (the 7x7 stem), then per bottleneck conv1+bn1, conv2+bn2, conv3+bn3 and, for the first block of each
 * layer, downsample.0+downsample.1.                                                                               */
typedef struct evk_trunk_cfg {
  int32_t blocks[4];             /* bottlenecks per layer (resnet101: 3, 4, 23, 3)                                */
  int32_t planes[4];             /* bottleneck width per layer (64, 128, 256, 512); outputs are 4 x planes        */
  int32_t stride[4];             /* stride of the first block of each layer (1, 2, 2, 2)                          */
  float eps, momentum;           /* batch-norm epsilon and running-statistics momentum                           */
} evk_trunk_cfg;
typedef struct evk_trunk_layer {
  const void* w;                 /* bf16 KRSC conv weight; [0]: the f32 OIHW master weight of the 7x7 stem        */
  float* dw;                     /* f32 gradient to accumulate into (layout of the master weight), NULL = frozen  */
  const float* gamma; const float* beta; float* running_mean; float* running_var;
  float* dgamma; float* dbeta;   /* f32 gradients to accumulate into, or NULL                                     */
} evk_trunk_layer;
int evk_trunk_num_pairs(const evk_trunk_cfg* cfg);
int64_t evk_trunk_ws_bytes(const evk_trunk_cfg* cfg, int32_t N, int32_t H, int32_t W);   /* -1 on bad arguments */
/* images f32 NCHW [N][3][H][W] -> out bf16 NHWC [N][H/32][W/32][4*planes[3]]; ws keeps what the backward needs */
int evk_trunk_forward(const evk_trunk_cfg* cfg, const evk_trunk_layer* layers, int32_t n_layers, const float* images, int32_t N,
                      int32_t H, int32_t W, void* ws, int64_t ws_bytes, void* out, int32_t training, evk_stream_t stream);
/* Inference forward of the trunk (nn.Module.eval() + torch.no_grad() around modules/visual_extractor.py:30-43 -- the test / sample /
 * validation loops of modules/trainer_v0401.py:470-494, 592-658): the eval-mode batch norms as per-channel scale / shift applied to the
 * convolutions' f32 accumulators, identity / ReLU in the same epilogue.  fold_ws = evk_trunk_fold_bytes(cfg) bytes kept by the caller between
 * calls (the scale / shift vectors); refold bit 0 recomputes them (affine parameters or running statistics changed), bit 1 keeps the
 * convolutions that add an identity unfused (conv + evk_bn_apply with the cached vectors), bit 2 keeps every layer unfused (the eval forward
 * of evk_trunk_forward bit for bit, without its bn_finalize launches and on evk_trunk_infer_ws_bytes of workspace). */
int64_t evk_trunk_fold_bytes(const evk_trunk_cfg* cfg);
int64_t evk_trunk_infer_ws_bytes(const evk_trunk_cfg* cfg, int32_t N, int32_t H, int32_t W);   /* `ws` of the inference forward: two arenas the blocks
                                                                                                   alternate between, no per-layer activations */
int evk_trunk_forward_inference(const evk_trunk_cfg* cfg, const evk_trunk_layer* layers, int32_t n_layers, const float* images, int32_t N,
                                int32_t H, int32_t W, void* ws, int64_t ws_bytes, void* out, void* fold_ws, int64_t fold_bytes, int32_t refold,
                                evk_stream_t stream);
/* pieces of it, callable directly (nn.BatchNorm2d in eval mode behind nn.Conv2d, modules/visual_extractor.py:30-43): the coefficients, and a
 * convolution with the scale / bias + optional residual + optional ReLU epilogue on the routes that have it (weight-stationary / strip / halo
 * kernels: evk_conv2d_fwd_affine_routes = 1; the tile GEMM's geometries return EVK_EUNSUPPORTED). */
int evk_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, float* scale, float* shift,
                       int32_t C, evk_stream_t stream);
int evk_conv2d_fwd_affine(const void* x, const void* w, void* y, const evk_conv_geom* g, const float* scale, const float* bias, const void* resid,
                          int32_t relu, evk_stream_t stream);
int evk_conv2d_fwd_affine_routes(const evk_conv_geom* g);
int evk_conv1x1_ws_fwd_affine(const void* x, const void* w, void* y, int64_t M, int32_t K, int32_t N, const float* scale, const float* bias,
                              const void* resid, int32_t relu, evk_stream_t stream);
int evk_conv3x3_halo_affine(const void* x, const void* w, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Co, const float* scale,
                            const float* bias, const void* resid, int64_t ldr, int32_t relu, evk_stream_t stream);
int evk_gemm_strip_affine(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int32_t N, int32_t K,
                          const float* scale, const float* bias, const void* resid, int64_t ldr, int32_t relu, evk_stream_t stream);

/* dout bf16 (shape of out); parameter gradients are accumulated in place; weight-gradient GEMMs run on wgrad_stream
 * (ordered after their inputs by events; pass NULL or `stream` to keep everything on one stream)                 */
int evk_trunk_backward(const evk_trunk_cfg* cfg, const evk_trunk_layer* layers, int32_t n_layers, int32_t N, int32_t H, int32_t W,
                       void* ws, int64_t ws_bytes, const void* dout, int32_t training, evk_stream_t stream, evk_stream_t wgrad_stream);

/* ---- row-wise kernels (norm.hip): one wavefront per row, shuffle reductions ----------------------------
 * LayerNorm family.  mode 0 = torch.nn.LayerNorm (biased variance, eps inside the sqrt; bert_model.py:355,433,
 * ...v0623_large_res.py:34-35); mode 1 = R2Gen LayerNorm / ConditionalLayerNorm (encoder_decoder.py:93-103,
 * 166-179: unbiased std, eps added to the std).  dgam/dbet: optional per-row deltas [rows][D] added to gamma/beta.
 * mean/rstd [rows] f32 are saved for the backward.  D % 8 == 0, D <= 2048.                                   */
int evk_layernorm_fwd(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta,
                      const void* dgam, const void* dbet, int d_dtype, float* mean, float* rstd,
                      int64_t rows, int32_t D, int32_t mode, float eps, evk_stream_t stream);
/* dgamma/dbeta (f32 [D]) are accumulated (+=); ddgam/ddbet (per-row, dtype d_dtype) are written.            */
int evk_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const void* dgam, int d_dtype,
                      const float* mean, const float* rstd, void* dx, int dx_dtype, float* dgamma, float* dbeta,
                      void* ddgam, void* ddbet, int64_t rows, int32_t D, int32_t mode, float eps, evk_stream_t stream);
/* softmax over scores[batch][heads][Tq][ld_in] f32 (S valid columns) -> probabilities (p_dtype) [..][ld_out] (pad = 0).
 * mask: uint8, 1 = attend, element (b, q, col) at mask[b*mask_batch_stride + q*mask_q_stride + col]; masked or
 * (causal && col > q) entries get probability 0 (== the reference's -1e9 fill / finfo.min add whenever a row keeps
 * at least one key: encoder_decoder.py:24, bert_model.py:322-325).  p_drop > 0 writes the dropped copy to pdrop_out.*/
int evk_softmax_fwd(const float* scores, void* p_out, void* pdrop_out, int p_dtype, const unsigned char* mask, int64_t mask_batch_stride,
                    int32_t mask_q_stride, int32_t causal, int64_t batch, int32_t heads, int32_t Tq, int32_t S,
                    int32_t ld_in, int32_t ld_out, float p_drop, uint64_t seed, evk_stream_t stream);
/* dS = alpha * P * (dP - sum_j dP_j P_j), dP first passed through the same dropout mask                      */
int evk_softmax_bwd(const void* dp, int dp_dtype, int32_t ld_dp, const void* probs, void* ds, int p_dtype, int64_t rows, int32_t S, int32_t ld,
                    float alpha, float p_drop, uint64_t seed, evk_stream_t stream);
/* log_softmax over logits[rows][ld] (V valid); optional outputs: logp [rows][ld_out], lse [rows]; with target:
 * acc2[0] += sum(-logp[target]*wmask), acc2[1] += sum(wmask)   (encoder_decoder.py:393 + loss.py:9-16)       */
/* single-query attention of the incremental decode step (MultiHeadedAttention with one new token per hypothesis,
 * encoder_decoder.py:182-214 as driven by caption_model.py beam_search): out[r] = concat_h softmax(scale q_h.K_h^T [masked]) V_h;
 * q/out bf16 [R][heads*64], k/v bf16 [R/kv_div][S][heads*64], mask uint8 [R/kv_div][S] (1 = attend) or NULL; kv_div consecutive
 * query rows (the beams of one sample) share one K/V row -- 1 for the self-attention caches; head_dim 64, S <= 256            */
int evk_decode_attention(const void* q, const void* k, const void* v, const unsigned char* mask, void* out, int32_t R, int32_t S,
                         int32_t heads, int32_t head_dim, int32_t kv_div, float scale, evk_stream_t stream);
/* the same over self-attention caches that beam search does NOT re-order: rowmap int32 [R][S] names, for hypothesis r and
 * position s, the cache row that holds that position (the row of the ancestor that wrote it); k/v bf16 [R][S][heads*64],
 * mask uint8 [R][S] or NULL; last_pos (device int64 scalar, or NULL): only positions <= *last_pos have been written --
 * later ones are neither read nor attended (the causal mask of the incremental step).  Replaces the per-step state gather
 * of caption_model.py:beam_step (`new_state[_][:, beam_ix]`, caption_model.py:74-86) on the K/V caches by a gather of the
 * [R][S] index table.                                                                                                */
int evk_decode_attention_indirect(const void* q, const void* k, const void* v, const unsigned char* mask, const int32_t* rowmap,
                                  const int64_t* last_pos, void* out, int32_t R, int32_t S, int32_t heads, int32_t head_dim, float scale,
                                  evk_stream_t stream);
/* evk_decode_attention_indirect fed by the fused q | k | v projection of the step (qkv bf16 [R][ldq], ldq >= 3 * heads * 64): q is read in
 * place, this step's key / value are attended to as position *last_pos straight from qkv and appended to row r of the caches for the
 * later steps -- the `self.ks[:, t] = k; self.vs[:, t] = v` state update of the incremental MultiHeadedAttention
 * (encoder_decoder.py:182-214 driven by caption_model.py:beam_search) without a copy kernel.  rowmap[r][*last_pos] must be r. */
int evk_decode_attention_qkv(const void* qkv, int64_t ldq, void* k_cache, void* v_cache, const int32_t* rowmap, const int64_t* last_pos, void* out,
                             int32_t R, int32_t S, int32_t heads, int32_t head_dim, float scale, evk_stream_t stream);
int evk_log_softmax_nll_fwd(const float* logits, float* logp, float* lse, const int64_t* target, const float* wmask, float* acc2,
                            int64_t rows, int32_t V, int32_t ld, int32_t ld_out, evk_stream_t stream);
/* the masked-NLL forward with one value per row (row_nll[r] = -logp[r][target[r]] * wmask[r]) instead of the two atomically accumulated
 * sums: the caller adds the rows up (a tree reduction: the same bits on every run), lse as above.  loss.py:9-22 */
int evk_log_softmax_nll_rows(const float* logits, float* lse, const int64_t* target, const float* wmask, float* row_nll, int64_t rows,
                             int32_t V, int32_t ld, evk_stream_t stream);
int evk_nll_bwd(const float* logits, const float* lse, const int64_t* target, const float* wmask, const float* gscale,
                void* dlogits, int64_t rows, int32_t V, int32_t ld, int32_t ld_out, evk_stream_t stream);
/* beam step (caption_model.py:70-74): k <= 8 largest of each row, descending, ties -> lowest index first */
int evk_topk_rows(const float* x, float* vals, int64_t* idx, int64_t rows, int32_t n, int32_t k, evk_stream_t stream);
/* Row-block kernels of the per-token decode step (modules/encoder_decoder.py:118-131 DecoderLayer.forward -> :134-141 ConditionalSublayerConnection,
 * :206-214 PositionwiseFeedForward, as driven by modules/caption_model.py:beam_search through EncoderDecoder.core :396-404): for 16 hypotheses
 * per workgroup, in ONE launch,
 *     v = x + g . W2^T + b2,  g = a  (output projection of an attention)  or  g = relu(a . W1^T + b1)  (the whole feed-forward);
 *     y = v rounded to the 16-bit storage format (the residual stream);
 *     n = (gamma + dgam[r]) * (v - mean) / (std_unbiased + eps) + (beta + dbet[r])   -- the NEXT sub-layer's (conditional) layer norm
 *         (:93-103 / :144-179; gamma = null: no norm, n unused).
 * a, x, y, n: [R][512] 16-bit (y may alias x); w*_packed: evk_decode_rb_pack of a row-major [512 out][512 in] 16-bit matrix (fragment-major
 * [n tile][k step][lane][8]: the kernel streams it from L2 into MFMA operand registers); b1, b2, gamma, beta f32[512]; dgam / dbet
 * [R][ld_delta] 16-bit per-hypothesis deltas or null. */
int evk_decode_rb_pack(const void* w, void* packed, evk_stream_t stream);
/* sync_ws: null, or evk_decode_rowblock_sync_bytes(R) bytes of device memory, ZERO-INITIALISED ONCE and then owned by the calling stream's
 * launch sequence (never shared by launches that may overlap): with it the projection variant splits every row block over four workgroups
 * (a quarter of the weight stream each) that exchange their partial row sums for the norm through it. */
int64_t evk_decode_rowblock_sync_bytes(int32_t R);
int evk_decode_rowblock(const void* a, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2, const void* x, void* y,
                        const float* gamma, const float* beta, const void* dgam, const void* dbet, int64_t ld_delta, float eps, void* n, int32_t R,
                        void* sync_ws, evk_stream_t stream);

/* One launch for the whole beam bookkeeping of a generated token (beam.hip): CaptionModel.beam_step + the finished-beam
 * handling of CaptionModel.beam_search (modules/caption_model.py:51-106, 174-189; att_model.py:133-135) for hypothesis counts
 * that no longer change (every step after the first).  logp [B*beam][ld] f32 log-probs (V1 = V+1 valid columns); *pos = position
 * being written.  In place: beam_sum [B][beam], beam_seq [B][beam][max_len], best_p [B], best_seq [B][max_len], and the
 * per-hypothesis state rows that follow their hypothesis -- mem [B*beam][mem_row] (16-bit relational memory) and anc
 * [B*beam][anc_cols] (self-attention cache row table), either may be NULL.  words [B*beam] receives the next input tokens.
 * force_end != 0 at the last position (every live beam is closed).  Ties -> lowest flat index.
 * pos_advance (optional, may be `pos` itself) with ticket (one zero-initialised int32 in device memory, left at zero): after the
 * bookkeeping the kernel seeds column pos + 1 of the row table with each hypothesis's own row and the LAST workgroup to finish writes
 * *pos_advance = pos + 1 -- the loop counter of the captured decode step advances without a launch of its own. */
int evk_beam_step(const float* logp, int32_t ld, int32_t V1, int32_t beam, int32_t B, int32_t max_len, const int64_t* pos, int32_t eos,
                  int32_t force_end, float* beam_sum, int64_t* beam_seq, float* best_p, int64_t* best_seq, int64_t* words, void* mem,
                  int32_t mem_row, int32_t* anc, int32_t anc_cols, int64_t* pos_advance, int32_t* ticket, void* mem2, evk_stream_t stream);
/* mem2 (optional): a second per-hypothesis state of mem_row 16-bit values re-ordered like mem (tanh of the relational memory) */
/* F.normalize(p=2, eps=1e-12) rows, f32 */
int evk_l2norm_fwd(const float* x, float* y, float* nrm, int64_t rows, int32_t D, evk_stream_t stream);
int evk_l2norm_bwd(const float* dy, const float* y, const float* nrm, float* dx, int64_t rows, int32_t D, evk_stream_t stream);
/* soft-label cross entropy (F.cross_entropy with probability targets; ...v0623_large_res.py:271-281,316-328,343-349):
 * loss_acc[0] += scale * sum_rows(lse*sum(t) - sum(t*z)); dz = scale*gscale[0]*(softmax*sum(t) - t);
 * diag_mask: z[r][r] treated as -1e9 (fill_diagonal_, line 276)                                                 */
int evk_softce(const float* z, const float* t, float* loss_acc, float* dz, const float* gscale, int64_t rows, int32_t Cn,
               float scale, int32_t diag_mask, evk_stream_t stream);

/* ---- BatchNorm / pooling (bn.hip): x[M][C] bf16 channels-last, C a power-of-two multiple of 8 <= 2048 --------
 * train-mode torch.nn.BatchNorm2d/1d (torchvision resnet101 via visual_extractor.py:30-38; utils_v0511.py:137,177-181) */
/* column reductions are two-stage (per-block partials in `ws`, then a final sum): deterministic, no atomics */
int64_t evk_colreduce_ws_bytes(int32_t C);
int evk_bn_stats(const void* x, float* sum, float* sumsq, void* ws, int64_t ws_bytes, int64_t M, int32_t C, evk_stream_t stream);
int evk_bn_stats_from_partials(const float* part, int32_t nblk, float* sum, float* sumsq, int32_t C, evk_stream_t stream);
int evk_bn_finalize(const float* sum, const float* sumsq, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float* scale, float* shift, float* mean, float* invstd, int32_t C, float count,
                    float momentum, float eps, int32_t training, evk_stream_t stream);
/* (`part` is SCRATCH for the three *_from_partials entry points: with >= 1024 partial rows they first fold rows r, r + h, r + 2h, ...
 * into row r in place with a fully parallel launch, h = nblk / 16 -- layer1's 9216 rows of 64-256 columns kept 8-32 workgroups in a
 * 46-58 us dependent row loop otherwise -- so its contents are undefined afterwards.)
 * Batch-norm backward sums without a pass over the gradient: `part` = nblk rows of [2][C] partial sums (sum g, sum g*z) written by
 * the epilogue of the data-gradient GEMM that produced g gated by z > 0 (evk_conv2d_dgrad_gated_stats), z = relu(gamma*xhat + beta)
 * the layer's own forward output.  Where the gate is open xhat = (z - beta) / gamma, so sum_g = sum g and sum_gx = (sum g*z - beta *
 * sum g) / gamma (0 for gamma == 0, where dx vanishes anyway); dbeta_acc += sum_g, dgamma_acc += sum_gx when given.
 * That identity divides the rounding of the 16-bit stored z by gamma: channels with |beta| > 16 |gamma| (2 |gamma| in the bf16 build)
 * -- the near-dead channels of an ImageNet-pretrained resnet101 -- are recomputed exactly inside the same launch from dz (the gated
 * gradient the producing GEMM wrote, [M][C]) and y (the layer's raw convolution output), xhat = (y - mean) * invstd; pass dz = NULL to
 * switch the fallback off.
 * Replaces evk_bn_bwd_reduce_acc for bn1 / bn2 of every bottleneck (nn.BatchNorm2d backward, torchvision resnet101). */
int evk_bn_bwd_sums_from_gate_partials(const float* part, int32_t nblk, const float* gamma, const float* beta, float* sum_g, float* sum_gx,
                                       float* dbeta_acc, float* dgamma_acc, int32_t C, const void* dz, const void* y, const float* mean,
                                       const float* invstd, int64_t M, evk_stream_t stream);
/* the same second stage for the [3][C] partial rows of evk_conv2d_dgrad_gated_xstat: sum_g = sum of row 0, sum_gx = invstd * sum of row 2
 * (row 2 already carries x - mean) */
int evk_bn_bwd_sums_from_xstat_partials(const float* part, int32_t nblk, const float* invstd, float* sum_g, float* sum_gx,
                                        float* dbeta_acc, float* dgamma_acc, int32_t C, evk_stream_t stream);
/* evk_bn_stats_from_partials + evk_bn_finalize (training) in one launch: the trunk runner's per-convolution critical path */
int evk_bn_stats_finalize_from_partials(const float* part, int32_t nblk, float* sum, float* sumsq, const float* gamma, const float* beta,
                                        float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* invstd,
                                        int32_t C, float count, float momentum, float eps, evk_stream_t stream);
int evk_bn_apply(const void* x, const float* scale, const float* shift, const void* resid, void* y, int64_t M, int32_t C,
                 int32_t relu, evk_stream_t stream);
int evk_bn_bwd_reduce(const void* dz, const void* z, const void* x, const float* mean, const float* invstd, float* sum_g,
                      float* sum_gx, void* ws, int64_t ws_bytes, int64_t M, int32_t C, int32_t relu, evk_stream_t stream);
int evk_bn_bwd_reduce_acc(const void* dz, const void* z, const void* x, const float* mean, const float* invstd, float* sum_g,
                          float* sum_gx, float* dbeta_acc, float* dgamma_acc, void* ws, int64_t ws_bytes, int64_t M, int32_t C,
                          int32_t relu, evk_stream_t stream);   /* + accumulates the affine gradients in place */
int evk_bn_bwd_apply(const void* dz, const void* z, const void* x, const float* scale, const float* mean, const float* invstd,
                     const float* sum_g, const float* sum_gx, void* dx, void* dres, int64_t M, int32_t C, int32_t relu,
                     evk_stream_t stream);
int evk_maxpool3x3s2_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, evk_stream_t stream);
int evk_maxpool3x3s2_bwd(const void* x, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, evk_stream_t stream);
/* same pooling with the window tap (kh*3+kw, one byte per output element) of the first maximum recorded, so the backward needs
 * neither the input tensor nor a recomputation of the 3x3 windows                                                 */
int evk_maxpool3x3s2_fwd_idx(const void* x, void* y, void* idx, int32_t N, int32_t H, int32_t W, int32_t C, evk_stream_t stream);
int evk_maxpool3x3s2_bwd_idx(const void* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, evk_stream_t stream);
/* visual_extractor.py:40-42: avg_feats = mean over patches */
int evk_patch_mean_fwd(const void* att, void* fc, int32_t N, int32_t P, int32_t C, evk_stream_t stream);
int evk_patch_mean_bwd(const void* datt_in, const void* dfc, void* datt, int32_t N, int32_t P, int32_t C, evk_stream_t stream);

/* ---- elementwise / gather / optimizer (eltwise.hip) ------------------------------------------------------- */
int evk_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, evk_stream_t stream);
int evk_act_fwd(const void* x, void* y, int64_t n, int32_t act, evk_stream_t stream);
/* ref = output (relu/tanh/sigmoid) or pre-activation (gelu) */
int evk_act_bwd(const void* dy, const void* ref, void* dx, int64_t n, int32_t act, evk_stream_t stream);
/* y = x * keep / (1-p) (+ resid), keep from a stateless hash of (seed, index): the backward is the same call on dy
 * (HF BertSelfOutput / BertOutput: dense -> dropout -> + residual, bert_model.py:359-362,437-440)               */
int evk_dropout(const void* x, const void* resid, void* y, int64_t n, float p, uint64_t seed, evk_stream_t stream);
int evk_embedding_fwd(const float* table, const int64_t* ids, const float* pos, const float* extra, void* out, int out_dtype,
                      int64_t rows, int32_t D, int32_t L, float scale, int64_t table_rows, const int64_t* pos0_dev, float* out_f32_copy, evk_stream_t stream);
/* pos0_dev (optional): device scalar added to the position index, pos[(r % L) + *pos0_dev] -- the decode step embeds one token per
 * hypothesis at a position only the device knows (modules/encoder_decoder.py:226-243 PositionalEncoding at the current step);
 * out_f32_copy (optional): the same rows unrounded, for the f32 relational memory of the decode step */
/* ids outside [0, table_rows) are skipped in both directions (forward: the table term is 0): never a wild access */
int evk_embedding_bwd(const void* dout, int d_dtype, const int64_t* ids, float* dtable, int64_t rows, int32_t D, float scale,
                      int64_t padding_idx, int64_t table_rows, evk_stream_t stream);
int evk_colsum(const void* x, float* out, int64_t M, int32_t N, int64_t ld, evk_stream_t stream);
/* fused clip_grad_value_ + optimizer step + bf16 shadow refresh over a flat buffer (trainer_v0401.py:262,434;
 * optimizers.py:17-53).  kind 0 = torch.optim.RAdam, kind 1 = torch.optim.Adam (vmax != NULL -> amsgrad)      */
int evk_optim_step(float* p, const float* g, float* m, float* v, float* vmax, void* shadow, int64_t n, int32_t kind, float lr,
                   float beta1, float beta2, float eps, float weight_decay, float clip, int64_t step, evk_stream_t stream);
/* same with the gradients multiplied by grad_scale first (= 1 / loss scale: the fp16-storage build back-propagates a loss
 * scaled by a constant so that its 16-bit gradients stay in fp16's normal range; torch.cuda.amp.GradScaler.unscale_ is
 * the closest public counterpart, the reference itself trains in fp32 and has none).  Elements whose scaled gradient is
 * not finite are skipped for the step. */
int evk_optim_step_scaled(float* p, const float* g, float* m, float* v, float* vmax, void* shadow, int64_t n, int32_t kind, float lr,
                          float beta1, float beta2, float eps, float weight_decay, float clip, int64_t step, float grad_scale,
                          evk_stream_t stream);

/* ---- fused attention core (attn.hip): softmax(alpha * Q.K^T [masked]) [dropout] . V per head, forward and backward ----------
 * replaces `attention` (modules/encoder_decoder.py:20-28), BertSelfAttention's matmul/softmax/dropout/matmul
 * (models/language_encoder/bert_model.py:262-341), ScaledDotProductAttention's core (modules/utils_v0511.py:263-278) and HF
 * GPT2Attention._attn of the distilgpt2 backend.  q [B][T][heads*dh], k / v [B][S][heads*dh] (heads interleaved in the feature
 * dimension), out [B][T][heads*dh], all in the library's 16-bit storage format; dh a multiple of 64, S <= 640
 * (evk_attention_supported).  The f32 score tile of 32 query rows lives in LDS only.
 *   fwd: probs [B][heads][T][pad8(S)] receives the pre-dropout probabilities (the backward's input), probs_dropped the
 *        post-dropout ones (required iff p_drop > 0; dropout = stateless hash of (seed + epoch, element index)); mask uint8, 1 =
 *        attend: key mask (mask_q_stride 0, mask_batch_stride = S) or per-query mask (mask_q_stride = S, batch stride T*S).
 *   bwd: dq = dS.K with dS = P * (dropout'(dO.V^T) - rowsum) * scale; ds [B][heads][T][pad8(S)] (16-bit) is written for the
 *        caller's dK = dS^T.Q product, dV = P'^T.dO uses probs_dropped -- both are K-strided batched evk_gemm launches.     */
int evk_attention_supported(int32_t S, int32_t dh);
int evk_attention_fwd(const void* q, const void* k, const void* v, void* out, void* probs, void* probs_dropped, const unsigned char* mask,
                      int64_t mask_batch_stride, int32_t mask_q_stride, int32_t causal, int64_t B, int32_t heads, int32_t T, int32_t S,
                      int32_t dh, float scale, float p_drop, uint64_t seed, evk_stream_t stream);
int evk_attention_bwd(const void* dout, const void* k, const void* v, const void* probs, void* ds, void* dq, int64_t B, int32_t heads,
                      int32_t T, int32_t S, int32_t dh, float scale, float p_drop, uint64_t seed, evk_stream_t stream);

/* Dynamic loss scaling with a GLOBAL overflow skip, entirely on the device (no host read-back, HIP-graph capturable).  The
 * reference trains in fp32 (modules/trainer_v0401.py:256-263, 426-435) and has none; torch.cuda.amp.GradScaler is the public
 * counterpart (growth 2, backoff 0.5).  scale_state = float[4] in device memory: [0] loss scale, [1] consecutive good steps,
 * [2] "a non-finite gradient was seen this step", [3] total skipped steps.
 *   evk_grad_nonfinite     sets state[2] when any element of the flat f32 gradient buffer is inf / NaN (run it AFTER the
 *                          gradient all-reduce: a sum carries every rank's inf / NaN, so all ranks take the same decision);
 *   evk_optim_step_dyn     evk_optim_step with (a) the step count read from *step_dev (+1), (b) gradients multiplied by
 *                          inv_world / state[0] (inv_world = 1 / ranks: the all-reduce sums, the loss is NOT pre-divided, so
 *                          16-bit activation gradients keep the full loss scale on every rank), (c) no update at all when
 *                          state[2] is set; zero_grad != 0: the kernel also writes 0 over every gradient it read (taken or
 *                          skipped step), which replaces the optimizer.zero_grad() pass over the buffer
 *                          (trainer_v0401.py:428).  scale_state may be NULL (bf16 storage: scale 1, never skips);
 *   evk_optim_bump         step_dev[0..count) += 1 unless state[2] is set (per-parameter step counts live on the device so
 *                          that a skipped step does not advance them, as GradScaler.step skips optimizer.step);
 *   evk_loss_scale_update  after the optimizer: overflow -> scale *= backoff (>= min_scale), good steps = 0, skipped += 1;
 *                          else good steps += 1 and scale *= growth (<= max_scale) every `interval` good steps; clears [2]. */
int evk_optim_step_dyn(float* p, float* g, float* m, float* v, float* vmax, void* shadow, int64_t n, int32_t kind, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float clip, const int32_t* step_dev,
                       const float* scale_state, float inv_world, int32_t zero_grad, evk_stream_t stream);
int evk_optim_bump(int32_t* step_dev, int32_t count, const float* scale_state, evk_stream_t stream);
/* One optimizer step of a whole parameter group (modules/optimizers.py:27-46 groups, flat buffers) in two launches, with EVERYTHING
 * that can change between steps read from device memory, so that a step captured in a HIP graph never bakes it in:
 *   hp_dev      float[8]: lr, beta1, beta2, eps, weight_decay, clip value (torch lr schedulers / load_state_dict rewrite the buffer);
 *   offsets_dev int64[n_params]: first element of each parameter in the flat buffers (multiples of 8), ascending;
 *   steps_dev   int32[n_params]: torch.optim's state['step'] PER PARAMETER -- bias corrections and RAdam's rectification are evaluated
 *               per parameter, a parameter that sat out some steps (statically unused branch of that step kind) keeps its own count;
 *   touched_dev uint8[n_params]: 1 = the parameter received a gradient this step (others are left alone: no decay, no count);
 *   coef_dev    float[4 n_params] scratch.   Overflow verdict / loss scale / world as evk_optim_step_dyn. */
int evk_optim_group_step(float* p, float* g, float* m, float* v, float* vmax, void* shadow, int64_t n, int32_t kind, const float* hp_dev,
                         const int64_t* offsets_dev, int32_t n_params, int32_t* steps_dev, const unsigned char* touched_dev, float* coef_dev,
                         const float* scale_state, float inv_world, int32_t zero_grad, evk_stream_t stream);
int evk_grad_nonfinite(const float* g, int64_t n, float* scale_state, evk_stream_t stream);
int evk_loss_scale_update(float* scale_state, float growth, float backoff, int32_t interval, float min_scale, float max_scale,
                          evk_stream_t stream);
/* Forward-overflow guard (no counterpart: the reference's forward is fp32, modules/visual_extractor.py:37-43 cannot overflow): *count +=
 * the number of wave-level hits of inf / NaN in a 16-bit activation tensor (the trunk output).  fp16 storage has a range cliff at
 * 65504 that the loss scale cannot repair in the FORWARD -- a network whose batch-norm running statistics are untrained, run in eval
 * mode, reaches it; the host raises and names EVK_STORE=bf16 (evoke_amd/ops.py: guard_finite). */
int evk_act_nonfinite(const void* x, int64_t n, int32_t* count, evk_stream_t stream);
/* Gradients of a model whose parameters NO fused optimizer owns -- the reference's own step, modules/trainer_v0401.py:432-435
 * `loss.backward(); clip_grad_value_(model.parameters(), 0.1); optimizer.step()` with a torch.optim optimizer (modules/optimizers.py:
 * 27-46): the gradients must leave backward() UNSCALED.  chunk_table = device array of {float* g; int32 n; int32 pad} (one entry per
 * <= 65536-element piece of a parameter's f32 gradient tensor, any number of tensors).  mode 0: state[2] = 1 when any element is
 * inf / NaN;  mode 1: g *= 1 / state[0], or g = 0 everywhere when state[2] is set (an overflowed step surfaces as zero gradients);
 * mode 2: g *= state[0] (gradients left over from an earlier backward are brought back under the scale before accumulation). */
int evk_grads_multi(const void* chunk_table, int32_t n_chunks, int32_t mode, float* scale_state, evk_stream_t stream);

/* ---- relational memory runner (rm.hip): RelationalMemory.forward / forward_step, encoder_decoder.py:274-300 ------
 * The host loop over tokens lives in the library (5 GEMM launches + 2 fused kernels per token forward, BPTT backward,
 * one K = L*B*slots weight-gradient GEMM per parameter).  Fixed geometry: 3 slots, d = 512, 8 heads.
 *   xk, xv (B,L,512) = x_t.Wk^T + bk / x_t.Wv^T + bv,  gw (B,L,1024) = W(x_t): hoisted out of the recurrence by the caller.
 *   Wqkv [1536][512] = [attn.linears.0; .1; .2] bf16, biases f32.  out (B,L,1536) bf16; m_last (B,3,512) optional.   */
int64_t evk_rm_ws_bytes(int32_t B, int32_t L);
/* evk_rm_set_persistent(1) (or EVK_RM_PERSIST=1): walks of L >= 8 tokens run as ONE persistent kernel per direction -- one
 * workgroup per sample, the sample's memory in LDS for all L tokens, the 4 MB of weights streamed from L2 into MFMA operand
 * registers once per token, no inter-workgroup synchronisation (cannot dead-lock).  2000 launches per training step become 2,
 * but each CU then pulls all weights through its own L1 (~27 GB/s per CU): 15 + 12.5 ms against 5.5 + 5.8 ms for the default
 * per-token launch sequence, which therefore stays the default; shorter walks (decode, L = 1) always use it. */
int evk_rm_set_persistent(int32_t on);
int evk_rm_forward(const void* xk, const void* xv, const void* gw, const void* m0, const void* Wqkv, const float* bqkv, const void* Wo,
                   const float* bo, const void* W0, const float* b0, const void* W2, const float* b2, const void* U, const float* bU,
                   void* out, void* m_last, void* ws, int64_t ws_bytes, int32_t B, int32_t L, float p_drop, uint64_t seed,
                   evk_stream_t stream);
/* ONE generated token of the relational memory for B hypotheses (RelationalMemory.forward_step, modules/encoder_decoder.py:274-291, as
 * CaptionModel.beam_search drives it through EncoderDecoder.core, :396-404): 8 launches instead of the 13 of evk_rm_forward(L = 1) +
 * its three hoisted projections, and the state updated IN PLACE.  x [B][512] token embeddings; Wx [2048][512] / bx = attn.linears.1,
 * attn.linears.2 and W stacked (the three projections of x_t as one product); mem, tmem [B][3][512] = the memory and tanh(memory), in /
 * out (the caller re-orders both with the hypotheses: evk_beam_step mem / mem2); out [B][1536] = the new memory row for the decoder's
 * conditional layer norms.  Same kernels and arithmetic as evk_rm_forward: bit-identical results. */
int64_t evk_rm_decode_ws_bytes(int32_t B);
int evk_rm_decode_step(const void* x, const void* Wx, const float* bx, void* mem, void* tmem, const void* Wqkv, const float* bqkv, const void* Wo,
                       const float* bo, const void* W0, const float* b0, const void* W2, const float* b2, const void* U, const float* bU, void* out,
                       void* ws, int64_t ws_bytes, int32_t B, evk_stream_t stream);
/* evk_rm_decode_step in f32 (rm_f32.hip): the relational memory is a recurrence over every generated position and, on the fixtures' weights,
 * an expanding one (1e-6 relative on its weights -> 3e-3 on the log-probabilities of position 99; 16-bit operands -> 0.3-0.5), so everything
 * that feeds back into it stays f32: x [B][512] f32 token embeddings, mem [B][3][512] f32 in / out, the f32 MASTER weights Wx [2048][512] (=
 * attn.linears.1 / .2 / W stacked), Wqkv [1536][512], Wo, W0, W2 [512][512], U [1024][512] and biases; products on the f32-input MFMA
 * (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain).  out16 [B][1536] = the new memory in the 16-bit storage format for the decoder's
 * conditional layer norms (that path does not feed back).  8 launches, like the 16-bit step. */
int64_t evk_rm_decode_f32_ws_bytes(int32_t B);
/* Arithmetic of the f32 relational-memory products (evk_rm_decode_step_f32, evk_rm_forward_f32; RelationalMemory.forward_step,
 * modules/encoder_decoder.py:274-291): mode 0 (default) uses v_mfma_f32_16x16x4_f32; mode 1 (fp16-storage build only, opt-in) splits every
 * f32 operand element into hi + lo fp16 halves and accumulates hi.hi + hi.lo + lo.hi in f32 on the 16-bit MFMA -- operands to 2^-22, +1 %
 * throughput, 9 x the rounding of mode 0 in a long recurrence --; mode < 0 queries.  Returns the mode in force.  Env EVK_RM_SPLIT16 = initial value. */
int evk_rm_f32_split16(int32_t mode);
int evk_rm_decode_step_f32(const float* x, const float* Wx, const float* bx, float* mem, const float* Wqkv, const float* bqkv, const float* Wo,
                           const float* bo, const float* W0, const float* b0, const float* W2, const float* b2, const float* U, const float* bU,
                           void* out16, void* ws, int64_t ws_bytes, int32_t B, evk_stream_t stream);
/* dxk, dxv, dgw are written; the f32 parameter gradients are accumulated (+=); W*t = transposed bf16 weights */
int evk_rm_backward(const void* dout, const void* xk, const void* xv, const void* Wqkvt, const void* Wot, const void* W0t, const void* W2t,
                    const void* Ut, void* dxk, void* dxv, void* dgw, float* dWqkv, float* dbqkv, float* dWo, float* dbo, float* dW0,
                    float* db0, float* dW2, float* db2, float* dU, float* dbU, void* ws, int64_t ws_bytes, int32_t B, int32_t L,
                    float p_drop, uint64_t seed, evk_stream_t stream);
/* evk_rm_forward with the recurrence carried in f32 (same reference lines; the training / teacher-forced default): x32 (B, L, 512) f32 token
 * embeddings, Wx32 [2048][512] = [attn.linears.1; attn.linears.2; W] and the six f32 MASTER matrices; out / m_last / ws as evk_rm_forward (the
 * 16-bit BPTT of evk_rm_backward reads the same workspace), ws32 = evk_rm_f32_ws_bytes(B, L) bytes of f32 scratch. */
int64_t evk_rm_f32_ws_bytes(int32_t B, int32_t L);
int evk_rm_forward_f32(const float* x32, const float* Wx32, const float* bx, const void* m0, const float* Wqkv32, const float* bqkv, const float* Wo32,
                       const float* bo, const float* W032, const float* b0, const float* W232, const float* b2, const float* U32, const float* bU,
                       void* out, void* m_last, void* ws, int64_t ws_bytes, void* ws32, int64_t ws32_bytes, int32_t B, int32_t L, float p_drop,
                       uint64_t seed, evk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
