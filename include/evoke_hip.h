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
enum evk_act { EVK_ACT_NONE = 0, EVK_ACT_RELU = 1, EVK_ACT_GELU = 2, EVK_ACT_TANH = 3, EVK_ACT_SIGMOID = 4 };

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
  evk_conv_geom g;               /* used by the gather modes                                            */
} evk_gemm;

int evk_version(void);
const char* evk_last_error(void);

/* ---- profiling hooks used by bench.py: HIP-event timing of every launch of a kernel family ---------- */
enum evk_family { EVK_FAM_GEMM = 0, EVK_FAM_NORM = 1, EVK_FAM_ELTWISE = 2, EVK_FAM_REDUCE = 3, EVK_FAM_OPTIM = 4,
                  EVK_FAM_COUNT = 5 };
int evk_prof_enable(int on);                     /* records a hipEvent pair around every launch when on  */
int evk_prof_collect(double* ms_per_family, int64_t* launches_per_family, double* flops_gemm); /* syncs+resets */

/* ---- GEMM / implicit-GEMM family (MFMA) -------------------------------------------------------------
 * replaces: nn.Linear / torch.matmul (encoder_decoder.py:20-28,192-214; bert_model.py:262-341;
 * utils_v0511.py:263-278), nn.Conv1d k=1 (utils_v0511.py:135-147), nn.Conv2d of the ResNet-101 trunk
 * (visual_extractor.py:30-38 -> torchvision), and their autograd backward passes.                      */
int evk_gemm_launch(const evk_gemm* desc, evk_stream_t stream);

/* NHWC bf16 convolution, weights KRSC bf16 ([Co][KH][KW][Ci]); y = conv(x, w) [+ nothing]: BN is separate.
 * fwd:   y[N,Ho,Wo,Co]      dgrad: dx[N,Hi,Wi,Ci]      wgrad: dw[Co,KH,KW,Ci] (f32, accumulated)       */
int evk_conv2d_fwd(const void* x, const void* w, void* y, const evk_conv_geom* g, evk_stream_t stream);
int evk_conv2d_dgrad(const void* dy, const void* w, void* dx, const evk_conv_geom* g, evk_stream_t stream);
int evk_conv2d_wgrad(const void* dy, const void* x, float* dw, const evk_conv_geom* g, evk_stream_t stream);

/* ResNet stem (conv 7x7 s2 p3, 3->64): images f32 NCHW -> zero-padded NHWC4 bf16 staging buffer
 * [N][H+6][W+8][4]; the conv then runs as an implicit GEMM with K = 7 x (8 taps x 4 ch) = 224.        */
int evk_stem_pack_image(const float* img_nchw, void* xpad, int32_t N, int32_t H, int32_t W, evk_stream_t stream);
int evk_stem_pack_weight(const float* w_oihw, void* w_packed, evk_stream_t stream);        /* [64][7][8][4] bf16 */
int evk_stem_unpack_wgrad(const float* dw_packed, float* dw_oihw, evk_stream_t stream);    /* += into OIHW grad  */
int evk_stem_fwd(const void* xpad, const void* w_packed, void* y, int32_t N, int32_t H, int32_t W, evk_stream_t stream);
int evk_stem_wgrad(const void* dy, const void* xpad, float* dw_packed, int32_t N, int32_t H, int32_t W, evk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
