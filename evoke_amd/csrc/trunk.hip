// trunk.hip -- native runner of the ResNet bottleneck trunk (visual_extractor.py:27-43 -> torchvision resnet101
// children 0-7): one C call runs the whole forward (stem, max-pool, 33 bottlenecks) and one the whole backward, so the
// ~1200 kernel launches of a training step are issued from C++ instead of one Python autograd node per op.
//
// Workspace (one caller-provided buffer, laid out by plan()): for every conv+BN pair the raw conv output y, the
// post-BN activation z and its statistics (saved for the backward), and for the backward a gradient buffer dy per pair
// plus six rotating gradient buffers for the block-level tensors.  Every dy buffer is unique, so the weight-gradient
// GEMMs can run on a second stream (they feed nothing before the optimizer) without slot-reuse hazards.
#include <vector>
#include <stdlib.h>
#include "common.h"

namespace {

struct Pair {               // one conv + batch-norm
  evk_conv_geom g;          // conv geometry (unused for the stem)
  int C;                    // output channels
  long M;                   // output rows N*Ho*Wo
  long y, z, stats, dy, sums;   // workspace offsets (bytes)
};

struct Plan {
  std::vector<Pair> pairs;  // [0] stem, then per block conv1, conv2, conv3, (downsample)
  std::vector<int> has_down;  // per block
  int N, H, W;
  long xpad, wp, dwp, pooled, pool_idx, red, slab, zeros, part, gbuf[6];
  std::vector<long> wflip;   // per pair: offset of the flipped / transposed 16-bit weights (stride-1 3x3 convolutions), or -1
  std::vector<long> wcls;    // per pair: offset of the parity-class weights of a stride-2 3x3 convolution (evk_conv3x3s2_class_weights), or -1
  long gcap, slab_bytes, red_bytes, part_bytes;
  long total;
};

inline long align256(long x) { return (x + 255) & ~255L; }

evk_conv_geom geom(int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad) {
  evk_conv_geom g{};
  g.N = N; g.Hi = Hi; g.Wi = Wi; g.Ci = Ci; g.Co = Co; g.KH = g.KW = k; g.stride_h = g.stride_w = stride; g.pad_h = g.pad_w = pad;
  g.Ho = (Hi + 2 * pad - k) / stride + 1; g.Wo = (Wi + 2 * pad - k) / stride + 1;
  g.sN = (int64_t)Hi * Wi * Ci; g.sH = (int64_t)Wi * Ci; g.sW = Ci;
  return g;
}

// infer: the plan of a forward that no backward follows.  Nothing but a block's input has to outlive the block, so the activations
// live in TWO arenas the blocks alternate between (block k writes arena k & 1 and reads its input -- block k-1's output, or the pooled
// stem output -- from the other one): 6 GB instead of 40 for 128 images at 384^2.  No gradient / slab / flipped-weight regions.
int make_plan(const evk_trunk_cfg* cfg, int N, int H, int W, Plan& P, bool infer = false) {
  EVK_REQUIRE(cfg && N > 0 && H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0, "trunk: image size must be a multiple of 32 (got %dx%d)", H, W);
  P.N = N; P.H = H; P.W = W;
  long off = 0;
  auto take = [&](long bytes) { long o = off; off = align256(off + bytes); return o; };
  long arena[2] = {0, 0}, acur = 0, asize = 0;
  auto atake = [&](long bytes) { long o = acur; acur = align256(acur + bytes); return o; };   // (inference: from the current arena)
  if (infer) {
    // the largest arena any block (or the stem) needs
    long need = align256((long)N * (H + 6) * (W + 8) * 4 * 2) + 2 * align256((long)N * (H / 2) * (W / 2) * 64 * 2) + align256((long)N * (H / 4) * (W / 4) * 64 * 2);
    int h0 = H / 4, w0 = W / 4;
    for (int L = 0; L < 4; ++L)
      for (int b = 0; b < cfg->blocks[L]; ++b) {
        const int stride = b == 0 ? cfg->stride[L] : 1, pl = cfg->planes[L];
        const int ho = (h0 - 1) / stride + 1, wo = (w0 - 1) / stride + 1;
        long nb = 2 * align256((long)N * h0 * w0 * pl * 2) + 2 * align256((long)N * ho * wo * pl * 2) + 2 * align256((long)N * ho * wo * pl * 4 * 2);
        if (b == 0) nb += 2 * align256((long)N * ho * wo * pl * 4 * 2);
        if (nb > need) need = nb;
        h0 = ho; w0 = wo;
      }
    asize = need;
    arena[0] = take(need); arena[1] = take(need);
    acur = arena[1];                                    // the stem plays block "-1"
  }
  P.xpad = infer ? atake((long)N * (H + 6) * (W + 8) * 4 * 2) : take((long)N * (H + 6) * (W + 8) * 4 * 2);
  P.wp = take(64 * 224 * 2);
  P.dwp = infer ? -1 : take(64 * 224 * 4);
  long gcap = 0, slab = infer ? 0 : evk_stem_wgrad_ws_bytes(N, H, W), part = 0;
  auto add_pair = [&](const evk_conv_geom& g, int C, long M) {
    Pair pr{};
    pr.g = g; pr.C = C; pr.M = M;
    if (infer) {
      pr.y = atake(M * C * 2); pr.z = atake(M * C * 2); pr.stats = take(6L * C * 4);
      pr.dy = pr.sums = -1;
    } else {
      pr.y = take(M * C * 2); pr.z = take(M * C * 2); pr.stats = take(6L * C * 4);
      pr.dy = take(M * C * 2); pr.sums = take(2L * C * 4);
      if (M * C * 2 > gcap) gcap = M * C * 2;
      if (evk_conv_stats_bytes(M, C) > part) part = evk_conv_stats_bytes(M, C);
    }
    P.pairs.push_back(pr);
  };
  int h = H / 2, w = W / 2;
  add_pair(evk_conv_geom{}, 64, (long)N * h * w);                 // stem
  h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;
  P.pooled = infer ? atake((long)N * h * w * 64 * 2) : take((long)N * h * w * 64 * 2);
  P.pool_idx = infer ? -1 : take((long)N * h * w * 64);
  int inpl = 64, blk = 0;
  for (int L = 0; L < 4; ++L) {
    const int planes = cfg->planes[L];
    EVK_REQUIRE(cfg->blocks[L] >= 1 && planes >= 8 && (planes & (planes - 1)) == 0, "trunk: bad layer %d config", L);
    for (int b = 0; b < cfg->blocks[L]; ++b) {
      const int stride = b == 0 ? cfg->stride[L] : 1;
      const bool down = b == 0;
      const int ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
      if (infer) {
        EVK_REQUIRE(acur <= arena[(blk + 1) & 1] + asize, "trunk: inference arena overflow before block %d", blk);
        acur = arena[blk & 1];
      }
      ++blk;
      evk_conv_geom g1 = geom(N, h, w, inpl, planes, 1, 1, 0);
      evk_conv_geom g2 = geom(N, h, w, planes, planes, 3, stride, 1);
      evk_conv_geom g3 = geom(N, ho, wo, planes, planes * 4, 1, 1, 0);
      add_pair(g1, planes, (long)N * h * w);
      if (!infer && evk_conv_xstat_bytes(&g1) > part) part = evk_conv_xstat_bytes(&g1);
      add_pair(g2, planes, (long)N * ho * wo);
      add_pair(g3, planes * 4, (long)N * ho * wo);
      if (down) {
        evk_conv_geom gd = geom(N, h, w, inpl, planes * 4, 1, stride, 0);
        add_pair(gd, planes * 4, (long)N * ho * wo);
      }
      P.has_down.push_back(down ? 1 : 0);
      if (infer) {
        EVK_REQUIRE(acur <= arena[(blk - 1) & 1] + asize, "trunk: inference arena overflow (block %d)", blk - 1);
        inpl = planes * 4; h = ho; w = wo;
        continue;
      }
      for (size_t i = P.pairs.size() - (down ? 4 : 3); i < P.pairs.size(); ++i) {
        const long nb = evk_conv2d_wgrad_ws_bytes(&P.pairs[i].g);
        if (nb > slab) slab = nb;
      }
      if ((long)N * h * w * inpl * 2 > gcap) gcap = (long)N * h * w * inpl * 2;
      inpl = planes * 4; h = ho; w = wo;
    }
  }
  P.wflip.assign(P.pairs.size(), -1);
  P.wcls.assign(P.pairs.size(), -1);
  if (infer) {
    P.gcap = 0; P.red_bytes = P.slab_bytes = P.part_bytes = 0;
    P.red = P.slab = P.part = -1;
    for (int i = 0; i < 6; ++i) P.gbuf[i] = -1;
    P.zeros = take(2 * 2048 * 4);
    P.total = off;
    return EVK_OK;
  }
  P.gcap = gcap;
  for (int i = 0; i < 6; ++i) P.gbuf[i] = take(gcap);
  P.red_bytes = evk_colreduce_ws_bytes(2048);
  P.red = take(P.red_bytes);
  P.slab_bytes = slab;
  P.slab = take(slab);
  P.zeros = take(2 * 2048 * 4);
  P.part_bytes = part;
  P.part = take(part);
  for (size_t i = 1; i < P.pairs.size(); ++i) {
    const evk_conv_geom& g = P.pairs[i].g;
    if (evk_conv3x3s2_dgrad_parity_supported(&g) && evk_conv3x3s2_dgrad_parity_ws_bytes(&g) <= gcap) P.wcls[i] = take((long)g.Co * 9 * g.Ci * 2);
    if (g.KH == 3 && g.KW == 3 && g.stride_h == 1 && g.stride_w == 1 && g.pad_h == 1 && g.pad_w == 1)
      P.wflip[i] = take((long)g.Co * 9 * g.Ci * 2);
    // contracting pointwise data gradients (Bottleneck.conv3) the strip GEMM takes: their weights transposed to [Ci][Co]
    else if (g.KH == 1 && g.KW == 1 && g.stride_h == 1 && g.stride_w == 1 && g.pad_h == 0 && g.pad_w == 0 && g.Co > g.Ci &&
             evk_gemm_strip_routes((int64_t)g.N * g.Hi * g.Wi, g.Ci, g.Co, 0, 0))
      P.wflip[i] = take((long)g.Co * g.Ci * 2);
  }
  P.total = off;
  return EVK_OK;
}

#define TRY(expr) do { int rc_ = (expr); if (rc_ != EVK_OK) return rc_; } while (0)

struct Ctx {
  const evk_trunk_cfg* cfg; const evk_trunk_layer* L; char* ws; Plan* P; evk_stream_t s; int training;
  template <typename T = void> T* at(long off) const { return reinterpret_cast<T*>(ws + off); }
};

// y = conv(x), batch statistics / running statistics -> scale, shift; z = relu?(y * scale + shift + resid)
int bn_forward(const Ctx& c, int i, const void* resid, int relu, int nblk) {
  const Pair& pr = c.P->pairs[i];
  const evk_trunk_layer& l = c.L[i];
  float* st = c.at<float>(pr.stats);
  const int C = pr.C;
  if (c.training)        // the conv epilogue left the partial sums: second reduction stage + statistics in one launch
    TRY(evk_bn_stats_finalize_from_partials(c.at<float>(c.P->part), nblk, st, st + C, l.gamma, l.beta, l.running_mean, l.running_var, st + 2 * C,
                                            st + 3 * C, st + 4 * C, st + 5 * C, C, (float)pr.M, c.cfg->momentum, c.cfg->eps, c.s));
  else
    TRY(evk_bn_finalize(st, st + C, l.gamma, l.beta, l.running_mean, l.running_var, st + 2 * C, st + 3 * C, st + 4 * C, st + 5 * C, C,
                        (float)pr.M, c.cfg->momentum, c.cfg->eps, c.training, c.s));
  return evk_bn_apply(c.at(pr.y), st + 2 * C, st + 3 * C, resid, c.at(pr.z), pr.M, C, relu, c.s);
}

// bn_backward whose two column sums come from the partial rows the producing data-gradient GEMM's epilogue wrote (no pass over dz)
int bn_backward_from_gate(const Ctx& c, int i, const void* dz, int nblk) {
  const Pair& pr = c.P->pairs[i];
  const evk_trunk_layer& l = c.L[i];
  float* st = c.at<float>(pr.stats);
  float* sums = c.at<float>(pr.sums);
  const int C = pr.C;
  TRY(evk_bn_bwd_sums_from_gate_partials(c.at<float>(c.P->part), nblk, l.gamma, l.beta, sums, sums + C, l.dbeta, l.dgamma, C, dz, c.at(pr.y),
                                         st + 4 * C, st + 5 * C, pr.M, c.s));
  const float* sg = c.training ? sums : c.at<float>(c.P->zeros);
  const float* sgx = c.training ? sums + C : c.at<float>(c.P->zeros);
  return evk_bn_bwd_apply(dz, c.at(pr.z), c.at(pr.y), st + 2 * C, st + 4 * C, st + 5 * C, sg, sgx, c.at(pr.dy), nullptr, pr.M, C, 0, c.s);
}

// bn_backward whose column sums (sum g, sum g*(x - mean)) were left by the data gradient that produced dz (evk_conv2d_dgrad_gated_xstat)
int bn_backward_from_xstat(const Ctx& c, int i, const void* dz, int nblk) {
  const Pair& pr = c.P->pairs[i];
  const evk_trunk_layer& l = c.L[i];
  float* st = c.at<float>(pr.stats);
  float* sums = c.at<float>(pr.sums);
  const int C = pr.C;
  TRY(evk_bn_bwd_sums_from_xstat_partials(c.at<float>(c.P->part), nblk, st + 5 * C, sums, sums + C, l.dbeta, l.dgamma, C, c.s));
  const float* sg = c.training ? sums : c.at<float>(c.P->zeros);
  const float* sgx = c.training ? sums + C : c.at<float>(c.P->zeros);
  return evk_bn_bwd_apply(dz, c.at(pr.z), c.at(pr.y), st + 2 * C, st + 4 * C, st + 5 * C, sg, sgx, c.at(pr.dy), nullptr, pr.M, C, 0, c.s);
}

// dz (gradient w.r.t. z) -> dy (gradient w.r.t. the conv output), optional dres (= masked dz, the skip-branch gradient)
int bn_backward(const Ctx& c, int i, const void* dz, void* dres, int relu) {
  const Pair& pr = c.P->pairs[i];
  const evk_trunk_layer& l = c.L[i];
  float* st = c.at<float>(pr.stats);
  float* sums = c.at<float>(pr.sums);
  const int C = pr.C;
  TRY(evk_bn_bwd_reduce_acc(dz, c.at(pr.z), c.at(pr.y), st + 4 * C, st + 5 * C, sums, sums + C, l.dbeta, l.dgamma, c.at(c.P->red),
                            c.P->red_bytes, pr.M, C, relu, c.s));
  const float* sg = c.training ? sums : c.at<float>(c.P->zeros);       // eval-mode BN is a fixed affine map
  const float* sgx = c.training ? sums + C : c.at<float>(c.P->zeros);
  return evk_bn_bwd_apply(dz, c.at(pr.z), c.at(pr.y), st + 2 * C, st + 4 * C, st + 5 * C, sg, sgx, c.at(pr.dy), dres, pr.M, C, relu, c.s);
}

struct WgradQueue {      // weight gradients on a second stream, ordered after the main-stream kernel that produced dy
  hipStream_t main, side;
  hipEvent_t ev[8];
  int n = 0;
  bool ok = false;
  int init(evk_stream_t m, evk_stream_t sd) {
    main = reinterpret_cast<hipStream_t>(m);
    side = reinterpret_cast<hipStream_t>(sd);
    ok = sd != nullptr && sd != m;
    if (ok) {
      static thread_local hipEvent_t pool[8];
      static thread_local bool made = false;
      if (!made) {
        for (int i = 0; i < 8; ++i)
          if (hipEventCreateWithFlags(&pool[i], hipEventDisableTiming) != hipSuccess) { evk_set_error("trunk: hipEventCreate failed"); return EVK_ELAUNCH; }
        made = true;
      }
      for (int i = 0; i < 8; ++i) ev[i] = pool[i];
    }
    return EVK_OK;
  }
  evk_stream_t fork() {        // returns the stream to launch the weight gradient on
    if (!ok) return reinterpret_cast<evk_stream_t>(main);
    hipEvent_t e = ev[n++ & 7];
    (void)hipEventRecord(e, main);
    (void)hipStreamWaitEvent(side, e, 0);
    return reinterpret_cast<evk_stream_t>(side);
  }
};

}  // namespace

extern "C" {

int64_t evk_trunk_ws_bytes(const evk_trunk_cfg* cfg, int32_t N, int32_t H, int32_t W) {
  Plan P;
  if (make_plan(cfg, N, H, W, P) != EVK_OK) return -1;
  return P.total;
}

/* workspace of evk_trunk_forward_inference: two arenas the blocks alternate between (6 GB for 128 images at 384^2; the training layout
 * of evk_trunk_ws_bytes keeps every activation for the backward: 40 GB) */
int64_t evk_trunk_infer_ws_bytes(const evk_trunk_cfg* cfg, int32_t N, int32_t H, int32_t W) {
  Plan P;
  if (make_plan(cfg, N, H, W, P, true) != EVK_OK) return -1;
  return P.total;
}

int evk_trunk_num_pairs(const evk_trunk_cfg* cfg) {
  if (!cfg) return -1;
  int n = 1;
  for (int L = 0; L < 4; ++L) n += 3 * cfg->blocks[L] + 1;
  return n;
}

int evk_trunk_forward(const evk_trunk_cfg* cfg, const evk_trunk_layer* layers, int32_t n_layers, const float* images, int32_t N,
                      int32_t H, int32_t W, void* ws, int64_t ws_bytes, void* out, int32_t training, evk_stream_t stream) {
  Plan P;
  TRY(make_plan(cfg, N, H, W, P));
  EVK_REQUIRE(layers && images && ws && out, "trunk_forward: null argument");
  EVK_REQUIRE(n_layers == (int)P.pairs.size(), "trunk_forward: expected %d conv+bn pairs, got %d", (int)P.pairs.size(), n_layers);
  EVK_REQUIRE(ws_bytes >= P.total, "trunk_forward: workspace too small (%ld bytes needed)", P.total);
  Ctx c{cfg, layers, static_cast<char*>(ws), &P, stream, training};
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(c.at(P.zeros), 0, 2 * 2048 * 4, hs) != hipSuccess) { evk_set_error("trunk: memset failed"); return EVK_ELAUNCH; }

  // conv + (training) batch statistics in its epilogue, then BN finalize + apply
  float* part = training ? c.at<float>(P.part) : nullptr;
  int nblk = 0;
  auto conv_bn = [&](int i, const void* xin, const void* resid, int relu) -> int {
    TRY(evk_conv2d_fwd_stats(xin, layers[i].w, c.at(P.pairs[i].y), &P.pairs[i].g, part, P.part_bytes, &nblk, stream));
    return bn_forward(c, i, resid, relu, nblk);
  };

  // stem: pack, conv 7x7 s2, BN + ReLU, max-pool 3x3 s2
  TRY(evk_stem_pack_image(images, c.at(P.xpad), N, H, W, stream));
  TRY(evk_stem_pack_weight(reinterpret_cast<const float*>(layers[0].w), c.at(P.wp), stream));
  TRY(evk_stem_fwd_stats(c.at(P.xpad), c.at(P.wp), c.at(P.pairs[0].y), N, H, W, part, P.part_bytes, &nblk, stream));
  TRY(bn_forward(c, 0, nullptr, 1, nblk));
  TRY(evk_maxpool3x3s2_fwd_idx(c.at(P.pairs[0].z), c.at(P.pooled), c.at(P.pool_idx), N, H / 2, W / 2, 64, stream));

  const void* x = c.at(P.pooled);
  int i = 1;
  for (size_t b = 0; b < P.has_down.size(); ++b) {
    const bool down = P.has_down[b];
    const void* idt = x;
    if (down) {
      TRY(conv_bn(i + 3, x, nullptr, 0));
      idt = c.at(P.pairs[i + 3].z);
    }
    TRY(conv_bn(i, x, nullptr, 1));
    TRY(conv_bn(i + 1, c.at(P.pairs[i].z), nullptr, 1));
    TRY(conv_bn(i + 2, c.at(P.pairs[i + 1].z), idt, 1));
    x = c.at(P.pairs[i + 2].z);
    i += down ? 4 : 3;
  }
  const Pair& last = P.pairs[i - (P.has_down.back() ? 4 : 3) + 2];
  // a copy KERNEL, not hipMemcpyAsync: in a stream capture the 37 MB copy becomes a memcpy node that ROCm 7.2 reports as multi-dimensional,
  // which the step replayer (replay.hip) cannot re-issue
  TRY(evk_cast(x, EVK_BF16, out, EVK_BF16, last.M * last.C, stream));
  return EVK_OK;
}

/* Inference forward (no backward will follow): every batch norm behind a convolution is applied by that convolution's epilogue to its f32
 * accumulators as a per-channel scale / shift (evk_bn_eval_coeffs: gamma / sqrt(var + eps), beta - mean * gamma / sqrt(var + eps)), together
 * with the identity and the ReLU -- the 2 x 100 bn_finalize / bn_apply launches of the eval-mode forward (6.3 ms of a 15.4 ms pass over 128
 * images at 384^2: one more read and write of every activation) are gone, and the activations are rounded to 16 bits once instead of twice.
 *   fold_ws: evk_trunk_fold_bytes(cfg) bytes the CALLER keeps between calls (the scale / shift vectors); refold bit 0: (re)compute them (the
 *   affine parameters or the running statistics changed since the last call), else reuse them; bit 1: see conv_affine below.  The stem and the 9 convolutions the tile GEMM
 *   takes (stride 2, Co = 64) keep the unfused path. */
int64_t evk_trunk_fold_bytes(const evk_trunk_cfg* cfg) {
  Plan P;
  if (make_plan(cfg, 1, 32, 32, P) != EVK_OK) return -1;
  long tot = 0;
  for (size_t i = 1; i < P.pairs.size(); ++i) {
    const evk_conv_geom& g = P.pairs[i].g;
    tot += 2 * align256((long)g.Co * 4);
  }
  return tot + 256;
}

int evk_trunk_forward_inference(const evk_trunk_cfg* cfg, const evk_trunk_layer* layers, int32_t n_layers, const float* images, int32_t N,
                                int32_t H, int32_t W, void* ws, int64_t ws_bytes, void* out, void* fold_ws, int64_t fold_bytes, int32_t refold,
                                evk_stream_t stream) {
  Plan P;
  TRY(make_plan(cfg, N, H, W, P, true));
  EVK_REQUIRE(layers && images && ws && out && fold_ws, "trunk_forward_inference: null argument");
  EVK_REQUIRE(n_layers == (int)P.pairs.size(), "trunk_forward_inference: expected %d conv+bn pairs, got %d", (int)P.pairs.size(), n_layers);
  EVK_REQUIRE(ws_bytes >= P.total && fold_bytes >= evk_trunk_fold_bytes(cfg), "trunk_forward_inference: workspace too small");
  Ctx c{cfg, layers, static_cast<char*>(ws), &P, stream, 0};
  // per pair the eval-mode batch norm as scale / shift vectors (the layout does not depend on the image size)
  std::vector<float*> sc(P.pairs.size(), nullptr), sh(P.pairs.size(), nullptr);
  {
    char* f = static_cast<char*>(fold_ws);
    for (size_t i = 1; i < P.pairs.size(); ++i) {
      const evk_conv_geom& g = P.pairs[i].g;
      sc[i] = reinterpret_cast<float*>(f); f += align256((long)g.Co * 4);
      sh[i] = reinterpret_cast<float*>(f); f += align256((long)g.Co * 4);
      if (refold & 1)
        TRY(evk_bn_eval_coeffs(layers[i].gamma, layers[i].beta, layers[i].running_mean, layers[i].running_var, cfg->eps, sc[i], sh[i], g.Co, stream));
    }
  }
  int nblk = 0;
  // z = relu?(conv(x, w) * scale + shift (+ resid)) in one launch where a route has the epilogue; conv, then the eval-mode bn pass, otherwise
  // refold == 2 / the caller's mode bit 1: the convolutions with an identity keep conv + bn_apply (with the cached vectors: no bn_finalize
  // launch) -- their fused form holds one 8-wave workgroup per CU for ~130 us, which a serving loop's small decode kernels then queue behind
  // bit 2: nothing fused at all -- conv + bn_apply per layer, i.e. the eval forward of evk_trunk_forward minus its bn_finalize launches
  const bool fuse_identity = !(refold & 2), fuse = !(refold & 4);
  auto conv_affine = [&](int i, const void* xin, const void* resid, int relu) -> int {
    const evk_conv_geom& g = P.pairs[i].g;
    if (fuse && evk_conv2d_fwd_affine_routes(&g) && (!resid || fuse_identity))
      return evk_conv2d_fwd_affine(xin, layers[i].w, c.at(P.pairs[i].z), &g, sc[i], sh[i], resid, relu, stream);
    TRY(evk_conv2d_fwd_stats(xin, layers[i].w, c.at(P.pairs[i].y), &g, nullptr, 0, &nblk, stream));
    return evk_bn_apply(c.at(P.pairs[i].y), sc[i], sh[i], resid, c.at(P.pairs[i].z), P.pairs[i].M, P.pairs[i].C, relu, stream);
  };
  TRY(evk_stem_pack_image(images, c.at(P.xpad), N, H, W, stream));
  TRY(evk_stem_pack_weight(reinterpret_cast<const float*>(layers[0].w), c.at(P.wp), stream));
  TRY(evk_stem_fwd_stats(c.at(P.xpad), c.at(P.wp), c.at(P.pairs[0].y), N, H, W, nullptr, 0, &nblk, stream));
  TRY(bn_forward(c, 0, nullptr, 1, nblk));
  TRY(evk_maxpool3x3s2_fwd(c.at(P.pairs[0].z), c.at(P.pooled), N, H / 2, W / 2, 64, stream));
  const void* x = c.at(P.pooled);
  int i = 1;
  for (size_t b = 0; b < P.has_down.size(); ++b) {
    const bool down = P.has_down[b];
    const void* idt = x;
    if (down) {
      TRY(conv_affine(i + 3, x, nullptr, 0));
      idt = c.at(P.pairs[i + 3].z);
    }
    TRY(conv_affine(i, x, nullptr, 1));
    TRY(conv_affine(i + 1, c.at(P.pairs[i].z), nullptr, 1));
    TRY(conv_affine(i + 2, c.at(P.pairs[i + 1].z), idt, 1));
    x = c.at(P.pairs[i + 2].z);
    i += down ? 4 : 3;
  }
  const Pair& last = P.pairs[i - (P.has_down.back() ? 4 : 3) + 2];
  TRY(evk_cast(x, EVK_BF16, out, EVK_BF16, last.M * last.C, stream));
  return EVK_OK;
}

int evk_trunk_backward(const evk_trunk_cfg* cfg, const evk_trunk_layer* layers, int32_t n_layers, int32_t N, int32_t H, int32_t W,
                       void* ws, int64_t ws_bytes, const void* dout, int32_t training, evk_stream_t stream, evk_stream_t wgrad_stream) {
  Plan P;
  TRY(make_plan(cfg, N, H, W, P));
  EVK_REQUIRE(layers && ws && dout, "trunk_backward: null argument");
  EVK_REQUIRE(n_layers == (int)P.pairs.size() && ws_bytes >= P.total, "trunk_backward: layer count / workspace mismatch");
  Ctx c{cfg, layers, static_cast<char*>(ws), &P, stream, training};
  WgradQueue q;
  TRY(q.init(stream, wgrad_stream));
  static const bool gate_stats = evk_tunable("EVK_BN_GATE_STATS", 1) != 0;
  static const bool x_stats = evk_tunable("EVK_BN_XSTATS", 1) != 0;
  int nbz = 0;                           // > 0: the kernel that produced gZ left bn3's backward sums of the current block in P.part
  static const bool flip_on = evk_tunable("EVK_DGRAD_FLIP", 1) != 0;
  if (flip_on) {        // flipped / transposed weights of every stride-1 3x3 convolution and every conv3 the strip GEMM takes, one launch (evk_conv_flip_weights)
    std::vector<const void*> ws_; std::vector<void*> wt_; std::vector<int32_t> co_, ci_, k_;
    for (size_t i = 1; i < P.pairs.size(); ++i)
      if (P.wflip[i] >= 0) {
        ws_.push_back(layers[i].w); wt_.push_back(c.at(P.wflip[i])); co_.push_back(P.pairs[i].g.Co); ci_.push_back(P.pairs[i].g.Ci); k_.push_back(P.pairs[i].g.KH);
      }
    if (!ws_.empty()) TRY(evk_conv_flip_weights(ws_.data(), wt_.data(), co_.data(), ci_.data(), k_.data(), k_.data(), (int32_t)ws_.size(), stream));
  }

  for (size_t i = 1; i < P.pairs.size(); ++i)         // parity-class weights of the stride-2 3x3 convolutions (three small launches)
    if (P.wcls[i] >= 0) TRY(evk_conv3x3s2_class_weights(layers[i].w, c.at(P.wcls[i]), P.pairs[i].g.Co, P.pairs[i].g.Ci, stream));

  // timing probe (EVK_EXPERIMENTAL=1 EVK_PROBE_SKIP_WGRAD=1: wrong gradients): how much of the step's wall time do the trunk's weight gradients cost?
  static const bool probe_skip_wgrad = evk_tunable("EVK_PROBE_SKIP_WGRAD", 0) != 0;
  auto wgrad = [&](int i, const void* xin) -> int {
    if (!layers[i].dw || probe_skip_wgrad) return EVK_OK;
    evk_stream_t s2 = q.fork();
    return evk_conv2d_wgrad(c.at(P.pairs[i].dy), xin, layers[i].dw, &P.pairs[i].g, c.at(P.slab), P.slab_bytes, s2);
  };

  // first pair index of every block
  std::vector<int> first(P.has_down.size());
  {
    int i = 1;
    for (size_t b = 0; b < P.has_down.size(); ++b) { first[b] = i; i += P.has_down[b] ? 4 : 3; }
  }
  // ReLU gates travel with the producer: every gradient tensor is zeroed where its forward activation was clipped by the
  // epilogue of the data-gradient GEMM that writes it (evk_conv2d_dgrad_gated), so the batch-norm backward passes read
  // neither a mask tensor nor write a separate masked copy for the skip branch.
  const Pair& lastp = P.pairs[first.back() + 2];
  void* g0 = c.at(P.gbuf[1]);
  TRY(evk_act_bwd(dout, c.at(lastp.z), g0, lastp.M * lastp.C, EVK_ACT_RELU, stream));
  const void* gZ = g0;                   // gated gradient w.r.t. the current block's output
  int flip = 0;
  for (int b = (int)P.has_down.size() - 1; b >= 0; --b) {
    const int i = first[b];
    const bool down = P.has_down[b];
    const void* X = b == 0 ? c.at(P.pooled) : c.at(P.pairs[first[b - 1] + 2].z);     // block input
    const void* xgate = b == 0 ? nullptr : X;      // X is the previous block's post-ReLU output (the pooled stem output has no ReLU)
    void* S1 = c.at(P.gbuf[3]);
    void* S2 = c.at(P.gbuf[4]);
    void* T = c.at(P.gbuf[5]);
    void* gX = c.at(P.gbuf[flip]);
    flip ^= 1;
    // bn2 / bn1 of the block: their backward sums ride on the data-gradient GEMM that produces their dz (gate statistics)
    float* part = c.at<float>(P.part);
    const bool gs1 = gate_stats && P.part_bytes >= evk_conv_stats_bytes(P.pairs[i + 1].M, P.pairs[i + 1].C);
    const bool gs0 = gate_stats && P.part_bytes >= evk_conv_stats_bytes(P.pairs[i].M, P.pairs[i].C);
    int nb1 = 0, nb0 = 0;
    if (nbz > 0) TRY(bn_backward_from_xstat(c, i + 2, gZ, nbz));
    else TRY(bn_backward(c, i + 2, gZ, nullptr, 0));
    if (flip_on && P.wflip[i + 2] >= 0)
      TRY(evk_conv2d_dgrad_flipped_gated_stats(c.at(P.pairs[i + 2].dy), c.at(P.wflip[i + 2]), nullptr, c.at(P.pairs[i + 1].z), S1, &P.pairs[i + 2].g,
                                               gs1 ? part : nullptr, P.part_bytes, &nb1, stream));
    else
      TRY(evk_conv2d_dgrad_gated_stats(c.at(P.pairs[i + 2].dy), layers[i + 2].w, nullptr, c.at(P.pairs[i + 1].z), S1, &P.pairs[i + 2].g,
                                       gs1 ? part : nullptr, P.part_bytes, &nb1, stream));
    TRY(wgrad(i + 2, c.at(P.pairs[i + 1].z)));
    if (gs1) TRY(bn_backward_from_gate(c, i + 1, S1, nb1));
    else TRY(bn_backward(c, i + 1, S1, nullptr, 0));
    if (P.wcls[i + 1] >= 0)        // stride 2: by output parity (2.25 taps per pixel instead of 9); gbuf[2] is free until the max-pool backward
      TRY(evk_conv3x3s2_dgrad_parity(c.at(P.pairs[i + 1].dy), c.at(P.wcls[i + 1]), c.at(P.pairs[i].z), S2, &P.pairs[i + 1].g, c.at(P.gbuf[2]),
                                     P.gcap, gs0 ? part : nullptr, P.part_bytes, &nb0, stream));
    else if (flip_on && P.wflip[i + 1] >= 0)
      TRY(evk_conv2d_dgrad_flipped_gated_stats(c.at(P.pairs[i + 1].dy), c.at(P.wflip[i + 1]), nullptr, c.at(P.pairs[i].z), S2, &P.pairs[i + 1].g,
                                               gs0 ? part : nullptr, P.part_bytes, &nb0, stream));
    else
      TRY(evk_conv2d_dgrad_gated_stats(c.at(P.pairs[i + 1].dy), layers[i + 1].w, nullptr, c.at(P.pairs[i].z), S2, &P.pairs[i + 1].g,
                                       gs0 ? part : nullptr, P.part_bytes, &nb0, stream));
    TRY(wgrad(i + 1, c.at(P.pairs[i].z)));
    if (gs0) TRY(bn_backward_from_gate(c, i, S2, nb0));
    else TRY(bn_backward(c, i, S2, nullptr, 0));
    const void* skip = gZ;
    if (down) {
      TRY(bn_backward(c, i + 3, gZ, nullptr, 0));
      {
        // down-sampling shortcut: a pointwise stride-2 convolution gives gradient to the even pixels only -- the product on the compact
        // rows (a quarter of them), then one pass that spreads them and writes the zeros (EVK_DOWN_COMPACT=0: the gathering GEMM)
        static const bool compact = evk_tunable("EVK_DOWN_COMPACT", 1) != 0;
        const evk_conv_geom& gd = P.pairs[i + 3].g;
        if (compact && gd.KH == 1 && gd.KW == 1 && gd.stride_h == 2 && gd.stride_w == 2 && gd.pad_h == 0 && gd.pad_w == 0 &&
            gd.Hi == 2 * gd.Ho && gd.Wi == 2 * gd.Wo && gd.Ci % 8 == 0) {
          evk_conv_geom gc = gd;
          gc.Hi = gd.Ho; gc.Wi = gd.Wo; gc.stride_h = gc.stride_w = 1;
          void* Tc = c.at(P.gbuf[2]);                 // free until the max-pool backward at the very end
          TRY(evk_conv2d_dgrad(c.at(P.pairs[i + 3].dy), layers[i + 3].w, Tc, &gc, stream));
          TRY(evk_upsample2_zero(Tc, T, gd.N, gd.Ho, gd.Wo, gd.Ci, stream));
        } else {
          TRY(evk_conv2d_dgrad(c.at(P.pairs[i + 3].dy), layers[i + 3].w, T, &P.pairs[i + 3].g, stream));
        }
      }
      TRY(wgrad(i + 3, X));
      skip = T;
    }
    // the gradient w.r.t. the block input IS the previous block's output gradient: its bn3 sums ride on this epilogue (weight-stationary
    // kernel only; nbz = 0 otherwise and that block reduces gX itself).  Training-mode statistics only: st + 4C holds the batch mean.
    nbz = 0;
    if (b > 0 && x_stats && training) {
      const int ip = first[b - 1] + 2;
      TRY(evk_conv2d_dgrad_gated_xstat(c.at(P.pairs[i].dy), layers[i].w, skip, xgate, gX, &P.pairs[i].g, c.at(P.pairs[ip].y),
                                       c.at<float>(P.pairs[ip].stats) + 4 * P.pairs[ip].C, part, P.part_bytes, &nbz, stream));
    } else {
      TRY(evk_conv2d_dgrad_gated(c.at(P.pairs[i].dy), layers[i].w, skip, xgate, gX, &P.pairs[i].g, stream));
    }
    TRY(wgrad(i, X));
    gZ = gX;
  }
  // max-pool, stem BN, stem weight gradient (images get no gradient)
  void* gS = c.at(P.gbuf[2]);
  TRY(evk_maxpool3x3s2_bwd_idx(c.at(P.pool_idx), gZ, gS, N, H / 2, W / 2, 64, stream));
  TRY(bn_backward(c, 0, gS, nullptr, 1));
  if (layers[0].dw) {
    evk_stream_t s2 = q.fork();
    hipStream_t h2 = reinterpret_cast<hipStream_t>(s2);
    if (hipMemsetAsync(c.at(P.dwp), 0, 64 * 224 * 4, h2) != hipSuccess) { evk_set_error("trunk: memset failed"); return EVK_ELAUNCH; }
    TRY(evk_stem_wgrad(c.at(P.pairs[0].dy), c.at(P.xpad), c.at<float>(P.dwp), N, H, W, c.at(P.slab), P.slab_bytes, s2));
    TRY(evk_stem_unpack_wgrad(c.at<float>(P.dwp), layers[0].dw, s2));
  }
  return EVK_OK;
}

}  // extern "C"
