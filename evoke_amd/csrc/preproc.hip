// preproc.hip -- GPU image pre-processing of the input pipeline: uint8 HWC pixels ->
// transforms.Resize (PIL bilinear, antialiased) -> crop -> [horizontal flip] -> [RandomRotation, nearest] -> ToTensor ->
// Normalize, written as one f32 CHW image of the batch tensor (modules/dataloaders_v0623.py:22-37 / dataloaders_v0401.py:25-37;
// the arithmetic lives in torchvision 0.16.2's PIL backend = Pillow's ImagingResample / ImagingTransformAffine, restated here).
//
// Bit-exactness plan: Pillow resamples uint8 images in two integer passes (horizontal, then vertical) with 22-bit
// fixed-point coefficients derived from double-precision weights; coeff_kernel evaluates the same double expressions
// (contraction off, so no FMA changes a rounding), resize_h_kernel is the horizontal pass (uint8 out, as Pillow rounds
// between the passes) and finish_kernel evaluates the vertical pass only at the pixels the crop / flip / rotation
// actually read, then applies x / 255, (x - mean) / std in f32.
// HBM-bound byte work: per output image the kernels read src_h x src_w x C bytes once and write 3 x S x S floats.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct CoefP { int in_size, out_size, ksize; int* bounds; int* kk; };

// Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1.0), box = whole axis
__global__ void coeff_kernel(const CoefP p) {
#pragma clang fp contract(off)
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= p.out_size) return;
  const double scale = (double)((float)p.in_size - 0.0f) / p.out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const double center = 0.0f + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > p.in_size) xmax = p.in_size;
  xmax -= xmin;
  int* kk = p.kk + (long)xx * p.ksize;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double t = (x + xmin - center + 0.5) * ss;
    if (t < 0.0) t = -t;
    const double w = t < 1.0 ? 1.0 - t : 0.0;
    ww += w;
  }
  for (int x = 0; x < p.ksize; ++x) {
    double k = 0.0;
    if (x < xmax) {
      double t = (x + xmin - center + 0.5) * ss;
      if (t < 0.0) t = -t;
      k = t < 1.0 ? 1.0 - t : 0.0;
      if (ww != 0.0) k /= ww;
    }
    kk[x] = k < 0 ? (int)(-0.5 + k * (1 << PRECISION_BITS)) : (int)(0.5 + k * (1 << PRECISION_BITS));
  }
  p.bounds[2 * xx] = xmin;
  p.bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct ResHP { const unsigned char* src; unsigned char* tmp; int src_h, src_w, ch, rw, ksize; const int* bounds; const int* kk; };

// horizontal pass: tmp[y][x][c] for every source row y and resized column x
__global__ void resize_h_kernel(const ResHP p) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)p.src_h * p.rw;
  if (i >= total) return;
  const int y = (int)(i / p.rw), x = (int)(i - (long)y * p.rw);
  const int xmin = p.bounds[2 * x], n = p.bounds[2 * x + 1];
  const int* k = p.kk + (long)x * p.ksize;
  const unsigned char* row = p.src + ((long)y * p.src_w + xmin) * p.ch;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  if (p.ch == 3) {
    for (int j = 0; j < n; ++j) { const int w = k[j]; s0 += row[3 * j] * w; s1 += row[3 * j + 1] * w; s2 += row[3 * j + 2] * w; }
  } else {
    for (int j = 0; j < n; ++j) s0 += row[j] * k[j];
  }
  unsigned char* o = p.tmp + i * p.ch;
  o[0] = (unsigned char)clip8(s0);
  if (p.ch == 3) { o[1] = (unsigned char)clip8(s1); o[2] = (unsigned char)clip8(s2); }
}

struct FinP {
  const unsigned char* tmp; float* out; int src_h, rw, rh, ch, S, top, left, flip, rotate, ksize;
  int a0, a1, a2, a3, a4, a5;
  const int* bounds; const int* kk; float mean[3], stdv[3];
};

// out[c][y][x] = normalize(rotate(flip(crop(vertical pass))))
__global__ void finish_kernel(const FinP p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.S * p.S) return;
  const int y = i / p.S, x = i - y * p.S;
  int sx = x, sy = y;
  bool inside = true;
  if (p.rotate) {      // Pillow Geometry.c affine_fixed: 16.16 fixed point, nearest neighbour, fill 0 outside
    const int xx = p.a2 + x * p.a0 + y * p.a1, yy = p.a5 + x * p.a3 + y * p.a4;
    sx = xx >> 16;
    sy = yy >> 16;
    inside = sx >= 0 && sx < p.S && sy >= 0 && sy < p.S;
  }
  int v[3] = {0, 0, 0};
  if (inside) {
    if (p.flip) sx = p.S - 1 - sx;
    const int ry = sy + p.top, rx = sx + p.left;       // coordinates in the resized image
    const int ymin = p.bounds[2 * ry], n = p.bounds[2 * ry + 1];
    const int* k = p.kk + (long)ry * p.ksize;
    const unsigned char* col = p.tmp + ((long)ymin * p.rw + rx) * p.ch;
    const long stride = (long)p.rw * p.ch;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    if (p.ch == 3) {
      for (int j = 0; j < n; ++j) { const int w = k[j]; s0 += col[j * stride] * w; s1 += col[j * stride + 1] * w; s2 += col[j * stride + 2] * w; }
      v[0] = clip8(s0); v[1] = clip8(s1); v[2] = clip8(s2);
    } else {
      for (int j = 0; j < n; ++j) s0 += col[j * stride] * k[j];
      v[0] = v[1] = v[2] = clip8(s0);                  // PIL .convert('RGB') of a grey image replicates the band
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float t = (float)v[c] / 255.0f;              // ToTensor
    p.out[(long)c * p.S * p.S + i] = (t - p.mean[c]) / p.stdv[c];   // Normalize
  }
}

inline int ksize_for(int in_size, int out_size) {
  double fs = (double)((float)in_size) / out_size;
  if (fs < 1.0) fs = 1.0;
  return (int)ceil(1.0 * fs) * 2 + 1;
}

}  // namespace

extern "C" {

int64_t evk_preprocess_ws_bytes(const evk_preproc_desc* d) {
  if (!d || d->src_h <= 0 || d->src_w <= 0 || d->resize_h <= 0 || d->resize_w <= 0 || (d->channels != 1 && d->channels != 3)) return -1;
  const int64_t kh = ksize_for(d->src_w, d->resize_w), kv = ksize_for(d->src_h, d->resize_h);
  int64_t b = 0;
  b += ((int64_t)d->resize_w * (kh + 2) + (int64_t)d->resize_h * (kv + 2)) * 4;
  b = (b + 255) & ~255LL;
  b += (int64_t)d->src_h * d->resize_w * d->channels;
  return (b + 255) & ~255LL;
}

int evk_preprocess_image(const evk_preproc_desc* d, void* ws, int64_t ws_bytes, float* out, evk_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EVK_REQUIRE(d && d->src && ws && out, "preprocess: null argument");
  const int64_t need = evk_preprocess_ws_bytes(d);
  EVK_REQUIRE(need > 0 && ws_bytes >= need, "preprocess: bad descriptor or workspace too small (%ld bytes needed)", (long)need);
  EVK_REQUIRE(d->out_size > 0 && d->crop_top >= 0 && d->crop_left >= 0 && d->crop_top + d->out_size <= d->resize_h &&
              d->crop_left + d->out_size <= d->resize_w, "preprocess: crop window %d+%d x %d+%d outside the %dx%d resized image",
              d->crop_top, d->out_size, d->crop_left, d->out_size, d->resize_h, d->resize_w);
  EVK_REQUIRE(d->src_h < 32768 && d->src_w < 32768, "preprocess: image too large");
  const int kh = ksize_for(d->src_w, d->resize_w), kv = ksize_for(d->src_h, d->resize_h);
  int* hb = static_cast<int*>(ws);
  int* hk = hb + 2 * d->resize_w;
  int* vb = hk + (long)d->resize_w * kh;
  int* vk = vb + 2 * d->resize_h;
  const int64_t tabs = (((int64_t)d->resize_w * (kh + 2) + (int64_t)d->resize_h * (kv + 2)) * 4 + 255) & ~255LL;
  unsigned char* tmp = static_cast<unsigned char*>(ws) + tabs;
  ProfScope ps(EVK_FAM_ELTWISE, s);
  hipLaunchKernelGGL(coeff_kernel, dim3((int)cdiv(d->resize_w, 128)), dim3(128), 0, s, CoefP{d->src_w, d->resize_w, kh, hb, hk});
  hipLaunchKernelGGL(coeff_kernel, dim3((int)cdiv(d->resize_h, 128)), dim3(128), 0, s, CoefP{d->src_h, d->resize_h, kv, vb, vk});
  ResHP h{static_cast<const unsigned char*>(d->src), tmp, d->src_h, d->src_w, d->channels, d->resize_w, kh, hb, hk};
  hipLaunchKernelGGL(resize_h_kernel, dim3((int)cdiv((long)d->src_h * d->resize_w, 256)), dim3(256), 0, s, h);
  FinP f{tmp, out, d->src_h, d->resize_w, d->resize_h, d->channels, d->out_size, d->crop_top, d->crop_left, d->flip, d->rotate, kv,
         d->affine[0], d->affine[1], d->affine[2], d->affine[3], d->affine[4], d->affine[5], vb, vk,
         {d->mean[0], d->mean[1], d->mean[2]}, {d->std[0], d->std[1], d->std[2]}};
  hipLaunchKernelGGL(finish_kernel, dim3((int)cdiv((long)d->out_size * d->out_size, 256)), dim3(256), 0, s, f);
  return evk_check_launch("preprocess");
}

}  // extern "C"
