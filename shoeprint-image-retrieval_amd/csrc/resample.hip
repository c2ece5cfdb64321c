// Rotation / scale variants of query feature maps (reference similarity.py:230-284: every channel goes
// through a Pillow mode-"F" image and Image.rotate(angle) / Image.resize(size)).  The kernels restate
// Pillow's C paths exactly so that variants are bit-identical to the reference's:
//
//  * rotate: NEAREST, no expand, zero fill.  Pillow's affine fast path (Geometry.c, affine_fixed) walks
//    the destination in 16.16 fixed point: source x = (a2 + y*a1 + x*a0) >> 16 (arithmetic shift), same
//    for y; 180 degrees is an exact flip, 90/270 on square maps exact transposes.  Integer arithmetic only.
//  * resize: BICUBIC with Pillow's separable two-pass resampler (Resample.c): per output pixel a window
//    [xmin, xmin+xmax) with double-precision normalised coefficients (computed on the host, see
//    variants.py), double accumulation in window order, result rounded to float32 after each pass.
//    Contraction into FMA is disabled so that each product and each sum rounds as in the C code.
// Both are pure gather streams: HBM-bound, 4 bytes read + 4 written per output pixel.
#include "spr_common.h"

namespace spr {
namespace {

// mode 0: copy, 1: rotate 180, 2: rotate 90 (h == w), 3: rotate 270 (h == w), 4: affine fixed point
__global__ void __launch_bounds__(kThreads)
rotate_kernel(const float* __restrict__ in, float* __restrict__ out, long long n_maps, int h, int w, int mode,
              long long a0, long long a1, long long a2, long long a3, long long a4, long long a5) {
  const long long per = static_cast<long long>(h) * w;
  const long long total = n_maps * per;
  for (long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * kThreads) {
    const long long map = i / per;
    const int rem = static_cast<int>(i - map * per);
    const int y = rem / w, x = rem - y * w;
    const float* src = in + map * per;
    float v = 0.0f;
    if (mode == 0) {
      v = src[rem];
    } else if (mode == 1) {
      v = src[(h - 1 - y) * w + (w - 1 - x)];
    } else if (mode == 2) {  // Pillow ROTATE_90:  out[w-1-xx][yy] = in[yy][xx]
      v = src[x * w + (w - 1 - y)];
    } else if (mode == 3) {  // Pillow ROTATE_270: out[xx][h-1-yy] = in[yy][xx]
      v = src[(h - 1 - x) * w + y];
    } else {
      const long long xin = (a2 + y * a1 + x * a0) >> 16;
      const long long yin = (a5 + y * a4 + x * a3) >> 16;
      if (xin >= 0 && xin < w && yin >= 0 && yin < h) v = src[yin * w + xin];
    }
    out[i] = v;
  }
}

// One pass of the separable resampler along `axis` (1: width, 0: height).
__global__ void __launch_bounds__(kThreads)
resample_kernel(const float* __restrict__ in, float* __restrict__ out, long long n_maps, int h, int w, int axis,
                int out_size, const int* __restrict__ bounds, const double* __restrict__ coeffs, int ksize) {
#pragma clang fp contract(off)
  const int oh = axis == 0 ? out_size : h, ow = axis == 1 ? out_size : w;
  const long long per_out = static_cast<long long>(oh) * ow, per_in = static_cast<long long>(h) * w;
  const long long total = n_maps * per_out;
  for (long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * kThreads) {
    const long long map = i / per_out;
    const int rem = static_cast<int>(i - map * per_out);
    const int y = rem / ow, x = rem - y * ow;
    const float* src = in + map * per_in;
    const int o = axis == 1 ? x : y;
    const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
    const double* k = coeffs + static_cast<long long>(o) * ksize;
    double ss = 0.0;
    for (int t = 0; t < cnt; ++t) {
      const float p = axis == 1 ? src[y * w + (t + lo)] : src[(t + lo) * w + x];
      const double prod = static_cast<double>(p) * k[t];
      ss = ss + prod;
    }
    out[i] = static_cast<float>(ss);
  }
}

inline unsigned grid_for(long long total) {
  long long b = (total + kThreads - 1) / kThreads;
  if (b > 4096) b = 4096;
  return static_cast<unsigned>(b < 1 ? 1 : b);
}

}  // namespace
}  // namespace spr

extern "C" int spr_rotate_nearest(const float* in, float* out, int64_t n_maps, int32_t h, int32_t w, int32_t mode,
                                  const int64_t* fixed6, spr_stream_t stream) {
  using namespace spr;
  if (n_maps < 0 || h < 1 || w < 1 || mode < 0 || mode > 4) { set_error("spr_rotate_nearest: bad sizes / mode"); return SPR_ERR_ARG; }
  if ((mode == 2 || mode == 3) && h != w) { set_error("spr_rotate_nearest: 90/270 need square maps"); return SPR_ERR_ARG; }
  if (n_maps == 0) return SPR_OK;
  if (!in || !out || (mode == 4 && !fixed6)) { set_error("spr_rotate_nearest: null pointer"); return SPR_ERR_ARG; }
  long long a[6] = {0, 0, 0, 0, 0, 0};
  if (mode == 4) for (int i = 0; i < 6; ++i) a[i] = fixed6[i];
  const long long total = n_maps * static_cast<long long>(h) * w;
  hipLaunchKernelGGL(rotate_kernel, dim3(grid_for(total)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), in, out,
                     static_cast<long long>(n_maps), h, w, mode, a[0], a[1], a[2], a[3], a[4], a[5]);
  return check_launch("rotate_kernel");
}

extern "C" int spr_resample_axis(const float* in, float* out, int64_t n_maps, int32_t h, int32_t w, int32_t axis,
                                 int32_t out_size, const int32_t* bounds, const double* coeffs, int32_t ksize,
                                 spr_stream_t stream) {
  using namespace spr;
  if (n_maps < 0 || h < 1 || w < 1 || out_size < 1 || ksize < 1 || (axis != 0 && axis != 1)) {
    set_error("spr_resample_axis: bad sizes / axis");
    return SPR_ERR_ARG;
  }
  if (n_maps == 0) return SPR_OK;
  if (!in || !out || !bounds || !coeffs) { set_error("spr_resample_axis: null pointer"); return SPR_ERR_ARG; }
  const long long total = n_maps * static_cast<long long>(axis == 0 ? out_size : h) * (axis == 1 ? out_size : w);
  hipLaunchKernelGGL(resample_kernel, dim3(grid_for(total)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), in, out,
                     static_cast<long long>(n_maps), h, w, axis, out_size, bounds, coeffs, ksize);
  return check_launch("resample_kernel");
}
