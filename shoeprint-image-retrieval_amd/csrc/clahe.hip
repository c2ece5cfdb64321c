// CLAHE on uint8 images (reference network.py:108-111, 197-208: cv2.createCLAHE(clipLimit, tileGridSize).apply),
// restating OpenCV's published 8-bit algorithm (modules/imgproc/src/clahe.cpp) — see oracle/clahe_oracle.py
// for the same algorithm in numpy; both are unpinned by the reference (no cv2 offline).
//
//   clahe_lut_kernel     one workgroup per (tile, image): 256-bin histogram in LDS (integer atomics), clip at
//                        max(1, int(clipLimit*tileArea/256)), redistribute the excess (uniform part + strided
//                        residual), inclusive scan, LUT = saturate(round_half_even(cdf * 255/tileArea)).
//                        Images whose size is not a multiple of the grid are extended by reflect-101.
//   clahe_interp_kernel  one lane per pixel: bilinear blend of the four neighbouring tile LUTs in float32
//                        (contraction off: same roundings as the numpy / OpenCV expression order).
// Integer work + a 4-tap gather per pixel: HBM-bound (1 byte read + 1 written per pixel, LUTs stay in L2).
#include "spr_common.h"

namespace spr {
namespace {

__device__ __forceinline__ int reflect101(int i, int n) {  // BORDER_REFLECT_101 for i in [n, 2n-2]
  return i < n ? i : 2 * n - 2 - i;
}

// grid = (tiles_x * tiles_y, n)
__global__ void __launch_bounds__(kThreads)
clahe_lut_kernel(const uint8_t* __restrict__ in, int h, int w, int tiles_x, int tiles_y, int th, int tw, int clip,
                 float lut_scale, uint8_t* __restrict__ luts) {
  __shared__ int hist[256];
  __shared__ int scan[256];
  __shared__ int excess_total;
  const int tid = static_cast<int>(threadIdx.x);
  const int ty = static_cast<int>(blockIdx.x) / tiles_x, tx = static_cast<int>(blockIdx.x) % tiles_x;
  const size_t img = blockIdx.y;
  hist[tid] = 0;
  if (tid == 0) excess_total = 0;
  __syncthreads();
  const uint8_t* src = in + img * static_cast<size_t>(h) * w;
  for (int i = tid; i < th * tw; i += kThreads) {
    const int yy = ty * th + i / tw, xx = tx * tw + i % tw;
    atomicAdd(&hist[src[static_cast<size_t>(reflect101(yy, h)) * w + reflect101(xx, w)]], 1);
  }
  __syncthreads();
  int v = hist[tid];
  if (clip > 0) {
    const int over = v > clip ? v - clip : 0;
    if (over) atomicAdd(&excess_total, over);
    v = v > clip ? clip : v;
    __syncthreads();
    const int excess = excess_total;
    const int batch = excess / 256, residual = excess - batch * 256;
    v += batch;
    if (residual) {
      const int step = 256 / residual > 1 ? 256 / residual : 1;
      if (tid % step == 0 && tid / step < residual) v += 1;
    }
  }
  // inclusive scan over the 256 bins (Hillis-Steele in LDS)
  scan[tid] = v;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int add = tid >= off ? scan[tid - off] : 0;
    __syncthreads();
    scan[tid] += add;
    __syncthreads();
  }
  const float f = rintf(static_cast<float>(scan[tid]) * lut_scale);  // cvRound: half to even
  const int q = f < 0.0f ? 0 : (f > 255.0f ? 255 : static_cast<int>(f));
  luts[((img * tiles_y + ty) * tiles_x + tx) * 256 + tid] = static_cast<uint8_t>(q);
}

__global__ void __launch_bounds__(kThreads)
clahe_interp_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long long n_pixels, int h, int w,
                    int tiles_x, int tiles_y, float inv_th, float inv_tw, const uint8_t* __restrict__ luts) {
#pragma clang fp contract(off)
  const long long per = static_cast<long long>(h) * w;
  for (long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x; i < n_pixels;
       i += static_cast<long long>(gridDim.x) * kThreads) {
    const long long img = i / per;
    const int rem = static_cast<int>(i - img * per);
    const int y = rem / w, x = rem - y * w;
    const float tyf = static_cast<float>(y) * inv_th - 0.5f, txf = static_cast<float>(x) * inv_tw - 0.5f;
    int ty1 = static_cast<int>(floorf(tyf)), tx1 = static_cast<int>(floorf(txf));
    const float ya = tyf - static_cast<float>(ty1), xa = txf - static_cast<float>(tx1);
    int ty2 = ty1 + 1 < tiles_y - 1 ? ty1 + 1 : tiles_y - 1;
    int tx2 = tx1 + 1 < tiles_x - 1 ? tx1 + 1 : tiles_x - 1;
    ty1 = ty1 > 0 ? ty1 : 0;
    tx1 = tx1 > 0 ? tx1 : 0;
    const int v = in[i];
    const uint8_t* base = luts + img * static_cast<size_t>(tiles_y) * tiles_x * 256;
    const float l11 = base[(ty1 * tiles_x + tx1) * 256 + v], l12 = base[(ty1 * tiles_x + tx2) * 256 + v];
    const float l21 = base[(ty2 * tiles_x + tx1) * 256 + v], l22 = base[(ty2 * tiles_x + tx2) * 256 + v];
    const float one = 1.0f;
    const float top = l11 * (one - xa) + l12 * xa, bot = l21 * (one - xa) + l22 * xa;
    const float res = rintf(top * (one - ya) + bot * ya);
    out[i] = static_cast<uint8_t>(res < 0.0f ? 0 : (res > 255.0f ? 255 : static_cast<int>(res)));
  }
}

}  // namespace
}  // namespace spr

extern "C" size_t spr_clahe_workspace_bytes(int64_t n, int32_t tiles_x, int32_t tiles_y) {
  if (n < 0 || tiles_x < 1 || tiles_y < 1) return 0;
  return spr::align_up(static_cast<size_t>(n) * tiles_x * tiles_y * 256, 256);
}

extern "C" int spr_clahe_u8(const uint8_t* in, uint8_t* out, int64_t n, int32_t h, int32_t w, float clip_limit,
                            int32_t tiles_x, int32_t tiles_y, void* workspace, spr_stream_t stream) {
  using namespace spr;
  if (n < 0 || n > 65535 || h < 1 || w < 1 || tiles_x < 1 || tiles_y < 1 || tiles_x > w || tiles_y > h) {
    set_error("spr_clahe_u8: bad sizes");
    return SPR_ERR_ARG;
  }
  if (n == 0) return SPR_OK;
  if (!in || !out || !workspace) { set_error("spr_clahe_u8: null pointer"); return SPR_ERR_ARG; }
  // OpenCV extends BOTH dimensions by tiles - size % tiles as soon as EITHER is not a multiple of the grid
  // (copyMakeBorder(src, ext, 0, tilesY - h % tilesY, 0, tilesX - w % tilesX, REFLECT_101)): a divisible
  // dimension then grows by one whole tile row / column.
  const bool divisible = h % tiles_y == 0 && w % tiles_x == 0;
  const int eh = divisible ? h : h + tiles_y - h % tiles_y, ew = divisible ? w : w + tiles_x - w % tiles_x;
  if (eh > 2 * h - 1 || ew > 2 * w - 1) { set_error("spr_clahe_u8: image smaller than the reflect-101 extension"); return SPR_ERR_SHAPE; }
  const int th = eh / tiles_y, tw = ew / tiles_x, area = th * tw;
  int clip = 0;
  if (clip_limit > 0.0f) {
    clip = static_cast<int>(static_cast<double>(clip_limit) * area / 256);
    if (clip < 1) clip = 1;
  }
  const float lut_scale = 255.0f / static_cast<float>(area);
  const float inv_th = static_cast<float>(1.0 / th), inv_tw = static_cast<float>(1.0 / tw);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(clahe_lut_kernel, dim3(tiles_x * tiles_y, static_cast<unsigned>(n)), dim3(kThreads), 0, s, in, h, w,
                     tiles_x, tiles_y, th, tw, clip, lut_scale, static_cast<uint8_t*>(workspace));
  int rc = check_launch("clahe_lut_kernel");
  if (rc != SPR_OK) return rc;
  const long long n_pixels = static_cast<long long>(n) * h * w;
  long long blocks = (n_pixels + kThreads - 1) / kThreads;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(clahe_interp_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s, in, out, n_pixels,
                     h, w, tiles_x, tiles_y, inv_th, inv_tw, static_cast<const uint8_t*>(workspace));
  return check_launch("clahe_interp_kernel");
}
