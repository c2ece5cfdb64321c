// 8-bit RGB <-> CIE L*a*b* for the reference's RGB route: CLAHE of a colour image runs on the L channel
// (network.py:199-204: cv2.cvtColor(img, COLOR_RGB2LAB) -> split -> clahe.apply(L) -> merge -> COLOR_LAB2RGB).
//
// OpenCV is not importable offline and the reference holds no fixtures, so these kernels restate the PUBLISHED 8-bit
// convention (L in [0,255] = L* * 255/100, a and b offset by 128, sRGB primaries, D65) - parity with cv2 itself is
// UNPINNED; what is pinned is bit-for-bit agreement with oracle/color_oracle.py:
//   forward  fixed-point, table-driven (the scheme of OpenCV's 8-bit path): gamma table (256 x u16, 3 extra bits),
//            XYZ by a 12-bit integer matrix, f(t) from a 3072-entry table (15 bits), integer L / a / b with rounding;
//   inverse  float32 with one rounding per operation (contraction off) and a 4096-entry table for the sRGB transfer
//            function.
// Tables come from the caller (host numpy, float64 -> integers): one source for the kernels and the oracle.
// Elementwise, HBM-bound: 6 bytes per pixel.
#include "spr_common.h"

namespace spr {
namespace {

struct ColorTables {  // layout of the caller's table buffer (int32 each)
  int gamma[256];     // 255 * 8 * srgb_to_linear(i / 255), rounded
  int cbrt[3072];     // 32768 * f(i / (255 * 8)), rounded
  int coeff[9];       // 4096 * M[i][j] / white[i], rounded
  int inv_gamma[4096];  // 255 * linear_to_srgb(i / 4095), rounded
};

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ void __launch_bounds__(kThreads)
rgb2lab_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long long n, const ColorTables* __restrict__ t) {
  const long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  const int R = t->gamma[src[3 * i]], G = t->gamma[src[3 * i + 1]], B = t->gamma[src[3 * i + 2]];
  const int* c = t->coeff;
  const int fx = t->cbrt[descale(R * c[0] + G * c[1] + B * c[2], 12)];
  const int fy = t->cbrt[descale(R * c[3] + G * c[4] + B * c[5], 12)];
  const int fz = t->cbrt[descale(R * c[6] + G * c[7] + B * c[8], 12)];
  constexpr int kLscale = (116 * 255 + 50) / 100;
  constexpr int kLshift = -((16 * 255 * (1 << 15) + 50) / 100);
  dst[3 * i] = static_cast<uint8_t>(sat8(descale(kLscale * fy + kLshift, 15)));
  dst[3 * i + 1] = static_cast<uint8_t>(sat8(descale(500 * (fx - fy) + 128 * (1 << 15), 15)));
  dst[3 * i + 2] = static_cast<uint8_t>(sat8(descale(200 * (fy - fz) + 128 * (1 << 15), 15)));
}

// one rounding per operation (contraction off): the oracle computes the same sequence in numpy float32
__device__ __forceinline__ float mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float finv(float t) {  // inverse of f: t^3 above 6/29, the linear branch below
  return t > 0.20689655f ? mul(mul(t, t), t) : add(t, -0.13793103f) / 7.787f;
}

__global__ void __launch_bounds__(kThreads)
lab2rgb_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long long n, const ColorTables* __restrict__ t) {
#pragma clang fp contract(off)
  const long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  const float L = mul(static_cast<float>(src[3 * i]), 100.0f) / 255.0f;
  const float a = static_cast<float>(static_cast<int>(src[3 * i + 1]) - 128);
  const float b = static_cast<float>(static_cast<int>(src[3 * i + 2]) - 128);
  const float fy = add(L, 16.0f) / 116.0f;
  const float fx = add(fy, a / 500.0f);
  const float fz = add(fy, -(b / 200.0f));
  const float X = mul(0.950456f, finv(fx)), Y = finv(fy), Z = mul(1.088754f, finv(fz));
  const float lin[3] = {add(add(mul(3.240479f, X), mul(-1.53715f, Y)), mul(-0.498535f, Z)),
                        add(add(mul(-0.969256f, X), mul(1.875991f, Y)), mul(0.041556f, Z)),
                        add(add(mul(0.055648f, X), mul(-0.204043f, Y)), mul(1.057311f, Z))};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = lin[k] < 0.0f ? 0.0f : (lin[k] > 1.0f ? 1.0f : lin[k]);
    const int idx = static_cast<int>(add(mul(v, 4095.0f), 0.5f));  // round half up (v >= 0)
    dst[3 * i + k] = static_cast<uint8_t>(t->inv_gamma[idx > 4095 ? 4095 : idx]);
  }
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" size_t spr_color_tables_bytes(void) { return sizeof(ColorTables); }

static int color_launch(bool forward, const uint8_t* src, uint8_t* dst, int64_t n_pixels, const void* tables,
                        spr_stream_t stream, const char* who) {
  if (n_pixels < 0) { set_error("%s: bad size", who); return SPR_ERR_ARG; }
  if (n_pixels == 0) return SPR_OK;
  if (!src || !dst || !tables) { set_error("%s: null pointer", who); return SPR_ERR_ARG; }
  const long long blocks = (n_pixels + kThreads - 1) / kThreads;
  if (blocks > 0x7fffffffLL) { set_error("%s: too many pixels for one call", who); return SPR_ERR_ARG; }
  if (forward)
    hipLaunchKernelGGL(rgb2lab_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       src, dst, static_cast<long long>(n_pixels), static_cast<const ColorTables*>(tables));
  else
    hipLaunchKernelGGL(lab2rgb_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       src, dst, static_cast<long long>(n_pixels), static_cast<const ColorTables*>(tables));
  return check_launch(who);
}

extern "C" int spr_rgb_to_lab_u8(const uint8_t* rgb, uint8_t* lab, int64_t n_pixels, const void* tables, spr_stream_t stream) {
  return color_launch(true, rgb, lab, n_pixels, tables, stream, "spr_rgb_to_lab_u8");
}
extern "C" int spr_lab_to_rgb_u8(const uint8_t* lab, uint8_t* rgb, int64_t n_pixels, const void* tables, spr_stream_t stream) {
  return color_launch(false, lab, rgb, n_pixels, tables, stream, "spr_lab_to_rgb_u8");
}
