// Per-channel preparation shared by the direct and FFT methods (one workgroup = one channel of
// one item):  crop + centre (similarity.py:92-93, 48-49), template energy (:67) and the float64
// window statistics that give the 1/sigma map of a search image (:57-65).
#pragma once
#include "spr_common.h"

namespace spr {

constexpr int kMaxPixPerThread = 48;  // two-sweep 1/sigma path: cropped maps of up to 48 pixels per work-item

// All helpers here run on whatever workgroup size the kernel was launched with (a multiple of 64).
__device__ __forceinline__ int wg_size() { return static_cast<int>(blockDim.x); }

// x0[y*w + x] = crop(map)[y][x] - mean(crop(map)), float32 arithmetic on a float64-accumulated mean.
// Returns after a workgroup barrier.
__device__ __forceinline__ void load_centred(const void* maps, size_t chan_base, int raw_w, int crop, int h, int w,
                                             int dtype, float* x0, double* red, float* mean_out = nullptr) {
  const int tid = static_cast<int>(threadIdx.x);
  const int n = h * w;
  double s = 0.0;
  // (y, x) of pixel i = tid + k*wg_size() advance incrementally: one division per lane, not per pixel.
  // Eight loads are requested before the first is used, so a lane waits for HBM/L2 once per batch, not per pixel.
  const int nthr = wg_size();
  const int dy = nthr / w, dx = nthr - dy * w;
  int y = tid / w, x = tid - y * w;
  constexpr int B = 8;
  for (int i0 = tid; i0 < n; i0 += B * nthr) {
    float v[B];
    int yy = y, xx = x;
#pragma unroll
    for (int k = 0; k < B; ++k) {
      v[k] = i0 + k * nthr < n
                 ? load_feature(maps, chan_base + static_cast<size_t>(yy + crop) * raw_w + (xx + crop), dtype)
                 : 0.0f;
      xx += dx; yy += dy;
      if (xx >= w) { xx -= w; ++yy; }
    }
#pragma unroll
    for (int k = 0; k < B; ++k) {
      if (i0 + k * nthr < n) {
        x0[i0 + k * nthr] = v[k];
        s += static_cast<double>(v[k]);
      }
    }
    y = yy; x = xx;
  }
  const double total = block_sum(s, red);
  const float mean = static_cast<float>(total / static_cast<double>(n));
  for (int i = tid; i < n; i += wg_size()) x0[i] = x0[i] - mean;
  if (mean_out) *mean_out = mean;
  __syncthreads();
}

// 1/sqrt(sum x0^2) as float (0 for an all-zero map: the reference's 0/0 -> NaN -> 0, :68-70).
__device__ __forceinline__ float template_scale(const float* x0, int n, double* red) {
  const int tid = static_cast<int>(threadIdx.x);
  double s = 0.0;
  for (int i = tid; i < n; i += wg_size()) {
    const float v = x0[i];
    const float sq = v * v;  // np.square keeps float32 (:67)
    s += static_cast<double>(sq);
  }
  const float energy = static_cast<float>(block_sum(s, red));
  return energy > 0.0f ? static_cast<float>(1.0 / sqrt(static_cast<double>(energy))) : 0.0f;
}

// Summed-area table of x0 (or of fl32(x0^2)) in float64: sat[(y)*(w+1) + x] = sum of rows < y, cols < x.
// One lane per row, then one lane per column; the serial scans run in register batches of 8 so that the
// LDS round trip is paid once per batch and only the float64 adds form the dependent chain.
__device__ __forceinline__ void build_sat(const float* __restrict__ x0, int h, int w, double* __restrict__ sat,
                                          bool squared) {
  const int tid = static_cast<int>(threadIdx.x);
  const int stride = w + 1;
  constexpr int B = 8;
  for (int x = tid; x <= w; x += wg_size()) sat[x] = 0.0;
  for (int y = tid; y < h; y += wg_size()) {
    double run = 0.0;
    double* row = sat + static_cast<size_t>(y + 1) * stride;
    const float* src = x0 + y * w;
    row[0] = 0.0;
    for (int xb = 0; xb < w; xb += B) {
      float v[B];
#pragma unroll
      for (int k = 0; k < B; ++k) v[k] = xb + k < w ? src[xb + k] : 0.0f;
#pragma unroll
      for (int k = 0; k < B; ++k) {
        if (squared) {
          const float sq = v[k] * v[k];
          run += static_cast<double>(sq);
        } else {
          run += static_cast<double>(v[k]);
        }
        if (xb + k < w) row[xb + k + 1] = run;
      }
    }
  }
  __syncthreads();
  for (int x = tid; x <= w; x += wg_size()) {
    double run = 0.0;
    for (int yb = 1; yb <= h; yb += B) {
      double v[B];
#pragma unroll
      for (int k = 0; k < B; ++k) v[k] = yb + k <= h ? sat[static_cast<size_t>(yb + k) * stride + x] : 0.0;
#pragma unroll
      for (int k = 0; k < B; ++k) {
        run += v[k];
        if (yb + k <= h) sat[static_cast<size_t>(yb + k) * stride + x] = run;
      }
    }
  }
  __syncthreads();
}

// Sum over the th x tw window that 'same'-mode correlation places at output pixel (y, x),
// clipped to the h x w image (zero padding contributes nothing).
__device__ __forceinline__ double window_sum(const double* sat, int h, int w, int th, int tw, int y, int x) {
  const int stride = w + 1;
  int y0 = y - th / 2, y1 = y0 + th, x0 = x - tw / 2, x1 = x0 + tw;
  y0 = y0 < 0 ? 0 : (y0 > h ? h : y0);
  y1 = y1 < 0 ? 0 : (y1 > h ? h : y1);
  x0 = x0 < 0 ? 0 : (x0 > w ? w : x0);
  x1 = x1 < 0 ? 0 : (x1 > w ? w : x1);
  return sat[static_cast<size_t>(y1) * stride + x1] - sat[static_cast<size_t>(y0) * stride + x1] -
         sat[static_cast<size_t>(y1) * stride + x0] + sat[static_cast<size_t>(y0) * stride + x0];
}

// Both tables in one sweep (two running sums per scan): sat1 of x0, sat2 of fl32(x0^2).
__device__ __forceinline__ void build_sat_pair(const float* __restrict__ x0, int h, int w, double* __restrict__ sat1,
                                               double* __restrict__ sat2) {
  const int tid = static_cast<int>(threadIdx.x);
  const int stride = w + 1;
  constexpr int B = 8;
  for (int x = tid; x <= w; x += wg_size()) { sat1[x] = 0.0; sat2[x] = 0.0; }
  for (int y = tid; y < h; y += wg_size()) {
    double r1 = 0.0, r2 = 0.0;
    double* row1 = sat1 + static_cast<size_t>(y + 1) * stride;
    double* row2 = sat2 + static_cast<size_t>(y + 1) * stride;
    const float* src = x0 + y * w;
    row1[0] = 0.0; row2[0] = 0.0;
    for (int xb = 0; xb < w; xb += B) {
      float v[B];
#pragma unroll
      for (int k = 0; k < B; ++k) v[k] = xb + k < w ? src[xb + k] : 0.0f;
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const float sq = v[k] * v[k];  // np.square keeps float32 (similarity.py:57)
        r1 += static_cast<double>(v[k]);
        r2 += static_cast<double>(sq);
        if (xb + k < w) { row1[xb + k + 1] = r1; row2[xb + k + 1] = r2; }
      }
    }
  }
  __syncthreads();
  // column scans: lanes [0, w] take table 1, lanes [w+1, 2w+1] table 2
  for (int j = tid; j < 2 * stride; j += wg_size()) {
    double* sat = j < stride ? sat1 : sat2;
    const int x = j < stride ? j : j - stride;
    double run = 0.0;
    for (int yb = 1; yb <= h; yb += B) {
      double v[B];
#pragma unroll
      for (int k = 0; k < B; ++k) v[k] = yb + k <= h ? sat[static_cast<size_t>(yb + k) * stride + x] : 0.0;
#pragma unroll
      for (int k = 0; k < B; ++k) {
        run += v[k];
        if (yb + k <= h) sat[static_cast<size_t>(yb + k) * stride + x] = run;
      }
    }
  }
  __syncthreads();
}

// 1/sigma from the two window sums: var = S2 - S1^2/(th*tw) in float64 (that is where the cancellation is),
// var <= 0 -> 0 (similarity.py:65, :70); the reciprocal square root of the positive result only needs float32
// accuracy (hardware v_rsq_f32, ~1 ulp: relative error ~2e-7).
__device__ __forceinline__ float inv_sigma_from_sums(double s1, double s2, double inv_n) {
  const double var = s2 - s1 * s1 * inv_n;
  float inv = 0.0f;
  if (var > 0.0) {
    inv = rsqrtf(static_cast<float>(var));
    if (!(inv <= 3.0e38f)) inv = 0.0f;  // variance below the float32 range: treat as non-finite -> 0
  }
  return inv;
}

// Single-sweep form of inv_sigma_map for maps whose two tables fit LDS together (2*(h+1)*(w+1) doubles).
// `store(y, x, value)`.
template <class Store>
__device__ __forceinline__ void inv_sigma_map_fused(const float* x0, int h, int w, int th, int tw, double* sat1,
                                                    double* sat2, Store store) {
  const int tid = static_cast<int>(threadIdx.x);
  const int n = h * w;
  build_sat_pair(x0, h, w, sat1, sat2);
  const double inv_n = 1.0 / (static_cast<double>(th) * static_cast<double>(tw));
  const int stride = w + 1;
  const int dy = wg_size() / w, dx = wg_size() - dy * w;
  int y = tid / w, x = tid - y * w;
  for (int i = tid; i < n; i += wg_size()) {
    int y0 = y - th / 2, y1 = y0 + th, xa = x - tw / 2, xb = xa + tw;
    y0 = y0 < 0 ? 0 : y0;  y1 = y1 > h ? h : y1;  // (y0 <= h and y1 >= 0 always: 0 <= y < h, th >= 1)
    xa = xa < 0 ? 0 : xa;  xb = xb > w ? w : xb;
    y0 = y0 > h ? h : y0;  xa = xa > w ? w : xa;  y1 = y1 < 0 ? 0 : y1;  xb = xb < 0 ? 0 : xb;
    const int i11 = y1 * stride + xb, i01 = y0 * stride + xb, i10 = y1 * stride + xa, i00 = y0 * stride + xa;
    const double s1 = sat1[i11] - sat1[i01] - sat1[i10] + sat1[i00];
    const double s2 = sat2[i11] - sat2[i01] - sat2[i10] + sat2[i00];
    store(y, x, inv_sigma_from_sums(s1, s2, inv_n));
    x += dx; y += dy;
    if (x >= w) { x -= w; ++y; }
  }
  __syncthreads();
}

// inv_sigma for every pixel of the centred map x0 (h x w), for a th x tw template:
// var = S2 - S1^2/(th*tw) in float64, var <= 0 -> 0 (the reference clamps negatives to 0 and
// turns the resulting division by zero into 0, :65, :70).  `store(i, value)` receives pixel
// i = y*w + x.  `sat` needs (h+1)*(w+1) doubles of LDS.
template <class Store>
__device__ __forceinline__ void inv_sigma_map(const float* x0, int h, int w, int th, int tw, double* sat, Store store) {
  const int tid = static_cast<int>(threadIdx.x);
  const int n = h * w;
  double s1[kMaxPixPerThread];
  build_sat(x0, h, w, sat, false);
#pragma unroll
  for (int k = 0; k < kMaxPixPerThread; ++k) {
    const int i = tid + k * wg_size();
    s1[k] = 0.0;
    if (i < n) {
      const int y = i / w, x = i - y * w;
      s1[k] = window_sum(sat, h, w, th, tw, y, x);
    }
  }
  __syncthreads();
  build_sat(x0, h, w, sat, true);
  const double inv_n = 1.0 / (static_cast<double>(th) * static_cast<double>(tw));
#pragma unroll
  for (int k = 0; k < kMaxPixPerThread; ++k) {
    const int i = tid + k * wg_size();
    if (i < n) {
      const int y = i / w, x = i - y * w;
      const double s2 = window_sum(sat, h, w, th, tw, y, x);
      store(i, inv_sigma_from_sums(s1[k], s2, inv_n));
    }
  }
  __syncthreads();
}

}  // namespace spr
