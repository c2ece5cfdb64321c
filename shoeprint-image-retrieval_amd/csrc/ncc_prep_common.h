// Per-channel preparation shared by the direct and FFT methods (one workgroup = one channel of
// one item):  crop + centre (similarity.py:92-93, 48-49), template energy (:67) and the float64
// window statistics that give the 1/sigma map of a search image (:57-65).
#pragma once
#include "spr_common.h"

namespace spr {

// -DSPR_PREP_STAMPS (ncc_fft.hip only): diagnostic build for tools/ubench/stamps_prep.py - one workgroup of the FFT prep
// kernel records the shader clock at its phase boundaries
#ifdef SPR_PREP_STAMPS
__device__ unsigned long long g_prep_stamps[16];
#define SPR_PSTAMP(i) do { if (blockIdx.x == 100 && blockIdx.y == 700 && threadIdx.x == 0) g_prep_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SPR_PSTAMP(i)
#endif

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
  // Sixteen loads are requested before the first is used, so a lane waits for HBM/L2 once per batch, not per pixel
  // (a conv3_3 map on 512 work-items is one batch).
  const int nthr = wg_size();
  const int dy = nthr / w, dx = nthr - dy * w;
  int y = tid / w, x = tid - y * w;
  constexpr int B = 16;
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

// The same two tables with every scan cut into kSatSeg segments: a row (column) is scanned by kSatSeg work-items, each from
// zero over its own stretch, then the totals of the stretches before it are added - four times as many work-items on chains
// a quarter as long (the serial form keeps h (2w + 2) work-items of the workgroup busy, the rest wait at the barrier).
// The float64 sums are formed in another order than build_sat_pair's: differences of a few ulp of float64.
constexpr int kSatSeg = 4;
constexpr int kSatRowSeg = 16;  // longest row stretch the register batches below hold (w <= 64)
__device__ __forceinline__ bool sat_blocked_fits(int h, int w) {
  return h * kSatSeg <= 2 * wg_size() && 2 * kSatSeg * w <= 4 * wg_size() && (w + kSatSeg - 1) / kSatSeg <= kSatRowSeg;
}
__device__ __forceinline__ void build_sat_pair_blocked(const float* __restrict__ x0, int h, int w, double* __restrict__ sat1,
                                                       double* __restrict__ sat2) {
  // No element is predicated: a batch slot beyond the end of a stretch re-reads and re-writes the stretch's LAST element
  // (its addend is 0, so the value written again is the same), which costs far fewer instructions than masking it.
  const int tid = static_cast<int>(threadIdx.x), nthr = wg_size();
  const int stride = w + 1;
  const int wseg = (w + kSatSeg - 1) / kSatSeg, hseg = (h + kSatSeg - 1) / kSatSeg;
  for (int x = tid; x <= w; x += nthr) { sat1[x] = 0.0; sat2[x] = 0.0; }
  // ---- rows: unit (y, stretch); every LDS round trip is paid once per register batch, not once per element ----
  for (int u = tid; u < h * kSatSeg; u += nthr) {
    const int y = u / kSatSeg, sg = u - y * kSatSeg;
    const int xa = sg * wseg, last = (xa + wseg < w ? xa + wseg : w) - xa - 1;
    if (last < 0) continue;  // narrow maps: a stretch past the last column
    double* row1 = sat1 + (y + 1) * stride + xa + 1;
    double* row2 = sat2 + (y + 1) * stride + xa + 1;
    const float* src = x0 + y * w + xa;
    float v[kSatRowSeg];
#pragma unroll
    for (int k = 0; k < kSatRowSeg; ++k) {
      const float t = src[k < last ? k : last];
      v[k] = k <= last ? t : 0.0f;
    }
    if (sg == 0) { row1[-1] = 0.0; row2[-1] = 0.0; }
    double r1 = 0.0, r2 = 0.0;
#pragma unroll
    for (int k = 0; k < kSatRowSeg; ++k) {
      const float sq = v[k] * v[k];  // np.square keeps float32 (similarity.py:57)
      r1 += static_cast<double>(v[k]);
      r2 += static_cast<double>(sq);
      const int kk = k < last ? k : last;
      row1[kk] = r1; row2[kk] = r2;
    }
  }
  __syncthreads();
  {
    // totals of the stretches before this one (read by everybody before anybody adds: two barriers)
    double o1[2], o2[2];  // a workgroup of 256 covers h * kSatSeg <= 512 units in two turns
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int u = tid + k * nthr < h * kSatSeg ? tid + k * nthr : h * kSatSeg - 1;
      const int y = u / kSatSeg, sg = u - y * kSatSeg;
      const double* row1 = sat1 + (y + 1) * stride;
      const double* row2 = sat2 + (y + 1) * stride;
      o1[k] = 0.0; o2[k] = 0.0;
#pragma unroll
      for (int j = 0; j < kSatSeg - 1; ++j) {
        const int e = (j + 1) * wseg < w ? (j + 1) * wseg : w;
        const double a1 = row1[e], a2 = row2[e];
        o1[k] += j < sg ? a1 : 0.0;
        o2[k] += j < sg ? a2 : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int u = tid + k * nthr;
      if (u >= h * kSatSeg) continue;
      const int y = u / kSatSeg, sg = u - y * kSatSeg;
      if (sg == 0) continue;
      const int xa = sg * wseg, last = (xa + wseg < w ? xa + wseg : w) - xa - 1;
      if (last < 0) continue;
      double* row1 = sat1 + (y + 1) * stride + xa + 1;
      double* row2 = sat2 + (y + 1) * stride + xa + 1;
      double a1[kSatRowSeg], a2[kSatRowSeg];
#pragma unroll
      for (int i = 0; i < kSatRowSeg; ++i) { a1[i] = row1[i < last ? i : last]; a2[i] = row2[i < last ? i : last]; }
#pragma unroll
      for (int i = 0; i < kSatRowSeg; ++i) { row1[i < last ? i : last] = a1[i] + o1[k]; row2[i < last ? i : last] = a2[i] + o2[k]; }
    }
  }
  __syncthreads();
  SPR_PSTAMP(7);
  // ---- columns 1..w of both tables: unit (table, stretch, column), consecutive work-items on consecutive columns ----
  const int units = 2 * kSatSeg * w;
  constexpr int B = 8;
  for (int u = tid; u < units; u += nthr) {
    const int rest = u / w, x = u - rest * w + 1, sg = rest % kSatSeg;
    const int ya = sg * hseg, n = (ya + hseg < h ? ya + hseg : h) - ya;  // rows ya + 1 .. ya + n of the table
    double* col = (rest / kSatSeg ? sat2 : sat1) + x + (ya + 1) * stride;
    double run = 0.0;
    for (int r0 = 0; r0 < n; r0 += B) {
      double v[B];
      double* at[B];  // walks down the column, standing still on its last row (no multiply per element)
      at[0] = col;
#pragma unroll
      for (int k = 1; k < B; ++k) at[k] = at[k - 1] + (r0 + k < n ? stride : 0);
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const double t = *at[k];
        v[k] = r0 + k < n ? t : 0.0;
      }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        run += v[k];
        *at[k] = run;
      }
      col = at[B - 1] + stride;
    }
  }
  __syncthreads();
  double off[4];  // up to four turns (256 work-items, w <= 128)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int u = tid + k * nthr < units ? tid + k * nthr : units - 1;
    const int rest = u / w, x = u - rest * w + 1, sg = rest % kSatSeg;
    const double* col = (rest / kSatSeg ? sat2 : sat1) + x;
    off[k] = 0.0;
#pragma unroll
    for (int j = 0; j < kSatSeg - 1; ++j) {
      const int e = (j + 1) * hseg < h ? (j + 1) * hseg : h;
      const double a = col[e * stride];
      off[k] += j < sg ? a : 0.0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int u = tid + k * nthr;
    if (u >= units) continue;
    const int rest = u / w, x = u - rest * w + 1, sg = rest % kSatSeg;
    if (sg == 0) continue;
    const int ya = sg * hseg, n = (ya + hseg < h ? ya + hseg : h) - ya;
    double* col = (rest / kSatSeg ? sat2 : sat1) + x + (ya + 1) * stride;
    for (int r0 = 0; r0 < n; r0 += B) {
      double v[B];
      double* at[B];
      at[0] = col;
#pragma unroll
      for (int i = 1; i < B; ++i) at[i] = at[i - 1] + (r0 + i < n ? stride : 0);
#pragma unroll
      for (int i = 0; i < B; ++i) v[i] = *at[i];
#pragma unroll
      for (int i = 0; i < B; ++i) *at[i] = v[i] + off[k];
      col = at[B - 1] + stride;
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
  if (sat_blocked_fits(h, w))
    build_sat_pair_blocked(x0, h, w, sat1, sat2);
  else
    build_sat_pair(x0, h, w, sat1, sat2);
  SPR_PSTAMP(8);
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
