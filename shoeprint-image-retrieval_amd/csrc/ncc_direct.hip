// Direct (sliding-window) NCC scorer: the shape-agnostic method.
//
// One workgroup scores one (query, gallery) pair, channel by channel (similarity.py:100-108):
// the zero-padded, centred search map and the normalised template of the current channel live in
// LDS; every lane owns register strips of kStrip output pixels and walks the template, so each
// template tap is one LDS broadcast and 2*kStrip-1 search-map reads feed kStrip*kStrip FMAs.
// The channel sum stays in registers; one wave/LDS max-reduction at the end gives the score.
//
// This is the simple path: arithmetic = th*tw*pad... multiply-adds per output pixel including
// the zero padding.  The FFT method (ncc_pair_fft.hip) is the fast path; this one covers every
// shape that fits LDS and cross-checks the FFT path on the GPU.
#include "ncc_prep_common.h"

namespace spr {

namespace {

__host__ __device__ inline int round_up8(int v) { return (v + 7) / 8 * 8; }

// LDS layouts ------------------------------------------------------------------------------
struct PrepLds {
  size_t red_off, x0_off, sat_off, total;
};
inline PrepLds prep_lds(int h, int w, bool need_sat) {
  PrepLds l;
  l.red_off = 0;
  l.x0_off = 64;
  l.sat_off = align_up(l.x0_off + sizeof(float) * h * w, 16);
  l.total = l.sat_off + (need_sat ? sizeof(double) * (h + 1) * (w + 1) : 0);
  return l;
}

// grid = (channels, n_items)
__global__ void __launch_bounds__(kThreads)
prep_direct_kernel(NccGeom g, int is_query, const void* __restrict__ maps, float* __restrict__ prepared,
                   size_t item_floats, unsigned x0_off, unsigned sat_off) {
  unsigned char* lds = dyn_lds();
  double* red = reinterpret_cast<double*>(lds);
  float* x0 = reinterpret_cast<float*>(lds + x0_off);
  double* sat = reinterpret_cast<double*>(lds + sat_off);
  const int c = static_cast<int>(blockIdx.x);
  const size_t item = blockIdx.y;
  const int tid = static_cast<int>(threadIdx.x);
  if (is_query) {
    const int n = g.th * g.tw;
    const size_t base = (item * g.channels + c) * static_cast<size_t>(g.q_h) * g.q_w;
    load_centred(maps, base, g.q_w, g.crop, g.th, g.tw, g.dtype, x0, red);
    const float scale = template_scale(x0, n, red);
    float* out = prepared + item * item_floats + static_cast<size_t>(c) * n;
    for (int i = tid; i < n; i += kThreads) out[i] = x0[i] * scale;
  } else {
    const int n = g.ih * g.iw;
    const size_t base = (item * g.channels + c) * static_cast<size_t>(g.g_h) * g.g_w;
    load_centred(maps, base, g.g_w, g.crop, g.ih, g.iw, g.dtype, x0, red);
    float* out = prepared + item * item_floats + static_cast<size_t>(c) * n;
    float* inv = out + static_cast<size_t>(g.channels) * n;
    for (int i = tid; i < n; i += kThreads) out[i] = x0[i];
    inv_sigma_map(x0, g.ih, g.iw, g.th, g.tw, sat, [&](int i, float v) { inv[i] = v; });
  }
}

// grid = (n_gallery, n_queries)
template <int SPT>
__global__ void __launch_bounds__(kThreads)
pair_direct_kernel(NccGeom g, const float* __restrict__ pq, size_t q_item_floats, const float* __restrict__ pg,
                   size_t g_item_floats, float* __restrict__ scores, long long ld, long long col0, int accumulate,
                   float* __restrict__ maps_out, int pws, int tws, unsigned t_off) {
  unsigned char* lds = dyn_lds();
  float* red = reinterpret_cast<float*>(lds);
  float* P = reinterpret_cast<float*>(lds + 64);
  float* T = reinterpret_cast<float*>(lds + t_off);
  const int tid = static_cast<int>(threadIdx.x);
  const size_t gi = blockIdx.x, qi = blockIdx.y;
  const int cy = g.th / 2, cx = g.tw / 2;
  const int n_img = g.ih * g.iw, n_tpl = g.th * g.tw;
  const float* q_base = pq + qi * q_item_floats;
  const float* g_base = pg + gi * g_item_floats;
  const float* inv_base = g_base + static_cast<size_t>(g.channels) * n_img;

  // zero the padded image (borders stay zero for every channel) and the template tail columns
  for (int i = tid; i < g.pad_h * pws; i += kThreads) P[i] = 0.0f;
  for (int i = tid; i < g.th * tws; i += kThreads) T[i] = 0.0f;

  int sy[SPT], sx[SPT];
  float total[SPT][kStrip];
#pragma unroll
  for (int k = 0; k < SPT; ++k) {
    const int s = tid + k * kThreads;
    sy[k] = s / g.strips_per_row;
    sx[k] = (s - sy[k] * g.strips_per_row) * kStrip;
#pragma unroll
    for (int j = 0; j < kStrip; ++j) total[k][j] = 0.0f;
  }

  for (int c = 0; c < g.channels; ++c) {
    __syncthreads();  // previous channel's reads are done (and the initial zero fill is visible)
    const float* img = g_base + static_cast<size_t>(c) * n_img;
    for (int i = tid; i < n_img; i += kThreads) {
      const int y = i / g.iw, x = i - y * g.iw;
      P[(y + cy) * pws + (x + cx)] = img[i];
    }
    const float* tpl = q_base + static_cast<size_t>(c) * n_tpl;
    for (int i = tid; i < n_tpl; i += kThreads) {
      const int u = i / g.tw, v = i - u * g.tw;
      T[u * tws + v] = tpl[i];
    }
    __syncthreads();
    const float* inv = inv_base + static_cast<size_t>(c) * n_img;
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
      if (sy[k] >= g.ih) continue;
      float acc[kStrip];
#pragma unroll
      for (int j = 0; j < kStrip; ++j) acc[j] = 0.0f;
      for (int u = 0; u < g.th; ++u) {
        const float* prow = P + (sy[k] + u) * pws + sx[k];
        const float* trow = T + u * tws;
        for (int v0 = 0; v0 < tws; v0 += kStrip) {
          float seg[2 * kStrip - 1], tv[kStrip];
#pragma unroll
          for (int j = 0; j < 2 * kStrip - 1; ++j) seg[j] = prow[v0 + j];
#pragma unroll
          for (int j = 0; j < kStrip; ++j) tv[j] = trow[v0 + j];
#pragma unroll
          for (int vv = 0; vv < kStrip; ++vv) {
#pragma unroll
            for (int j = 0; j < kStrip; ++j) acc[j] = fmaf(tv[vv], seg[vv + j], acc[j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < kStrip; ++j) {
        const int x = sx[k] + j;
        if (x < g.iw) {
          const float val = acc[j] * inv[sy[k] * g.iw + x];
          total[k][j] += val;
          if (maps_out) maps_out[(static_cast<size_t>(c) * g.ih + sy[k]) * g.iw + x] = val;
        }
      }
    }
  }

  float best = -3.0e38f;
#pragma unroll
  for (int k = 0; k < SPT; ++k) {
#pragma unroll
    for (int j = 0; j < kStrip; ++j) {
      if (sy[k] < g.ih && sx[k] + j < g.iw) best = fmaxf(best, total[k][j]);
    }
  }
  best = block_max(best, red);
  if (tid == 0 && scores) {
    float s = best / static_cast<float>(g.channels);
    float* dst = scores + qi * ld + col0 + gi;
    const float prev = accumulate ? *dst : 0.0f;
    *dst = s > prev ? s : prev;  // similarity.py:355, 366-367: zero-initialised running maximum
  }
}

struct DirectLds {
  int pws, tws;
  size_t t_off, total;
};
inline DirectLds direct_lds(const NccGeom& g) {
  DirectLds l;
  l.tws = round_up8(g.tw);
  l.pws = kStrip * g.strips_per_row + l.tws;
  if (l.pws < g.pad_w) l.pws = g.pad_w;
  l.t_off = align_up(64 + sizeof(float) * static_cast<size_t>(g.pad_h) * l.pws, 16);
  l.total = l.t_off + sizeof(float) * static_cast<size_t>(g.th) * l.tws;
  return l;
}

}  // namespace

bool direct_geometry(NccGeom& g) {
  g.pad_h = g.ih + g.th - 1;
  g.pad_w = g.iw + g.tw - 1;
  g.strips_per_row = ceil_div(g.iw, kStrip);
  const int strips = g.ih * g.strips_per_row;
  g.strips_per_thread = 1;
  while (g.strips_per_thread < ceil_div(strips, kThreads)) g.strips_per_thread *= 2;
  if (g.strips_per_thread > 8) return false;
  if (direct_lds(g).total > static_cast<size_t>(kLdsLimit)) return false;
  if (g.ih * g.iw > kMaxPixPerThread * kThreads || g.th * g.tw > kMaxPixPerThread * kThreads) return false;
  if (prep_lds(g.ih, g.iw, true).total > static_cast<size_t>(kLdsLimit)) return false;
  return true;
}

int launch_prep_direct(const NccGeom& g, bool is_query, const void* maps, int64_t n, void* prepared,
                       hipStream_t stream) {
  if (n == 0) return SPR_OK;
  const PrepLds l = is_query ? prep_lds(g.th, g.tw, false) : prep_lds(g.ih, g.iw, true);
  const size_t item_floats =
      (is_query ? prepared_query_item_bytes(g, SPR_NCC_DIRECT) : prepared_gallery_item_bytes(g, SPR_NCC_DIRECT)) /
      sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(prep_direct_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
  hipLaunchKernelGGL(prep_direct_kernel, dim3(g.channels, static_cast<unsigned>(n)), dim3(kThreads), l.total, stream,
                     g, is_query ? 1 : 0, maps, static_cast<float*>(prepared), item_floats,
                     static_cast<unsigned>(l.x0_off), static_cast<unsigned>(l.sat_off));
  return check_launch("prep_direct_kernel");
}

template <int SPT>
static int launch_pair_direct_t(const NccGeom& g, const void* pq, int64_t nq, const void* pg, int64_t ng,
                                float* scores, int64_t ld, int64_t col0, int accumulate, float* maps_out,
                                hipStream_t stream) {
  const DirectLds l = direct_lds(g);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pair_direct_kernel<SPT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
  // grid = (gallery, query) workgroups; HIP refuses 2^32 work-items and more along x: slices of the gallery
  const size_t g_item_floats = prepared_gallery_item_bytes(g, SPR_NCC_DIRECT) / sizeof(float);
  const int64_t max_g = pair_tiles_per_launch(1, kThreads);
  for (int64_t g0 = 0; g0 < ng; g0 += max_g) {
    const int64_t n = ng - g0 < max_g ? ng - g0 : max_g;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_direct_kernel<SPT>), dim3(static_cast<unsigned>(n), static_cast<unsigned>(nq)),
                       dim3(kThreads), l.total, stream, g, static_cast<const float*>(pq),
                       prepared_query_item_bytes(g, SPR_NCC_DIRECT) / sizeof(float),
                       static_cast<const float*>(pg) + static_cast<size_t>(g0) * g_item_floats, g_item_floats, scores,
                       static_cast<long long>(ld), static_cast<long long>(col0 + g0), accumulate, maps_out, l.pws, l.tws,
                       static_cast<unsigned>(l.t_off));
    const int rc = check_launch("pair_direct_kernel");
    if (rc != SPR_OK) return rc;
  }
  return SPR_OK;
}

int launch_pair_direct(const NccGeom& g, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores,
                       int64_t ld, int64_t col0, int accumulate, float* maps_out, hipStream_t stream) {
  if (nq == 0 || ng == 0) return SPR_OK;
  switch (g.strips_per_thread) {
    case 1: return launch_pair_direct_t<1>(g, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, stream);
    case 2: return launch_pair_direct_t<2>(g, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, stream);
    case 4: return launch_pair_direct_t<4>(g, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, stream);
    case 8: return launch_pair_direct_t<8>(g, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, stream);
    default: set_error("direct NCC: unsupported strips per thread %d", g.strips_per_thread); return SPR_ERR_UNSUPPORTED;
  }
}

}  // namespace spr
