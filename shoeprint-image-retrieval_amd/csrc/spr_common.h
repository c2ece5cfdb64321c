// Internal declarations shared by the HIP translation units of libshoeprint_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/shoeprint_mi355x.h"
#include <spr_intrinsics.h>  // angle form: the CPU-emulation test build shadows it by include path

namespace spr {

constexpr int kThreads = 256;          // work-items per workgroup of the prep / direct / rank kernels (4 waves)
constexpr int kLdsLimit = 160 * 1024;  // LDS per CU on gfx950

// complex<float> as a 2-wide vector: 8-byte aligned (single b64 LDS/global accesses) and complex
// add / sub / multiply lower to packed fp32 VALU (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32), which
// issue at the scalar rate on gfx950: half the instructions at the low occupancy these kernels run at.
typedef float cf __attribute__((ext_vector_type(2)));
static_assert(sizeof(cf) == 8, "cf must be 8 bytes");

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// ---------------------------------------------------------------------------------------------
// Geometry of one plan, passed by value to the kernels.
//
// Cropped template (query) maps are th x tw, cropped search (gallery) maps ih x iw.  'same'-mode
// output (similarity.py:55, scipy 'same') has the search map's size; output pixel (y, x)
// correlates the template with the window whose top-left corner is (y - th/2, x - tw/2).
struct NccGeom {
  int channels;
  int q_h, q_w, g_h, g_w;  // raw map sizes as stored by the caller
  int crop;
  int th, tw, ih, iw;  // cropped sizes
  int dtype;
  // FFT method only ------------------------------------------------------------------------
  int nh, nw;          // FFT grid (rows, cols); nh = eh*tgh, nw = ew*tgw
  int nt;              // work-items per workgroup of the pair kernel (prepared layouts are tiled by it)
  int eh, tgh, ew, tgw;
  int big;             // 1: working set in the plan's global workspace instead of LDS (maps too large for LDS)
  int six;             // 1: prepared layouts of the six-wave pair kernel (ncc_pair6.hip)
  int tight;           // 1: ih <= nh/2 and iw <= nw/2 (the pruned kernel variant), 0: general variant
  int rounds_c;        // column-pass rounds of (kThreads/tgh) columns covering nw/2 columns
  int r_rows;          // rows of the intermediate LDS image kept after the column pass (ih rounded up to 8)
  int r_stride;        // row stride (complex elements) of the transposed LDS image RT[column][row]
  int rounds_r;        // row-pass rounds of (kThreads/tgw) row pairs covering r_rows/2 pairs (r_rows is even)
  int keep_w;          // kept outputs per row sub-transform (covers iw)
  int nv;              // 1/sigma values (= accumulators) per lane and row round, a multiple of 4
  int spec_per_chan;   // complex elements of one channel's spectrum (tiled layout)
  int inv_per_chan;    // floats of one channel's 1/sigma map in pair-kernel register order
  // direct method only ---------------------------------------------------------------------
  int pad_h, pad_w;    // zero-padded search map: ih+th-1, iw+tw-1
  int strips_per_row;  // strips of kStrip output pixels per output row
  int strips_per_thread;
  // matrix-core method only -----------------------------------------------------------------
  int mfma_exact;      // 1: raw search map on the matrix cores + correction matrix (ncc_mfma.hip), 0: hi + lo
  int mfma_general;    // 1: the general instance (template up to 30 x 16 on a map up to 28 x 12), 0: the 28 x 12 / 28 x 12 one
};

constexpr int kStrip = 8;  // output pixels per register strip in the direct kernel

struct spr_ncc_plan_impl;

// Launchers (each in its own .hip file).  All enqueue on `stream` and return SPR_OK / SPR_ERR_HIP.
int launch_prep_direct(const NccGeom& g, bool is_query, const void* maps, int64_t n, void* prepared,
                       hipStream_t stream);
int launch_pair_direct(const NccGeom& g, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores,
                       int64_t ld, int64_t col0, int accumulate, float* maps_out, hipStream_t stream);
struct FftWorkspace {  // the plan's scratch for "big" FFT geometries (null / 0 otherwise)
  void* base;
  size_t bytes;
  const float* six_ctab;  // six-wave pair kernel: its pre-twist table (nw/2 floats), null otherwise
};
// six-wave pair kernel (ncc_pair6.hip)
size_t pair6_lds_bytes();
int pair6_max_rows();  // cropped search-map rows / columns it covers
int pair6_max_cols();
int launch_pair6(const NccGeom& g, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores, int64_t ld,
                 int64_t col0, int accumulate, float* maps_out, const cf* tw_h, const float* ctab, hipStream_t stream);
// Pair kernels run one workgroup per pair in tiles of `pairs_per_tile`; HIP refuses a grid of 2^32 work-items or
// more, so a launch takes at most this many tiles (SPR_NCC_MAX_TILES lowers it: tests of the slicing).
int64_t pair_tiles_per_launch(int pairs_per_tile, int threads);
size_t fft_workspace_bytes(const NccGeom& g);  // what a plan with this geometry must allocate (0: none)
int launch_prep_fft(const NccGeom& g, bool is_query, const void* maps, int64_t n, void* prepared, const cf* tw_h,
                    const cf* tw_w, const FftWorkspace& ws, hipStream_t stream);
int launch_pair_fft(const NccGeom& g, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores,
                    int64_t ld, int64_t col0, int accumulate, float* maps_out, const cf* tw_h, const cf* tw_w,
                    unsigned* team_sync, const FftWorkspace& ws,
                    hipStream_t stream);  // team_sync: 8 x 32 counters, or null (tile mode only)
// direct form on the bf16 matrix cores (ncc_mfma.hip): small maps stored as bfloat16
bool mfma_geometry(NccGeom& g);  // true if an instantiated kernel covers this plan (fills mfma_exact)
size_t mfma_query_item_bytes(const NccGeom& g);
size_t mfma_gallery_item_bytes(const NccGeom& g);
size_t mfma_workspace_bytes(const NccGeom& g);  // the plan's correction matrix of the exact form (0: none)
int launch_prep_mfma(const NccGeom& g, bool is_query, const void* maps, int64_t n, void* prepared, hipStream_t stream);
int launch_pair_mfma(const NccGeom& g, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores, int64_t ld,
                     int64_t col0, int accumulate, float* maps_out, float* xws, hipStream_t stream);
bool fft_geometry(NccGeom& g, bool pow2_only);  // fills the FFT fields; false if no instantiated kernel fits
bool direct_geometry(NccGeom& g);  // fills the direct fields; false if the maps do not fit LDS

// Prepared-buffer sizes (bytes per item), both 256-byte multiples.
size_t prepared_query_item_bytes(const NccGeom& g, int method);
size_t prepared_gallery_item_bytes(const NccGeom& g, int method);

// 16-bit 3x3 / stride 1 convolution of vgg_conv.hip, shared with the ResNet plans (NHWC 16-bit in / out)
int pack_conv16_3x3(int kind, const float* w, const float* b, float* packed, size_t w_off, size_t b_off, int cin, int cout,
                    hipStream_t s);
int launch_conv16_3x3(int kind, const uint16_t* in, int64_t n, int h, int w, int cin, int cout, const uint16_t* w16,
                      const float* bias, int relu, uint16_t* out, hipStream_t s);

// first convolution of a plain VGG in a 16-bit plan (resnet.hip: the stem kernel's 3x3 / stride 1 instance)
int pack_first16(int kind, const float* w, const float* b, float* packed, size_t w_off, size_t b_off, hipStream_t s);
int launch_first16(int kind, const uint8_t* images, int64_t n, int h, int w, int in_channels, const float* mean3,
                   const float* inv_std3, const uint16_t* w16, const float* bias, int relu, uint16_t* out, hipStream_t s);

// Load one feature value of any supported storage type as float.
__device__ __forceinline__ float load_feature(const void* base, size_t idx, int dtype) {
  if (dtype == SPR_F32) return static_cast<const float*>(base)[idx];
  const uint16_t bits = static_cast<const uint16_t*>(base)[idx];
  if (dtype == SPR_BF16) {
    union { uint32_t u; float f; } v;
    v.u = static_cast<uint32_t>(bits) << 16;
    return v.f;
  }
  union { uint16_t u; _Float16 h; } v;
  v.u = bits;
  return static_cast<float>(v.h);
}

// float32 -> float16 / bfloat16 bit pattern, round to nearest even (what a stored 16-bit activation or weight of the 16-bit
// extractor plans holds), and back.  KIND = SPR_F16 | SPR_BF16.
__host__ __device__ inline uint16_t round_bf16(float v) {
  union { float f; uint32_t u; } c;
  c.f = v;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;  // NaN
  c.u += 0x7fffu + ((c.u >> 16) & 1u);                    // ties to even
  return static_cast<uint16_t>(c.u >> 16);
}
__device__ __forceinline__ uint16_t round_f16(float v) {
  union { _Float16 h; uint16_t u; } c;
  c.h = static_cast<_Float16>(v);  // v_cvt_f16_f32: round to nearest even
  return c.u;
}
template <int KIND>
__device__ __forceinline__ uint16_t round16(float v) { return KIND == SPR_F16 ? round_f16(v) : round_bf16(v); }
template <int KIND>
__device__ __forceinline__ float value16(uint16_t b) {
  if (KIND == SPR_F16) {
    union { uint16_t u; _Float16 h; } c;
    c.u = b;
    return static_cast<float>(c.h);
  }
  union { uint32_t u; float f; } c;
  c.u = static_cast<uint32_t>(b) << 16;
  return c.f;
}

// Workgroup reductions over the launch's work-items (a multiple of 64); `scratch` holds one value per wave.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
  const int tid = static_cast<int>(threadIdx.x);
  __syncthreads();  // scratch may still be read by a previous reduction
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < static_cast<int>(blockDim.x) / 64; ++w) s += scratch[w];
  return s;
}
template <int NT = kThreads>
__device__ __forceinline__ float block_max(float v, float* scratch) {
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
  const int tid = static_cast<int>(threadIdx.x);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  float s = scratch[0];
  for (int w = 1; w < NT / 64; ++w) s = fmaxf(s, scratch[w]);
  return s;
}

}  // namespace spr
