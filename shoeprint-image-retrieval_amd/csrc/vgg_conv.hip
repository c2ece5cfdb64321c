// VGG16 feature extractor: torchvision vgg16().features[:block] (network.py:125-134, 185-186, 235).
//
// Every 3x3 / stride 1 / pad 1 convolution is an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32, bit-for-bit a k-ordered fmaf chain), with bias, ReLU and the
// 2x2 max-pool that follows it fused into the epilogue.  Activations between layers are NHWC in HBM
// (the GEMM's K index = input channel is then contiguous); the last layer writes NCHW float32, the
// layout Model.get_feature_maps returns (network.py:241-244) and the NCC prep kernels read.
//
// conv_mfma_kernel, one workgroup = 16x16 output pixels x 64 output channels, 4 waves, wave w owns
// output rows 4w..4w+3 (4 row segments of 16 pixels = 4 A blocks) x 4 blocks of 16 channels:
//   for every chunk of 16 input channels:   stage the 18x18x16 input patch and the 9x64x16 filter slab
//                                           in LDS (16-byte loads, zero fill outside the image)
//     for every tap (dy,dx):                4 A fragments (ds_read_b128: 4 input channels per lane) and
//                                           4 B fragments feed 4x4x4 MFMAs; the k index of MFMA j is
//                                           {4q+j : q = 0..3} for both operands (any partition of the 16
//                                           channels works as long as A and B agree).
// Arithmetic per image at 512x256 up to conv3_3: 48.77 GFLOP (SURVEY §8 a-E2); bound: fp32 MFMA 157 TFLOP/s.
//
// conv_first_kernel: conv1_1 has 3 input planes (K = 27, 0.9 % of the flops): plain FMA kernel that
// also applies ToTensor / repeat(3) / Normalize (network.py:60-71) — x'_c = (u8/255 - mean_c) * inv_std_c,
// zero padding AFTER normalisation.
#include <new>
#include <vector>

#include "spr_common.h"

namespace spr {
namespace {

constexpr int kTile = 16;       // output tile edge (pixels)
constexpr int kPatch = kTile + 2;
constexpr int kCk = 16;         // input channels per chunk
constexpr int kCS = 16;         // LDS stride (floats) of one patch pixel / one filter row (16 channels, no padding)
// 16-byte quarter `quarter` of LDS row `row` sits at a swizzled position: rows 4 apart start on the same bank,
// the XOR moves their quarters apart (measured against 4 floats of padding per row: bank conflicts 50 % -> lower,
// 1 % faster, 14 KB less LDS per workgroup).
__host__ __device__ inline int qoff(int row, int quarter) { return ((quarter ^ ((row >> 2) & 3)) << 2); }
constexpr int kTN = 64;         // output channels per workgroup
constexpr size_t kConvLdsBytes = sizeof(float) * ((16 + 2) * (16 + 2) * 16 + 9 * 64 * 16);  // = kConvLds below
constexpr int kCk16 = 32;       // 16-bit matrix cores: input channels per chunk = the K of one v_mfma_f32_16x16x32_{bf16,f16}

// Compute types of a plan (spr_vgg_plan_create_ex): the f32 matrix cores (exact, the reference's arithmetic, network.py:235),
// or 16-bit operands with f32 accumulation - weights and the activations BETWEEN layers rounded to float16 / bfloat16
// (round to nearest even), bias / ReLU / pool and the last layer's output in f32.
enum { kF32 = SPR_F32, kF16 = SPR_F16, kBF16 = SPR_BF16 };

struct Stage {
  int cin, cout;
  int relu, pool;
  size_t w_off, b_off;  // float offsets into the packed parameter buffer
};

// ---------------------------------------------------------------- parameter packing
// first conv: [tap*3 + c][64]  |  MFMA convs: [cout/64][cin/16][tap][n:64][c:16]
__global__ void __launch_bounds__(kThreads)
pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ packed,
                    size_t w_off, size_t b_off, int cin, int cout, int first) {
  const size_t total = static_cast<size_t>(cout) * cin * 9;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * kThreads) {
    // i indexes the torch layout [n][c][ky][kx]
    const int tap = static_cast<int>(i % 9);
    const int c = static_cast<int>((i / 9) % cin);
    const int n = static_cast<int>(i / (static_cast<size_t>(9) * cin));
    size_t dst;
    if (first) {
      dst = static_cast<size_t>(tap * 3 + c) * cout + n;
    } else {
      const int cb = n / kTN, nn = n % kTN, cc = c / kCk, ci = c % kCk;
      dst = ((((static_cast<size_t>(cb) * (cin / kCk) + cc) * 9 + tap) * kTN + nn) * kCk) + ci;
    }
    packed[w_off + dst] = w[i];
  }
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < cout; i += gridDim.x * kThreads) packed[b_off + i] = b[i];
}

// 16-bit plans: the MFMA convolutions' weights as [cout/64][cin/32][tap][n:64][c:32] float16 / bfloat16 (64-byte rows: the
// same bytes per row as the f32 layout, so the LDS image and its swizzle are shared); bias stays f32.
template <int KIND>
__global__ void __launch_bounds__(kThreads)
pack_weights16_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ packed, size_t w_off,
                      size_t b_off, int cin, int cout) {
  uint16_t* dst16 = reinterpret_cast<uint16_t*>(packed + w_off);
  const size_t total = static_cast<size_t>(cout) * cin * 9;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * kThreads) {
    const int tap = static_cast<int>(i % 9);
    const int c = static_cast<int>((i / 9) % cin);
    const int n = static_cast<int>(i / (static_cast<size_t>(9) * cin));
    const int cb = n / kTN, nn = n % kTN, cc = c / kCk16, ci = c % kCk16;
    dst16[((((static_cast<size_t>(cb) * (cin / kCk16) + cc) * 9 + tap) * kTN + nn) * kCk16) + ci] = round16<KIND>(w[i]);
  }
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < cout; i += gridDim.x * kThreads) packed[b_off + i] = b[i];
}

// ---------------------------------------------------------------- conv1_1 (+ pre-processing)
// grid = (tiles, n images); out NHWC [n][H][W][64] or NCHW when it is the last stage.
__global__ void __launch_bounds__(kThreads)
conv_first_kernel(const uint8_t* __restrict__ images, int H, int W, int in_channels, float m0, float m1, float m2,
                  float s0, float s1, float s2, const float* __restrict__ wts, const float* __restrict__ bias,
                  int relu, int nchw, float* __restrict__ out, int kind16) {
  // kind16 != 0 (16-bit plans, not the last stage): the NHWC activation is stored rounded to float16 / bfloat16
  // Lane = output channel (its 27 weights live in registers), wave = strips of 8 output pixels: a strip reads
  // its 3 x 10 x 3 input values with wave-uniform (broadcast) 16-byte LDS reads and does 216 FMAs on them, and a
  // pixel's 64 channels leave as one 256-byte store.
  constexpr int kRowF = 56;  // floats per patch row: 18 pixels x 3 planes, padded to 16-byte multiples
  __shared__ __attribute__((aligned(16))) float patch[kPatch * kRowF];
  const int tiles_x = ceil_div(W, kTile);
  const int ty = static_cast<int>(blockIdx.x) / tiles_x, tx = static_cast<int>(blockIdx.x) % tiles_x;
  const int y0 = ty * kTile, x0 = tx * kTile;
  const size_t img = blockIdx.y;
  const int tid = static_cast<int>(threadIdx.x);
  const float mean[3] = {m0, m1, m2}, istd[3] = {s0, s1, s2};
  for (int i = tid; i < kPatch * kPatch; i += kThreads) {
    const int py = i / kPatch, px = i % kPatch;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    const bool in = y >= 0 && y < H && x >= 0 && x < W;
    for (int c = 0; c < 3; ++c) {
      float v = 0.0f;  // zero padding of the NORMALISED tensor (network.py:69 then conv padding)
      if (in) {
        const size_t pix = (img * H + y) * static_cast<size_t>(W) + x;
        const float u = static_cast<float>(in_channels == 1 ? images[pix] : images[pix * 3 + c]);
        v = (u / 255.0f - mean[c]) * istd[c];  // ToTensor then Normalize, in this order (network.py:64-69)
      }
      patch[py * kRowF + px * 3 + c] = v;
    }
  }
  if (tid < kPatch) { patch[tid * kRowF + 54] = 0.0f; patch[tid * kRowF + 55] = 0.0f; }
  const int n = tid & 63, wave = tid >> 6;
  float w[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) w[k] = wts[k * 64 + n];  // packed [tap*3 + c][64]
  const float b = bias[n];
  __syncthreads();
  for (int strip = wave; strip < 2 * kTile; strip += kThreads / 64) {
    const int py = strip >> 1, px0 = (strip & 1) * 8;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = b;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const float4* row = reinterpret_cast<const float4*>(patch + (py + dy) * kRowF + px0 * 3);  // 30 floats used
      float v[32];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 r = row[k];
        v[4 * k] = r.x; v[4 * k + 1] = r.y; v[4 * k + 2] = r.z; v[4 * k + 3] = r.w;
      }
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = fmaf(v[(i + dx) * 3 + c], w[(dy * 3 + dx) * 3 + c], acc[i]);
    }
    const int y = y0 + py;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int x = x0 + px0 + i;
      if (y < H && x < W) {
        const float r = relu ? fmaxf(acc[i], 0.0f) : acc[i];
        if (nchw) out[((img * 64 + n) * H + y) * static_cast<size_t>(W) + x] = r;
        else if (kind16 == 0) out[((img * H + y) * static_cast<size_t>(W) + x) * 64 + n] = r;
        else reinterpret_cast<uint16_t*>(out)[((img * H + y) * static_cast<size_t>(W) + x) * 64 + n] =
                 kind16 == kF16 ? round_f16(r) : round_bf16(r);
      }
    }
  }
}

// ---------------------------------------------------------------- 3x3 conv on fp32 MFMA
// grid = (tiles, cout/64, n images)
__global__ void __launch_bounds__(kThreads, 2)
conv_mfma_kernel(const float* __restrict__ in, int H, int W, int cin, int cout, const float* __restrict__ wts,
                 const float* __restrict__ bias, int relu, int pool, int nchw, float* __restrict__ out,
                 float* __restrict__ tap) {
  unsigned char* lds = dyn_lds();
  float* patch = reinterpret_cast<float*>(lds);                 // [18*18][kCS]
  float* wl = patch + kPatch * kPatch * kCS;                    // [9*64][kCS]
  const int tiles_x = ceil_div(W, kTile);
  const int ty = static_cast<int>(blockIdx.x) / tiles_x, tx = static_cast<int>(blockIdx.x) % tiles_x;
  const int y0 = ty * kTile, x0 = tx * kTile;
  const int cb = static_cast<int>(blockIdx.y);
  const size_t img = blockIdx.z;
  const int tid = static_cast<int>(threadIdx.x);
  const int wave = tid >> 6, lane = tid & 63;
  const int p = lane & 15, q = lane >> 4;  // MFMA lane coordinates: row/col index, k index
  const int nchunks = cin / kCk;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* in_img = in + img * static_cast<size_t>(H) * W * cin;
  // Staging is software-pipelined: the 16-byte pieces of chunk cc+1 (input patch + filter slab) are requested
  // into registers before the MFMA loop of chunk cc and written to LDS after it, so their HBM/L2 latency runs
  // under ~18 k cycles of matrix work instead of in front of it.
  constexpr int kPatchPieces = kPatch * kPatch * 4, kFilterPieces = 9 * kTN * 4;
  constexpr int kPP = (kPatchPieces + kThreads - 1) / kThreads, kFP = kFilterPieces / kThreads;
  static_assert(kFilterPieces % kThreads == 0, "filter slab pieces divide over the workgroup");
  float4 pre_p[kPP];
  auto request = [&](int cc) {
#pragma unroll
    for (int k = 0; k < kPP; ++k) {
      const int i = tid + k * kThreads;
      const int pp = i >> 2, qq = i & 3;
      const int y = y0 - 1 + pp / kPatch, x = x0 - 1 + pp % kPatch;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < kPatchPieces && y >= 0 && y < H && x >= 0 && x < W)
        v = *reinterpret_cast<const float4*>(in_img + (static_cast<size_t>(y) * W + x) * cin + cc * kCk + qq * 4);
      pre_p[k] = v;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < kPP; ++k) {
      const int i = tid + k * kThreads;
      if (i < kPatchPieces) *reinterpret_cast<float4*>(patch + (i >> 2) * kCS + qoff(i >> 2, i & 3)) = pre_p[k];
    }
  };
  request(0);
  for (int cc = 0; cc < nchunks; ++cc) {
    __syncthreads();  // the previous chunk's fragments are consumed
    commit();
    {  // filter slab [tap][n][16] of this (cout block, chunk): contiguous in the packed buffer, L2-resident
      const float* wsrc = wts + (static_cast<size_t>(cb) * nchunks + cc) * (9 * kTN * kCk);
#pragma unroll
      for (int k = 0; k < kFP; ++k) {
        const int i = tid + k * kThreads;
        *reinterpret_cast<float4*>(wl + (i >> 2) * kCS + qoff(i >> 2, i & 3)) =
            *reinterpret_cast<const float4*>(wsrc + (i >> 2) * kCk + (i & 3) * 4);
      }
    }
    __syncthreads();
    if (cc + 1 < nchunks) request(cc + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      float4 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a[i] = *reinterpret_cast<const float4*>(patch + ((wave * 4 + i + dy) * kPatch + (p + dx)) * kCS +
                                                qoff((wave * 4 + i + dy) * kPatch + (p + dx), q));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        b[j] = *reinterpret_cast<const float4*>(wl + ((tap * kTN) + j * 16 + p) * kCS + qoff((tap * kTN) + j * 16 + p, q));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = mfma_f32_16x16x4(a[i].x, b[j].x, acc[i][j]);
          acc[i][j] = mfma_f32_16x16x4(a[i].y, b[j].y, acc[i][j]);
          acc[i][j] = mfma_f32_16x16x4(a[i].z, b[j].z, acc[i][j]);
          acc[i][j] = mfma_f32_16x16x4(a[i].w, b[j].w, acc[i][j]);
        }
      }
    }
  }

  // ---- epilogue: lane (q, p) owns output pixels x0 + 4q + jj (jj = 0..3) of rows y0 + 4*wave + i,
  //      channel cb*64 + 16j + p  (MFMA C/D map: row = 4*(lane>>4) + reg, col = lane & 15)
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = cb * kTN + j * 16 + p;
    const float bv = bias[ch];
    float v[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float t = acc[i][j][jj] + bv;
        v[i][jj] = relu ? fmaxf(t, 0.0f) : t;
      }
    if (tap) {  // a feature tap (multi-layer scoring): this layer's activation before the pool, NCHW float32
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int y = y0 + 4 * wave + i, x = x0 + 4 * q + jj;
          if (y < H && x < W) tap[((img * cout + ch) * H + y) * static_cast<size_t>(W) + x] = v[i][jj];
        }
    }
    if (pool) {
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
          const float m = fmaxf(fmaxf(v[2 * i2][2 * j2], v[2 * i2][2 * j2 + 1]),
                                fmaxf(v[2 * i2 + 1][2 * j2], v[2 * i2 + 1][2 * j2 + 1]));
          const int y = (y0 + 4 * wave) / 2 + i2, x = (x0 + 4 * q) / 2 + j2;
          if (y < Ho && x < Wo) {
            if (nchw) out[((img * cout + ch) * Ho + y) * static_cast<size_t>(Wo) + x] = m;
            else out[((img * Ho + y) * static_cast<size_t>(Wo) + x) * cout + ch] = m;
          }
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int y = y0 + 4 * wave + i, x = x0 + 4 * q + jj;
          if (y < Ho && x < Wo) {
            if (nchw) out[((img * cout + ch) * Ho + y) * static_cast<size_t>(Wo) + x] = v[i][jj];
            else out[((img * Ho + y) * static_cast<size_t>(Wo) + x) * cout + ch] = v[i][jj];
          }
        }
    }
  }
}

// ---------------------------------------------------------------- 3x3 conv on the 16-bit matrix cores
// The same implicit GEMM, tile and LDS image as conv_mfma_kernel with 32 input channels per chunk: a pixel's 64-byte LDS
// row now holds 32 float16 / bfloat16 channels, and the 16-byte quarter a lane reads (8 channels) is exactly its operand
// of ONE v_mfma_f32_16x16x32 (lane (p, q): A row p = pixel, k = 8q .. 8q + 7; B column p = output channel, same k).  The
// matrix work per chunk is 16 x shorter than on the f32 cores while the bytes staged are the same, so BOTH the input patch
// and the filter slab of the next chunk are requested into registers before the MFMA loop of the current one.
// in: NHWC 16-bit; out: NHWC 16-bit (rounded, the next layer's operand) or, for the last layer, NCHW float32.
template <int KIND>
__global__ void __launch_bounds__(kThreads, 2)
conv16_kernel(const uint16_t* __restrict__ in, int H, int W, int cin, int cout, const uint16_t* __restrict__ wts,
              const float* __restrict__ bias, int relu, int pool, int nchw, float* __restrict__ out,
              float* __restrict__ tap) {
  unsigned char* lds = dyn_lds();
  float* patch = reinterpret_cast<float*>(lds);  // [18*18][16 dwords]: dword = two channels
  float* wl = patch + kPatch * kPatch * kCS;     // [9*64][16 dwords]
  const int tiles_x = ceil_div(W, kTile);
  const int ty = static_cast<int>(blockIdx.x) / tiles_x, tx = static_cast<int>(blockIdx.x) % tiles_x;
  const int y0 = ty * kTile, x0 = tx * kTile;
  const int cb = static_cast<int>(blockIdx.y);
  const size_t img = blockIdx.z;
  const int tid = static_cast<int>(threadIdx.x);
  const int wave = tid >> 6, lane = tid & 63;
  const int p = lane & 15, q = lane >> 4;
  const int nchunks = cin / kCk16;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint16_t* in_img = in + img * static_cast<size_t>(H) * W * cin;
  constexpr int kPatchPieces = kPatch * kPatch * 4, kFilterPieces = 9 * kTN * 4;
  constexpr int kPP = (kPatchPieces + kThreads - 1) / kThreads, kFP = kFilterPieces / kThreads;
  // (native vector types: as arrays of HIP's float4 struct the nine filter pieces stayed in scratch memory - a store and a
  // load of 160 bytes per lane and chunk behind every global load, which made the prefetch synchronous)
  u32x4 pre_p[kPP], pre_f[kFP];
  auto request = [&](int cc) {
#pragma unroll
    for (int k = 0; k < kPP; ++k) {
      const int i = tid + k * kThreads;
      const int pp = i >> 2, qq = i & 3;
      const int y = y0 - 1 + pp / kPatch, x = x0 - 1 + pp % kPatch;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (i < kPatchPieces && y >= 0 && y < H && x >= 0 && x < W)
        v = *reinterpret_cast<const u32x4*>(in_img + (static_cast<size_t>(y) * W + x) * cin + cc * kCk16 + qq * 8);
      pre_p[k] = v;
    }
    const uint16_t* wsrc = wts + (static_cast<size_t>(cb) * nchunks + cc) * (9 * kTN * kCk16);
#pragma unroll
    for (int k = 0; k < kFP; ++k) {
      const int i = tid + k * kThreads;
      pre_f[k] = *reinterpret_cast<const u32x4*>(wsrc + (i >> 2) * kCk16 + (i & 3) * 8);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < kPP; ++k) {
      const int i = tid + k * kThreads;
      if (i < kPatchPieces) *reinterpret_cast<u32x4*>(patch + (i >> 2) * kCS + qoff(i >> 2, i & 3)) = pre_p[k];
    }
#pragma unroll
    for (int k = 0; k < kFP; ++k) {
      const int i = tid + k * kThreads;
      *reinterpret_cast<u32x4*>(wl + (i >> 2) * kCS + qoff(i >> 2, i & 3)) = pre_f[k];
    }
  };
  request(0);
  for (int cc = 0; cc < nchunks; ++cc) {
    __syncthreads();  // the previous chunk's fragments are consumed
    commit();
    __syncthreads();
    if (cc + 1 < nchunks) request(cc + 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3, dx = t % 3;
      u32x4 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a[i] = *reinterpret_cast<const u32x4*>(patch + ((wave * 4 + i + dy) * kPatch + (p + dx)) * kCS +
                                               qoff((wave * 4 + i + dy) * kPatch + (p + dx), q));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        b[j] = *reinterpret_cast<const u32x4*>(wl + ((t * kTN) + j * 16 + p) * kCS + qoff((t * kTN) + j * 16 + p, q));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = KIND == kF16 ? mfma_f16_16x16x32(a[i], b[j], acc[i][j]) : mfma_bf16_16x16x32(a[i], b[j], acc[i][j]);
    }
  }

  // ---- epilogue.  The C/D map is conv_mfma_kernel's: lane (q, p) owns pixels x0 + 4q + jj of rows y0 + 4 wave + i, channel
  // cb*64 + 16j + p.  A 16-bit NHWC result stored from there would be 64 two-byte stores per lane, each instruction touching
  // 32-byte pieces of four pixels (measured: as many store as load instructions, waves parked 65 % of their time).  So the
  // activations (after bias / ReLU / pool, f32) go through LDS, 128 pixels x 64 channels at a time, and leave as 16-byte
  // pieces of eight channels.  The float32 NCHW result of the last layer and the feature taps keep the direct form.
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  uint16_t* out16 = reinterpret_cast<uint16_t*>(out);
  constexpr int kET = 68;  // row stride (floats) of the staging tile
  float* T = reinterpret_cast<float*>(lds);
  static_assert(128 * kET * sizeof(float) <= kConvLdsBytes, "the staging tile fits the operand tiles");
  // channel block j of this lane after bias / ReLU, and its feature tap (float32 NCHW, before any pool)
  auto activations = [&](int j, float (&v)[4][4]) {
    const int ch = cb * kTN + j * 16 + p;
    const float bv = bias[ch];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float t = acc[i][j][jj] + bv;
        v[i][jj] = relu ? fmaxf(t, 0.0f) : t;
      }
    if (tap) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int y = y0 + 4 * wave + i, x = x0 + 4 * q + jj;
          if (y < H && x < W) tap[((img * cout + ch) * H + y) * static_cast<size_t>(W) + x] = v[i][jj];
        }
    }
  };
  if (nchw) {  // the last layer: float32 NCHW, direct
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = cb * kTN + j * 16 + p;
      float v[4][4];
      activations(j, v);
      if (pool) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int j2 = 0; j2 < 2; ++j2) {
            const float m = fmaxf(fmaxf(v[2 * i2][2 * j2], v[2 * i2][2 * j2 + 1]),
                                  fmaxf(v[2 * i2 + 1][2 * j2], v[2 * i2 + 1][2 * j2 + 1]));
            const int y = (y0 + 4 * wave) / 2 + i2, x = (x0 + 4 * q) / 2 + j2;
            if (y < Ho && x < Wo) out[((img * cout + ch) * Ho + y) * static_cast<size_t>(Wo) + x] = m;
          }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int y = y0 + 4 * wave + i, x = x0 + 4 * q + jj;
            if (y < Ho && x < Wo) out[((img * cout + ch) * Ho + y) * static_cast<size_t>(Wo) + x] = v[i][jj];
          }
      }
    }
    return;
  }
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {  // tile rows 8h .. 8h + 7: the waves 2h and 2h + 1
    __syncthreads();             // the operand tiles / the previous half are consumed
    if ((wave >> 1) == h) {
      const int lw = wave & 1;   // this wave's four (pooled: two) pixel rows inside the half
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4][4];
        activations(j, v);
        if (pool) {  // the half holds 4 x 8 pooled pixels
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
              const float m = fmaxf(fmaxf(v[2 * i2][2 * j2], v[2 * i2][2 * j2 + 1]),
                                    fmaxf(v[2 * i2 + 1][2 * j2], v[2 * i2 + 1][2 * j2 + 1]));
              T[((2 * lw + i2) * 8 + 2 * q + j2) * kET + j * 16 + p] = m;
            }
        } else {     // 8 x 16 pixels
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) T[((4 * lw + i) * 16 + 4 * q + jj) * kET + j * 16 + p] = v[i][jj];
        }
      }
    }
    __syncthreads();
    // a 16-byte piece (pixel row of T, eight channels) per work-item and pass
    const int rows = pool ? 32 : 128, tw = pool ? 8 : 16;
    for (int e = tid; e < rows * 8; e += kThreads) {
      const int row = e >> 3, piece = e & 7;
      const int ly = row / tw, lx = row - ly * tw;
      const int y = (pool ? y0 / 2 + 4 * h : y0 + 8 * h) + ly, x = (pool ? x0 / 2 : x0) + lx;
      if (y >= Ho || x >= Wo) continue;
      const float4 lo = *reinterpret_cast<const float4*>(T + row * kET + piece * 8);
      const float4 hi = *reinterpret_cast<const float4*>(T + row * kET + piece * 8 + 4);
      u32x4 o;
      o[0] = static_cast<uint32_t>(round16<KIND>(lo.x)) | (static_cast<uint32_t>(round16<KIND>(lo.y)) << 16);
      o[1] = static_cast<uint32_t>(round16<KIND>(lo.z)) | (static_cast<uint32_t>(round16<KIND>(lo.w)) << 16);
      o[2] = static_cast<uint32_t>(round16<KIND>(hi.x)) | (static_cast<uint32_t>(round16<KIND>(hi.y)) << 16);
      o[3] = static_cast<uint32_t>(round16<KIND>(hi.z)) | (static_cast<uint32_t>(round16<KIND>(hi.w)) << 16);
      *reinterpret_cast<u32x4*>(out16 + ((img * Ho + y) * static_cast<size_t>(Wo) + x) * cout + cb * kTN + piece * 8) = o;
    }
  }
}

constexpr size_t kConvLds = sizeof(float) * (kPatch * kPatch * kCS + 9 * kTN * kCS);
static_assert(kConvLds == kConvLdsBytes, "one LDS size");

}  // namespace

// The 16-bit 3x3 / stride 1 convolution for other plans of the library (the ResNet's bottleneck 3x3 layers: the input patch
// staged once serves all nine taps, 164 flop per staged byte against 43 of the tap-by-tap GEMM tiles of resnet.hip).
// in / out: NHWC 16-bit; weights as pack_conv16_3x3 writes them.
int pack_conv16_3x3(int kind, const float* w, const float* b, float* packed, size_t w_off, size_t b_off, int cin, int cout,
                    hipStream_t s) {
  if (kind == SPR_F16)
    hipLaunchKernelGGL(pack_weights16_kernel<kF16>, dim3(256), dim3(kThreads), 0, s, w, b, packed, w_off, b_off, cin, cout);
  else
    hipLaunchKernelGGL(pack_weights16_kernel<kBF16>, dim3(256), dim3(kThreads), 0, s, w, b, packed, w_off, b_off, cin, cout);
  return check_launch("pack_weights16_kernel");
}
int launch_conv16_3x3(int kind, const uint16_t* in, int64_t n, int h, int w, int cin, int cout, const uint16_t* w16,
                      const float* bias, int relu, uint16_t* out, hipStream_t s) {
  const dim3 grid(static_cast<unsigned>(ceil_div(h, kTile) * ceil_div(w, kTile)), static_cast<unsigned>(cout / kTN),
                  static_cast<unsigned>(n));
  if (kind == SPR_F16) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv16_kernel<kF16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(kConvLds));
    hipLaunchKernelGGL(conv16_kernel<kF16>, grid, dim3(kThreads), kConvLds, s, in, h, w, cin, cout, w16, bias, relu, 0, 0,
                       reinterpret_cast<float*>(out), static_cast<float*>(nullptr));
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv16_kernel<kBF16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(kConvLds));
    hipLaunchKernelGGL(conv16_kernel<kBF16>, grid, dim3(kThreads), kConvLds, s, in, h, w, cin, cout, w16, bias, relu, 0, 0,
                       reinterpret_cast<float*>(out), static_cast<float*>(nullptr));
  }
  return check_launch("conv16_kernel");
}
}  // namespace spr

struct spr_vgg16_plan {
  int block;
  int compute;  // SPR_F32 (f32 matrix cores, exact) | SPR_F16 | SPR_BF16 (16-bit operands, f32 accumulation)
  std::vector<spr::Stage> stages;
  std::vector<int> feature_index;  // position of every convolution in model.features
  std::vector<int> bn_inside;      // 1: its BatchNorm2d lies inside features[:block] (folded by the host)
  size_t packed_floats;
};

using namespace spr;

// torchvision configurations: "D" (vgg16) and "E" (vgg19); *_bn puts a BatchNorm2d between every convolution
// and its ReLU.  BatchNorm in eval mode is an affine map per channel: the host folds it into the convolution's
// weights and bias when the layer lies inside features[:block], so the kernels never see it.
extern "C" int spr_vgg_plan_create(int32_t arch, int32_t block, spr_vgg16_plan** plan_out) {
  return spr_vgg_plan_create_ex(arch, block, SPR_F32, plan_out);
}

extern "C" int spr_vgg_plan_compute(const spr_vgg16_plan* plan) { return plan ? plan->compute : SPR_ERR_ARG; }

extern "C" int spr_vgg_plan_create_ex(int32_t arch, int32_t block, int32_t compute, spr_vgg16_plan** plan_out) {
  if (!plan_out) { set_error("spr_vgg_plan_create: null pointer"); return SPR_ERR_ARG; }
  *plan_out = nullptr;
  if (compute != SPR_F32 && compute != SPR_F16 && compute != SPR_BF16) {
    set_error("spr_vgg_plan_create_ex: compute type %d (SPR_F32 | SPR_F16 | SPR_BF16)", compute);
    return SPR_ERR_ARG;
  }
  static const int cfg_d[] = {64, 64, -1, 128, 128, -1, 256, 256, 256, -1, 512, 512, 512, -1, 512, 512, 512, -1};
  static const int cfg_e[] = {64, 64, -1, 128, 128, -1, 256, 256, 256, 256, -1,
                              512, 512, 512, 512, -1, 512, 512, 512, 512, -1};
  const int* cfg = cfg_d;
  int n_cfg = static_cast<int>(sizeof(cfg_d) / sizeof(int));
  bool bn = false;
  if (arch == SPR_VGG19 || arch == SPR_VGG19_BN) { cfg = cfg_e; n_cfg = static_cast<int>(sizeof(cfg_e) / sizeof(int)); }
  else if (arch != SPR_VGG16) { set_error("spr_vgg_plan_create: unknown architecture %d", arch); return SPR_ERR_ARG; }
  if (arch == SPR_VGG19_BN) bn = true;
  struct Op { char kind; int cin, cout; };
  std::vector<Op> ops;
  int c = 3;
  for (int k = 0; k < n_cfg; ++k) {
    const int v = cfg[k];
    if (v < 0) ops.push_back({'P', c, c});
    else {
      ops.push_back({'C', c, v});
      if (bn) ops.push_back({'B', v, v});
      ops.push_back({'R', v, v});
      c = v;
    }
  }
  if (block < 1 || block > static_cast<int>(ops.size())) {
    set_error("spr_vgg_plan_create: block %d outside [1, %zu] (= len(model.features))", block, ops.size());
    return SPR_ERR_ARG;
  }
  spr_vgg16_plan* plan = new (std::nothrow) spr_vgg16_plan();
  if (!plan) { set_error("out of host memory"); return SPR_ERR_ARG; }
  plan->block = block;
  plan->compute = compute;
  size_t off = 0;
  for (int i = 0; i < block; ++i) {
    if (ops[i].kind != 'C') continue;  // B is folded by the host, R and P are fused into the preceding convolution
    Stage s{};
    s.cin = ops[i].cin; s.cout = ops[i].cout;
    int j = i + 1;
    const bool has_bn = j < block && ops[j].kind == 'B';
    if (has_bn) ++j;
    s.relu = (j < block && ops[j].kind == 'R') ? 1 : 0;
    s.pool = (s.relu && j + 1 < block && ops[j + 1].kind == 'P') ? 1 : 0;
    // (16-bit plans: two weights per float slot; the first convolution's 32 x 64 padded 16-bit matrix fits its f32 slab)
    s.w_off = off; off += static_cast<size_t>(s.cout) * s.cin * 9 / ((compute != SPR_F32 && s.cin != 3) ? 2 : 1);
    s.b_off = off; off += static_cast<size_t>(s.cout);
    off = (off + 3) / 4 * 4;  // keep every slab 16-byte aligned
    plan->stages.push_back(s);
    plan->feature_index.push_back(i);
    plan->bn_inside.push_back(has_bn ? 1 : 0);
  }
  plan->packed_floats = off;
  *plan_out = plan;
  return SPR_OK;
}

extern "C" int spr_vgg16_plan_create(int32_t block, spr_vgg16_plan** plan_out) {
  return spr_vgg_plan_create(SPR_VGG16, block, plan_out);
}

extern "C" void spr_vgg16_plan_destroy(spr_vgg16_plan* plan) { delete plan; }

extern "C" int spr_vgg_conv_info(const spr_vgg16_plan* plan, int32_t i, int32_t* feature_index, int32_t* bn_inside) {
  if (!plan || !feature_index || !bn_inside || i < 0 || i >= static_cast<int>(plan->stages.size())) {
    set_error("spr_vgg_conv_info: bad argument");
    return SPR_ERR_ARG;
  }
  *feature_index = plan->feature_index[i];
  *bn_inside = plan->bn_inside[i];
  return SPR_OK;
}

extern "C" int spr_vgg16_num_convs(const spr_vgg16_plan* plan) {
  return plan ? static_cast<int>(plan->stages.size()) : SPR_ERR_ARG;
}

extern "C" int spr_vgg16_conv_shape(const spr_vgg16_plan* plan, int32_t i, int32_t* cin, int32_t* cout) {
  if (!plan || !cin || !cout || i < 0 || i >= static_cast<int>(plan->stages.size())) {
    set_error("spr_vgg16_conv_shape: bad argument");
    return SPR_ERR_ARG;
  }
  *cin = plan->stages[i].cin;
  *cout = plan->stages[i].cout;
  return SPR_OK;
}

extern "C" int spr_vgg16_output_shape(const spr_vgg16_plan* plan, int32_t in_h, int32_t in_w, int32_t* channels,
                                      int32_t* out_h, int32_t* out_w) {
  if (!plan || !channels || !out_h || !out_w || in_h < 1 || in_w < 1) {
    set_error("spr_vgg16_output_shape: bad argument");
    return SPR_ERR_ARG;
  }
  int h = in_h, w = in_w, c = 3;
  for (const Stage& s : plan->stages) {
    c = s.cout;
    if (s.pool) { h /= 2; w /= 2; }
  }
  if (h < 1 || w < 1) { set_error("image %dx%d vanishes under the pools of features[:%d]", in_h, in_w, plan->block); return SPR_ERR_SHAPE; }
  *channels = c; *out_h = h; *out_w = w;
  return SPR_OK;
}

extern "C" size_t spr_vgg16_packed_bytes(const spr_vgg16_plan* plan) {
  return plan ? plan->packed_floats * sizeof(float) : 0;
}

extern "C" int spr_vgg16_pack_weights(spr_vgg16_plan* plan, const float* const* weights, const float* const* biases,
                                      void* packed, spr_stream_t stream) {
  if (!plan || !weights || !biases || !packed) { set_error("spr_vgg16_pack_weights: null pointer"); return SPR_ERR_ARG; }
  for (size_t i = 0; i < plan->stages.size(); ++i) {
    const Stage& s = plan->stages[i];
    if (!weights[i] || !biases[i]) { set_error("spr_vgg16_pack_weights: null parameter %zu", i); return SPR_ERR_ARG; }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (i == 0 && plan->compute != SPR_F32 && plan->stages.size() > 1) {
      // 16-bit plans: the first convolution runs on the matrix cores too (K = 27 padded to 32), unless it is the whole plan
      const int rc0 = pack_first16(plan->compute, weights[i], biases[i], static_cast<float*>(packed), s.w_off, s.b_off, hs);
      if (rc0 != SPR_OK) return rc0;
      continue;
    }
    if (i > 0 && plan->compute == SPR_F16)
      hipLaunchKernelGGL(pack_weights16_kernel<kF16>, dim3(256), dim3(kThreads), 0, hs, weights[i], biases[i],
                         static_cast<float*>(packed), s.w_off, s.b_off, s.cin, s.cout);
    else if (i > 0 && plan->compute == SPR_BF16)
      hipLaunchKernelGGL(pack_weights16_kernel<kBF16>, dim3(256), dim3(kThreads), 0, hs, weights[i], biases[i],
                         static_cast<float*>(packed), s.w_off, s.b_off, s.cin, s.cout);
    else
      hipLaunchKernelGGL(pack_weights_kernel, dim3(256), dim3(kThreads), 0, hs, weights[i],
                         biases[i], static_cast<float*>(packed), s.w_off, s.b_off, s.cin, s.cout, i == 0 ? 1 : 0);
    const int rc = check_launch("pack_weights_kernel");
    if (rc != SPR_OK) return rc;
  }
  return SPR_OK;
}

static size_t stage_out_floats(const Stage& s, int64_t n, int h, int w) {
  const int ho = s.pool ? h / 2 : h, wo = s.pool ? w / 2 : w;
  return static_cast<size_t>(n) * ho * wo * s.cout;
}

extern "C" size_t spr_vgg16_workspace_bytes(const spr_vgg16_plan* plan, int64_t n, int32_t in_h, int32_t in_w) {
  if (!plan || n < 0) return 0;
  // two ping-pong activation buffers, each as large as the largest intermediate tensor
  size_t biggest = 0;
  int h = in_h, w = in_w;
  for (size_t i = 0; i + 1 < plan->stages.size(); ++i) {
    const Stage& s = plan->stages[i];
    const size_t f = stage_out_floats(s, n, h, w);
    if (f > biggest) biggest = f;
    if (s.pool) { h /= 2; w /= 2; }
  }
  return 2 * align_up(biggest * (plan->compute == SPR_F32 ? sizeof(float) : sizeof(uint16_t)), 256);
}

static int vgg_forward(spr_vgg16_plan* plan, const uint8_t* images, int64_t n, int32_t in_h, int32_t in_w,
                       int32_t in_channels, const float* mean3, const float* inv_std3, const void* packed, void* workspace,
                       float* out, int32_t n_taps, const int32_t* tap_convs, float* const* tap_out, spr_stream_t stream,
                       const char* who) {
  if (!plan) { set_error("%s: null plan", who); return SPR_ERR_ARG; }
  if (n < 0 || n > 65535 || in_h < 1 || in_w < 1 || (in_channels != 1 && in_channels != 3)) {
    set_error("%s: bad sizes (n in [0, 65535], in_channels 1 or 3)", who);
    return SPR_ERR_ARG;
  }
  if (n_taps < 0 || (n_taps > 0 && (!tap_convs || !tap_out))) { set_error("%s: bad taps", who); return SPR_ERR_ARG; }
  for (int t = 0; t < n_taps; ++t) {
    if (tap_convs[t] < 1 || tap_convs[t] >= static_cast<int>(plan->stages.size()) || !tap_out[t]) {
      set_error("%s: tap %d names convolution %d: taps are convolutions 1 .. %zu (not the first) and need a buffer", who, t,
                tap_convs[t], plan->stages.size() - 1);
      return SPR_ERR_ARG;
    }
  }
  if (n == 0) return SPR_OK;
  if (!images || !mean3 || !inv_std3 || !packed || !out || (plan->stages.size() > 1 && !workspace)) {
    set_error("%s: null pointer", who);
    return SPR_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* pk = static_cast<const float*>(packed);
  const size_t half = spr_vgg16_workspace_bytes(plan, n, in_h, in_w) / 2;
  float* buf[2] = {static_cast<float*>(workspace),
                   reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + half)};
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kConvLds));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv16_kernel<kF16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kConvLds));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv16_kernel<kBF16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kConvLds));
  int h = in_h, w = in_w;
  const float* cur = nullptr;
  for (size_t i = 0; i < plan->stages.size(); ++i) {
    const Stage& st = plan->stages[i];
    const bool last = i + 1 == plan->stages.size();
    float* dst = last ? out : buf[i & 1];
    float* tap = nullptr;
    for (int t = 0; t < n_taps; ++t)
      if (tap_convs[t] == static_cast<int>(i)) tap = tap_out[t];
    const unsigned tiles = static_cast<unsigned>(ceil_div(h, kTile) * ceil_div(w, kTile));
    if (i == 0 && plan->compute != SPR_F32 && !last) {
      const int rc = launch_first16(plan->compute, images, n, h, w, in_channels, mean3, inv_std3,
                                    reinterpret_cast<const uint16_t*>(pk + st.w_off), pk + st.b_off, st.relu,
                                    reinterpret_cast<uint16_t*>(dst), s);
      if (rc != SPR_OK) return rc;
    } else if (i == 0) {
      hipLaunchKernelGGL(conv_first_kernel, dim3(tiles, static_cast<unsigned>(n)), dim3(kThreads), 0, s, images, h, w,
                         in_channels, mean3[0], mean3[1], mean3[2], inv_std3[0], inv_std3[1], inv_std3[2],
                         pk + st.w_off, pk + st.b_off, st.relu, last ? 1 : 0, dst, plan->compute == SPR_F32 ? 0 : plan->compute);
      const int rc = check_launch("conv_first_kernel");
      if (rc != SPR_OK) return rc;
    } else if (plan->compute != SPR_F32) {
      const dim3 grid(tiles, static_cast<unsigned>(st.cout / kTN), static_cast<unsigned>(n));
      const uint16_t* src16 = reinterpret_cast<const uint16_t*>(cur);
      const uint16_t* w16 = reinterpret_cast<const uint16_t*>(pk + st.w_off);
      if (plan->compute == SPR_F16)
        hipLaunchKernelGGL(conv16_kernel<kF16>, grid, dim3(kThreads), kConvLds, s, src16, h, w, st.cin, st.cout, w16,
                           pk + st.b_off, st.relu, st.pool, last ? 1 : 0, dst, tap);
      else
        hipLaunchKernelGGL(conv16_kernel<kBF16>, grid, dim3(kThreads), kConvLds, s, src16, h, w, st.cin, st.cout, w16,
                           pk + st.b_off, st.relu, st.pool, last ? 1 : 0, dst, tap);
      const int rc = check_launch("conv16_kernel");
      if (rc != SPR_OK) return rc;
    } else {
      hipLaunchKernelGGL(conv_mfma_kernel, dim3(tiles, static_cast<unsigned>(st.cout / kTN), static_cast<unsigned>(n)),
                         dim3(kThreads), kConvLds, s, cur, h, w, st.cin, st.cout, pk + st.w_off, pk + st.b_off, st.relu,
                         st.pool, last ? 1 : 0, dst, tap);
      const int rc = check_launch("conv_mfma_kernel");
      if (rc != SPR_OK) return rc;
    }
    if (st.pool) { h /= 2; w /= 2; }
    cur = dst;
  }
  return SPR_OK;
}

extern "C" int spr_vgg16_forward(spr_vgg16_plan* plan, const uint8_t* images, int64_t n, int32_t in_h, int32_t in_w,
                                 int32_t in_channels, const float* mean3, const float* inv_std3, const void* packed,
                                 void* workspace, float* out, spr_stream_t stream) {
  return vgg_forward(plan, images, n, in_h, in_w, in_channels, mean3, inv_std3, packed, workspace, out, 0, nullptr, nullptr,
                     stream, "spr_vgg16_forward");
}

extern "C" int spr_vgg16_forward_taps(spr_vgg16_plan* plan, const uint8_t* images, int64_t n, int32_t in_h, int32_t in_w,
                                      int32_t in_channels, const float* mean3, const float* inv_std3, const void* packed,
                                      void* workspace, float* out, int32_t n_taps, const int32_t* tap_convs,
                                      float* const* tap_out, spr_stream_t stream) {
  return vgg_forward(plan, images, n, in_h, in_w, in_channels, mean3, inv_std3, packed, workspace, out, n_taps, tap_convs,
                     tap_out, stream, "spr_vgg16_forward_taps");
}
