// Feature extractors on the implicit-GEMM convolution kernel of this file: ResNet50 through layer1 / layer2 / layer3
// (BASELINE.json config 3: "ResNet50 layer3 summed maps"; first part), the EfficientNet B-series and V2 truncations
// (network.py:139-175; spr_effnet_*, second part) and DenseNet_201 (network.py:176-179; spr_densenet_*, third part).
//
// The reference has no ResNet (network.py:121-182 lists VGG, EfficientNet and DenseNet) and its truncation
// `list(model.features.children())[:block]` (network.py:185) would not apply to torchvision's resnet50, which has no
// `.features`; this extractor is therefore BUILD-DEFINED: torchvision's resnet50 v1.5 graph (stride on the 3x3 convolution
// of a bottleneck), truncated after `block` of its top-level children [conv1, bn1, relu, maxpool, layer1, layer2, layer3],
// block = 5 / 6 / 7.  Eval-mode BatchNorm is an affine map per channel and is folded into the preceding convolution by
// the host, so the kernels see convolution + bias only.
//
// Kernels (activations NHWC float32 between layers, NCHW out of the last one, as the NCC prep kernels read them):
//   stem_kernel      7x7 / stride 2 / pad 3, 3 -> 64, with ToTensor / repeat(3) / Normalize fused in front (zero padding
//                    of the NORMALISED tensor) and ReLU behind; plain FMA (K = 147, 3.6 % of the flops).
//   maxpool3_kernel  3x3 / stride 2 / pad 1.
//   conv_gemm_kernel every other convolution (1x1 and 3x3, stride 1 or 2) as an implicit GEMM on the fp32 matrix cores
//                    (v_mfma_f32_16x16x4_f32, exact f32): M = images x output pixels, N = output channels, K = taps x
//                    input channels.  Workgroup = 64 pixels x 64 channels, 4 waves x (16 pixels x 64 channels); per K
//                    chunk of 16 the A tile (gathered rows of 16 contiguous channels, zero fill = padding) and the B
//                    tile (packed filter slab) are staged in LDS from registers loaded one chunk ahead.  Epilogue: bias,
//                    residual add, ReLU.
// Arithmetic: 17.13 GFLOP per 512x256 image through layer3; bound: fp32 MFMA 157 TFLOP/s.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

#include "spr_common.h"

namespace spr {
namespace {

constexpr int kGM = 64, kGN = 64, kGK = 16;  // GEMM tile of a workgroup: pixels x channels x K chunk
constexpr int kGS = 20;                        // LDS row stride (floats) of a 16-float row: 16-byte aligned, 5 quads
                                               // -> the 16 lanes of an MFMA operand read hit different banks

struct RConv {
  int cin, cout, ks, stride;
  int relu;      // ReLU in the epilogue
  int res;       // 0: none, 1: add the block input, 2: add the downsample branch's output
  int role;      // 0 stem, 1 conv1, 2 conv2, 3 conv3, 4 downsample
  size_t w_off, b_off;
};

// ---------------------------------------------------------------- parameter packing
// stem: [tap*3 + c][64]   |   GEMM convs: [cout/64][K/16][n:64][k:16], K index = tap * cin + c
__global__ void __launch_bounds__(kThreads)
rpack_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ packed, size_t w_off,
             size_t b_off, int cin, int cout, int ks, int stem) {
  const int taps = ks * ks;
  const size_t total = static_cast<size_t>(cout) * cin * taps;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * kThreads) {
    const int tap = static_cast<int>(i % taps);  // torch layout [n][c][ky][kx]
    const int c = static_cast<int>((i / taps) % cin);
    const int n = static_cast<int>(i / (static_cast<size_t>(taps) * cin));
    size_t dst;
    if (stem) {
      dst = static_cast<size_t>(tap * 3 + c) * cout + n;
    } else {
      const int k = tap * cin + c;
      const int chunks = taps * cin / kGK;
      dst = ((static_cast<size_t>(n / kGN) * chunks + k / kGK) * kGN + n % kGN) * kGK + k % kGK;
    }
    packed[w_off + dst] = w[i];
  }
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < cout; i += gridDim.x * kThreads) packed[b_off + i] = b[i];
}

// ---------------------------------------------------------------- stem: 7x7 s2 p3, 3 -> 64, pre-processing + ReLU fused
// grid = (tiles of 8x8 output pixels, images); out NHWC [n][Ho][Wo][64]
__global__ void __launch_bounds__(kThreads)
stem_kernel(const uint8_t* __restrict__ images, int H, int W, int in_channels, float m0, float m1, float m2, float s0,
            float s1, float s2, const float* __restrict__ wts, const float* __restrict__ bias, float* __restrict__ out,
            int relu, int kind16) {
  // kind16 != 0 (16-bit plans): the activation is stored rounded to float16 / bfloat16 (rounding is monotonic, so the max
  // pool behind it may take its maximum over the rounded values)
  constexpr int kT = 8, kP = 2 * kT + 5;  // 21 x 21 input patch
  __shared__ float patch[kP * kP * 3];
  __shared__ float wl[147 * 64];
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int tiles_x = ceil_div(Wo, kT);
  const int ty = static_cast<int>(blockIdx.x) / tiles_x, tx = static_cast<int>(blockIdx.x) % tiles_x;
  const int oy0 = ty * kT, ox0 = tx * kT;
  const size_t img = blockIdx.y;
  const int tid = static_cast<int>(threadIdx.x);
  const float mean[3] = {m0, m1, m2}, istd[3] = {s0, s1, s2};
  for (int i = tid; i < 147 * 64; i += kThreads) wl[i] = wts[i];
  for (int i = tid; i < kP * kP; i += kThreads) {
    const int py = i / kP, px = i % kP;
    const int y = 2 * oy0 - 3 + py, x = 2 * ox0 - 3 + px;
    const bool in = y >= 0 && y < H && x >= 0 && x < W;
    for (int c = 0; c < 3; ++c) {
      float v = 0.0f;  // zero padding of the NORMALISED tensor
      if (in) {
        const size_t pix = (img * H + y) * static_cast<size_t>(W) + x;
        const float u = static_cast<float>(in_channels == 1 ? images[pix] : images[pix * 3 + c]);
        v = (u / 255.0f - mean[c]) * istd[c];
      }
      patch[i * 3 + c] = v;
    }
  }
  __syncthreads();
  const int n = tid & 63, part = tid >> 6;  // lane = output channel; wave `part` takes output rows 2 part, 2 part + 1
  const float b = bias[n];
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = b;
  for (int dy = 0; dy < 7; ++dy)
    for (int dx = 0; dx < 7; ++dx)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float wv = wl[((dy * 7 + dx) * 3 + c) * 64 + n];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int py = 2 * (2 * part + i / 8) + dy, px = 2 * (i % 8) + dx;
          acc[i] = fmaf(patch[(py * kP + px) * 3 + c], wv, acc[i]);
        }
      }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int oy = oy0 + 2 * part + i / 8, ox = ox0 + i % 8;
    if (oy < Ho && ox < Wo) {
      const float v = relu ? fmaxf(acc[i], 0.0f) : acc[i];
      const size_t at = ((img * Ho + oy) * static_cast<size_t>(Wo) + ox) * 64 + n;
      if (kind16 == 0) out[at] = v;
      else reinterpret_cast<uint16_t*>(out)[at] = kind16 == SPR_F16 ? round_f16(v) : round_bf16(v);
    }
  }
}

// ---------------------------------------------------------------- 3x3 / stride 2 / pad 1 max pool, NHWC
__global__ void __launch_bounds__(kThreads)
maxpool3_kernel(const float* __restrict__ in, int H, int W, int C, float* __restrict__ out, size_t total, int ldo) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % C);
    size_t p = i / C;
    const int ox = static_cast<int>(p % Wo); p /= Wo;
    const int oy = static_cast<int>(p % Ho);
    const size_t img = p / Ho;
    float m = -3.402823466e38f;  // (padding never wins: every window holds at least one real pixel)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int y = 2 * oy + dy, x = 2 * ox + dx;
        if (y >= 0 && y < H && x >= 0 && x < W) m = fmaxf(m, in[((img * H + y) * static_cast<size_t>(W) + x) * C + c]);
      }
    out[(i / C) * ldo + c] = m;  // ldo: channel stride of the output tensor (>= C)
  }
}

// ---------------------------------------------------------------- implicit-GEMM convolution on fp32 MFMA
// grid = (ceil(M / 64), cout / 64).  in NHWC [n][H][W][cin]; out NHWC [n][Ho][Wo][cout] (or NCHW); res NHWC like out.
template <int KS, int STRIDE>
__global__ void __launch_bounds__(kThreads, 2)
conv_gemm_kernel(const float* __restrict__ in, int n_img, int H, int W, int cin, int cout, const float* __restrict__ wts,
                 const float* __restrict__ bias, const float* __restrict__ res, int relu, int nchw,
                 float* __restrict__ out, const float* __restrict__ in_scale, int cout_real, int lda, int ldc, int c_off,
                 const float* __restrict__ pre_s, const float* __restrict__ pre_t) {
  // lda / ldc: channel strides of the input / NHWC output tensors (>= cin / cout: a convolution may read a prefix of a wider
  // tensor and write a channel range [c_off, c_off + cout_real) of one - DenseNet's concatenation); pre_s / pre_t: per input
  // channel, max(x * s + t, 0) applied while the operand is loaded (BatchNorm + ReLU in FRONT of a 1x1 convolution), or null
  // relu: activation code (0 none, 1 ReLU, 2 SiLU); in_scale: [image][cin] factors on the input (squeeze-excitation), or null;
  // cout_real: channels of an NCHW result when cout is padded (0: all of them)
  __shared__ __attribute__((aligned(16))) float A[kGM * kGS];
  __shared__ __attribute__((aligned(16))) float B[kGN * kGS];
  constexpr int PAD = KS / 2;
  const int Ho = (H + 2 * PAD - KS) / STRIDE + 1, Wo = (W + 2 * PAD - KS) / STRIDE + 1;
  const long long M = static_cast<long long>(n_img) * Ho * Wo;
  const int tid = static_cast<int>(threadIdx.x);
  const int wave = tid >> 6, lane = tid & 63;
  const int p = lane & 15, q = lane >> 4;  // MFMA lane coordinates: row/col index, k index
  const int cb = static_cast<int>(blockIdx.y);
  const long long m0 = static_cast<long long>(blockIdx.x) * kGM;
  const int cchunks = cin / kGK, chunks = KS * KS * cchunks;

  // this thread stages quarter `sq` (4 floats) of row `sr` of both tiles
  const int sr = tid >> 2, sq = tid & 3;
  const long long pm = m0 + sr;  // pixel of the A row
  const bool pm_ok = pm < M;
  int py = 0, px = 0;
  size_t pimg = 0;
  if (pm_ok) {
    px = static_cast<int>(pm % Wo);
    py = static_cast<int>((pm / Wo) % Ho);
    pimg = static_cast<size_t>(pm / (static_cast<long long>(Wo) * Ho));
  }
  const float* wbase = wts + static_cast<size_t>(cb) * chunks * (kGN * kGK) + sr * kGK + sq * 4;

  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 ra, rb;
  auto request = [&](int ch) {
    const int tap = ch / cchunks, cc = ch - tap * cchunks;
    const int dy = tap / KS, dx = tap - dy * KS;
    const int y = py * STRIDE + dy - PAD, x = px * STRIDE + dx - PAD;
    ra = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pm_ok && y >= 0 && y < H && x >= 0 && x < W)
      ra = *reinterpret_cast<const float4*>(in + ((pimg * H + y) * static_cast<size_t>(W) + x) * lda + cc * kGK + sq * 4);
    if (pre_s) {
      const float4 ps = *reinterpret_cast<const float4*>(pre_s + cc * kGK + sq * 4);
      const float4 pt = *reinterpret_cast<const float4*>(pre_t + cc * kGK + sq * 4);
      ra.x = fmaxf(fmaf(ra.x, ps.x, pt.x), 0.f); ra.y = fmaxf(fmaf(ra.y, ps.y, pt.y), 0.f);
      ra.z = fmaxf(fmaf(ra.z, ps.z, pt.z), 0.f); ra.w = fmaxf(fmaf(ra.w, ps.w, pt.w), 0.f);
    }
    if (in_scale) {
      const float4 sc = *reinterpret_cast<const float4*>(in_scale + pimg * cin + cc * kGK + sq * 4);
      ra.x *= sc.x; ra.y *= sc.y; ra.z *= sc.z; ra.w *= sc.w;
    }
    rb = *reinterpret_cast<const float4*>(wbase + static_cast<size_t>(ch) * (kGN * kGK));
  };
  request(0);
  for (int ch = 0; ch < chunks; ++ch) {
    __syncthreads();  // the previous chunk's fragments are consumed
    *reinterpret_cast<float4*>(A + sr * kGS + sq * 4) = ra;
    *reinterpret_cast<float4*>(B + sr * kGS + sq * 4) = rb;
    __syncthreads();
    if (ch + 1 < chunks) request(ch + 1);
    // the k index of MFMA step j is {4 q + j}: any partition of the 16 works as long as A and B agree
    const float4 a = *reinterpret_cast<const float4*>(A + (wave * 16 + p) * kGS + q * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 b = *reinterpret_cast<const float4*>(B + (j * 16 + p) * kGS + q * 4);
      acc[j] = mfma_f32_16x16x4(a.x, b.x, acc[j]);
      acc[j] = mfma_f32_16x16x4(a.y, b.y, acc[j]);
      acc[j] = mfma_f32_16x16x4(a.z, b.z, acc[j]);
      acc[j] = mfma_f32_16x16x4(a.w, b.w, acc[j]);
    }
  }
  // ---- epilogue: lane (q, p) owns pixels m0 + 16 wave + 4 q + r (r = 0..3), channel cb*64 + 16 j + p
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long m = m0 + wave * 16 + 4 * q + r;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = cb * kGN + j * 16 + p;
      float v = acc[j][r] + bias[ch];
      if (relu == 2) v = v / (1.0f + expf(-v));  // SiLU, in FRONT of the residual sum (EfficientNet blocks)
      if (res) v += res[static_cast<size_t>(m) * cout + ch];
      if (relu == 1) v = fmaxf(v, 0.0f);         // ReLU, behind it (ResNet bottlenecks)
      if (nchw) {
        const int creal = cout_real ? cout_real : cout;
        if (ch >= creal) continue;
        const int ox = static_cast<int>(m % Wo), oy = static_cast<int>((m / Wo) % Ho);
        const size_t img = static_cast<size_t>(m / (static_cast<long long>(Wo) * Ho));
        out[((img * creal + ch) * Ho + oy) * static_cast<size_t>(Wo) + ox] = v;
      } else if (!cout_real || ch < cout_real) {
        out[static_cast<size_t>(m) * ldc + c_off + ch] = v;
      }
    }
  }
}

// ---------------------------------------------------------------- implicit-GEMM convolution on the 16-bit matrix cores
// spr_resnet_plan_create_ex(SPR_F16 | SPR_BF16): the same GEMM view with float16 / bfloat16 operands and f32 accumulation
// (v_mfma_f32_16x16x32: K = 32 per instruction).  The matrix work per byte staged is 16 x shorter than on the f32 cores, so
// the tile is larger: workgroup = 128 pixels x 64 channels x a K chunk of 64 (two MFMA k-steps), 4 waves x (32 pixels x 64
// channels) = 16 MFMAs per wave and chunk; the A tile (128 gathered rows of 64 contiguous channels = 128 bytes each) and the
// B tile (64 filter rows) are loaded into registers one chunk ahead and written to LDS behind the barrier.  LDS rows are 128
// bytes = eight 16-byte slots; slot s of row r sits at s ^ ((r >> 1) & 7), so the 16 lanes x 4 k-groups of an operand read
// fall into different banks.  Activations between layers: NHWC, rounded to the 16-bit type; the residual operand is such a
// stored activation; bias / residual sum / ReLU in f32; the last layer writes float32 NCHW.
constexpr int kHM = 128, kHN = 64, kHK = 64;
constexpr int kHRowDw = 32;  // dwords per LDS row (128 bytes)

// GEMM convs of a 16-bit plan: [cout/64][K/64][n:64][k:64] float16 / bfloat16, K index = tap * cin + c; bias f32
template <int KIND>
__global__ void __launch_bounds__(kThreads)
rpack16_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ packed, size_t w_off,
               size_t b_off, int cin, int cout, int ks) {
  uint16_t* dst16 = reinterpret_cast<uint16_t*>(packed + w_off);
  const int taps = ks * ks;
  const size_t total = static_cast<size_t>(cout) * cin * taps;
  const int chunks = taps * cin / kHK;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * kThreads) {
    const int tap = static_cast<int>(i % taps);
    const int c = static_cast<int>((i / taps) % cin);
    const int n = static_cast<int>(i / (static_cast<size_t>(taps) * cin));
    const int k = tap * cin + c;
    dst16[((static_cast<size_t>(n / kHN) * chunks + k / kHK) * kHN + n % kHN) * kHK + k % kHK] = round16<KIND>(w[i]);
  }
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < cout; i += gridDim.x * kThreads) packed[b_off + i] = b[i];
}

// 3x3 / stride 2 / pad 1 max pool of the stem's (rounded, post-ReLU: non-negative) 16-bit NHWC output into the 16-bit NHWC
// tensor layer1 reads: eight channels (16 bytes) per work-item; non-negative float16 / bfloat16 values order like their bit
// patterns, so the maximum is taken on the 16-bit integers
__global__ void __launch_bounds__(kThreads)
maxpool3_16_kernel(const uint16_t* __restrict__ in, int H, int W, int C, uint16_t* __restrict__ out, size_t total8) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, c8 = C / 8;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total8;
       i += static_cast<size_t>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % c8) * 8;
    size_t p = i / c8;
    const int ox = static_cast<int>(p % Wo); p /= Wo;
    const int oy = static_cast<int>(p % Ho);
    const size_t img = p / Ho;
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int y = 2 * oy + dy, x = 2 * ox + dx;
        if (y < 0 || y >= H || x < 0 || x >= W) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(in + ((img * H + y) * static_cast<size_t>(W) + x) * C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t h = (v[e >> 1] >> (16 * (e & 1))) & 0xffffu;
          m[e] = h > m[e] ? h : m[e];
        }
      }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = m[2 * e] | (m[2 * e + 1] << 16);
    *reinterpret_cast<u32x4*>(out + i * 8) = o;
  }
}

// ---------------------------------------------------------------- the stem on the 16-bit matrix cores (16-bit plans)
// The plain FMA stem_kernel runs at half of the (unpacked) f32 vector peak and is LDS-bound on broadcast reads: 0.5 ms per 32
// images, a fifth of a 16-bit forward pass.  Here the 7x7 / stride 2 convolution is a GEMM of 128 output pixels (8 x 16) x 64
// channels x K = 147 taps-and-planes padded to 160 = five v_mfma_f32_16x16x32 k-steps: the normalised input patch (ToTensor,
// repeat(3), Normalize; zero outside the image) is rounded to the 16-bit type into LDS, every work-item gathers ten 16-byte
// pieces (8 consecutive k each) of the im2col tile from it through an offset table, and the operand tiles lie K-major
// ([16-byte slot][row]) so that the sixteen rows x four k-groups of a fragment read fall into different banks as they are.
// Weights: [k / 8][n: 64][8] 16-bit, zero for k >= 147 (rstem16_pack_kernel).  Output: NHWC 16-bit, ReLU applied.
// The same kernel with KS = 3, STRIDE = 1 (K = 27 padded to 32: one k-step) is the first convolution of the plain VGGs in their
// 16-bit plans (vgg_conv.hip calls launch_first16).
constexpr int kSTH = 8, kSTW = 16;  // output pixels per workgroup
template <int KS, int STRIDE>
struct First16 {
  static constexpr int TAPS = KS * KS, KREAL = TAPS * 3, K = (KREAL + 31) / 32 * 32, SLOTS = K / 8;
  static constexpr int PH = STRIDE * kSTH + KS - STRIDE, PW = STRIDE * kSTW + KS - STRIDE, PAD = KS / 2;
};

template <int KIND, int KS>
__global__ void __launch_bounds__(kThreads)
rstem16_pack_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ packed, size_t w_off,
                    size_t b_off) {
  using F = First16<KS, 1>;
  uint16_t* dst = reinterpret_cast<uint16_t*>(packed + w_off);
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < F::K * 64; i += gridDim.x * kThreads) {
    const int k = i / 64, n = i % 64;  // k = tap * 3 + c
    float v = 0.0f;
    if (k < F::KREAL) v = w[(static_cast<size_t>(n) * 3 + k % 3) * F::TAPS + k / 3];  // torch layout [n][c][ky][kx]
    dst[(static_cast<size_t>(k / 8) * 64 + n) * 8 + k % 8] = round16<KIND>(v);
  }
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < 64; i += gridDim.x * kThreads) packed[b_off + i] = b[i];
}

// grid = (tiles of 8 x 16 output pixels, images)
template <int KIND, int KS, int STRIDE>
__global__ void __launch_bounds__(kThreads, 2)
stem16_kernel(const uint8_t* __restrict__ images, int H, int W, int in_channels, float m0, float m1, float m2, float s0,
              float s1, float s2, const uint16_t* __restrict__ wts, const float* __restrict__ bias, uint16_t* __restrict__ out,
              int relu) {
  using F = First16<KS, STRIDE>;
  constexpr int kSK = F::K, kSSlots = F::SLOTS, kSPH = F::PH, kSPW = F::PW;
  constexpr int kHT = 68;
  constexpr int kPatchElems = kSPH * kSPW * 3;                  // stem: 21 x 37 x 3
  constexpr int kPatchBytes = (kPatchElems * 2 + 2 + 15) / 16 * 16;  // + one zero element the padded k read
  constexpr int kABytes = kSSlots * 128 * 16 > 128 * kHT * 4 ? kSSlots * 128 * 16 : 128 * kHT * 4;  // (or the f32 output tile)
  constexpr int kBBytes = kSSlots * 64 * 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kABytes + kBBytes + kPatchBytes + kSK * 2];
  uint32_t* A = reinterpret_cast<uint32_t*>(lds);
  uint32_t* B = reinterpret_cast<uint32_t*>(lds + kABytes);
  uint16_t* patch = reinterpret_cast<uint16_t*>(lds + kABytes + kBBytes);
  uint16_t* koff = reinterpret_cast<uint16_t*>(lds + kABytes + kBBytes + kPatchBytes);
  const int Ho = (H + 2 * F::PAD - KS) / STRIDE + 1, Wo = (W + 2 * F::PAD - KS) / STRIDE + 1;
  const int tiles_x = ceil_div(Wo, kSTW);
  const int ty = static_cast<int>(blockIdx.x) / tiles_x, tx = static_cast<int>(blockIdx.x) % tiles_x;
  const int oy0 = ty * kSTH, ox0 = tx * kSTW;
  const size_t img = blockIdx.y;
  const int tid = static_cast<int>(threadIdx.x);
  const int wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  const float mean[3] = {m0, m1, m2}, istd[3] = {s0, s1, s2};
  // the weights of this layer (20 KB, L2-resident) and the offset table: k -> element of the patch, relative to the pixel's
  // window origin; the padded k point at the zero element behind the patch
  for (int i = tid; i < kBBytes / 16; i += kThreads) reinterpret_cast<float4*>(B)[i] = reinterpret_cast<const float4*>(wts)[i];
  if (tid < kSK) {
    const int tap = tid / 3, c = tid % 3, dy = tap / KS, dx = tap % KS;
    koff[tid] = tid < F::KREAL ? static_cast<uint16_t>((dy * kSPW + dx) * 3 + c) : static_cast<uint16_t>(0xffff);
  }
  for (int i = tid; i < kSPH * kSPW; i += kThreads) {
    const int py = i / kSPW, px = i % kSPW;
    const int y = STRIDE * oy0 - F::PAD + py, x = STRIDE * ox0 - F::PAD + px;
    const bool in = y >= 0 && y < H && x >= 0 && x < W;
    for (int c = 0; c < 3; ++c) {
      float v = 0.0f;  // zero padding of the NORMALISED tensor
      if (in) {
        const size_t pix = (img * H + y) * static_cast<size_t>(W) + x;
        const float u = static_cast<float>(in_channels == 1 ? images[pix] : images[pix * 3 + c]);
        v = (u / 255.0f - mean[c]) * istd[c];
      }
      patch[i * 3 + c] = round16<KIND>(v);
    }
  }
  if (tid == 0) patch[kPatchElems] = 0;
  __syncthreads();
  {  // im2col: row = output pixel (8 x 16, row-major), ten 16-byte pieces per work-item
    const int row = tid & 127, half = tid >> 7;
    const int base = ((row >> 4) * STRIDE * kSPW + (row & 15) * STRIDE) * 3;
#pragma unroll
    for (int j = 0; j < kSSlots / 2; ++j) {
      const int sl = half * (kSSlots / 2) + j;
      const u32x4 ko = *reinterpret_cast<const u32x4*>(koff + 8 * sl);
      u32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned k0 = ko[e] & 0xffffu, k1 = ko[e] >> 16;
        const unsigned a0 = patch[k0 == 0xffffu ? kPatchElems : base + static_cast<int>(k0)];
        const unsigned a1 = patch[k1 == 0xffffu ? kPatchElems : base + static_cast<int>(k1)];
        v[e] = a0 | (a1 << 16);
      }
      *reinterpret_cast<u32x4*>(A + (sl * 128 + row) * 4) = v;
    }
  }
  __syncthreads();
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < kSK / 32; ++ks) {
    u32x4 a[2], b[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const u32x4*>(A + ((ks * 4 + q) * 128 + wave * 32 + i * 16 + p) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const u32x4*>(B + ((ks * 4 + q) * 64 + j * 16 + p) * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = KIND == SPR_F16 ? mfma_f16_16x16x32(a[i], b[j], acc[i][j]) : mfma_bf16_16x16x32(a[i], b[j], acc[i][j]);
  }
  __syncthreads();  // the A tile is consumed: the f32 output tile takes its place
  float* T = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float bv = bias[j * 16 + p];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(wave * 32 + i * 16 + 4 * q + r) * kHT + j * 16 + p] = acc[i][j][r] + bv;
  }
  __syncthreads();
  const int sr = tid >> 3, ss = tid & 7;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = sr + 32 * k;
    const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
    if (oy >= Ho || ox >= Wo) continue;
    const float4 lo = *reinterpret_cast<const float4*>(T + row * kHT + ss * 8);
    const float4 hi = *reinterpret_cast<const float4*>(T + row * kHT + ss * 8 + 4);
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // relu: activation code (0 none, 1 ReLU, 2 SiLU)
      const float a0 = relu == 1 ? fmaxf(v[2 * e], 0.0f) : relu == 2 ? v[2 * e] / (1.0f + expf(-v[2 * e])) : v[2 * e];
      const float a1 = relu == 1 ? fmaxf(v[2 * e + 1], 0.0f) : relu == 2 ? v[2 * e + 1] / (1.0f + expf(-v[2 * e + 1])) : v[2 * e + 1];
      o[e] = static_cast<uint32_t>(round16<KIND>(a0)) | (static_cast<uint32_t>(round16<KIND>(a1)) << 16);
    }
    *reinterpret_cast<u32x4*>(out + ((img * Ho + oy) * static_cast<size_t>(Wo) + ox) * 64 + ss * 8) = o;
  }
}

// grid = (ceil(M / 128), cout / BN).  in / res / out: NHWC 16-bit with cin / cout channels; out32: float32 NCHW (last layer).
// BN = 64: four waves x (32 pixels x 64 channels); BN = 128: 2 x 2 waves x (64 pixels x 64 channels) - twice the matrix
// work per byte staged (these GEMMs run against the L2 -> CU bandwidth, not against the matrix cores: a 128 x 64 x 64 chunk
// is 43 flop per staged byte, a 128 x 128 x 64 one 64).
template <int KS, int STRIDE, int KIND, int BN>
__global__ void __launch_bounds__(kThreads, BN == 128 ? 2 : 3)
conv_gemm16_kernel(const uint16_t* __restrict__ in, int n_img, int H, int W, int cin, int cout,
                   const uint16_t* __restrict__ wts, const float* __restrict__ bias, const uint16_t* __restrict__ res,
                   int relu, uint16_t* __restrict__ out, float* __restrict__ out32, const float* __restrict__ in_scale,
                   int cout_real) {
  // relu: activation code (0 none, 1 ReLU behind the residual sum: ResNet bottlenecks, 2 SiLU in FRONT of it: EfficientNet
  // blocks); in_scale: [image][cin] f32 factors on the input (squeeze-excitation), applied while the operand is staged and
  // rounded again, or null; cout_real: channels of a float32 NCHW result when cout is padded (0: all of them)
  // one LDS array: the A and B operand tiles in the main loop, the f32 output tile [128][kHT] (64 channels at a time) in the
  // epilogue
  constexpr int kHT = 68;  // row stride (floats) of the output tile: 16-byte aligned, rows 4 apart half a bank row apart
  constexpr int kLdsDw = kHM * kHT > (kHM + BN) * kHRowDw ? kHM * kHT : (kHM + BN) * kHRowDw;
  __shared__ __attribute__((aligned(16))) uint32_t lds16[kLdsDw];
  uint32_t* A = lds16;
  uint32_t* B = lds16 + kHM * kHRowDw;
  constexpr int PAD = KS / 2;
  constexpr int MI = BN == 128 ? 4 : 2;  // 16-pixel blocks per wave
  constexpr int BK = BN / 32;            // 16-byte pieces of the B tile per work-item
  const int Ho = (H + 2 * PAD - KS) / STRIDE + 1, Wo = (W + 2 * PAD - KS) / STRIDE + 1;
  const long long M = static_cast<long long>(n_img) * Ho * Wo;
  const int tid = static_cast<int>(threadIdx.x);
  const int wave = tid >> 6, lane = tid & 63;
  const int p = lane & 15, q = lane >> 4;
  const int cb = static_cast<int>(blockIdx.y);
  const long long m0 = static_cast<long long>(blockIdx.x) * kHM;
  const int cchunks = cin / kHK, chunks = KS * KS * cchunks;
  const int wm = BN == 128 ? (wave >> 1) * 64 : wave * 32;  // first pixel row / first channel of this wave's part of the tile
  const int wn = BN == 128 ? (wave & 1) * 64 : 0;

  // staging role: 16-byte slot `ss` of rows sr + 32 k (A: k = 0..3, B: k = 0 .. BK - 1)
  const int sr = tid >> 3, ss = tid & 7;
  int ay[4], ax[4], aimg[4];
  long long abase[4];  // element offset of pixel (img, 0, 0); negative marks a row beyond M
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long pm = m0 + sr + 32 * k;
    if (pm < M) {
      const int px = static_cast<int>(pm % Wo), py = static_cast<int>((pm / Wo) % Ho);
      const long long pimg = pm / (static_cast<long long>(Wo) * Ho);
      ay[k] = py * STRIDE - PAD; ax[k] = px * STRIDE - PAD;
      abase[k] = pimg * H * static_cast<long long>(W) * cin;
      aimg[k] = static_cast<int>(pimg);
    } else {
      ay[k] = ax[k] = 0; abase[k] = -1; aimg[k] = 0;
    }
  }
  // packed weights: [cout / 64][chunk][n: 64][k: 64]; B tile row r belongs to 64-channel block cb * (BN / 64) + r / 64
  const uint16_t* wbase = wts + ss * 8;
  const size_t wblock = static_cast<size_t>(chunks) * (kHN * kHK);

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[BK];  // (native vectors: arrays of HIP's float4 struct stayed in scratch memory and made the prefetch synchronous)
  auto request = [&](int ch) {
    const int tap = ch / cchunks, cc = ch - tap * cchunks;
    const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = ay[k] + dy, x = ax[k] + dx;
      ra[k] = u32x4{0u, 0u, 0u, 0u};
      if (abase[k] >= 0 && y >= 0 && y < H && x >= 0 && x < W) {
        ra[k] = *reinterpret_cast<const u32x4*>(in + abase[k] + (static_cast<long long>(y) * W + x) * cin + cc * kHK + ss * 8);
        if (in_scale) {  // x * factor, rounded to the operand type again
          const float* sc = in_scale + static_cast<size_t>(aimg[k]) * cin + cc * kHK + ss * 8;
          const float4 s0 = *reinterpret_cast<const float4*>(sc), s1 = *reinterpret_cast<const float4*>(sc + 4);
          const float f[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
          u32x4 r;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = value16<KIND>(static_cast<uint16_t>(ra[k][e] & 0xffffu)) * f[2 * e];
            const float hi = value16<KIND>(static_cast<uint16_t>(ra[k][e] >> 16)) * f[2 * e + 1];
            r[e] = static_cast<uint32_t>(round16<KIND>(lo)) | (static_cast<uint32_t>(round16<KIND>(hi)) << 16);
          }
          ra[k] = r;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < BK; ++k) {
      const int r = sr + 32 * k;
      rb[k] = *reinterpret_cast<const u32x4*>(wbase + (static_cast<size_t>(cb) * (BN / 64) + r / 64) * wblock +
                                               (static_cast<size_t>(ch) * kHN + r % 64) * kHK);
    }
  };
  auto slot = [](int row, int s) { return (s ^ ((row >> 1) & 7)) << 2; };  // dword offset of 16-byte slot s inside row `row`
  request(0);
  for (int ch = 0; ch < chunks; ++ch) {
    __syncthreads();  // the previous chunk's fragments are consumed
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(A + (sr + 32 * k) * kHRowDw + slot(sr + 32 * k, ss)) = ra[k];
#pragma unroll
    for (int k = 0; k < BK; ++k) *reinterpret_cast<u32x4*>(B + (sr + 32 * k) * kHRowDw + slot(sr + 32 * k, ss)) = rb[k];
    __syncthreads();
    if (ch + 1 < chunks) request(ch + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 a[MI], b[4];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm + i * 16 + p;
        a[i] = *reinterpret_cast<const u32x4*>(A + row * kHRowDw + slot(row, ks * 4 + q));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn + j * 16 + p;
        b[j] = *reinterpret_cast<const u32x4*>(B + row * kHRowDw + slot(row, ks * 4 + q));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = KIND == SPR_F16 ? mfma_f16_16x16x32(a[i], b[j], acc[i][j]) : mfma_bf16_16x16x32(a[i], b[j], acc[i][j]);
    }
  }
  // ---- epilogue.  Lane (q, p) owns pixels wm + 16 i + 4 q + r, channels wn + 16 j + p of the tile: scattered 2-byte stores
  // from there would touch 32-byte pieces of 4 rows per instruction.  The accumulators (+ bias) go through LDS as an f32 tile
  // of 64 channels at a time instead, and leave in the layout of the destination: NHWC 16-bit rows as 16-byte pieces of 8
  // channels (the residual operand is read the same way), NCHW float32 as runs of consecutive pixels of one channel.
  float* T = reinterpret_cast<float*>(lds16);
#pragma unroll
  for (int h = 0; h < BN / 64; ++h) {
    __syncthreads();  // the operand tiles (or the previous half of the output tile) are consumed
    const int cbase = cb * BN + h * 64;  // first channel of this half
    if (wn == h * 64) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bv = bias[cbase + j * 16 + p];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) T[(wm + i * 16 + 4 * q + r) * kHT + j * 16 + p] = acc[i][j][r] + bv;
      }
    }
    __syncthreads();
    if (out32) {
      // thread = (pixel row of the tile, half of the channels): for one channel, 128 consecutive work-items store 128
      // consecutive pixels of its plane
      const int row = tid & 127, c0 = tid >> 7;
      const long long m = m0 + row;
      if (m < M) {
        const size_t plane = static_cast<size_t>(Ho) * Wo;
        const size_t img = static_cast<size_t>(m / static_cast<long long>(plane));
        const size_t pix = static_cast<size_t>(m - static_cast<long long>(img) * plane);
        const int creal = cout_real ? cout_real : cout;
#pragma unroll 4
        for (int k = 0; k < 32; ++k) {
          const int c = 2 * k + c0, chn = cbase + c;
          if (chn >= creal) continue;
          float v = T[row * kHT + c];
          if (relu == 2) v = v / (1.0f + expf(-v));
          if (res) v += value16<KIND>(res[static_cast<size_t>(m) * cout + chn]);
          if (relu == 1) v = fmaxf(v, 0.0f);
          out32[(img * creal + chn) * plane + pix] = v;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = sr + 32 * k;  // (the staging role again: 16-byte piece ss of rows sr + 32 k)
        const long long m = m0 + row;
        if (m >= M) continue;
        const float4 lo = *reinterpret_cast<const float4*>(T + row * kHT + ss * 8);
        const float4 hi = *reinterpret_cast<const float4*>(T + row * kHT + ss * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const size_t at = static_cast<size_t>(m) * cout + cbase + ss * 8;
        if (relu == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
        }
        if (res) {
          const u32x4 rv = *reinterpret_cast<const u32x4*>(res + at);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += value16<KIND>(static_cast<uint16_t>(rv[e >> 1] >> (16 * (e & 1))));
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = relu == 1 ? fmaxf(v[2 * e], 0.0f) : v[2 * e], a1 = relu == 1 ? fmaxf(v[2 * e + 1], 0.0f) : v[2 * e + 1];
          o[e] = static_cast<uint32_t>(round16<KIND>(a0)) | (static_cast<uint32_t>(round16<KIND>(a1)) << 16);
        }
        *reinterpret_cast<u32x4*>(out + at) = o;
      }
    }
  }
}

// ================================================================ EfficientNetV2 (network.py:163-175) building blocks
// Activations NHWC float32 with the channel count padded to a multiple of 64 (the GEMM tile; padded channels hold zeros:
// zero weights and biases, SiLU(0) = 0); eval-mode BatchNorm folded into the convolutions by the host.

// uint8 grey [n][H][W] (repeated to 3, network.py:60-71) or RGB [n][H][W][3] -> normalised NHWC with 16 channels (3 + zeros)
__global__ void __launch_bounds__(kThreads)
enet_input_kernel(const uint8_t* __restrict__ images, size_t pixels, int in_channels, float m0, float m1, float m2, float s0,
                  float s1, float s2, float* __restrict__ out) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < pixels;
       i += static_cast<size_t>(gridDim.x) * kThreads) {
    float v[3];
    for (int c = 0; c < 3; ++c) v[c] = static_cast<float>(images[in_channels == 3 ? i * 3 + c : i]) * (1.0f / 255.0f);
    float4* o = reinterpret_cast<float4*>(out + i * 16);
    o[0] = float4{(v[0] - m0) * s0, (v[1] - m1) * s1, (v[2] - m2) * s2, 0.0f};
    o[1] = o[2] = o[3] = float4{0.f, 0.f, 0.f, 0.f};
  }
}

// depthwise ks x ks (3 or 5), stride 1 or 2, pad ks / 2, + bias + SiLU.  One work-item = four channels of one output pixel.
// weights [tap][C] (channels contiguous), C a multiple of 64
__global__ void __launch_bounds__(kThreads)
enet_dw_kernel(const float* __restrict__ in, int n_img, int H, int W, int C, int stride, int ks, const float* __restrict__ wts,
               const float* __restrict__ bias, float* __restrict__ out) {
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1, pad = ks / 2;
  const int c4 = C / 4;
  const size_t total = static_cast<size_t>(n_img) * Ho * Wo * c4;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % c4) * 4;
    size_t p = i / c4;
    const int ox = static_cast<int>(p % Wo); p /= Wo;
    const int oy = static_cast<int>(p % Ho);
    const size_t img = p / Ho;
    float4 acc = *reinterpret_cast<const float4*>(bias + c);
    for (int dy = 0; dy < ks; ++dy)
      for (int dx = 0; dx < ks; ++dx) {
        const int y = oy * stride + dy - pad, x = ox * stride + dx - pad;
        if (y < 0 || y >= H || x < 0 || x >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(in + ((img * H + y) * static_cast<size_t>(W) + x) * C + c);
        const float4 w = *reinterpret_cast<const float4*>(wts + static_cast<size_t>(dy * ks + dx) * C + c);
        acc.x = fmaf(v.x, w.x, acc.x); acc.y = fmaf(v.y, w.y, acc.y); acc.z = fmaf(v.z, w.z, acc.z); acc.w = fmaf(v.w, w.w, acc.w);
      }
    float4 o;
    o.x = acc.x / (1.0f + expf(-acc.x)); o.y = acc.y / (1.0f + expf(-acc.y));
    o.z = acc.z / (1.0f + expf(-acc.z)); o.w = acc.w / (1.0f + expf(-acc.w));
    *reinterpret_cast<float4*>(out + i * 4) = o;
  }
}

// squeeze-excitation, step 1: mean over the pixels.  grid = (C / 64, images)
__global__ void __launch_bounds__(kThreads)
enet_pool_kernel(const float* __restrict__ in, int HW, int C, float* __restrict__ pooled) {
  __shared__ float part[4][64];
  const int tid = static_cast<int>(threadIdx.x), c = tid & 63, r = tid >> 6;
  const size_t img = blockIdx.y;
  const float* base = in + img * static_cast<size_t>(HW) * C + static_cast<size_t>(blockIdx.x) * 64 + c;
  float s = 0.0f;
  for (int p = r; p < HW; p += 4) s += base[static_cast<size_t>(p) * C];
  part[r][c] = s;
  __syncthreads();
  if (r == 0) pooled[img * C + blockIdx.x * 64 + c] = (part[0][c] + part[1][c] + part[2][c] + part[3][c]) / static_cast<float>(HW);
}

// step 2: scale[img][c] = sigmoid(W2 SiLU(W1 pooled[img] + b1) + b2); w1 [sq][C], w2 [C][sq] (C padded, sq real).  Two small
// kernels with one unit of work per (image, output): a workgroup per image walking its outputs one after another paid a
// trip to memory per output (measured: 76 - 93 us per call at C = 1056, a quarter of a 16-bit forward pass).
// first layer: one WAVE per (image, hidden unit j): a dot product over C, lanes sweep the channels
__global__ void __launch_bounds__(kThreads)
enet_fc1_kernel(const float* __restrict__ pooled, int n_img, int C, int sq, const float* __restrict__ w1,
                const float* __restrict__ b1, float* __restrict__ hid) {
  const int tid = static_cast<int>(threadIdx.x), lane = tid & 63;
  const long long unit = static_cast<long long>(blockIdx.x) * (kThreads / 64) + (tid >> 6);
  if (unit >= static_cast<long long>(n_img) * sq) return;  // (whole waves leave: no barrier below)
  const int img = static_cast<int>(unit / sq), j = static_cast<int>(unit - static_cast<long long>(img) * sq);
  const float* row = w1 + static_cast<size_t>(j) * C;
  const float* pv = pooled + static_cast<size_t>(img) * C;
  float s0 = 0.0f, s1 = 0.0f;
  int c = lane;
  for (; c + 64 < C; c += 128) {
    s0 = fmaf(row[c], pv[c], s0);
    s1 = fmaf(row[c + 64], pv[c + 64], s1);
  }
  if (c < C) s0 = fmaf(row[c], pv[c], s0);
  float s = s0 + s1;
  for (int m = 32; m >= 1; m >>= 1) s += shfl_xor(s, m);
  if (lane == 0) {
    const float v = b1[j] + s;
    hid[static_cast<size_t>(img) * sq + j] = v / (1.0f + expf(-v));
  }
}
// second layer, grid = (blocks of 64 channels, images): four work-items per channel take every fourth hidden unit
__global__ void __launch_bounds__(kThreads)
enet_fc2_kernel(const float* __restrict__ hid, int C, int sq, const float* __restrict__ w2, const float* __restrict__ b2,
                float* __restrict__ scale) {
  __shared__ float part[4][64];
  const int tid = static_cast<int>(threadIdx.x), cl = tid & 63, r = tid >> 6;
  const size_t img = blockIdx.y;
  const int c = static_cast<int>(blockIdx.x) * 64 + cl;
  const float* h = hid + img * sq;
  float s = 0.0f;
  if (c < C) {
    const float* row = w2 + static_cast<size_t>(c) * sq;
    for (int j = r; j < sq; j += 4) s = fmaf(row[j], h[j], s);
  }
  part[r][cl] = s;
  __syncthreads();
  if (r == 0 && c < C) {
    const float t = b2[c] + ((part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]));
    scale[img * C + c] = 1.0f / (1.0f + expf(-t));
  }
}
// both layers; `hid` = n_img * sq floats of scratch
static int launch_enet_fc(const float* pooled, int64_t n, int C, int sq, const float* w1, const float* b1, const float* w2,
                          const float* b2, float* hid, float* scale, hipStream_t s) {
  const long long units = static_cast<long long>(n) * sq;
  hipLaunchKernelGGL(enet_fc1_kernel, dim3(static_cast<unsigned>((units + 3) / 4)), dim3(kThreads), 0, s, pooled,
                     static_cast<int>(n), C, sq, w1, b1, hid);
  int rc = check_launch("enet_fc1_kernel");
  if (rc != SPR_OK) return rc;
  hipLaunchKernelGGL(enet_fc2_kernel, dim3(static_cast<unsigned>(ceil_div(C, 64)), static_cast<unsigned>(n)), dim3(kThreads), 0, s,
                     hid, C, sq, w2, b2, scale);
  return check_launch("enet_fc2_kernel");
}

// ---- 16-bit plans (spr_effnet_plan_create_ex): activations NHWC float16 / bfloat16, the same padding to 64 channels
// depthwise ks x ks + bias + SiLU: eight channels (16 bytes) of one output pixel per work-item; weights / bias f32 as above
template <int KIND>
__global__ void __launch_bounds__(kThreads)
enet_dw16_kernel(const uint16_t* __restrict__ in, int n_img, int H, int W, int C, int stride, int ks,
                 const float* __restrict__ wts, const float* __restrict__ bias, uint16_t* __restrict__ out) {
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1, pad = ks / 2;
  const int c8 = C / 8;
  const size_t total = static_cast<size_t>(n_img) * Ho * Wo * c8;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % c8) * 8;
    size_t p = i / c8;
    const int ox = static_cast<int>(p % Wo); p /= Wo;
    const int oy = static_cast<int>(p % Ho);
    const size_t img = p / Ho;
    float acc[8];
    {
      const float4 b0 = *reinterpret_cast<const float4*>(bias + c), b1 = *reinterpret_cast<const float4*>(bias + c + 4);
      acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
    }
    for (int dy = 0; dy < ks; ++dy)
      for (int dx = 0; dx < ks; ++dx) {
        const int y = oy * stride + dy - pad, x = ox * stride + dx - pad;
        if (y < 0 || y >= H || x < 0 || x >= W) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(in + ((img * H + y) * static_cast<size_t>(W) + x) * C + c);
        const float* wp = wts + static_cast<size_t>(dy * ks + dx) * C + c;
        const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
          acc[e] = fmaf(value16<KIND>(static_cast<uint16_t>(v[e >> 1] >> (16 * (e & 1)))), wv[e], acc[e]);
      }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a0 = acc[2 * e] / (1.0f + expf(-acc[2 * e])), a1 = acc[2 * e + 1] / (1.0f + expf(-acc[2 * e + 1]));
      o[e] = static_cast<uint32_t>(round16<KIND>(a0)) | (static_cast<uint32_t>(round16<KIND>(a1)) << 16);
    }
    *reinterpret_cast<u32x4*>(out + i * 8) = o;
  }
}

// squeeze-excitation, step 1 on a 16-bit tensor: f32 mean over the pixels.  grid = (C / 64, images); a work-item reads eight
// channels (16 bytes) of every 32nd pixel
template <int KIND>
__global__ void __launch_bounds__(kThreads)
enet_pool16_kernel(const uint16_t* __restrict__ in, int HW, int C, float* __restrict__ pooled) {
  __shared__ float part[32][65];
  const int tid = static_cast<int>(threadIdx.x), g8 = tid & 7, r = tid >> 3;
  const size_t img = blockIdx.y;
  const uint16_t* base = in + img * static_cast<size_t>(HW) * C + static_cast<size_t>(blockIdx.x) * 64 + g8 * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int p = r; p < HW; p += 32) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(base + static_cast<size_t>(p) * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += value16<KIND>(static_cast<uint16_t>(v[e >> 1] >> (16 * (e & 1))));
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[r][g8 * 8 + e] = s[e];
  __syncthreads();
  if (tid < 64) {
    float t = 0.0f;
    for (int k = 0; k < 32; ++k) t += part[k][tid];
    pooled[img * C + blockIdx.x * 64 + tid] = t / static_cast<float>(HW);
  }
}

// ================================================================ DenseNet building blocks (network.py:176-179)
// 2x2 / stride 2 average pool (a transition's tail), NHWC [..][C] -> NHWC with channel stride ldo
__global__ void __launch_bounds__(kThreads)
dnet_avgpool_kernel(const float* __restrict__ in, int H, int W, int C, float* __restrict__ out, size_t total, int ldo) {
  const int Ho = H / 2, Wo = W / 2;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % C);
    size_t p = i / C;
    const int ox = static_cast<int>(p % Wo); p /= Wo;
    const int oy = static_cast<int>(p % Ho);
    const size_t img = p / Ho;
    const float* b = in + ((img * H + 2 * oy) * static_cast<size_t>(W) + 2 * ox) * C + c;
    out[(i / C) * ldo + c] = (b[0] + b[C] + b[static_cast<size_t>(W) * C] + b[static_cast<size_t>(W) * C + C]) * 0.25f;
  }
}

// NHWC (channel stride ld, C channels) -> NCHW with an optional per-channel x * s + t (the closing BatchNorm) and ReLU
__global__ void __launch_bounds__(kThreads)
dnet_out_kernel(const float* __restrict__ in, int HW, int C, int ld, const float* __restrict__ sc, const float* __restrict__ sh,
                int relu, float* __restrict__ out, size_t total) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * kThreads) {
    const int p = static_cast<int>(i % HW);
    const int c = static_cast<int>((i / HW) % C);
    const size_t img = i / (static_cast<size_t>(HW) * C);
    float v = in[(img * HW + p) * ld + c];
    if (sc) v = fmaf(v, sc[c], sh[c]);
    if (relu) v = fmaxf(v, 0.0f);
    out[i] = v;
  }
}

}  // namespace

// First convolution (3x3 / stride 1 / pad 1, 3 -> 64) of a plain VGG in a 16-bit plan, pre-processing fused: see stem16_kernel
int pack_first16(int kind, const float* w, const float* b, float* packed, size_t w_off, size_t b_off, hipStream_t s) {
  if (kind == SPR_F16)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rstem16_pack_kernel<SPR_F16, 3>), dim3(8), dim3(kThreads), 0, s, w, b, packed, w_off, b_off);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rstem16_pack_kernel<SPR_BF16, 3>), dim3(8), dim3(kThreads), 0, s, w, b, packed, w_off, b_off);
  return check_launch("rstem16_pack_kernel");
}
int launch_first16(int kind, const uint8_t* images, int64_t n, int h, int w, int in_channels, const float* mean3,
                   const float* inv_std3, const uint16_t* w16, const float* bias, int relu, uint16_t* out, hipStream_t s) {
  const dim3 grid(static_cast<unsigned>(ceil_div(h, kSTH) * ceil_div(w, kSTW)), static_cast<unsigned>(n));
  if (kind == SPR_F16)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(stem16_kernel<SPR_F16, 3, 1>), grid, dim3(kThreads), 0, s, images, h, w, in_channels,
                       mean3[0], mean3[1], mean3[2], inv_std3[0], inv_std3[1], inv_std3[2], w16, bias, out, relu);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(stem16_kernel<SPR_BF16, 3, 1>), grid, dim3(kThreads), 0, s, images, h, w, in_channels,
                       mean3[0], mean3[1], mean3[2], inv_std3[0], inv_std3[1], inv_std3[2], w16, bias, out, relu);
  return check_launch("stem16_kernel");
}
}  // namespace spr

struct spr_resnet_plan {
  int block;                       // top-level children kept: 5 = layer1, 6 = layer2, 7 = layer3
  int compute;                     // SPR_F32 (exact) | SPR_F16 | SPR_BF16 (16-bit operands, f32 accumulation)
  std::vector<spr::RConv> convs;   // in torchvision's module order (conv1; per bottleneck conv1, conv2, conv3, [downsample])
  size_t packed_floats;
};

using namespace spr;

extern "C" int spr_resnet_plan_create(int32_t block, spr_resnet_plan** plan_out) {
  return spr_resnet_plan_create_ex(block, SPR_F32, plan_out);
}

extern "C" int spr_resnet_plan_compute(const spr_resnet_plan* plan) { return plan ? plan->compute : SPR_ERR_ARG; }

extern "C" int spr_resnet_plan_create_ex(int32_t block, int32_t compute, spr_resnet_plan** plan_out) {
  if (!plan_out) { set_error("spr_resnet_plan_create: null pointer"); return SPR_ERR_ARG; }
  *plan_out = nullptr;
  if (compute != SPR_F32 && compute != SPR_F16 && compute != SPR_BF16) {
    set_error("spr_resnet_plan_create_ex: compute type %d (SPR_F32 | SPR_F16 | SPR_BF16)", compute);
    return SPR_ERR_ARG;
  }
  if (block < 5 || block > 7) {
    set_error("spr_resnet_plan_create: block %d: the truncation must end after layer1 (5), layer2 (6) or layer3 (7)", block);
    return SPR_ERR_ARG;
  }
  spr_resnet_plan* plan = new (std::nothrow) spr_resnet_plan();
  if (!plan) { set_error("out of host memory"); return SPR_ERR_ARG; }
  plan->block = block;
  plan->compute = compute;
  size_t off = 0;
  auto add = [&](int cin, int cout, int ks, int stride, int relu, int res, int role) {
    RConv c{};
    c.cin = cin; c.cout = cout; c.ks = ks; c.stride = stride; c.relu = relu; c.res = res; c.role = role;
    // (16-bit plans: two weights per float slot; the stem's 160 x 64 padded 16-bit matrix fits its f32 allocation)
    c.w_off = off; off += static_cast<size_t>(cout) * cin * ks * ks / ((compute != SPR_F32 && role != 0) ? 2 : 1);
    c.b_off = off; off += static_cast<size_t>(cout);
    off = (off + 3) / 4 * 4;
    plan->convs.push_back(c);
  };
  add(3, 64, 7, 2, 1, 0, 0);
  static const int blocks_per_layer[3] = {3, 4, 6};
  int cin = 64;
  for (int layer = 0; layer < block - 4; ++layer) {
    const int mid = 64 << layer;
    for (int b = 0; b < blocks_per_layer[layer]; ++b) {
      const int stride = (b == 0 && layer > 0) ? 2 : 1;
      add(cin, mid, 1, 1, 1, 0, 1);
      add(mid, mid, 3, stride, 1, 0, 2);
      add(mid, 4 * mid, 1, 1, 1, b == 0 ? 2 : 1, 3);
      if (b == 0) add(cin, 4 * mid, 1, stride, 0, 0, 4);
      cin = 4 * mid;
    }
  }
  plan->packed_floats = off;
  *plan_out = plan;
  return SPR_OK;
}

extern "C" void spr_resnet_plan_destroy(spr_resnet_plan* plan) { delete plan; }

extern "C" int spr_resnet_num_convs(const spr_resnet_plan* plan) {
  return plan ? static_cast<int>(plan->convs.size()) : SPR_ERR_ARG;
}

extern "C" int spr_resnet_conv_shape(const spr_resnet_plan* plan, int32_t i, int32_t* cin, int32_t* cout, int32_t* ksize,
                                     int32_t* stride, int32_t* role) {
  if (!plan || !cin || !cout || !ksize || !stride || !role || i < 0 || i >= static_cast<int>(plan->convs.size())) {
    set_error("spr_resnet_conv_shape: bad argument");
    return SPR_ERR_ARG;
  }
  const RConv& c = plan->convs[i];
  *cin = c.cin; *cout = c.cout; *ksize = c.ks; *stride = c.stride; *role = c.role;
  return SPR_OK;
}

static void resnet_dims(const spr_resnet_plan* plan, int in_h, int in_w, int* c, int* h, int* w) {
  int hh = (in_h + 1) / 2, ww = (in_w + 1) / 2;  // conv1: 7x7 s2 p3
  hh = (hh + 1) / 2; ww = (ww + 1) / 2;          // maxpool 3x3 s2 p1
  for (int layer = 1; layer < plan->block - 4; ++layer) { hh = (hh + 1) / 2; ww = (ww + 1) / 2; }  // 3x3 s2 p1
  *c = 256 << (plan->block - 5); *h = hh; *w = ww;
}

extern "C" int spr_resnet_output_shape(const spr_resnet_plan* plan, int32_t in_h, int32_t in_w, int32_t* channels,
                                       int32_t* out_h, int32_t* out_w) {
  if (!plan || !channels || !out_h || !out_w || in_h < 1 || in_w < 1) {
    set_error("spr_resnet_output_shape: bad argument");
    return SPR_ERR_ARG;
  }
  int c, h, w;
  resnet_dims(plan, in_h, in_w, &c, &h, &w);
  *channels = c; *out_h = h; *out_w = w;
  return SPR_OK;
}

extern "C" size_t spr_resnet_packed_bytes(const spr_resnet_plan* plan) {
  return plan ? plan->packed_floats * sizeof(float) : 0;
}

extern "C" int spr_resnet_pack_weights(spr_resnet_plan* plan, const float* const* weights, const float* const* biases,
                                       void* packed, spr_stream_t stream) {
  if (!plan || !weights || !biases || !packed) { set_error("spr_resnet_pack_weights: null pointer"); return SPR_ERR_ARG; }
  for (size_t i = 0; i < plan->convs.size(); ++i) {
    const RConv& c = plan->convs[i];
    if (!weights[i] || !biases[i]) { set_error("spr_resnet_pack_weights: null parameter %zu", i); return SPR_ERR_ARG; }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (i == 0 && plan->compute == SPR_F16)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rstem16_pack_kernel<SPR_F16, 7>), dim3(40), dim3(kThreads), 0, hs, weights[i], biases[i],
                         static_cast<float*>(packed), c.w_off, c.b_off);
    else if (i == 0 && plan->compute == SPR_BF16)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rstem16_pack_kernel<SPR_BF16, 7>), dim3(40), dim3(kThreads), 0, hs, weights[i], biases[i],
                         static_cast<float*>(packed), c.w_off, c.b_off);
    else if (plan->compute != SPR_F32 && c.ks == 3 && c.stride == 1) {
      // the 3x3 / stride 1 layers of a 16-bit plan run on vgg_conv.hip's patch kernel: its weight layout
      const int rc3 = pack_conv16_3x3(plan->compute, weights[i], biases[i], static_cast<float*>(packed), c.w_off, c.b_off, c.cin,
                                      c.cout, hs);
      if (rc3 != SPR_OK) return rc3;
    } else if (i > 0 && plan->compute == SPR_F16)
      hipLaunchKernelGGL(rpack16_kernel<SPR_F16>, dim3(256), dim3(kThreads), 0, hs, weights[i], biases[i],
                         static_cast<float*>(packed), c.w_off, c.b_off, c.cin, c.cout, c.ks);
    else if (i > 0 && plan->compute == SPR_BF16)
      hipLaunchKernelGGL(rpack16_kernel<SPR_BF16>, dim3(256), dim3(kThreads), 0, hs, weights[i], biases[i],
                         static_cast<float*>(packed), c.w_off, c.b_off, c.cin, c.cout, c.ks);
    else
      hipLaunchKernelGGL(rpack_kernel, dim3(256), dim3(kThreads), 0, hs, weights[i], biases[i],
                         static_cast<float*>(packed), c.w_off, c.b_off, c.cin, c.cout, c.ks, i == 0 ? 1 : 0);
    const int rc = check_launch("rpack_kernel");
    if (rc != SPR_OK) return rc;
  }
  return SPR_OK;
}

// four activation buffers (block input, two bottleneck intermediates / the downsample branch, block output), each as
// large as the largest tensor between layers: the stem's output
static size_t resnet_big16_bytes(int64_t n, int in_h, int in_w) {  // layer1's output, the largest 16-bit tensor
  const int hp = ((in_h + 1) / 2 + 1) / 2, wp = ((in_w + 1) / 2 + 1) / 2;
  return align_up(static_cast<size_t>(n) * hp * wp * 256 * sizeof(uint16_t), 256);
}
extern "C" size_t spr_resnet_workspace_bytes(const spr_resnet_plan* plan, int64_t n, int32_t in_h, int32_t in_w) {
  if (!plan || n < 0) return 0;
  const size_t stem = static_cast<size_t>(n) * ((in_h + 1) / 2) * ((in_w + 1) / 2) * 64;
  // 16-bit plans: the stem's f32 tensor, then four 16-bit activation buffers
  if (plan->compute != SPR_F32) return align_up(stem * sizeof(float), 256) + 4 * resnet_big16_bytes(n, in_h, in_w);
  return 4 * align_up(stem * sizeof(float), 256);
}

// (cin, cout: the channel counts of the tensors = the padded widths of an EfficientNet layer)
template <int KS, int STRIDE>
static int launch_gemm16_raw(int kind, int cin, int cout, int act, const uint16_t* w16, const float* bias, const uint16_t* in,
                             int64_t n, int h, int w, const uint16_t* res, uint16_t* out, float* out32, const float* in_scale,
                             int cout_real, hipStream_t s) {
  const int pad = KS / 2;
  const int ho = (h + 2 * pad - KS) / STRIDE + 1, wo = (w + 2 * pad - KS) / STRIDE + 1;
  const long long m = static_cast<long long>(n) * ho * wo;
  const dim3 grid(static_cast<unsigned>((m + kHM - 1) / kHM), static_cast<unsigned>(cout / 64));
  if (kind == SPR_F16)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_gemm16_kernel<KS, STRIDE, SPR_F16, 64>), grid, dim3(kThreads), 0, s, in,
                       static_cast<int>(n), h, w, cin, cout, w16, bias, res, act, out, out32, in_scale, cout_real);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_gemm16_kernel<KS, STRIDE, SPR_BF16, 64>), grid, dim3(kThreads), 0, s, in,
                       static_cast<int>(n), h, w, cin, cout, w16, bias, res, act, out, out32, in_scale, cout_real);
  return check_launch("conv_gemm16_kernel");
}

template <int KS, int STRIDE>
static int launch_gemm16(int kind, const RConv& c, const uint16_t* in, int64_t n, int h, int w, const float* pk,
                         const uint16_t* res, uint16_t* out, float* out32, hipStream_t s) {
  const int pad = KS / 2;
  const int ho = (h + 2 * pad - KS) / STRIDE + 1, wo = (w + 2 * pad - KS) / STRIDE + 1;
  const long long m = static_cast<long long>(n) * ho * wo;
  const unsigned mt = static_cast<unsigned>((m + kHM - 1) / kHM);
  // 128-channel tiles only on request (SPR_GEMM16_BN=128; tests and A/B runs)
  static const int forced = [] { const char* v = std::getenv("SPR_GEMM16_BN"); return v && *v ? std::atoi(v) : 0; }();
  // (measured on ResNet50 through layer3, batch 32: 17.9 k images/s with 64-channel tiles throughout, 16.9 k with 128)
  const bool wide = c.cout % 128 == 0 && forced == 128;
  const uint16_t* w16 = reinterpret_cast<const uint16_t*>(pk + c.w_off);
#define SPR_LAUNCH16(KIND_, BN_)                                                                                              \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_gemm16_kernel<KS, STRIDE, KIND_, BN_>), dim3(mt, static_cast<unsigned>(c.cout / BN_)), \
                     dim3(kThreads), 0, s, in, static_cast<int>(n), h, w, c.cin, c.cout, w16, pk + c.b_off, res, c.relu, out, out32, \
                     static_cast<const float*>(nullptr), 0)
  if (kind == SPR_F16) { if (wide) SPR_LAUNCH16(SPR_F16, 128); else SPR_LAUNCH16(SPR_F16, 64); }
  else { if (wide) SPR_LAUNCH16(SPR_BF16, 128); else SPR_LAUNCH16(SPR_BF16, 64); }
#undef SPR_LAUNCH16
  return check_launch("conv_gemm16_kernel");
}

// the bottlenecks of a 16-bit plan: x (the pooled stem output, 16-bit NHWC) lives in buf[0]; t1 / t2 / y in buf[1..3]
static int resnet_blocks16(const spr_resnet_plan* plan, int64_t n, int h, int w, const float* pk, uint16_t* const buf[4],
                           float* out, hipStream_t s) {
  uint16_t* x = buf[0];
  uint16_t* t1 = buf[1];
  uint16_t* t2 = buf[2];
  uint16_t* y = buf[3];
  const int kind = plan->compute;
  size_t i = 1;
  while (i < plan->convs.size()) {
    const RConv& c1 = plan->convs[i];
    const RConv& c2 = plan->convs[i + 1];
    const RConv& c3 = plan->convs[i + 2];
    const bool down = c3.res == 2;
    const bool last = i + (down ? 4 : 3) == plan->convs.size();
    const int ho = c2.stride == 2 ? (h + 1) / 2 : h, wo = c2.stride == 2 ? (w + 1) / 2 : w;
    int rc = launch_gemm16<1, 1>(kind, c1, x, n, h, w, pk, nullptr, t1, nullptr, s);
    if (rc != SPR_OK) return rc;
    rc = c2.stride == 2 ? launch_gemm16<3, 2>(kind, c2, t1, n, h, w, pk, nullptr, t2, nullptr, s)
                        : launch_conv16_3x3(kind, t1, n, h, w, c2.cin, c2.cout, reinterpret_cast<const uint16_t*>(pk + c2.w_off),
                                            pk + c2.b_off, c2.relu, t2, s);
    if (rc != SPR_OK) return rc;
    const uint16_t* resid = x;
    if (down) {
      const RConv& cd = plan->convs[i + 3];
      rc = cd.stride == 2 ? launch_gemm16<1, 2>(kind, cd, x, n, h, w, pk, nullptr, t1, nullptr, s)
                          : launch_gemm16<1, 1>(kind, cd, x, n, h, w, pk, nullptr, t1, nullptr, s);
      if (rc != SPR_OK) return rc;
      resid = t1;
    }
    rc = launch_gemm16<1, 1>(kind, c3, t2, n, ho, wo, pk, resid, y, last ? out : nullptr, s);
    if (rc != SPR_OK) return rc;
    h = ho; w = wo;
    uint16_t* old = x;
    x = y;
    y = old;
    i += down ? 4 : 3;
  }
  return SPR_OK;
}

template <int KS, int STRIDE>
static int launch_gemm(const RConv& c, const float* in, int64_t n, int h, int w, const float* pk, const float* res, int nchw,
                       float* out, hipStream_t s) {
  const int pad = KS / 2;
  const int ho = (h + 2 * pad - KS) / STRIDE + 1, wo = (w + 2 * pad - KS) / STRIDE + 1;
  const long long m = static_cast<long long>(n) * ho * wo;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_gemm_kernel<KS, STRIDE>), dim3(static_cast<unsigned>((m + kGM - 1) / kGM),
                     static_cast<unsigned>(c.cout / kGN)), dim3(kThreads), 0, s, in, static_cast<int>(n), h, w, c.cin, c.cout,
                     pk + c.w_off, pk + c.b_off, res, c.relu, nchw, out, static_cast<const float*>(nullptr), 0, c.cin, c.cout, 0,
                     static_cast<const float*>(nullptr), static_cast<const float*>(nullptr));
  return check_launch("conv_gemm_kernel");
}

extern "C" int spr_resnet_forward(spr_resnet_plan* plan, const uint8_t* images, int64_t n, int32_t in_h, int32_t in_w,
                                  int32_t in_channels, const float* mean3, const float* inv_std3, const void* packed,
                                  void* workspace, float* out, spr_stream_t stream) {
  if (!plan) { set_error("spr_resnet_forward: null plan"); return SPR_ERR_ARG; }
  if (n < 0 || n > 65535 || in_h < 32 || in_w < 32 || (in_channels != 1 && in_channels != 3)) {
    set_error("spr_resnet_forward: bad sizes (n in [0, 65535], images at least 32 x 32, in_channels 1 or 3)");
    return SPR_ERR_ARG;
  }
  if (n == 0) return SPR_OK;
  if (!images || !mean3 || !inv_std3 || !packed || !out || !workspace) { set_error("spr_resnet_forward: null pointer"); return SPR_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* pk = static_cast<const float*>(packed);
  const bool f32 = plan->compute == SPR_F32;
  const size_t quarter = f32 ? spr_resnet_workspace_bytes(plan, n, in_h, in_w) / 4 : 0;
  float* buf[4];
  for (int i = 0; i < 4; ++i) buf[i] = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + i * quarter);
  uint16_t* b16[4] = {nullptr, nullptr, nullptr, nullptr};
  if (!f32) {  // [stem f32 tensor][x][t1][t2][y]
    const size_t stem_bytes = align_up(static_cast<size_t>(n) * ((in_h + 1) / 2) * ((in_w + 1) / 2) * 64 * sizeof(float), 256);
    buf[1] = static_cast<float*>(workspace);
    for (int i = 0; i < 4; ++i)
      b16[i] = reinterpret_cast<uint16_t*>(static_cast<unsigned char*>(workspace) + stem_bytes + i * resnet_big16_bytes(n, in_h, in_w));
  }
  // stem + max pool
  int h = (in_h + 1) / 2, w = (in_w + 1) / 2;
  {
    const RConv& c = plan->convs[0];
    const unsigned tiles = static_cast<unsigned>(ceil_div(h, 8) * ceil_div(w, 8));
    if (f32) {
      hipLaunchKernelGGL(stem_kernel, dim3(tiles, static_cast<unsigned>(n)), dim3(kThreads), 0, s, images, in_h, in_w,
                         in_channels, mean3[0], mean3[1], mean3[2], inv_std3[0], inv_std3[1], inv_std3[2], pk + c.w_off,
                         pk + c.b_off, buf[1], 1, 0);
    } else {
      const dim3 sgrid(static_cast<unsigned>(ceil_div(h, kSTH) * ceil_div(w, kSTW)), static_cast<unsigned>(n));
      const uint16_t* w16 = reinterpret_cast<const uint16_t*>(pk + c.w_off);
      uint16_t* o16 = reinterpret_cast<uint16_t*>(buf[1]);
      if (plan->compute == SPR_F16)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem16_kernel<SPR_F16, 7, 2>), sgrid, dim3(kThreads), 0, s, images, in_h, in_w, in_channels, mean3[0],
                           mean3[1], mean3[2], inv_std3[0], inv_std3[1], inv_std3[2], w16, pk + c.b_off, o16, 1);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem16_kernel<SPR_BF16, 7, 2>), sgrid, dim3(kThreads), 0, s, images, in_h, in_w, in_channels, mean3[0],
                           mean3[1], mean3[2], inv_std3[0], inv_std3[1], inv_std3[2], w16, pk + c.b_off, o16, 1);
    }
    int rc = check_launch("stem_kernel");
    if (rc != SPR_OK) return rc;
    const int hp = (h + 1) / 2, wp = (w + 1) / 2;
    const size_t total = static_cast<size_t>(n) * hp * wp * 64;
    const dim3 pgrid(static_cast<unsigned>(std::min<size_t>((total + kThreads - 1) / kThreads, 65535 * 16)));
    if (plan->compute == SPR_F32) {
      hipLaunchKernelGGL(maxpool3_kernel, pgrid, dim3(kThreads), 0, s, buf[1], h, w, 64, buf[0], total, 64);
    } else {
      // (the stem stored its activation rounded to the 16-bit type; the pooled tensor is layer1's operand)
      const dim3 pgrid8(static_cast<unsigned>(std::min<size_t>((total / 8 + kThreads - 1) / kThreads, 65535 * 16)));
      hipLaunchKernelGGL(maxpool3_16_kernel, pgrid8, dim3(kThreads), 0, s, reinterpret_cast<const uint16_t*>(buf[1]), h, w, 64,
                         b16[0], total / 8);
    }
    rc = check_launch("maxpool3_kernel");
    if (rc != SPR_OK) return rc;
    h = hp; w = wp;
  }
  if (!f32) return resnet_blocks16(plan, n, h, w, pk, b16, out, s);
  // bottlenecks: x = buf[0]
  float* x = buf[0];
  float* t1 = buf[1];
  float* t2 = buf[2];
  float* y = buf[3];
  size_t i = 1;
  while (i < plan->convs.size()) {
    const RConv& c1 = plan->convs[i];
    const RConv& c2 = plan->convs[i + 1];
    const RConv& c3 = plan->convs[i + 2];
    const bool down = c3.res == 2;
    const bool last = i + (down ? 4 : 3) == plan->convs.size();
    const int ho = c2.stride == 2 ? (h + 1) / 2 : h, wo = c2.stride == 2 ? (w + 1) / 2 : w;
    int rc = launch_gemm<1, 1>(c1, x, n, h, w, pk, nullptr, 0, t1, s);
    if (rc != SPR_OK) return rc;
    rc = c2.stride == 2 ? launch_gemm<3, 2>(c2, t1, n, h, w, pk, nullptr, 0, t2, s)
                        : launch_gemm<3, 1>(c2, t1, n, h, w, pk, nullptr, 0, t2, s);
    if (rc != SPR_OK) return rc;
    const float* resid = x;
    if (down) {
      const RConv& cd = plan->convs[i + 3];
      rc = cd.stride == 2 ? launch_gemm<1, 2>(cd, x, n, h, w, pk, nullptr, 0, t1, s)
                          : launch_gemm<1, 1>(cd, x, n, h, w, pk, nullptr, 0, t1, s);
      if (rc != SPR_OK) return rc;
      resid = t1;
    }
    float* dst = last ? out : y;
    rc = launch_gemm<1, 1>(c3, t2, n, ho, wo, pk, resid, last ? 1 : 0, dst, s);
    if (rc != SPR_OK) return rc;
    h = ho; w = wo;
    float* old = x;
    x = y;
    y = old;
    i += down ? 4 : 3;
  }
  return SPR_OK;
}

// ================================================================ EfficientNetV2 truncations (network.py:163-175, :185-186)
// torchvision's efficientnet_v2_{s,m,l}: features = [stem, stage 1 .. stage N, last conv]; the reference keeps
// features[:block].  Stages of FusedMBConv (3x3 expansion convolution, 1x1 projection) and MBConv (1x1 expansion, depthwise
// 3x3, squeeze-excitation, 1x1 projection), SiLU, residual where stride 1 and equal widths; stochastic depth is the identity
// in eval mode.  Flattened here into a list of layers; the host folds BatchNorm (eps 1e-3) and packs the parameters.
namespace {

struct EOp {
  int kind;         // 0 convolution (implicit GEMM), 1 depthwise 3x3, 2 squeeze-excitation
  int cin, cout;    // real channels (squeeze-excitation: cin = cout = expanded width)
  int cin_p, cout_p;
  int ks, stride, act;
  int res;          // convolution: add the block input behind it
  int scaled;       // convolution: its input is multiplied by the squeeze-excitation factors
  int sq;           // squeeze-excitation: hidden width
  int block_end;    // last layer of a residual block (or of the stem)
  int feature;      // index of the top-level child of `features` this layer belongs to
  size_t w_off, b_off, w2_off, b2_off;  // floats into the packed buffer (multiples of 4)
};

struct EStage { int fused, expand, stride, cin, cout, layers, ks; };

const EStage kV2S[] = {{1, 1, 1, 24, 24, 2, 3}, {1, 4, 2, 24, 48, 4, 3}, {1, 4, 2, 48, 64, 4, 3}, {0, 4, 2, 64, 128, 6, 3},
                       {0, 6, 1, 128, 160, 9, 3}, {0, 6, 2, 160, 256, 15, 3}};
const EStage kV2M[] = {{1, 1, 1, 24, 24, 3, 3}, {1, 4, 2, 24, 48, 5, 3}, {1, 4, 2, 48, 80, 5, 3}, {0, 4, 2, 80, 160, 7, 3},
                       {0, 6, 1, 160, 176, 14, 3}, {0, 6, 2, 176, 304, 18, 3}, {0, 6, 1, 304, 512, 5, 3}};
const EStage kV2L[] = {{1, 1, 1, 32, 32, 4, 3}, {1, 4, 2, 32, 64, 7, 3}, {1, 4, 2, 64, 96, 7, 3}, {0, 4, 2, 96, 192, 10, 3},
                       {0, 6, 1, 192, 224, 19, 3}, {0, 6, 2, 224, 384, 25, 3}, {0, 6, 1, 384, 640, 7, 3}};
// EfficientNet_B0's stages (all MBConv); B1 .. B7 scale the widths and depths (network.py:139-162)
const EStage kB0[] = {{0, 1, 1, 32, 16, 1, 3}, {0, 6, 2, 16, 24, 2, 3}, {0, 6, 2, 24, 40, 2, 5}, {0, 6, 2, 40, 80, 3, 3},
                      {0, 6, 1, 80, 112, 3, 5}, {0, 6, 2, 112, 192, 4, 5}, {0, 6, 1, 192, 320, 1, 3}};
// arch 3 .. 8 = EfficientNet_B1, B2, B3, B4, B5, B7: width and depth multipliers in tenths
const int kBWidth[6] = {10, 11, 12, 14, 16, 20}, kBDepth[6] = {11, 12, 14, 18, 22, 31};

inline int make_divisible8(double v) {  // torchvision's _make_divisible(v, 8)
  int n = static_cast<int>(v + 4.0) / 8 * 8;
  if (n < 8) n = 8;
  if (n < 0.9 * v) n += 8;
  return n;
}

inline int pad64(int c) { return (c + 63) / 64 * 64; }

}  // namespace

struct spr_effnet_plan {
  int arch, block;
  int compute;  // SPR_F32 | SPR_F16 | SPR_BF16
  std::vector<EOp> ops;
  size_t packed_floats;
  int max_expand_p;  // widest expanded tensor (squeeze-excitation scratch)
};

extern "C" int spr_effnet_plan_create(int32_t arch, int32_t block, spr_effnet_plan** plan_out) {
  return spr_effnet_plan_create_ex(arch, block, SPR_F32, plan_out);
}

extern "C" int spr_effnet_plan_compute(const spr_effnet_plan* plan) { return plan ? plan->compute : SPR_ERR_ARG; }

extern "C" int spr_effnet_plan_create_ex(int32_t arch, int32_t block, int32_t compute, spr_effnet_plan** plan_out) {
  if (!plan_out) { set_error("spr_effnet_plan_create: null pointer"); return SPR_ERR_ARG; }
  *plan_out = nullptr;
  if (compute != SPR_F32 && compute != SPR_F16 && compute != SPR_BF16) {
    set_error("spr_effnet_plan_create_ex: compute type %d (SPR_F32 | SPR_F16 | SPR_BF16)", compute);
    return SPR_ERR_ARG;
  }
  EStage scaled[7];
  const EStage* stages = arch == 0 ? kV2S : arch == 1 ? kV2M : arch == 2 ? kV2L : nullptr;
  const int n_stages = arch == 0 ? 6 : 7;
  if (arch >= 3 && arch <= 8) {
    const double wm = kBWidth[arch - 3] / 10.0, dm = kBDepth[arch - 3] / 10.0;
    for (int i = 0; i < 7; ++i) {
      scaled[i] = kB0[i];
      scaled[i].cin = make_divisible8(kB0[i].cin * wm);
      scaled[i].cout = make_divisible8(kB0[i].cout * wm);
      scaled[i].layers = static_cast<int>(std::ceil(kB0[i].layers * dm - 1e-9));
    }
    stages = scaled;
  }
  if (!stages) {
    set_error("spr_effnet_plan_create: arch %d (0 .. 2 = EfficientNetV2_S / _M / _L, 3 .. 8 = EfficientNet_B1 / B2 / B3 / B4 / B5 / B7)", arch);
    return SPR_ERR_ARG;
  }
  if (block < 1 || block > n_stages + 2) {
    set_error("spr_effnet_plan_create: block %d: features[:block] with block in [1, %d] (= len(model.features))", block,
              n_stages + 2);
    return SPR_ERR_ARG;
  }
  spr_effnet_plan* plan = new (std::nothrow) spr_effnet_plan();
  if (!plan) { set_error("out of host memory"); return SPR_ERR_ARG; }
  plan->arch = arch; plan->block = block; plan->max_expand_p = 64; plan->compute = compute;
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off += (n + 3) / 4 * 4; return o; };
  auto conv = [&](int cin, int cout, int ks, int stride, int act, int res, int scaled, int end, int feature, int cin_p) {
    EOp o{};
    o.kind = 0; o.cin = cin; o.cout = cout; o.cin_p = cin_p; o.cout_p = pad64(cout); o.ks = ks; o.stride = stride; o.act = act;
    o.res = res; o.scaled = scaled; o.block_end = end; o.feature = feature;
    o.w_off = take(static_cast<size_t>(o.cout_p) * o.cin_p * ks * ks);
    o.b_off = take(o.cout_p);
    plan->ops.push_back(o);
  };
  const int stem_out = stages[0].cin;
  conv(3, stem_out, 3, 2, 2, 0, 0, 1, 0, 16);
  for (int st = 0; st < block - 1 && st < n_stages; ++st) {
    const EStage& g = stages[st];
    for (int l = 0; l < g.layers; ++l) {
      const int cin = l == 0 ? g.cin : g.cout, stride = l == 0 ? g.stride : 1;
      const int exp = arch >= 3 ? make_divisible8(static_cast<double>(cin) * g.expand) : cin * g.expand;
      const int res = stride == 1 && cin == g.cout;
      if (g.fused) {
        if (g.expand == 1) {
          conv(cin, g.cout, 3, stride, 2, res, 0, 1, st + 1, pad64(cin));
        } else {
          conv(cin, exp, 3, stride, 2, 0, 0, 0, st + 1, pad64(cin));
          conv(exp, g.cout, 1, 1, 0, res, 0, 1, st + 1, pad64(exp));
        }
      } else {
        if (exp != cin) conv(cin, exp, 1, 1, 2, 0, 0, 0, st + 1, pad64(cin));  // (no expansion convolution at ratio 1)
        EOp d{};
        d.kind = 1; d.cin = d.cout = exp; d.cin_p = d.cout_p = pad64(exp); d.ks = g.ks; d.stride = stride; d.act = 2; d.feature = st + 1;
        d.w_off = take(static_cast<size_t>(g.ks) * g.ks * d.cin_p);
        d.b_off = take(d.cin_p);
        plan->ops.push_back(d);
        EOp e{};
        e.kind = 2; e.cin = e.cout = exp; e.cin_p = e.cout_p = pad64(exp); e.sq = cin / 4 > 1 ? cin / 4 : 1; e.feature = st + 1;
        e.w_off = take(static_cast<size_t>(e.sq) * e.cin_p);
        e.b_off = take(e.sq);
        e.w2_off = take(static_cast<size_t>(e.cin_p) * e.sq);
        e.b2_off = take(e.cin_p);
        plan->ops.push_back(e);
        if (e.cin_p > plan->max_expand_p) plan->max_expand_p = e.cin_p;
        conv(exp, g.cout, 1, 1, 0, res, 1, 1, st + 1, pad64(exp));
      }
    }
  }
  if (block == n_stages + 2) {
    // the closing 1x1 convolution + BatchNorm + SiLU of `features`: 1280 channels in the V2 models, four times the last stage's
    // width in the B-series (torchvision: last_channel or 4 * lastconv_input_channels)
    const int cin = stages[n_stages - 1].cout;
    conv(cin, arch <= 2 ? 1280 : 4 * cin, 1, 1, 2, 0, 0, 1, n_stages + 1, pad64(cin));
  }
  plan->packed_floats = off;
  *plan_out = plan;
  return SPR_OK;
}

extern "C" void spr_effnet_plan_destroy(spr_effnet_plan* plan) { delete plan; }
extern "C" int spr_effnet_num_ops(const spr_effnet_plan* plan) { return plan ? static_cast<int>(plan->ops.size()) : SPR_ERR_ARG; }
extern "C" size_t spr_effnet_packed_bytes(const spr_effnet_plan* plan) { return plan ? plan->packed_floats * sizeof(float) : 0; }

// info[16] = kind, cin, cout, cin_p, cout_p, ks, stride, act, res, sq, feature, then the four packed offsets (in floats,
// each < 2^31) w, b, w2, b2, then block_end (1: last layer of a residual block or of the stem)
extern "C" int spr_effnet_op_info(const spr_effnet_plan* plan, int32_t i, int32_t* info) {
  if (!plan || !info || i < 0 || i >= static_cast<int>(plan->ops.size())) { set_error("spr_effnet_op_info: bad argument"); return SPR_ERR_ARG; }
  const EOp& o = plan->ops[i];
  const int32_t v[16] = {o.kind, o.cin, o.cout, o.cin_p, o.cout_p, o.ks, o.stride, o.act, o.res, o.sq, o.feature,
                         static_cast<int32_t>(o.w_off), static_cast<int32_t>(o.b_off), static_cast<int32_t>(o.w2_off),
                         static_cast<int32_t>(o.b2_off), o.block_end};
  for (int k = 0; k < 16; ++k) info[k] = v[k];
  return SPR_OK;
}

static void effnet_dims(const spr_effnet_plan* plan, int in_h, int in_w, int* c, int* h, int* w) {
  int hh = in_h, ww = in_w, cc = 3;
  for (const EOp& o : plan->ops) {
    if (o.kind == 2) continue;
    if (o.stride == 2) { hh = (hh - 1) / 2 + 1; ww = (ww - 1) / 2 + 1; }
    cc = o.cout;
  }
  *c = cc; *h = hh; *w = ww;
}

extern "C" int spr_effnet_output_shape(const spr_effnet_plan* plan, int32_t in_h, int32_t in_w, int32_t* channels,
                                       int32_t* out_h, int32_t* out_w) {
  if (!plan || !channels || !out_h || !out_w || in_h < 1 || in_w < 1) { set_error("spr_effnet_output_shape: bad argument"); return SPR_ERR_ARG; }
  int c, h, w;
  effnet_dims(plan, in_h, in_w, &c, &h, &w);
  *channels = c; *out_h = h; *out_w = w;
  return SPR_OK;
}

// four activation buffers as large as the largest tensor between layers + the normalised input + the squeeze-excitation
// vectors (mean and factors)
static size_t effnet_buf_floats(const spr_effnet_plan* plan, int64_t n, int in_h, int in_w) {
  size_t best = static_cast<size_t>(n) * in_h * in_w * 16;
  int hh = in_h, ww = in_w;
  for (const EOp& o : plan->ops) {
    if (o.kind == 2) continue;
    if (o.stride == 2) { hh = (hh - 1) / 2 + 1; ww = (ww - 1) / 2 + 1; }
    const size_t f = static_cast<size_t>(n) * hh * ww * o.cout_p;
    if (f > best) best = f;
  }
  return best;
}
extern "C" size_t spr_effnet_workspace_bytes(const spr_effnet_plan* plan, int64_t n, int32_t in_h, int32_t in_w) {
  if (!plan || n < 0) return 0;
  const size_t buf = align_up(effnet_buf_floats(plan, n, in_h, in_w) * sizeof(float), 256);
  // four activation buffers, the squeeze-excitation means and factors, and the hidden units of its first layer
  return 4 * buf + 2 * align_up(static_cast<size_t>(n) * plan->max_expand_p * sizeof(float), 256) +
         align_up(static_cast<size_t>(n) * 256 * sizeof(float), 256);
}

template <int KS, int STRIDE>
static int launch_egemm(const EOp& o, const float* in, int64_t n, int h, int w, const float* pk, const float* res,
                        const float* scale, int nchw, float* out, hipStream_t s) {
  const int pad = KS / 2;
  const int ho = (h + 2 * pad - KS) / STRIDE + 1, wo = (w + 2 * pad - KS) / STRIDE + 1;
  const long long m = static_cast<long long>(n) * ho * wo;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_gemm_kernel<KS, STRIDE>), dim3(static_cast<unsigned>((m + kGM - 1) / kGM),
                     static_cast<unsigned>(o.cout_p / kGN)), dim3(kThreads), 0, s, in, static_cast<int>(n), h, w, o.cin_p,
                     o.cout_p, pk + o.w_off, pk + o.b_off, res, o.act, nchw, out, scale, nchw ? o.cout : 0, o.cin_p, o.cout_p, 0,
                     static_cast<const float*>(nullptr), static_cast<const float*>(nullptr));
  return check_launch("conv_gemm_kernel");
}

// The 16-bit plans' forward pass: the same walk over the flattened layers with float16 / bfloat16 activations (padded to 64
// channels, the same four buffers: the f32 sizes are kept, half of each is used), the stem on stem16_kernel's 3x3 / stride 2
// instance, every other convolution on conv_gemm16_kernel (SiLU in front of the residual sum, squeeze-excitation factors on
// the operand), depthwise convolutions and the squeeze-excitation mean on their 16-bit kernels; float32 NCHW out.
static int effnet_forward16(const spr_effnet_plan* plan, const uint8_t* images, int64_t n, int in_h, int in_w, int in_channels,
                            const float* mean3, const float* inv_std3, const float* pk, unsigned char* ws, size_t buf_bytes,
                            float* out, hipStream_t s) {
  const int kind = plan->compute;
  uint16_t* x = reinterpret_cast<uint16_t*>(ws);
  uint16_t* t1 = reinterpret_cast<uint16_t*>(ws + buf_bytes);
  uint16_t* t2 = reinterpret_cast<uint16_t*>(ws + 2 * buf_bytes);
  uint16_t* y = reinterpret_cast<uint16_t*>(ws + 3 * buf_bytes);
  float* pooled = reinterpret_cast<float*>(ws + 4 * buf_bytes);
  float* factors = pooled + align_up(static_cast<size_t>(n) * plan->max_expand_p * sizeof(float), 256) / sizeof(float);
  float* hidden = factors + align_up(static_cast<size_t>(n) * plan->max_expand_p * sizeof(float), 256) / sizeof(float);
  int h = in_h, w = in_w;
  int rc;
  {  // stem: 3x3 / stride 2, 3 -> 64 (padded), SiLU, pre-processing fused
    const EOp& o = plan->ops[0];
    if (plan->ops.size() == 1 || o.cout_p != 64) { set_error("spr_effnet_forward: a 16-bit plan needs layers behind a 64-wide stem"); return SPR_ERR_UNSUPPORTED; }
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    const dim3 sgrid(static_cast<unsigned>(ceil_div(ho, kSTH) * ceil_div(wo, kSTW)), static_cast<unsigned>(n));
    const uint16_t* w16 = reinterpret_cast<const uint16_t*>(pk + o.w_off);
    if (kind == SPR_F16)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(stem16_kernel<SPR_F16, 3, 2>), sgrid, dim3(kThreads), 0, s, images, h, w, in_channels,
                         mean3[0], mean3[1], mean3[2], inv_std3[0], inv_std3[1], inv_std3[2], w16, pk + o.b_off, x, 2);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(stem16_kernel<SPR_BF16, 3, 2>), sgrid, dim3(kThreads), 0, s, images, h, w, in_channels,
                         mean3[0], mean3[1], mean3[2], inv_std3[0], inv_std3[1], inv_std3[2], w16, pk + o.b_off, x, 2);
    rc = check_launch("stem16_kernel");
    if (rc != SPR_OK) return rc;
    h = ho; w = wo;
  }
  const uint16_t* cur = x;
  uint16_t* tmp[2] = {t1, t2};
  int ti = 0;
  const float* scale = nullptr;
  for (size_t i = 1; i < plan->ops.size(); ++i) {
    const EOp& o = plan->ops[i];
    const bool last = i + 1 == plan->ops.size();
    if (o.kind == 0) {
      uint16_t* dst = o.block_end ? y : tmp[ti];
      const uint16_t* res = o.res ? x : nullptr;
      const float* sc = o.scaled ? scale : nullptr;
      const uint16_t* w16 = reinterpret_cast<const uint16_t*>(pk + o.w_off);
      float* o32 = last ? out : nullptr;
      if (o.ks == 3 && o.stride == 2)
        rc = launch_gemm16_raw<3, 2>(kind, o.cin_p, o.cout_p, o.act, w16, pk + o.b_off, cur, n, h, w, res, dst, o32, sc, o.cout, s);
      else if (o.ks == 3)
        rc = launch_gemm16_raw<3, 1>(kind, o.cin_p, o.cout_p, o.act, w16, pk + o.b_off, cur, n, h, w, res, dst, o32, sc, o.cout, s);
      else
        rc = launch_gemm16_raw<1, 1>(kind, o.cin_p, o.cout_p, o.act, w16, pk + o.b_off, cur, n, h, w, res, dst, o32, sc, o.cout, s);
      if (rc != SPR_OK) return rc;
      if (o.stride == 2) { h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1; }
      if (o.block_end) {
        uint16_t* old = x; x = y; y = old;
        cur = x;
        ti = 0;
      } else {
        cur = dst;
        ti ^= 1;
      }
    } else if (o.kind == 1) {
      uint16_t* dst = tmp[ti];
      const int ho = (h - 1) / o.stride + 1, wo = (w - 1) / o.stride + 1;
      const size_t total = static_cast<size_t>(n) * ho * wo * (o.cin_p / 8);
      const dim3 grid(static_cast<unsigned>(std::min<size_t>((total + kThreads - 1) / kThreads, 65535 * 16)));
      if (kind == SPR_F16)
        hipLaunchKernelGGL(enet_dw16_kernel<SPR_F16>, grid, dim3(kThreads), 0, s, cur, static_cast<int>(n), h, w, o.cin_p, o.stride,
                           o.ks, pk + o.w_off, pk + o.b_off, dst);
      else
        hipLaunchKernelGGL(enet_dw16_kernel<SPR_BF16>, grid, dim3(kThreads), 0, s, cur, static_cast<int>(n), h, w, o.cin_p, o.stride,
                           o.ks, pk + o.w_off, pk + o.b_off, dst);
      rc = check_launch("enet_dw16_kernel");
      if (rc != SPR_OK) return rc;
      h = ho; w = wo;
      cur = dst;
      ti ^= 1;
    } else {
      const dim3 grid(o.cin_p / 64, static_cast<unsigned>(n));
      if (kind == SPR_F16)
        hipLaunchKernelGGL(enet_pool16_kernel<SPR_F16>, grid, dim3(kThreads), 0, s, cur, h * w, o.cin_p, pooled);
      else
        hipLaunchKernelGGL(enet_pool16_kernel<SPR_BF16>, grid, dim3(kThreads), 0, s, cur, h * w, o.cin_p, pooled);
      rc = check_launch("enet_pool16_kernel");
      if (rc != SPR_OK) return rc;
      rc = launch_enet_fc(pooled, n, o.cin_p, o.sq, pk + o.w_off, pk + o.b_off, pk + o.w2_off, pk + o.b2_off, hidden, factors, s);
      if (rc != SPR_OK) return rc;
      scale = factors;
    }
  }
  return SPR_OK;
}

extern "C" int spr_effnet_forward(spr_effnet_plan* plan, const uint8_t* images, int64_t n, int32_t in_h, int32_t in_w,
                                  int32_t in_channels, const float* mean3, const float* inv_std3, const void* packed,
                                  void* workspace, float* out, spr_stream_t stream) {
  if (!plan) { set_error("spr_effnet_forward: null plan"); return SPR_ERR_ARG; }
  if (n < 0 || n > 65535 || in_h < 32 || in_w < 32 || (in_channels != 1 && in_channels != 3)) {
    set_error("spr_effnet_forward: bad sizes (n in [0, 65535], images at least 32 x 32, in_channels 1 or 3)");
    return SPR_ERR_ARG;
  }
  if (n == 0) return SPR_OK;
  if (!images || !mean3 || !inv_std3 || !packed || !out || !workspace) { set_error("spr_effnet_forward: null pointer"); return SPR_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* pk = static_cast<const float*>(packed);
  const size_t buf_bytes = align_up(effnet_buf_floats(plan, n, in_h, in_w) * sizeof(float), 256);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  if (plan->compute != SPR_F32)
    return effnet_forward16(plan, images, n, in_h, in_w, in_channels, mean3, inv_std3, pk, ws, buf_bytes, out, s);
  float* x = reinterpret_cast<float*>(ws);                    // block input
  float* t1 = reinterpret_cast<float*>(ws + buf_bytes);
  float* t2 = reinterpret_cast<float*>(ws + 2 * buf_bytes);
  float* y = reinterpret_cast<float*>(ws + 3 * buf_bytes);    // block output
  float* pooled = reinterpret_cast<float*>(ws + 4 * buf_bytes);
  float* factors = pooled + align_up(static_cast<size_t>(n) * plan->max_expand_p * sizeof(float), 256) / sizeof(float);
  float* hidden = factors + align_up(static_cast<size_t>(n) * plan->max_expand_p * sizeof(float), 256) / sizeof(float);
  const size_t pixels = static_cast<size_t>(n) * in_h * in_w;
  hipLaunchKernelGGL(enet_input_kernel, dim3(static_cast<unsigned>(std::min<size_t>((pixels + kThreads - 1) / kThreads, 65535 * 16))),
                     dim3(kThreads), 0, s, images, pixels, in_channels, mean3[0], mean3[1], mean3[2], inv_std3[0], inv_std3[1],
                     inv_std3[2], x);
  int rc = check_launch("enet_input_kernel");
  if (rc != SPR_OK) return rc;
  int h = in_h, w = in_w;
  const float* cur = x;   // what the next layer reads
  float* tmp[2] = {t1, t2};
  int ti = 0;
  const float* scale = nullptr;
  for (size_t i = 0; i < plan->ops.size(); ++i) {
    const EOp& o = plan->ops[i];
    const bool last = i + 1 == plan->ops.size();
    if (o.kind == 0) {
      float* dst = last ? out : o.block_end ? y : tmp[ti];
      const float* res = o.res ? x : nullptr;
      const float* sc = o.scaled ? scale : nullptr;
      if (o.ks == 3 && o.stride == 2) rc = launch_egemm<3, 2>(o, cur, n, h, w, pk, res, sc, last ? 1 : 0, dst, s);
      else if (o.ks == 3) rc = launch_egemm<3, 1>(o, cur, n, h, w, pk, res, sc, last ? 1 : 0, dst, s);
      else rc = launch_egemm<1, 1>(o, cur, n, h, w, pk, res, sc, last ? 1 : 0, dst, s);
      if (rc != SPR_OK) return rc;
      if (o.stride == 2) { h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1; }
      if (o.block_end) {  // the block's output becomes the next block's input
        float* old = x; x = y; y = old;
        cur = x;
        ti = 0;
      } else {
        cur = dst;
        ti ^= 1;
      }
    } else if (o.kind == 1) {
      float* dst = tmp[ti];
      const int ho = (h - 1) / o.stride + 1, wo = (w - 1) / o.stride + 1;
      const size_t total = static_cast<size_t>(n) * ho * wo * (o.cin_p / 4);
      hipLaunchKernelGGL(enet_dw_kernel, dim3(static_cast<unsigned>(std::min<size_t>((total + kThreads - 1) / kThreads, 65535 * 16))),
                         dim3(kThreads), 0, s, cur, static_cast<int>(n), h, w, o.cin_p, o.stride, o.ks, pk + o.w_off, pk + o.b_off, dst);
      rc = check_launch("enet_dw_kernel");
      if (rc != SPR_OK) return rc;
      h = ho; w = wo;
      cur = dst;
      ti ^= 1;
    } else {
      hipLaunchKernelGGL(enet_pool_kernel, dim3(o.cin_p / 64, static_cast<unsigned>(n)), dim3(kThreads), 0, s, cur, h * w, o.cin_p,
                         pooled);
      rc = check_launch("enet_pool_kernel");
      if (rc != SPR_OK) return rc;
      rc = launch_enet_fc(pooled, n, o.cin_p, o.sq, pk + o.w_off, pk + o.b_off, pk + o.w2_off, pk + o.b2_off, hidden, factors, s);
      if (rc != SPR_OK) return rc;
      scale = factors;
    }
  }
  return SPR_OK;
}


// ================================================================ DenseNet_201 truncations (network.py:176-179, :185-186)
// torchvision's densenet201: features = [conv0, norm0, relu0, pool0, denseblock1, transition1, denseblock2, transition2,
// denseblock3, transition3, denseblock4, norm5]; the reference keeps features[:block], block in [1, 12].  A dense layer is
// BatchNorm + ReLU -> 1x1 convolution (128) -> BatchNorm + ReLU -> 3x3 convolution (32), concatenated behind its input: the
// first BatchNorm + ReLU runs while the 1x1 convolution loads its operand (per-channel affine + ReLU), the second is folded
// into that convolution, and the 3x3 convolution stores its 32 channels into the block's tensor at their offset.
namespace {
struct DOp {
  int kind;     // 0 stem (7x7 / 2 convolution [+ BatchNorm] [+ ReLU] [+ 3x3 / 2 max pool]), 1 dense 1x1, 2 dense 3x3,
                // 3 transition (BatchNorm + ReLU + 1x1 + 2x2 average pool), 4 closing BatchNorm
  int cin, cout;
  int c_off;    // dense 3x3: channel offset of its output in the block's tensor
  int ctot;     // channels of the tensor this layer reads (kinds 1, 3, 4) or writes into (kind 2)
  int flags;    // stem: 1 BatchNorm folded, 2 ReLU, 4 max pool
  int feature;
  size_t w_off, b_off, s_off, t_off;  // packed offsets (floats): weights, bias, pre-activation scale / shift
};
const int kDenseLayers[4] = {6, 12, 48, 32};
}  // namespace

struct spr_densenet_plan {
  int block;
  std::vector<DOp> ops;
  size_t packed_floats;
};

extern "C" int spr_densenet_plan_create(int32_t block, spr_densenet_plan** plan_out) {
  if (!plan_out) { set_error("spr_densenet_plan_create: null pointer"); return SPR_ERR_ARG; }
  *plan_out = nullptr;
  if (block < 1 || block > 12) { set_error("spr_densenet_plan_create: block %d: features[:block] with block in [1, 12]", block); return SPR_ERR_ARG; }
  spr_densenet_plan* plan = new (std::nothrow) spr_densenet_plan();
  if (!plan) { set_error("out of host memory"); return SPR_ERR_ARG; }
  plan->block = block;
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off += (n + 3) / 4 * 4; return o; };
  {
    DOp o{};
    o.kind = 0; o.cin = 3; o.cout = 64; o.feature = 0;
    o.flags = (block >= 2 ? 1 : 0) | (block >= 3 ? 2 : 0) | (block >= 4 ? 4 : 0);
    o.w_off = take(147 * 64); o.b_off = take(64);
    plan->ops.push_back(o);
  }
  int c = 64;
  for (int b = 0; b < 4 && 4 + 2 * b < block; ++b) {
    const int ctot = c + 32 * kDenseLayers[b];
    for (int l = 0; l < kDenseLayers[b]; ++l) {
      DOp a{};
      a.kind = 1; a.cin = c + 32 * l; a.cout = 128; a.ctot = ctot; a.feature = 4 + 2 * b;
      a.s_off = take(a.cin); a.t_off = take(a.cin);
      a.w_off = take(static_cast<size_t>(128) * a.cin); a.b_off = take(128);
      plan->ops.push_back(a);
      DOp d{};
      d.kind = 2; d.cin = 128; d.cout = 32; d.c_off = c + 32 * l; d.ctot = ctot; d.feature = 4 + 2 * b;
      d.w_off = take(static_cast<size_t>(64) * 128 * 9); d.b_off = take(64);  // output channels padded to the 64-wide tile
      plan->ops.push_back(d);
    }
    c = ctot;
    if (b < 3 && 5 + 2 * b < block) {
      DOp t{};
      t.kind = 3; t.cin = c; t.cout = c / 2; t.ctot = c; t.feature = 5 + 2 * b;
      t.s_off = take(c); t.t_off = take(c);
      t.w_off = take(static_cast<size_t>(c / 2) * c); t.b_off = take(c / 2);
      plan->ops.push_back(t);
      c /= 2;
    }
  }
  if (block == 12) {
    DOp n{};
    n.kind = 4; n.cin = n.cout = c; n.ctot = c; n.feature = 11;
    n.s_off = take(c); n.t_off = take(c);
    plan->ops.push_back(n);
  }
  plan->packed_floats = off;
  *plan_out = plan;
  return SPR_OK;
}

extern "C" void spr_densenet_plan_destroy(spr_densenet_plan* plan) { delete plan; }
extern "C" int spr_densenet_num_ops(const spr_densenet_plan* plan) { return plan ? static_cast<int>(plan->ops.size()) : SPR_ERR_ARG; }
extern "C" size_t spr_densenet_packed_bytes(const spr_densenet_plan* plan) { return plan ? plan->packed_floats * sizeof(float) : 0; }

// info[12] = kind, cin, cout, c_off, ctot, flags, feature, then the packed offsets (floats) w, b, s, t, then 0
extern "C" int spr_densenet_op_info(const spr_densenet_plan* plan, int32_t i, int32_t* info) {
  if (!plan || !info || i < 0 || i >= static_cast<int>(plan->ops.size())) { set_error("spr_densenet_op_info: bad argument"); return SPR_ERR_ARG; }
  const DOp& o = plan->ops[i];
  const int32_t v[12] = {o.kind, o.cin, o.cout, o.c_off, o.ctot, o.flags, o.feature, static_cast<int32_t>(o.w_off),
                         static_cast<int32_t>(o.b_off), static_cast<int32_t>(o.s_off), static_cast<int32_t>(o.t_off), 0};
  for (int k = 0; k < 12; ++k) info[k] = v[k];
  return SPR_OK;
}

static void densenet_dims(const spr_densenet_plan* plan, int in_h, int in_w, int* c, int* h, int* w) {
  int hh = (in_h + 1) / 2, ww = (in_w + 1) / 2, cc = 64;  // conv0: 7x7 s2 p3
  if (plan->block >= 4) { hh = (hh + 1) / 2; ww = (ww + 1) / 2; }
  for (const DOp& o : plan->ops) {
    if (o.kind == 2) cc = o.c_off + 32;
    if (o.kind == 3) { hh /= 2; ww /= 2; cc = o.cout; }
  }
  *c = cc; *h = hh; *w = ww;
}

extern "C" int spr_densenet_output_shape(const spr_densenet_plan* plan, int32_t in_h, int32_t in_w, int32_t* channels,
                                         int32_t* out_h, int32_t* out_w) {
  if (!plan || !channels || !out_h || !out_w || in_h < 1 || in_w < 1) { set_error("spr_densenet_output_shape: bad argument"); return SPR_ERR_ARG; }
  int c, h, w;
  densenet_dims(plan, in_h, in_w, &c, &h, &w);
  *channels = c; *out_h = h; *out_w = w;
  return SPR_OK;
}

// three buffers as large as the largest tensor: the stem's output (64 channels at half resolution) or a block's tensor
static size_t densenet_buf_floats(const spr_densenet_plan* plan, int64_t n, int in_h, int in_w) {
  int hh = (in_h + 1) / 2, ww = (in_w + 1) / 2;
  size_t best = static_cast<size_t>(n) * hh * ww * 64;
  if (plan->block >= 4) { hh = (hh + 1) / 2; ww = (ww + 1) / 2; }
  for (const DOp& o : plan->ops) {
    if (o.kind == 1 || o.kind == 3) {
      const size_t f = static_cast<size_t>(n) * hh * ww * o.ctot;
      if (f > best) best = f;
    }
    if (o.kind == 3) { hh /= 2; ww /= 2; }
  }
  return best;
}
extern "C" size_t spr_densenet_workspace_bytes(const spr_densenet_plan* plan, int64_t n, int32_t in_h, int32_t in_w) {
  if (!plan || n < 0) return 0;
  return 3 * align_up(densenet_buf_floats(plan, n, in_h, in_w) * sizeof(float), 256);
}

extern "C" int spr_densenet_forward(spr_densenet_plan* plan, const uint8_t* images, int64_t n, int32_t in_h, int32_t in_w,
                                    int32_t in_channels, const float* mean3, const float* inv_std3, const void* packed,
                                    void* workspace, float* out, spr_stream_t stream) {
  if (!plan) { set_error("spr_densenet_forward: null plan"); return SPR_ERR_ARG; }
  if (n < 0 || n > 65535 || in_h < 32 || in_w < 32 || (in_channels != 1 && in_channels != 3)) {
    set_error("spr_densenet_forward: bad sizes (n in [0, 65535], images at least 32 x 32, in_channels 1 or 3)");
    return SPR_ERR_ARG;
  }
  if (n == 0) return SPR_OK;
  if (!images || !mean3 || !inv_std3 || !packed || !out || !workspace) { set_error("spr_densenet_forward: null pointer"); return SPR_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* pk = static_cast<const float*>(packed);
  const size_t buf_bytes = align_up(densenet_buf_floats(plan, n, in_h, in_w) * sizeof(float), 256);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  float* cat = reinterpret_cast<float*>(ws);                 // the current block's tensor (or the stem's output)
  float* tmp = reinterpret_cast<float*>(ws + buf_bytes);     // a dense layer's 128-channel intermediate / a transition's output
  float* nxt = reinterpret_cast<float*>(ws + 2 * buf_bytes); // the next block's tensor
  auto blocks_of = [](size_t total) { return dim3(static_cast<unsigned>(std::min<size_t>((total + kThreads - 1) / kThreads, 65535 * 16))); };
  auto gemm = [&](int ks, const float* in, int h, int w, int cin, int cout_p, const DOp& o, int relu, float* dst, int cout_real,
                  int lda, int ldc, int c_off, bool pre) {
    const long long m = static_cast<long long>(n) * h * w;
    const dim3 grid(static_cast<unsigned>((m + kGM - 1) / kGM), static_cast<unsigned>(cout_p / kGN));
    const float* ps = pre ? pk + o.s_off : nullptr;
    const float* pt = pre ? pk + o.t_off : nullptr;
    if (ks == 3)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_gemm_kernel<3, 1>), grid, dim3(kThreads), 0, s, in, static_cast<int>(n), h, w, cin,
                         cout_p, pk + o.w_off, pk + o.b_off, static_cast<const float*>(nullptr), relu, 0, dst,
                         static_cast<const float*>(nullptr), cout_real, lda, ldc, c_off, ps, pt);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_gemm_kernel<1, 1>), grid, dim3(kThreads), 0, s, in, static_cast<int>(n), h, w, cin,
                         cout_p, pk + o.w_off, pk + o.b_off, static_cast<const float*>(nullptr), relu, 0, dst,
                         static_cast<const float*>(nullptr), cout_real, lda, ldc, c_off, ps, pt);
    return check_launch("conv_gemm_kernel");
  };
  int h = (in_h + 1) / 2, w = (in_w + 1) / 2, c = 64, ld = 64;
  int rc = SPR_OK;
  size_t i = 0;
  {
    const DOp& o = plan->ops[0];
    // the width of the first block's tensor, if there is one: the pooled stem output goes straight into its first 64 channels
    const int next_ld = plan->ops.size() > 1 && plan->ops[1].kind == 1 ? plan->ops[1].ctot : 64;
    const unsigned tiles = static_cast<unsigned>(ceil_div(h, 8) * ceil_div(w, 8));
    float* stem_out = (o.flags & 4) ? tmp : cat;
    hipLaunchKernelGGL(stem_kernel, dim3(tiles, static_cast<unsigned>(n)), dim3(kThreads), 0, s, images, in_h, in_w, in_channels,
                       mean3[0], mean3[1], mean3[2], inv_std3[0], inv_std3[1], inv_std3[2], pk + o.w_off, pk + o.b_off, stem_out,
                       (o.flags & 2) ? 1 : 0, 0);
    rc = check_launch("stem_kernel");
    if (rc != SPR_OK) return rc;
    if (o.flags & 4) {
      const int hp = (h + 1) / 2, wp = (w + 1) / 2;
      const size_t total = static_cast<size_t>(n) * hp * wp * 64;
      hipLaunchKernelGGL(maxpool3_kernel, blocks_of(total), dim3(kThreads), 0, s, tmp, h, w, 64, cat, total, next_ld);
      rc = check_launch("maxpool3_kernel");
      if (rc != SPR_OK) return rc;
      h = hp; w = wp; ld = next_ld;
    }
    i = 1;
  }
  const float* fin_s = nullptr;
  const float* fin_t = nullptr;
  for (; i < plan->ops.size(); ++i) {
    const DOp& o = plan->ops[i];
    if (o.kind == 1) {          // BatchNorm + ReLU (operand load) -> 1x1 -> BatchNorm (folded) + ReLU
      rc = gemm(1, cat, h, w, o.cin, 128, o, 1, tmp, 0, o.ctot, 128, 0, true);
    } else if (o.kind == 2) {   // 3x3, its 32 channels behind the layer's input
      rc = gemm(3, tmp, h, w, 128, 64, o, 0, cat, 32, 128, o.ctot, o.c_off, false);
      c = o.c_off + 32; ld = o.ctot;
    } else if (o.kind == 3) {   // BatchNorm + ReLU -> 1x1 -> 2x2 average pool into the next block's tensor
      rc = gemm(1, cat, h, w, o.cin, o.cout, o, 0, tmp, 0, o.ctot, o.cout, 0, true);
      if (rc != SPR_OK) return rc;
      const int next_ld = i + 1 < plan->ops.size() && plan->ops[i + 1].kind == 1 ? plan->ops[i + 1].ctot : o.cout;
      const size_t total = static_cast<size_t>(n) * (h / 2) * (w / 2) * o.cout;
      hipLaunchKernelGGL(dnet_avgpool_kernel, blocks_of(total), dim3(kThreads), 0, s, tmp, h, w, o.cout, nxt, total, next_ld);
      rc = check_launch("dnet_avgpool_kernel");
      float* old = cat; cat = nxt; nxt = old;
      h /= 2; w /= 2; c = o.cout; ld = next_ld;
    } else {                    // the closing BatchNorm rides on the layout change below
      fin_s = pk + o.s_off; fin_t = pk + o.t_off;
    }
    if (rc != SPR_OK) return rc;
  }
  const size_t total = static_cast<size_t>(n) * c * h * w;
  hipLaunchKernelGGL(dnet_out_kernel, blocks_of(total), dim3(kThreads), 0, s, cat, h * w, c, ld, fin_s, fin_t, 0, out, total);
  return check_launch("dnet_out_kernel");
}
