// Direct-form NCC on the bf16 matrix cores: the scorer for SMALL feature maps stored as bfloat16 (ResNet50 layer3
// and VGG16 conv5_3 shaped, cropped 28 x 12: BASELINE configs 3 and 5), where the FFT form spends its time on the
// prepared spectra (15 KB per pair and channel against 0.7 KB of raw features).
//
// Per channel the numerator of similarity.py:48-55 is a product of a [queries x taps] with a [taps x positions] matrix:
//     num[q, p] = sum_tap t0[q, tap] * I0z[p + tap]        (I0z = centred search map, zero outside: scipy 'same')
// A = the RAW template values: bfloat16 as stored, so every product below is exact.  Two forms of B:
//   exact (default): B = the RAW search map, zero outside - one MFMA per tile step, products exact on both sides - and
//       num = R - mean(I) * St0[q,p] - mean(t) * SI[p]   (St0: the centred template summed over the taps that fall inside
//       the map at p; SI: window sum of the raw map).  The last term is a weight of the epilogue; the middle one, summed
//       over the channels with the weights, is a [queries x channels] x [channels x gallery] product PER POSITION:
//       corr_mfma_kernel (f32 matrix cores, 0.6 % of the flops) writes it before the pair kernel subtracts it at the end.
//   split (SPR_NCC_MFMA_EXACT=0): B = I0z = hi + lo, two bfloat16 numbers (16 significant bits), two MFMAs per tile step,
//       num = R - mean(t) * S1[p]  (S1 = window sum of I0z, :59); self-contained, half as fast.
// B is never materialised: it is a Toeplitz gather from the zero-padded map of the channel, kept in LDS in eight copies
// shifted by one element each so that every lane's eight consecutive taps are one aligned ds_read_b128.
// A fragment (8 taps of 16 positions) depends on (row of the position + row of the tap) only, so one fragment read
// feeds up to NTG tiles of 16 positions, each against another template row pair: 78 fragment reads for 294 tile steps.
//
// One workgroup = one gallery item x 64 queries (one wave = 16 queries = the M side of v_mfma_f32_16x16x32_bf16), walking
// the channels: per channel 294 (split: 588) MFMAs per wave, then per position  sum += a[q] * (b[p] * acc) - (a*mean)[q] * (b*S)[p]
// (a = 1/sqrt(sum t0^2), b = 1/sigma of the window, both float32 from float64 statistics as in the other methods),
// channel sum in registers, spatial maximum and the running maximum over variants at the end (similarity.py:100-108,
// :355-367).
#include <cstdlib>
#include <type_traits>

#include "ncc_prep_common.h"

// -DSPR_MFMA_ABL=n: timing ablations (wrong results), tools/ubench/ablate_mfma.sh.  1: no epilogues, 2: one fragment read per
// period instead of one per step, 3: no staging of the next channel, 4: no reloads of the template fragments, 5: no barrier
// per channel
#ifndef SPR_MFMA_ABL
#define SPR_MFMA_ABL 0
#endif
// -DSPR_MFMA_GROUPS=n: n vector instructions behind every MFMA of a fragment step (scheduler groups); 0: the compiler's order
#ifndef SPR_MFMA_GROUPS
#define SPR_MFMA_GROUPS 0
#endif
// fragment steps between the LDS read of a fragment and its MFMAs
#ifndef SPR_MFMA_AHEAD
#define SPR_MFMA_AHEAD 2
#endif

namespace spr {

namespace {

// compile-time loop: f(integral_constant<int, I>) for I in [0, N) - register arrays indexed by I stay in registers
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// TH x TW: the FRAME of positions = the largest cropped search map of the instance (a smaller map sits in its top-left corner:
// the other positions carry 1/sigma = 0 and zero pixels, which is what the zero padding of 'same' mode puts there anyway).
// FH x 16: the template frame, FH even; a cropped template of th x tw taps sits in it so that its centre tap (th/2, tw/2) -
// the tap 'same' mode lays on the output pixel (similarity.py:55) - lands on (CY, CX) = (FH/2, CXF).  MCfg<28, 12> is the
// equal-size instance of BASELINE config 3 (template = map = 28 x 12: no spare taps, no spare positions);
// MCfg<28, 12, 30, 8> takes any template up to 30 x 16 on any map up to 28 x 12 (scaled / rotated query variants,
// similarity.py:230-284, and ragged sets) for one k-step more.
template <int TH_, int TW_, int FH_ = TH_, int CXF_ = TW_ / 2>
struct MCfg {
  static constexpr int TH = TH_, TW = TW_;  // frame of positions (cropped search map)
  static constexpr int FH = FH_;               // template frame rows
  static constexpr int NPOS = TH * TW;
  static constexpr int NTILE = NPOS / 16;      // tiles of 16 positions (row-major positions)
  static constexpr int KS = FH / 2;            // k-steps: two template rows of 16 taps (TW padded to 16 with zeros)
  static constexpr int TP = TW == 12 ? 3 : 2;  // tiles t and t + TP cover the same columns DY rows further down
  static constexpr int DY = 16 * TP / TW;
  static constexpr int NTG = NTILE / TP;
  static constexpr int SMAX = DY * (NTG - 1) + 2 * (KS - 1);
  static constexpr int PR = TH + FH - 1;  // rows of the zero-padded map
  static constexpr int CY = FH / 2, CX = CXF_;
  // largest template / map the instance takes
  static constexpr int kMaxTh = 2 * (FH - CY) < 2 * CY + 1 ? 2 * (FH - CY) : 2 * CY + 1;
  static constexpr int kMaxTw = 2 * (16 - CX) < 2 * CX + 1 ? 2 * (16 - CX) : 2 * CX + 1;
  static_assert(NPOS % 16 == 0 && FH % 2 == 0 && NTILE % TP == 0 && DY % 2 == 0 && TW <= 16, "unsupported map size");
  // LDS image of one channel: copy k (0..7) holds P shifted left by k elements, in chunks of 8 elements (16 bytes): chunk
  // (copy k, chunk column jj, row r) at jj * JS + r * RS + k * 16 - the eight copies of a row side by side (128 bytes), JS a
  // multiple of the 256-byte bank row.  ds_read_b128 is served in four groups of sixteen lanes that are NOT contiguous
  // ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS): of the layouts with a linear chunk -> bank-slot map this is the best,
  // 16 LDS cycles for the twelve (phase, group) reads of a fragment step against 12 without conflicts and 24 for the
  // first layout of this kernel (measured there: SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE).
  static constexpr int RS = 128;
  static constexpr int JS = (PR + 1) / 2 * 2 * RS;
  static constexpr int HL = 3 * JS;  // lo plane behind the hi plane
  static constexpr int kCopyBytes = 2 * HL;
  static constexpr int kEbOff = kCopyBytes;                // {b, b*S1}[NPOS] (float pairs)
  static constexpr int kDummyOff = kEbOff + 2 * 4 * NPOS;  // 16 bytes nobody reads: where masked-out stores go (no branches)
  static constexpr int kBufBytes = kDummyOff + 16;         // one channel; three buffers: channels c - 1 and c are read while
  static constexpr int kLdsBytes = 3 * kBufBytes;          // channel c + 1 is staged
  static constexpr int PERIOD = 2 * KS;                    // row offsets (even) one channel takes per tile group
  static_assert(DY * (NTG - 1) < PERIOD, "a tile group may lag the first one by less than a channel");
  static_assert(kBufBytes % 16 == 0, "buffer alignment");
  // prepared layouts
  static constexpr int kQMapBytes = FH * 16 * 2;           // per channel: the template frame, rows of 16 taps (16-bit)
  static constexpr int kGChanBytes = 3 * 4 * NPOS;         // per channel: b, b*S1 (float), hi|lo (one word per pixel)
  // exact form: U[position][channel] (query) and V[position][channel] (gallery) follow, channels padded to 16 (zeros)
  static constexpr int kXQ = 64;                           // queries per block = row length of the correction matrix
  // A fragment (phase ph, row offset s) reads rows s + y + {0, 1} of the padded map, y = the rows of the phase's sixteen
  // positions; the map proper occupies rows [CY, CY + TH): a fragment entirely above or below it is all zeros - neither read
  // nor multiplied (10 of the 26 row offsets of a phase; the tile steps on them are 18 % of all)
  static constexpr bool frag_zero(int ph, int s) {
    const int ymin = (16 * ph) / TW, ymax = (16 * ph + 15) / TW;
    return s + ymax + 1 < CY || s + ymin >= CY + TH;
  }
  // the first k-step of tile (tg, ph) whose fragment is not all zeros: its MFMA starts the accumulator
  static constexpr int first_ks(int tg, int ph) {
    for (int ks = 0; ks < KS; ++ks)
      if (!frag_zero(ph, DY * tg + 2 * ks)) return ks;
    return KS;
  }
};
__host__ __device__ inline int pad16(int c) { return (c + 15) / 16 * 16; }

struct MfmaArgs {
  int channels, nq, ng;
  long long ld, col0;
  int accumulate;
  unsigned q_item_bytes, g_item_bytes;
  // exact form: x[position][gallery item of this launch][query of the block] = sum_c U V, subtracted from the channel sums
  const float* x;
  int x_items;
  int ih, iw;  // the real (cropped) search map inside the frame of positions: what spr_ncc_maps writes
};

__device__ __forceinline__ unsigned bf16_round(float v) {  // round to nearest even, finite inputs
  const unsigned u = __float_as_uint(v);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_value(unsigned bits) { return __uint_as_float(bits << 16); }
// a float as the plan's 16-bit storage type (bit pattern) and back
__device__ __forceinline__ unsigned to_storage(float v, int dtype) {
  if (dtype == SPR_BF16) return bf16_round(v);
  union { _Float16 h; uint16_t u; } c;
  c.h = static_cast<_Float16>(v);
  return c.u;
}
__device__ __forceinline__ float from_storage(unsigned bits, int dtype) {
  if (dtype == SPR_BF16) return bf16_value(bits);
  union { _Float16 h; uint16_t u; } c;
  c.u = static_cast<uint16_t>(bits);
  return static_cast<float>(c.h);
}

// LDS of the prep kernel: the centred map (template or search map, whichever is larger) and two float64 summed-area tables.
// The equal-size instance holds exactly its 28 x 12 map (7.4 KB per wave-sized workgroup: 21 of them share a CU).
template <class M>
struct MPrepSizes {
  static constexpr bool kFixed = M::FH == M::TH;
  static constexpr int kMaxPix = kFixed ? M::NPOS : (M::FH * 16 > M::NPOS ? M::FH * 16 : M::NPOS);
  static constexpr int kSatElems = kFixed ? (M::TH + 1) * (M::TW + 1)
                                          : ((M::FH + 1) * 17 > (M::TH + 1) * (M::TW + 1) ? (M::FH + 1) * 17 : (M::TH + 1) * (M::TW + 1));
};

// ---- preparation: grid = (channels, items) ---------------------------------------------------------------------------
// EXACT form: the search map enters the matrix cores RAW (bf16 as stored: one MFMA per tile step, every product exact) and
// the centring of both sides becomes two corrections of the raw product R[q,p] = sum_tap t[q,tap] Iz[p + tap]:
//     num = R - mean(I) * St0[q,p] - mean(t) * SI[p],     St0 = sum of the centred template over the taps whose pixel lies
//                                                          inside the map at position p, SI = window sum of the raw map
// The last term is a weight of the pair kernel's epilogue like before (b * SI in place of b * S1); the middle one is a
// contraction over the channels per position, X[q,g,p] = sum_c (a St0)[q,c,p] * (b mean(I))[g,c,p], left to corr_mfma_kernel.
template <class M, bool EXACT, bool FIXED>
__global__ void __launch_bounds__(kThreads)
prep_mfma_kernel(NccGeom g, int is_query, const void* __restrict__ maps, unsigned char* __restrict__ prepared,
                 size_t item_bytes) {
  // Real sizes (cropped): template th x tw inside the FH x 16 frame, search map ih x iw inside the TH x TW frame.  FIXED (the
  // equal-size instance: template = map = the frame of positions): compile-time constants - with run-time sizes the kernel took
  // 51 ms instead of 31 for BASELINE config 3's gallery.
  const int th = FIXED ? M::TH : g.th, tw = FIXED ? M::TW : g.tw, ih = FIXED ? M::TH : g.ih, iw = FIXED ? M::TW : g.iw;
  const int fy = M::CY - th / 2, fx = M::CX - tw / 2;  // frame position of template tap (0, 0)
  constexpr int kMaxPix = MPrepSizes<M>::kMaxPix, kSatElems = MPrepSizes<M>::kSatElems;
  unsigned char* lds = dyn_lds();
  double* red = reinterpret_cast<double*>(lds);
  float* x0 = reinterpret_cast<float*>(lds + 64);
  double* sat1 = reinterpret_cast<double*>(lds + align_up(64 + sizeof(float) * kMaxPix, 16));
  double* sat2 = sat1 + kSatElems;
  const int c = static_cast<int>(blockIdx.x);
  const size_t item = blockIdx.y;
  const int tid = static_cast<int>(threadIdx.x);
  const int cp = pad16(g.channels);
  unsigned char* out_item = prepared + item * item_bytes;
  const uint16_t* raw = static_cast<const uint16_t*>(maps);
  auto build_tables = [&](int h, int w) {
    if (sat_blocked_fits(h, w))
      build_sat_pair_blocked(x0, h, w, sat1, sat2);
    else
      build_sat_pair(x0, h, w, sat1, sat2);
  };
  // Conditioning.  The raw operands make  num = R - corrections  a difference of numbers (mean / sigma)^2 times larger than
  // num (measured: 4e-4 on scores of maps offset by 100 sigma).  A channel whose values all sit near its mean - the only way
  // to have a large mean / sigma - can be shifted EXACTLY in its storage type: kappa = the mean rounded to the storage type,
  // x - kappa representable for every pixel (checked here, pixel by pixel).  Then x - kappa enters the matrix cores and
  // mean - kappa the corrections: the same algebra, exact products as before, and the cancellation is gone.  Channels spread
  // over several binades fail the check and stay as they are: their mean / sigma is small.
  auto exact_shift = [&](size_t base, int raw_w, int h, int w, float mean) {  // returns kappa (0: no exact shift exists)
    const float kappa = from_storage(to_storage(mean, g.dtype), g.dtype);
    double bad = 0.0;
    for (int i = tid; i < h * w; i += wg_size()) {
      const int y = i / w, x = i - y * w;
      const float d = from_storage(raw[base + static_cast<size_t>(y + g.crop) * raw_w + (x + g.crop)], g.dtype) - kappa;
      if (from_storage(to_storage(d, g.dtype), g.dtype) != d) bad += 1.0;
    }
    return block_sum(bad, red) == 0.0 ? kappa : 0.0f;
  };
  auto shifted_bits = [&](size_t idx, float kappa) -> unsigned {
    const unsigned bits = raw[idx];
    return kappa == 0.0f ? bits : to_storage(from_storage(bits, g.dtype) - kappa, g.dtype);
  };
  // the channel columns that pad U / V to a multiple of 16 are zero: the last channel's workgroup writes them
  auto store_column = [&](float* mat, int pos, float v) {
    mat[static_cast<size_t>(pos) * cp + c] = v;
    if (c == g.channels - 1)
      for (int k = g.channels; k < cp; ++k) mat[static_cast<size_t>(pos) * cp + k] = 0.0f;
  };
  if (is_query) {
    const size_t base = (item * g.channels + c) * static_cast<size_t>(g.q_h) * g.q_w;
    float mean;
    load_centred(maps, base, g.q_w, g.crop, th, tw, g.dtype, x0, red, &mean);
    const float scale = template_scale(x0, th * tw, red);
    const float kappa = exact_shift(base, g.q_w, th, tw, mean);
    // the template frame in the storage type (taps as stored, or shifted by kappa; zero outside the template)
    uint16_t* rows = reinterpret_cast<uint16_t*>(out_item + static_cast<size_t>(c) * M::kQMapBytes);
    for (int i = tid; i < M::FH * 16; i += wg_size()) {
      const int u = (i >> 4) - fy, v = (i & 15) - fx;
      rows[i] = (u >= 0 && u < th && v >= 0 && v < tw)
                    ? static_cast<uint16_t>(shifted_bits(base + static_cast<size_t>(u + g.crop) * g.q_w + (v + g.crop), kappa))
                    : static_cast<uint16_t>(0);
    }
    float* sc = reinterpret_cast<float*>(out_item + static_cast<size_t>(g.channels) * M::kQMapBytes);
    if (tid == 0) {
      sc[2 * c] = scale;
      sc[2 * c + 1] = scale * (mean - kappa);
    }
    if constexpr (EXACT) {
      float* U = sc + 2 * g.channels;
      build_tables(th, tw);
      const int stride = tw + 1;
      for (int i = tid; i < M::NPOS; i += wg_size()) {
        const int y = i / M::TW, x = i - y * M::TW;
        // taps (u, v) whose pixel (y + u - th/2, x + v - tw/2) lies inside the map
        const int u0 = th / 2 - y > 0 ? th / 2 - y : 0, u1 = th / 2 - y + ih < th ? th / 2 - y + ih : th;
        const int v0 = tw / 2 - x > 0 ? tw / 2 - x : 0, v1 = tw / 2 - x + iw < tw ? tw / 2 - x + iw : tw;
        double st0 = 0.0;
        if (y < ih && x < iw && u1 > u0 && v1 > v0)
          st0 = sat1[u1 * stride + v1] - sat1[u0 * stride + v1] - sat1[u1 * stride + v0] + sat1[u0 * stride + v0];
        store_column(U, i, scale * static_cast<float>(st0));
      }
    }
  } else {
    const size_t base = (item * g.channels + c) * static_cast<size_t>(g.g_h) * g.g_w;
    float mean;
    load_centred(maps, base, g.g_w, g.crop, ih, iw, g.dtype, x0, red, &mean);
    const float kappa = EXACT ? exact_shift(base, g.g_w, ih, iw, mean) : 0.0f;
    if constexpr (EXACT) mean -= kappa;  // from here on: the mean of the map as the matrix cores see it
    float* eb = reinterpret_cast<float*>(out_item + static_cast<size_t>(c) * M::kGChanBytes);
    float* ebs = eb + M::NPOS;
    unsigned* hl = reinterpret_cast<unsigned*>(ebs + M::NPOS);
    for (int i = tid; i < M::NPOS; i += wg_size()) {
      const int y = i / M::TW, x = i - y * M::TW;
      const bool inside = y < ih && x < iw;
      if constexpr (EXACT) {
        hl[i] = inside ? shifted_bits(base + static_cast<size_t>(y + g.crop) * g.g_w + (x + g.crop), kappa) << 16 : 0u;
      } else {
        const float v = inside ? x0[y * iw + x] : 0.0f;
        const unsigned hi = bf16_round(v);
        const unsigned lo = bf16_round(v - bf16_value(hi));
        hl[i] = (hi << 16) | lo;
      }
    }
    build_tables(ih, iw);
    if constexpr (EXACT) {  // the mean of the channel, behind V: vcol_mfma_kernel turns b and the means into V[position][channel]
      float* means = reinterpret_cast<float*>(out_item + static_cast<size_t>(g.channels) * M::kGChanBytes) +
                     static_cast<size_t>(M::NPOS) * cp;
      if (tid == 0) means[c] = mean;
    }
    const double inv_n = 1.0 / static_cast<double>(th * tw);
    for (int i = tid; i < M::NPOS; i += wg_size()) {
      const int y = i / M::TW, x = i - y * M::TW;
      if (y >= ih || x >= iw) {  // a position of the frame outside the map: weight 0
        eb[i] = 0.0f;
        ebs[i] = 0.0f;
        continue;
      }
      const double s1 = window_sum(sat1, ih, iw, th, tw, y, x);
      const double s2 = window_sum(sat2, ih, iw, th, tw, y, x);
      const float inv = inv_sigma_from_sums(s1, s2, inv_n);
      eb[i] = inv;
      if constexpr (EXACT) {
        const int y0 = y - th / 2 > 0 ? y - th / 2 : 0, y1 = y - th / 2 + th < ih ? y - th / 2 + th : ih;
        const int xa = x - tw / 2 > 0 ? x - tw / 2 : 0, xb = x - tw / 2 + tw < iw ? x - tw / 2 + tw : iw;
        const double si = s1 + static_cast<double>(mean) * static_cast<double>((y1 - y0) * (xb - xa));  // window sum of the raw map
        ebs[i] = inv * static_cast<float>(si);
      } else {
        ebs[i] = inv * static_cast<float>(s1);
      }
    }
  }
}

// ---- gallery preparation of the equal-size instance (template = map = frame): one wave = TWO channels -------------------
// The general kernel above spends ~7 000 SIMD cycles on a 336-pixel map, most of them in the summed-area tables built for maps
// of any size.  With template = map every window is a corner window - rows [max(0, y - TH/2), min(TH, y + TH/2)), columns
// likewise - so the two window sums are a prefix or a suffix in either direction and no table is needed:
//   lane = (channel of the pair, row): its 12 pixels stay in registers; the column-range sums of its row (float64 prefix, 12
//   values per table) go to LDS; 48 lanes = (table, channel, column) run down the 28 rows (prefix, then the row-range sum of
//   every y) and write them back in place; the row lanes pick their 2 x 12 window sums up again and finish 1/sigma, the raw
//   window sum and the pixel words.  Same statistics as prep_mfma_kernel (float64 sums of float32 values and float32 squares,
//   similarity.py:57-62), formed in another order: a few ulp of float64 apart.  grid = (ceil(channels / 2), items), 64 lanes.
template <class M, bool EXACT>
__global__ void __launch_bounds__(64)
prep_gallery_fixed_kernel(NccGeom g, const void* __restrict__ maps, unsigned char* __restrict__ prepared, size_t item_bytes) {
  constexpr int H = M::TH, W = M::TW, N = M::NPOS;
  static_assert(H <= 32 && W % 4 == 0 && 4 * W <= 64 && M::FH == M::TH, "one row per lane, two channels per wave");
  __shared__ double tab[2][2][H][W];
  const int lane = static_cast<int>(threadIdx.x), half = lane >> 5, row = lane & 31;
  const int c = 2 * static_cast<int>(blockIdx.x) + half;
  const size_t item = blockIdx.y;
  const bool valid = row < H && c < g.channels;
  const int cc = c < g.channels ? c : g.channels - 1, rr = row < H ? row : H - 1;
  const uint16_t* raw = static_cast<const uint16_t*>(maps) +
                        ((item * g.channels + cc) * static_cast<size_t>(g.g_h) + (rr + g.crop)) * g.g_w + g.crop;
  unsigned bits[W];
#pragma unroll
  for (int k = 0; k < W; ++k) bits[k] = raw[k];
  float v[W];
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < W; ++k) {
    v[k] = valid ? from_storage(bits[k], g.dtype) : 0.0f;
    s += static_cast<double>(v[k]);
  }
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) s += shfl_xor(s, m);  // over the 32 lanes of this channel
  float mean = static_cast<float>(s / static_cast<double>(N));
  // exact shift (see prep_mfma_kernel): kappa = the mean in the storage type, taken if x - kappa is representable everywhere
  float kappa = 0.0f;
  if constexpr (EXACT) {
    kappa = from_storage(to_storage(mean, g.dtype), g.dtype);
    int bad = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float d = v[k] - kappa;
      if (valid && from_storage(to_storage(d, g.dtype), g.dtype) != d) bad = 1;
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) bad |= shfl_xor(bad, m);
    if (bad) kappa = 0.0f;
  }
  unsigned char* out_item = prepared + item * item_bytes;
  float* eb = reinterpret_cast<float*>(out_item + static_cast<size_t>(cc) * M::kGChanBytes);
  float* ebs = eb + N;
  unsigned* hl = reinterpret_cast<unsigned*>(ebs + N);
  float x0[W];
  unsigned word[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    x0[k] = v[k] - mean;
    if constexpr (EXACT) {
      word[k] = (kappa == 0.0f ? bits[k] : to_storage(v[k] - kappa, g.dtype)) << 16;
    } else {
      const unsigned hi = bf16_round(x0[k]);
      word[k] = (hi << 16) | bf16_round(x0[k] - bf16_value(hi));
    }
  }
  if (valid) {
#pragma unroll
    for (int k = 0; k < W; k += 4)
      *reinterpret_cast<u32x4*>(hl + row * W + k) = u32x4{word[k], word[k + 1], word[k + 2], word[k + 3]};
  }
  if constexpr (EXACT) {
    mean -= kappa;  // from here on: the mean of the map as the matrix cores see it
    if (valid && row == 0) {
      float* means = reinterpret_cast<float*>(out_item + static_cast<size_t>(g.channels) * M::kGChanBytes) +
                     static_cast<size_t>(N) * pad16(g.channels);
      means[c] = mean;
    }
  }
  // column-range sums of this row: window columns [max(0, x - W/2), min(W, x + W/2))
  {
    double p1[W + 1], p2[W + 1];
    p1[0] = 0.0; p2[0] = 0.0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float sq = x0[k] * x0[k];  // np.square keeps float32 (similarity.py:57)
      p1[k + 1] = p1[k] + static_cast<double>(valid ? x0[k] : 0.0f);
      p2[k + 1] = p2[k] + static_cast<double>(valid ? sq : 0.0f);
    }
    if (row < H) {
#pragma unroll
      for (int x = 0; x < W; ++x) {
        const int a = x - W / 2 > 0 ? x - W / 2 : 0, b = x + W / 2 < W ? x + W / 2 : W;
        tab[0][half][row][x] = p1[b] - p1[a];
        tab[1][half][row][x] = p2[b] - p2[a];
      }
    }
  }
  __syncthreads();
  if (lane < 4 * W) {  // (table, channel, column): down the rows
    double* col = &tab[0][0][0][0] + (lane / W) * (H * W) + lane % W;
    double P[H + 1];
    P[0] = 0.0;
#pragma unroll
    for (int r = 0; r < H; ++r) P[r + 1] = P[r] + col[r * W];
#pragma unroll
    for (int y = 0; y < H; ++y) {
      const int a = y - H / 2 > 0 ? y - H / 2 : 0, b = y + H / 2 < H ? y + H / 2 : H;
      col[y * W] = P[b] - P[a];
    }
  }
  __syncthreads();
  if (!valid) return;
  const double inv_n = 1.0 / static_cast<double>(N);
  const int y0 = row - H / 2 > 0 ? row - H / 2 : 0, y1 = row + H / 2 < H ? row + H / 2 : H;
  float o_b[W], o_bs[W];
#pragma unroll
  for (int x = 0; x < W; ++x) {
    const double s1 = tab[0][half][row][x], s2 = tab[1][half][row][x];
    const float inv = inv_sigma_from_sums(s1, s2, inv_n);
    o_b[x] = inv;
    if constexpr (EXACT) {
      const int xa = x - W / 2 > 0 ? x - W / 2 : 0, xb = x + W / 2 < W ? x + W / 2 : W;
      const double si = s1 + static_cast<double>(mean) * static_cast<double>((y1 - y0) * (xb - xa));  // raw window sum
      o_bs[x] = inv * static_cast<float>(si);
    } else {
      o_bs[x] = inv * static_cast<float>(s1);
    }
  }
#pragma unroll
  for (int k = 0; k < W; k += 4) {
    *reinterpret_cast<float4*>(eb + row * W + k) = float4{o_b[k], o_b[k + 1], o_b[k + 2], o_b[k + 3]};
    *reinterpret_cast<float4*>(ebs + row * W + k) = float4{o_bs[k], o_bs[k + 1], o_bs[k + 2], o_bs[k + 3]};
  }
}

// ---- gallery preparation of the general instance: the same two-channels-per-wave scheme for ANY template on any map of the
// frame.  The windows are no longer corner windows (rows [y - th/2, y - th/2 + th) clipped to the map), so the range sums are
// differences of INCLUSIVE prefixes picked at run-time positions: the prefixes go through LDS (a register array cannot be
// indexed at run time without scratch) - along the row (written and read back by the same lane), then down the columns.
template <class M, bool EXACT>
__global__ void __launch_bounds__(64)
prep_gallery_wave_kernel(NccGeom g, const void* __restrict__ maps, unsigned char* __restrict__ prepared, size_t item_bytes) {
  constexpr int H = M::TH, W = M::TW, N = M::NPOS;
  static_assert(H <= 32 && W % 4 == 0 && 4 * W <= 64, "one row per lane, two channels per wave");
  __shared__ double tab[2][2][H][W];
  const int th = g.th, tw = g.tw, ih = g.ih, iw = g.iw;
  const int lane = static_cast<int>(threadIdx.x), half = lane >> 5, row = lane & 31;
  const int c = 2 * static_cast<int>(blockIdx.x) + half;
  const size_t item = blockIdx.y;
  const bool chan_ok = c < g.channels;
  const bool in_map = row < ih && chan_ok;  // this lane holds a row of the map
  const int cc = chan_ok ? c : g.channels - 1, rr = row < ih ? row : ih - 1;
  const uint16_t* raw = static_cast<const uint16_t*>(maps) +
                        ((item * g.channels + cc) * static_cast<size_t>(g.g_h) + (rr + g.crop)) * g.g_w + g.crop;
  unsigned bits[W];
#pragma unroll
  for (int k = 0; k < W; ++k) bits[k] = raw[k < iw ? k : iw - 1];
  float v[W];
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < W; ++k) {
    v[k] = in_map && k < iw ? from_storage(bits[k], g.dtype) : 0.0f;
    s += static_cast<double>(v[k]);
  }
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) s += shfl_xor(s, m);
  float mean = static_cast<float>(s / static_cast<double>(ih * iw));
  float kappa = 0.0f;
  if constexpr (EXACT) {
    kappa = from_storage(to_storage(mean, g.dtype), g.dtype);
    int bad = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float d = v[k] - kappa;
      if (in_map && k < iw && from_storage(to_storage(d, g.dtype), g.dtype) != d) bad = 1;
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) bad |= shfl_xor(bad, m);
    if (bad) kappa = 0.0f;
  }
  unsigned char* out_item = prepared + item * item_bytes;
  float* eb = reinterpret_cast<float*>(out_item + static_cast<size_t>(cc) * M::kGChanBytes);
  float* ebs = eb + N;
  unsigned* hl = reinterpret_cast<unsigned*>(ebs + N);
  float x0[W];
  unsigned word[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const bool inside = in_map && k < iw;
    x0[k] = inside ? v[k] - mean : 0.0f;
    if constexpr (EXACT) {
      word[k] = inside ? (kappa == 0.0f ? bits[k] : to_storage(v[k] - kappa, g.dtype)) << 16 : 0u;
    } else {
      const unsigned hi = bf16_round(x0[k]);
      word[k] = (hi << 16) | bf16_round(x0[k] - bf16_value(hi));
    }
  }
  const bool frame_row = row < H && chan_ok;  // every row of the frame is written (zeros outside the map)
  if (frame_row) {
#pragma unroll
    for (int k = 0; k < W; k += 4)
      *reinterpret_cast<u32x4*>(hl + row * W + k) = u32x4{word[k], word[k + 1], word[k + 2], word[k + 3]};
  }
  if constexpr (EXACT) {
    mean -= kappa;
    if (chan_ok && row == 0) {
      float* means = reinterpret_cast<float*>(out_item + static_cast<size_t>(g.channels) * M::kGChanBytes) +
                     static_cast<size_t>(N) * pad16(g.channels);
      means[c] = mean;
    }
  }
  // inclusive prefixes along the row -> LDS -> the column-range sums of the row, back in place
  if (row < H) {
    double r1 = 0.0, r2 = 0.0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float sq = x0[k] * x0[k];  // np.square keeps float32 (similarity.py:57)
      r1 += static_cast<double>(x0[k]);
      r2 += static_cast<double>(sq);
      tab[0][half][row][k] = r1;
      tab[1][half][row][k] = r2;
    }
    double c1[W], c2[W];
#pragma unroll
    for (int x = 0; x < W; ++x) {
      int xa = x - tw / 2, xb = xa + tw;
      xa = xa < 0 ? 0 : (xa > iw ? iw : xa);
      xb = xb < 0 ? 0 : (xb > iw ? iw : xb);
      const double hi1 = xb > 0 ? tab[0][half][row][xb - 1] : 0.0, lo1 = xa > 0 ? tab[0][half][row][xa - 1] : 0.0;
      const double hi2 = xb > 0 ? tab[1][half][row][xb - 1] : 0.0, lo2 = xa > 0 ? tab[1][half][row][xa - 1] : 0.0;
      c1[x] = hi1 - lo1;
      c2[x] = hi2 - lo2;
    }
#pragma unroll
    for (int x = 0; x < W; ++x) {
      tab[0][half][row][x] = c1[x];
      tab[1][half][row][x] = c2[x];
    }
  }
  __syncthreads();
  if (lane < 4 * W) {  // (table, channel, column): inclusive prefix down the rows, in place
    double* col = &tab[0][0][0][0] + (lane / W) * (H * W) + lane % W;
    double P[H];
    double run = 0.0;
#pragma unroll
    for (int r = 0; r < H; ++r) {
      run += col[r * W];
      P[r] = run;
    }
#pragma unroll
    for (int r = 0; r < H; ++r) col[r * W] = P[r];
  }
  __syncthreads();
  if (!frame_row) return;
  const double inv_n = 1.0 / static_cast<double>(th * tw);
  int y0 = row - th / 2, y1 = y0 + th;
  y0 = y0 < 0 ? 0 : (y0 > ih ? ih : y0);
  y1 = y1 < 0 ? 0 : (y1 > ih ? ih : y1);
  float o_b[W], o_bs[W];
#pragma unroll
  for (int x = 0; x < W; ++x) {
    const double s1 = (y1 > 0 ? tab[0][half][y1 - 1][x] : 0.0) - (y0 > 0 ? tab[0][half][y0 - 1][x] : 0.0);
    const double s2 = (y1 > 0 ? tab[1][half][y1 - 1][x] : 0.0) - (y0 > 0 ? tab[1][half][y0 - 1][x] : 0.0);
    const bool inside = row < ih && x < iw;
    const float inv = inside ? inv_sigma_from_sums(s1, s2, inv_n) : 0.0f;
    o_b[x] = inv;
    if constexpr (EXACT) {
      int xa = x - tw / 2, xb = xa + tw;
      xa = xa < 0 ? 0 : (xa > iw ? iw : xa);
      xb = xb < 0 ? 0 : (xb > iw ? iw : xb);
      const double si = s1 + static_cast<double>(mean) * static_cast<double>((y1 - y0) * (xb - xa));  // raw window sum
      o_bs[x] = inside ? inv * static_cast<float>(si) : 0.0f;
    } else {
      o_bs[x] = inside ? inv * static_cast<float>(s1) : 0.0f;
    }
  }
#pragma unroll
  for (int k = 0; k < W; k += 4) {
    *reinterpret_cast<float4*>(eb + row * W + k) = float4{o_b[k], o_b[k + 1], o_b[k + 2], o_b[k + 3]};
    *reinterpret_cast<float4*>(ebs + row * W + k) = float4{o_bs[k], o_bs[k + 1], o_bs[k + 2], o_bs[k + 3]};
  }
}

// ---- exact form: V[p][c] = b[c][p] * mean[c] of one gallery item, 32 channels per workgroup through an LDS tile (the prep
// workgroups own one channel each: written from there, V would be 4-byte stores 4 KB apart).  grid = (channel blocks, items)
template <class M>
__global__ void __launch_bounds__(kThreads)
vcol_mfma_kernel(int channels, unsigned char* __restrict__ prepared, size_t item_bytes) {
  __shared__ float tile[32][M::NPOS + 1];
  const int tid = static_cast<int>(threadIdx.x);
  const int cp = pad16(channels), c0 = static_cast<int>(blockIdx.x) * 32;
  unsigned char* item = prepared + static_cast<size_t>(blockIdx.y) * item_bytes;
  float* V = reinterpret_cast<float*>(item + static_cast<size_t>(channels) * M::kGChanBytes);
  const float* means = V + static_cast<size_t>(M::NPOS) * cp;
  const int p1 = tid + kThreads < M::NPOS ? tid + kThreads : tid;  // second pixel of this work-item (or the first again)
  for (int cb = 0; cb < 32; cb += 8) {
    float v0[8], v1[8];  // eight channel rows in flight
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = c0 + cb + k < channels ? c0 + cb + k : channels - 1;
      const float* b = reinterpret_cast<const float*>(item + static_cast<size_t>(c) * M::kGChanBytes);
      const float m = c0 + cb + k < channels ? means[c] : 0.0f;
      v0[k] = b[tid] * m;
      v1[k] = b[p1] * m;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      tile[cb + k][tid] = v0[k];
      tile[cb + k][p1] = v1[k];
    }
  }
  __syncthreads();
  const int cc = tid & 31;
  if (c0 + cc < cp)
    for (int p = tid >> 5; p < M::NPOS; p += kThreads / 32) V[static_cast<size_t>(p) * cp + c0 + cc] = tile[cc][p];
}

// ---- exact form: X[p][g][q] = sum_c U[q][p][c] * V[g][p][c] on the f32 matrix cores (exact f32 products) ---------------
// grid = (blocks of 64 gallery items, positions); a wave = 16 queries x 64 gallery items, K = channels in steps of 16.
template <class M>
__global__ void __launch_bounds__(kThreads)
corr_mfma_kernel(int channels, int nq_here, int ng, const unsigned char* __restrict__ q_block, unsigned q_item_bytes,
                 unsigned u_off, const unsigned char* __restrict__ pg, unsigned g_item_bytes, unsigned v_off,
                 float* __restrict__ x, int x_items) {
  const int tid = static_cast<int>(threadIdx.x);
  const int lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kgrp = lane >> 4;
  const int cp = pad16(channels);
  const int p = static_cast<int>(blockIdx.y);
  const int g0 = static_cast<int>(blockIdx.x) * 64;
  const int q = wave * 16 + i16 < nq_here ? wave * 16 + i16 : nq_here - 1;
  const float* up = reinterpret_cast<const float*>(q_block + static_cast<size_t>(q) * q_item_bytes + u_off) +
                    static_cast<size_t>(p) * cp + 4 * kgrp;
  const float* vp[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int gi = g0 + 16 * nt + i16 < ng ? g0 + 16 * nt + i16 : ng - 1;
    vp[nt] = reinterpret_cast<const float*>(pg + static_cast<size_t>(gi) * g_item_bytes + v_off) + static_cast<size_t>(p) * cp +
             4 * kgrp;
  }
  f32x4 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the operands of the next k-step are requested before the MFMAs of this one (plain loop: load, wait, multiply - 8.5 ms
  // for config 3's gallery; so: 6.x)
  float4 a_next = *reinterpret_cast<const float4*>(up);
  float4 b_next[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) b_next[nt] = *reinterpret_cast<const float4*>(vp[nt]);
  for (int c0 = 0; c0 < cp; c0 += 16) {
    const float4 a = a_next;
    float4 b[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b[nt] = b_next[nt];
    const int cn = c0 + 16 < cp ? c0 + 16 : c0;  // (the last step re-reads its own operands)
    a_next = *reinterpret_cast<const float4*>(up + cn);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b_next[nt] = *reinterpret_cast<const float4*>(vp[nt] + cn);
    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float bv = m == 0 ? b[nt].x : m == 1 ? b[nt].y : m == 2 ? b[nt].z : b[nt].w;
        acc[nt] = mfma_f32_16x16x4(av[m], bv, acc[nt]);
      }
    }
  }
  // lane owns rows (queries) 4 * kgrp .. + 3 of its wave's sixteen, column (gallery item) i16 of every tile
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int gl = g0 + 16 * nt + i16;
    if (gl < ng)
      *reinterpret_cast<float4*>(x + (static_cast<size_t>(p) * x_items + gl) * M::kXQ + wave * 16 + 4 * kgrp) =
          float4{acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]};
  }
}

// ---- pair kernel: grid = (gallery items, blocks of 64 queries) ---------------------------------------------------------
template <class M, bool MAPS, bool EXACT, bool F16>
__global__ void __launch_bounds__(kThreads, 1)
pair_mfma_kernel(MfmaArgs g, const unsigned char* __restrict__ pq, const unsigned char* __restrict__ pg,
                 float* __restrict__ scores, float* __restrict__ maps_out) {
  unsigned char* lds = dyn_lds();
  const int tid = static_cast<int>(threadIdx.x);
  const int lane = tid & 63, wave = tid >> 6, col = lane & 15, kg = lane >> 4;
  const size_t gi = blockIdx.x;
  const int q0 = static_cast<int>(blockIdx.y) * 64;
  const int nq_here = g.nq - q0 < 64 ? g.nq - q0 : 64;
  const bool active = wave * 16 < nq_here;  // a wave without queries still stages the channel images

  // the borders of the padded map stay zero for every channel
  for (int i = tid; i < M::kLdsBytes / 16; i += kThreads) reinterpret_cast<float4*>(lds)[i] = float4{0.f, 0.f, 0.f, 0.f};

  // ---- gallery side: this lane stages pixels e0 = tid and e1 = tid + 256 of every channel ---------------------------
  const unsigned char* g_item = pg + gi * static_cast<size_t>(g.g_item_bytes);
  const bool has1 = tid + kThreads < M::NPOS;
  const int e1 = has1 ? tid + kThreads : tid;
  float st_b[2], st_bs[2];
  unsigned st_hl[2];
  auto stage_load = [&](int c) {
    const float* cb = reinterpret_cast<const float*>(g_item + static_cast<size_t>(c) * M::kGChanBytes);
    st_b[0] = cb[tid];                 st_b[1] = cb[e1];
    st_bs[0] = cb[M::NPOS + tid];      st_bs[1] = cb[M::NPOS + e1];
    st_hl[0] = __float_as_uint(cb[2 * M::NPOS + tid]);
    st_hl[1] = __float_as_uint(cb[2 * M::NPOS + e1]);
  };
  // low half: byte offset of the pixel in copy 0 (row, column); high half: its column in the padded map
  auto pixel_base = [&](int e) {
    const int iy = e / M::TW, ix = e - iy * M::TW;
    return ((iy + M::CY) * M::RS + 2 * (ix + M::CX)) | ((ix + M::CX) << 16);
  };
  const int pb0 = pixel_base(tid), pb1 = pixel_base(e1);
  // Branch-free on purpose: the channel loop below must stay ONE basic block, or the compiler sinks the epilogues of the
  // fragment steps before a branch into the block behind it, where they run with the matrix cores idle.  A copy that does
  // not hold the pixel stores to the buffer's dummy slot; a work-item without a second pixel stores its first one twice.
  // part k of 8: both pixels of this work-item into copy k (part 0 also the two epilogue weights); the parts go to different
  // fragment steps of the period - one burst of all 32 + 4 stores per wave holds up the fragment reads of all four waves
  auto stage_store_part = [&](unsigned char* buf, auto k_c) {
    constexpr int k = decltype(k_c)::value;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pb = e ? pb1 : pb0;
      const unsigned hl = st_hl[e];
      const int w = pb >> 16, rowb = pb & 0xffff;
      const int mm = w - k;  // column of the pixel in copy k
      const int addr = k * 16 - 2 * k + rowb + (M::JS - 16) * (mm >> 3);
      const bool held = k <= M::CX || mm >= 0;  // w >= CX: only the far copies can push the first columns out
      *reinterpret_cast<uint16_t*>(buf + (held ? addr : M::kDummyOff)) = static_cast<uint16_t>(hl >> 16);
      if constexpr (!EXACT)
        *reinterpret_cast<uint16_t*>(buf + (held ? addr + M::HL : M::kDummyOff + 2)) = static_cast<uint16_t>(hl & 0xffffu);
    }
    if constexpr (k == 0) {  // the two epilogue weights of a pixel side by side: one 8-byte store here, one 8-byte read there
      f32x2* eb = reinterpret_cast<f32x2*>(buf + M::kEbOff);
      eb[tid] = f32x2{st_b[0], st_bs[0]};
      eb[e1] = f32x2{st_b[1], st_bs[1]};
    }
  };
  auto stage_store = [&](unsigned char* buf) {
    static_for<0, 8>([&](auto k_c) { stage_store_part(buf, k_c); });
  };

  // ---- query side --------------------------------------------------------------------------------------------------
  const unsigned char* q_block = pq + static_cast<size_t>(q0) * g.q_item_bytes;
  const BufRsrc q_rs = make_rsrc(q_block, static_cast<size_t>(nq_here) * g.q_item_bytes);
  auto clampq = [&](int q) { return q < nq_here ? q : nq_here - 1; };
  const unsigned a_off = static_cast<unsigned>(clampq(wave * 16 + col)) * g.q_item_bytes + static_cast<unsigned>(kg) * 16u;
  unsigned sc_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    sc_off[r] = static_cast<unsigned>(clampq(wave * 16 + 4 * kg + r)) * g.q_item_bytes +
                static_cast<unsigned>(g.channels) * M::kQMapBytes;
  auto load_a = [&](int c, int ks) { return buf_ld16v(q_rs, a_off, static_cast<unsigned>(c) * M::kQMapBytes + ks * 64); };

  // fragment read addresses: tile phase ph covers positions 16*ph + col (+ 16*TP per tile group)
  int frag_base[M::TP];
#pragma unroll
  for (int ph = 0; ph < M::TP; ++ph) {
    const int p = 16 * ph + col, y = p / M::TW, x = p - y * M::TW;
    const int m0 = x + 8 * (kg & 1), k = m0 & 7;
    frag_base[ph] = k * 16 + (m0 >> 3) * M::JS + (y + (kg >> 1)) * M::RS;
  }

  auto mfma = [](u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (F16) return mfma_f16_16x16x32(a, b, c);
    else return mfma_bf16_16x16x32(a, b, c);
  };
  // ---- steady-state schedule --------------------------------------------------------------------------------------
  // Time runs in row offsets tau = PERIOD * c + sigma (sigma = 0, 2, .. PERIOD - 2; three tile phases per step).  Tile group
  // tg works DY * tg behind the first one: at (c, sigma) it is at d = sigma - DY * tg of channel c if d >= 0, else at
  // d + PERIOD of channel c - 1 - so every step issues the same 2 * NTG MFMAs per phase, one tile in TP * DY / 2 steps is
  // finished and weighted, and the ramp-up / ramp-down of a channel-by-channel schedule (where a fragment read feeds one
  // or two MFMAs and the epilogues bunch at the end) is gone.  Channels c - 1 and c are both read from LDS while c + 1 is
  // staged (three buffers); the template fragments are double-buffered in registers (two periods unrolled), each reloaded
  // with channel c + 1 more than a channel before its first use.  The loop runs channels + 1 periods: in the first the
  // "previous channel" is all zeros (weights 0), in the last the "current" one is read but never weighted.
  f32x4 run[M::NTILE], acc[M::NTILE];
#pragma unroll
  for (int t = 0; t < M::NTILE; ++t) { run[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  u32x4 A[2][M::KS];
  const int last_c = g.channels - 1;
  if (active) {
#pragma unroll
    for (int ks = 0; ks < M::KS; ++ks) {
      A[0][ks] = load_a(0, ks);
      A[1][ks] = u32x4{0u, 0u, 0u, 0u};
    }
    // (the first PERIOD - DY * (NTG - 1) offsets of a channel are done with before the previous period ends)
#pragma unroll
    for (int ks = 0; ks < (M::PERIOD - M::DY * (M::NTG - 1)) / 2; ++ks) A[1][ks] = load_a(last_c < 1 ? last_c : 1, ks);
  }
  stage_load(0);
  __syncthreads();  // the zero fill is complete
  stage_store(lds);
  stage_load(last_c < 1 ? last_c : 1);
  __syncthreads();

  constexpr int NIT = M::KS * M::TP;       // fragment steps per period
  constexpr int IT_STAGE = 4, kStageEvery = 2;  // channel c + 1 is written to LDS in eight parts, every other step from here
  static_assert(IT_STAGE + 7 * kStageEvery < NIT, "staging must end inside the period");
  f32x2 sc_cur[4], sc_prev[4], sc_ld[4], sc_ld2[4];  // {a, a * mean} of this lane's four queries, channels c and c - 1; in flight
#pragma unroll
  for (int r = 0; r < 4; ++r) sc_cur[r] = f32x2{0.f, 0.f};
  float ebv = 0.f, ebsv = 0.f;             // 1/sigma and S1/sigma of the tile whose epilogue is due
  int b_prev = 2 * M::kBufBytes, b_cur = 0, b_next = M::kBufBytes;  // byte offsets of the three channel buffers

  auto period = [&](auto par_c, int c) {
    constexpr int PAR = decltype(par_c)::value;
    unsigned char* buf_cur = lds + b_cur;
    unsigned char* buf_prev = lds + b_prev;
    unsigned char* buf_next = lds + b_next;
    const int c1 = c + 1 < g.channels ? c + 1 : last_c, c2 = c + 2 < g.channels ? c + 2 : last_c;
    if (active) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (the loaded pair is first needed by the last steps of the period: its zeroing for the extra period waits until
        // mid-period - done here, the select made every period start with a full memory latency in front of its first MFMA)
        sc_prev[r] = sc_cur[r];
        if constexpr (PAR == 0) {  // (c is even here: one 16-byte load holds the pairs of channels c and c + 1)
          const u32x4 v = buf_ld16v(q_rs, sc_off[r], static_cast<unsigned>(c < g.channels ? c : (last_c & ~1)) * 8u);
          sc_ld[r] = f32x2{__uint_as_float(v[0]), __uint_as_float(v[1])};
          sc_ld2[r] = f32x2{__uint_as_float(v[2]), __uint_as_float(v[3])};
        } else {
          sc_ld[r] = sc_ld2[r];
        }
      }
      const f32x2* eb_cur = reinterpret_cast<const f32x2*>(buf_cur + M::kEbOff);
      const f32x2* eb_prev = reinterpret_cast<const f32x2*>(buf_prev + M::kEbOff);
      auto load_frag = [&](const unsigned char* buf, int ph, int s, int plane) {
        return *reinterpret_cast<const u32x4*>(buf + frag_base[ph] + s * M::RS + plane * M::HL);
      };
      // tile finished by fragment step `it` of a period: group 0 finishes the CURRENT channel in the last sigma step, group
      // tg > 0 the PREVIOUS channel at sigma = DY * tg - 2.  -1: none
      auto finished = [](int it) constexpr {
        const int sg = 2 * (it / M::TP), ph = it % M::TP;
        if (sg == M::PERIOD - 2) return ph;
        if ((sg + 2) % M::DY == 0 && (sg + 2) / M::DY >= 1 && (sg + 2) / M::DY < M::NTG) return ((sg + 2) / M::DY) * M::TP + ph;
        return -1;
      };
      auto epilogue = [&](int t, bool of_prev, int chan) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x2 w = of_prev ? sc_prev[r] : sc_cur[r];
          const float v = ebv * acc[t][r];
          run[t][r] = fmaf(-w.y, ebsv, fmaf(w.x, v, run[t][r]));
          if constexpr (MAPS) {  // spr_ncc_maps: the channel's own map of the first query
            if (r == 0 && wave == 0 && kg == 0 && chan >= 0 && chan < g.channels) {
              float m = fmaf(-w.y, ebsv, w.x * v);
              if constexpr (EXACT) {  // this channel's term of the contraction, for the one pair of spr_ncc_maps
                const int cp = pad16(g.channels), pos = 16 * t + col;
                const float* U = reinterpret_cast<const float*>(q_block + static_cast<size_t>(g.channels) * (M::kQMapBytes + 8));
                const float* V = reinterpret_cast<const float*>(g_item + static_cast<size_t>(g.channels) * M::kGChanBytes);
                m -= U[static_cast<size_t>(pos) * cp + chan] * V[static_cast<size_t>(pos) * cp + chan];
              }
              const int py = (16 * t + col) / M::TW, px = (16 * t + col) - py * M::TW;
              if (py < g.ih && px < g.iw) maps_out[(static_cast<size_t>(chan) * g.ih + py) * g.iw + px] = m;
            }
          }
        }
      };
      // fragment ring: the reads of step it + kAhead are issued before the MFMAs of step it (the scheduling fences keep
      // them there: left alone, the compiler sinks every read to just in front of its first use)
      constexpr int kAhead = SPR_MFMA_AHEAD, kRing = SPR_MFMA_AHEAD + 1;
      u32x4 fch[kRing], fcl[kRing], fph[kRing], fpl[kRing];
      auto issue = [&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        constexpr int sg = 2 * (j / M::TP), ph = j % M::TP, slot = j % kRing;
        if constexpr (!M::frag_zero(ph, sg)) {
          fch[slot] = load_frag(buf_cur, ph, sg, 0);
          if constexpr (!EXACT) fcl[slot] = load_frag(buf_cur, ph, sg, 1);
        }
        if constexpr (sg + M::PERIOD <= M::SMAX && !M::frag_zero(ph, sg + M::PERIOD)) {
          fph[slot] = load_frag(buf_prev, ph, sg + M::PERIOD, 0);
          if constexpr (!EXACT) fpl[slot] = load_frag(buf_prev, ph, sg + M::PERIOD, 1);
        }
      };
      static_for<0, kAhead>(issue);
      static_for<0, NIT>([&](auto it_c) {
        constexpr int it = decltype(it_c)::value;
        constexpr int sg = 2 * (it / M::TP), ph = it % M::TP, slot = it % kRing;
        if constexpr (it + kAhead < NIT && (SPR_MFMA_ABL != 2 || it + kAhead < kRing)) issue(std::integral_constant<int, it + kAhead>{});
        // the tile finished by the previous step (its MFMAs have had time to drain) is weighted and added now; the two
        // weights of the tile this step finishes are requested for the next
        constexpr int tdone = finished((it + NIT - 1) % NIT);
        float eb_now = ebv, ebs_now = ebsv;
        (void)eb_now; (void)ebs_now;
        constexpr int tnext = finished(it);
        float nb = 0.f, nbs = 0.f;
        if constexpr (tnext >= 0) {
          const f32x2 w = (tnext < M::TP ? eb_cur : eb_prev)[16 * tnext + col];
          nb = w.x;
          nbs = w.y;
        }
        sched_fence();
        if constexpr (tdone >= 0 && SPR_MFMA_ABL != 1) {
          // (group 0's last tile is finished by the last step of the PREVIOUS period: by now it belongs to channel c - 1)
          constexpr bool cur_chan = tdone < M::TP && it != 0;
          epilogue(tdone, !cur_chan, cur_chan ? c : c - 1);
        }
        if constexpr (tnext >= 0) { ebv = nb; ebsv = nbs; }
        if constexpr (it == NIT / 2) {
          static_assert(NIT / 2 < NIT - M::TP, "the current channel's weights are first used by the last sigma step");
#pragma unroll
          for (int r = 0; r < 4; ++r) sc_cur[r] = c < g.channels ? sc_ld[r] : f32x2{0.f, 0.f};
        }
        if constexpr (SPR_MFMA_ABL != 3 && it >= IT_STAGE && it < IT_STAGE + 8 * kStageEvery && (it - IT_STAGE) % kStageEvery == 0) {
          stage_store_part(buf_next, std::integral_constant<int, (it - IT_STAGE) / kStageEvery>{});
          if constexpr (it == IT_STAGE + 7 * kStageEvery) stage_load(c2);
        }
        static_for<0, M::NTG>([&](auto tg_c) {
          constexpr int tg = decltype(tg_c)::value;
          constexpr int d = sg - M::DY * tg;
          constexpr int t = tg * M::TP + ph;
          if constexpr (d >= 0) {
            constexpr int ks = d / 2;
            static_assert(M::first_ks(tg, ph) * 2 + M::DY * tg < M::PERIOD, "a tile starts in its own channel's period");
            if constexpr (!M::frag_zero(ph, sg)) {
              if constexpr (ks == M::first_ks(tg, ph))
                acc[t] = mfma(A[PAR][ks], fch[slot], f32x4{0.f, 0.f, 0.f, 0.f});
              else
                acc[t] = mfma(A[PAR][ks], fch[slot], acc[t]);
              if constexpr (!EXACT) acc[t] = mfma(A[PAR][ks], fcl[slot], acc[t]);
            }
          } else if constexpr (!M::frag_zero(ph, sg + M::PERIOD)) {
            constexpr int ks = (d + M::PERIOD) / 2;
            acc[t] = mfma(A[PAR ^ 1][ks], fph[slot], acc[t]);
            if constexpr (!EXACT) acc[t] = mfma(A[PAR ^ 1][ks], fpl[slot], acc[t]);
          }
        });
        // template row pairs nobody needs any more: the previous channel's (the slowest group has just used it) is
        // replaced by channel c + 1's, and at the end of the period the first ones of this channel by channel c + 2's
        // All of them go to the other buffer: its first (PERIOD - lag) / 2 row pairs hold channel c - 1's, which no tile
        // group reads in this period (the slowest one is at k-step (PERIOD - lag) / 2 when the period starts) - they are
        // reloaded in the FIRST steps.  Nothing is requested in the last steps of a period: the memory counter is in order,
        // so the first wait of the next period (for the staged pixels, requested half a period earlier) would sit out the
        // latency of a load requested just before it.
        if constexpr (SPR_MFMA_ABL != 4) {
          constexpr int lag = M::DY * (M::NTG - 1);
          constexpr int kFree = (M::PERIOD - lag) / 2;
          if constexpr (ph == M::TP - 1 && sg + M::PERIOD - lag < M::PERIOD) {
            constexpr int ks = (sg + M::PERIOD - lag) / 2;
            A[PAR ^ 1][ks] = load_a(c1, ks);
          } else if constexpr (ph == 0 && sg / 2 < kFree) {
            A[PAR ^ 1][sg / 2] = load_a(c1, sg / 2);
          }
        }
#if SPR_MFMA_GROUPS
        // the step's vector instructions (epilogue, staging addresses) in groups behind one MFMA each, not in one run
        static_for<0, M::NTG>([&](auto) {
          sched_group<0x8, 1>();
          sched_group<0x2, SPR_MFMA_GROUPS>();
        });
#endif
        sched_fence();
      });
    } else {
      stage_store(buf_next);
      stage_load(c2);
    }
    if (SPR_MFMA_ABL != 5) __syncthreads();  // channel c + 1 is staged; nobody reads channel c - 1's image any more
    const int t0 = b_prev;
    b_prev = b_cur; b_cur = b_next; b_next = t0;
  };
  for (int c = 0; c <= g.channels; c += 2) {
    period(std::integral_constant<int, 0>{}, c);
    if (c + 1 <= g.channels) period(std::integral_constant<int, 1>{}, c + 1);
  }
  // group 0's last tile of the last period is still to be weighted - with weight 0 (it belongs to "channel" channels)

  if (!active || !scores) return;  // spr_ncc_maps passes no score matrix
  // spatial maximum per query: over this lane's tiles, then over the sixteen lanes holding the other positions
  if constexpr (EXACT) {
#pragma unroll
    for (int t = 0; t < M::NTILE; ++t) {
      const float4 xv = *reinterpret_cast<const float4*>(g.x + (static_cast<size_t>(16 * t + col) * g.x_items + gi) * M::kXQ +
                                                         wave * 16 + 4 * kg);
      run[t][0] -= xv.x; run[t][1] -= xv.y; run[t][2] -= xv.z; run[t][3] -= xv.w;
    }
  }
  float best[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    best[r] = run[0][r];
#pragma unroll
    for (int t = 1; t < M::NTILE; ++t) best[r] = fmaxf(best[r], run[t][r]);
    for (int m = 8; m >= 1; m >>= 1) best[r] = fmaxf(best[r], shfl_xor(best[r], m));
  }
  if (col == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = wave * 16 + 4 * kg + r;
      if (q < nq_here) {
        const float s = best[r] / static_cast<float>(g.channels);
        float* dst = scores + static_cast<size_t>(q0 + q) * g.ld + g.col0 + gi;
        const float prev = g.accumulate ? *dst : 0.0f;
        *dst = s > prev ? s : prev;  // similarity.py:355, 366-367: zero-initialised running maximum
      }
    }
  }
}

using M2812 = MCfg<28, 12>;          // template = map = 28 x 12 (BASELINE config 3 / conv5_3 of config 5)
using MGEN = MCfg<28, 12, 30, 8>;    // any template up to 30 x 16 on any map up to 28 x 12
static_assert(MGEN::kMaxTh == 30 && MGEN::kMaxTw == 16, "general instance");

bool mfma_tuned_shape(const NccGeom& g) {
  return g.th == M2812::TH && g.tw == M2812::TW && g.ih == M2812::TH && g.iw == M2812::TW;
}
bool mfma_shape_ok(const NccGeom& g) {
  if (g.dtype != SPR_BF16 && g.dtype != SPR_F16) return false;
  if (mfma_tuned_shape(g)) return true;
  return g.th >= 1 && g.tw >= 1 && g.ih >= 1 && g.iw >= 1 && g.th <= MGEN::kMaxTh && g.tw <= MGEN::kMaxTw && g.ih <= MGEN::TH &&
         g.iw <= MGEN::TW;
}

// the instance of a plan: f(M2812{}) or f(MGEN{})
template <class F>
auto with_instance(const NccGeom& g, F&& f) {
  return g.mfma_general ? f(MGEN{}) : f(M2812{});
}

}  // namespace

constexpr int kCorrItems = 1024;  // gallery items per launch of the exact form = what the plan's correction matrix holds

bool mfma_geometry(NccGeom& g) {
  if (!mfma_shape_ok(g)) return false;
  g.mfma_general = mfma_tuned_shape(g) ? 0 : 1;
  // SPR_NCC_MFMA_EXACT=0: the centred search map as hi + lo (two MFMAs per tile step, no correction matrix)
  const char* e = std::getenv("SPR_NCC_MFMA_EXACT");
  g.mfma_exact = !(e && e[0] == '0') || g.dtype == SPR_F16;  // half-precision maps: the exact form only (both operands raw)
  // buffer-load offsets of a 64-query block and the scalar channel offsets are 32-bit
  return mfma_query_item_bytes(g) * 64 < (static_cast<size_t>(1) << 31);
}

size_t mfma_query_item_bytes(const NccGeom& g) {
  return with_instance(g, [&](auto m) {
    using M = decltype(m);
    size_t b = static_cast<size_t>(g.channels) * (M::kQMapBytes + 8);
    if (g.mfma_exact) b += sizeof(float) * M::NPOS * static_cast<size_t>(pad16(g.channels));
    return align_up(b, 256);
  });
}
size_t mfma_gallery_item_bytes(const NccGeom& g) {
  size_t b = static_cast<size_t>(g.channels) * M2812::kGChanBytes;  // (both instances: the 28 x 12 frame of positions)
  static_assert(M2812::kGChanBytes == MGEN::kGChanBytes && M2812::NPOS == MGEN::NPOS, "one frame of positions");
  if (g.mfma_exact) b += sizeof(float) * (M2812::NPOS + 1) * static_cast<size_t>(pad16(g.channels));  // V, then the means
  return align_up(b, 256);
}
size_t mfma_workspace_bytes(const NccGeom& g) {
  return g.mfma_exact ? sizeof(float) * M2812::NPOS * static_cast<size_t>(kCorrItems) * M2812::kXQ : 0;
}

template <class M>
static int launch_prep_mfma_m(const NccGeom& g, bool is_query, const void* maps, int64_t n, void* prepared, hipStream_t stream) {
  constexpr int kMaxPix = MPrepSizes<M>::kMaxPix, kSatElems = MPrepSizes<M>::kSatElems;
  const size_t lds = align_up(64 + sizeof(float) * kMaxPix, 16) + 2 * sizeof(double) * kSatElems;
  const size_t item_bytes = is_query ? mfma_query_item_bytes(g) : mfma_gallery_item_bytes(g);
  constexpr bool kFixed = M::FH == M::TH;  // (the equal-size instance is only chosen for template = map = frame)
  int rc;
  static const bool general_prep = std::getenv("SPR_MFMA_PREP") && std::atoi(std::getenv("SPR_MFMA_PREP")) == 0;  // A/B switch
  if constexpr (kFixed) {
    if (!is_query && !general_prep) {  // the gallery of the equal-size instance: two channels per wave, no tables
      auto fixed = g.mfma_exact ? prep_gallery_fixed_kernel<M, true> : prep_gallery_fixed_kernel<M, false>;
      hipLaunchKernelGGL(fixed, dim3((g.channels + 1) / 2, static_cast<unsigned>(n)), dim3(64), 0, stream, g, maps,
                         static_cast<unsigned char*>(prepared), item_bytes);
      rc = check_launch("prep_gallery_fixed_kernel");
      if (rc == SPR_OK && g.mfma_exact) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(vcol_mfma_kernel<M>), dim3((pad16(g.channels) + 31) / 32, static_cast<unsigned>(n)),
                           dim3(kThreads), 0, stream, g.channels, static_cast<unsigned char*>(prepared), item_bytes);
        rc = check_launch("vcol_mfma_kernel");
      }
      return rc;
    }
  }
  if constexpr (!kFixed) {
    if (!is_query && !general_prep) {  // the gallery of the general instance: the same scheme with run-time windows
      auto wave = g.mfma_exact ? prep_gallery_wave_kernel<M, true> : prep_gallery_wave_kernel<M, false>;
      hipLaunchKernelGGL(wave, dim3((g.channels + 1) / 2, static_cast<unsigned>(n)), dim3(64), 0, stream, g, maps,
                         static_cast<unsigned char*>(prepared), item_bytes);
      rc = check_launch("prep_gallery_wave_kernel");
      if (rc == SPR_OK && g.mfma_exact) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(vcol_mfma_kernel<M>), dim3((pad16(g.channels) + 31) / 32, static_cast<unsigned>(n)),
                           dim3(kThreads), 0, stream, g.channels, static_cast<unsigned char*>(prepared), item_bytes);
        rc = check_launch("vcol_mfma_kernel");
      }
      return rc;
    }
  }
  auto kernel = g.mfma_exact ? prep_mfma_kernel<M, true, kFixed> : prep_mfma_kernel<M, false, kFixed>;
  // one wave per (item, channel): maps of 336 pixels leave a 256-lane workgroup waiting at its ~20 barriers
  hipLaunchKernelGGL(kernel, dim3(g.channels, static_cast<unsigned>(n)), dim3(64), lds, stream, g, is_query ? 1 : 0, maps,
                     static_cast<unsigned char*>(prepared), item_bytes);
  rc = check_launch("prep_mfma_kernel");
  if (rc == SPR_OK && g.mfma_exact && !is_query) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(vcol_mfma_kernel<M>), dim3((pad16(g.channels) + 31) / 32, static_cast<unsigned>(n)),
                       dim3(kThreads), 0, stream, g.channels, static_cast<unsigned char*>(prepared), item_bytes);
    rc = check_launch("vcol_mfma_kernel");
  }
  return rc;
}

int launch_prep_mfma(const NccGeom& g, bool is_query, const void* maps, int64_t n, void* prepared, hipStream_t stream) {
  if (n == 0) return SPR_OK;
  return with_instance(g, [&](auto m) { return launch_prep_mfma_m<decltype(m)>(g, is_query, maps, n, prepared, stream); });
}

template <class M, bool EXACT, bool F16>
static int launch_pair_mfma_t(const NccGeom& g, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores,
                              int64_t ld, int64_t col0, int accumulate, float* maps_out, float* xws, hipStream_t stream) {
  auto kernel = maps_out ? pair_mfma_kernel<M, true, EXACT, F16> : pair_mfma_kernel<M, false, EXACT, F16>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
  const size_t q_item = mfma_query_item_bytes(g), g_item = mfma_gallery_item_bytes(g);
  const unsigned char* pqb = static_cast<const unsigned char*>(pq);
  const unsigned char* pgb = static_cast<const unsigned char*>(pg);
  if (!EXACT || maps_out) {
    const unsigned q_blocks = static_cast<unsigned>((nq + 63) / 64);
    // HIP refuses a grid of 2^32 work-items or more: slices of the gallery
    int64_t max_g = pair_tiles_per_launch(1, kThreads) / q_blocks;
    if (max_g < 1) max_g = 1;
    for (int64_t g0 = 0; g0 < ng; g0 += max_g) {
      const int64_t n = ng - g0 < max_g ? ng - g0 : max_g;
      MfmaArgs a{g.channels, static_cast<int>(nq), static_cast<int>(n), static_cast<long long>(ld),
                 static_cast<long long>(col0 + g0), accumulate, static_cast<unsigned>(q_item), static_cast<unsigned>(g_item),
                 nullptr, 0, g.ih, g.iw};
      hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(n), q_blocks), dim3(kThreads), M::kLdsBytes, stream, a, pqb,
                         pgb + static_cast<size_t>(g0) * g_item, scores, maps_out);
      const int rc = check_launch("pair_mfma_kernel");
      if (rc != SPR_OK) return rc;
    }
    return SPR_OK;
  }
  // exact form: per block of 64 queries and slice of the gallery, the correction matrix first, then the pairs
  if (!xws) { set_error("pair_mfma_kernel: the exact form needs the plan's correction matrix"); return SPR_ERR_ARG; }
  int64_t max_g = pair_tiles_per_launch(1, kThreads);
  if (max_g > kCorrItems) max_g = kCorrItems;
  const unsigned u_off = static_cast<unsigned>(static_cast<size_t>(g.channels) * (M::kQMapBytes + 8));
  const unsigned v_off = static_cast<unsigned>(static_cast<size_t>(g.channels) * M::kGChanBytes);
  for (int64_t q0 = 0; q0 < nq; q0 += 64) {
    const int nq_here = static_cast<int>(nq - q0 < 64 ? nq - q0 : 64);
    for (int64_t g0 = 0; g0 < ng; g0 += max_g) {
      const int n = static_cast<int>(ng - g0 < max_g ? ng - g0 : max_g);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(corr_mfma_kernel<M>), dim3((n + 63) / 64, M::NPOS), dim3(kThreads), 0, stream,
                         g.channels, nq_here, n, pqb + static_cast<size_t>(q0) * q_item, static_cast<unsigned>(q_item), u_off,
                         pgb + static_cast<size_t>(g0) * g_item, static_cast<unsigned>(g_item), v_off, xws, n);
      int rc = check_launch("corr_mfma_kernel");
      if (rc != SPR_OK) return rc;
      MfmaArgs a{g.channels, nq_here, n, static_cast<long long>(ld), static_cast<long long>(col0 + g0), accumulate,
                 static_cast<unsigned>(q_item), static_cast<unsigned>(g_item), xws, n, g.ih, g.iw};
      hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(n), 1), dim3(kThreads), M::kLdsBytes, stream, a,
                         pqb + static_cast<size_t>(q0) * q_item, pgb + static_cast<size_t>(g0) * g_item,
                         scores ? scores + static_cast<size_t>(q0) * ld : nullptr, maps_out);
      rc = check_launch("pair_mfma_kernel");
      if (rc != SPR_OK) return rc;
    }
  }
  return SPR_OK;
}

int launch_pair_mfma(const NccGeom& g, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores, int64_t ld,
                     int64_t col0, int accumulate, float* maps_out, float* xws, hipStream_t stream) {
  if (nq == 0 || ng == 0) return SPR_OK;
  return with_instance(g, [&](auto m) {
    using M = decltype(m);
    if (g.dtype == SPR_F16)
      return launch_pair_mfma_t<M, true, true>(g, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, xws, stream);
    return g.mfma_exact ? launch_pair_mfma_t<M, true, false>(g, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, xws, stream)
                        : launch_pair_mfma_t<M, false, false>(g, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, xws, stream);
  });
}

}  // namespace spr
