// Six-wave pair kernel of the FFT-method NCC scorer (the 192 x 96 grid: VGG16 conv3_3 maps of a 512x256 print).
//
// Same arithmetic as pair_fft_kernel (ncc_fft.hip): per channel, product of the two prepared half spectra ->
// inverse column transforms -> intermediate image in LDS -> inverse row transforms -> * 1/sigma -> channel sum in
// registers -> spatial maximum (similarity.py:100-108, :355-367).  What differs is how the work lies on the CU:
//
//   * 384 work-items (six waves) per pair and <= 168 VGPRs, so two pairs = TWELVE waves share a CU (three per
//     SIMD instead of two): the kernel is paced by LDS / L2 latency between short vector bursts, and the third
//     wave is what fills those gaps.  The two pairs are the two halves of ONE 768-lane workgroup (two queries
//     against the same gallery item): two independent 384-lane workgroups do not become co-resident - measured,
//     one per CU - because six waves land 2+2+1+1 on the four SIMDs and a second such workgroup only fits the
//     3-waves-per-SIMD register budget if the dispatcher happens to start it on the other SIMD pair.
//   * column pass: 24 groups of 16 lanes = 48 columns in exactly two rounds (the 256-lane kernel needs three).
//     12-point register stage as a twiddle-free 3 x 4 prime-factor transform.
//   * row pass: ONE round.  Every image row is a real-output transform of length 96 = a complex transform of
//     48 = 16 x 3 points on a group of three lanes (126 groups for the 124 rows): 16-point register stage,
//     3-point lane stage, no twiddles in between (prime-factor index maps).  The Hermitian pre-twist
//     Z[k] = Y[k] (1 + i w^k) + conj(Y[48-k]) (1 - i w^k) collapses to ONE packed fma per element because the
//     prep kernel scales query column k by (1 - i w^k):  Z[k] = i cot(pi k/96 + pi/4) Y'[k] + conj(Y'[48-k]).
//   * accumulators: 24 per lane (40 in the 256-lane kernel), 1/sigma slice 36.9 KB per channel (40.9).
//
// LDS per workgroup (151 KB): per pair {image 48 x 139 complex | per-wave exchange rows | Nyquist columns},
// shared {column twiddles | pre-twist table}.
#include <cstdlib>

#include "fft_core.h"
#include "ncc_fft_cfg.h"

namespace spr {
namespace {

// -DSPR_STAMPS: diagnostic build only (tools/ubench/stamps_pair6.py): lane 0 of every wave of ONE workgroup records
// the shader clock at the phase boundaries of eight channels into a buffer nothing else reads.
// -DSPR_ABL=n: timing ablations (wrong results), tools/ubench/ablate_pair6.sh.  1: no operand loads, 2: no pair
// barriers, 3: no LDS stores, 4: no 16-point stages, 5: no LDS reads of the exchanges / image
#ifndef SPR_ABL
#define SPR_ABL 0
#endif
// -DSPR_X6=n: layout experiments (right results), A/B builds of tools/ubench/x6_builds.sh.
// 1: conflict-free map of the row-pass exchange (Six::xpos).  Measured 3 % SLOWER than the round-2 map (304 k against 294 k
// pairs/s, kernel only, same box, both orders): three read bases instead of one; the conflicts it removes (~80 LDS cycles per
// wave and channel) are not what the kernel waits for.  Kept for A/B builds.
#ifndef SPR_X6
#define SPR_X6 0
#endif
#ifdef SPR_STAMPS
constexpr int kStampPoints = 12, kStampChannels = 8, kStampFirst = 8;
__device__ unsigned long long g_stamps[12 * kStampChannels * kStampPoints];
#define SPR_STAMP(i) st[i] = __builtin_amdgcn_s_memtime()
#else
#define SPR_STAMP(i)
#endif

constexpr int kTileQ6 = 16, kTileG6 = 16;  // pair -> workgroup tiling, as in ncc_fft.hip

struct Pair6Args {
  int channels, nq, ng;
  int ih, iw;
  int inv_per_chan;
  int accumulate;
  int tile0;    // first pair tile of this launch (launches are split so that the grid stays within HIP's limit)
  int tiles_g;  // pair tiles along the gallery
  int prio_mode;  // experiment: static wave priorities (SPR_P6_PRIO)
  unsigned q_flags_off, g_flags_off;  // byte offset of the per-channel dead flags inside a prepared item
};

// 12-point transform as 3 x 4 (Good-Thomas): input n = (4 n1 + 3 n2) mod 12, output k = (4 k1 + 9 k2) mod 12,
// X[k] = sum x[n] W3^(n1 k1) W4^(n2 k2).  48 packed instructions, no twiddle factors.
template <int DIR>
__device__ __forceinline__ void pfa12(cf (&x)[12]) {
  constexpr float h = DIR > 0 ? 0.86602540378443860f : -0.86602540378443860f;  // +-sin(2 pi / 3)
  cf a[3][4];
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    const cf x0 = x[(3 * n2) % 12], x1 = x[(4 + 3 * n2) % 12], x2 = x[(8 + 3 * n2) % 12];
    const cf s = x1 + x2, d = x1 - x2;
    const cf m = x0 - 0.5f * s;
    a[0][n2] = x0 + s;
    a[1][n2] = pk_iaxpy_u(d, h, m);   // m + i h d
    a[2][n2] = pk_iaxpy_u(d, -h, m);  // m - i h d
  }
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    const cf e0 = a[k1][0] + a[k1][2], e1 = a[k1][0] - a[k1][2];
    const cf o0 = a[k1][1] + a[k1][3], o1 = a[k1][1] - a[k1][3];
    x[(4 * k1) % 12] = e0 + o0;
    x[(4 * k1 + 18) % 12] = e0 - o0;
    x[(4 * k1 + 9) % 12] = DIR > 0 ? pk_add_i(e1, o1) : pk_sub_i(e1, o1);
    x[(4 * k1 + 27) % 12] = DIR > 0 ? pk_sub_i(e1, o1) : pk_add_i(e1, o1);
  }
}

template <class C>
struct Six {
  static_assert(C::SIX == 1 && C::EH == 12 && C::TGH == 16 && C::NW == 96 && C::NT == 384, "192 x 96 on six waves");
  static constexpr int NT = C::NT, WAVES = NT / 64;
  static constexpr int RS = 139;  // image row stride (complex): == 11 (mod 32), so the three lanes of a row group
                                  // and the 10 2/3 groups of a 32-lane LDS phase read 32 different banks
  // Row-pass exchange (16 rows of 63 lanes through the image rows the wave has just read): value r of lane u lies at
  // image slot 3 r + a(u), row offset o(u) of the wave's 21 rows.  a = u / 21, o = u % 21 (round 2) serves the reads - three
  // lanes of a row group read the values of lanes 3 grp + tt - with up to eleven two-way bank conflicts per instruction (the
  // 21-lane blocks of one exchange row start 11 banks apart, the groups inside a block 3).  This map makes the bank of a
  // value, 11 a + o, equal to u modulo 32 for every lane but one, which is what the 32-lane halves of ds_read_b64 need to be
  // conflict-free (the stores stay conflict-free per 16 lanes): {a, o} in image elements from the block's base.
  static __device__ __forceinline__ int xpos(int u) {
    if (SPR_X6 == 0) return (u / 21) * RS + u % 21;
    const int a = u < 21 ? 0 : (u < 32 ? 1 : (u < 43 ? 2 : (u < 53 ? 1 : 2)));
    const int o = u < 21 ? u : (u < 32 ? u - 11 : (u < 43 ? u - 22 : (u < 53 ? u - 43 : (u == 53 ? 9 : u - 54))));
    return a * RS + o;
  }
  static constexpr int XR = 65;   // exchange row: one slot per lane of the wave + 1 (bank skew between rows)
  static constexpr int XROWS = 6; // exchange rows in the wave's own buffer (the column pass puts six more into
                                  // the image columns the wave is about to write)
  static constexpr int kXbWave = XROWS * XR + 6;
  // byte offsets in LDS: per pair (half of the workgroup) ...
  static constexpr int kImgOff = 0;
  static constexpr int kImgBytes = C::COLS * RS * 8;
  static constexpr int kXbOff = kImgOff + kImgBytes;
  static constexpr int kNyqOff = kXbOff + WAVES * kXbWave * 8;
  static constexpr int kHalfBytes = kNyqOff + 2 * 2 * C::NH * 8;  // two buffers of Nyquist columns (see the channel loop)
  // ... and shared by both
  static constexpr int kRedOff = 2 * kHalfBytes;  // 2 x 8 floats of reduction scratch, then 2 barrier counters
  static constexpr int kTwOff = kRedOff + 128;
  static constexpr int kCtabOff = kTwOff + C::NH * 8;
  static constexpr int kListOff = kCtabOff + 48 * 4;  // two lists of live channels (uint16), sized at launch
  static constexpr int kLdsBytes = kListOff;
  static_assert(kLdsBytes + 2 * 2 * 1024 <= 160 * 1024, "one workgroup of two pairs per CU, up to 1024 channels");
  static_assert(kXbOff % 8 == 0 && kNyqOff % 8 == 0 && kHalfBytes % 16 == 0 && kTwOff % 8 == 0 && kCtabOff % 8 == 0, "");
};

template <class C>
__global__ void __launch_bounds__(2 * C::NT, 3)
pair6_kernel(Pair6Args g, const unsigned char* __restrict__ pq, size_t q_item_bytes,
             const unsigned char* __restrict__ pg, size_t g_item_bytes, float* __restrict__ scores, long long ld,
             long long col0, float* __restrict__ maps_out, const cf* __restrict__ tw_h,
             const float* __restrict__ ctab) {
  using S = Six<C>;
  constexpr int NT = S::NT, RS = S::RS, XR = S::XR;
  constexpr int H2 = C::EH / 2;  // 16-byte loads per operand and column unit

  // ---- which pairs ------------------------------------------------------------------------------
  // A workgroup scores two queries against one gallery item: lanes [0, 384) the first pair, [384, 768) the second, each
  // at its own pace (the gallery item's lines are then in L2 for whichever pair comes second).  Workgroups are dealt
  // round-robin over the 8 XCDs: workgroup w of a 16 x 16 tile (w % 8 = XCD) takes a 16-query x 2-gallery sub-tile,
  // so a gallery item's 112 KB per channel has 16 readers on one L2.
  constexpr int kWgPerTile = kTileQ6 * kTileG6 / 2;
  const int tile = g.tile0 + static_cast<int>(blockIdx.x) / kWgPerTile;
  const int within = static_cast<int>(blockIdx.x) % kWgPerTile;
  const int tq = tile / g.tiles_g, tg = tile - tq * g.tiles_g;
  const int half = uniform(static_cast<int>(threadIdx.x) >= NT ? 1 : 0);
  const int q_first = tq * kTileQ6 + 2 * (within >> 4);
  const int gi_item = tg * kTileG6 + 2 * (within & 7) + ((within >> 3) & 1);
  if (q_first >= g.nq || gi_item >= g.ng) return;  // uniform per workgroup
  const bool live = q_first + half < g.nq;          // an odd query count: the last workgroup's second half idles
  const int qi = live ? q_first + half : q_first;   // (it redoes the first half's pair and writes nothing)

  unsigned char* lds = dyn_lds();
  unsigned char* hl = lds + half * S::kHalfBytes;
  float* red = reinterpret_cast<float*>(lds + S::kRedOff) + half * 8;
  unsigned* bar = reinterpret_cast<unsigned*>(lds + S::kRedOff + 64) + half * 8;  // this pair's barrier counter
  unsigned* nlive_p = bar + 2;                                                   // its number of live channels
  unsigned bar_target = 0;
  cf* R = reinterpret_cast<cf*>(hl + S::kImgOff);
  cf* nyq = reinterpret_cast<cf*>(hl + S::kNyqOff);      // [buffer][0, NH): gallery column nw/2, [NH, 2NH): query's
  cf* twt_h = reinterpret_cast<cf*>(lds + S::kTwOff);    // w_NH^(+t p) at [p][t]
  cf* ctab2 = reinterpret_cast<cf*>(lds + S::kCtabOff);  // pre-twist cotangents, {c(a), c(a+1)} at [a/2][t]
  unsigned short* chan = reinterpret_cast<unsigned short*>(lds + S::kListOff) + half * g.channels;

  const int tid0 = static_cast<int>(threadIdx.x) - half * NT;  // lane of the pair
  const int wv = uniform(tid0 >> 6);
  cf* xb = reinterpret_cast<cf*>(hl + S::kXbOff) + wv * S::kXbWave;

  const unsigned char* g_item = pg + static_cast<size_t>(gi_item) * g_item_bytes;
  const unsigned char* q_item = pq + static_cast<size_t>(qi) * q_item_bytes;

  if (tid0 == 0) *bar = 0u;
  if (half == 0) {
    for (int k = tid0; k < C::NH; k += NT) twt_h[k] = cconj(tw_h[(k / C::TGH) * (k % C::TGH)]);
    if (tid0 < 24) {  // pair index = (a / 2) * 3 + t
      const int a = (tid0 / 3) * 2, t = tid0 % 3;
      ctab2[tid0] = cmake(ctab[(3 * a + 16 * t) % 48], ctab[(3 * (a + 1) + 16 * t) % 48]);
    }
  }
  // ---- live channels -------------------------------------------------------------------------------------------------
  // A channel that is constant (all zero after ReLU, typically) in the query or in the gallery item contributes exactly
  // 0 to every position (similarity.py:68-70: 0/0 -> NaN -> 0) while still counting in the divisor: the prep kernels
  // flag such channels per item, and the pair walks the list of channels live on BOTH sides.
  for (int k = tid0; k < g.channels; k += NT)
    chan[k] = (q_item[g.q_flags_off + k] | g_item[g.g_flags_off + k]) ? 0xFFFFu : static_cast<unsigned short>(k);
  __syncthreads();
  if (tid0 == 0) {  // in-place compaction, in channel order (the write index never passes the read index)
    int n = 0;
    for (int k = 0; k < g.channels; ++k) {
      const unsigned short v = chan[k];
      if (v != 0xFFFFu) chan[n++] = v;
    }
    if (n == 0) chan[0] = 0;  // (nothing live: the loop below does not run; the prologue's loads still need an address)
    *nlive_p = static_cast<unsigned>(n);
  }
  __syncthreads();
  const int nlive = uniform(static_cast<int>(*nlive_p));
  const int last_i = nlive > 0 ? nlive - 1 : 0;
  auto chan_at = [&](int i) { return static_cast<int>(chan[i < last_i ? i : last_i]); };  // (prefetches past the end re-read)

  // ---- operand streams ---------------------------------------------------------------------------------
  // Buffer loads: descriptor = one prepared item, lane offset = 16 * lane of the pair, the rest (channel, column
  // round, register) is a scalar offset.
  float4 nxt[2 * H2];  // operands of the next column unit: H2 x (2 complex of G), H2 x (2 complex of Q)
  cf nyq_g, nyq_q;     // (every lane loads both sides' Nyquist value and keeps its own: a select between two resource
                       // descriptors would put the load into a waterfall loop)
  const BufRsrc g_rs = make_rsrc(g_item, g_item_bytes), q_rs = make_rsrc(q_item, q_item_bytes);
  const unsigned voff16 = static_cast<unsigned>(tid0) * 16u;
  const unsigned voff_nyq = static_cast<unsigned>(tid0 % C::NH) * 8u + C::kNyqOffset * 8u;
  constexpr unsigned kChanBytes = C::kSpecPerChan * 8u;
  const unsigned inv_base = static_cast<unsigned>(g.channels) * kChanBytes;
  auto issue_unit = [&](int c, int rc) {  // column round rc of channel c: this wave's four columns
    const unsigned so = static_cast<unsigned>(c) * kChanBytes + static_cast<unsigned>(rc * H2 * NT * 16);
    if (SPR_ABL == 1) {
      for (int mm = 0; mm < 2 * H2; ++mm) nxt[mm] = make_float4(0.001f * so, 0.5f, 0.25f, 0.125f);
      return;
    }
#pragma unroll
    for (int mm = 0; mm < H2; ++mm) {
      nxt[mm] = buf_ld16(g_rs, voff16, so + mm * NT * 16);
      nxt[H2 + mm] = buf_ld16(q_rs, voff16, so + mm * NT * 16);
    }
  };
  auto issue_nyq = [&](int c) {  // 2 NH = NT values: lanes [0, NH) keep the gallery's, [NH, 2 NH) the query's
    const unsigned so = static_cast<unsigned>(c) * kChanBytes;
    nyq_g = buf_ld8(g_rs, voff_nyq, so);
    nyq_q = buf_ld8(q_rs, voff_nyq, so);
  };
  static_assert(2 * C::NH == NT, "one Nyquist value per lane");

#ifndef SPR_EMU
  {
    // Static wave priorities.  The SIMD arbitrates by priority, then age, and the twelve waves sit three to a SIMD
    // in dispatch order: without this the first pair holds the four oldest waves and the second pair the four
    // youngest, and every workgroup ends with the first pair's SIMD slots idle while the second finishes.
    // Raising waves 4, 5 and 8-11 gives each pair two waves of each rank (measured: +7 %; rotating the ranks per
    // channel instead: -20 %).
    const int gw = half * S::WAVES + wv;  // wave of the workgroup, in dispatch (= age) order
    if (g.prio_mode == 2 && (gw == 4 || gw == 5 || gw >= 8)) __builtin_amdgcn_s_setprio(1);
  }
#endif
  cf acc[6][2];
#pragma unroll
  for (int pp = 0; pp < 6; ++pp) acc[pp][0] = acc[pp][1] = cmake(0.0f, 0.0f);

  // ---- prologue ---------------------------------------------------------------------------------------------------------
  issue_nyq(chan_at(0));
  nyq[tid0] = tid0 < C::NH ? nyq_g : nyq_q;
  issue_nyq(chan_at(1));
  nyq[NT + tid0] = tid0 < C::NH ? nyq_g : nyq_q;
  issue_unit(chan_at(0), 0);
  __syncthreads();

  // The channel loop, per wave (A, B = its unit - four image columns - of column round 0 / 1; the LDS reads of an
  // exchange are issued before the arithmetic that follows them and consumed after it):
  //
  //   product + 12-point stage of A | B2 | product of B | exchange A | 12-point stage of B | 16-point stage + stores of A |
  //   exchange B | 16-point stage + stores of B | B1 | row pass: image reads, pre-twist, 16-point stage, exchange,
  //   3-point stage, * 1/sigma, accumulate
  //
  // The six waves of a pair meet twice per channel on a counter in LDS (group_barrier); the other pair of the workgroup
  // runs at its own pace - s_barrier would march all twelve waves through the same phase at the same time, each pipe busy
  // in bursts and idle in between (measured: 254 k pairs/s with s_barrier, 260 k with the pair barriers, before the rest):
  //   B1  every image column of this channel is stored - before the row pass reads the image;
  //   B2  every wave has finished reading the image of the previous channel - before the first LDS store of this one; it
  //       sits behind part 1 of A (registers only), so a wave that leaves the row pass early has work before it waits.
  // The Nyquist values of the live channel after next are loaded during this channel and stored to LDS at its end, i.e.
  // before the next B2; wave 0 reads them in its part 1 two channels on: two buffers.
  for (int ci = 0; ci < nlive; ++ci) {
    const int c = chan_at(ci), c1 = chan_at(ci + 1), c2 = chan_at(ci + 2);  // this live channel and the next two
#ifdef SPR_STAMPS
    unsigned long long st[kStampPoints] = {};
#endif
    SPR_STAMP(0);
    // lane coordinates, re-derived from an opaque copy of the lane id every channel: otherwise the ~20 lane addresses
    // and constants below are hoisted out of the channel loop and sit in registers through every phase (spr::opaque)
    const int tid = opaque(tid0);
    const int lane = tid & 63;
    const int g4 = lane >> 4, tc = lane & 15;  // column role: group of the wave, lane of the group

    // part 1, registers only: operands -> product spectrum (two operand registers become one) ...
    auto product = [&](cf (&z)[12], int rc) {
#pragma unroll
      for (int mm = 0; mm < H2; ++mm) {
        const float4 a = nxt[mm], b = nxt[H2 + mm];
        z[2 * mm] = cmul(cmake(a.x, a.y), cmake(b.x, b.y));
        z[2 * mm + 1] = cmul(cmake(a.z, a.w), cmake(b.z, b.w));
      }
      if (rc == 0 && tid < 16) {  // image column 0: pack column nw/2 into its imaginary part
        const cf* nq = nyq + (ci & 1) * NT;
#pragma unroll
        for (int m = 0; m < 12; ++m) {
          const int k1 = tc + 16 * m;
          z[m] = pk_add_i(z[m], cmul(nq[k1], nq[C::NH + k1]));
        }
      }
    };
    // ... -> 12-point stage -> twiddles
    auto stage12 = [&](cf (&z)[12]) {
      pfa12<+1>(z);
#pragma unroll
      for (int p = 1; p < 12; ++p) z[p] = cmul(z[p], twt_h[p * 16 + tc]);
    };
    // exchange: row p of the wave's image holds U[p] of all 64 lanes.  Rows 0..5 in the wave's buffer, rows 6..11 in the
    // image columns of this unit (written at its end, dead until then); `skew` makes row 6 continue the bank sequence of
    // rows 0..5.  The reads are ISSUED here and consumed by part 2: whatever the caller puts in between hides them.
    auto exchange = [&](const cf (&z)[12], cf (&y)[16], int rc) {
      const int own0 = (rc * 24 + 4 * wv) * RS;
      const int xb_el = (S::kXbOff - S::kImgOff) / 8 + wv * S::kXbWave;  // the wave's buffer, in elements from R
      const int skew = ((xb_el + 6 - own0) % 32 + 32) % 32;
      cf* own = R + own0 + skew;
      wave_sync();  // every lane has read the previous unit's rows
      cf* w0 = xb + lane;
      cf* w1 = own + lane;
      if (SPR_ABL != 3) {
#pragma unroll
        for (int p = 0; p < 6; ++p) w0[p * XR] = z[p];
#pragma unroll
        for (int p = 6; p < 12; ++p) w1[(p - 6) * XR] = z[p];
      }
      wave_sync();
      // lane tc < 12 owns sub-transform p = tc; the four idle lanes of a group re-read row 0 (finite values)
      const cf* rd = (tc < 6 ? xb + tc * XR : (tc < 12 ? own + (tc - 6) * XR : xb)) + 16 * g4;
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) y[tt] = rd[tt];
    };
    // part 2: 16-point stage of the owned sub-transform, rows tc + 12 s of image slot rc * 24 + 4 wv + g4
    // (s = 10 reaches past the rows anyone reads: harmless)
    auto part2 = [&](cf (&y)[16], int rc) {
      wave_sync();  // (every lane has its row before the image stores below overwrite rows 6..11 of the exchange)
      if (tc < 12) {
        if (SPR_ABL != 4) Dft<16, +1>::run(y);
        cf* col = R + (rc * 24 + 4 * wv + g4) * RS + tc;
        if (SPR_ABL != 3) {
#pragma unroll
          for (int s = 0; s < 11; ++s) col[12 * s] = y[s];
        }
      }
    };

    // =================================== column pass ===================================
    {
      cf za[12], ya[16], zb[12], yb[16];
      issue_nyq(c2);
      product(za, 0);
      issue_unit(c, 1);  // unit B's operands fly during the 12-point stage of A and the wait
      stage12(za);
#pragma unroll
      for (int p = 0; p < 12; ++p) pin(za[p]);  // (or the compiler sinks the arithmetic below the wait)
      SPR_STAMP(1);
      bar_target += S::WAVES;
      if (SPR_ABL != 2) group_barrier(bar, bar_target);  // B2
      SPR_STAMP(2);
      product(zb, 1);  // before A's exchange: its 16 reads then overlap 24 live registers, not 48
      sched_fence();
      exchange(za, ya, 0);
      sched_fence();
      stage12(zb);  // hides exchange A
      sched_fence();
      issue_unit(c1, 0);  // consumed after the row pass
      SPR_STAMP(3);
      part2(ya, 0);
      SPR_STAMP(4);
      exchange(zb, yb, 1);
      SPR_STAMP(5);
      part2(yb, 1);
      SPR_STAMP(6);
    }
    bar_target += S::WAVES;
    if (SPR_ABL != 2) group_barrier(bar, bar_target);  // B1
    SPR_STAMP(7);
    // =================================== row pass ===================================
    {
      const int grp = lane / 3;
      const int t3 = lane - 3 * grp;
      // (lane 63 has no row group and shadows group 20.  Rows >= ih: every row below 132 is written by the column
      // pass, so surplus groups transform real - finite - rows of the circular correlation; their 1/sigma is 0.)
      const int grp_r = grp < C::kRowGroups ? grp : C::kRowGroups - 1;
      const int row = wv * C::kRowGroups + grp_r;  // a wave reads ONLY its own 21 rows (it reuses them below)
      const int tm = t3 == 0 ? 0 : 3 - t3;         // lane of the mirrored column 48 - k
      const cf* dbase = R + t3 * RS + row;
      const cf* mbase = R + tm * RS + row;
      const cf* cc = ctab2 + t3;
      cf z[16];
#pragma unroll
      for (int a2 = 0; a2 < 8; ++a2) {
        const cf ck = cc[a2 * 3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int a = 2 * a2 + h;
          const int ra = (3 * a) % 16, rm = (3 * ((16 - a) % 16)) % 16;
          const cf av = dbase[3 * ra * RS];
          const cf bv = mbase[3 * rm * RS];
          cf v = h == 0 ? pk_conj_iaxpy<0>(av, ck, bv) : pk_conj_iaxpy<1>(av, ck, bv);
          if (a == 0) {  // k = 0 (lane 0 of the group): image slot 0 holds (Y[0], Y[nw/2]), both real
            const cf v0 = cmake(av.x + av.y, av.x - av.y);
            v = t3 == 0 ? v0 : v;
          }
          z[a] = v;
        }
      }
      SPR_STAMP(8);
      if (SPR_ABL != 4) Dft<16, +1>::run(z);
      sched_fence();
      // the 1/sigma slice is needed at the very end: requested only now, it does not sit in registers during the
      // 16-point stage (the exchange below is what hides its latency)
      float4 iv[6];
      const unsigned so_inv = inv_base + static_cast<unsigned>(c) * static_cast<unsigned>(g.inv_per_chan) * 4u;
#pragma unroll
      for (int pp = 0; pp < 6; ++pp)
        iv[pp] = SPR_ABL == 1 ? make_float4(1.f, 1.f, 1.f, 1.f) : buf_ld16(g_rs, voff16, so_inv + pp * NT * 16);
      // 3-point lane stage.  The exchange image (16 rows of 64 lanes) lies in the image rows this wave has just
      // consumed - rows 21 wv .. 21 wv + 20 of all 48 slots, which no other wave reads: exchange row r takes slots
      // 3r .. 3r+2 (21 lanes each, a row group never straddles two) - so all 16 values leave in one burst and
      // the 18 reads come back in one round trip.
      const float kH = 0.86602540378443860f;
      const cf kl = t3 == 1 ? cmake(-0.5f, kH) : cmake(1.0f, 0.0f);  // output a = y0 + kappa s + lambda (i d)
      const cf hb = t3 == 0 ? cmake(kH, 0.0f) : cmake(-kH, 0.0f);    // output b = m + hb (i d)
      cf* wx = R + S::xpos(lane < 63 ? lane : 62) + wv * C::kRowGroups;  // (lane 63 writes nothing)
      const cf* rxt[3];
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) rxt[tt] = R + S::xpos(3 * grp_r + tt) + wv * C::kRowGroups;
      wave_sync();  // every lane of the wave has read its image rows
      if (lane < 63 && SPR_ABL != 3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) wx[3 * r * RS] = z[r];
      }
      SPR_STAMP(9);
      wave_sync();
      cf yv[6][3];
#pragma unroll
      for (int pp = 0; pp < 6; ++pp)
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)  // row n1 = t3 + 3 pp; rows 16, 17 do not exist: those lanes re-read row 15
          yv[pp][tt] = pp < 5 ? rxt[tt][(9 * pp + 3 * t3) * RS] : rxt[tt][45 * RS];
#pragma unroll
      for (int pp = 0; pp < 6; ++pp) {
        const cf y0 = yv[pp][0], y1 = yv[pp][1], y2 = yv[pp][2];
        const cf s = y1 + y2, d = y1 - y2;
        const cf m = y0 - 0.5f * s;
        const cf ob = pk_iaxpy<0>(d, hb, m);
        const cf oa = pk_iaxpy<1>(d, kl, pk_axpy<0>(s, kl, y0));
        const cf ia = cmake(iv[pp].x, iv[pp].y), ib = cmake(iv[pp].z, iv[pp].w);
        acc[pp][0] = oa * ia + acc[pp][0];
        acc[pp][1] = ob * ib + acc[pp][1];
        if (maps_out && half == 0) {  // debug / parity output of the per-channel maps (spr_ncc_maps): one uniform branch
          const int n1 = t3 + 3 * pp;
          const int real_row = wv * C::kRowGroups + grp;
          if (n1 < 16 && grp < C::kRowGroups && real_row < g.ih) {
            const int ma = t3 == 2 ? n1 + 16 : n1, mb = t3 == 2 ? n1 : n1 + 16;
            float* dst = maps_out + (static_cast<size_t>(c) * g.ih + real_row) * g.iw;
            if (2 * ma < g.iw) dst[2 * ma] = oa.x * ia.x;
            if (2 * ma + 1 < g.iw) dst[2 * ma + 1] = oa.y * ia.y;
            if (2 * mb < g.iw) dst[2 * mb] = ob.x * ib.x;
            if (2 * mb + 1 < g.iw) dst[2 * mb + 1] = ob.y * ib.y;
          }
        }
      }
      nyq[(ci & 1) * NT + tid] = tid < C::NH ? nyq_g : nyq_q;  // the live channel after next's
    }
    SPR_STAMP(10);
#ifdef SPR_STAMPS
    if (blockIdx.x == 777 && (tid0 & 63) == 0 && ci >= kStampFirst && ci < kStampFirst + kStampChannels) {
      for (int i = 0; i < kStampPoints; ++i)
        g_stamps[((half * 6 + wv) * kStampChannels + (ci - kStampFirst)) * kStampPoints + i] = st[i];
    }
#endif
  }

  // Slots outside the ih x iw map carry 1/sigma = 0 and stay 0; the score is floored at 0 anyway
  // (similarity.py:355), so they cannot change the result.
  float best = 0.0f;
#pragma unroll
  for (int pp = 0; pp < 6; ++pp) {
    best = fmaxf(best, fmaxf(fmaxf(acc[pp][0].x, acc[pp][0].y), fmaxf(acc[pp][1].x, acc[pp][1].y)));
  }
  // maximum over the six waves of this pair
  for (int m = 32; m >= 1; m >>= 1) best = fmaxf(best, shfl_xor(best, m));
  if ((tid0 & 63) == 0) red[wv] = best;
  __syncthreads();
  if (tid0 == 0 && scores && live) {
    float b = red[0];
    for (int w = 1; w < S::WAVES; ++w) b = fmaxf(b, red[w]);
    const float s = b / static_cast<float>(g.channels);
    float* dst = scores + static_cast<size_t>(qi) * ld + col0 + gi_item;
    const float prev = g.accumulate ? *dst : 0.0f;
    *dst = s > prev ? s : prev;
  }
}

using C6 = Cfg<12, 16, 12, 8, 384, 5, 2, 1>;

}  // namespace

#ifdef SPR_STAMPS
extern "C" int spr_debug_read_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

size_t pair6_lds_bytes() { return Six<C6>::kLdsBytes; }
int pair6_max_rows() { return C6::kRows6; }
int pair6_max_cols() { return 64; }  // outputs x[2m], x[2m+1] for m < 32

int launch_pair6(const NccGeom& g, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores, int64_t ld,
                 int64_t col0, int accumulate, float* maps_out, const cf* tw_h, const float* ctab, hipStream_t stream) {
  if (nq == 0 || ng == 0) return SPR_OK;
  if (!ctab) { set_error("pair6_kernel: the plan has no pre-twist table"); return SPR_ERR_ARG; }
  using S = Six<C6>;
  Pair6Args a{};
  a.channels = g.channels; a.nq = static_cast<int>(nq); a.ng = static_cast<int>(ng);
  a.ih = g.ih; a.iw = g.iw; a.inv_per_chan = g.inv_per_chan; a.accumulate = accumulate;
  a.tiles_g = ceil_div(static_cast<int>(ng), kTileG6);
  a.q_flags_off = static_cast<unsigned>(sizeof(cf) * static_cast<size_t>(g.channels) * g.spec_per_chan);
  a.g_flags_off = static_cast<unsigned>(static_cast<size_t>(g.channels) * (sizeof(cf) * g.spec_per_chan + sizeof(float) * g.inv_per_chan));
  if (g.channels > 1024) { set_error("pair6_kernel: at most 1024 channels"); return SPR_ERR_UNSUPPORTED; }
  const size_t lds_bytes = S::kLdsBytes + 2 * sizeof(unsigned short) * static_cast<size_t>(g.channels);
  { const char* v = std::getenv("SPR_P6_PRIO"); a.prio_mode = v && *v ? std::atoi(v) : 2; }
  const int64_t tiles = static_cast<int64_t>(ceil_div(static_cast<int>(nq), kTileQ6)) * a.tiles_g;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pair6_kernel<C6>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLdsLimit);
  // HIP refuses grids of 2^32 work-items and more: launch in slices of pair tiles
  const int64_t max_tiles = pair_tiles_per_launch(kTileQ6 * kTileG6 / 2, 2 * C6::NT);
  for (int64_t t0 = 0; t0 < tiles; t0 += max_tiles) {
    const int64_t n = tiles - t0 < max_tiles ? tiles - t0 : max_tiles;
    a.tile0 = static_cast<int>(t0);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(pair6_kernel<C6>), dim3(static_cast<unsigned>(n * (kTileQ6 * kTileG6 / 2))), dim3(2 * C6::NT),
                       lds_bytes, stream, a, static_cast<const unsigned char*>(pq),
                       prepared_query_item_bytes(g, SPR_NCC_FFT), static_cast<const unsigned char*>(pg),
                       prepared_gallery_item_bytes(g, SPR_NCC_FFT), scores, static_cast<long long>(ld),
                       static_cast<long long>(col0), maps_out, tw_h, ctab);
    const int rc = check_launch("pair6_kernel");
    if (rc != SPR_OK) return rc;
  }
  return SPR_OK;
}

}  // namespace spr
