// Grid configurations of the FFT-method NCC kernels: which lanes hold which spectrum elements, and where a
// prepared item keeps them (shared by the prep kernel, the 4-wave pair kernels of ncc_fft.hip and the 6-wave
// pair kernel of ncc_pair6.hip).
#pragma once
#include "fft_core.h"

namespace spr {

// EH x TGH: column transforms (length nh), EW x TGW: row transforms (length nw); NT = work-items per
// workgroup of the PAIR kernel (the prep kernel always runs kThreads and writes the prepared data in the
// pair kernel's lane order); KWA / RRA: the tuned variant's kept outputs per row sub-transform and row
// rounds (the general variant keeps everything: KW = TGW, RR = all row pairs of the grid).
// SIX = 1: the prepared data feed the six-wave pair kernel (ncc_pair6.hip): its row pass is one real-output
// transform per image row (length NW as a complex transform of NW/2 = 16 x 3 points, prime-factor split, three
// lanes per row), so the intermediate image keeps its columns in the order those lanes read them (slot()),
// the query spectrum carries the transform's pre-twist factor (c2r_beta()) and the 1/sigma map follows that
// kernel's accumulator order (inv6_index()).
template <int EH_, int TGH_, int EW_, int TGW_, int NT_, int KWA_, int RRA_, int SIX_ = 0>
struct Cfg {
  static constexpr int EH = EH_, TGH = TGH_, EW = EW_, TGW = TGW_, NT = NT_, SIX = SIX_;
  using GH = GroupFft<EH, TGH>;
  using GW = GroupFft<EW, TGW>;
  static constexpr int NH = EH * TGH, NW = EW * TGW;
  static constexpr int CPR = NT / TGH;  // columns per column-pass round of the pair kernel
  static constexpr int PPR = NT / TGW;  // row pairs per row-pass round of the pair kernel
  static constexpr int COLS = NW / 2;   // columns of the intermediate image (column nw/2 rides in column 0)
  static constexpr int RC = (COLS + CPR - 1) / CPR;  // column rounds per channel
  static constexpr int KW_A = KWA_, RR_A = RRA_;
  static constexpr int KW_B = TGW, RR_B = ((NH + 1) / 2 + PPR - 1) / PPR;
  static_assert(EH % 2 == 0 && EW % 2 == 0, "register pairs are loaded with 16-byte accesses");
  static constexpr int xbuf_elems(int threads) {
    return GH::block_elems(threads) > GW::block_elems(threads) ? GH::block_elems(threads) : GW::block_elems(threads);
  }
  // prep kernel: a row group's buffer also stages the NW outputs of its transform for the two-row split
  // (group stride == TGW (mod 16), as in GroupFft)
  static constexpr int kRowRaw = GW::kGroupElems > NW ? GW::kGroupElems : NW;
  static constexpr int kRowGroupElems = kRowRaw + ((TGW % 16) - (kRowRaw % 16) + 16) % 16;
  static constexpr int prep_xbuf_elems(int threads) {
    return GH::block_elems(threads) > (threads / TGW) * kRowGroupElems ? GH::block_elems(threads)
                                                                      : (threads / TGW) * kRowGroupElems;
  }
  // Position of spectrum element (column j < COLS, k1) in a channel's prepared data: the pair kernel's
  // lane (group j % CPR, lane-in-group k1 % TGH) loads registers (2mm, 2mm+1), m = k1 / TGH, of column
  // round j / CPR with one 16-byte access.
  static __host__ __device__ constexpr int spec_index(int j, int k1) {
    return (((j / CPR) * (EH / 2) + (k1 / TGH) / 2) * NT + (j % CPR) * TGH + (k1 % TGH)) * 2 + ((k1 / TGH) & 1);
  }
  static constexpr int kNyqOffset = RC * EH * NT;  // column nw/2 follows in natural k1 order
  static constexpr int kSpecPerChan = kNyqOffset + NH;

  // ---- six-wave kernel (SIX): NW/2 = 16 * 3 ---------------------------------------------------------------
  // Row transform of one image row: z[m] = sum_k Z[k] e^{+2 pi i k m / (NW/2)}, k = (3a + 16t) mod 48 held by lane
  // t = 0..2 in register a = 0..15, output m = (33 n1 + 16 n2) mod 48 (Good-Thomas maps: no twiddles between the
  // 16-point register stage and the 3-point lane stage); x[2m] = Re z[m], x[2m+1] = Im z[m].
  // slot(k): position of column k in the intermediate image.  Lane t reads slot 3*((3a) mod 16) + t for its
  // register a, so the three columns k, k+16, k+32 (mod 48) that share a residue mod 16 sit side by side, starting
  // with the one lane 0 reads.
  static constexpr int kRowGroups = 21;               // 3-lane row groups per wave (lane 63 idles)
  static constexpr int kRows6 = (NT / 64) * kRowGroups;  // image rows one workgroup transforms per channel
  static constexpr int kInv6PerChan = 6 * NT * 4;     // floats: six sub-transform sets x {a.re, a.im, b.re, b.im}
  static __host__ __device__ constexpr int slot(int k) {
    if (!SIX) return k;
    const int r = k % 16, a = (11 * r) % 16, q = (3 * a) / 16, sg = k / 16;
    return 3 * r + (sg - q + 3) % 3;
  }
  // 1/sigma slot of output pixel (row n1, column n2) of the search map: lane (row group, n1' mod 3) owns the
  // 3-point sub-transform n1' = m mod 16 (m = n2 / 2) and keeps two of its three outputs, a and b.
  static __host__ __device__ constexpr int inv6_index(int row, int n2) {
    const int m = n2 >> 1, ri = n2 & 1;
    const int n1 = m & 15, tq = n1 % 3, pp = n1 / 3;
    const int b = (33 * n1) % 48;
    const int ma = tq == 1 ? b - 32 : b;  // output a of lanes 0 / 1 / 2: X0 (m = b), X1 (m = b - 32), X0 (m = b)
    const int slot_ab = m == ma ? 0 : 1;
    const int lane = (row / kRowGroups) * 64 + (row % kRowGroups) * 3 + tq;
    return (pp * NT + lane) * 4 + slot_ab * 2 + ri;
  }
};

}  // namespace spr
