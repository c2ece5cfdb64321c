// FFT-method NCC scorer (the fast path).
//
// Correlation theorem:  num = IFFT2( FFT2(I0z) . conj(FFT2(t^)) )  on an nh x nw grid large enough
// that the 'same'-mode lags are free of circular aliasing (nh >= ih + max(th/2, th-1-th/2), same
// for nw).  The reference does exactly this through scipy's fftconvolve for every pair and channel
// (similarity.py:55: 2 forward + 1 inverse FFT per call, 3 calls per channel); here
//
//   prep kernels (once per item and channel)
//     gallery: centre -> float64 window sums -> 1/sigma map; forward 2-D FFT of the zero-padded map
//     query:   centre -> scale by 1/sqrt(sum t0^2) -> forward 2-D FFT, conjugated, with the 'same'
//              centre shift (th/2, tw/2) and the 1/(nh*nw) inverse-FFT factor folded in
//   pair kernel (one workgroup per (query, gallery) pair, loop over channels)
//     spectrum product -> inverse column FFTs (only the rows that cover ih are kept, in LDS)
//     -> inverse row FFTs, two rows per complex transform (rows are real) -> multiply by the
//     1/sigma map -> accumulate the channel sum in registers -> final wave/LDS max-reduction.
//
// Grids are 2^k or 3*2^k per axis (fft_core.h GroupFft): the VGG16 conv3_3 maps of a 512x256 print
// (124x60 after the crop) need >= 186 x 90 and run on 192 x 96 — 56 % of the 256 x 128 power-of-two
// grid in bytes, flops and LDS, which also lets two 256-lane workgroups share a CU.
//
// All FFTs are LDS/register resident: the pair kernel's only HBM/L2 traffic is the two half-spectra
// and the 1/sigma map of the current channel, laid out in exactly the lane/register order the kernel
// consumes (fully coalesced 16-byte loads).
//
// Half-spectrum bookkeeping (rows are real => X[k1][nw-k2] = conj(X[-k1][k2])):
//   columns k2 = 0 .. nw/2 are stored; the pair kernel runs nw/2 column transforms, the first of
//   which carries columns 0 and nw/2 packed as  P0 + i*Pn  (both give real column results).
#include <cstdlib>

#include "fft_core.h"
#include "ncc_fft_cfg.h"
#include "ncc_prep_common.h"

#ifndef SPR_BIG_ABL
#define SPR_BIG_ABL 0  // timing ablations of the workspace instance (wrong results): 1 no image stores, 2 image reads from one
#endif                 // cached line set, 3 operands of channel 0 every channel
namespace spr {
namespace {

// The intermediate image is stored TRANSPOSED: RT[j][n1], column j of the half spectrum, row n1, with a
// row stride == 8 (mod 32) complex values.  Column transforms then write consecutive values per group with
// compile-time offsets, and a row-pair lane reads (row 2pr, row 2pr+1) of one column as a single 16-byte
// access; with that stride the four 8-lane groups of every ds_read_b128 lane group cover all 64 banks once.
inline int rt_stride(int r_rows) { return r_rows + ((8 - r_rows % 32) + 32) % 32; }

// ============================================================================================
// Forward (prep) kernel.  grid = (channels, n_items), kThreads lanes
// ============================================================================================
// BIG: maps too large for LDS.  The centred map, the two float64 tables and the row-pass output then live in a
// per-workgroup slot of a global workspace (x0_off / f_off / sat2_off are offsets into the slot); only the
// exchange buffers stay in LDS.  Same code, same arithmetic - the slot is L2-resident scratch.
// PT = work-items of this kernel's workgroups: LDS allows one workgroup per CU on the larger grids, so those run
// 8 waves (two per SIMD) to overlap the LDS round trips of the transform rounds; small grids keep 4.
template <class C, bool BIG, int PT>
__global__ void __launch_bounds__(PT)
prep_fft_kernel(NccGeom g, int is_query, const void* __restrict__ maps, unsigned char* __restrict__ prepared,
                size_t item_bytes, const cf* __restrict__ tw_h, const cf* __restrict__ tw_w, unsigned x0_off,
                unsigned f_off, unsigned xbuf_off, unsigned zbuf_off, unsigned sat2_off, int f_stride,
                unsigned char* __restrict__ ws, size_t slot_bytes) {
  using GH = typename C::GH;
  using GW = typename C::GW;
  unsigned char* lds = dyn_lds();
  unsigned char* big = lds;
  if constexpr (BIG) big = ws + (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * slot_bytes;
  double* red = reinterpret_cast<double*>(lds);
  float* x0 = reinterpret_cast<float*>(big + x0_off);
  double* sat = reinterpret_cast<double*>(big + f_off);  // dead before F is written
  cf* F = reinterpret_cast<cf*>(big + f_off);
  cf* xbuf = reinterpret_cast<cf*>(lds + xbuf_off);
  (void)zbuf_off;
  const int tid = static_cast<int>(threadIdx.x);
  const int c = static_cast<int>(blockIdx.x);
  const size_t item = blockIdx.y;
  const int h = is_query ? g.th : g.ih, w = is_query ? g.tw : g.iw;
  const int raw_h = is_query ? g.q_h : g.g_h, raw_w = is_query ? g.q_w : g.g_w;

  unsigned char* item_base = prepared + item * item_bytes;
  cf* spec = reinterpret_cast<cf*>(item_base) + static_cast<size_t>(c) * C::kSpecPerChan;

  // the six-wave layout with both tables in LDS writes its 1/sigma slots in slot order, zeros included (below)
  const bool inv_by_slot = C::SIX && sat2_off != 0 && sat_blocked_fits(h, w);
  if (!is_query && !inv_by_slot) {
    // 1/sigma slots that no pixel maps to (rows >= ih, columns >= iw, surplus lanes) must read as 0: clear the
    // channel's slot first; the workgroup barriers below order these stores before the values written later
    float4* inv4 = reinterpret_cast<float4*>(item_base + sizeof(cf) * static_cast<size_t>(g.channels) * C::kSpecPerChan) +
                   static_cast<size_t>(c) * (g.inv_per_chan / 4);
    for (int i = tid; i < g.inv_per_chan / 4; i += PT) inv4[i] = float4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  SPR_PSTAMP(0);
  load_centred(maps, (item * g.channels + c) * static_cast<size_t>(raw_h) * raw_w, raw_w, g.crop, h, w, g.dtype, x0,
               red);
  SPR_PSTAMP(1);
  float scale = 1.0f;
  {
    // dead flag of this (item, channel): a constant channel - all zero after ReLU, typically - has a zero centred map,
    // hence a zero spectrum, and contributes exactly 0 to every pair (similarity.py:68-70); the six-wave pair kernel
    // skips channels flagged on either side.  The flags follow the spectra (and the 1/sigma maps) of the item.
    const float rs = template_scale(x0, h * w, red);  // 1 / sqrt(sum x0^2), 0 for an all-zero centred map
    unsigned char* flags = item_base + static_cast<size_t>(g.channels) *
                           (sizeof(cf) * C::kSpecPerChan + (is_query ? 0 : sizeof(float) * static_cast<size_t>(g.inv_per_chan)));
    if (tid == 0) flags[c] = rs == 0.0f ? 1 : 0;
    if (is_query) scale = rs * (1.0f / (static_cast<float>(C::NH) * static_cast<float>(C::NW)));
  }
  SPR_PSTAMP(2);
  if (!is_query) {
    // 1/sigma map in the pair kernel's register order; slots no pixel maps to stay 0.
    float* inv = reinterpret_cast<float*>(item_base + sizeof(cf) * static_cast<size_t>(g.channels) * C::kSpecPerChan) +
                 static_cast<size_t>(c) * g.inv_per_chan;
    const int nv = g.nv;
    auto store = [&](int n1, int n2, float v) {
      if constexpr (C::SIX) {  // the six-wave pair kernel's accumulator order (ncc_fft_cfg.h)
        inv[C::inv6_index(n1, n2)] = v;
        return;
      }
      const int pr = n1 >> 1, ab = n1 & 1;
      const int rr = pr / C::PPR, giw = pr - rr * C::PPR;
      const int p = n2 % C::EW, s = n2 / C::EW;       // output n2 = p + EW*s of the row transform
      const int pp = p / C::TGW, t = p - pp * C::TGW;  // owned by lane t of the group, sub-transform pp
      const int e2 = (pp * g.keep_w + s) * 2 + ab;
      const int lane = giw * C::TGW + t;
      inv[((rr * (nv / 4) + (e2 >> 2)) * C::NT + lane) * 4 + (e2 & 3)] = v;
    };
    if (inv_by_slot) {
      // One float4 per (sub-transform set, pair-kernel lane) = the 1/sigma values of the four pixels that lane weights:
      // every slot is written exactly once, in address order (16 bytes per work-item, coalesced), zeros where no pixel
      // maps to it.  Inverse of Cfg::inv6_index.
      double* sat2 = reinterpret_cast<double*>(big + sat2_off);
      build_sat_pair_blocked(x0, h, w, sat, sat2);
      SPR_PSTAMP(8);
      const double inv_n = 1.0 / (static_cast<double>(g.th) * static_cast<double>(g.tw));
      const int stride = w + 1;
      float4* inv4 = reinterpret_cast<float4*>(inv);
      for (int idx4 = tid; idx4 < C::kInv6PerChan / 4; idx4 += PT) {
        const int pp = idx4 / C::NT, lane6 = idx4 - pp * C::NT;
        const int l64 = lane6 & 63, rg = l64 / 3, tq = l64 - 3 * rg;
        const int row = (lane6 >> 6) * C::kRowGroups + rg, n1 = 3 * pp + tq;
        float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (l64 < 63 && n1 < 16 && row < h) {
          const int b = (33 * n1) % 48;
          const int ma = tq == 1 ? b - 32 : b, mb = ma == n1 ? n1 + 16 : n1;
          int y0 = row - g.th / 2, y1 = y0 + g.th;
          y0 = y0 < 0 ? 0 : (y0 > h ? h : y0);
          y1 = y1 > h ? h : (y1 < 0 ? 0 : y1);
          const double* t1a = sat + static_cast<size_t>(y1) * stride;
          const double* t1b = sat + static_cast<size_t>(y0) * stride;
          const double* t2a = sat2 + static_cast<size_t>(y1) * stride;
          const double* t2b = sat2 + static_cast<size_t>(y0) * stride;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int x = 2 * (e < 2 ? ma : mb) + (e & 1);
            if (x >= w) continue;
            int xa = x - g.tw / 2, xb = xa + g.tw;
            xa = xa < 0 ? 0 : (xa > w ? w : xa);
            xb = xb > w ? w : (xb < 0 ? 0 : xb);
            const double s1 = t1a[xb] - t1b[xb] - t1a[xa] + t1b[xa];
            const double s2 = t2a[xb] - t2b[xb] - t2a[xa] + t2b[xa];
            o[e] = inv_sigma_from_sums(s1, s2, inv_n);
          }
        }
        inv4[idx4] = float4{o[0], o[1], o[2], o[3]};
      }
      __syncthreads();
    } else if (sat2_off != 0) {
      inv_sigma_map_fused(x0, h, w, g.th, g.tw, sat, reinterpret_cast<double*>(big + sat2_off), store);
    } else {
      inv_sigma_map(x0, h, w, g.th, g.tw, sat, [&](int i, float v) { store(i / w, i % w, v); });
    }
  }

  SPR_PSTAMP(3);
  // ---- row pass: two real rows per complex transform of length NW -------------------------------
  {
    const int giw = tid / C::TGW, t = tid - giw * C::TGW;
    RegTwiddles<C::EW> twr;
    load_twiddles<C::EW, C::TGW, -1>(twr, tw_w, t);
    cf* gbuf = xbuf + giw * C::kRowGroupElems;
    cf* zb = gbuf;  // the group's exchange buffer doubles as the staging row of the two-row split (>= NW elements)
    const int pairs = (h + 1) / 2;
    constexpr int kPairsPerRound = PT / C::TGW;
    const int rounds = ceil_div(pairs, kPairsPerRound);
    for (int rr = 0; rr < rounds; ++rr) {
      const int pr = rr * kPairsPerRound + giw;
      const int ra = 2 * pr, rb = ra + 1;
      cf x[C::EW], y[GW::SPL][C::TGW];
#pragma unroll
      for (int m = 0; m < C::EW; ++m) {
        const int n2 = GW::in_index(t, m);
        const bool in = n2 < w;
        x[m].x = (in && ra < h) ? x0[ra * w + n2] * scale : 0.0f;
        x[m].y = (in && rb < h) ? x0[rb * w + n2] * scale : 0.0f;
      }
      group_fft<C::EW, C::TGW, -1>(x, y, t, twr, gbuf);
      wave_sync();  // every lane of the group has read its exchange values before the buffer is reused
      // publish Z[k] for the group, then split the two real rows:  Xa = (Z[k] + conj Z[-k])/2,
      // Xb = (Z[k] - conj Z[-k])/(2i)
#pragma unroll
      for (int pp = 0; pp < GW::SPL; ++pp)
#pragma unroll
        for (int s = 0; s < C::TGW; ++s)
          if (GW::out_valid(t, pp)) zb[GW::out_index(t, pp, s)] = y[pp][s];
      wave_sync();
      if (pr < pairs) {
        for (int k = t; k <= C::NW / 2; k += C::TGW) {
          const cf zk = zb[k];
          const cf zm = zb[k == 0 ? 0 : C::NW - k];
          F[ra * f_stride + k] = cmake(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
          F[rb * f_stride + k] = cmake(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
        }
      }
      wave_sync();
    }
  }
  __syncthreads();
  SPR_PSTAMP(4);

  // ---- column pass: nw/2 + 1 columns of length NH ----------------------------------------------
  {
    const int gi = tid / C::TGH, t = tid - gi * C::TGH;
    RegTwiddles<C::EH> twr;
    load_twiddles<C::EH, C::TGH, -1>(twr, tw_h, t);
    cf* gbuf = xbuf + gi * GH::kGroupElems;
    const int rows_f = 2 * ((h + 1) / 2);
    const int cy = g.th / 2, cx = g.tw / 2;
    constexpr int kColsPerRound = PT / C::TGH;
    constexpr int kRounds = (C::COLS + 1 + kColsPerRound - 1) / kColsPerRound;  // COLS columns + column nw/2
    for (int rc = 0; rc < kRounds; ++rc) {
      const int j = rc * kColsPerRound + gi;
      const bool nyq = j == C::COLS;
      const bool active = j <= C::COLS;
      cf x[C::EH], y[GH::SPL][C::TGH];
#pragma unroll
      for (int m = 0; m < C::EH; ++m) {
        const int n1 = GH::in_index(t, m);
        x[m] = (active && n1 < rows_f) ? F[n1 * f_stride + j] : cmake(0.0f, 0.0f);
      }
      group_fft<C::EH, C::TGH, -1>(x, y, t, twr, gbuf);
      if (active) {
        cf wx = cmake(1.0f, 0.0f);
        if (is_query) {
          wx = tw_w[(cx * j) % C::NW];
          if (C::SIX && j > 0 && !nyq) {  // pre-twist factor of the six-wave row pass: 1 - i w^j, w = e^(+2 pi i / nw)
            const cf wj = tw_w[j];        // = conj(w^j) = (cos, -sin)
            wx = cmul(wx, cmake(1.0f - wj.y, -wj.x));
          }
        }
#pragma unroll
        for (int pp = 0; pp < GH::SPL; ++pp) {
          if (!GH::out_valid(t, pp)) continue;
#pragma unroll
          for (int s = 0; s < C::TGH; ++s) {
            const int k1 = GH::out_index(t, pp, s);
            cf v = y[pp][s];
            if (is_query) {
              const cf wy = tw_h[(cy * k1) % C::NH];
              v = cmul(cmul(cconj(v), wy), wx);  // conj(A) * w^(cy k1) * w^(cx k2): centre shift folded in
            }
            spec[nyq ? C::kNyqOffset + k1 : C::spec_index(C::slot(j), k1)] = v;
          }
        }
      }
    }
  }
  SPR_PSTAMP(5);
}

// ============================================================================================
// Pair kernel.  One workgroup per (query, gallery) pair, looping over the channels.
//
// Memory-latency structure (the kernel streams ~180-300 KB of spectra + 1/sigma per pair and channel,
// far more than it can keep in LDS, so everything is register-prefetched ahead of use):
//   * the column pass is a flat sequence of "units" (channel c, column round rc); the two half-spectra
//     of a later unit are loaded (16-byte loads, lane-ordered layout) before the current one is computed:
//     with PF == RC there is one buffer per round, refilled right after it is consumed (a full channel of
//     lead), with PF == 1 a single buffer holds the next unit (fewer registers);
//   * the 1/sigma slice is requested at the start of a channel's row pass and used at its end, the
//     Nyquist column (k2 = nw/2: staged through a small LDS buffer for the lanes of column group 0) of
//     channel c+1 is loaded during channel c;
//   * no wait is placed by hand: loads are issued early and the compiler's counted s_waitcnt sits at
//     the first use.
// Pair -> workgroup mapping: 1-D grid in tiles of 16 queries x 16 gallery items.  Workgroups are dealt
// round-robin over the 8 XCDs, so workgroup w of a tile (w % 8 = XCD group) takes a 16-query x 2-gallery
// sub-tile: a gallery item's spectrum + 1/sigma slice (116 KB per channel, the larger side) then has 16
// readers on the same L2 and a query spectrum (75 KB) 2 (4 with the co-resident next tile).  Measured at
// Q=100 x G=1500: 228 k pairs/s against 173 k with 4-query x 8-gallery sub-tiles — workgroups drift apart
// by a channel or more, and the more readers a line has the likelier one of them is still close in time.
// ============================================================================================
constexpr int kTileQ = 16, kTileG = 16;

// What the pair kernel needs of the plan, kept small: kernel arguments live in SGPRs, and the packed-math
// twiddle constants want those too.
struct PairArgs {
  int channels, nq, ng;
  int ih, iw;        // cropped search-map size (debug map output only)
  int r_rows;        // rows of the intermediate image that matter (pairs = r_rows / 2)
  int r_stride;      // row stride of the transposed image
  int rounds_r;      // row rounds actually needed (<= RR)
  int inv_per_chan;  // floats of 1/sigma per channel
  int accumulate;
  int tile0;         // tile mode: first 16 x 16 pair tile of this launch (launches are sliced: HIP's grid limit)
  // team mode (team_size > 0): persistent grid of 8 teams (one per XCD) x team_size resident workgroups
  int team_size;     // workgroups per team = pairs per epoch
  int strip_q;       // queries per strip (the last strip may hold fewer)
  int strips;        // number of query strips
  int epochs_full;   // epochs of a full strip = ceil(strip_q * ng / team_size)
  int epochs_total;  // over all strips
  int sync_polls;    // bound of the soft team barrier
  int sync_every;    // channels between mid-pair team barriers (0: only at the start of a pair)
};

// RK = rows of a column transform's output kept in the LDS image (compile-time for the tuned variant: the
// stage-2 outputs beyond it are dead code and the stores need no per-row test; 0 = runtime r_rows)
// BIG: the intermediate image does not fit LDS; it lives in this workgroup's slot of a global workspace (the
// launch is then always the persistent team grid, one slot per resident workgroup).
// TEAM: the persistent-grid schedule is its own instantiation, so the default one-pair-per-workgroup kernel
// carries none of its scalar state (measured: 1.8 % when both lived in one kernel).  BIG implies TEAM.
template <class C, int RR, int KW, int PF, int RK, bool BIG, bool TEAM>
__global__ void __launch_bounds__(C::NT, BIG ? 1 : ((C::NT == 64 && RK > 0) ? 4 : 2))
pair_fft_kernel(PairArgs g, const unsigned char* __restrict__ pq, size_t q_item_bytes,
                const unsigned char* __restrict__ pg, size_t g_item_bytes, float* __restrict__ scores,
                long long ld, long long col0, float* __restrict__ maps_out,
                const cf* __restrict__ tw_h, const cf* __restrict__ tw_w, unsigned r_off, unsigned xbuf_off,
                unsigned nyq_off_lds, unsigned* __restrict__ team_sync, unsigned char* __restrict__ ws,
                size_t slot_bytes) {
  using GH = typename C::GH;
  using GW = typename C::GW;
  const int nq = g.nq, ng = g.ng;

  unsigned char* lds = dyn_lds();
  float* red = reinterpret_cast<float*>(lds);
  cf* R = reinterpret_cast<cf*>(lds + r_off);
  if constexpr (BIG) R = reinterpret_cast<cf*>(ws + static_cast<size_t>(blockIdx.x) * slot_bytes);
  cf* xbuf = reinterpret_cast<cf*>(lds + xbuf_off);
  cf* nyq = reinterpret_cast<cf*>(lds + nyq_off_lds);  // [0, NH): gallery column nw/2, [NH, 2NH): query's
  const int tid = static_cast<int>(threadIdx.x);
  // LEAN (the workspace instance, 24-point column units on 512 lanes): nothing but the accumulators lives across a
  // transform - twiddles from the LDS table at use, one row round's 1/sigma at a time, operands requested where they are
  // multiplied.  With the prefetch state of the LDS instances (2 x 96 + 48 registers) the kernel spilled 345 registers.
  constexpr bool LEAN = BIG;
  constexpr int NVR = GW::SPL * KW * 2;          // accumulators per lane and row round ...
  constexpr int NV = (NVR + 3) / 4 * 4;          // ... padded to whole 16-byte loads of 1/sigma
  constexpr int RC = C::RC;                      // column rounds per channel
  constexpr int H2 = C::EH / 2;                  // 16-byte loads per operand and unit
  constexpr int NYQ = (2 * C::NH + C::NT - 1) / C::NT;  // Nyquist values prefetched per lane
  static_assert(PF == 1 || PF == RC, "prefetch depth: one unit or one buffer per round");

  const cf* qspec = nullptr;   // set per pair below
  const cf* gspec = nullptr;
  const float* ginv = nullptr;
  const int last_c = g.channels - 1;
  const int tid0 = tid;

  // inverse twiddle tables w^(+t*p), [p][t], in LDS (read at use: keeps 2*(EH+EW) VGPRs free)
  cf* twt_h = nyq + 2 * C::NH;
  cf* twt_w = twt_h + C::NH;
  for (int k = tid; k < C::NH; k += C::NT) twt_h[k] = cconj(tw_h[(k / C::TGH) * (k % C::TGH)]);
  for (int k = tid; k < C::NW; k += C::NT) twt_w[k] = cconj(tw_w[(k / C::TGW) * (k % C::TGW)]);

  float acc[RR][NV];
  const int pairs = g.r_rows / 2;
  const int rs = g.r_stride;                    // row stride of the transposed image RT[j][n1]
  // column outputs p + EH*s with s < s_full are rows < (RK or, for RK == 0, the runtime r_rows) for every p;
  // at s == s_full only p < p_part.  Compile-time for the tuned variant.
  const int rk = RK > 0 ? RK : g.r_rows;
  const int s_full = rk / C::EH;
  const int p_part = rk - s_full * C::EH;

  // ---- prefetch state --------------------------------------------------------------------------
  float4 nxt[PF][2 * H2];  // H2 x (2 complex of G), H2 x (2 complex of Q) per buffer (LEAN: unused)
  float4 inv_nxt[LEAN ? 1 : RR][NV / 4];
  cf nyq_nxt[NYQ];
  auto issue_unit = [&](int c, int rc, float4 (&buf)[2 * H2]) {
    c = c > last_c ? last_c : c;  // the one-past-the-end prefetch re-reads the last channel (never used)
    const float4* gs4 = reinterpret_cast<const float4*>(gspec + static_cast<size_t>(c) * C::kSpecPerChan);
    const float4* qs4 = reinterpret_cast<const float4*>(qspec + static_cast<size_t>(c) * C::kSpecPerChan);
#pragma unroll
    for (int mm = 0; mm < H2; ++mm) {
      const size_t idx = (static_cast<size_t>(rc) * H2 + mm) * C::NT + tid;
      buf[mm] = gs4[idx];
      buf[H2 + mm] = qs4[idx];
    }
  };
  auto issue_inv = [&](int c) {
    if constexpr (LEAN) return;
    const float4* inv4 = reinterpret_cast<const float4*>(ginv + static_cast<size_t>(c) * g.inv_per_chan);
#pragma unroll
    for (int rr = 0; rr < (LEAN ? 1 : RR); ++rr)
#pragma unroll
      for (int i = 0; i < NV / 4; ++i)
        inv_nxt[rr][i] = rr < g.rounds_r ? inv4[(rr * (NV / 4) + i) * C::NT + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto issue_inv_round = [&](int c, int rr) {  // LEAN: one row round's weights, requested as the round starts
    const float4* inv4 = reinterpret_cast<const float4*>(ginv + static_cast<size_t>(c) * g.inv_per_chan);
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) inv_nxt[0][i] = inv4[(rr * (NV / 4) + i) * C::NT + tid];
  };
  auto issue_nyq = [&](int c) {
    c = c > last_c ? last_c : c;
    const cf* gs = gspec + static_cast<size_t>(c) * C::kSpecPerChan + C::kNyqOffset;
    const cf* qs = qspec + static_cast<size_t>(c) * C::kSpecPerChan + C::kNyqOffset;
#pragma unroll
    for (int i = 0; i < NYQ; ++i) {
      const int k = tid + i * C::NT;
      nyq_nxt[i] = k < C::NH ? gs[k] : (k < 2 * C::NH ? qs[k - C::NH] : cmake(0.f, 0.f));
    }
  };
  auto store_nyq = [&]() {
#pragma unroll
    for (int i = 0; i < NYQ; ++i) {
      const int k = tid + i * C::NT;
      if (k < 2 * C::NH) nyq[k] = nyq_nxt[i];
    }
  };

  // ---- which pairs ------------------------------------------------------------------------------
  // tile mode: this workgroup's one pair from the 16 x 16 tile mapping above the kernel.
  // team mode: workgroup = member of team (blockIdx % 8 = XCD); the team walks its share of the epochs, an
  // epoch being team_size consecutive pairs of a query strip in gallery-major order (so ~strip_q queries x
  // team_size/strip_q gallery items: every spectrum line has many readers on this L2), and the members
  // start each pair together (soft barrier) so that those readers are close in time as well.
  static_assert(TEAM || !BIG, "the workspace mode runs on the persistent grid");
  const int team = static_cast<int>(blockIdx.x) & 7, member = static_cast<int>(blockIdx.x) >> 3;
  unsigned* sync_ctr = team_sync + team * 32;
  unsigned sync_target = 0;
  int epoch = 0, epoch_end = 1, syncs_per_pair = 1;
  if constexpr (TEAM) {
    syncs_per_pair = g.sync_every > 0 ? ceil_div(g.channels, g.sync_every) : 1;
    epoch = static_cast<int>(static_cast<long long>(g.epochs_total) * team / 8);
    epoch_end = static_cast<int>(static_cast<long long>(g.epochs_total) * (team + 1) / 8);
  }
  for (; epoch < epoch_end; ++epoch) {
  int qi, gi_item;
  if constexpr (TEAM) {
    int strip = epoch / g.epochs_full;
    strip = strip < g.strips ? strip : g.strips - 1;
    const int q0 = strip * g.strip_q;
    const int qn = nq - q0 < g.strip_q ? nq - q0 : g.strip_q;
    const long long p = static_cast<long long>(epoch - strip * g.epochs_full) * g.team_size + member;
    gi_item = static_cast<int>(p / qn);
    qi = q0 + static_cast<int>(p - static_cast<long long>(gi_item) * qn);
    if (gi_item >= ng) {  // ragged last epoch of a strip: keep the team's count whole and move on
      if (tid == 0) team_arrive(sync_ctr, static_cast<unsigned>(syncs_per_pair));
      sync_target += static_cast<unsigned>(syncs_per_pair) * g.team_size;
      continue;
    }
    sync_target += g.team_size;
    if (tid == 0) {
      team_arrive(sync_ctr, 1u);
      team_wait(sync_ctr, sync_target, g.sync_polls);
    }
    __syncthreads();
  } else {
    const int tiles_g = ceil_div(ng, kTileG);
    const int tile = g.tile0 + static_cast<int>(blockIdx.x) / (kTileQ * kTileG);
    const int within = static_cast<int>(blockIdx.x) % (kTileQ * kTileG);
    const int tq = tile / tiles_g, tg = tile - tq * tiles_g;
    const int xcd = within & 7, slot = within >> 3;  // 8 XCD groups x 32 slots
    qi = tq * kTileQ + (slot >> 1);
    gi_item = tg * kTileG + 2 * xcd + (slot & 1);
    if (qi >= nq || gi_item >= ng) return;  // uniform per workgroup
  }
  {
    const unsigned char* g_item = pg + static_cast<size_t>(gi_item) * g_item_bytes;
    qspec = reinterpret_cast<const cf*>(pq + static_cast<size_t>(qi) * q_item_bytes);
    gspec = reinterpret_cast<const cf*>(g_item);
    ginv = reinterpret_cast<const float*>(g_item + sizeof(cf) * static_cast<size_t>(g.channels) * C::kSpecPerChan);
  }
#pragma unroll
  for (int r = 0; r < RR; ++r)
#pragma unroll
    for (int e = 0; e < NV; ++e) acc[r][e] = 0.0f;

  issue_nyq(0);
  if constexpr (LEAN) {
  } else if constexpr (PF == RC) {
#pragma unroll
    for (int rc = 0; rc < RC; ++rc) issue_unit(0, rc, nxt[rc]);
  } else {
    issue_unit(0, 0, nxt[0]);
  }
  store_nyq();
  __syncthreads();

  for (int c = 0; c < g.channels; ++c) {
    if (TEAM && g.sync_every > 0 && c > 0 && c % g.sync_every == 0) {
      sync_target += g.team_size;
      if (tid == 0) {  // the other waves run on to the next workgroup barrier
        team_arrive(sync_ctr, 1u);
        team_wait(sync_ctr, sync_target, g.sync_polls);
      }
    }
    // lane coordinates, re-derived from an opaque copy of the lane id every channel (see spr::opaque)
    const int tidv = opaque(tid0);
    const int gc = tidv / C::TGH, tc = tidv - gc * C::TGH;  // column-pass group / lane in group
    const int gr = tidv / C::TGW, tr = tidv - gr * C::TGW;  // row-pass group / lane in group
    // column twiddles: with several column rounds per channel they are read from the LDS table once per
    // channel into registers (live only across the column pass); the row pass reads its table at use
    RegTwiddles<C::EH> twc_reg;
    const LdsTwiddles<C::TGH> twc_lds{twt_h, tc};
    if constexpr (RC > 1 && !LEAN) {
#pragma unroll
      for (int p = 0; p < C::EH; ++p) twc_reg.w[p] = twc_lds.get(p);
    }
    const LdsTwiddles<C::TGW> twr{twt_w, tr};
    cf* cbuf = xbuf + gc * GH::kGroupElems;
    cf* rbuf = xbuf + gr * GW::kGroupElems;
    // ---- column pass: product spectrum -> inverse transforms along k1 -> RT (rows < r_rows) -------
#pragma unroll
    for (int rc = 0; rc < RC; ++rc) {
      const int j = rc * C::CPR + gc;
      const bool active = j < C::COLS;
      float4 (&buf)[2 * H2] = nxt[PF == RC ? rc : 0];
      cf z[C::EH], y[GH::SPL][C::TGH];
      if constexpr (LEAN) {
        // requested and consumed here, in pieces of kPiece register pairs per operand: no operand lives across a transform
        // and at most 2 * kPiece 16-byte loads are in registers beside the products
        constexpr int kPiece = 4;
#if SPR_BIG_ABL == 3
        const float4* gs4 = reinterpret_cast<const float4*>(gspec + static_cast<size_t>(c & 0) * C::kSpecPerChan);
        const float4* qs4 = reinterpret_cast<const float4*>(qspec + static_cast<size_t>(c & 0) * C::kSpecPerChan);
#else
        const float4* gs4 = reinterpret_cast<const float4*>(gspec + static_cast<size_t>(c) * C::kSpecPerChan);
        const float4* qs4 = reinterpret_cast<const float4*>(qspec + static_cast<size_t>(c) * C::kSpecPerChan);
#endif
#pragma unroll
        for (int m0 = 0; m0 < H2; m0 += kPiece) {
          float4 a[kPiece], b[kPiece];
#pragma unroll
          for (int i = 0; i < kPiece; ++i) {
            const size_t idx = (static_cast<size_t>(rc) * H2 + (m0 + i < H2 ? m0 + i : H2 - 1)) * C::NT + tid;
            a[i] = gs4[idx];
            b[i] = qs4[idx];
          }
#pragma unroll
          for (int i = 0; i < kPiece; ++i) {
            if (m0 + i < H2) {
              z[2 * (m0 + i)] = cmul(cmake(a[i].x, a[i].y), cmake(b[i].x, b[i].y));
              z[2 * (m0 + i) + 1] = cmul(cmake(a[i].z, a[i].w), cmake(b[i].z, b[i].w));
            }
          }
        }
      } else {
#pragma unroll
        for (int mm = 0; mm < H2; ++mm) {
          const float4 a = buf[mm], b = buf[H2 + mm];
          z[2 * mm] = cmul(cmake(a.x, a.y), cmake(b.x, b.y));
          z[2 * mm + 1] = cmul(cmake(a.z, a.w), cmake(b.z, b.w));
        }
      }
      // operands of a later unit start flying now
      if constexpr (LEAN) {
      } else if constexpr (PF == RC) issue_unit(c + 1, rc, buf);
      else if (rc + 1 < RC) issue_unit(c, rc + 1, buf);
      else issue_unit(c + 1, 0, buf);
      if constexpr (RC * C::CPR != C::COLS) {  // surplus groups of the last round transform zeros
        if (!active) {
#pragma unroll
          for (int m = 0; m < C::EH; ++m) z[m] = cmake(0.0f, 0.0f);
        }
      }
      if (j == 0) {  // pack column nw/2 into the imaginary part of column 0
#pragma unroll
        for (int m = 0; m < C::EH; ++m) {
          const int k1 = GH::in_index(tc, m);
          z[m] = pk_add_i(z[m], cmul(nyq[k1], nyq[C::NH + k1]));
        }
      }
      auto store_set = [&](int pp) {
        const int p = tc + C::TGH * pp;
        if (!active || ((C::EH % C::TGH != 0) && p >= C::EH)) return;  // surplus groups / idle stage-2 lanes
#if SPR_BIG_ABL == 1
        if (BIG && g.channels > 0) return;
#endif
        cf* col = R + j * rs + p;
#pragma unroll
        for (int s = 0; s < C::TGH; ++s) {
          if (s < s_full) col[C::EH * s] = y[pp][s];                      // compile-time
          else if (s == s_full && p_part > 0) {
            if (p < p_part) col[C::EH * s] = y[pp][s];                    // last, partial row block
          }
        }
      };
      if constexpr (LEAN) {  // every sub-transform set leaves its registers as soon as it is complete
        group_fft<C::EH, C::TGH, +1>(z, y, tc, twc_lds, cbuf, store_set);
      } else {
        if constexpr (RC > 1) group_fft<C::EH, C::TGH, +1>(z, y, tc, twc_reg, cbuf);
        else group_fft<C::EH, C::TGH, +1>(z, y, tc, twc_lds, cbuf);
#pragma unroll
        for (int pp = 0; pp < GH::SPL; ++pp) store_set(pp);
      }
    }
    __syncthreads();
    issue_inv(c);  // consumed after this channel's row transforms
    issue_nyq(c + 1);
    RegTwiddles<C::EW> twr_reg;  // row twiddles: same reasoning, live only across the row pass
    if constexpr (RR > 1 && !LEAN) {
#pragma unroll
      for (int p = 0; p < C::EW; ++p) twr_reg.w[p] = twr.get(p);
    }
    // ---- row pass: two real rows per inverse transform along k2 -> * 1/sigma -> accumulate --------
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      if (rr >= g.rounds_r) break;  // uniform: RR is the variant's compile-time maximum
      if constexpr (LEAN) issue_inv_round(c, rr);
      int pr = rr * C::PPR + gr;
      if (pr >= pairs) pr = pairs - 1;  // duplicate work on surplus lanes; their 1/sigma slots are 0
      // W[k] = Ya[k] + i*Yb[k] for k = tr + TGW*m, built from the stored half spectrum (columns 0..nw/2-1;
      // column nw/2 rides in the imaginary part of column 0).  Which form applies is a compile-time
      // property of the register index m, except for lane tr == 0 of the two registers holding k = 0, nw/2.
#if SPR_BIG_ABL == 2
      const cf* direct = BIG ? R + 2 * (tid & 63) - 0 * rs : R + tr * rs + 2 * pr;
      const cf* mirror = BIG ? R + 2 * (tid & 63) - 0 * rs : R + (C::TGW - tr) * rs + 2 * pr;
#else
      const cf* direct = R + tr * rs + 2 * pr;             // column k          (k < nw/2)
      const cf* mirror = R + (C::TGW - tr) * rs + 2 * pr;  // column nw - k     (k > nw/2), from m = EW-1 down
#endif
      cf wv[C::EW], y[GW::SPL][C::TGW];
#pragma unroll
      for (int m = 0; m < C::EW; ++m) {
        if (m < C::EW / 2) {
          const float4 ab = *reinterpret_cast<const float4*>(direct + m * C::TGW * rs);
          cf v = pk_add_i(cmake(ab.x, ab.y), cmake(ab.z, ab.w));  // Ya + i*Yb
          if (m == 0) v = tr == 0 ? cmake(ab.x, ab.z) : v;        // k = 0: both columns real, values in .x
          wv[m] = v;
        } else if (m == C::EW / 2) {
          // k = nw/2 + tr: lane 0 takes the packed Nyquist column (.y of column 0), the others column nw/2 - tr
          const cf* src = tr == 0 ? R + 2 * pr : mirror + (C::EW - 1 - m) * C::TGW * rs;
          const float4 ab = *reinterpret_cast<const float4*>(src);
          wv[m] = tr == 0 ? cmake(ab.y, ab.w) : pk_conj_add_i(cmake(ab.x, ab.y), cmake(ab.z, ab.w));
        } else {
          const float4 ab = *reinterpret_cast<const float4*>(mirror + (C::EW - 1 - m) * C::TGW * rs);
          wv[m] = pk_conj_add_i(cmake(ab.x, ab.y), cmake(ab.z, ab.w));  // conj(Ya) + i*conj(Yb)
        }
      }
      if constexpr (RR > 1 && !LEAN) group_fft<C::EW, C::TGW, +1>(wv, y, tr, twr_reg, rbuf);
      else group_fft<C::EW, C::TGW, +1>(wv, y, tr, twr, rbuf);
      const float* ivf = reinterpret_cast<const float*>(inv_nxt[LEAN ? 0 : rr]);
#pragma unroll
      for (int pp = 0; pp < GW::SPL; ++pp) {
#pragma unroll
        for (int s = 0; s < KW; ++s) {
          const cf v = y[pp][s];  // lanes without a sub-transform pp hold finite junk and their 1/sigma slots are 0
          const int e = (pp * KW + s) * 2;
          acc[rr][e] = fmaf(v.x, ivf[e], acc[rr][e]);
          acc[rr][e + 1] = fmaf(v.y, ivf[e + 1], acc[rr][e + 1]);
        }
      }
      if (maps_out) {  // debug / parity output of the per-channel maps (spr_ncc_maps): one uniform branch
#pragma unroll
        for (int pp = 0; pp < GW::SPL; ++pp) {
#pragma unroll
          for (int s = 0; s < KW; ++s) {
            const cf v = y[pp][s];
            const int e = (pp * KW + s) * 2;
            const int n2 = GW::out_index(tr, pp, s);
            const int n1 = 2 * (rr * C::PPR + gr);
            const bool ok = GW::out_valid(tr, pp) && n2 < g.iw;
            if (ok && n1 < g.ih) maps_out[(static_cast<size_t>(c) * g.ih + n1) * g.iw + n2] = v.x * ivf[e];
            if (ok && n1 + 1 < g.ih) maps_out[(static_cast<size_t>(c) * g.ih + n1 + 1) * g.iw + n2] = v.y * ivf[e + 1];
          }
        }
      }
    }
    store_nyq();      // channel c+1's Nyquist column, consumed after the barrier
    __syncthreads();  // RT is rewritten by the next channel
  }

  // Slots outside the ih x iw map carry 1/sigma = 0 and stay 0; the score is floored at 0 anyway
  // (similarity.py:355), so they cannot change the result.
  float best = 0.0f;
#pragma unroll
  for (int r = 0; r < RR; ++r)
#pragma unroll
    for (int e = 0; e < NV; ++e) best = fmaxf(best, acc[r][e]);
  best = block_max<C::NT>(best, red);
  if (tid == 0 && scores) {
    const float s = best / static_cast<float>(g.channels);
    float* dst = scores + static_cast<size_t>(qi) * ld + col0 + gi_item;
    const float prev = g.accumulate ? *dst : 0.0f;
    *dst = s > prev ? s : prev;
  }
  }  // epochs
}

// ============================================================================================
// Host side: configurations, LDS layouts, dispatch
// ============================================================================================
struct PrepFftLds {
  size_t x0_off, f_off, xbuf_off, zbuf_off, sat2_off, total;  // sat2_off = 0: the two tables do not fit together
  int f_stride;
  size_t slot_bytes;  // big mode: bytes of one workgroup's workspace slot (x0_off / f_off / sat2_off index it)
};
template <class C>
PrepFftLds prep_fft_lds(const NccGeom& g, bool is_query, int pt) {
  const int h = is_query ? g.th : g.ih, w = is_query ? g.tw : g.iw;
  PrepFftLds l;
  l.f_stride = C::NW / 2 + 1;
  const size_t f_bytes = sizeof(cf) * static_cast<size_t>(2 * ((h + 1) / 2)) * l.f_stride;
  const size_t sat_bytes = align_up(sizeof(double) * (h + 1) * (w + 1), 16);
  if (g.big) {
    // LDS: reduction scratch + exchange buffers; slot: centred map | table 1 (later the row-pass output) | table 2
    l.xbuf_off = 64;
    l.zbuf_off = l.xbuf_off;  // (the two-row split stages through the exchange buffer)
    l.total = l.xbuf_off + sizeof(cf) * C::prep_xbuf_elems(pt);
    l.x0_off = 0;
    l.f_off = align_up(sizeof(float) * h * w, 256);
    const size_t first = f_bytes > sat_bytes ? f_bytes : sat_bytes;
    l.sat2_off = align_up(l.f_off + first, 256);
    l.slot_bytes = align_up(l.sat2_off + (is_query ? 0 : sat_bytes), 256);
    return l;
  }
  l.slot_bytes = 0;
  l.x0_off = 64;
  l.f_off = align_up(l.x0_off + sizeof(float) * h * w, 16);
  l.xbuf_off = align_up(l.f_off + f_bytes, 16);
  l.zbuf_off = l.xbuf_off;
  const size_t fft_total = l.xbuf_off + sizeof(cf) * C::prep_xbuf_elems(pt);
  size_t sat_total = is_query ? 0 : l.f_off + sat_bytes;
  l.sat2_off = 0;
  if (!is_query && l.f_off + 2 * sat_bytes <= static_cast<size_t>(kLdsLimit)) {  // single-sweep 1/sigma
    l.sat2_off = l.f_off + sat_bytes;
    sat_total = l.f_off + 2 * sat_bytes;
  }
  l.total = fft_total > sat_total ? fft_total : sat_total;
  return l;
}

struct PairFftLds {
  size_t r_off, xbuf_off, nyq_off, total, slot_bytes;
};
// rows kept by the column pass: the tuned variant covers RR_A row rounds, the general one the whole grid
template <class C>
constexpr int rk_tuned() { return C::RR_A * C::PPR * 2 < C::NH ? C::RR_A * C::PPR * 2 : C::NH; }

template <class C>
PairFftLds pair_fft_lds(const NccGeom& g) {
  PairFftLds l;
  l.r_off = 64;
  const size_t r_bytes = sizeof(cf) * static_cast<size_t>(C::COLS) * g.r_stride;
  l.slot_bytes = g.big ? align_up(r_bytes, 256) : 0;  // big mode: the image lives in a workspace slot
  l.xbuf_off = align_up(l.r_off + (g.big ? 0 : r_bytes), 16);
  l.nyq_off = align_up(l.xbuf_off + sizeof(cf) * C::xbuf_elems(C::NT), 16);
  l.total = l.nyq_off + sizeof(cf) * (2 * C::NH + C::NH + C::NW);  // Nyquist columns + inverse twiddle tables
  return l;
}

// One entry per instantiated (nh, nw) grid.
struct FftEntry {
  int nh, nw, eh, tgh, ew, tgw, nt, spl_w;
  int kw_a, rr_a, kw_b, rr_b, rk_a;
  bool pow2;
  int spec_per_chan;
  size_t (*prep_lds_total)(const NccGeom&, bool);
  size_t (*pair_lds_total)(const NccGeom&);
  int (*prep)(const NccGeom&, bool, const void*, int64_t, void*, const cf*, const cf*, const FftWorkspace&, hipStream_t);
  int (*pair)(const NccGeom&, bool, const void*, int64_t, const void*, int64_t, float*, int64_t, int64_t, int, float*,
              const cf*, const cf*, unsigned*, const FftWorkspace&, hipStream_t);
  size_t (*prep_slot_bytes)(const NccGeom&, bool);
  size_t (*pair_slot_bytes)(const NccGeom&);
  bool big_only;  // grid whose working set never fits LDS: always the workspace ("big") kernels
  bool six;       // prepared layouts + pair kernel of ncc_pair6.hip (six waves per pair)
};

// Work-items of the prep kernel: 8 waves where the grid gives them work and their exchange buffers fit
template <class C>
int prep_threads(const NccGeom& g, bool q) {
  if (C::NH * C::NW >= 128 * 64 && prep_fft_lds<C>(g, q, 512).total <= static_cast<size_t>(kLdsLimit)) return 512;
  return kThreads;
}
template <class C>
size_t prep_lds_total_t(const NccGeom& g, bool q) { return prep_fft_lds<C>(g, q, prep_threads<C>(g, q)).total; }
template <class C>
size_t pair_lds_total_t(const NccGeom& g) { return pair_fft_lds<C>(g).total; }
template <class C>
size_t prep_slot_bytes_t(const NccGeom& g, bool q) { return prep_fft_lds<C>(g, q, kThreads).slot_bytes; }
template <class C>
size_t pair_slot_bytes_t(const NccGeom& g) { return pair_fft_lds<C>(g).slot_bytes; }

template <class C, bool BIG, int PT>
int prep_launch(const NccGeom& g, bool is_query, const void* maps, int64_t n, void* prepared, const cf* tw_h,
                const cf* tw_w, const FftWorkspace& ws, hipStream_t stream) {
  const PrepFftLds l = prep_fft_lds<C>(g, is_query, PT);
  const size_t item_bytes = is_query ? prepared_query_item_bytes(g, SPR_NCC_FFT) : prepared_gallery_item_bytes(g, SPR_NCC_FFT);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(prep_fft_kernel<C, BIG, PT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
  // big mode: one workspace slot per workgroup of a launch, so the items go in batches the workspace can hold
  int64_t batch = n;
  if (BIG) {
    const size_t per_item = l.slot_bytes * static_cast<size_t>(g.channels);
    if (!ws.base || ws.bytes < per_item) { set_error("prep_fft_kernel: workspace too small for one item"); return SPR_ERR_WORKSPACE; }
    batch = static_cast<int64_t>(ws.bytes / per_item);
  }
  const int raw_h = is_query ? g.q_h : g.g_h, raw_w = is_query ? g.q_w : g.g_w;
  const size_t elem = g.dtype == SPR_F32 ? 4 : 2;
  const size_t raw_item_bytes = static_cast<size_t>(g.channels) * raw_h * raw_w * elem;
  for (int64_t first = 0; first < n; first += batch) {
    const int64_t m = n - first < batch ? n - first : batch;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(prep_fft_kernel<C, BIG, PT>), dim3(g.channels, static_cast<unsigned>(m)), dim3(PT),
                       l.total, stream, g, is_query ? 1 : 0,
                       static_cast<const void*>(static_cast<const unsigned char*>(maps) + first * raw_item_bytes),
                       static_cast<unsigned char*>(prepared) + first * item_bytes, item_bytes, tw_h,
                       tw_w, static_cast<unsigned>(l.x0_off), static_cast<unsigned>(l.f_off),
                       static_cast<unsigned>(l.xbuf_off), static_cast<unsigned>(l.zbuf_off),
                       static_cast<unsigned>(l.sat2_off), l.f_stride, static_cast<unsigned char*>(ws.base), l.slot_bytes);
    const int rc = check_launch("prep_fft_kernel");
    if (rc != SPR_OK) return rc;
  }
  return SPR_OK;
}
template <class C>
int prep_t(const NccGeom& g, bool is_query, const void* maps, int64_t n, void* prepared, const cf* tw_h, const cf* tw_w,
           const FftWorkspace& ws, hipStream_t stream) {
#ifndef SPR_SAN_SUBSET
  if (g.big) return prep_launch<C, true, kThreads>(g, is_query, maps, n, prepared, tw_h, tw_w, ws, stream);
#endif
  if (prep_threads<C>(g, is_query) == 512)
    return prep_launch<C, false, 512>(g, is_query, maps, n, prepared, tw_h, tw_w, ws, stream);
  return prep_launch<C, false, kThreads>(g, is_query, maps, n, prepared, tw_h, tw_w, ws, stream);
}

constexpr int kTeamCounters = 8 * 32;  // one 128-byte line per team

inline int env_int(const char* name, int fallback) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : fallback;
}

template <class C, int RR, int KW, int PF, int RK, bool BIG, bool TEAM>
int pair_launch(const NccGeom& g, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores, int64_t ld,
                int64_t col0, int accumulate, float* maps_out, const cf* tw_h, const cf* tw_w, unsigned* team_sync,
                const FftWorkspace& ws, hipStream_t stream) {
  const PairFftLds l = pair_fft_lds<C>(g);
  const int64_t tiles = static_cast<int64_t>(ceil_div(static_cast<int>(nq), kTileQ)) * ceil_div(static_cast<int>(ng), kTileG);
  PairArgs a{};
  a.channels = g.channels; a.nq = static_cast<int>(nq); a.ng = static_cast<int>(ng);
  a.ih = g.ih; a.iw = g.iw; a.r_rows = g.r_rows; a.r_stride = g.r_stride; a.rounds_r = g.rounds_r;
  a.inv_per_chan = g.inv_per_chan; a.accumulate = accumulate;
  const void* kernel = reinterpret_cast<const void*>(pair_fft_kernel<C, RR, KW, PF, RK, BIG, TEAM>);
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
  unsigned grid = 0;
  // ---- team mode: a persistent grid that exactly fills the device, 8 teams of co-resident workgroups ----
  int per_cu = 0, cus = 0, dev = 0;
  if (TEAM && !team_sync) { set_error("pair_fft_kernel: the team schedule needs the plan's counters"); return SPR_ERR_ARG; }
  if (TEAM && hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pair_fft_kernel<C, RR, KW, PF, RK, BIG, TEAM>, C::NT, l.total) == hipSuccess &&
      per_cu > 0 && cus >= 8) {
    int team_size = cus / 8 * per_cu;
    if (BIG) {  // one workspace slot per resident workgroup: the persistent grid is the only launch form
      const size_t slots = ws.base ? ws.bytes / l.slot_bytes : 0;
      if (slots < 8) { set_error("pair_fft_kernel: workspace too small"); return SPR_ERR_WORKSPACE; }
      if (static_cast<size_t>(team_size) * 8 > slots) team_size = static_cast<int>(slots / 8);
    }
    {
      const int strip_max = env_int("SPR_NCC_STRIP_Q", 16) > 0 ? env_int("SPR_NCC_STRIP_Q", 16) : 16;
      const int strips = ceil_div(static_cast<int>(nq), strip_max);
      a.team_size = team_size;
      a.strips = strips;
      a.strip_q = ceil_div(static_cast<int>(nq), strips);
      a.strips = ceil_div(static_cast<int>(nq), a.strip_q);
      a.epochs_full = static_cast<int>((static_cast<int64_t>(a.strip_q) * ng + team_size - 1) / team_size);
      const int q_last = static_cast<int>(nq) - (a.strips - 1) * a.strip_q;
      a.epochs_total = (a.strips - 1) * a.epochs_full + static_cast<int>((static_cast<int64_t>(q_last) * ng + team_size - 1) / team_size);
      a.sync_polls = env_int("SPR_NCC_TEAM_POLLS", BIG ? 0 : 256);
      a.sync_every = env_int("SPR_NCC_TEAM_EVERY", BIG ? 0 : 32);
      if (a.sync_every < 0) a.sync_every = 0;
      grid = 8u * static_cast<unsigned>(team_size);
      if (hipMemsetAsync(team_sync, 0, sizeof(unsigned) * kTeamCounters, stream) != hipSuccess) {
        set_error("hipMemsetAsync(team counters) failed");
        return SPR_ERR_HIP;
      }
    }
  }
  if (TEAM && a.team_size == 0) { set_error("pair_fft_kernel: could not size the persistent grid"); return SPR_ERR_HIP; }
  // tile mode: HIP refuses grids of 2^32 work-items and more (65 536 tiles of 256-lane workgroups, e.g. 256 queries
  // against a 65 535-item gallery chunk of small maps), so a launch takes a slice of the tiles
  const int64_t max_tiles = TEAM ? tiles : pair_tiles_per_launch(kTileQ * kTileG, C::NT);
  for (int64_t t0 = 0; t0 < (TEAM ? 1 : tiles); t0 += max_tiles) {
    if (!TEAM) {
      const int64_t n = tiles - t0 < max_tiles ? tiles - t0 : max_tiles;
      a.tile0 = static_cast<int>(t0);
      grid = static_cast<unsigned>(n * kTileQ * kTileG);
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_fft_kernel<C, RR, KW, PF, RK, BIG, TEAM>), dim3(grid),
                       dim3(C::NT), l.total, stream, a, static_cast<const unsigned char*>(pq),
                       prepared_query_item_bytes(g, SPR_NCC_FFT), static_cast<const unsigned char*>(pg),
                       prepared_gallery_item_bytes(g, SPR_NCC_FFT), scores,
                       static_cast<long long>(ld), static_cast<long long>(col0), maps_out, tw_h, tw_w,
                       static_cast<unsigned>(l.r_off), static_cast<unsigned>(l.xbuf_off),
                       static_cast<unsigned>(l.nyq_off), team_sync, static_cast<unsigned char*>(ws.base), l.slot_bytes);
    const int rc = check_launch("pair_fft_kernel");
    if (rc != SPR_OK) return rc;
  }
  (void)kernel;
  return SPR_OK;
}

// The workspace instance keeps its accumulators in registers beside 24-point column units: every accumulator it does not
// need is a register it does not spill.  Tuned variant 9 x 4 (maps up to 256 x 108: conv3_3 of an 800 x 400 print), middle
// variant 11 x 4 (up to 256 x 128), general variant for the rest.
constexpr int kBigMidKw = 11, kBigMidRr = 4;
// tuned variant: KW_A x RR_A with PFA prefetch buffers; general variant: everything kept, one buffer
template <class C, int PFA, bool BIG, bool TEAM>
int pair_tb(const NccGeom& g, bool tuned, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores,
            int64_t ld, int64_t col0, int accumulate, float* maps_out, const cf* tw_h, const cf* tw_w,
            unsigned* team_sync, const FftWorkspace& ws, hipStream_t stream) {
  if (tuned)
    return pair_launch<C, C::RR_A, C::KW_A, PFA, rk_tuned<C>(), BIG, TEAM>(g, pq, nq, pg, ng, scores, ld, col0, accumulate,
                                                                           maps_out, tw_h, tw_w, team_sync, ws, stream);
  if constexpr (BIG) {  // the workspace instance's middle variant (fill_geometry): the widest map of its grid, 4 row rounds
    if (g.keep_w == kBigMidKw)
      return pair_launch<C, kBigMidRr, kBigMidKw, 1, 0, BIG, TEAM>(g, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out,
                                                                   tw_h, tw_w, team_sync, ws, stream);
  }
  return pair_launch<C, C::RR_B, C::KW_B, 1, 0, BIG, TEAM>(g, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out,
                                                           tw_h, tw_w, team_sync, ws, stream);
}
template <class C, int PFA>
int pair_t(const NccGeom& g, bool tuned, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores,
           int64_t ld, int64_t col0, int accumulate, float* maps_out, const cf* tw_h, const cf* tw_w,
           unsigned* team_sync, const FftWorkspace& ws, hipStream_t stream) {
#ifndef SPR_SAN_SUBSET  // (the sanitizer build of the CPU emulation compiles the default schedule only)
  if (g.big)
    return pair_tb<C, PFA, true, true>(g, tuned, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, tw_h, tw_w,
                                       team_sync, ws, stream);
  if (team_sync && !maps_out && env_int("SPR_NCC_TEAM", 0) == 1)
    return pair_tb<C, PFA, false, true>(g, tuned, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, tw_h, tw_w,
                                        team_sync, ws, stream);
#endif
  return pair_tb<C, PFA, false, false>(g, tuned, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, tw_h, tw_w,
                                       team_sync, ws, stream);
}

template <class C, int PFA, bool BIG_ONLY = false>
constexpr FftEntry entry() {
  return FftEntry{C::NH,   C::NW,   C::EH,   C::TGH,  C::EW,   C::TGW, C::NT, C::GW::SPL,
                  C::KW_A, C::RR_A, C::KW_B, C::RR_B, rk_tuned<C>(),
                  (C::NH & (C::NH - 1)) == 0 && (C::NW & (C::NW - 1)) == 0,
                  C::kSpecPerChan, prep_lds_total_t<C>, pair_lds_total_t<C>, prep_t<C>, pair_t<C, PFA>,
                  prep_slot_bytes_t<C>, pair_slot_bytes_t<C>, BIG_ONLY, false};
}

// The six-wave pair kernel lives in ncc_pair6.hip; the prep kernel here writes its layouts (Cfg<..., SIX = 1>).
size_t pair6_lds_total(const NccGeom&) { return pair6_lds_bytes(); }
size_t no_slot_bytes(const NccGeom&) { return 0; }
int pair6_t(const NccGeom& g, bool, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores, int64_t ld,
            int64_t col0, int accumulate, float* maps_out, const cf* tw_h, const cf*, unsigned*, const FftWorkspace& ws,
            hipStream_t stream) {
  return launch_pair6(g, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, tw_h, ws.six_ctab, stream);
}
template <class C>
constexpr FftEntry entry6() {
  return FftEntry{C::NH, C::NW, C::EH, C::TGH, C::EW, C::TGW, C::NT, C::GW::SPL, 0, 0, 0, 0, 0, false,
                  C::kSpecPerChan, prep_lds_total_t<C>, pair6_lds_total, prep_t<C>, pair6_t,
                  prep_slot_bytes_t<C>, no_slot_bytes, false, true};
}

// (E, TG) factorisations: 256 = 16*16, 192 = 12*16, 128 = 16*8, 96 = 12*8, 64 = 8*8, 32 = 8*4, 16 = 4*4.
// The workgroup size follows the grid: a transform group is TG lanes, and a 256-lane workgroup on a small
// grid leaves most groups idle (48 x 24 on 256 lanes was slower than 64 x 32).  Measured on an MI355X
// (kernel only): conv5_3 maps [512,32,16]  408 k pairs/s (64 x 32, 256 lanes) -> 879 k (64 x 32, one wave)
// -> 1.22 M (48 x 24, one wave);  conv4_3 maps [512,64,32]  255 k (128 x 64, 256 lanes) -> 295 k (96 x 48,
// 256 lanes) -> 307 k (96 x 48, 192 lanes);  one wave per pair on 128 x 64 spills: 69 k.
//                 EH TGH EW TGW  NT KWA RRA      prefetch buffers of the tuned variant
#ifdef SPR_SAN_SUBSET  // sanitizer build: one grid of each workgroup shape keeps its compile time in minutes
const FftEntry kEntries[] = {
    entry<Cfg<8, 4, 4, 4, 64, 2, 1>, 1>(),
    entry<Cfg<12, 8, 6, 8, 192, 5, 2>, 1>(),
    entry6<Cfg<12, 16, 12, 8, 384, 5, 2, 1>>(),
    entry<Cfg<12, 16, 12, 8, 256, 5, 2>, 1>(),
};
#else
const FftEntry kEntries[] = {
    entry<Cfg<8, 4, 4, 4, 64, 2, 1>, 1>(),        // 32 x 16   (grids this small: one WAVE per pair, no workgroup
    entry<Cfg<8, 4, 8, 4, 64, 2, 1>, 1>(),        // 32 x 32    barriers, up to 16 independent waves per CU)
    entry<Cfg<12, 4, 6, 4, 64, 2, 1>, 1>(),       // 48 x 24: conv5_3 / ResNet layer3 maps of a 512x256 print
    entry<Cfg<8, 8, 8, 4, 64, 2, 1>, 1>(),        // 64 x 32: conv5_3 / ResNet layer3 maps; one WAVE per pair
    entry<Cfg<8, 8, 8, 8, 256, 4, 1>, 1>(),       // 64 x 64
    entry<Cfg<12, 8, 6, 8, 192, 5, 2>, 1>(),      // 96 x 48: conv4_3 maps; 3 waves = its 24 columns x 8 lanes exactly
    entry<Cfg<16, 8, 8, 8, 256, 4, 1>, 1>(),      // 128 x 64
    entry<Cfg<16, 8, 16, 8, 256, 4, 1>, 1>(),     // 128 x 128
    entry6<Cfg<12, 16, 12, 8, 384, 5, 2, 1>>(),   // 192 x 96 on SIX waves per pair (ncc_pair6.hip): maps up to 126 x 64
    entry<Cfg<12, 16, 12, 8, 256, 5, 2>, 1>(),    // 192 x 96: conv3_3 of a 512x256 print; two workgroups per CU
    entry<Cfg<16, 16, 16, 8, 512, 4, 1>, 2>(),    // 256 x 128: 8 waves per workgroup, one workgroup per CU
    entry<Cfg<24, 16, 12, 16, 512, 9, 4>, 1, true>(),  // 384 x 192: maps up to 256 x 128 (conv3_3 of a 1024x512 print,
                                                        // conv2_2 of 512x256); working set in the global workspace
};
#endif

const FftEntry* find_entry(int nh, int nw, int six) {
  for (const FftEntry& e : kEntries)
    if (e.nh == nh && e.nw == nw && (e.six ? 1 : 0) == six) return &e;
  return nullptr;
}

inline int fft_need(int img, int tpl) {
  const int c = tpl / 2;
  const int a = img + c, b = img + tpl - 1 - c;
  return a > b ? a : b;
}

bool fill_geometry(NccGeom& g, const FftEntry& e, bool big) {
  g.big = big ? 1 : 0;
  g.six = e.six ? 1 : 0;
  g.nh = e.nh; g.nw = e.nw; g.eh = e.eh; g.tgh = e.tgh; g.ew = e.ew; g.tgw = e.tgw; g.nt = e.nt;
  if (e.six) {
    // one real-output row transform per image row on three lanes; SPR_NCC_SIX=0 keeps the four-wave kernel (A/B runs)
    if (big || env_int("SPR_NCC_SIX", 1) == 0) return false;
    if (g.ih > pair6_max_rows() || g.iw > pair6_max_cols()) return false;
    g.rounds_c = 2; g.r_rows = g.ih; g.r_stride = 0; g.rounds_r = 1; g.tight = 1; g.keep_w = 0; g.nv = 24;
    g.spec_per_chan = e.spec_per_chan;
    g.inv_per_chan = 6 * e.nt * 4;
    if (g.ih * g.iw > kMaxPixPerThread * kThreads || g.th * g.tw > kMaxPixPerThread * kThreads) return false;
    if (e.prep_lds_total(g, true) > static_cast<size_t>(kLdsLimit)) return false;
    if (e.prep_lds_total(g, false) > static_cast<size_t>(kLdsLimit)) return false;
    return e.pair_lds_total(g) <= static_cast<size_t>(kLdsLimit);
  }
  const int cpr = e.nt / e.tgh, ppr = e.nt / e.tgw;
  g.rounds_c = ceil_div(e.nw / 2, cpr);
  g.r_rows = (g.ih + 7) / 8 * 8;  // rows kept after the column pass (even; rows >= ih carry 1/sigma = 0)
  if (g.r_rows > e.nh) g.r_rows = e.nh;
  g.rounds_r = ceil_div(g.r_rows / 2, ppr);
  const int kw_need = ceil_div(g.iw, e.ew);  // row outputs n2 = p + ew*s with s < kw_need cover iw
  const bool tuned = kw_need <= e.kw_a && g.rounds_r <= e.rr_a;
  g.r_stride = rt_stride(tuned ? e.rk_a : g.r_rows);  // the image holds every row the variant's column pass keeps
  g.tight = tuned ? 1 : 0;
  g.keep_w = tuned ? e.kw_a : e.kw_b;
  if (!tuned && (kw_need > e.kw_b || g.rounds_r > e.rr_b)) return false;
  if (big && !tuned && kw_need <= kBigMidKw && g.rounds_r <= kBigMidRr && kBigMidKw < e.kw_b) g.keep_w = kBigMidKw;
  g.nv = (e.spl_w * g.keep_w * 2 + 3) / 4 * 4;
  g.spec_per_chan = e.spec_per_chan;
  g.inv_per_chan = g.rounds_r * g.nv * e.nt;
  // (the two-sweep 1/sigma path of the LDS mode keeps per-lane register arrays; big mode is always single-sweep)
  if (!big && (g.ih * g.iw > kMaxPixPerThread * kThreads || g.th * g.tw > kMaxPixPerThread * kThreads)) return false;
  if (e.prep_lds_total(g, true) > static_cast<size_t>(kLdsLimit)) return false;
  if (e.prep_lds_total(g, false) > static_cast<size_t>(kLdsLimit)) return false;
  if (e.pair_lds_total(g) > static_cast<size_t>(kLdsLimit)) return false;
  return true;
}

}  // namespace

bool fft_geometry(NccGeom& g, bool pow2_only) {
  const int need_h = fft_need(g.ih, g.th), need_w = fft_need(g.iw, g.tw);
  // the template must also fit the grid (it always does when the image does not alias, except
  // for templates larger than the image)
  const int min_h = need_h > g.th ? need_h : g.th, min_w = need_w > g.tw ? need_w : g.tw;
  const FftEntry* best = nullptr;
  NccGeom best_g = g;
  // smallest grid whose working set fits LDS; failing that, smallest grid with the working set in the global
  // workspace ("big" mode: any size the largest grid covers).  SPR_NCC_FORCE_BIG=1 skips the first pass (tests).
  const bool force_big = env_int("SPR_NCC_FORCE_BIG", 0) == 1;
#ifdef SPR_SAN_SUBSET
  constexpr int kPasses = 1;  // the workspace kernels are not compiled into the sanitizer build
#else
  constexpr int kPasses = 2;
#endif
  for (int pass = force_big ? 1 : 0; pass < kPasses && !best; ++pass) {
    for (const FftEntry& e : kEntries) {
      if (e.nh < min_h || e.nw < min_w) continue;
      if (pow2_only && !e.pow2) continue;
      if (pass == 0 && e.big_only) continue;
      if (best && static_cast<long long>(e.nh) * e.nw >= static_cast<long long>(best->nh) * best->nw) continue;
      NccGeom trial = g;
      if (!fill_geometry(trial, e, pass == 1)) continue;
      best = &e;
      best_g = trial;
    }
  }
  if (!best) return false;
  g = best_g;
  return true;
}

size_t fft_workspace_bytes(const NccGeom& g) {
  if (!g.big) return 0;
  const FftEntry* e = find_entry(g.nh, g.nw, g.six);
  if (!e) return 0;
  const size_t prep_q = e->prep_slot_bytes(g, true), prep_g = e->prep_slot_bytes(g, false);
  const size_t prep = (prep_q > prep_g ? prep_q : prep_g) * static_cast<size_t>(g.channels) * 2;  // two items per launch
  const size_t pair = e->pair_slot_bytes(g) * 512;  // a persistent grid of up to 512 workgroups
  return prep > pair ? prep : pair;
}

int launch_prep_fft(const NccGeom& g, bool is_query, const void* maps, int64_t n, void* prepared, const cf* tw_h,
                    const cf* tw_w, const FftWorkspace& ws, hipStream_t stream) {
  if (n == 0) return SPR_OK;
  const FftEntry* e = find_entry(g.nh, g.nw, g.six);
  if (!e) { set_error("no FFT kernel for grid %dx%d", g.nh, g.nw); return SPR_ERR_UNSUPPORTED; }
  return e->prep(g, is_query, maps, n, prepared, tw_h, tw_w, ws, stream);
}

int launch_pair_fft(const NccGeom& g, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores,
                    int64_t ld, int64_t col0, int accumulate, float* maps_out, const cf* tw_h, const cf* tw_w,
                    unsigned* team_sync, const FftWorkspace& ws, hipStream_t stream) {
  if (nq == 0 || ng == 0) return SPR_OK;
  const FftEntry* e = find_entry(g.nh, g.nw, g.six);
  if (!e) { set_error("no FFT kernel for grid %dx%d", g.nh, g.nw); return SPR_ERR_UNSUPPORTED; }
  return e->pair(g, g.tight != 0, pq, nq, pg, ng, scores, ld, col0, accumulate, maps_out, tw_h, tw_w, team_sync, ws, stream);
}

#ifdef SPR_PREP_STAMPS
extern "C" int spr_debug_read_prep_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prep_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace spr
