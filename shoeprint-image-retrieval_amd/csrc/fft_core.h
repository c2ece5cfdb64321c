// In-register DFTs and the two-stage "group FFT" used by the NCC kernels.
//
// A transform of length N = E * TG is computed by a group of TG consecutive lanes of one wave,
// each lane holding E complex values in registers:
//
//   on entry   x[m]  = data[t + TG*m]                 (t = lane in group, m = 0..E-1)
//   stage 1    length-E DFT over m in registers      -> U_t[p], p = 0..E-1
//   twiddle    U_t[p] *= w_N^(±t*p)
//   exchange   through LDS: lane t receives U_tt[t + TG*pp] for tt = 0..TG-1, pp = 0..E/TG-1
//   stage 2    E/TG length-TG DFTs over tt in registers
//   on exit    x[m'] = result[t + TG*m'],  m' = pp + (E/TG)*s
//
// i.e. input and output use the same lane/register <-> index map, so a forward transform's
// output can be stored as it lies and loaded as the inverse transform's input.
//
// DIR = -1: forward, kernel exp(-2*pi*i*k*n/N).  DIR = +1: inverse (unnormalised).
#pragma once
#include "spr_common.h"

namespace spr {

__device__ __forceinline__ cf cmake(float x, float y) { return cf{x, y}; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
// multiply by +i / -i
__device__ __forceinline__ cf cmul_i(cf a) { return cf{-a.y, a.x}; }
__device__ __forceinline__ cf cmul_mi(cf a) { return cf{a.y, -a.x}; }
// (a.x b.x - a.y b.y, a.x b.y + a.y b.x) = a.xx * b + a.yy * (i*b): one packed multiply + one packed fma
__device__ __forceinline__ cf cmul(cf a, cf b) { return pk_cmul(a, b); }
__device__ __forceinline__ cf cconj(cf a) { return cf{a.x, -a.y}; }

// cos / sin of 2*pi*k/48, k = 0..47 (covers the 16th and 24th roots of unity), rounded from
// double precision.
__device__ constexpr float kCos48[48] = {
    1.0f, 0.99144486137381038f, 0.96592582628906831f, 0.92387953251128674f, 0.86602540378443860f,
    0.79335334029123517f, 0.70710678118654757f, 0.60876142900872066f, 0.5f, 0.38268343236508984f,
    0.25881904510252074f, 0.13052619222005171f, 0.0f, -0.13052619222005160f, -0.25881904510252063f,
    -0.38268343236508973f, -0.5f, -0.60876142900872054f, -0.70710678118654746f, -0.79335334029123517f,
    -0.86602540378443871f, -0.92387953251128674f, -0.96592582628906831f, -0.99144486137381038f, -1.0f,
    -0.99144486137381049f, -0.96592582628906842f, -0.92387953251128685f, -0.86602540378443882f,
    -0.79335334029123528f, -0.70710678118654768f, -0.60876142900872088f, -0.5f, -0.38268343236509034f,
    -0.25881904510252152f, -0.13052619222005160f, 0.0f, 0.13052619222005127f, 0.25881904510252035f,
    0.38268343236509000f, 0.5f, 0.60876142900872066f, 0.70710678118654735f, 0.79335334029123494f,
    0.86602540378443837f, 0.92387953251128652f, 0.96592582628906820f, 0.99144486137381038f};
__device__ constexpr float kSin48[48] = {
    0.0f, 0.13052619222005157f, 0.25881904510252074f, 0.38268343236508978f, 0.5f, 0.60876142900872066f,
    0.70710678118654746f, 0.79335334029123517f, 0.86602540378443860f, 0.92387953251128674f,
    0.96592582628906831f, 0.99144486137381038f, 1.0f, 0.99144486137381038f, 0.96592582628906831f,
    0.92387953251128674f, 0.86602540378443871f, 0.79335334029123528f, 0.70710678118654757f,
    0.60876142900872066f, 0.5f, 0.38268343236508989f, 0.25881904510252102f, 0.13052619222005157f, 0.0f,
    -0.13052619222005132f, -0.25881904510252079f, -0.38268343236508967f, -0.5f, -0.60876142900872054f,
    -0.70710678118654746f, -0.79335334029123494f, -0.86602540378443849f, -0.92387953251128652f,
    -0.96592582628906809f, -0.99144486137381038f, -1.0f, -0.99144486137381049f, -0.96592582628906842f,
    -0.92387953251128663f, -0.86602540378443882f, -0.79335334029123517f, -0.70710678118654779f,
    -0.60876142900872088f, -0.5f, -0.38268343236509039f, -0.25881904510252157f, -0.13052619222005168f};

// x *= exp(DIR * 2*pi*i * K / N) with K, N compile-time (N divides 48); trivial factors cost nothing.
template <int N, int K, int DIR>
__device__ __forceinline__ cf rot(cf a) {
  constexpr int k = ((K % N) + N) % N;
  if constexpr (k == 0) {
    return a;
  } else if constexpr (2 * k == N) {
    return cmake(-a.x, -a.y);
  } else if constexpr (4 * k == N) {
    return DIR > 0 ? cmul_i(a) : cmul_mi(a);
  } else if constexpr (4 * k == 3 * N) {
    return DIR > 0 ? cmul_mi(a) : cmul_i(a);
  } else {
    constexpr float c = kCos48[k * (48 / N)];
    constexpr float s = DIR > 0 ? kSin48[k * (48 / N)] : -kSin48[k * (48 / N)];
    return pk_rot(a, c, s);
  }
}

template <int N, int DIR>
struct Dft;

template <int DIR>
struct Dft<1, DIR> {
  static __device__ __forceinline__ void run(cf (&)[1]) {}
};
template <int DIR>
struct Dft<2, DIR> {
  static __device__ __forceinline__ void run(cf (&x)[2]) {
    const cf a = x[0], b = x[1];
    x[0] = cadd(a, b);
    x[1] = csub(a, b);
  }
};
template <int DIR>
struct Dft<3, DIR> {
  static __device__ __forceinline__ void run(cf (&x)[3]) {
    // X0 = a+b+c; X1 = a + w b + w^2 c; X2 = a + w^2 b + w c, w = exp(DIR*2pi*i/3)
    const cf a = x[0], s = cadd(x[1], x[2]), d = csub(x[1], x[2]);
    constexpr float h = 0.86602540378443860f;  // sin(2*pi/3)
    const cf m = a - 0.5f * s;
    const cf j = DIR > 0 ? cmul_i(d) * h : cmul_mi(d) * h;  // ±i*h*d
    x[0] = cadd(a, s);
    x[1] = cadd(m, j);
    x[2] = csub(m, j);
  }
};

// Decimation-in-time step: N = 2 * (N/2), even/odd split, natural-order output.
template <int N, int DIR, int K>
struct Combine2 {
  static __device__ __forceinline__ void run(cf (&x)[N], const cf (&e)[N / 2], const cf (&o)[N / 2]) {
    constexpr int k = ((K % N) + N) % N;
    if constexpr (4 * k == N) {  // twiddle = +-i: fused into the packed add (no rotate instruction)
      x[K] = DIR > 0 ? pk_add_i(e[K], o[K]) : pk_sub_i(e[K], o[K]);
      x[K + N / 2] = DIR > 0 ? pk_sub_i(e[K], o[K]) : pk_add_i(e[K], o[K]);
    } else {
      const cf t = rot<N, K, DIR>(o[K]);
      x[K] = cadd(e[K], t);
      x[K + N / 2] = csub(e[K], t);
    }
    if constexpr (K + 1 < N / 2) Combine2<N, DIR, K + 1>::run(x, e, o);
  }
};

template <int N, int DIR>
struct Dft {
  static __device__ __forceinline__ void run(cf (&x)[N]) {
    if constexpr (N % 2 == 0) {
      cf e[N / 2], o[N / 2];
#pragma unroll
      for (int k = 0; k < N / 2; ++k) {
        e[k] = x[2 * k];
        o[k] = x[2 * k + 1];
      }
      Dft<N / 2, DIR>::run(e);
      Dft<N / 2, DIR>::run(o);
      Combine2<N, DIR, 0>::run(x, e, o);
    } else {
      static_assert(N % 3 == 0, "radices 2 and 3 only");
      // N = 3 * (N/3): three interleaved sub-transforms, then length-3 butterflies
      cf a[N / 3], b[N / 3], c[N / 3];
#pragma unroll
      for (int k = 0; k < N / 3; ++k) {
        a[k] = x[3 * k];
        b[k] = x[3 * k + 1];
        c[k] = x[3 * k + 2];
      }
      Dft<N / 3, DIR>::run(a);
      Dft<N / 3, DIR>::run(b);
      Dft<N / 3, DIR>::run(c);
      Combine3<0>(x, a, b, c);
    }
  }
  template <int K>
  static __device__ __forceinline__ void Combine3(cf (&x)[N], const cf (&a)[N / 3], const cf (&b)[N / 3],
                                                   const cf (&c)[N / 3]) {
    cf t[3] = {a[K], rot<N, K, DIR>(b[K]), rot<N, 2 * K, DIR>(c[K])};
    Dft<3, DIR>::run(t);
    x[K] = t[0];
    x[K + N / 3] = t[1];
    x[K + 2 * N / 3] = t[2];
    if constexpr (K + 1 < N / 3) Combine3<K + 1>(x, a, b, c);
  }
};

// ---------------------------------------------------------------------------------------------------
// Group FFT, general form: length N = E * TG on TG consecutive lanes with E complex registers each.
//
//   input    x[m]      = data[t + TG*m]                                  m = 0..E-1
//   stage 1  length-E DFT over m (registers), then twiddle w_N^(DIR*t*p)  -> U_t[p], p = 0..E-1
//   stage 2  E sub-transforms of length TG over the lanes; lane t owns p = t + TG*pp (pp = 0..SPL-1,
//            SPL = ceil(E/TG); lanes whose p >= E idle for that pp — e.g. E = 12, TG = 16 uses 12 of 16
//            lanes in stage 2, E = 12, TG = 8 runs 8 + 4)
//   output   y[pp][s]  = result[(t + TG*pp) + E*s]                       s = 0..TG-1
//
// The all-to-all between the stages goes through LDS in SPL phases: in phase pp every lane writes the
// (up to TG) values with p in [TG*pp, TG*pp + TG) and the owning lanes read one row of TG values, so the
// exchange image of a group is only min(E,TG) x (TG+1) complex values (not E x TG), and the stage-2
// DFT of phase pp can overlap the LDS traffic of phase pp+1 of the other wave on the SIMD.
// E need not be a multiple of TG: that is what admits the 3*2^k grids (192 = 12*16, 96 = 12*8).
template <int E, int TG>
struct GroupFft {
  static constexpr int N = E * TG;
  static constexpr int SPL = (E + TG - 1) / TG;
  // rows of the exchange image = values of p exchanged per phase: one stage-2 sub-transform set (TG rows)
  // when E >= TG; when E < TG (a single set, E rows) at most 8 rows at a time to keep the image small
  static constexpr int PH = E >= TG ? TG : (E > 8 ? 8 : E);
  static constexpr int NPH = (E + PH - 1) / PH;  // phases
  static constexpr int kRow = TG + 1;         // +1: the strided row reads of stage 2 spread over the banks
  static constexpr int kRaw = PH * kRow;
  // group stride == TG (mod 16): the 16/TG groups sharing a 16-lane ds_write_b64 slice start on different banks
  static constexpr int kGroupElems = kRaw + ((TG % 16) - (kRaw % 16) + 16) % 16;
  static constexpr int block_elems(int threads) { return (threads / TG) * kGroupElems; }
  static __host__ __device__ constexpr int in_index(int t, int m) { return t + TG * m; }
  static __host__ __device__ constexpr bool out_valid(int t, int pp) { return t + TG * pp < E; }
  static __host__ __device__ constexpr int out_index(int t, int pp, int s) { return (t + TG * pp) + E * s; }
};

// Twiddle sources: registers (w[p] = w_N^(DIR*t*p), loaded once per kernel) ...
template <int E>
struct RegTwiddles {
  cf w[E];
  __device__ __forceinline__ cf get(int p) const { return w[p]; }
};
template <int E, int TG, int DIR>
__device__ __forceinline__ void load_twiddles(RegTwiddles<E>& r, const cf* __restrict__ tw, int t) {
#pragma unroll
  for (int p = 0; p < E; ++p) {
    const cf w = tw[t * p];  // forward table tw[k] = exp(-2*pi*i*k/N), t*p < N
    r.w[p] = DIR > 0 ? cconj(w) : w;
  }
}
// ... or a table in LDS laid out [p][t] = w_N^(DIR*t*p) (E rows of TG entries), read at use: saves 2E VGPRs
// per lane, and the TG lanes of a group read TG consecutive entries (conflict-free, all groups broadcast).
template <int TG>
struct LdsTwiddles {
  const cf* table;
  int t;
  __device__ __forceinline__ cf get(int p) const { return table[p * TG + t]; }
};

// Every lane of the wave must call this together; `xbuf` is the group's exchange image
// (GroupFft<E,TG>::kGroupElems complex values); it may be reused as soon as the function returns.
struct NoSink {
  __device__ __forceinline__ void operator()(int) const {}
};
// `sink(pp)` is called as soon as sub-transform set pp is complete (E >= TG): a caller that stores y[pp] there does not
// hold it in registers through the remaining phases.
template <int E, int TG, int DIR, class Tw, int SPL, class Sink = NoSink>
__device__ __forceinline__ void group_fft(cf (&x)[E], cf (&y)[SPL][TG], int t, const Tw& tw, cf* xbuf,
                                          const Sink& sink = Sink{}) {
  using G = GroupFft<E, TG>;
  static_assert(SPL == G::SPL, "y must be [GroupFft<E,TG>::SPL][TG]");
  Dft<E, DIR>::run(x);
#pragma unroll
  for (int ph = 0; ph < G::NPH; ++ph) {
#pragma unroll
    for (int r = 0; r < G::PH; ++r) {
      const int p = G::PH * ph + r;
      if (p < E) xbuf[r * G::kRow + t] = p == 0 ? x[0] : cmul(x[p], tw.get(p));
    }
    wave_sync();
    constexpr int kE = E, kTG = TG;
    const int kValid = kE - G::PH * ph < G::PH ? kE - G::PH * ph : G::PH;  // image rows this phase wrote (folds after unrolling)
    const int pp = kE >= kTG ? ph : 0;         // the sub-transform set this phase feeds
    const int row = t + TG * pp - G::PH * ph;  // image row holding this lane's p = t + TG*pp, if in this phase
    if constexpr (E % TG == 0) {
#pragma unroll
      for (int tt = 0; tt < TG; ++tt) y[pp][tt] = xbuf[row * G::kRow + tt];
    } else {
      // Lanes that own no sub-transform in this set (p >= E) still need defined, finite inputs: they read some
      // valid row — their results are never stored (column pass) or meet a zero 1/sigma (row pass) — which
      // avoids zero-filling registers and an exec-masked branch.  When E < TG spans several phases, a later
      // phase may only overwrite the lanes whose row it carries.
      const bool own = row >= 0 && row < kValid;
      const int rsafe = own ? row : (t % kValid);
      if (kE >= kTG || ph == 0) {
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) y[pp][tt] = xbuf[rsafe * G::kRow + tt];
      } else if (own) {
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) y[pp][tt] = xbuf[row * G::kRow + tt];
      }
    }
    wave_sync();  // the image is rewritten by the next phase / the next call
    // Lanes without a sub-transform in this set sit the stage-2 DFT out (exec-masked): their registers keep the
    // finite values read above.  The kernels run at the board's power limit (~1.4 kW, sclk ~2.0 GHz instead of
    // 2.4), so work that idle lanes do not do is clock headroom for the others.
    if constexpr (E >= TG) {
      if (E % TG == 0 || G::out_valid(t, ph)) Dft<TG, DIR>::run(y[ph]);
      sink(ph);
    }
  }
  if constexpr (E < TG) {
    if (G::out_valid(t, 0)) Dft<TG, DIR>::run(y[0]);
  }
}

}  // namespace spr
