// CPU-emulation shadow of csrc/spr_intrinsics.h (test infrastructure, see hip/hip_runtime.h here).
#pragma once
#include <hip/hip_runtime.h>

namespace spr {

constexpr int kWave = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

inline unsigned char* dyn_lds() { return hipemu::t_blk->lds; }

inline int linear_tid() {
  return static_cast<int>(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
}
inline int lane_id() { return linear_tid() & (kWave - 1); }

inline void wave_sync() { hipemu::yield(hipemu::WAIT_WAVE); }

template <class T>
inline T exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "exchange slot is 8 bytes");
  hipemu::BlockCtx* b = hipemu::t_blk;
  const int me = b->cur, w0 = me & ~(kWave - 1);
  std::memcpy(&b->xchg[me], &v, sizeof(T));
  hipemu::yield(hipemu::WAIT_WAVE);
  int src = w0 + (src_lane & (kWave - 1));
  if (src >= b->nthreads) src = me;
  T r;
  std::memcpy(&r, &b->xchg[src], sizeof(T));
  hipemu::yield(hipemu::WAIT_WAVE);
  return r;
}
inline float shfl_xor(float v, int mask) { return exchange(v, lane_id() ^ mask); }
inline int shfl_xor(int v, int mask) { return exchange(v, lane_id() ^ mask); }
inline double shfl_xor(double v, int mask) { return exchange(v, lane_id() ^ mask); }
inline float shfl(float v, int src) { return exchange(v, src); }
inline int shfl(int v, int src) { return exchange(v, src); }

inline int opaque(int v) { return v; }
typedef float f32x2 __attribute__((ext_vector_type(2)));
inline void pin(f32x2&) {}
inline void sched_fence() {}
template <int MASK, int SIZE>
inline void sched_group() {}

// soft team barrier: workgroups run one after another here, so the wait is the timeout case (returns at once)
inline void team_arrive(unsigned* counter, unsigned n) { __atomic_fetch_add(counter, n, __ATOMIC_RELAXED); }
inline void team_wait(unsigned*, unsigned, int) {}

// buffer loads: the hardware's range check (lane offset + access size against the descriptor's size, the uniform
// offset unchecked) returns zeros; here a violation aborts - the kernels never rely on it
struct BufRsrc {
  const char* base;
  size_t bytes;
};
inline BufRsrc make_rsrc(const void* base, size_t bytes) { return BufRsrc{static_cast<const char*>(base), bytes}; }
inline void buf_check(const BufRsrc& r, unsigned lane_off, unsigned n) {
  if (static_cast<size_t>(lane_off) + n > r.bytes || r.bytes >= (static_cast<size_t>(1) << 32)) {
    std::fprintf(stderr, "hipemu: buffer load outside its descriptor (lane offset %u + %u > %zu)\n", lane_off, n, r.bytes);
    std::abort();
  }
}
inline u32x4 buf_ld16v(BufRsrc r, unsigned lane_off, unsigned uniform_off) {
  buf_check(r, lane_off, 16);
  u32x4 v;
  std::memcpy(&v, r.base + lane_off + uniform_off, 16);
  return v;
}
inline float4 buf_ld16(BufRsrc r, unsigned lane_off, unsigned uniform_off) {
  buf_check(r, lane_off, 16);
  float4 v;
  std::memcpy(&v, r.base + lane_off + uniform_off, 16);
  return v;
}
// barrier of a group of waves on an LDS counter: every work-item spins (yielding to the other fibers)
inline void group_barrier(unsigned* ctr, unsigned target) {
  hipemu::yield(hipemu::WAIT_WAVE);  // the wave's earlier LDS traffic is complete
  if (lane_id() == 0) __atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED);
  while (static_cast<int>(__atomic_load_n(ctr, __ATOMIC_RELAXED) - target) < 0) hipemu::yield(hipemu::RUN);
}

inline f32x2 buf_ld8(BufRsrc r, unsigned lane_off, unsigned uniform_off) {
  buf_check(r, lane_off, 8);
  f32x2 v;
  std::memcpy(&v, r.base + lane_off + uniform_off, 8);
  return v;
}
inline f32x2 pk_add_i(f32x2 e, f32x2 o) { return f32x2{e.x - o.y, e.y + o.x}; }
inline f32x2 pk_sub_i(f32x2 e, f32x2 o) { return f32x2{e.x + o.y, e.y - o.x}; }
inline f32x2 pk_conj_add_i(f32x2 a, f32x2 b) { return f32x2{a.x + b.y, b.x - a.y}; }
inline f32x2 pk_cmul(f32x2 a, f32x2 b) {
  const f32x2 t = {a.x * b.x, a.x * b.y};
  return f32x2{std::fmaf(a.y, -b.y, t.x), std::fmaf(a.y, b.x, t.y)};
}
inline f32x2 pk_rot(f32x2 a, float c, float s) {
  const f32x2 t = {a.x * c, a.y * c};
  return f32x2{std::fmaf(-a.y, s, t.x), std::fmaf(a.x, s, t.y)};
}

template <int SEL>
inline f32x2 pk_axpy(f32x2 a, f32x2 kk, f32x2 c) {
  const float k = SEL ? kk.y : kk.x;
  return f32x2{std::fmaf(a.x, k, c.x), std::fmaf(a.y, k, c.y)};
}
template <int SEL>
inline f32x2 pk_iaxpy(f32x2 a, f32x2 kk, f32x2 c) {
  const float k = SEL ? kk.y : kk.x;
  return f32x2{std::fmaf(-a.y, k, c.x), std::fmaf(a.x, k, c.y)};
}
inline f32x2 pk_iaxpy_u(f32x2 a, float k, f32x2 c) { return f32x2{std::fmaf(-a.y, k, c.x), std::fmaf(a.x, k, c.y)}; }
template <int SEL>
inline f32x2 pk_conj_iaxpy(f32x2 a, f32x2 kk, f32x2 b) {
  const float k = SEL ? kk.y : kk.x;
  return f32x2{std::fmaf(-a.y, k, b.x), std::fmaf(a.x, k, -b.y)};
}
inline int uniform(int v) { return v; }

// Emulated MFMA: every lane publishes its A/B element, then computes its own D entries as the
// k-ordered fmaf chain the hardware produces.
inline f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
  hipemu::BlockCtx* blk = hipemu::t_blk;
  const int me = blk->cur, w0 = me & ~(kWave - 1), l = me - w0;
  blk->xf[8 * me] = a;
  blk->xf[8 * me + 1] = b;
  hipemu::yield(hipemu::WAIT_WAVE);
  f32x4 d = c;
  const int col = l & 15;
  for (int j = 0; j < 4; ++j) {
    const int row = (l >> 4) * 4 + j;
    float acc = c[j];
    for (int k = 0; k < 4; ++k) acc = std::fmaf(blk->xf[8 * (w0 + row + 16 * k)], blk->xf[8 * (w0 + col + 16 * k) + 1], acc);
    d[j] = acc;
  }
  hipemu::yield(hipemu::WAIT_WAVE);
  return d;
}
inline f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
  hipemu::BlockCtx* blk = hipemu::t_blk;
  const int me = blk->cur, w0 = me & ~(kWave - 1), l = me - w0;
  blk->xf[8 * me] = a;
  blk->xf[8 * me + 1] = b;
  hipemu::yield(hipemu::WAIT_WAVE);
  f32x16 d = c;
  const int col = l & 31;
  for (int j = 0; j < 16; ++j) {
    const int row = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5);
    float acc = c[j];
    for (int k = 0; k < 2; ++k) acc = std::fmaf(blk->xf[8 * (w0 + row + 32 * k)], blk->xf[8 * (w0 + col + 32 * k) + 1], acc);
    d[j] = acc;
  }
  hipemu::yield(hipemu::WAIT_WAVE);
  return d;
}

// bf16 MFMA: every lane publishes its two 16-byte fragments; products of bf16 values are exact in float, the
// accumulation is a k-ordered float chain (the hardware's internal order is not documented: tests use tolerances)
template <bool F16>
inline f32x4 mfma_16bit_16x16x32(u32x4 a, u32x4 b, f32x4 c);
inline f32x4 mfma_bf16_16x16x32(u32x4 a, u32x4 b, f32x4 c) { return mfma_16bit_16x16x32<false>(a, b, c); }
inline f32x4 mfma_f16_16x16x32(u32x4 a, u32x4 b, f32x4 c) { return mfma_16bit_16x16x32<true>(a, b, c); }
template <bool F16>
inline f32x4 mfma_16bit_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
  hipemu::BlockCtx* blk = hipemu::t_blk;
  const int me = blk->cur, w0 = me & ~(kWave - 1), l = me - w0;
  std::memcpy(&blk->xf[8 * me], &a, 16);
  std::memcpy(&blk->xf[8 * me + 4], &b, 16);
  hipemu::yield(hipemu::WAIT_WAVE);
  auto elem = [&](int lane, int which, int j) {
    uint16_t bits;
    std::memcpy(&bits, reinterpret_cast<const char*>(&blk->xf[8 * (w0 + lane) + 4 * which]) + 2 * j, 2);
    if (F16) {
      _Float16 h;
      std::memcpy(&h, &bits, 2);
      return static_cast<float>(h);
    }
    const uint32_t u = static_cast<uint32_t>(bits) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  };
  f32x4 d = c;
  const int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k) acc += elem(row + 16 * (k >> 3), 0, k & 7) * elem(col + 16 * (k >> 3), 1, k & 7);
    d[r] = acc;
  }
  hipemu::yield(hipemu::WAIT_WAVE);
  return d;
}

}  // namespace spr
