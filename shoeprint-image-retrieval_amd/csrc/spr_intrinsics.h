// Wave-level primitives of gfx950 used by the kernels, in one place.
// (tests/emu/spr_intrinsics.h shadows this file in the CPU emulation build.)
#pragma once
#include <hip/hip_runtime.h>

namespace spr {

constexpr int kWave = 64;  // CDNA wavefront width

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Base of the dynamic LDS allocation (16-byte aligned; no static __shared__ precedes it).
extern __shared__ __attribute__((aligned(16))) unsigned char spr_lds_raw[];
__device__ __forceinline__ unsigned char* dyn_lds() { return spr_lds_raw; }

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

// Rendezvous of the lanes of one wave for data exchanged through LDS: all LDS writes issued
// by the wave before it are visible to LDS reads issued after it.  (A wave executes its LDS
// instructions in order; the fences keep the compiler from moving accesses across.)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, kWave); }
__device__ __forceinline__ int shfl_xor(int v, int mask) { return __shfl_xor(v, mask, kWave); }
__device__ __forceinline__ double shfl_xor(double v, int mask) { return __shfl_xor(v, mask, kWave); }
__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, kWave); }
__device__ __forceinline__ int shfl(int v, int src) { return __shfl(v, src, kWave); }

// Make a lane-constant value opaque to the optimiser at this point: address arithmetic derived from it
// cannot be hoisted out of the enclosing loop (loop-invariant hoisting of ~100 LDS/global addresses costs
// more VGPRs than recomputing them).
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// The instruction scheduler may not move anything across this point (software-pipelined loops: keeps LDS reads
// issued where the source issues them, ahead of the arithmetic that is meant to hide them).
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// Ordering request to the scheduler inside one fenced region: the next group holds up to SIZE instructions of the kinds in
// MASK (0x8 MFMA, 0x2 VALU, 0x4 SALU, 0x20 VMEM read, 0x100 LDS read, 0x200 LDS write); groups are laid down in call order.
template <int MASK, int SIZE>
__device__ __forceinline__ void sched_group() { __builtin_amdgcn_sched_group_barrier(MASK, SIZE, 0); }

// Pin a value in its vector registers at this point of the program (the compiler may not move its computation below).
__device__ __forceinline__ void pin(f32x2& v) { asm volatile("" : "+v"(v)); }

// ---- soft team barrier (locality hint, never a correctness dependency) ------------------------------
// Workgroups that share an L2 announce themselves on a monotonic counter and wait, for a BOUNDED number of
// polls, until the whole team has arrived.  A team member that is late (or not resident at all) only costs
// the others the timeout: every wave leaves the wait, and results never depend on it.
__device__ __forceinline__ void team_arrive(unsigned* counter, unsigned n) {
  __hip_atomic_fetch_add(counter, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void team_wait(unsigned* counter, unsigned target, int max_polls) {
  for (int i = 0; i < max_polls; ++i) {
    const unsigned seen = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (static_cast<int>(seen - target) >= 0) break;
    __builtin_amdgcn_s_sleep(16);
  }
}

// Streamed operands through buffer loads: a wave-uniform resource descriptor (base + size, four scalar registers),
// a 32-bit byte offset per lane and a wave-uniform byte offset in a scalar register.  No 64-bit vector address
// arithmetic per load (three vector instructions each with flat global loads), and the compiler still counts the
// loads in vmcnt.  Raw buffer (stride 0): the hardware checks lane offset + access size against `bytes` and returns
// zeros beyond it; the scalar offset takes no part in that check.  `bytes` < 4 GiB.
struct BufRsrc {
  __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ BufRsrc make_rsrc(const void* uniform_base, size_t bytes) {
  // dword 3 = 0x00020000: 32-bit data format, no swizzle, no add-tid (the value composable_kernel uses on gfx90a+)
  return BufRsrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_base), 0, static_cast<int>(bytes), 0x00020000)};
}
__device__ __forceinline__ float4 buf_ld16(BufRsrc rs, unsigned lane_off, unsigned uniform_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs.r, static_cast<int>(lane_off), static_cast<int>(uniform_off), 0);
  return __builtin_bit_cast(float4, v);
}
__device__ __forceinline__ u32x4 buf_ld16v(BufRsrc rs, unsigned lane_off, unsigned uniform_off) {  // as a native vector
  return __builtin_amdgcn_raw_buffer_load_b128(rs.r, static_cast<int>(lane_off), static_cast<int>(uniform_off), 0);
}
__device__ __forceinline__ f32x2 buf_ld8(BufRsrc rs, unsigned lane_off, unsigned uniform_off) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs.r, static_cast<int>(lane_off), static_cast<int>(uniform_off), 0);
  return __builtin_bit_cast(f32x2, v);
}

// ---- barrier of a GROUP of waves inside a workgroup (s_barrier is all-or-nothing) -------------------------------
// `ctr` is a monotonic arrival counter in LDS, `target` the count that completes this barrier.  A wave's LDS
// instructions execute in order and LDS is one memory per CU, so data a wave wrote to LDS before it arrived is
// visible to every wave that has seen the count; the wavefront-scope fences only pin the compiler's ordering.
__device__ __forceinline__ void group_barrier(unsigned* ctr, unsigned target) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  if (lane_id() == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (;;) {
    const unsigned seen = static_cast<unsigned>(
        __builtin_amdgcn_readfirstlane(static_cast<int>(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))));
    if (static_cast<int>(seen - target) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- complex arithmetic on packed fp32 (a complex number = one 64-bit VGPR pair {re, im}) -----------
// hipcc materialises i*x (swap halves, flip one sign) as v_mov + v_xor before a packed op; the VOP3P
// operand modifiers do it for free.  op_sel / op_sel_hi pick the source half feeding the low / high
// result, neg_lo / neg_hi negate it.

// e + i*o = (e.x - o.y, e.y + o.x)
__device__ __forceinline__ f32x2 pk_add_i(f32x2 e, f32x2 o) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(e), "v"(o));
  return r;
}
// e - i*o = (e.x + o.y, e.y - o.x)
__device__ __forceinline__ f32x2 pk_sub_i(f32x2 e, f32x2 o) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(e), "v"(o));
  return r;
}
// conj(a) + i*conj(b) = (a.x + b.y, b.x - a.y)
__device__ __forceinline__ f32x2 pk_conj_add_i(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a * b (complex): a.xx*b, then += a.yy * (i*b).  One asm statement: between two statements the compiler pads a
// wait state whenever the second reads what the first wrote (it cannot see that these are full-register packed
// writes, which forward like any VALU result).
__device__ __forceinline__ f32x2 pk_cmul(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,0,0]"
      : "=&v"(r) : "v"(a), "v"(b));
  return r;
}
// a * (c + i*s) with wave-uniform c, s (compile-time twiddles live in SGPR pairs): a*c + (i*a)*s
__device__ __forceinline__ f32x2 pk_rot(f32x2 a, float c, float s) {
  const f32x2 t = a * c;
  const f32x2 ss = {s, s};
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0] neg_hi:[0,0,0]"
      : "=v"(r) : "v"(a), "s"(ss), "v"(t));
  return r;
}

// a * k + c with k = one half of the register pair kk (SEL = 0: kk.x, 1: kk.y): lane constants travel two to a pair
template <int SEL>
__device__ __forceinline__ f32x2 pk_axpy(f32x2 a, f32x2 kk, f32x2 c) {
  f32x2 r;
  if constexpr (SEL == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(kk), "v"(c));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(kk), "v"(c));
  return r;
}
// (i*a) * k + c = (c.x - a.y*k, c.y + a.x*k)
template <int SEL>
__device__ __forceinline__ f32x2 pk_iaxpy(f32x2 a, f32x2 kk, f32x2 c) {
  f32x2 r;
  if constexpr (SEL == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0] neg_hi:[0,0,0]"
        : "=v"(r) : "v"(a), "v"(kk), "v"(c));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[0,0,0]"
        : "=v"(r) : "v"(a), "v"(kk), "v"(c));
  return r;
}
// (i*a) * k + c with a wave-uniform k (SGPR pair)
__device__ __forceinline__ f32x2 pk_iaxpy_u(f32x2 a, float k, f32x2 c) {
  const f32x2 kk = {k, k};
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0] neg_hi:[0,0,0]"
      : "=v"(r) : "v"(a), "s"(kk), "v"(c));
  return r;
}
// conj(b) + (i*a) * k = (b.x - a.y*k, a.x*k - b.y)
template <int SEL>
__device__ __forceinline__ f32x2 pk_conj_iaxpy(f32x2 a, f32x2 kk, f32x2 b) {
  f32x2 r;
  if constexpr (SEL == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0] neg_hi:[0,0,1]"
        : "=v"(r) : "v"(a), "v"(kk), "v"(b));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[0,0,1]"
        : "=v"(r) : "v"(a), "v"(kk), "v"(b));
  return r;
}
// The wave-uniform value of v as the compiler should see it (scalar register)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// D = A(16x4) * B(4x16) + C, exact f32 (k-ordered fmaf chain).  Lane l supplies
// A[l&15][l>>4] and B[l>>4][l&15]; it owns D[(l>>4)*4 + j][l&15], j = 0..3.
__device__ __forceinline__ f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// D = A(32x2) * B(2x32) + C.  Lane l supplies A[l&31][l>>5], B[l>>5][l&31]; it owns
// D[(j&3) + 8*(j>>2) + 4*(l>>5)][l&31], j = 0..15.
__device__ __forceinline__ f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// D = A(16x32) * B(32x16) + C on the bf16 matrix cores (exact products, f32 accumulation).  Lane l supplies, as eight
// bf16 bit patterns packed in four dwords (element j in the low / high half of dword j/2), A[l&15][8*(l>>4) + j] and
// B[8*(l>>4) + j][l&15]; it owns D[(l>>4)*4 + r][l&15], r = 0..3.
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// The same with IEEE half-precision operands (products of two halves are exact in f32 as well).
__device__ __forceinline__ f32x4 mfma_f16_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

}  // namespace spr
