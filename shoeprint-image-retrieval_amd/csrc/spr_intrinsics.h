// Wave-level primitives of gfx950 used by the kernels, in one place.
// (tests/emu/spr_intrinsics.h shadows this file in the CPU emulation build.)
#pragma once
#include <hip/hip_runtime.h>

namespace spr {

constexpr int kWave = 64;  // CDNA wavefront width

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

}  // namespace spr
