// Micro-benchmark: issue rate of scalar vs packed fp32 VALU on gfx950, at 1/2/4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int ITER>
__global__ void k_scalar(float* out, float s) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(0.5f));
  }
  float r = 0;
  for (int i = 0; i < 16; ++i) r += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int ITER>
__global__ void k_packed(float* out, float s) {
  v2f a[8];
  for (int i = 0; i < 8; ++i) a[i] = v2f{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const v2f sv = {s, s}, h = {0.5f, 0.25f};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], sv, h);
  }
  v2f r = {0, 0};
  for (int i = 0; i < 8; ++i) r += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;
}
template <int ITER>
__global__ void k_packed_add(float* out, float s) {
  v2f a[8];
  for (int i = 0; i < 8; ++i) a[i] = v2f{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const v2f h = {s, 0.25f};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = a[i] + h;
  }
  v2f r = {0, 0};
  for (int i = 0; i < 8; ++i) r += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;
}
template <class K>
double run(K kern, int blocks, int threads, float* out, double flop_per_thread) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return flop_per_thread * blocks * threads * 5 / (ms * 1e-3) / 1e12;
}
int main() {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 16 * 1024);
  constexpr int IT = 4096;
  for (int waves_per_simd : {1, 2, 4, 8}) {
    int threads = 256, blocks = 256 * waves_per_simd;  // 4 waves per block -> blocks/CU = waves/SIMD
    double s = run(k_scalar<IT>, blocks, threads, out, 2.0 * 16 * IT);
    double p = run(k_packed<IT>, blocks, threads, out, 2.0 * 16 * IT);
    double pa = run(k_packed_add<IT>, blocks, threads, out, 1.0 * 16 * IT);
    printf("waves/SIMD %d: v_fma_f32 %.1f TFLOP/s | v_pk_fma_f32 %.1f TFLOP/s | v_pk_add_f32 %.1f Tadd/s\n", waves_per_simd, s, p, pa);
  }
  return 0;
}
