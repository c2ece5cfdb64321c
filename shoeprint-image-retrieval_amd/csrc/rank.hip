// Rank of the true match (similarity.py:378-386) by counting, not sorting:
// rank = 1 + #{s_j > s_m} + #{j > m : s_j == s_m}  (ties ordered as stable argsort + flip would).
// One workgroup per query row; lanes stride the row with coalesced loads, then a wave-shuffle
// and LDS reduction.  HBM-bound: 4 bytes per gallery item per query.
#include "spr_common.h"

namespace spr {
namespace {

__device__ __forceinline__ int block_sum_int(int v, int* scratch) {
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
  const int tid = static_cast<int>(threadIdx.x);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  int s = 0;
  for (int w = 0; w < kThreads / 64; ++w) s += scratch[w];
  return s;
}

// counts[q] (+1 if `final_rank`) for the columns [0, n_local) of this shard; the true-match score
// comes from the row itself (match_scores == nullptr, single shard) or from match_scores[q].
__global__ void __launch_bounds__(kThreads)
rank_kernel(const float* __restrict__ scores, long long ld, long long n_local, long long global_col0,
            const float* __restrict__ match_scores, const int* __restrict__ match, int* __restrict__ out,
            int final_rank) {
  __shared__ int scratch[kThreads / 64];
  const size_t q = blockIdx.x;
  const int tid = static_cast<int>(threadIdx.x);
  const float* row = scores + q * ld;
  const long long m = match[q];
  float sm;
  if (match_scores) {
    sm = match_scores[q];
  } else {
    if (m < 0 || m >= n_local) {  // the reference raises IndexError here (:386); the host mirror does too
      if (tid == 0) out[q] = 0;
      return;
    }
    sm = row[m];
  }
  int cnt = 0;
  for (long long j = tid; j < n_local; j += kThreads) {
    const float s = row[j];
    cnt += (s > sm) || (s == sm && (j + global_col0) > m);
  }
  cnt = block_sum_int(cnt, scratch);
  if (tid == 0) out[q] = cnt + (final_rank ? 1 : 0);
}

// dst[i] = dst[i] * keep + src[i] * weight: the layer fusion of a multi-layer score (build-defined mean of per-layer
// similarities), on the device so that no layer's matrix travels to the host.  HBM-bound, 12 bytes per element.
__global__ void __launch_bounds__(kThreads)
scores_fuse_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n, float keep, float weight) {
  const long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) dst[i] = keep == 0.0f ? src[i] * weight : fmaf(dst[i], keep, src[i] * weight);
}

}  // namespace
}  // namespace spr

extern "C" int spr_scores_fuse(float* dst, const float* src, int64_t n, float keep, float weight, spr_stream_t stream) {
  using namespace spr;
  if (n < 0) { set_error("spr_scores_fuse: bad size"); return SPR_ERR_ARG; }
  if (n == 0) return SPR_OK;
  if (!dst || !src) { set_error("spr_scores_fuse: null pointer"); return SPR_ERR_ARG; }
  const long long blocks = (n + kThreads - 1) / kThreads;
  if (blocks > 0x7fffffffLL) { set_error("spr_scores_fuse: matrix too large for one call"); return SPR_ERR_ARG; }
  hipLaunchKernelGGL(scores_fuse_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), dst, src, static_cast<long long>(n), keep, weight);
  return check_launch("scores_fuse_kernel");
}

extern "C" int spr_rank_true_match(const float* scores, int64_t ld, int64_t n_queries, int64_t n_gallery,
                                   const int32_t* match, int32_t* ranks, spr_stream_t stream) {
  using namespace spr;
  if (n_queries < 0 || n_gallery < 0 || ld < n_gallery) { set_error("spr_rank_true_match: bad sizes"); return SPR_ERR_ARG; }
  if (n_queries == 0) return SPR_OK;
  // (an empty gallery has no score storage: every rank then comes back 0 = "match not in the gallery")
  if ((!scores && n_gallery > 0) || !match || !ranks) { set_error("spr_rank_true_match: null pointer"); return SPR_ERR_ARG; }
  hipLaunchKernelGGL(rank_kernel, dim3(static_cast<unsigned>(n_queries)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), scores, static_cast<long long>(ld),
                     static_cast<long long>(n_gallery), 0LL, static_cast<const float*>(nullptr), match, ranks, 1);
  return check_launch("rank_kernel");
}

extern "C" int spr_rank_count_greater(const float* scores, int64_t ld, int64_t n_queries, int64_t n_local,
                                      int64_t global_col0, const float* match_scores, const int32_t* match,
                                      int32_t* counts, spr_stream_t stream) {
  using namespace spr;
  if (n_queries < 0 || n_local < 0 || ld < n_local) { set_error("spr_rank_count_greater: bad sizes"); return SPR_ERR_ARG; }
  if (n_queries == 0) return SPR_OK;
  if (!scores || !match || !counts || !match_scores) { set_error("spr_rank_count_greater: null pointer"); return SPR_ERR_ARG; }
  hipLaunchKernelGGL(rank_kernel, dim3(static_cast<unsigned>(n_queries)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), scores, static_cast<long long>(ld),
                     static_cast<long long>(n_local), static_cast<long long>(global_col0), match_scores, match,
                     counts, 0);
  return check_launch("rank_count_kernel");
}
