// torch.ops.shoeprint_mi355x.* — PyTorch-ROCm custom ops over the C ABI of include/shoeprint_mi355x.h.
//
// SURVEY §8(b) names three device-level operators behind the reference's Python call surface (run.py:20-28):
//   ncc_scores(q [Q,C,h,w], g [G,C,h',w']) -> f32 [Q,G]     compare_maps / _comparison_worker (similarity.py:129-227, :287-375)
//   ranks(scores [Q,G], match i32 [Q])     -> i32 [Q]       _get_rank (similarity.py:378-386)
//   extract(images u8 [N,H,W(,3)], ...)    -> f32 [N,C,h,w] Model.get_feature_maps (network.py:210-244), plain-VGG branches
// This file is plain host C++ (no device code): every operator checks its tensors, takes PyTorch's CURRENT HIP stream and
// calls the same extern "C" entry points the ctypes binding (_lib.py) calls, so both routes run the same kernels bit for
// bit.  Scratch (prepared spectra, workspaces) comes from PyTorch's caching allocator on that stream; plans are cached per
// shape class for the life of the process.  Nothing synchronises.
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <torch/library.h>

#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/shoeprint_mi355x.h"

namespace {

void check(int rc, const char* what) {
  TORCH_CHECK(rc == SPR_OK, "shoeprint_mi355x::", what, ": ", spr_last_error(), " (status ", rc, ")");
}

spr_stream_t current_stream(const at::Tensor& t) {
  return static_cast<spr_stream_t>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

// Makes the tensor's GPU the current HIP device for the call (plans allocate, kernels launch on the current device).
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(const at::Device& d) {
    TORCH_CHECK(hipGetDevice(&prev) == hipSuccess, "hipGetDevice failed");
    if (prev != d.index()) {
      TORCH_CHECK(hipSetDevice(d.index()) == hipSuccess, "hipSetDevice(", int(d.index()), ") failed");
    } else {
      prev = -1;
    }
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void check_device_tensor(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.device().is_cuda(), name, " must live in HBM (a GPU tensor); there is no CPU path");  // torch calls the ROCm device type "cuda"
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

int dtype_code(const at::Tensor& t, const char* name) {
  switch (t.scalar_type()) {
    case at::kFloat: return SPR_F32;
    case at::kHalf: return SPR_F16;
    case at::kBFloat16: return SPR_BF16;
    case at::kUInt16: return SPR_BF16;  // bfloat16 bit patterns (numpy has no bfloat16: the host mirror uploads uint16)
    default: TORCH_CHECK(false, name, ": feature maps are float32, float16 or bfloat16 (or uint16 bfloat16 bit patterns), got ",
                         t.scalar_type());
  }
  return -1;
}

int method_code(const std::string& m) {
  if (m == "auto") return SPR_NCC_AUTO;
  if (m == "fft") return SPR_NCC_FFT;
  if (m == "direct") return SPR_NCC_DIRECT;
  if (m == "fft_pow2") return SPR_NCC_FFT_POW2;
  if (m == "mfma") return SPR_NCC_MFMA;
  TORCH_CHECK(false, "unknown NCC method '", m, "' (auto | fft | direct | fft_pow2 | mfma)");
  return -1;
}

std::mutex g_mutex;
using NccKey = std::tuple<int, int, int, int, int, int, int, int, int>;  // device, C, qh, qw, gh, gw, crop, dtype, method
std::map<NccKey, spr_ncc_plan*> g_ncc_plans;
std::map<std::tuple<int, int, int, int>, spr_vgg16_plan*> g_vgg_plans;  // device, arch, block, compute type

spr_ncc_plan* ncc_plan(int device, int c, int qh, int qw, int gh, int gw, int crop, int dtype, int method) {
  std::lock_guard<std::mutex> lock(g_mutex);
  const NccKey key{device, c, qh, qw, gh, gw, crop, dtype, method};
  auto it = g_ncc_plans.find(key);
  if (it != g_ncc_plans.end()) return it->second;
  spr_ncc_shape shape{c, qh, qw, gh, gw, crop, dtype, method};
  spr_ncc_plan* plan = nullptr;
  check(spr_ncc_plan_create(&shape, &plan), "ncc_scores (plan)");
  g_ncc_plans[key] = plan;
  return plan;
}

// scores[q, g] = get_similarity(q, g) as float32, floored at 0 (similarity.py:355-367 without variants).  The prepared
// gallery is built chunk by chunk when it would not fit `max_prepared_bytes` (0: a third of the free HBM, at most 64 GiB).
at::Tensor ncc_scores(const at::Tensor& q, const at::Tensor& g, int64_t crop, std::string method, int64_t max_prepared_bytes) {
  check_device_tensor(q, "q");
  check_device_tensor(g, "g");
  TORCH_CHECK(q.dim() == 4 && g.dim() == 4, "q and g are [N, C, h, w] batches");
  TORCH_CHECK(q.size(1) == g.size(1), "channel mismatch: queries ", q.size(1), ", gallery ", g.size(1));
  TORCH_CHECK(q.scalar_type() == g.scalar_type() && q.device() == g.device(), "q and g must share storage type and device");
  const DeviceGuard guard(q.device());
  const int64_t nq = q.size(0), ng = g.size(0);
  at::Tensor scores = at::zeros({nq, ng}, q.options().dtype(at::kFloat));
  if (nq == 0 || ng == 0) return scores;
  spr_ncc_plan* plan = ncc_plan(q.device().index(), static_cast<int>(q.size(1)), static_cast<int>(q.size(2)), static_cast<int>(q.size(3)),
                                static_cast<int>(g.size(2)), static_cast<int>(g.size(3)), static_cast<int>(crop), dtype_code(q, "q"),
                                method_code(method));
  const spr_stream_t stream = current_stream(q);
  const auto bytes = q.options().dtype(at::kByte);
  const size_t q_item = spr_ncc_query_bytes(plan, 1), g_item = spr_ncc_gallery_bytes(plan, 1);
  int64_t budget = max_prepared_bytes;
  if (budget <= 0) {
    size_t free_b = 0, total_b = 0;
    TORCH_CHECK(hipMemGetInfo(&free_b, &total_b) == hipSuccess, "hipMemGetInfo failed");
    budget = static_cast<int64_t>(std::min<size_t>(free_b / 3, size_t{64} << 30));
    budget = std::max<int64_t>(budget, int64_t{256} << 20);
  }
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>({ng, budget / static_cast<int64_t>(std::max<size_t>(1, g_item)), 65535}));
  at::Tensor pg = at::empty({static_cast<int64_t>(g_item) * chunk}, bytes);
  const size_t g_stride = static_cast<size_t>(g.stride(0)) * g.element_size();
  for (int64_t q0 = 0; q0 < nq; q0 += 65535) {  // (the pair grid takes at most 65 535 queries per call)
    const int64_t qn = std::min<int64_t>(65535, nq - q0);
    at::Tensor pq = at::empty({static_cast<int64_t>(std::max<size_t>(1, q_item)) * qn}, bytes);
    check(spr_ncc_prepare_queries(plan, static_cast<const char*>(q.data_ptr()) + static_cast<size_t>(q0) * q.stride(0) * q.element_size(),
                                  qn, pq.data_ptr(), stream), "ncc_scores (prepare queries)");
    for (int64_t start = 0; start < ng; start += chunk) {
      const int64_t n = std::min(chunk, ng - start);
      if (q0 == 0 || ng > chunk)
        check(spr_ncc_prepare_gallery(plan, static_cast<const char*>(g.data_ptr()) + static_cast<size_t>(start) * g_stride, n,
                                      pg.data_ptr(), stream), "ncc_scores (prepare gallery)");
      check(spr_ncc_score(plan, pq.data_ptr(), qn, pg.data_ptr(), n, scores.data_ptr<float>() + q0 * ng, ng, start, 0, stream),
            "ncc_scores (score)");
    }
  }
  return scores;
}

// ranks[q] = 1-based place of gallery item match[q] in the descending order of row q (similarity.py:378-386; ties as a
// stable argsort + flip orders them); 0 where match[q] is outside the gallery (the host mirror raises IndexError there).
at::Tensor ranks(const at::Tensor& scores, const at::Tensor& match) {
  check_device_tensor(scores, "scores");
  check_device_tensor(match, "match");
  TORCH_CHECK(scores.dim() == 2 && scores.scalar_type() == at::kFloat, "scores is a float32 [Q, G] matrix");
  TORCH_CHECK(match.dim() == 1 && match.scalar_type() == at::kInt && match.size(0) == scores.size(0), "match is int32 [Q]");
  const DeviceGuard guard(scores.device());
  at::Tensor out = at::zeros({scores.size(0)}, match.options());
  if (scores.size(0) == 0) return out;
  check(spr_rank_true_match(scores.data_ptr<float>(), scores.size(1), scores.size(0), scores.size(1), match.data_ptr<int32_t>(),
                            out.data_ptr<int32_t>(), current_stream(scores)), "ranks");
  return out;
}

// features[:block] of a plain VGG (arch 0 VGG16, 1 VGG19, 2 VGG19_BN; network.py:121-139, :185-186) on a uint8 batch
// [N,H,W] (grey, repeated over three planes: network.py:67) or [N,H,W,3]; `packed` = the weights as spr_vgg16_pack_weights
// wrote them; mean / std as the reference's transforms take them (network.py:60-71); compute = spr_dtype of the plan
// (spr_vgg_plan_create_ex: 0 the exact f32 matrix cores, 1 / 2 float16 / bfloat16 operands).  float32 [N,C,h,w] out.
at::Tensor extract(const at::Tensor& images, const at::Tensor& packed, int64_t arch, int64_t block, std::vector<double> mean,
                   std::vector<double> std_, int64_t compute) {
  check_device_tensor(images, "images");
  check_device_tensor(packed, "packed");
  TORCH_CHECK(images.scalar_type() == at::kByte && (images.dim() == 3 || (images.dim() == 4 && images.size(3) == 3)),
              "images are uint8 [N, H, W] or [N, H, W, 3]");
  TORCH_CHECK(mean.size() == 3 && std_.size() == 3, "mean and std hold three values");
  const DeviceGuard guard(images.device());
  spr_vgg16_plan* plan = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    const auto key = std::make_tuple(static_cast<int>(images.device().index()), static_cast<int>(arch), static_cast<int>(block),
                                     static_cast<int>(compute));
    auto it = g_vgg_plans.find(key);
    if (it == g_vgg_plans.end()) {
      check(spr_vgg_plan_create_ex(static_cast<int32_t>(arch), static_cast<int32_t>(block), static_cast<int32_t>(compute), &plan),
            "extract (plan)");
      g_vgg_plans[key] = plan;
    } else {
      plan = it->second;
    }
  }
  TORCH_CHECK(static_cast<size_t>(packed.numel()) * packed.element_size() >= spr_vgg16_packed_bytes(plan),
              "packed weights: ", packed.numel() * packed.element_size(), " bytes, the plan needs ", spr_vgg16_packed_bytes(plan));
  const int64_t n = images.size(0);
  const int32_t h = static_cast<int32_t>(images.size(1)), w = static_cast<int32_t>(images.size(2));
  int32_t c = 0, oh = 0, ow = 0;
  check(spr_vgg16_output_shape(plan, h, w, &c, &oh, &ow), "extract (shape)");
  at::Tensor out = at::empty({n, c, oh, ow}, images.options().dtype(at::kFloat));
  if (n == 0) return out;
  at::Tensor ws = at::empty({static_cast<int64_t>(std::max<size_t>(16, spr_vgg16_workspace_bytes(plan, n, h, w)))},
                            images.options().dtype(at::kByte));
  const float mean3[3] = {static_cast<float>(mean[0]), static_cast<float>(mean[1]), static_cast<float>(mean[2])};
  const float inv3[3] = {1.0f / static_cast<float>(std_[0]), 1.0f / static_cast<float>(std_[1]), 1.0f / static_cast<float>(std_[2])};
  check(spr_vgg16_forward(plan, images.data_ptr<uint8_t>(), n, h, w, images.dim() == 4 ? 3 : 1, mean3, inv3, packed.data_ptr(),
                          ws.data_ptr(), out.data_ptr<float>(), current_stream(images)), "extract");
  return out;
}

}  // namespace

TORCH_LIBRARY(shoeprint_mi355x, m) {
  m.def("ncc_scores(Tensor q, Tensor g, int crop=2, str method='auto', int max_prepared_bytes=0) -> Tensor");
  m.def("ranks(Tensor scores, Tensor match) -> Tensor");
  m.def("extract(Tensor images, Tensor packed, int arch, int block, float[] mean, float[] std, int compute=0) -> Tensor");
}

// Backend-independent registration: the operators check for GPU tensors themselves (there is no CPU kernel to dispatch to).
TORCH_LIBRARY_IMPL(shoeprint_mi355x, CompositeExplicitAutograd, m) {
  m.impl("ncc_scores", &ncc_scores);
  m.impl("ranks", &ranks);
  m.impl("extract", &extract);
}
