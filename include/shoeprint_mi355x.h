/*
 * shoeprint_mi355x.h — C ABI of the MI355X-native shoeprint retrieval hot path.
 *
 * The reference (struan-robertson/shoeprint-image-retrieval) is pure Python and has
 * no FFI layer of its own; its hot path is entered through the Python call surface
 * run.py:20-34 uses.  This header is the C-level boundary our host-side mirror of
 * that surface (shoeprint-image-retrieval_amd/similarity.py, network.py) binds with
 * ctypes, and what a maintainer of the reference would bind to replace
 *
 *   similarity.py:26-72    normxcorr            -> spr_ncc_* (per-channel NCC maps)
 *   similarity.py:75-108   get_similarity       -> spr_ncc_score (one pair = Q=G=1)
 *   similarity.py:355-367  score matrix, floor 0, max over variants -> spr_ncc_score
 *   similarity.py:378-386  _get_rank            -> spr_rank_true_match
 *   network.py:185-244     truncated VGG16 forward -> spr_vgg16_* (see below)
 *   network.py:60-71       ToTensor / repeat(3) / Normalize -> fused into the first conv layer
 *
 * (INTEGRATION.md shows the binding stubs.)
 *
 * Conventions
 *  - Every pointer marked "device" is a HIP device pointer owned by the caller
 *    (PyTorch-ROCm allocations in our host code).  Nothing here allocates or frees
 *    caller-visible memory, and nothing synchronises the stream: all work is
 *    enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream).
 *  - All entry points return SPR_OK (0) or a negative error code; spr_last_error()
 *    gives the message for the calling thread.
 *  - Feature maps are dense row-major [item][channel][row][col] ("[N,C,h,w]"),
 *    float32 unless the plan's dtype says otherwise (customtypes.py:7-14,
 *    network.py:241).
 *  - Thread safety: a plan may be used from one host thread at a time; different
 *    plans are independent.
 *  - A plan may own device scratch its kernels write (the workspace of the 384 x 192 grid, the counters of the team
 *    schedule, the correction matrix of SPR_NCC_MFMA).  Such a plan orders its own calls: spr_ncc_score / spr_ncc_maps record
 *    an event behind their launches and a call arriving on another stream waits for it, so two streams sharing a plan
 *    serialise on it instead of racing; use one plan per stream to overlap layers.
 *  - spr_*_plan_create / spr_*_plan_destroy are synchronous (hipMalloc / hipMemcpy / hipFree on the current device): create
 *    plans outside the hot path, with the device that will run them current.
 */
#ifndef SHOEPRINT_MI355X_H
#define SHOEPRINT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPR_ABI_VERSION 1

enum spr_status {
  SPR_OK = 0,
  SPR_ERR_ARG = -1,         /* null pointer, negative count, bad enum */
  SPR_ERR_SHAPE = -2,       /* shape not representable (e.g. map smaller than the crop) */
  SPR_ERR_UNSUPPORTED = -3, /* valid request this build has no kernel for */
  SPR_ERR_HIP = -4,         /* a HIP runtime call or kernel launch failed */
  SPR_ERR_WORKSPACE = -5    /* caller-provided buffer too small */
};

enum spr_dtype { SPR_F32 = 0, SPR_F16 = 1, SPR_BF16 = 2 };

/* Which pair kernel scores a (query, gallery) pair. */
enum spr_ncc_method {
  SPR_NCC_AUTO = 0,   /* bf16 matrix cores when SPR_NCC_MFMA covers the plan, else FFT when an instantiated grid covers the
                         padded maps, else direct */
  SPR_NCC_FFT = 1,    /* frequency-domain correlation, inverse 2-D FFT per channel: LDS-resident for maps up to
                         ~12 k cropped pixels, working set in a plan-owned device workspace beyond that (maps
                         up to 256 x 128 on the 384 x 192 grid); the plan allocates the workspace itself and
                         spr_ncc_plan_create returns SPR_ERR_WORKSPACE if that allocation fails */
  SPR_NCC_DIRECT = 2, /* sliding-window correlation in LDS (any shape that fits LDS) */
  SPR_NCC_FFT_POW2 = 3, /* as SPR_NCC_FFT but restricted to power-of-two grids (A/B and fallback for the 3*2^k grids) */
  SPR_NCC_MFMA = 4    /* sliding-window correlation as a [queries x taps] x [taps x positions] product on the bf16 / f16 matrix
                         cores: bfloat16 or float16 storage, cropped search maps up to 28 x 12 and cropped templates up to
                         30 x 16 (ResNet50 layer3 / VGG16 conv5_3 / EfficientNet stride-16 maps of a 512 x 256 image and their
                         scaled / rotated query variants; 28 x 12 on both sides has its own tuned instance);
                         SPR_ERR_UNSUPPORTED for anything else */
};

typedef void* spr_stream_t;
typedef struct spr_ncc_plan spr_ncc_plan;

const char* spr_last_error(void);
int spr_abi_version(void);

/* ------------------------------------------------------------------ NCC scorer
 *
 * A plan fixes one (query shape, gallery shape) class: all queries are
 * [C, q_h, q_w], all gallery items [C, g_h, g_w] (raw sizes, before the crop of
 * `crop` pixels per edge that get_similarity applies, similarity.py:92-93; the
 * reference uses crop = 2).  Ragged data sets are handled by the host code with one
 * plan per shape class.
 */
typedef struct spr_ncc_shape {
  int32_t channels;
  int32_t q_h, q_w;
  int32_t g_h, g_w;
  int32_t crop;
  int32_t dtype;   /* spr_dtype of the feature maps handed to spr_ncc_prepare_* */
  int32_t method;  /* spr_ncc_method */
} spr_ncc_shape;

int spr_ncc_plan_create(const spr_ncc_shape* shape, spr_ncc_plan** plan_out);
void spr_ncc_plan_destroy(spr_ncc_plan* plan);
/* The method the plan resolved SPR_NCC_AUTO to (SPR_NCC_FFT / SPR_NCC_DIRECT / SPR_NCC_MFMA). */
int spr_ncc_plan_method(const spr_ncc_plan* plan);
/* FFT grid the plan uses ({0,0} for the direct method): rows, cols. */
int spr_ncc_plan_fft_size(const spr_ncc_plan* plan, int32_t* rows, int32_t* cols);

/* Bytes of device memory the prepared form of n items needs. */
size_t spr_ncc_query_bytes(const spr_ncc_plan* plan, int64_t n_queries);
size_t spr_ncc_gallery_bytes(const spr_ncc_plan* plan, int64_t n_gallery);

/* Prepare n items (device, [n,C,h,w]) into `prepared` (device, >= *_bytes(n)):
 *  queries  (templates): crop, subtract the per-channel mean (similarity.py:48), scale by
 *           1/sqrt(sum t0^2) (:67-68), and for the FFT method transform to the frequency
 *           domain with the 'same'-mode centre shift (h//2, w//2) folded in.
 *  gallery  (search images): crop, subtract the per-channel mean (:49), float64 window
 *           sums -> 1/sqrt(var) map with var<=0 -> 0 (:57-65, :70), and for the FFT method
 *           the forward transform of the zero-padded map. */
int spr_ncc_prepare_queries(spr_ncc_plan* plan, const void* maps, int64_t n, void* prepared,
                            spr_stream_t stream);
int spr_ncc_prepare_gallery(spr_ncc_plan* plan, const void* maps, int64_t n, void* prepared,
                            spr_stream_t stream);

/* scores[q*ld + col0 + g] = max(prev, get_similarity(query q, gallery g)) as float32, where
 * prev is 0 when accumulate_max == 0 (the reference's zero-initialised matrix,
 * similarity.py:355) and the value already stored otherwise (max over rotation/scale
 * variants, :364-367).  scores: device float32 [n_queries, ld]. */
int spr_ncc_score(spr_ncc_plan* plan, const void* prepared_queries, int64_t n_queries,
                  const void* prepared_gallery, int64_t n_gallery, float* scores, int64_t ld,
                  int64_t col0, int accumulate_max, spr_stream_t stream);

/* Debug / parity entry point (scripts/summed_feature_maps.py:1-6, similarity.py:100-104):
 * per-channel NCC maps of ONE prepared query against ONE prepared gallery item, float32
 * [C, g_h-2*crop, g_w-2*crop] (device). */
int spr_ncc_maps(spr_ncc_plan* plan, const void* prepared_query, const void* prepared_gallery,
                 float* maps_out, spr_stream_t stream);

/* ------------------------------------------------------------------ ranking
 * ranks[q] = 1-based position of gallery item match[q] in the descending order of row q
 * (similarity.py:378-386): 1 + #{s_j > s_m} + #{j > m : s_j == s_m} (ties as a stable
 * argsort + flip orders them).  A match index outside [0, n_gallery) gives rank 0 and the
 * call returns SPR_OK; the host mirror turns that into the reference's IndexError.
 * scores: device float32 [n_queries, ld]; match, ranks: device int32 [n_queries]. */
int spr_rank_true_match(const float* scores, int64_t ld, int64_t n_queries, int64_t n_gallery,
                        const int32_t* match, int32_t* ranks, spr_stream_t stream);

/* Partial form for a gallery sharded over ranks: counts[q] = #{j in this shard : s_j > s_m}
 * + #{j in this shard, global index > m : s_j == s_m}, given the true-match score of every
 * query (match_scores, device float32 [n_queries]) and the global index of local column 0.
 * Summing counts over shards and adding 1 gives the rank. */
int spr_rank_count_greater(const float* scores, int64_t ld, int64_t n_queries, int64_t n_local,
                           int64_t global_col0, const float* match_scores, const int32_t* match,
                           int32_t* counts, spr_stream_t stream);

/* dst[i] = keep * dst[i] + weight * src[i] over n float32 score-matrix elements (keep = 0: dst is only written).
 * The reference scores ONE feature layer (run.py:20); a multi-layer score (BASELINE config 5: mean of the conv3_3,
 * conv4_3 and conv5_3 similarities, build-defined) is fused with this on the device. */
int spr_scores_fuse(float* dst, const float* src, int64_t n, float keep, float weight, spr_stream_t stream);

/* ------------------------------------------------------------------ query variants (rotation / scale)
 * similarity.py:230-284 pushes every query feature map through Pillow: Image.rotate(angle) (NEAREST, no
 * expand, zero fill) or Image.resize((int(w*s), int(h*s))) (BICUBIC on mode "F").  These two entry
 * points restate Pillow's C paths bit for bit; the small parameter sets are computed by the host
 * (shoeprint-image-retrieval_amd/variants.py).  in/out: device float32 [n_maps, h, w] -> [n_maps, h', w'].
 *
 * spr_rotate_nearest: mode 0 copy (0 deg), 1 exact flip (180 deg), 2 / 3 exact 90 / 270 transposes (square
 *   maps), 4 Pillow's 16.16 fixed-point affine walk with fixed6 = {a0,a1,a2,a3,a4,a5} (host array).
 * spr_resample_axis: one pass of Pillow's separable resampler along axis 1 (width) or 0 (height):
 *   bounds = device int32 [out_size][2] (first source index, count), coeffs = device float64
 *   [out_size][ksize] (normalised weights), double accumulation, float32 result. */
int spr_rotate_nearest(const float* in, float* out, int64_t n_maps, int32_t h, int32_t w, int32_t mode,
                       const int64_t* fixed6, spr_stream_t stream);
int spr_resample_axis(const float* in, float* out, int64_t n_maps, int32_t h, int32_t w, int32_t axis,
                      int32_t out_size, const int32_t* bounds, const double* coeffs, int32_t ksize,
                      spr_stream_t stream);

/* ------------------------------------------------------------------ feature extractor (VGG16)
 *
 * network.py:125-134, 185-186: torchvision vgg16().features truncated to its first `block` children
 * (index -> layer: 0 conv1_1 1 ReLU 2 conv1_2 3 ReLU 4 pool | 5 conv2_1 6 R 7 conv2_2 8 R 9 pool |
 * 10 conv3_1 11 R 12 conv3_2 13 R 14 conv3_3 15 R 16 pool | 17 conv4_1 .. 22 R 23 pool | 24 conv5_1 .. 29 R
 * 30 pool), convolutions 3x3 / stride 1 / zero pad 1 / bias, pools 2x2 stride 2 (floor).  Every
 * convolution runs as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32
 * fma chains) with bias, ReLU and the following pool fused into its epilogue.
 *
 * Pre-processing fused into the first layer (network.py:60-71, 127-130): x'_c = (pixel/255 - mean_c)/std_c
 * with the zero padding applied AFTER normalisation.  `mean`/`inv_std` are the three per-channel values in
 * [0,1] units (VGG16: mean (0.48235, 0.45882, 0.40784), std 1/255 each).  CLAHE (network.py:197-208) is
 * a separate entry point (spr_clahe_u8) the caller runs before this one.
 */
/* CLAHE of n uint8 images [n, h, w] (device), OpenCV's 8-bit algorithm: tile grid tiles_x x tiles_y
 * (cv2 tileGridSize = (x, y)), clip limit as in cv2.createCLAHE.  workspace: device buffer of
 * spr_clahe_workspace_bytes() (the per-tile look-up tables).  in and out may not alias. */
size_t spr_clahe_workspace_bytes(int64_t n, int32_t tiles_x, int32_t tiles_y);
int spr_clahe_u8(const uint8_t* in, uint8_t* out, int64_t n, int32_t h, int32_t w, float clip_limit,
                 int32_t tiles_x, int32_t tiles_y, void* workspace, spr_stream_t stream);

typedef struct spr_vgg16_plan spr_vgg16_plan;

/* Plain-VGG backbones of network.py:121-139: the truncation model.features[:block] of torchvision's vgg16
 * (cfg "D"), vgg19 (cfg "E") or vgg19_bn.  The plan type keeps its first name; spr_vgg16_plan_create(block)
 * is spr_vgg_plan_create(SPR_VGG16, block).  BatchNorm2d (eval mode) is folded into the preceding
 * convolution's weights and bias by the caller when spr_vgg_conv_info reports it inside the truncation. */
typedef enum spr_vgg_arch { SPR_VGG16 = 0, SPR_VGG19 = 1, SPR_VGG19_BN = 2 } spr_vgg_arch;
int spr_vgg_plan_create(int32_t arch, int32_t block, spr_vgg16_plan** plan_out);
/* As spr_vgg_plan_create with a compute type for the convolutions: SPR_F32 = the f32 matrix cores
 * (exact, the reference's arithmetic: network.py:235 runs the model in float32), SPR_F16 / SPR_BF16 = 16-bit operands with
 * f32 accumulation on v_mfma_f32_16x16x32_{f16,bf16} (BASELINE configs 3 and 5 ask for reduced precision): the weights and
 * the activations BETWEEN layers (and the normalised image the first convolution reads) are rounded to that type (nearest even),
 * bias / ReLU / pool and the last layer's output stay f32; a plan that consists of the first convolution alone stays f32.  Every later call on the plan (packed bytes, pack, workspace, forward)
 * follows the plan's type; the float32 NCHW output and the argument lists do not change. */
int spr_vgg_plan_create_ex(int32_t arch, int32_t block, int32_t compute, spr_vgg16_plan** plan_out);
int spr_vgg_plan_compute(const spr_vgg16_plan* plan);
/* For convolution conv_index: its position in model.features (state-dict key features.<k>.weight) and whether
 * the BatchNorm2d that follows it (features.<k+1>) lies inside features[:block]. */
int spr_vgg_conv_info(const spr_vgg16_plan* plan, int32_t conv_index, int32_t* feature_index, int32_t* bn_inside);
int spr_vgg16_plan_create(int32_t block, spr_vgg16_plan** plan_out);
void spr_vgg16_plan_destroy(spr_vgg16_plan* plan);
/* Number of convolution layers inside features[:block] and their (cin, cout). */
int spr_vgg16_num_convs(const spr_vgg16_plan* plan);
int spr_vgg16_conv_shape(const spr_vgg16_plan* plan, int32_t conv_index, int32_t* cin, int32_t* cout);
/* Output shape [channels, h, w] for an in_h x in_w image. */
int spr_vgg16_output_shape(const spr_vgg16_plan* plan, int32_t in_h, int32_t in_w, int32_t* channels,
                           int32_t* out_h, int32_t* out_w);

/* Re-pack torch-layout parameters into the kernels' layout.  weights[i]: device float32
 * [cout_i, cin_i, 3, 3]; biases[i]: device float32 [cout_i] (host arrays of device pointers, one per
 * convolution).  packed: device buffer of spr_vgg16_packed_bytes(). */
size_t spr_vgg16_packed_bytes(const spr_vgg16_plan* plan);
int spr_vgg16_pack_weights(spr_vgg16_plan* plan, const float* const* weights, const float* const* biases,
                           void* packed, spr_stream_t stream);

/* Forward pass of n images.  images: device uint8, [n, in_h, in_w] (in_channels = 1: the grey value is
 * repeated over the 3 input planes, network.py:67) or [n, in_h, in_w, 3] (in_channels = 3, RGB).
 * workspace: device buffer of spr_vgg16_workspace_bytes(); out: device float32 [n, C, h, w] (NCHW, the
 * layout Model.get_feature_maps returns per image, network.py:241-244). */
size_t spr_vgg16_workspace_bytes(const spr_vgg16_plan* plan, int64_t n, int32_t in_h, int32_t in_w);
int spr_vgg16_forward(spr_vgg16_plan* plan, const uint8_t* images, int64_t n, int32_t in_h, int32_t in_w,
                      int32_t in_channels, const float* mean3, const float* inv_std3, const void* packed,
                      void* workspace, float* out, spr_stream_t stream);

/* As spr_vgg16_forward, and convolution tap_convs[t] (ordinal among the plan's convolutions, not the first one) also
 * writes its activation - after bias / BatchNorm / ReLU, BEFORE any max-pool fused behind it - to tap_out[t] as float32
 * NCHW [n, cout, h, w] of that layer.  One pass of the extractor then feeds a multi-layer score (BASELINE config 5:
 * conv3_3 + conv4_3 + conv5_3 = convolutions 6, 9 and 12 of features[:30]; the reference runs one block per cluster,
 * run.py:20). */
int spr_vgg16_forward_taps(spr_vgg16_plan* plan, const uint8_t* images, int64_t n, int32_t in_h, int32_t in_w,
                           int32_t in_channels, const float* mean3, const float* inv_std3, const void* packed,
                           void* workspace, float* out, int32_t n_taps, const int32_t* tap_convs, float* const* tap_out,
                           spr_stream_t stream);

/* ------------------------------------------------------------------ RGB route of CLAHE (network.py:199-204)
 * The reference equalises a colour image on its L channel: cv2.cvtColor(RGB2LAB) -> CLAHE(L) -> cv2.cvtColor(LAB2RGB).
 * 8-bit convention (L * 255/100, a + 128, b + 128; sRGB, D65); interleaved uint8 [n_pixels, 3] in and out.  `tables`:
 * device buffer of spr_color_tables_bytes() bytes filled by the caller (int32: 256 gamma, 3072 cube-root, 9 matrix,
 * 4096 inverse-gamma entries; shoeprint_image_retrieval_amd/color.py builds them).  Parity with OpenCV itself is
 * unpinned (no cv2 offline): the kernels are bit-identical to oracle/color_oracle.py. */
size_t spr_color_tables_bytes(void);
int spr_rgb_to_lab_u8(const uint8_t* rgb, uint8_t* lab, int64_t n_pixels, const void* tables, spr_stream_t stream);
int spr_lab_to_rgb_u8(const uint8_t* lab, uint8_t* rgb, int64_t n_pixels, const void* tables, spr_stream_t stream);

/* ------------------------------------------------------------------ ResNet50 extractor (build-defined)
 * BASELINE.json config 3 asks for "ResNet50 layer3 summed maps"; the reference has no ResNet branch (network.py:121-182)
 * and its truncation rule `list(model.features.children())[:block]` (network.py:185) has no `.features` to act on there.
 * Defined here as torchvision's resnet50 (v1.5: stride on the 3x3 convolution) cut after `block` of its top-level children
 * [conv1, bn1, relu, maxpool, layer1, layer2, layer3]: block = 5 / 6 / 7 -> [256|512|1024, H/4|H/8|H/16, W/4|W/8|W/16].
 * Same calling sequence as the VGG plan: conv shapes in module order (conv1; per bottleneck conv1, conv2, conv3 and, in a
 * layer's first block, the downsample convolution), weights [cout][cin][k][k] + bias with eval-mode BatchNorm folded in
 * by the caller, packed once; forward = pre-processing (network.py:60-71) + stem + max pool + bottlenecks on the fp32
 * matrix cores, float32 NCHW out.  role: 0 stem, 1/2/3 a bottleneck's convolutions, 4 downsample. */
typedef struct spr_resnet_plan spr_resnet_plan;
int spr_resnet_plan_create(int32_t block, spr_resnet_plan** plan_out);
/* With a compute type, as spr_vgg_plan_create_ex: SPR_F32 (the f32 matrix cores) or SPR_F16 / SPR_BF16 - every convolution,
 * the 7x7 stem included (its operand is the normalised image), on v_mfma_f32_16x16x32 with its (BatchNorm-folded) weights and
 * the activations between layers, the residual operand included, rounded to that type; f32 accumulation, bias, residual sum
 * and ReLU; the float32 NCHW output is unchanged (BASELINE config 3: "ResNet50 layer3 summed maps, bf16"). */
int spr_resnet_plan_create_ex(int32_t block, int32_t compute, spr_resnet_plan** plan_out);
int spr_resnet_plan_compute(const spr_resnet_plan* plan);
void spr_resnet_plan_destroy(spr_resnet_plan* plan);
int spr_resnet_num_convs(const spr_resnet_plan* plan);
int spr_resnet_conv_shape(const spr_resnet_plan* plan, int32_t i, int32_t* cin, int32_t* cout, int32_t* ksize,
                          int32_t* stride, int32_t* role);
int spr_resnet_output_shape(const spr_resnet_plan* plan, int32_t in_h, int32_t in_w, int32_t* channels, int32_t* out_h,
                            int32_t* out_w);
size_t spr_resnet_packed_bytes(const spr_resnet_plan* plan);
int spr_resnet_pack_weights(spr_resnet_plan* plan, const float* const* weights, const float* const* biases, void* packed,
                            spr_stream_t stream);
size_t spr_resnet_workspace_bytes(const spr_resnet_plan* plan, int64_t n, int32_t in_h, int32_t in_w);
int spr_resnet_forward(spr_resnet_plan* plan, const uint8_t* images, int64_t n, int32_t in_h, int32_t in_w,
                       int32_t in_channels, const float* mean3, const float* inv_std3, const void* packed, void* workspace,
                       float* out, spr_stream_t stream);

/* ------------------------------------------------------------------ EfficientNetV2 feature extractor
 * network.py:163-175 (model choice), :185-186 (`list(model.features.children())[:block]`), :60-71 / :74-87 (transforms).
 * arch: 0 = EfficientNetV2_S, 1 = EfficientNetV2_M (the reference's run.toml default), 2 = EfficientNetV2_L, 3 .. 8 =
 * EfficientNet_B1, B2, B3, B4, B5, B7 (network.py:139-162); block in [1, stages + 2] = [1, len(model.features)]: the stem, block - 1
 * stages and, with the last value, the closing 1x1 convolution of torchvision's efficientnet / efficientnet_v2.
 * The plan flattens the graph into layers (spr_effnet_op_info: int32[16] = kind {0 convolution, 1 depthwise 3x3, 2 squeeze-
 * excitation}, cin, cout, cin_p, cout_p, ksize, stride, act {0 none, 2 SiLU}, res, sq, feature index, offsets in floats of
 * w, b, w2, b2 in the packed buffer, block_end).  The caller folds eval-mode BatchNorm (eps 1e-3) into the convolutions and writes
 * the packed buffer itself (device, spr_effnet_packed_bytes):
 *   convolution      w at [cout_p/64][K/16][64][16] with K = tap * cin_p + c (zero where padded), b [cout_p]
 *   depthwise        w [9][c_p], b [c_p]
 *   squeeze-excite   w = fc1 [sq][c_p], b = fc1 bias [sq], w2 = fc2 [c_p][sq], b2 = fc2 bias [c_p]
 * images / mean3 / inv_std3 / out as for spr_vgg16_forward (out: device float32 [n, C, h, w], real channels only). */
typedef struct spr_effnet_plan spr_effnet_plan;
int spr_effnet_plan_create(int32_t arch, int32_t block, spr_effnet_plan** plan_out);
/* With a compute type (as spr_vgg_plan_create_ex): SPR_F16 / SPR_BF16 run every convolution on v_mfma_f32_16x16x32 with the
 * activations between layers stored in that type (the stem reads the rounded normalised image; squeeze-excitation means,
 * factors and the depthwise weights stay f32).  The caller then packs, at the same offsets:
 *   stem             w as 16-bit [k / 8][64][8], k = tap * 3 + plane (zero for k >= 27 and for padded channels), b f32 [64]
 *   convolution      w as 16-bit [cout_p/64][K/64][64][64] with K = tap * cin_p + c, b f32 [cout_p]
 *   depthwise / squeeze-excitation   as in the f32 plans
 * A plan that is the stem alone (block 1) must be f32. */
int spr_effnet_plan_create_ex(int32_t arch, int32_t block, int32_t compute, spr_effnet_plan** plan_out);
int spr_effnet_plan_compute(const spr_effnet_plan* plan);
void spr_effnet_plan_destroy(spr_effnet_plan* plan);
int spr_effnet_num_ops(const spr_effnet_plan* plan);
int spr_effnet_op_info(const spr_effnet_plan* plan, int32_t i, int32_t* info16);
int spr_effnet_output_shape(const spr_effnet_plan* plan, int32_t in_h, int32_t in_w, int32_t* channels, int32_t* out_h,
                            int32_t* out_w);
size_t spr_effnet_packed_bytes(const spr_effnet_plan* plan);
size_t spr_effnet_workspace_bytes(const spr_effnet_plan* plan, int64_t n, int32_t in_h, int32_t in_w);
int spr_effnet_forward(spr_effnet_plan* plan, const uint8_t* images, int64_t n, int32_t in_h, int32_t in_w,
                       int32_t in_channels, const float* mean3, const float* inv_std3, const void* packed, void* workspace,
                       float* out, spr_stream_t stream);

/* ------------------------------------------------------------------ DenseNet_201 feature extractor
 * network.py:176-179, :185-186: torchvision's densenet201 `features` = [conv0, norm0, relu0, pool0, denseblock1, transition1,
 * denseblock2, transition2, denseblock3, transition3, denseblock4, norm5], truncated to features[:block], block in [1, 12].
 * spr_densenet_op_info: int32[12] = kind {0 stem, 1 dense 1x1, 2 dense 3x3, 3 transition, 4 closing BatchNorm}, cin, cout,
 * c_off, ctot, flags {stem: 1 BatchNorm folded, 2 ReLU, 4 max pool}, feature index, offsets in floats of w, b, s, t in the
 * packed buffer, 0.  The caller writes the packed buffer (device, spr_densenet_packed_bytes):
 *   stem        w [tap * 3 + c][64] (norm0 folded when flag 1), b [64]
 *   dense 1x1   s, t [cin] = the BatchNorm in front of it as x * s + t (ReLU follows); w GEMM-packed [128/64][cin/16][64][16]
 *               with the second BatchNorm folded, b [128]
 *   dense 3x3   w GEMM-packed [1][9 * 128 / 16][64][16] (output channels 32 .. 63 zero), b [64] zeros
 *   transition  s, t [cin]; w GEMM-packed [cout/64][cin/16][64][16], b [cout] zeros
 *   closing     s, t [C]
 * images / mean3 / inv_std3 / out as for spr_vgg16_forward. */
typedef struct spr_densenet_plan spr_densenet_plan;
int spr_densenet_plan_create(int32_t block, spr_densenet_plan** plan_out);
void spr_densenet_plan_destroy(spr_densenet_plan* plan);
int spr_densenet_num_ops(const spr_densenet_plan* plan);
int spr_densenet_op_info(const spr_densenet_plan* plan, int32_t i, int32_t* info12);
int spr_densenet_output_shape(const spr_densenet_plan* plan, int32_t in_h, int32_t in_w, int32_t* channels, int32_t* out_h,
                              int32_t* out_w);
size_t spr_densenet_packed_bytes(const spr_densenet_plan* plan);
size_t spr_densenet_workspace_bytes(const spr_densenet_plan* plan, int64_t n, int32_t in_h, int32_t in_w);
int spr_densenet_forward(spr_densenet_plan* plan, const uint8_t* images, int64_t n, int32_t in_h, int32_t in_w,
                         int32_t in_channels, const float* mean3, const float* inv_std3, const void* packed, void* workspace,
                         float* out, spr_stream_t stream);

/* ------------------------------------------------------------------ synthetic data
 * Bench/test support: the device twin of shoeprint_image_retrieval_amd/synth.py (bit-identical
 * float32 values).  out: device float32 [n, C, h, w]. */
int spr_synth_gallery(float* out, int64_t first_item, int64_t n, int32_t channels, int32_t h,
                      int32_t w, uint64_t seed, spr_stream_t stream);
/* match: device int32 [n] = gallery index each query is derived from. */
int spr_synth_queries(float* out, int64_t first_query, int64_t n, const int32_t* match,
                      int32_t channels, int32_t h, int32_t w, uint64_t seed, int32_t max_shift,
                      int32_t signal, int32_t noise, spr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SHOEPRINT_MI355X_H */
