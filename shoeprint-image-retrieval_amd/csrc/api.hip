// C ABI entry points of libshoeprint_mi355x.so (include/shoeprint_mi355x.h) for the NCC scorer.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include "spr_common.h"

namespace spr {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SPR_ERR_HIP;
  }
  return SPR_OK;
}

int64_t pair_tiles_per_launch(int pairs_per_tile, int threads) {
  int64_t t = ((static_cast<int64_t>(1) << 32) - 1) / (static_cast<int64_t>(pairs_per_tile) * threads);
  const int64_t by_blocks = ((static_cast<int64_t>(1) << 31) - 1) / pairs_per_tile;
  if (by_blocks < t) t = by_blocks;
  const char* v = std::getenv("SPR_NCC_MAX_TILES");
  if (v && *v) {
    const long long cap = std::atoll(v);
    if (cap >= 1 && cap < t) t = cap;
  }
  return t;
}

size_t prepared_query_item_bytes(const NccGeom& g, int method) {
  size_t b;
  if (method == SPR_NCC_MFMA) return mfma_query_item_bytes(g);
  if (method == SPR_NCC_FFT)
    b = sizeof(cf) * static_cast<size_t>(g.channels) * g.spec_per_chan + static_cast<size_t>(g.channels);  // + dead flags
  else
    b = sizeof(float) * static_cast<size_t>(g.channels) * g.th * g.tw;
  return align_up(b, 256);
}

size_t prepared_gallery_item_bytes(const NccGeom& g, int method) {
  size_t b;
  if (method == SPR_NCC_MFMA) return mfma_gallery_item_bytes(g);
  if (method == SPR_NCC_FFT)
    b = static_cast<size_t>(g.channels) * (sizeof(cf) * g.spec_per_chan + sizeof(float) * g.inv_per_chan) +
        static_cast<size_t>(g.channels);  // + one dead flag per channel
  else
    b = sizeof(float) * static_cast<size_t>(g.channels) * g.ih * g.iw * 2;
  return align_up(b, 256);
}

}  // namespace spr

struct spr_ncc_plan {
  spr::NccGeom geom;
  int method;              // resolved: SPR_NCC_FFT, SPR_NCC_DIRECT or SPR_NCC_MFMA
  spr::cf* tw_h = nullptr;  // device: exp(-2*pi*i*k/nh), k < nh
  spr::cf* tw_w = nullptr;  // device: exp(-2*pi*i*k/nw), k < nw
  unsigned* team_sync = nullptr;  // device: arrival counters of the pair kernel's 8 workgroup teams
  spr::FftWorkspace ws{nullptr, 0, nullptr};  // device: scratch of the "big" geometries (maps beyond LDS)
  float* six_ctab = nullptr;  // device: pre-twist table of the six-wave pair kernel
  float* mfma_x = nullptr;    // device: correction matrix of the matrix-core method's exact form (one call at a time per plan)
  // A plan that owns device scratch its kernels write (mfma_x, team_sync, ws) orders its own calls: every scoring call
  // records `done` behind its launches, and a call that arrives on ANOTHER stream waits for it first - two streams sharing a
  // plan (one scorer, equal-shaped layers on separate streams) then serialise instead of racing on the scratch.
  hipEvent_t done = nullptr;
  hipStream_t last_stream = nullptr;
  bool used = false;
  bool team_mode = false;  // SPR_NCC_TEAM at plan creation: the team schedule's counters are written by the kernels
  bool owns_scratch() const { return mfma_x || ws.base || (team_sync && team_mode); }
};

// before / after the launches of a scoring call on stream s
static int plan_enter(spr_ncc_plan* plan, hipStream_t s) {
  if (!plan->owns_scratch()) return SPR_OK;
  if (!plan->done && hipEventCreateWithFlags(&plan->done, hipEventDisableTiming) != hipSuccess) {
    spr::set_error("hipEventCreate failed");
    return SPR_ERR_HIP;
  }
  if (plan->used && plan->last_stream != s && hipStreamWaitEvent(s, plan->done, 0) != hipSuccess) {
    spr::set_error("hipStreamWaitEvent failed");
    return SPR_ERR_HIP;
  }
  return SPR_OK;
}
static void plan_leave(spr_ncc_plan* plan, hipStream_t s) {
  if (!plan->owns_scratch() || !plan->done) return;
  (void)hipEventRecord(plan->done, s);
  plan->last_stream = s;
  plan->used = true;
}

using namespace spr;

static int make_twiddles(int n, cf** out) {
  std::vector<cf> host(static_cast<size_t>(n));
  const double two_pi = 6.283185307179586476925286766559;
  for (int k = 0; k < n; ++k) {
    const double a = -two_pi * static_cast<double>(k) / static_cast<double>(n);
    host[k].x = static_cast<float>(std::cos(a));
    host[k].y = static_cast<float>(std::sin(a));
  }
  // exact values on the axes (cos/sin of multiples of pi/2 are not exact in floating point)
  if (n % 4 == 0) {
    host[n / 4] = cf{0.0f, -1.0f};
    host[n / 2] = cf{-1.0f, 0.0f};
    host[3 * n / 4] = cf{0.0f, 1.0f};
  } else if (n % 2 == 0) {
    host[n / 2] = cf{-1.0f, 0.0f};
  }
  void* dev = nullptr;
  if (hipMalloc(&dev, sizeof(cf) * n) != hipSuccess) { set_error("hipMalloc(twiddles) failed"); return SPR_ERR_HIP; }
  if (hipMemcpy(dev, host.data(), sizeof(cf) * n, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(dev);
    set_error("hipMemcpy(twiddles) failed");
    return SPR_ERR_HIP;
  }
  *out = static_cast<cf*>(dev);
  return SPR_OK;
}

extern "C" const char* spr_last_error(void) { return g_error; }
extern "C" int spr_abi_version(void) { return SPR_ABI_VERSION; }

extern "C" int spr_ncc_plan_create(const spr_ncc_shape* shape, spr_ncc_plan** plan_out) {
  if (!shape || !plan_out) { set_error("spr_ncc_plan_create: null pointer"); return SPR_ERR_ARG; }
  *plan_out = nullptr;
  if (shape->channels <= 0 || shape->crop < 0) { set_error("spr_ncc_plan_create: bad channels/crop"); return SPR_ERR_ARG; }
  if (shape->dtype != SPR_F32 && shape->dtype != SPR_F16 && shape->dtype != SPR_BF16) {
    set_error("spr_ncc_plan_create: unknown dtype %d", shape->dtype);
    return SPR_ERR_ARG;
  }
  if (shape->method != SPR_NCC_AUTO && shape->method != SPR_NCC_FFT && shape->method != SPR_NCC_DIRECT &&
      shape->method != SPR_NCC_FFT_POW2 && shape->method != SPR_NCC_MFMA) {
    set_error("spr_ncc_plan_create: unknown method %d", shape->method);
    return SPR_ERR_ARG;
  }
  NccGeom g{};
  g.channels = shape->channels;
  g.q_h = shape->q_h; g.q_w = shape->q_w; g.g_h = shape->g_h; g.g_w = shape->g_w;
  g.crop = shape->crop;
  g.dtype = shape->dtype;
  g.th = g.q_h - 2 * g.crop; g.tw = g.q_w - 2 * g.crop;
  g.ih = g.g_h - 2 * g.crop; g.iw = g.g_w - 2 * g.crop;
  if (g.th < 1 || g.tw < 1 || g.ih < 1 || g.iw < 1) {
    set_error("spr_ncc_plan_create: maps %dx%d / %dx%d vanish under a crop of %d", g.q_h, g.q_w, g.g_h, g.g_w, g.crop);
    return SPR_ERR_SHAPE;
  }
  int method = 0;
  NccGeom gf = g, gd = g;
  const bool fft_ok = fft_geometry(gf, shape->method == SPR_NCC_FFT_POW2), direct_ok = direct_geometry(gd);
  // SPR_NCC_MFMA=0 in the environment keeps SPR_NCC_AUTO off the matrix-core kernel (A/B runs)
  const char* mfma_env = std::getenv("SPR_NCC_MFMA");
  NccGeom gm = g;
  const bool mfma_ok = mfma_geometry(gm), mfma_auto = mfma_ok && !(mfma_env && mfma_env[0] == '0');
  if (shape->method == SPR_NCC_MFMA) {
    if (!mfma_ok) { set_error("matrix-core method: bfloat16 / float16 maps of 28x12 (cropped) on both sides only, got dtype %d, query %dx%d vs gallery %dx%d", g.dtype, g.th, g.tw, g.ih, g.iw); return SPR_ERR_UNSUPPORTED; }
    method = SPR_NCC_MFMA;
  } else if (shape->method == SPR_NCC_AUTO && mfma_auto) {
    method = SPR_NCC_MFMA;
  } else if (shape->method == SPR_NCC_FFT || shape->method == SPR_NCC_FFT_POW2) {
    if (!fft_ok) { set_error("FFT method: no instantiated LDS-resident grid fits query %dx%d vs gallery %dx%d", g.th, g.tw, g.ih, g.iw); return SPR_ERR_UNSUPPORTED; }
    method = SPR_NCC_FFT;
  } else if (shape->method == SPR_NCC_DIRECT) {
    if (!direct_ok) { set_error("direct method: maps %dx%d vs %dx%d do not fit LDS", g.th, g.tw, g.ih, g.iw); return SPR_ERR_UNSUPPORTED; }
    method = SPR_NCC_DIRECT;
  } else if (fft_ok) {
    method = SPR_NCC_FFT;
  } else if (direct_ok) {
    method = SPR_NCC_DIRECT;
  } else {
    set_error("no NCC kernel fits query %dx%d vs gallery %dx%d (cropped) in LDS", g.th, g.tw, g.ih, g.iw);
    return SPR_ERR_UNSUPPORTED;
  }
  spr_ncc_plan* p = new (std::nothrow) spr_ncc_plan();
  if (!p) { set_error("out of host memory"); return SPR_ERR_ARG; }
  p->method = method;
  p->geom = method == SPR_NCC_FFT ? gf : method == SPR_NCC_MFMA ? gm : gd;
  if (method == SPR_NCC_MFMA && mfma_workspace_bytes(p->geom) > 0 &&
      hipMalloc(reinterpret_cast<void**>(&p->mfma_x), mfma_workspace_bytes(p->geom)) != hipSuccess) {
    set_error("hipMalloc(%zu bytes of correction matrix) failed", mfma_workspace_bytes(p->geom));
    spr_ncc_plan_destroy(p);
    return SPR_ERR_WORKSPACE;
  }
  if (method == SPR_NCC_FFT) {
    int rc = make_twiddles(p->geom.nh, &p->tw_h);
    if (rc == SPR_OK) rc = make_twiddles(p->geom.nw, &p->tw_w);
    if (rc == SPR_OK && hipMalloc(reinterpret_cast<void**>(&p->team_sync), sizeof(unsigned) * 8 * 32) != hipSuccess) {
      set_error("hipMalloc(team counters) failed");
      rc = SPR_ERR_HIP;
    }
    const size_t ws_bytes = fft_workspace_bytes(p->geom);
    if (rc == SPR_OK && ws_bytes > 0) {
      if (hipMalloc(&p->ws.base, ws_bytes) != hipSuccess) {
        set_error("hipMalloc(%zu bytes of FFT workspace) failed", ws_bytes);
        rc = SPR_ERR_WORKSPACE;
      } else {
        p->ws.bytes = ws_bytes;
      }
    }
    if (rc == SPR_OK && p->geom.six) {
      // pre-twist factors of the six-wave row pass: cot(pi k / nw + pi / 4), k < nw / 2 (ncc_pair6.hip)
      const int half = p->geom.nw / 2;
      std::vector<float> host(static_cast<size_t>(half));
      for (int k = 0; k < half; ++k) {
        const double x = 3.14159265358979323846 * (static_cast<double>(k) / p->geom.nw + 0.25);
        host[k] = static_cast<float>(std::cos(x) / std::sin(x));
      }
      if (hipMalloc(reinterpret_cast<void**>(&p->six_ctab), sizeof(float) * half) != hipSuccess ||
          hipMemcpy(p->six_ctab, host.data(), sizeof(float) * half, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("pre-twist table: hipMalloc / hipMemcpy failed");
        rc = SPR_ERR_HIP;
      }
      p->ws.six_ctab = p->six_ctab;
    }
    if (rc != SPR_OK) { spr_ncc_plan_destroy(p); return rc; }
  }
  { const char* t = std::getenv("SPR_NCC_TEAM"); p->team_mode = t && t[0] == '1'; }
  *plan_out = p;
  return SPR_OK;
}

extern "C" void spr_ncc_plan_destroy(spr_ncc_plan* plan) {
  if (!plan) return;
  if (plan->tw_h) (void)hipFree(plan->tw_h);
  if (plan->tw_w) (void)hipFree(plan->tw_w);
  if (plan->team_sync) (void)hipFree(plan->team_sync);
  if (plan->ws.base) (void)hipFree(plan->ws.base);
  if (plan->six_ctab) (void)hipFree(plan->six_ctab);
  if (plan->mfma_x) (void)hipFree(plan->mfma_x);
  if (plan->done) (void)hipEventDestroy(plan->done);
  delete plan;
}

extern "C" int spr_ncc_plan_method(const spr_ncc_plan* plan) { return plan ? plan->method : SPR_ERR_ARG; }

extern "C" int spr_ncc_plan_fft_size(const spr_ncc_plan* plan, int32_t* rows, int32_t* cols) {
  if (!plan || !rows || !cols) { set_error("spr_ncc_plan_fft_size: null pointer"); return SPR_ERR_ARG; }
  *rows = plan->method == SPR_NCC_FFT ? plan->geom.nh : 0;
  *cols = plan->method == SPR_NCC_FFT ? plan->geom.nw : 0;
  return SPR_OK;
}

extern "C" size_t spr_ncc_query_bytes(const spr_ncc_plan* plan, int64_t n) {
  if (!plan || n < 0) return 0;
  return prepared_query_item_bytes(plan->geom, plan->method) * static_cast<size_t>(n);
}
extern "C" size_t spr_ncc_gallery_bytes(const spr_ncc_plan* plan, int64_t n) {
  if (!plan || n < 0) return 0;
  return prepared_gallery_item_bytes(plan->geom, plan->method) * static_cast<size_t>(n);
}

static int prepare(spr_ncc_plan* plan, bool is_query, const void* maps, int64_t n, void* prepared, spr_stream_t stream,
                   const char* who) {
  if (!plan) { set_error("%s: null plan", who); return SPR_ERR_ARG; }
  if (n < 0 || n > 65535) { set_error("%s: n = %lld outside [0, 65535] (call in chunks)", who, static_cast<long long>(n)); return SPR_ERR_ARG; }
  if (n == 0) return SPR_OK;
  if (!maps || !prepared) { set_error("%s: null pointer", who); return SPR_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (plan->method == SPR_NCC_FFT)
    return launch_prep_fft(plan->geom, is_query, maps, n, prepared, plan->tw_h, plan->tw_w, plan->ws, s);
  if (plan->method == SPR_NCC_MFMA) return launch_prep_mfma(plan->geom, is_query, maps, n, prepared, s);
  return launch_prep_direct(plan->geom, is_query, maps, n, prepared, s);
}

extern "C" int spr_ncc_prepare_queries(spr_ncc_plan* plan, const void* maps, int64_t n, void* prepared,
                                       spr_stream_t stream) {
  return prepare(plan, true, maps, n, prepared, stream, "spr_ncc_prepare_queries");
}
extern "C" int spr_ncc_prepare_gallery(spr_ncc_plan* plan, const void* maps, int64_t n, void* prepared,
                                       spr_stream_t stream) {
  return prepare(plan, false, maps, n, prepared, stream, "spr_ncc_prepare_gallery");
}

extern "C" int spr_ncc_score(spr_ncc_plan* plan, const void* pq, int64_t nq, const void* pg, int64_t ng, float* scores,
                             int64_t ld, int64_t col0, int accumulate_max, spr_stream_t stream) {
  if (!plan) { set_error("spr_ncc_score: null plan"); return SPR_ERR_ARG; }
  if (nq < 0 || ng < 0 || col0 < 0 || ld < col0 + ng) { set_error("spr_ncc_score: bad sizes"); return SPR_ERR_ARG; }
  if (nq == 0 || ng == 0) return SPR_OK;
  if (!pq || !pg || !scores) { set_error("spr_ncc_score: null pointer"); return SPR_ERR_ARG; }
  if (nq > 65535 || ng > (1 << 24)) { set_error("spr_ncc_score: too many items in one call (chunk the gallery)"); return SPR_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = plan_enter(plan, s);
  if (rc != SPR_OK) return rc;
  if (plan->method == SPR_NCC_FFT)
    rc = launch_pair_fft(plan->geom, pq, nq, pg, ng, scores, ld, col0, accumulate_max, nullptr, plan->tw_h,
                         plan->tw_w, plan->team_sync, plan->ws, s);
  else if (plan->method == SPR_NCC_MFMA)
    rc = launch_pair_mfma(plan->geom, pq, nq, pg, ng, scores, ld, col0, accumulate_max, nullptr, plan->mfma_x, s);
  else
    rc = launch_pair_direct(plan->geom, pq, nq, pg, ng, scores, ld, col0, accumulate_max, nullptr, s);
  plan_leave(plan, s);
  return rc;
}

extern "C" int spr_ncc_maps(spr_ncc_plan* plan, const void* pq, const void* pg, float* maps_out, spr_stream_t stream) {
  if (!plan || !pq || !pg || !maps_out) { set_error("spr_ncc_maps: null pointer"); return SPR_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = plan_enter(plan, s);
  if (rc != SPR_OK) return rc;
  if (plan->method == SPR_NCC_FFT)
    rc = launch_pair_fft(plan->geom, pq, 1, pg, 1, nullptr, 1, 0, 0, maps_out, plan->tw_h, plan->tw_w,
                         plan->geom.big ? plan->team_sync : nullptr, plan->ws, s);
  else if (plan->method == SPR_NCC_MFMA)
    rc = launch_pair_mfma(plan->geom, pq, 1, pg, 1, nullptr, 1, 0, 0, maps_out, plan->mfma_x, s);
  else
    rc = launch_pair_direct(plan->geom, pq, 1, pg, 1, nullptr, 1, 0, 0, maps_out, s);
  plan_leave(plan, s);
  return rc;
}
