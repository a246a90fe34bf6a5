// stereo_capi.hip -- C-ABI of include/flame_stereo.h: resident frames (padded image + gradients built on the
// device), the per-launch pose table, and the host/device feature-array entry points of the per-feature epipolar
// inverse-depth update.  All arithmetic of the path runs in stereo_kernels.hip; this file only moves bytes.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstring>
#include <new>
#include <unordered_map>
#include <vector>

#include "flame_stereo.h"
#include "stereo_kernels.h"
#include "roctx_ranges.hpp"

using namespace flame_hip;

namespace {

struct Frame {
  uint8_t* img_pad = nullptr;
  float* gx_pad = nullptr;
  float* gy_pad = nullptr;
};

}  // namespace

struct flame_stereo_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  int last_hip = 0;
  bool have_camera = false;
  StereoCamera cam{};
  std::unordered_map<uint32_t, Frame> frames;
  std::vector<Frame> spare;  // buffers of dropped frames, reused by the next add_frame: hipFree waits for every stream of the device, also
                             // for a solver that runs beside the tracker (tools/frame_loop.py --pipelined: 0.5 ms per frame)
  uint8_t* d_raw = nullptr;  // staging of the unpadded upload
  size_t raw_cap = 0;
  StereoPoseEntry* d_poses = nullptr;
  size_t poses_cap = 0;
  StereoPoseEntry* h_poses = nullptr;  // pinned
  size_t h_poses_cap = 0;
  StereoFeature* d_feats = nullptr;
  size_t feats_cap = 0;
  int* d_stats = nullptr;
  int* h_stats = nullptr;  // pinned, kStatWords ints
  StereoFeature* d_res = nullptr;  // the resident feature set (flame_stereo_set_features)
  size_t res_cap = 0;
  int n_res = 0;
  int lanes_per_feature = 0;  // 0 = by feature count (pick_lanes)
};

namespace {

#define SCHK(ctx, expr)                  \
  do {                                   \
    hipError_t _e = (expr);              \
    if (_e != hipSuccess) {              \
      (ctx)->last_hip = (int)_e;         \
      return _e == hipErrorOutOfMemory ? FLAME_NLTGV2_ERR_OOM : FLAME_NLTGV2_ERR_HIP; \
    }                                    \
  } while (0)

int enter(flame_stereo_ctx* ctx) {
  if (!ctx) return FLAME_NLTGV2_ERR_INVALID_ARG;
  SCHK(ctx, hipSetDevice(ctx->device));
  return 0;
}

void free_frame(Frame& f) {
  if (f.img_pad) (void)hipFree(f.img_pad);
  if (f.gx_pad) (void)hipFree(f.gx_pad);
  if (f.gy_pad) (void)hipFree(f.gy_pad);
  f = Frame{};
}

void drop_all_frames(flame_stereo_ctx* ctx) {
  for (auto& kv : ctx->frames) free_frame(kv.second);
  ctx->frames.clear();
  for (Frame& f : ctx->spare) free_frame(f);  // (the camera changes: another padded size)
  ctx->spare.clear();
}

size_t padded_pixels(const StereoCamera& c) { return (size_t)(c.width + 2 * c.border) * (size_t)(c.height + 2 * c.border); }

template <typename T>
int grow(flame_stereo_ctx* ctx, T** p, size_t* cap, size_t count) {
  if (*cap >= count) return 0;
  if (*p) SCHK(ctx, hipFree(*p));
  *p = nullptr, *cap = 0;
  const size_t want = count + count / 2 + 16;
  SCHK(ctx, hipMalloc((void**)p, want * sizeof(T)));
  *cap = want;
  return 0;
}

// Fills the device pose table and launches; the feature array is already on the device.
// 16 lanes per feature shorten a feature's dependent chain (3.1 k instead of 6.0 k instructions per wave: the epipolar walk
// is split over the row) but replicate its scalar part 16 times: 4 features per wave.  That pays while the chip has room for
// the 16x as many waves -- measured on MI355X (1024 SIMDs): 20.2 vs 24.0 us at 8.4 k features, 32 vs 26 us at 18 k,
// 75 vs 30 us at 57 k.  Above ~2.5 waves per SIMD one lane per feature is faster.
int pick_lanes(const flame_stereo_ctx* ctx, int n_feats) {
  if (ctx->lanes_per_feature) return ctx->lanes_per_feature;
  return n_feats <= 10240 ? 16 : 1;
}

int enqueue_update(flame_stereo_ctx* ctx, const flame_stereo_params* params, uint32_t new_frame_id, uint32_t curr_pf_id,
                   int n_poses, const flame_stereo_pose* poses, int n_feats, StereoFeature* d_feats) {
  if (!params || n_poses < 0 || n_feats < 0 || (n_poses > 0 && !poses)) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (!ctx->have_camera) return FLAME_NLTGV2_ERR_NO_GRAPH;
  auto nf = ctx->frames.find(new_frame_id);
  if (nf == ctx->frames.end()) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (ctx->h_poses_cap < (size_t)n_poses) {
    if (ctx->h_poses) SCHK(ctx, hipHostFree(ctx->h_poses));
    ctx->h_poses = nullptr, ctx->h_poses_cap = 0;
    const size_t want = (size_t)n_poses * 2 + 16;
    SCHK(ctx, hipHostMalloc((void**)&ctx->h_poses, want * sizeof(StereoPoseEntry), hipHostMallocDefault));
    ctx->h_poses_cap = want;
  }
  if (int rc = grow(ctx, &ctx->d_poses, &ctx->poses_cap, (size_t)n_poses + 1)) return rc;
  // the pinned table may still be in flight from the previous launch on this stream
  SCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < n_poses; ++k) {
    auto it = ctx->frames.find(poses[k].frame_id);
    if (it == ctx->frames.end()) return FLAME_NLTGV2_ERR_INVALID_ARG;
    StereoPoseEntry& e = ctx->h_poses[k];
    std::memset(&e, 0, sizeof e);
    e.frame_id = poses[k].frame_id;
    e.img_pad = it->second.img_pad;
    fill_pose_entry(&e, ctx->cam, poses[k]);
  }
  if (n_poses > 0)
    SCHK(ctx, hipMemcpyAsync(ctx->d_poses, ctx->h_poses, (size_t)n_poses * sizeof(StereoPoseEntry), hipMemcpyHostToDevice,
                             ctx->stream));
  std::memset(ctx->h_stats, 0, kStatWords * sizeof(int));
  ctx->h_stats[kStatAssert] = ctx->h_stats[kStatBadFrame] = INT_MAX;
  SCHK(ctx, hipMemcpyAsync(ctx->d_stats, ctx->h_stats, kStatWords * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  SCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  SCHK(ctx, launch_update_feature_idepths(*params, ctx->cam, n_poses, ctx->d_poses, nf->second.img_pad, nf->second.gx_pad,
                                          nf->second.gy_pad, curr_pf_id, n_feats, d_feats, ctx->d_stats, pick_lanes(ctx, n_feats),
                                          ctx->stream));
  SCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->timed = true;
  SCHK(ctx, hipMemcpyAsync(ctx->h_stats, ctx->d_stats, kStatWords * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  return 0;
}

int report(flame_stereo_ctx* ctx, flame_stereo_stats* stats) {
  int* s = ctx->h_stats;
  for (int c = 0; c < 6; ++c) {  // the counters were accumulated in kStatSlots copies (stereo_kernels.h)
    int sum = 0;
    for (int k = 0; k < kStatSlots; ++k) sum += s[kStatCount + k * kStatSlotStride + c];
    s[c] = sum;
  }
  stats->num_idepth_updates = s[0];
  stats->num_fail_max_var = s[1];
  stats->num_fail_max_dropouts = s[2];
  stats->num_fail_ref_patch_grad = s[3];
  stats->num_fail_ambiguous_match = s[4];
  stats->num_fail_max_cost = s[5];
  stats->success = s[0] > 0;
  stats->error_feature = -1;
  if (s[kStatBadFrame] != INT_MAX) {  // pfs.at() would throw
    stats->error_feature = s[kStatBadFrame];
    return FLAME_NLTGV2_ERR_INVALID_ARG;
  }
  if (s[kStatAssert] != INT_MAX) {
    stats->error_feature = s[kStatAssert];
    return FLAME_NLTGV2_ERR_ASSERT;
  }
  return 0;
}

}  // namespace

extern "C" {

void flame_stereo_default_params(flame_stereo_params* p) {
  if (!p) return;
  p->min_baseline = 0.01f;
  p->do_letterbox = 0;
  p->rescale_factor_min = 0.7f;
  p->rescale_factor_max = 1.4f;
  p->idepth_var_max = 0.5f * 0.5f;
  p->max_dropouts = 5;
  p->outlier_sigma_thresh = 3.0f;
  p->do_meas_fusion = 1;
  p->win_size = 5;
  p->search_sigma = 2.0f;
  p->min_grad_mag = 5.0f;
  p->idepth_min = 1e-3f;
  p->idepth_max = 2.0f;
  p->epilength_min = 3.0f;
  p->epilength_max = 32.0f;
  p->process_var_factor = 1.01f;
  p->process_fail_var_factor = 1.1f;
  p->max_cost = 1300.0f;
  p->do_subpixel = 1;
  p->sample_dist = 1.0f;
  p->second_best_factor = 1.5f;
  p->z_win_size = 5;
  p->pixel_var = 16.0f;
  p->epipolar_line_var = 1.0f;
}

int flame_stereo_create(flame_stereo_ctx** out, int device) {
  if (!out) return FLAME_NLTGV2_ERR_INVALID_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return FLAME_NLTGV2_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return FLAME_NLTGV2_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return FLAME_NLTGV2_ERR_NO_DEVICE;
  flame_stereo_ctx* ctx = new (std::nothrow) flame_stereo_ctx();
  if (!ctx) return FLAME_NLTGV2_ERR_OOM;
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
      hipMalloc((void**)&ctx->d_stats, kStatWords * sizeof(int)) != hipSuccess ||
      hipHostMalloc((void**)&ctx->h_stats, kStatWords * sizeof(int), hipHostMallocDefault) != hipSuccess) {
    flame_stereo_destroy(ctx);
    return FLAME_NLTGV2_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  *out = ctx;
  return 0;
}

void flame_stereo_destroy(flame_stereo_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  drop_all_frames(ctx);
  if (ctx->d_raw) (void)hipFree(ctx->d_raw);
  if (ctx->d_poses) (void)hipFree(ctx->d_poses);
  if (ctx->h_poses) (void)hipHostFree(ctx->h_poses);
  if (ctx->d_feats) (void)hipFree(ctx->d_feats);
  if (ctx->d_res) (void)hipFree(ctx->d_res);
  if (ctx->d_stats) (void)hipFree(ctx->d_stats);
  if (ctx->h_stats) (void)hipHostFree(ctx->h_stats);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int flame_stereo_set_stream(flame_stereo_ctx* ctx, void* hip_stream) {
  if (int rc = enter(ctx)) return rc;
  SCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return 0;
}

int flame_stereo_set_camera(flame_stereo_ctx* ctx, const float K[9], const float Kinv[9], int width, int height,
                            int border) {
  if (int rc = enter(ctx)) return rc;
  if (!K || !Kinv || width < 2 || height < 2 || border < 0 || width > 16384 || height > 16384 || border > 64)
    return FLAME_NLTGV2_ERR_INVALID_ARG;
  SCHK(ctx, hipStreamSynchronize(ctx->stream));
  drop_all_frames(ctx);
  std::memcpy(ctx->cam.K, K, sizeof ctx->cam.K);
  std::memcpy(ctx->cam.Kinv, Kinv, sizeof ctx->cam.Kinv);
  ctx->cam.width = width, ctx->cam.height = height, ctx->cam.border = border;
  ctx->have_camera = true;
  return 0;
}

int flame_stereo_add_frame(flame_stereo_ctx* ctx, uint32_t frame_id, const uint8_t* img, int row_stride_bytes) {
  flame_hip::RoctxRange roctx_range_("flame_stereo_add_frame");
  if (int rc = enter(ctx)) return rc;
  if (!ctx->have_camera) return FLAME_NLTGV2_ERR_NO_GRAPH;
  const int w = ctx->cam.width, h = ctx->cam.height;
  if (!img || row_stride_bytes < w) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (int rc = grow(ctx, &ctx->d_raw, &ctx->raw_cap, (size_t)w * h)) return rc;
  Frame& f = ctx->frames[frame_id];
  const size_t px = padded_pixels(ctx->cam);
  if (!f.img_pad && !ctx->spare.empty()) {
    f = ctx->spare.back();
    ctx->spare.pop_back();
  }
  if (!f.img_pad) {
    hipError_t e = hipMalloc((void**)&f.img_pad, px);
    if (e == hipSuccess) e = hipMalloc((void**)&f.gx_pad, px * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&f.gy_pad, px * sizeof(float));
    if (e != hipSuccess) {
      free_frame(f);
      ctx->frames.erase(frame_id);
      ctx->last_hip = (int)e;
      return e == hipErrorOutOfMemory ? FLAME_NLTGV2_ERR_OOM : FLAME_NLTGV2_ERR_HIP;
    }
  }
  SCHK(ctx, hipMemcpy2DAsync(ctx->d_raw, (size_t)w, img, (size_t)row_stride_bytes, (size_t)w, (size_t)h,
                             hipMemcpyHostToDevice, ctx->stream));
  SCHK(ctx, launch_frame_pad_gradient(ctx->d_raw, w, h, ctx->cam.border, f.img_pad, f.gx_pad, f.gy_pad, ctx->stream));
  // the staging buffer is reused by the next add_frame and `img` is pageable host memory
  SCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int flame_stereo_drop_frame(flame_stereo_ctx* ctx, uint32_t frame_id) {
  if (int rc = enter(ctx)) return rc;
  auto it = ctx->frames.find(frame_id);
  if (it == ctx->frames.end()) return FLAME_NLTGV2_ERR_INVALID_ARG;
  SCHK(ctx, hipStreamSynchronize(ctx->stream));
  constexpr size_t kSpareFrames = 4;
  if (ctx->spare.size() < kSpareFrames) ctx->spare.push_back(it->second); else free_frame(it->second);
  ctx->frames.erase(it);
  return 0;
}

int flame_stereo_frame_count(const flame_stereo_ctx* ctx) { return ctx ? (int)ctx->frames.size() : 0; }

int flame_stereo_download_frame(flame_stereo_ctx* ctx, uint32_t frame_id, uint8_t* img_pad, float* gradx_pad,
                                float* grady_pad) {
  if (int rc = enter(ctx)) return rc;
  auto it = ctx->frames.find(frame_id);
  if (it == ctx->frames.end()) return FLAME_NLTGV2_ERR_INVALID_ARG;
  const size_t px = padded_pixels(ctx->cam);
  SCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (img_pad) SCHK(ctx, hipMemcpy(img_pad, it->second.img_pad, px, hipMemcpyDeviceToHost));
  if (gradx_pad) SCHK(ctx, hipMemcpy(gradx_pad, it->second.gx_pad, px * sizeof(float), hipMemcpyDeviceToHost));
  if (grady_pad) SCHK(ctx, hipMemcpy(grady_pad, it->second.gy_pad, px * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int flame_stereo_update_feature_idepths(flame_stereo_ctx* ctx, const flame_stereo_params* params, uint32_t new_frame_id,
                                        uint32_t curr_pf_id, int n_poses, const flame_stereo_pose* poses, int n_feats,
                                        flame_stereo_feature* feats, flame_stereo_stats* stats) {
  flame_hip::RoctxRange roctx_range_("flame_stereo_update_feature_idepths");
  if (int rc = enter(ctx)) return rc;
  if (!stats || n_feats < 0 || (n_feats > 0 && !feats)) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (int rc = grow(ctx, &ctx->d_feats, &ctx->feats_cap, (size_t)n_feats + 1)) return rc;
  if (n_feats > 0)
    SCHK(ctx, hipMemcpyAsync(ctx->d_feats, feats, (size_t)n_feats * sizeof(StereoFeature), hipMemcpyHostToDevice, ctx->stream));
  if (int rc = enqueue_update(ctx, params, new_frame_id, curr_pf_id, n_poses, poses, n_feats, ctx->d_feats)) return rc;
  SCHK(ctx, hipStreamSynchronize(ctx->stream));
  const int rc = report(ctx, stats);
  if (rc == 0 && n_feats > 0)
    SCHK(ctx, hipMemcpy(feats, ctx->d_feats, (size_t)n_feats * sizeof(StereoFeature), hipMemcpyDeviceToHost));
  return rc;
}

int flame_stereo_update_feature_idepths_device(flame_stereo_ctx* ctx, const flame_stereo_params* params,
                                               uint32_t new_frame_id, uint32_t curr_pf_id, int n_poses,
                                               const flame_stereo_pose* poses, int n_feats, void* feats_device,
                                               flame_stereo_stats* stats) {
  flame_hip::RoctxRange roctx_range_("flame_stereo_update_feature_idepths_device");
  if (int rc = enter(ctx)) return rc;
  if (n_feats < 0 || (n_feats > 0 && !feats_device)) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (int rc = enqueue_update(ctx, params, new_frame_id, curr_pf_id, n_poses, poses, n_feats, (StereoFeature*)feats_device))
    return rc;
  if (!stats) return 0;
  SCHK(ctx, hipStreamSynchronize(ctx->stream));
  return report(ctx, stats);
}

int flame_stereo_set_option(flame_stereo_ctx* ctx, int option, int value) {
  if (!ctx) return FLAME_NLTGV2_ERR_INVALID_ARG;
  switch (option) {
    case FLAME_STEREO_OPT_LANES_PER_FEATURE:
      if (value != 0 && value != 1 && value != 16) return FLAME_NLTGV2_ERR_INVALID_ARG;
      ctx->lanes_per_feature = value;
      return 0;
    default:
      return FLAME_NLTGV2_ERR_INVALID_ARG;
  }
}

int flame_stereo_set_features(flame_stereo_ctx* ctx, int n_feats, const flame_stereo_feature* feats) {
  flame_hip::RoctxRange roctx_range_("flame_stereo_set_features");
  if (int rc = enter(ctx)) return rc;
  if (n_feats < 0 || (n_feats > 0 && !feats)) return FLAME_NLTGV2_ERR_INVALID_ARG;
  SCHK(ctx, hipStreamSynchronize(ctx->stream));  // (the array may be reallocated)
  if (int rc = grow(ctx, &ctx->d_res, &ctx->res_cap, (size_t)n_feats + 1)) return rc;
  if (n_feats > 0) {
    SCHK(ctx, hipMemcpyAsync(ctx->d_res, feats, (size_t)n_feats * sizeof(StereoFeature), hipMemcpyHostToDevice, ctx->stream));
    SCHK(ctx, hipStreamSynchronize(ctx->stream));  // `feats` is the caller's (pageable) memory
  }
  ctx->n_res = n_feats;
  return 0;
}

int flame_stereo_get_features(flame_stereo_ctx* ctx, int max_feats, flame_stereo_feature* feats, int* n_feats) {
  flame_hip::RoctxRange roctx_range_("flame_stereo_get_features");
  if (int rc = enter(ctx)) return rc;
  if (n_feats) *n_feats = ctx->n_res;
  if (!feats) return 0;
  if (max_feats < ctx->n_res) return FLAME_NLTGV2_ERR_INVALID_ARG;
  SCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->n_res > 0)
    SCHK(ctx, hipMemcpy(feats, ctx->d_res, (size_t)ctx->n_res * sizeof(StereoFeature), hipMemcpyDeviceToHost));
  return 0;
}

int flame_stereo_features_device(flame_stereo_ctx* ctx, void** feats_device, int* n_feats) {
  if (!ctx) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (feats_device) *feats_device = ctx->d_res;
  if (n_feats) *n_feats = ctx->n_res;
  return 0;
}

int flame_stereo_update_resident(flame_stereo_ctx* ctx, const flame_stereo_params* params, uint32_t new_frame_id,
                                 uint32_t curr_pf_id, int n_poses, const flame_stereo_pose* poses, flame_stereo_stats* stats) {
  flame_hip::RoctxRange roctx_range_("flame_stereo_update_resident");
  if (int rc = enter(ctx)) return rc;
  if (int rc = enqueue_update(ctx, params, new_frame_id, curr_pf_id, n_poses, poses, ctx->n_res, ctx->d_res)) return rc;
  if (!stats) return 0;
  SCHK(ctx, hipStreamSynchronize(ctx->stream));
  return report(ctx, stats);
}

float flame_stereo_last_kernel_ms(flame_stereo_ctx* ctx) {
  if (!ctx || !ctx->timed) return -1.0f;
  if (hipSetDevice(ctx->device) != hipSuccess || hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0f;
  float ms = -1.0f;
  if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0f;
  return ms;
}

int flame_stereo_last_hip_error(const flame_stereo_ctx* ctx) { return ctx ? ctx->last_hip : 0; }

}  // extern "C"
