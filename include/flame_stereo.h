/*
 * flame_stereo.h -- C-ABI of the per-feature epipolar inverse-depth update of robustrobotics/flame on MI355X
 * (gfx950); part of libflame_nltgv2_hip.so.  SURVEY.md section 8(f) rank 4: the per-frame loop that produces the
 * regularizer's data terms.
 *
 * Replaces, for the level-0 images the path reads:
 *   Flame::updateFeatureIDepths      /root/reference/src/flame/flame.cc:1280-1536  (the `omp parallel for` over features)
 *   Flame::trackFeature              flame.cc:1538-1752
 *   stereo::EpipolarGeometry<float>  src/flame/stereo/epipolar_geometry.h
 *   stereo::inverse_depth_filter::{predict,getSearchRegion,search,update}   src/flame/stereo/inverse_depth_filter.cc
 *   stereo::line_stereo::match       src/flame/stereo/line_stereo.h:73-385
 *   stereo::InverseDepthMeasModel::idepth   src/flame/stereo/inverse_depth_meas_model.cc:48-154
 *   utils::Frame::create (level 0: padded image + padded central gradients)   src/flame/utils/frame.cc:33-71
 *
 * The boundary sits where the reference hands Eigen/Sophus values to its stereo code: the caller keeps the
 * pose algebra (Sophus::SE3f products, flame.cc:1315-1316, 1614) and passes one (quaternion, translation) pair
 * per pose-frame; everything from EpipolarGeometry::loadGeometry down runs on the GPU, a 16-lane row per feature.
 * Results are bit-identical to the reference's scalar float code (same expression order, no FMA contraction).
 * Debug drawing (params.debug_draw_matches) and the stderr diagnostics are not part of the path.
 *
 * Status codes are flame_nltgv2_status (flame_nltgv2.h).  Where the reference would FLAME_ASSERT -> exit(1)
 * (negative inverse depth into project(), a sample outside the padded image, ...), the call returns
 * FLAME_NLTGV2_ERR_ASSERT and stats.error_feature names the lowest such feature index; the feature array is
 * then unspecified.  There is no CPU fallback.
 */
#ifndef FLAME_STEREO_H_
#define FLAME_STEREO_H_

#include <stdint.h>

#include "flame_nltgv2.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct flame_stereo_ctx flame_stereo_ctx;

/* The members of flame::Params this path reads (params.h), with the reference's defaults
 * (flame_stereo_default_params). */
typedef struct flame_stereo_params {
  float min_baseline;            /* Params::min_baseline            0.01  (params.h:72) */
  int32_t do_letterbox;          /* Params::do_letterbox            0     (params.h:47) */
  float rescale_factor_min;      /* Params::rescale_factor_min      0.7   (params.h:64) */
  float rescale_factor_max;      /* Params::rescale_factor_max      1.4   (params.h:65) */
  float idepth_var_max;          /* Params::idepth_var_max          0.25  (params.h:68) */
  int32_t max_dropouts;          /* Params::max_dropouts            5     (params.h:69) */
  float outlier_sigma_thresh;    /* Params::outlier_sigma_thresh    3     (params.h:70) */
  int32_t do_meas_fusion;        /* Params::do_meas_fusion          1     (params.h:73) */
  /* Params::fparams -- inverse_depth_filter::Params (inverse_depth_filter.h:50-71) */
  int32_t win_size;              /* 5 (the only supported value, as in the reference: inverse_depth_filter.cc:195) */
  float search_sigma;            /* 2 */
  float min_grad_mag;            /* 5 */
  float idepth_min;              /* 1e-3 */
  float idepth_max;              /* 2 */
  float epilength_min;           /* 3 */
  float epilength_max;           /* 32 */
  float process_var_factor;      /* 1.01 */
  float process_fail_var_factor; /* 1.1 */
  /* Params::fparams.sparams -- line_stereo::Params (line_stereo.h:47-60) */
  float max_cost;                /* 1300 */
  int32_t do_subpixel;           /* 1 */
  float sample_dist;             /* 1 */
  float second_best_factor;      /* 1.5 */
  /* Params::zparams -- InverseDepthMeasModel::Params (inverse_depth_meas_model.h:43-51) */
  int32_t z_win_size;            /* 5 */
  float pixel_var;               /* 16 */
  float epipolar_line_var;       /* 1 */
} flame_stereo_params;
void flame_stereo_default_params(flame_stereo_params* p);

/* == struct FeatureWithIDepth (flame.h:88-99); 40 bytes. */
typedef struct flame_stereo_feature {
  uint32_t id;
  uint32_t frame_id;     /* the pose-frame the feature is anchored in; rewritten when the feature is moved (flame.cc:1634) */
  float x, y;            /* xy in that frame (unpadded pixel coordinates) */
  float idepth_mu;
  float idepth_var;
  uint8_t valid;
  uint8_t reserved_[3];
  uint32_t num_updates;
  uint32_t num_dropouts;
  int32_t search_status; /* inverse_depth_filter::Status: 0 SUCCESS, 1 FAIL_REF_PATCH_GRADIENT, 2 FAIL_AMBIGUOUS_MATCH,
                            3 FAIL_MAX_COST (inverse_depth_filter.h:39-44) */
} flame_stereo_feature;

/* One entry per pose-frame that features may refer to (the reference's `pfs` map, flame.h:526).
 * q = (w, x, y, z). */
typedef struct flame_stereo_pose {
  uint32_t frame_id;
  float q_ref_to_new[4], t_ref_to_new[3]; /* fnew.pose.inverse() * pf.pose        (flame.cc:1315) */
  float q_ref_to_pf[4], t_ref_to_pf[3];   /* curr_pf.pose.inverse() * pf.pose     (flame.cc:1614), used when a feature
                                             is moved to the newest pose-frame */
} flame_stereo_pose;

/* The counters updateFeatureIDepths reports through StatsTracker (flame.cc:1497-1502) + its return value. */
typedef struct flame_stereo_stats {
  int32_t num_idepth_updates;
  int32_t num_fail_max_var;
  int32_t num_fail_max_dropouts;
  int32_t num_fail_ref_patch_grad;
  int32_t num_fail_ambiguous_match;
  int32_t num_fail_max_cost;
  int32_t success;       /* the bool updateFeatureIDepths returns (any feature updated) */
  int32_t error_feature; /* -1, or the lowest feature index that hit a reference assert / an unknown frame id */
} flame_stereo_stats;

int flame_stereo_create(flame_stereo_ctx** out, int device);
void flame_stereo_destroy(flame_stereo_ctx* ctx);
int flame_stereo_set_stream(flame_stereo_ctx* ctx, void* hip_stream);

/* Camera and image geometry (Flame::Flame, flame.cc:48-60: K_, Kinv_, width_, height_).  K, Kinv row-major.
 * `border` is the padding Frame::create gets (flame.cc:149-150: params.fparams.win_size = 5).  Drops all frames. */
int flame_stereo_set_camera(flame_stereo_ctx* ctx, const float K[9], const float Kinv[9], int width, int height,
                            int border);

/* utils::Frame::create, level 0 (frame.cc:33-71): uploads the width x height 8-bit image and builds, on the
 * device, img_pad (cv::BORDER_REFLECT_101) and gradx_pad / grady_pad (getCentralGradient, image_utils.h:425-470,
 * then cv::BORDER_CONSTANT 0).  The frame stays resident until dropped (a pose-frame is read by every later
 * frame); adding an existing id replaces it. */
int flame_stereo_add_frame(flame_stereo_ctx* ctx, uint32_t frame_id, const uint8_t* img, int row_stride_bytes);
int flame_stereo_drop_frame(flame_stereo_ctx* ctx, uint32_t frame_id);
int flame_stereo_frame_count(const flame_stereo_ctx* ctx);
/* Copies a resident frame's derived images back ((height + 2 border) x (width + 2 border) each; any may be NULL). */
int flame_stereo_download_frame(flame_stereo_ctx* ctx, uint32_t frame_id, uint8_t* img_pad, float* gradx_pad,
                                float* grady_pad);

/* Flame::updateFeatureIDepths (flame.cc:1280-1536).  `new_frame_id` is fnew, `curr_pf_id` is curr_pf.id; both and
 * every pose's frame must be resident.  `feats` (host) is updated in place. */
int flame_stereo_update_feature_idepths(flame_stereo_ctx* ctx, const flame_stereo_params* params, uint32_t new_frame_id,
                                        uint32_t curr_pf_id, int n_poses, const flame_stereo_pose* poses, int n_feats,
                                        flame_stereo_feature* feats, flame_stereo_stats* stats);
/* Same on a feature array that lives in device memory (n_feats * 40 bytes); enqueues on the context's stream and,
 * unless `stats` is NULL, waits and reports. */
int flame_stereo_update_feature_idepths_device(flame_stereo_ctx* ctx, const flame_stereo_params* params,
                                               uint32_t new_frame_id, uint32_t curr_pf_id, int n_poses,
                                               const flame_stereo_pose* poses, int n_feats, void* feats_device,
                                               flame_stereo_stats* stats);
/* The resident feature set -- the default way to run the path: the features live in device memory from detection to
 * removal (as Flame::feats_ lives in the Flame object, flame.h:529), every frame runs the update on them in place, and the
 * host reads them back only when it needs them (new detections, the data terms of the graph).
 *   set_features     replaces the resident set (host array of n_feats records);
 *   update_resident  Flame::updateFeatureIDepths on it; `stats` NULL = enqueue only (results are ordered on the stream);
 *   get_features     copies it back (feats may be NULL to query the count);
 *   features_device  its device address and count (valid until the next set_features). */
int flame_stereo_set_features(flame_stereo_ctx* ctx, int n_feats, const flame_stereo_feature* feats);
int flame_stereo_update_resident(flame_stereo_ctx* ctx, const flame_stereo_params* params, uint32_t new_frame_id,
                                 uint32_t curr_pf_id, int n_poses, const flame_stereo_pose* poses, flame_stereo_stats* stats);
int flame_stereo_get_features(flame_stereo_ctx* ctx, int max_feats, flame_stereo_feature* feats, int* n_feats);
int flame_stereo_features_device(flame_stereo_ctx* ctx, void** feats_device, int* n_feats);

/* Options.  LANES_PER_FEATURE: 16 (a 16-lane row shares a feature and splits the epipolar walk), 1 (one lane walks the
 * whole per-feature body) or 0 (default: 16 up to 10240 features, 1 above -- whichever is faster on MI355X); same results
 * bit for bit. */
enum { FLAME_STEREO_OPT_LANES_PER_FEATURE = 1 };
int flame_stereo_set_option(flame_stereo_ctx* ctx, int option, int value);

/* Device time of the last update kernel in milliseconds (HIP events on the context's stream); < 0 if none. */
float flame_stereo_last_kernel_ms(flame_stereo_ctx* ctx);
int flame_stereo_last_hip_error(const flame_stereo_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* FLAME_STEREO_H_ */
