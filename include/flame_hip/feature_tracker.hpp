// flame_hip/feature_tracker.hpp -- header-only C++11 binding of include/flame_stereo.h, shaped after the call the
// reference makes per image:
//
//   reference  bool Flame::updateFeatureIDepths(params, K, Kinv, pfs, fnew, curr_pf, &feats, &stats, &debug_img)
//              (/root/reference/src/flame/flame.cc:1280-1288, call site flame.cc:265-267)
//   here       flame_hip::FeatureTracker tracker(K, Kinv, width, height);      // next to K_, Kinv_ (flame.h:520-523)
//              tracker.addFrame(frame->id, frame->img[0].data, frame->img[0].step);   // where Frame::create ran
//              bool ok = tracker.updateFeatureIDepths(params, pfs, *fnew_, *curr_pf_, &feats_, &stats);
//
// Works with the reference's own types through templates (no Eigen/Sophus/OpenCV headers are needed here):
//   Matrix3    anything with operator()(row, col)                      (Eigen::Matrix3f)
//   SE3        .inverse(), operator*, .unit_quaternion().{w,x,y,z}(), .translation()(i)   (Sophus::SE3f)
//   Frame      .id, .pose                                               (utils::Frame, frame.h)
//   FrameMap   iterable of (id, pointer-to-Frame) pairs                 (FrameIDToFrame, flame.h:72)
//   Feature    layout of FeatureWithIDepth (flame.h:88-99); checked with static_asserts by adoptFeatures()
#ifndef FLAME_HIP_FEATURE_TRACKER_HPP_
#define FLAME_HIP_FEATURE_TRACKER_HPP_

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "flame_stereo.h"

namespace flame_hip {

struct StereoError : std::runtime_error {
  int status;
  int feature;  // lowest failing feature index, or -1
  StereoError(int s, int f, const std::string& what)
      : std::runtime_error(what + ": " + flame_nltgv2_status_string(s)), status(s), feature(f) {}
};

// flame::Params -> flame_stereo_params (the members updateFeatureIDepths / trackFeature read, params.h:36-126).
template <class FlameParams>
inline flame_stereo_params toStereoParams(const FlameParams& p) {
  flame_stereo_params o;
  flame_stereo_default_params(&o);
  o.min_baseline = p.min_baseline;
  o.do_letterbox = p.do_letterbox ? 1 : 0;
  o.rescale_factor_min = p.rescale_factor_min;
  o.rescale_factor_max = p.rescale_factor_max;
  o.idepth_var_max = p.idepth_var_max;
  o.max_dropouts = p.max_dropouts;
  o.outlier_sigma_thresh = p.outlier_sigma_thresh;
  o.do_meas_fusion = p.do_meas_fusion ? 1 : 0;
  o.win_size = p.fparams.win_size;
  o.search_sigma = p.fparams.search_sigma;
  o.min_grad_mag = p.fparams.min_grad_mag;
  o.idepth_min = p.fparams.idepth_min;
  o.idepth_max = p.fparams.idepth_max;
  o.epilength_min = p.fparams.epilength_min;
  o.epilength_max = p.fparams.epilength_max;
  o.process_var_factor = p.fparams.process_var_factor;
  o.process_fail_var_factor = p.fparams.process_fail_var_factor;
  o.max_cost = p.fparams.sparams.max_cost;
  o.do_subpixel = p.fparams.sparams.do_subpixel ? 1 : 0;
  o.sample_dist = p.fparams.sparams.sample_dist;
  o.second_best_factor = p.fparams.sparams.second_best_factor;
  o.z_win_size = p.zparams.win_size;
  o.pixel_var = p.zparams.pixel_var;
  o.epipolar_line_var = p.zparams.epipolar_line_var;
  return o;
}

// (quaternion, translation) of an SE3 into the C arrays of a flame_stereo_pose.
template <class SE3>
inline void toQuatTrans(const SE3& T, float q[4], float t[3]) {
  q[0] = T.unit_quaternion().w(), q[1] = T.unit_quaternion().x(), q[2] = T.unit_quaternion().y(),
  q[3] = T.unit_quaternion().z();
  t[0] = T.translation()(0), t[1] = T.translation()(1), t[2] = T.translation()(2);
}

// One pose table entry for pose-frame `pf`: the two relative poses the reference forms per feature
// (flame.cc:1315 and flame.cc:1614), formed once per pose-frame here.
template <class Frame>
inline flame_stereo_pose makePose(const Frame& pf, const Frame& fnew, const Frame& curr_pf) {
  flame_stereo_pose p;
  std::memset(&p, 0, sizeof p);
  p.frame_id = pf.id;
  toQuatTrans(fnew.pose.inverse() * pf.pose, p.q_ref_to_new, p.t_ref_to_new);
  toQuatTrans(curr_pf.pose.inverse() * pf.pose, p.q_ref_to_pf, p.t_ref_to_pf);
  return p;
}

// Reinterprets an array of the reference's FeatureWithIDepth as flame_stereo_feature after checking the layout.
template <class Feature>
inline flame_stereo_feature* adoptFeatures(Feature* feats) {
  static_assert(sizeof(Feature) == sizeof(flame_stereo_feature), "FeatureWithIDepth layout changed");
  static_assert(offsetof(Feature, frame_id) == offsetof(flame_stereo_feature, frame_id), "frame_id");
  static_assert(offsetof(Feature, xy) == offsetof(flame_stereo_feature, x), "xy");
  static_assert(offsetof(Feature, idepth_mu) == offsetof(flame_stereo_feature, idepth_mu), "idepth_mu");
  static_assert(offsetof(Feature, idepth_var) == offsetof(flame_stereo_feature, idepth_var), "idepth_var");
  static_assert(offsetof(Feature, valid) == offsetof(flame_stereo_feature, valid), "valid");
  static_assert(offsetof(Feature, num_updates) == offsetof(flame_stereo_feature, num_updates), "num_updates");
  static_assert(offsetof(Feature, num_dropouts) == offsetof(flame_stereo_feature, num_dropouts), "num_dropouts");
  static_assert(offsetof(Feature, search_status) == offsetof(flame_stereo_feature, search_status), "search_status");
  return reinterpret_cast<flame_stereo_feature*>(feats);
}

class FeatureTracker {
 public:
  template <class Matrix3>
  FeatureTracker(const Matrix3& K, const Matrix3& Kinv, int width, int height, int border = 5, int device = 0)
      : ctx_(nullptr) {
    float k[9], ki[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) k[3 * r + c] = K(r, c), ki[3 * r + c] = Kinv(r, c);
    check(flame_stereo_create(&ctx_, device), -1, "flame_stereo_create");
    const int rc = flame_stereo_set_camera(ctx_, k, ki, width, height, border);
    if (rc != 0) {
      flame_stereo_destroy(ctx_);
      ctx_ = nullptr;
      throw StereoError(rc, -1, "flame_stereo_set_camera");
    }
  }
  ~FeatureTracker() { flame_stereo_destroy(ctx_); }
  FeatureTracker(const FeatureTracker&) = delete;
  FeatureTracker& operator=(const FeatureTracker&) = delete;

  // utils::Frame::create, level 0 (frame.cc:33-71), on the device.  Call it where the reference creates fnew_
  // (flame.cc:150); a frame that becomes a pose-frame simply stays resident.
  void addFrame(uint32_t frame_id, const uint8_t* img, int row_stride_bytes) {
    check(flame_stereo_add_frame(ctx_, frame_id, img, row_stride_bytes), -1, "flame_stereo_add_frame");
  }
  void dropFrame(uint32_t frame_id) { check(flame_stereo_drop_frame(ctx_, frame_id), -1, "flame_stereo_drop_frame"); }
  int frameCount() const { return flame_stereo_frame_count(ctx_); }

  // == Flame::updateFeatureIDepths (flame.cc:1280-1536).  Returns what the reference returns; `stats` (optional)
  // receives the counters the reference hands to StatsTracker (flame.cc:1497-1502).
  bool updateFeatureIDepths(const flame_stereo_params& params, uint32_t new_frame_id, uint32_t curr_pf_id,
                            const std::vector<flame_stereo_pose>& poses, flame_stereo_feature* feats, int n_feats,
                            flame_stereo_stats* stats = nullptr) {
    flame_stereo_stats local;
    flame_stereo_stats* st = stats ? stats : &local;
    const int rc = flame_stereo_update_feature_idepths(ctx_, &params, new_frame_id, curr_pf_id, (int)poses.size(),
                                                       poses.empty() ? nullptr : poses.data(), n_feats, feats, st);
    check(rc, st->error_feature, "flame_stereo_update_feature_idepths");
    return st->success != 0;
  }

  // The reference's argument list: pfs (id -> shared_ptr<Frame>), fnew, curr_pf, std::vector<FeatureWithIDepth>.
  template <class FlameParams, class FrameMap, class Frame, class Feature>
  bool updateFeatureIDepths(const FlameParams& params, const FrameMap& pfs, const Frame& fnew, const Frame& curr_pf,
                            std::vector<Feature>* feats, flame_stereo_stats* stats = nullptr) {
    std::vector<flame_stereo_pose> poses;
    for (typename FrameMap::const_iterator it = pfs.begin(); it != pfs.end(); ++it)
      poses.push_back(makePose(*it->second, fnew, curr_pf));
    return updateFeatureIDepths(toStereoParams(params), fnew.id, curr_pf.id, poses,
                                feats->empty() ? nullptr : adoptFeatures(feats->data()), (int)feats->size(), stats);
  }

  flame_stereo_ctx* handle() const { return ctx_; }

 private:
  static void check(int rc, int feature, const char* what) {
    if (rc != 0) throw StereoError(rc, feature, what);
  }
  flame_stereo_ctx* ctx_;
};

}  // namespace flame_hip
#endif  // FLAME_HIP_FEATURE_TRACKER_HPP_
