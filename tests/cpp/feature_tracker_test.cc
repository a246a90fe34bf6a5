// tests/cpp/feature_tracker_test.cc -- include/flame_hip/feature_tracker.hpp used with look-alikes of the
// reference's own types (Params with fparams/zparams, a Frame with id + SE3 pose, a map of shared frames,
// FeatureWithIDepth), the way Flame::update() would call it; results are compared with the CPU checker
// (oracle/stereo_oracle.c, linked as test infrastructure).  Build+run: tests/test_cpp_facade.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include "flame_hip/feature_tracker.hpp"

// ---- look-alikes of the reference types the template binding touches (test-only) -----------------------------
struct Quat {
  float w_, x_, y_, z_;
  float w() const { return w_; }
  float x() const { return x_; }
  float y() const { return y_; }
  float z() const { return z_; }
};
struct Vec3 {
  float v[3];
  float operator()(int i) const { return v[i]; }
};
static Vec3 rotate(const Quat& q, const Vec3& p) {
  const double w = q.w_, x = q.x_, y = q.y_, z = q.z_;
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                       1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                       1 - 2 * (x * x + y * y)};
  Vec3 o;
  for (int i = 0; i < 3; ++i) o.v[i] = (float)(R[3 * i] * p.v[0] + R[3 * i + 1] * p.v[1] + R[3 * i + 2] * p.v[2]);
  return o;
}
struct SE3 {  // Sophus::SE3f look-alike
  Quat q;
  Vec3 t;
  const Quat& unit_quaternion() const { return q; }
  const Vec3& translation() const { return t; }
  SE3 inverse() const {
    SE3 o;
    o.q = Quat{q.w_, -q.x_, -q.y_, -q.z_};
    const Vec3 r = rotate(o.q, t);
    o.t = Vec3{{-r.v[0], -r.v[1], -r.v[2]}};
    return o;
  }
  SE3 operator*(const SE3& b) const {
    SE3 o;
    o.q = Quat{q.w_ * b.q.w_ - q.x_ * b.q.x_ - q.y_ * b.q.y_ - q.z_ * b.q.z_,
               q.w_ * b.q.x_ + q.x_ * b.q.w_ + q.y_ * b.q.z_ - q.z_ * b.q.y_,
               q.w_ * b.q.y_ - q.x_ * b.q.z_ + q.y_ * b.q.w_ + q.z_ * b.q.x_,
               q.w_ * b.q.z_ + q.x_ * b.q.y_ - q.y_ * b.q.x_ + q.z_ * b.q.w_};
    const Vec3 r = rotate(q, b.t);
    o.t = Vec3{{r.v[0] + t.v[0], r.v[1] + t.v[1], r.v[2] + t.v[2]}};
    return o;
  }
};
struct Frame {
  uint32_t id;
  SE3 pose;
  std::vector<uint8_t> img;
};
struct Point2f {
  float x, y;
};
struct FeatureWithIDepth {  // flame.h:88-99
  uint32_t id = 0;
  uint32_t frame_id = 0;
  Point2f xy;
  float idepth_mu = 0.0f;
  float idepth_var = 0.0f;
  bool valid = false;
  uint32_t num_updates = 0;
  uint32_t num_dropouts = 0;
  int search_status = 0;
};
struct LineStereoParams {
  float max_cost = 1300.0f;
  bool do_subpixel = true;
  float sample_dist = 1.0f;
  float second_best_factor = 1.5f;
};
struct FilterParams {
  int win_size = 5;
  float search_sigma = 2.0f, min_grad_mag = 5.0f, idepth_min = 1e-3f, idepth_max = 2.0f, epilength_min = 3.0f,
        epilength_max = 32.0f, process_var_factor = 1.01f, process_fail_var_factor = 1.1f;
  LineStereoParams sparams;
};
struct MeasParams {
  int win_size = 5;
  float pixel_var = 16.0f, epipolar_line_var = 1.0f;
};
struct FlameParams {
  float min_baseline = 0.01f;
  bool do_letterbox = false;
  float rescale_factor_min = 0.7f, rescale_factor_max = 1.4f, idepth_var_max = 0.25f;
  int max_dropouts = 5;
  float outlier_sigma_thresh = 3.0f;
  bool do_meas_fusion = true;
  FilterParams fparams;
  MeasParams zparams;
};
struct Mat3 {
  float m[9];
  float operator()(int r, int c) const { return m[3 * r + c]; }
};

extern "C" {
// oracle/stereo_oracle.c
struct stereo_frame_ref {
  uint32_t id;
  const uint8_t* img_pad;
  float q_to_new[4], t_to_new[3], q_to_pf[4], t_to_pf[3];
};
long stereo_update_feature_idepths(const flame_stereo_params* P, const float* K, const float* Kinv, int width, int height,
                                   int pad, int n_frames, const stereo_frame_ref* frames, const uint8_t* new_img_pad,
                                   const float* new_gradx_pad, const float* new_grady_pad, uint32_t curr_pf_id, int n,
                                   flame_stereo_feature* feats, int32_t* stats);
void stereo_make_frame(const uint8_t* img, int width, int height, int border, uint8_t* img_pad, float* gradx_pad,
                       float* grady_pad);
}

static unsigned long long sm(unsigned long long& s) {
  unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// A fronto-parallel textured wall at depth Z seen by cameras translated by (tx, 0, 0): view(u) = tex(u.x + f tx / Z).
static const int W = 320, H = 240;
static const float F = 262.5f, Z = 2.0f;
static float tex(double x, double y) {
  return (float)(128.0 + 50.0 * std::sin(0.31 * x + 0.05 * y) * std::cos(0.23 * y - 0.02 * x) + 40.0 * std::sin(0.11 * x * 1.7 + 0.4) +
                 25.0 * std::cos(0.57 * y + 0.13 * x));
}
static std::vector<uint8_t> render(float tx) {
  std::vector<uint8_t> img((size_t)W * H);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float v = tex(x - F * tx / Z, y);
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      img[(size_t)y * W + x] = (uint8_t)std::lrintf(v);
    }
  return img;
}

int main() {
  const Mat3 K = {{F, 0, W / 2.0f, 0, F, H / 2.0f, 0, 0, 1}};
  const Mat3 Kinv = {{1 / F, 0, -(W / 2.0f) / F, 0, 1 / F, -(H / 2.0f) / F, 0, 0, 1}};
  const Quat I = {1, 0, 0, 0};
  // camera poses in the world (Frame::pose): pose-frames 10, 11, new frame 12, all looking down +z
  std::map<uint32_t, std::shared_ptr<Frame>> pfs;
  const float cam_x[3] = {0.0f, 0.05f, 0.12f};
  std::shared_ptr<Frame> fr[3];
  for (int k = 0; k < 3; ++k) {
    fr[k].reset(new Frame());
    fr[k]->id = 10 + k;
    fr[k]->pose = SE3{I, Vec3{{cam_x[k], 0, 0}}};
    fr[k]->img = render(-cam_x[k]);  // a camera at +x sees the wall shifted to -x
  }
  pfs[10] = fr[0], pfs[11] = fr[1];
  const Frame& fnew = *fr[2];
  const Frame& curr_pf = *fr[1];

  std::vector<FeatureWithIDepth> feats;
  unsigned long long seed = 99;
  for (int y = 20; y < H - 20; y += 9)
    for (int x = 20; x < W - 20; x += 9) {
      FeatureWithIDepth f;
      f.id = (uint32_t)feats.size();
      f.frame_id = (feats.size() & 1) ? 11 : 10;
      f.xy.x = x + (sm(seed) >> 40) * (1.0f / 16777216.0f), f.xy.y = y + (sm(seed) >> 40) * (1.0f / 16777216.0f);
      f.idepth_mu = (1.0f / Z) * (0.9f + 0.2f * ((sm(seed) >> 40) * (1.0f / 16777216.0f)));
      f.idepth_var = 0.02f;
      f.valid = true;
      feats.push_back(f);
    }
  std::vector<FeatureWithIDepth> expect = feats;

  FlameParams params;
  bool ok = true;
  try {
    flame_hip::FeatureTracker tracker(K, Kinv, W, H);
    for (int k = 0; k < 3; ++k) tracker.addFrame(fr[k]->id, fr[k]->img.data(), W);
    flame_stereo_stats st;
    const bool success = tracker.updateFeatureIDepths(params, pfs, fnew, curr_pf, &feats, &st);

    // the checker on the same inputs
    const int pad = 5, pw = W + 2 * pad, ph = H + 2 * pad;
    std::vector<std::vector<uint8_t>> ip(3, std::vector<uint8_t>((size_t)pw * ph));
    std::vector<float> gx((size_t)pw * ph), gy((size_t)pw * ph), gx2((size_t)pw * ph), gy2((size_t)pw * ph);
    for (int k = 0; k < 3; ++k) stereo_make_frame(fr[k]->img.data(), W, H, pad, ip[k].data(), k == 2 ? gx.data() : gx2.data(),
                                                  k == 2 ? gy.data() : gy2.data());
    stereo_frame_ref refs[2];
    for (int k = 0; k < 2; ++k) {
      const flame_stereo_pose p = flame_hip::makePose(*fr[k], fnew, curr_pf);
      refs[k].id = p.frame_id, refs[k].img_pad = ip[k].data();
      std::memcpy(refs[k].q_to_new, p.q_ref_to_new, sizeof p.q_ref_to_new);
      std::memcpy(refs[k].t_to_new, p.t_ref_to_new, sizeof p.t_ref_to_new);
      std::memcpy(refs[k].q_to_pf, p.q_ref_to_pf, sizeof p.q_ref_to_pf);
      std::memcpy(refs[k].t_to_pf, p.t_ref_to_pf, sizeof p.t_ref_to_pf);
    }
    const flame_stereo_params sp = flame_hip::toStereoParams(params);
    int32_t ost[7];
    const long rc = stereo_update_feature_idepths(&sp, K.m, Kinv.m, W, H, pad, 2, refs, ip[2].data(), gx.data(), gy.data(), 11,
                                                  (int)expect.size(), flame_hip::adoptFeatures(expect.data()), ost);
    const bool same = rc == 0 && std::memcmp(expect.data(), feats.data(), feats.size() * sizeof(FeatureWithIDepth)) == 0;
    std::printf("updateFeatureIDepths: %d features, %d updated (checker %d), returned %d: %s\n", (int)feats.size(),
                st.num_idepth_updates, ost[0], (int)success, same && st.num_idepth_updates == ost[0] ? "ok" : "MISMATCH");
    ok = ok && same && st.num_idepth_updates == ost[0] && success == (ost[6] != 0);
    // the wall is at idepth 0.5: the fused estimates must have moved towards it
    double e0 = 0, e1 = 0;
    int n = 0;
    for (size_t i = 0; i < feats.size(); ++i)
      if (feats[i].num_updates == 1) e1 += std::fabs(feats[i].idepth_mu - 0.5), ++n;
    for (size_t i = 0; i < feats.size(); ++i) e0 += std::fabs((1.0f / Z) * 1.0f - 0.5f);
    std::printf("mean |idepth - truth| of %d updated features: %.5f %s\n", n, n ? e1 / n : -1.0,
                (n > (int)feats.size() / 2 && e1 / n < 0.02) ? "ok" : "BAD");
    ok = ok && n > (int)feats.size() / 2 && e1 / n < 0.02;
    (void)e0;
    // unknown frame id -> exception naming the feature
    std::vector<FeatureWithIDepth> bad = expect;
    bad[3].frame_id = 77;
    try {
      tracker.updateFeatureIDepths(params, pfs, fnew, curr_pf, &bad);
      std::printf("unknown frame: no exception BAD\n");
      ok = false;
    } catch (const flame_hip::StereoError& e) {
      std::printf("unknown frame: StereoError status %d feature %d %s\n", e.status, e.feature,
                  (e.status == FLAME_NLTGV2_ERR_INVALID_ARG && e.feature == 3) ? "ok" : "BAD");
      ok = ok && e.status == FLAME_NLTGV2_ERR_INVALID_ARG && e.feature == 3;
    }
  } catch (const flame_hip::StereoError& e) {
    std::printf("StereoError: %s (status %d)\n", e.what(), e.status);
    return e.status == FLAME_NLTGV2_ERR_NO_DEVICE ? 77 : 1;
  }
  return ok ? 0 : 1;
}
