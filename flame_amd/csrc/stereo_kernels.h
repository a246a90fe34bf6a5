// stereo_kernels.h -- launch wrappers of stereo_kernels.hip (include/flame_stereo.h's device side).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "flame_stereo.h"

namespace flame_hip {

using StereoParams = flame_stereo_params;    // passed to the kernel by value
using StereoFeature = flame_stereo_feature;  // 40-byte records, in place

struct StereoCamera {
  float K[9], Kinv[9];  // row-major
  int width, height, border;
};

struct V2 {
  float x, y;
};
struct V3 {
  float x, y, z;
};

// ---- Eigen semantics (Quaternionf * Vector3f, toRotationMatrix, fixed 3x3 products) ------------------------
__host__ __device__ inline V3 rotate(const float* q, V3 v) {  // q = (w, x, y, z); Eigen _transformVector
  const float w = q[0];
  const V3 u = {q[1], q[2], q[3]};
  V3 uv = {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
  uv.x += uv.x;
  uv.y += uv.y;
  uv.z += uv.z;
  const V3 c = {u.y * uv.z - u.z * uv.y, u.z * uv.x - u.x * uv.z, u.x * uv.y - u.y * uv.x};
  return {(v.x + w * uv.x) + c.x, (v.y + w * uv.y) + c.y, (v.z + w * uv.z) + c.z};
}

struct Geo {           // EpipolarGeometry<float> after loadGeometry (epipolar_geometry.h:84-102)
  float q[4];          // q_ref_to_cmp
  V3 t;                // t_ref_to_cmp
  V3 tcr;              // t_cmp_to_ref
  float M[9];          // KRKinv
  V3 Kt;
  V2 epipole;
};

__host__ __device__ inline void mul3(const float* a, const float* b, float* c) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}

__host__ __device__ inline void load_geometry(Geo& g, const StereoCamera& cam, const float* q, const float* t) {
  g.q[0] = q[0], g.q[1] = q[1], g.q[2] = q[2], g.q[3] = q[3];
  g.t = {t[0], t[1], t[2]};
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  // Quaternion::inverse(): conjugate / squaredNorm (packet reduction order of the 4 coefficients x,y,z,w)
  const float n2 = (x * x + z * z) + (y * y + w * w);
  float qi[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (n2 > 0.0f) qi[0] = w / n2, qi[1] = -x / n2, qi[2] = -y / n2, qi[3] = -z / n2;
  const V3 r = rotate(qi, g.t);
  g.tcr = {-r.x, -r.y, -r.z};
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const float R[9] = {1.0f - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0f - (txx + tzz),
                      tyz - twx,          txz - twy, tyz + twx, 1.0f - (txx + tyy)};
  float KR[9];
  mul3(cam.K, R, KR);
  mul3(KR, cam.Kinv, g.M);
  g.Kt.x = (cam.K[0] * t[0] + cam.K[1] * t[1]) + cam.K[2] * t[2];
  g.Kt.y = (cam.K[3] * t[0] + cam.K[4] * t[1]) + cam.K[5] * t[2];
  g.Kt.z = (cam.K[6] * t[0] + cam.K[7] * t[1]) + cam.K[8] * t[2];
  g.epipole = {0.0f, 0.0f};
  if (t[2] > 0) {
    g.epipole.x = (cam.K[0] * t[0] + cam.K[2] * t[2]) / t[2];
    g.epipole.y = (cam.K[4] * t[1] + cam.K[5] * t[2]) / t[2];
  }
}

// One per pose-frame the features may refer to; lives in device memory for the duration of a launch.  The epipolar
// geometry of the pose (towards the new frame, and towards the newest pose-frame for features that are moved) and the
// baseline are filled in by the host (fill_pose_entry): they are the same for every feature of the pose-frame.
struct StereoPoseEntry {
  uint32_t frame_id;
  float baseline;          // ||t_ref_to_new||, flame.cc:1318-1324
  const uint8_t* img_pad;  // device, (height + 2 border) x (width + 2 border)
  Geo geo_new, geo_pf;
};
inline void fill_pose_entry(StereoPoseEntry* e, const StereoCamera& cam, const flame_stereo_pose& p) {
  load_geometry(e->geo_new, cam, p.q_ref_to_new, p.t_ref_to_new);
  load_geometry(e->geo_pf, cam, p.q_ref_to_pf, p.t_ref_to_pf);
  const float* t = p.t_ref_to_new;
  e->baseline = sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);  // (correctly rounded on both sides)
}

// stats[0..5]: the reference's counters (flame.cc:1497-1502); [6]: lowest feature index that hit a reference
// assert; [7]: lowest feature index with an unknown frame id (both start at INT_MAX).
// The six counters themselves are accumulated in kStatSlots copies behind that block (kStatSlotStride ints apart: one
// 128-byte line each), chosen by workgroup, and summed by the host.
constexpr int kStatAssert = 6, kStatBadFrame = 7, kStatCount = 8;
constexpr int kStatSlots = 64, kStatSlotStride = 32;
constexpr int kStatWords = kStatCount + kStatSlots * kStatSlotStride;

hipError_t launch_update_feature_idepths(const StereoParams& P, const StereoCamera& cam, int n_poses,
                                         const StereoPoseEntry* poses, const uint8_t* new_img, const float* new_gx,
                                         const float* new_gy, uint32_t curr_pf_id, int n, StereoFeature* feats, int* stats,
                                         int lanes_per_feature, hipStream_t stream);
hipError_t launch_frame_pad_gradient(const uint8_t* img, int width, int height, int border, uint8_t* img_pad,
                                     float* gx_pad, float* gy_pad, hipStream_t stream);

}  // namespace flame_hip
