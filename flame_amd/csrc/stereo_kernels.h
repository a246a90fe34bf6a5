// stereo_kernels.h -- launch wrappers of stereo_kernels.hip (include/flame_stereo.h's device side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flame_stereo.h"

namespace flame_hip {

using StereoParams = flame_stereo_params;    // passed to the kernel by value
using StereoFeature = flame_stereo_feature;  // 40-byte records, in place

struct StereoCamera {
  float K[9], Kinv[9];  // row-major
  int width, height, border;
};

// One per pose-frame the features may refer to; lives in device memory for the duration of a launch.
struct StereoPoseEntry {
  uint32_t frame_id;
  uint32_t reserved_;
  const uint8_t* img_pad;  // device, (height + 2 border) x (width + 2 border)
  float q_ref_to_new[4], t_ref_to_new[3];
  float q_ref_to_pf[4], t_ref_to_pf[3];
  float reserved2_[2];
};

// stats[0..5]: the reference's counters (flame.cc:1497-1502); [6]: lowest feature index that hit a reference
// assert; [7]: lowest feature index with an unknown frame id (both start at INT_MAX).
constexpr int kStatAssert = 6, kStatBadFrame = 7, kStatCount = 8;

hipError_t launch_update_feature_idepths(const StereoParams& P, const StereoCamera& cam, int n_poses,
                                         const StereoPoseEntry* poses, const uint8_t* new_img, const float* new_gx,
                                         const float* new_gy, uint32_t curr_pf_id, int n, StereoFeature* feats, int* stats,
                                         hipStream_t stream);
hipError_t launch_frame_pad_gradient(const uint8_t* img, int width, int height, int border, uint8_t* img_pad,
                                     float* gx_pad, float* gy_pad, hipStream_t stream);

}  // namespace flame_hip
