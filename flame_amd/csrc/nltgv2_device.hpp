// nltgv2_device.hpp -- device functions shared by the solver's kernel files (nltgv2_kernels.hip, nltgv2_persistent.hip): the
// reference's scalar arithmetic, kept expression for expression.
//
// Reference arithmetic: /root/reference/src/flame/optimizers/nltgv2_l1_graph_regularizer.{h,cc} (cited per function).  Every
// file that includes this is compiled with -ffp-contract=off: the reference build is plain x86-64 (no FMA,
// CMakeLists.txt:24), every expression keeps the reference's left-to-right float evaluation order.
#ifndef FLAME_AMD_NLTGV2_DEVICE_HPP_
#define FLAME_AMD_NLTGV2_DEVICE_HPP_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nltgv2_kernels.h"

namespace flame_hip {
namespace {


constexpr uint32_t kRole = 0x80000000u;

// proxNLTGV2Conj, h:171-176:  q / max(1,|q|).  For finite q this is exactly clamp(q,-1,1):
// |q|<=1 -> q/1 == q, |q|>1 -> q/|q| == +-1 exactly (IEEE x/x == 1).  For NaN/Inf the reference
// produces NaN and FLAME_ASSERTs (h:174); we report that through `ok` instead of exiting.
__device__ __forceinline__ float prox_conj(float q, bool& ok) {
  ok = ok && (__builtin_fabsf(q) <= 3.402823466e+38f);
  return __builtin_fminf(__builtin_fmaxf(q, -1.0f), 1.0f);
}

// proxL1, h:179-197 (thresh = step_x * (data_factor * data_weight), call site cc:149-150).
__device__ __forceinline__ float prox_l1(float x_min, float x_max, float step_x, float data_weight,
                                         float x, float data) {
  const float diff = x - data;
  const float thresh = step_x * data_weight;
  float new_x;
  if (diff > thresh) {
    new_x = x - thresh;
  } else if (diff < -thresh) {
    new_x = x + thresh;
  } else {
    new_x = data;
  }
  new_x = (new_x < x_min) ? x_min : new_x;
  new_x = (new_x > x_max) ? x_max : new_x;
  return new_x;
}

// One edge seen from one endpoint: dual update of (q1,q2,q3) (cc:99-110) followed by this
// endpoint's share of the primal scatter (cc:126-141).  (xi..) = source vertex, (xj..) = target.
struct EdgeOut {
  float q1, q2, q3;
};

__device__ __forceinline__ EdgeOut edge_dual(const SolverParams& p, float alpha, float beta, float dx,
                                             float dy, float q1, float q2, float q3, float xbi,
                                             float w1bi, float w2bi, float xbj, float w1bj,
                                             float w2bj, bool& ok) {
  float K1x = alpha * (xbi - xbj);
  K1x -= alpha * dx * w1bi;
  K1x -= alpha * dy * w2bi;
  EdgeOut o;
  o.q1 = prox_conj(q1 + p.step_q * K1x, ok);
  const float K2x = beta * (w1bi - w1bj);
  o.q2 = prox_conj(q2 + p.step_q * K2x, ok);
  const float K3x = beta * (w2bi - w2bj);
  o.q3 = prox_conj(q3 + p.step_q * K3x, ok);
  return o;
}

// ------------------------------------------------------------------------------------------------
// Per-vertex photometric residual (BASELINE config 5): |I_cmp(project(u, idepth)) - I_ref(u)|,
// NaN where it is undefined.  No live reference code (only the commented-out block flame.cc:854-893); built from the
// live, test-pinned pieces EpipolarGeometry::project (stereo/epipolar_geometry.h:127-143, 191-201) and
// utils::bilinearInterp<uint8_t,float> (utils/image_utils.h:199-214, 230-255).  It reads x, never writes it.  Used by
// the stand-alone sweep k_photo_residual and, when a standing target is set, by the epilogue of the persistent runs
// (the residual of the run's final x as part of the solver's own launch).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float photo_bilinear_u8(const uint8_t* __restrict__ data, int step, float x, float y) {
  const int xf = (int)x, yf = (int)y;
  const float dx = x - xf, dy = y - yf;
  const float w11 = dx * dy;
  const float w01 = dx - w11;
  const float w10 = dy - w11;
  const float w00 = 1.0f - dx - dy + w11;
  const uint8_t* p = data + (size_t)yf * step + xf;
  return w00 * p[0] + w01 * p[1] + w10 * p[step] + w11 * p[1 + step];
}

__device__ __forceinline__ bool photo_inside(float x, float y, int rows, int cols, int border) {
  return x >= (float)border && y >= (float)border && x < (float)(cols - border) && y < (float)(rows - border);
}

__device__ __forceinline__ float photo_residual_at(float2 u, float idepth, const PhotoGeometry& geo,
                                                   const uint8_t* __restrict__ ref, const uint8_t* __restrict__ cmp, int rows,
                                                   int cols, int step, int border) {
  float out = __builtin_nanf("");
  if (!(idepth != idepth) && !(idepth < 0.0f) && photo_inside(u.x, u.y, rows, cols, border)) {
    float h0, h1, h2;
    const float* K = geo.KRKinv;
    if (idepth == 0.0f) {  // maxDepthProjection
      h0 = (K[0] * u.x + K[1] * u.y) + K[2] * 1.0f;
      h1 = (K[3] * u.x + K[4] * u.y) + K[5] * 1.0f;
      h2 = (K[6] * u.x + K[7] * u.y) + K[8] * 1.0f;
    } else {
      const float depth = 1.0f / idepth;
      const float a = u.x * depth, b = u.y * depth, c = depth;
      h0 = ((K[0] * a + K[1] * b) + K[2] * c) + geo.Kt[0];
      h1 = ((K[3] * a + K[4] * b) + K[5] * c) + geo.Kt[1];
      h2 = ((K[6] * a + K[7] * b) + K[8] * c) + geo.Kt[2];
    }
    const float inv = 1.0f / h2;
    const float cx = h0 * inv, cy = h1 * inv;
    if (cx == cx && cy == cy && photo_inside(cx, cy, rows, cols, border)) {
      const float d = photo_bilinear_u8(cmp, step, cx, cy) - photo_bilinear_u8(ref, step, u.x, u.y);
      out = (d > 0) ? d : -d;
    }
  }
  return out;
}

// ---- the record exchange of the persistent kernels -----------------------------------------------------------------
typedef int v4i_t __attribute__((ext_vector_type(4)));

constexpr int kAuxSc1 = 16;  // cache-policy bits of the raw buffer builtins: bit 4 = sc1
// Exchange buffer of a persistent run, ONE allocation addressed through one buffer descriptor:
//   [ R0 | L0 | R1 | L1 | XCC ]   R/L = remote/local records of step parity 0/1, S = 16*n_packed bytes
//   each, XCC = one dword per vertex.
// "remote" records are written through (sc1) and can be read from any XCD (one-way ~0.45-0.55 us,
// every read is a trip to the memory side).  "local" records are written with a PLAIN store, i.e.
// they stay in the writer's XCD L2, where a reader on the SAME XCD finds them with an sc1 load in
// ~0.26 us without any memory-side traffic -- but a reader on another XCD would never see them
// (tools/hop_bench.hip measured both).  Which copy a lane polls is decided from the TRUE XCC ids of
// both waves (HW_REG_XCC_ID, exchanged once per launch through the XCC table), never from an
// assumed workgroup->XCD placement.
__device__ __forceinline__ unsigned read_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
}

}  // namespace
}  // namespace flame_hip

#endif  // FLAME_AMD_NLTGV2_DEVICE_HPP_
