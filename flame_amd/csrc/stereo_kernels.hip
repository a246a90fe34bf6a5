// stereo_kernels.hip -- gfx950 kernels of the per-feature epipolar inverse-depth update (include/flame_stereo.h).
//
// k_update_feature_idepths: ONE LANE PER FEATURE walks the reference's per-feature body
// (/root/reference/src/flame/flame.cc:1307-1495, the `omp parallel for`): load the epipolar geometry of the
// feature's pose-frame, predict, pick the search segment, sample the 5-tap reference patch, slide it along the
// epipolar segment in the new image (line_stereo.h:73-385), convert the match to an inverse-depth measurement
// with its variance and fuse.  The features are independent, the per-feature work is ~40 dependent bilinear
// samples of 4 bytes each, and the images (0.3-2 MB) sit in L2: the kernel is bound by the latency of that
// chain, not by HBM (8.5 k features read 40 B and write 40 B each).  Lanes of a wave diverge on the early exits
// exactly where the reference's loop `continue`s; that costs nothing here because the longest lane (the full
// search) dominates every wave anyway.
//
// Arithmetic: every expression keeps the reference's order and width (float unless the reference's overload
// resolution makes it double: `1.0f / sqrt(norm2)`, `norm2 > 1e-10`, `mu < 1e-6`, `1e6 * epi`); the build has
// -ffp-contract=off and correctly rounded division and sqrt, so results match the x86 build bit for bit.
//
// k_frame_pad_gradient: utils::Frame::create level 0 (frame.cc:33-71), one thread per padded pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stereo_kernels.h"

namespace flame_hip {
namespace {

__device__ __forceinline__ V2 max_depth_projection(const Geo& g, V2 u) {  // h:191-201
  const float h0 = (g.M[0] * u.x + g.M[1] * u.y) + g.M[2] * 1.0f;
  const float h1 = (g.M[3] * u.x + g.M[4] * u.y) + g.M[5] * 1.0f;
  const float h2 = (g.M[6] * u.x + g.M[7] * u.y) + g.M[8] * 1.0f;
  const float inv = 1.0f / h2;
  return {h0 * inv, h1 * inv};
}

// project(u_ref, idepth) h:127-143; false = the reference asserts
__device__ __forceinline__ bool project(const Geo& g, V2 u, float idepth, V2* out) {
  if (!(idepth >= 0.0f)) return false;
  if (idepth == 0.0f) {
    *out = max_depth_projection(g, u);
    return true;
  }
  const float depth = 1.0f / idepth;
  const V3 h = {u.x * depth, u.y * depth, depth};
  const float c0 = ((g.M[0] * h.x + g.M[1] * h.y) + g.M[2] * h.z) + g.Kt.x;
  const float c1 = ((g.M[3] * h.x + g.M[4] * h.y) + g.M[5] * h.z) + g.Kt.y;
  const float c2 = ((g.M[6] * h.x + g.M[7] * h.y) + g.M[8] * h.z) + g.Kt.z;
  if (!(fabsf(c2) > 0.0f)) return false;
  const float inv = 1.0f / c2;
  *out = {c0 * inv, c1 * inv};
  return true;
}

// project(u_ref, idepth, &u_cmp, &new_idepth) h:152-180
__device__ __forceinline__ bool project_idepth(const Geo& g, const StereoCamera& cam, V2 u, float idepth, V2* out,
                                               float* new_idepth) {
  if (!(idepth >= 0.0f)) return false;
  if (idepth == 0.0f) {
    *out = max_depth_projection(g, u);
    *new_idepth = 0.0f;
    return true;
  }
  const float depth = 1.0f / idepth;
  V3 p = {cam.Kinv[0] * u.x + cam.Kinv[2], cam.Kinv[4] * u.y + cam.Kinv[5], 1.0f};
  p.x *= depth, p.y *= depth, p.z *= depth;
  const V3 r = rotate(g.q, p);
  const V3 pc = {r.x + g.t.x, r.y + g.t.y, r.z + g.t.z};
  const float u0 = cam.K[0] * pc.x + cam.K[2] * pc.z;
  const float u1 = cam.K[4] * pc.y + cam.K[5] * pc.z;
  if (!(fabsf(pc.z) > 0.0f)) return false;
  const float nid = 1.0f / pc.z;
  *new_idepth = nid;
  *out = {u0 * nid, u1 * nid};
  return true;
}

// minDepthProjection h:239-264
__device__ __forceinline__ bool min_depth_projection(const Geo& g, const StereoCamera& cam, V2 u, V2* out) {
  if (g.t.z > 0) {
    *out = g.epipole;
  } else if (g.t.z == 0) {
    const V2 e = {cam.K[0] * g.t.x, cam.K[4] * g.t.y};
    const V2 inf = max_depth_projection(g, u);
    // `1e6 * epi`: cv::operator*(double, Point2f) multiplies in double and narrows
    *out = {inf.x + (float)((double)e.x * 1e6), inf.y + (float)((double)e.y * 1e6)};
  } else {
    const V3 p = {cam.Kinv[0] * u.x + cam.Kinv[2], cam.Kinv[4] * u.y + cam.Kinv[5], 1.0f};
    const V3 qp = rotate(g.q, p);
    const float min_depth = (1.0f - g.t.z) / qp.z;
    const V3 c = {min_depth * qp.x + g.t.x, min_depth * qp.y + g.t.y, min_depth * qp.z + g.t.z};
    if (!(c.z > 0.0f)) return false;
    *out = {(cam.K[0] * c.x + cam.K[2] * c.z) / c.z, (cam.K[4] * c.y + cam.K[5] * c.z) / c.z};
  }
  return true;
}

// epiline h:271-292
__device__ __forceinline__ bool epiline(const Geo& g, const StereoCamera& cam, V2 u, V2* u_inf, V2* epi) {
  V2 zero;
  if (!min_depth_projection(g, cam, u, &zero)) return false;
  *u_inf = max_depth_projection(g, u);
  V2 e = {zero.x - u_inf->x, zero.y - u_inf->y};
  const float norm2 = e.x * e.x + e.y * e.y;
  if ((double)norm2 > 1e-10) {
    const float inv = (float)(1.0 / sqrt((double)norm2));  // `1.0f / sqrt(norm2)` resolves to the double sqrt
    e.x *= inv, e.y *= inv;
  } else {
    e = {0.0f, 0.0f};
  }
  *epi = e;
  return true;
}

// referenceEpiline h:303-325
__device__ __forceinline__ bool reference_epiline(const Geo& g, const StereoCamera& cam, V2 u, V2* epi) {
  V2 e = {-cam.K[0] * g.tcr.x + g.tcr.z * (u.x - cam.K[2]), -cam.K[4] * g.tcr.y + g.tcr.z * (u.y - cam.K[5])};
  const float n2 = e.x * e.x + e.y * e.y;
  if (!(n2 > 0)) return false;
  const float inv = (float)(1.0 / sqrt((double)n2));
  e.x *= inv, e.y *= inv;
  *epi = e;
  return true;
}

// disparityToInverseDepth h:389-405
__device__ __forceinline__ float disparity_to_idepth(const Geo& g, V2 u, V2 u_inf, V2 epi, float disparity) {
  const float w = g.M[6] * u.x + g.M[7] * u.y + g.M[8];
  const V2 A = {g.Kt.x - g.Kt.z * (u_inf.x + disparity * epi.x), g.Kt.y - g.Kt.z * (u_inf.y + disparity * epi.y)};
  const float wd = w * disparity;
  const V2 b = {epi.x * wd, epi.y * wd};
  const float ATA = A.x * A.x + A.y * A.y;
  const float ATb = A.x * b.x + A.y * b.y;
  return ATb / ATA;
}

// bilinearInterp (image_utils.h:199-255)
__device__ __forceinline__ bool sample_ok(int rows, int cols, float x, float y) {
  return x >= 0 && y >= 0 && x < (float)(uint32_t)(cols - 1) && y < (float)(uint32_t)(rows - 1);
}
template <typename T>
__device__ __forceinline__ float bilinear(const T* __restrict__ img, int step, float x, float y) {
  const int xf = (int)x, yf = (int)y;
  const float dx = x - xf, dy = y - yf;
  const float w11 = dx * dy;
  const float w01 = dx - w11;
  const float w10 = dy - w11;
  const float w00 = 1.0f - dx - dy + w11;
  const T* p = img + (size_t)yf * step + xf;
  return w00 * p[0] + w01 * p[1] + w10 * p[step] + w11 * p[step + 1];
}

// clipLineLiangBarsky (image_utils.cc:269-372)
__device__ bool clip_segment(float xmin, float xmax, float ymin, float ymax, V2* a, V2* b) {
  float t0 = 0.0f, t1 = 1.0f;
  const float xd = b->x - a->x, yd = b->y - a->y;
#pragma unroll
  for (int edge = 0; edge < 4; ++edge) {
    float p, q;
    if (edge == 0) p = -xd, q = -(xmin - a->x);
    else if (edge == 1) p = xd, q = (xmax - a->x);
    else if (edge == 2) p = -yd, q = -(ymin - a->y);
    else p = yd, q = (ymax - a->y);
    const float r = q / p;
    if (p == 0 && q < 0) return false;
    if (p < 0) {
      if (r > t1) return false;
      else if (r > t0) t0 = r;
    } else if (p > 0) {
      if (r < t0) return false;
      else if (r < t1) t1 = r;
    }
  }
  V2 c0 = {a->x + t0 * xd, a->y + t0 * yd}, c1 = {a->x + t1 * xd, a->y + t1 * yd};
  if (c0.x < xmin) c0.x = xmin;
  if (c0.x > xmax) c0.x = xmax;
  if (c0.y < ymin) c0.y = ymin;
  if (c0.y > ymax) c0.y = ymax;
  if (c1.x < xmin) c1.x = xmin;
  if (c1.x > xmax) c1.x = xmax;
  if (c1.y < ymin) c1.y = ymin;
  if (c1.y > ymax) c1.y = ymax;
  *a = c0, *b = c1;
  return true;
}

enum Outcome { kAssert = -1, kNo = 0, kYes = 1 };

// inverse_depth_filter::predict (inverse_depth_filter.cc:36-62)
__device__ Outcome predict(const Geo& g, const StereoCamera& cam, float process_var_factor, V2 u, float mu, float var,
                           V2* u_cmp, float* mu_pred, float* var_pred) {
  if (!project_idepth(g, cam, u, mu, u_cmp, mu_pred)) return kAssert;
  if (*mu_pred < 0.0f) {
    *mu_pred = 0.0f;
    *var_pred = 1e10f;
    return kNo;
  }
  float f4 = *mu_pred / mu;
  f4 *= f4;
  f4 *= f4;
  if ((double)mu < 1e-6) f4 = 1;
  *var_pred = process_var_factor * f4 * var;
  return kYes;
}

// inverse_depth_filter::getSearchRegion (inverse_depth_filter.cc:64-176)
__device__ Outcome search_region(const StereoParams& P, const Geo& g, int width, int height, V2 u, float mu, float var,
                                 V2* start, V2* end) {
  float id_min = P.idepth_min, id_max = P.idepth_max;
  if (!isnan(mu) && !isnan(var)) {
    const float sigma = sqrtf(var);
    id_min = mu - P.search_sigma * sigma;
    id_max = mu + P.search_sigma * sigma;
  }
  id_min = (id_min < P.idepth_min) ? P.idepth_min : id_min;
  id_max = (id_max > P.idepth_max) ? P.idepth_max : id_max;
  if (id_max < id_min) return kNo;
  V2 a, b;
  if (!project(g, u, id_min, &a) || !project(g, u, id_max, &b)) return kAssert;
  V2 d = {b.x - a.x, b.y - a.y};
  float epilength = sqrtf(d.x * d.x + d.y * d.y);
  if (epilength <= 0) return kNo;
  const V2 epi = {d.x / epilength, d.y / epilength};
  const float xmin = 1.0f, ymin = 1.0f, xmax = (float)(width - 1), ymax = (float)(height - 1);
  if (isnan(a.x) || isnan(a.y) || isnan(b.x) || isnan(b.y)) return kAssert;
  if (!clip_segment(xmin, xmax, ymin, ymax, &a, &b)) return kNo;
  d = {b.x - a.x, b.y - a.y};
  epilength = sqrtf(d.x * d.x + d.y * d.y);
  if (epilength <= 0) return kNo;
  if (epilength < P.epilength_min) {
    const float pad = (P.epilength_min - epilength) / 2.0f;
    a.x -= epi.x * pad, a.y -= epi.y * pad;
    b.x += epi.x * pad, b.y += epi.y * pad;
  }
  if (epilength > P.epilength_max) {
    epilength = P.epilength_max;
    b = {a.x + epi.x * epilength, a.y + epi.y * epilength};
  }
  if (isnan(a.x) || isnan(a.y) || isnan(b.x) || isnan(b.y)) return kAssert;
  if (!clip_segment(xmin, xmax, ymin, ymax, &a, &b)) return kNo;
  *start = a, *end = b;
  return kYes;
}

constexpr int kMaxSearchSteps = 1 << 16;  // a degenerate segment would spin forever in the reference

// line_stereo::match (line_stereo.h:73-385): 0 success, 1 ambiguous, 2 max cost, -1 assert
__device__ int line_match(const StereoParams& P, float rescale_factor, const float (&ref)[5],
                          const uint8_t* __restrict__ img, int rows, int cols, V2 start, V2 end, V2* match) {
  float incx = end.x - start.x, incy = end.y - start.y;
  const float epl = sqrtf(incx * incx + incy * incy);
  incx *= P.sample_dist / epl;
  incy *= P.sample_dist / epl;
  float cpx = start.x, cpy = start.y;
  float s[4];  // the four samples behind the leading one: m2, m1, centre, p1
  {
    const float xs[4] = {cpx - 2.0f * incx, cpx - incx, cpx, cpx + incx};
    const float ys[4] = {cpy - 2.0f * incy, cpy - incy, cpy, cpy + incy};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!sample_ok(rows, cols, xs[k], ys[k])) return -1;
      s[k] = bilinear<uint8_t>(img, cols, xs[k], ys[k]);
    }
  }
  int loop = 0, c_best = -1, c_second = -1;
  float best_x = -1, best_y = -1;
  float best = 3.402823466e+38f, second = 3.402823466e+38f;
  const float qnan = __builtin_nanf("");
  float err_pre = qnan, err_post = qnan, diff_pre = qnan, diff_post = qnan;
  bool best_was_last = false;
  float ee_last = -1;
  float ec[5] = {qnan, qnan, qnan, qnan, qnan};  // residuals of the current step (p2, p1, c, m1, m2)
  float ep[5] = {qnan, qnan, qnan, qnan, qnan};  // ... of the previous step
  while ((((incx < 0) == (cpx > end.x)) && ((incy < 0) == (cpy > end.y))) || loop == 0) {
    const float lx = cpx + 2 * incx, ly = cpy + 2 * incy;
    if (!sample_ok(rows, cols, lx, ly) || loop >= kMaxSearchSteps) return -1;
    const float lead = bilinear<uint8_t>(img, cols, lx, ly);
#pragma unroll
    for (int k = 0; k < 5; ++k) ep[k] = ec[k];
    float ee = 0.0f;
    ec[0] = lead - ref[4], ee += ec[0] * ec[0];
    ec[1] = s[3] - ref[3], ee += ec[1] * ec[1];
    ec[2] = s[2] - ref[2], ee += ec[2] * ec[2];
    ec[3] = s[1] - ref[1], ee += ec[3] * ec[3];
    ec[4] = s[0] - ref[0], ee += ec[4] * ec[4];
    // the reference keeps the residuals in two alternating buffers A/B and sums eA*eB; float products commute,
    // so (current, previous) order gives the same bits
    const float cross = ec[0] * ep[0] + ec[1] * ep[1] + ec[2] * ep[2] + ec[3] * ep[3] + ec[4] * ep[4];
    if (ee < best) {
      second = best, c_second = c_best;
      best = ee, c_best = loop;
      err_pre = ee_last, diff_pre = cross;
      err_post = -1, diff_post = -1;
      best_x = cpx, best_y = cpy;
      best_was_last = true;
    } else {
      if (best_was_last) {
        err_post = ee, diff_post = cross;
        best_was_last = false;
      }
      if (ee < second) second = ee, c_second = loop;
    }
    ee_last = ee;
    s[0] = s[1], s[1] = s[2], s[2] = s[3], s[3] = lead;
    cpx += incx, cpy += incy;
    ++loop;
  }
  if (best > 4.0f * P.max_cost) return 2;
  {
    const int d = c_best - c_second;
    if (((float)(d > 0 ? d : -d) > 1.0f) && (P.second_best_factor * best > second)) return 1;
  }
  if (P.do_subpixel) {
    const float g_pre_pre = -(err_pre - diff_pre);
    const float g_pre_this = +(best - diff_pre);
    const float g_post_this = -(best - diff_post);
    const float g_post_post = +(err_post - diff_post);
    bool interp_pre = false, interp_post = false;
    if (err_pre < 0 || err_post < 0) {
    } else if ((g_post_this < 0) ^ (g_pre_this < 0)) {
    } else if ((g_pre_pre < 0) ^ (g_pre_this < 0)) {
      if (!((g_post_post < 0) ^ (g_post_this < 0))) interp_pre = true;
    } else if ((g_post_post < 0) ^ (g_post_this < 0)) {
      interp_post = true;
    }
    if (interp_pre) {
      const float d = g_pre_this / (g_pre_this - g_pre_pre);
      best_x -= d * incx;
      best_y -= d * incy;
      best = best - 2 * d * g_pre_this - (g_pre_pre - g_pre_this) * d * d;
    } else if (interp_post) {
      const float d = g_post_this / (g_post_this - g_post_post);
      best_x += d * incx;
      best_y += d * incy;
      best = best + 2 * d * g_post_this + (g_post_post - g_post_this) * d * d;
    }
  }
  const float sample_dist = P.sample_dist * rescale_factor;
  float grad = 0;
  float tmp = ref[4] - ref[3];
  grad += tmp * tmp;
  tmp = ref[3] - ref[2];
  grad += tmp * tmp;
  tmp = ref[2] - ref[1];
  grad += tmp * tmp;
  tmp = ref[1] - ref[0];
  grad += tmp * tmp;
  grad /= sample_dist * sample_dist;
  if (best > P.max_cost + sqrtf(grad) * 20) return 2;
  *match = {best_x, best_y};
  return 0;
}

// ---- line_stereo::match, 16 lanes per feature ------------------------------------------------------------------------
// The walk along the epipolar segment is a chain of ~33 dependent bilinear samples for one lane.  Here the 16 lanes of a
// DPP row share one feature: lane k of round r owns step t = 16 r + k.  What makes the reference's loop sequential is
//   * the position  cp_t = cp_{t-1} + inc  (a rounded float accumulation: no closed form) -- reproduced by a 16-step
//     shift-and-add across the row (v_add_f32 with a row_shr:1 operand; lane 0 picks up lane 15 of the round before);
//   * the sliding sample window -- every sample has ONE position whichever step reads it (the leading sample of step t
//     at cp_t + 2 inc, the first four at cp_0 + {-2,-1,0,1} inc), so lane k computes the leading sample of its own step
//     and reads the other five of its window from its neighbours' registers (row_shr / row_ror);
//   * the best / second-best bookkeeping -- which is "the two smallest (cost, step) pairs in lexicographic order" with
//     the costs of the steps next to the best one: every lane keeps that for its own steps, one row reduction at the end.
// Same values, same expression order per value as line_match above: bit-identical results.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float old, float src) {  // lanes without a source in the row keep `old`
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false);
}
constexpr int kRowShl = 0x100, kRowShr = 0x110, kRowRor = 0x120;

// positions of one round: on entry (x, y) of lane 0 is the round's first position; after the call lane k holds that
// plus k sequentially rounded additions of (incx, incy)
__device__ __forceinline__ void row_position_chain(float& x, float& y, float incx, float incy, int k) {
  const float ax = k ? incx : 0.0f, ay = k ? incy : 0.0f;  // lane 0: + 0.0f keeps its (positive) value
#pragma unroll
  for (int i = 1; i < 16; ++i) {
    x = dpp_mov<kRowShr + 1>(x, x) + ax;
    y = dpp_mov<kRowShr + 1>(y, y) + ay;
  }
}

// lexicographic (value, index) minimum over the 16 lanes of the row, result in every lane
__device__ __forceinline__ void row_min_pair(float& v, int& i) {
#define FLAME_ROW_MIN_STEP(N)                                        \
  {                                                                  \
    const float ov = dpp_mov<kRowRor + N>(v, v);                     \
    const int oi = dpp_mov<kRowRor + N>(i, i);                       \
    const bool take = (ov < v) || (ov == v && oi < i);               \
    v = take ? ov : v, i = take ? oi : i;                            \
  }
  FLAME_ROW_MIN_STEP(8) FLAME_ROW_MIN_STEP(4) FLAME_ROW_MIN_STEP(2) FLAME_ROW_MIN_STEP(1)
#undef FLAME_ROW_MIN_STEP
}

__device__ __forceinline__ uint32_t row_ballot(bool pred, int lane) {
  return (uint32_t)(__ballot(pred) >> (lane & 48)) & 0xffffu;
}

__device__ int line_match_row(const StereoParams& P, float rescale_factor, const float (&ref)[5],
                              const uint8_t* __restrict__ img, int rows, int cols, V2 start, V2 end, int lane, V2* match) {
  const int k = lane & 15;
  float incx = end.x - start.x, incy = end.y - start.y;
  const float epl = sqrtf(incx * incx + incy * incy);
  incx *= P.sample_dist / epl;
  incy *= P.sample_dist / epl;
  const float inc2x = 2 * incx, inc2y = 2 * incy;
  const float qnan = __builtin_nanf("");
  const float fmax = 3.402823466e+38f;
  // the four samples behind the first leading one live in lanes 12..15 of the "previous round" Z (m2, m1, centre, p1);
  // lane 11 is the residual the reference has not computed yet at step 0 (NaN)
  float Z = qnan;
  {
    const float px = (k == 12) ? start.x - 2.0f * incx : (k == 13) ? start.x - incx : (k == 14) ? start.x : start.x + incx;
    const float py = (k == 12) ? start.y - 2.0f * incy : (k == 13) ? start.y - incy : (k == 14) ? start.y : start.y + incy;
    const bool ok = sample_ok(rows, cols, px, py);
    if (row_ballot(k >= 12 && !ok, lane)) return -1;
    if (k >= 12) Z = bilinear<uint8_t>(img, cols, px, py);
  }
  auto keeps_going = [&](float x, float y) { return ((incx < 0) == (x > end.x)) && ((incy < 0) == (y > end.y)); };
  auto lead_sample = [&](float x, float y, bool* ok) {
    const float lx = x + inc2x, ly = y + inc2y;
    *ok = sample_ok(rows, cols, lx, ly);
    return bilinear<uint8_t>(img, cols, *ok ? lx : 0.0f, *ok ? ly : 0.0f);
  };
  // round 0: positions, loop condition, leading samples
  float ax = start.x, ay = start.y;
  row_position_chain(ax, ay, incx, incy, k);
  bool okA;
  float A = lead_sample(ax, ay, &okA);
  uint32_t goA = row_ballot(keeps_going(ax, ay) || k == 0, lane);  // (`|| loop == 0`)
  float lb = fmax, ls = fmax;            // this lane's best and second-best cost over its own steps
  int lb_i = 0x7fffffff, ls_i = 0x7fffffff;
  float c_pre = -1, c_dpre = qnan, c_post = -1, c_dpost = -1, c_x = -1, c_y = -1;  // ... and what goes with its best
  for (int base = 0;; base += 16) {
    // the next round, one ahead: lane 15's step is followed by its lane 0
    float bx = dpp_mov<kRowRor + 1>(ax, ax) + incx, by = dpp_mov<kRowRor + 1>(ay, ay) + incy;
    row_position_chain(bx, by, incx, incy, k);
    bool okB;
    const float B = lead_sample(bx, by, &okB);
    const uint32_t goB = row_ballot(keeps_going(bx, by), lane);
    const int n_valid = __builtin_ctz(~goA | 0x10000u);  // steps base .. base + n_valid - 1 run
    const bool valid = k < n_valid;
    if (row_ballot(valid && (!okA || base + k >= kMaxSearchSteps), lane)) return -1;
    // window: L_m = leading sample of step t - m
    const float Lm1 = dpp_mov<kRowShl + 1>(dpp_mov<kRowRor + 15>(B, B), A);
    const float L0 = A;
    const float L1 = dpp_mov<kRowShr + 1>(dpp_mov<kRowRor + 1>(Z, Z), A);
    const float L2 = dpp_mov<kRowShr + 2>(dpp_mov<kRowRor + 2>(Z, Z), A);
    const float L3 = dpp_mov<kRowShr + 3>(dpp_mov<kRowRor + 3>(Z, Z), A);
    const float L4 = dpp_mov<kRowShr + 4>(dpp_mov<kRowRor + 4>(Z, Z), A);
    const float L5 = dpp_mov<kRowShr + 5>(dpp_mov<kRowRor + 5>(Z, Z), A);
    // residuals of this step (c), the one before (p) and the one after (n): e[j] = sample - ref[4 - j]
    const float c0 = L0 - ref[4], c1 = L1 - ref[3], c2 = L2 - ref[2], c3 = L3 - ref[1], c4 = L4 - ref[0];
    const float p0 = L1 - ref[4], p1 = L2 - ref[3], p2 = L3 - ref[2], p3 = L4 - ref[1], p4 = L5 - ref[0];
    const float n0 = Lm1 - ref[4], n1 = L0 - ref[3], n2 = L1 - ref[2], n3 = L2 - ref[1], n4 = L3 - ref[0];
    const float ee = (((c0 * c0 + c1 * c1) + c2 * c2) + c3 * c3) + c4 * c4;
    const float ee_prev = (((p0 * p0 + p1 * p1) + p2 * p2) + p3 * p3) + p4 * p4;
    const float ee_next = (((n0 * n0 + n1 * n1) + n2 * n2) + n3 * n3) + n4 * n4;
    const float cross = (((c0 * p0 + c1 * p1) + c2 * p2) + c3 * p3) + c4 * p4;
    const float cross_next = (((n0 * c0 + n1 * c1) + n2 * c2) + n3 * c3) + n4 * c4;
    const bool has_next = (k < 15) ? (k + 1 < n_valid) : (n_valid == 16 && (goB & 1u));
    if (valid) {
      if (ee < lb) {
        ls = lb, ls_i = lb_i;
        lb = ee, lb_i = base + k;
        c_pre = (base + k) ? ee_prev : -1.0f, c_dpre = cross;
        c_post = has_next ? ee_next : -1.0f, c_dpost = has_next ? cross_next : -1.0f;
        c_x = ax, c_y = ay;
      } else if (ee < ls) {
        ls = ee, ls_i = base + k;
      }
    }
    if (n_valid < 16 || !(goB & 1u)) break;
    Z = A, A = B, okA = okB, ax = bx, ay = by, goA = goB;
  }
  // the row's best step, and the best of the rest
  float best = lb;
  int c_best = lb_i;
  row_min_pair(best, c_best);
  const bool winner = (lb_i == c_best);
  float second = winner ? ls : lb;
  int c_second = winner ? ls_i : lb_i;
  row_min_pair(second, c_second);
  if (c_second == 0x7fffffff) c_second = -1;
  const int src = (lane & 48) | (c_best & 15);  // step t lives in lane t mod 16
  const float err_pre = __shfl(c_pre, src, 64), diff_pre = __shfl(c_dpre, src, 64);
  const float err_post = __shfl(c_post, src, 64), diff_post = __shfl(c_dpost, src, 64);
  float best_x = __shfl(c_x, src, 64), best_y = __shfl(c_y, src, 64);
  if (best > 4.0f * P.max_cost) return 2;
  {
    const int d = c_best - c_second;
    if (((float)(d > 0 ? d : -d) > 1.0f) && (P.second_best_factor * best > second)) return 1;
  }
  if (P.do_subpixel) {
    const float g_pre_pre = -(err_pre - diff_pre);
    const float g_pre_this = +(best - diff_pre);
    const float g_post_this = -(best - diff_post);
    const float g_post_post = +(err_post - diff_post);
    bool interp_pre = false, interp_post = false;
    if (err_pre < 0 || err_post < 0) {
    } else if ((g_post_this < 0) ^ (g_pre_this < 0)) {
    } else if ((g_pre_pre < 0) ^ (g_pre_this < 0)) {
      if (!((g_post_post < 0) ^ (g_post_this < 0))) interp_pre = true;
    } else if ((g_post_post < 0) ^ (g_post_this < 0)) {
      interp_post = true;
    }
    if (interp_pre) {
      const float d = g_pre_this / (g_pre_this - g_pre_pre);
      best_x -= d * incx;
      best_y -= d * incy;
      best = best - 2 * d * g_pre_this - (g_pre_pre - g_pre_this) * d * d;
    } else if (interp_post) {
      const float d = g_post_this / (g_post_this - g_post_post);
      best_x += d * incx;
      best_y += d * incy;
      best = best + 2 * d * g_post_this + (g_post_post - g_post_this) * d * d;
    }
  }
  const float sample_dist = P.sample_dist * rescale_factor;
  float grad = 0;
  float tmp = ref[4] - ref[3];
  grad += tmp * tmp;
  tmp = ref[3] - ref[2];
  grad += tmp * tmp;
  tmp = ref[2] - ref[1];
  grad += tmp * tmp;
  tmp = ref[1] - ref[0];
  grad += tmp * tmp;
  grad /= sample_dist * sample_dist;
  if (best > P.max_cost + sqrtf(grad) * 20) return 2;
  *match = {best_x, best_y};
  return 0;
}


// cv::Rect::contains(Point2f): the point becomes a Point2i through cvRound
__device__ __forceinline__ bool rect_contains(int rx, int ry, int rw, int rh, V2 p) {
  const int ix = __float2int_rn(p.x), iy = __float2int_rn(p.y);
  return rx <= ix && ix < rx + rw && ry <= iy && iy < ry + rh;
}

__device__ __forceinline__ uint32_t fail_feature(const StereoParams& P, StereoFeature& f) {  // -> counters to bump (bit c)
  uint32_t bits = 0;
  f.idepth_var *= P.process_fail_var_factor;
  if (f.idepth_var > P.idepth_var_max) {
    f.valid = 0;
    bits |= 1u << 1;
  }
  f.num_dropouts++;
  if (f.num_dropouts > (uint32_t)P.max_dropouts) {
    f.valid = 0;
    bits |= 1u << 2;
  }
  return bits;
}

// Returns the statistics counters this feature bumps (bit c = stats[c]); the kernel adds them up per wave.
template <int G>
__device__ uint32_t update_one_feature(const StereoParams& P, const StereoCamera& cam, const int n_poses,
                                       const StereoPoseEntry* __restrict__ poses, const uint8_t* __restrict__ new_img,
                                       const float* __restrict__ new_gx, const float* __restrict__ new_gy, const uint32_t curr_pf_id,
                                       const int i, const bool leader, StereoFeature* __restrict__ feats, int* __restrict__ stats) {
  uint32_t bits = 0;
  StereoFeature f = feats[i];
  // pfs.at(fii.frame_id)
  int slot = -1;
  for (int k = 0; k < n_poses; ++k)
    if (poses[k].frame_id == f.frame_id) {
      slot = k;
      break;
    }
  if (slot < 0) {
    if (leader) atomicMin(&stats[kStatBadFrame], i);
    return 0;
  }
  const StereoPoseEntry& pe = poses[slot];
  // EpipolarGeometry::loadGeometry depends on the pose-frame only: done once per pose by the host (stereo_kernels.h,
  // same float operations in the same order, no contraction on either side), not once per feature here.
  const Geo& geo = pe.geo_new;
  if (pe.baseline < P.min_baseline) return 0;
  const int width = cam.width, height = cam.height, pad = cam.border;
  const int rows = height + 2 * pad, cols = width + 2 * pad;
  bool asserted = false, tracked = false;
  V2 flow = {0.0f, 0.0f};
  // ---- trackFeature (flame.cc:1538-1752) -----------------------------------------------------------------
  do {
    const V2 xy = {f.x, f.y};
    V2 u_cmp;
    float idepth_cmp, var_cmp;
    const Outcome pr = predict(geo, cam, P.process_var_factor, xy, f.idepth_mu, f.idepth_var, &u_cmp, &idepth_cmp, &var_cmp);
    if (pr == kAssert) asserted = true;
    if (pr != kYes) break;
    const int row_offset = P.do_letterbox ? height / 3 : 0;
    const int border = (int)(P.rescale_factor_max * P.win_size / 2 + 1);
    const int vx = border, vy = border + row_offset, vw = width - 2 * border, vh = height - 2 * border - 2 * row_offset;
    float rescale = 1.0f;
    if ((f.idepth_mu > 0.0f) && (idepth_cmp > 0.0f)) rescale = idepth_cmp / f.idepth_mu;
    if (isnan(rescale) || !(rescale > 0)) {
      asserted = true;
      break;
    }
    if ((rescale <= P.rescale_factor_min) || (rescale >= P.rescale_factor_max)) {
      // the patch warp is too large: re-anchor the feature in the newest pose-frame (flame.cc:1596-1659)
      const Geo& gpf = pe.geo_pf;
      V2 u_pf;
      float idepth_pf, var_pf;
      const Outcome mr = predict(gpf, cam, P.process_var_factor, xy, f.idepth_mu, f.idepth_var, &u_pf, &idepth_pf, &var_pf);
      if (mr == kAssert) {
        asserted = true;
        break;
      }
      if (mr != kYes || !rect_contains(vx, vy, vw, vh, u_pf)) {
        f.valid = 0;
        break;
      }
      f.frame_id = curr_pf_id;
      f.x = u_pf.x, f.y = u_pf.y;
      const float old_idepth = f.idepth_mu;
      f.idepth_mu = idepth_pf;
      float v4 = idepth_pf / old_idepth;
      v4 *= v4;
      v4 *= v4;
      if ((double)idepth_pf < 1e-6) v4 = 1;
      f.idepth_var *= v4;
      break;
    }
    V2 u_start, u_end;
    const Outcome sr = search_region(P, geo, width, height, xy, f.idepth_mu, f.idepth_var, &u_start, &u_end);
    if (sr == kAssert) asserted = true;
    if (sr != kYes) break;
    if (!rect_contains(vx, vy, vw, vh, xy)) break;
    // inverse_depth_filter::search (inverse_depth_filter.cc:178-263), in padded coordinates
    const float off = (float)pad;
    const V2 ur = {xy.x + off, xy.y + off};
    V2 epi_ref;
    if (!reference_epiline(geo, cam, ur, &epi_ref) || P.win_size != 5) {
      asserted = true;
      break;
    }
    if (!((ur.x - 2 * epi_ref.x * rescale) >= 0) || !((ur.x + 2 * epi_ref.x * rescale) < cols - 1) ||
        !((ur.y - 2 * epi_ref.y * rescale) >= 0) || !((ur.y + 2 * epi_ref.y * rescale) < rows - 1)) {
      asserted = true;
      break;
    }
    const uint8_t* __restrict__ ref_img = pe.img_pad;
    float patch[5];
    patch[0] = bilinear<uint8_t>(ref_img, cols, ur.x - 2 * epi_ref.x * rescale, ur.y - 2 * epi_ref.y * rescale);
    patch[1] = bilinear<uint8_t>(ref_img, cols, ur.x - epi_ref.x * rescale, ur.y - epi_ref.y * rescale);
    patch[2] = bilinear<uint8_t>(ref_img, cols, ur.x, ur.y);
    patch[3] = bilinear<uint8_t>(ref_img, cols, ur.x + epi_ref.x * rescale, ur.y + epi_ref.y * rescale);
    patch[4] = bilinear<uint8_t>(ref_img, cols, ur.x + 2 * epi_ref.x * rescale, ur.y + 2 * epi_ref.y * rescale);
    float gmax = 0.0f;
#pragma unroll
    for (int k = 1; k < 5; ++k) {
      const float d = patch[k] - patch[k - 1];
      const float a = (d > 0) ? d : -d;
      if (a > gmax) gmax = a;
    }
    int status;
    V2 m = u_cmp;
    if (gmax < P.min_grad_mag) {
      status = 1;  // FAIL_REF_PATCH_GRADIENT
    } else {
      int r;
      if constexpr (G == 16)
        r = line_match_row(P, rescale, patch, new_img, rows, cols, {u_start.x + off, u_start.y + off},
                           {u_end.x + off, u_end.y + off}, (int)threadIdx.x, &m);
      else
        r = line_match(P, rescale, patch, new_img, rows, cols, {u_start.x + off, u_start.y + off},
                       {u_end.x + off, u_end.y + off}, &m);
      if (r < 0) {
        asserted = true;
        break;
      }
      status = (r == 1) ? 2 : (r == 2) ? 3 : 0;
    }
    f.search_status = status;
    if (status != 0) break;
    flow = {m.x - off, m.y - off};
    tracked = true;
  } while (false);
  if (asserted) {
    if (leader) atomicMin(&stats[kStatAssert], i);
    return 0;
  }
  // failure-type counters read the status field whatever wrote it last (flame.cc:1337-1345)
  if (f.search_status == 1) bits |= 1u << 3;
  else if (f.search_status == 2) bits |= 1u << 4;
  else if (f.search_status == 3) bits |= 1u << 5;
  bool updated = false;
  if (tracked) {
    // ---- InverseDepthMeasModel::idepth (inverse_depth_meas_model.cc:48-154) ------------------------------
    const V2 xy = {f.x, f.y};
    bool sensed = false;
    float mu_meas = 0.0f, var_meas = 1e10f;
    do {
      V2 u_inf, epi;
      if (!epiline(geo, cam, xy, &u_inf, &epi)) {
        asserted = true;
        break;
      }
      const float disp = epi.x * (flow.x - u_inf.x) + epi.y * (flow.y - u_inf.y);
      if ((double)disp < 1e-3) break;
      const float mu = disparity_to_idepth(geo, xy, u_inf, epi, disp);
      if (mu < 0.0f) break;
      const float off = (float)(P.z_win_size / 2 + 1);
      if (!sample_ok(rows, cols, flow.x + off, flow.y + off)) {
        asserted = true;
        break;
      }
      const float gx = bilinear<float>(new_gx, cols, flow.x + off, flow.y + off);
      const float gy = bilinear<float>(new_gy, cols, flow.x + off, flow.y + off);
      const float gnorm = sqrtf(gx * gx + gy * gy);
      if ((double)gnorm < 1e-3) break;
      const float ngx = gx / gnorm, ngy = gy / gnorm;
      const float edn = ngx * epi.x + ngy * epi.y;
      const float geo_var = P.epipolar_line_var / (edn * edn);
      if ((double)((edn > 0) ? edn : -edn) < 1e-3) break;
      const float edg = gx * epi.x + gy * epi.y;
      const float photo_var = 2 * P.pixel_var / (edg * edg);
      const float dmin = disp - disp / 10, dmax = disp + disp / 10;
      const float idmin = disparity_to_idepth(geo, xy, u_inf, epi, dmin);
      const float idmax = disparity_to_idepth(geo, xy, u_inf, epi, dmax);
      const float alpha = (idmax - idmin) / (dmax - dmin);
      const float meas_var = alpha * alpha * (geo_var + photo_var);
      if (isnan(meas_var) || isinf(meas_var)) {
        asserted = true;
        break;
      }
      mu_meas = mu, var_meas = meas_var;
      sensed = true;
    } while (false);
    if (asserted) {
      if (leader) atomicMin(&stats[kStatAssert], i);
      return 0;
    }
    if (sensed) {
      // ---- inverse_depth_filter::update (inverse_depth_filter.cc:265-303) --------------------------------
      const float mu_pred = f.idepth_mu, var_pred = f.idepth_var;
      float mu_post, var_post;
      if (!isnan(mu_pred) && (mu_pred > 0.0f)) {
        const float w = var_pred + var_meas;
        mu_post = (var_meas * mu_pred + var_pred * mu_meas) / w;
        var_post = (var_pred * var_meas) / w;
      } else {
        mu_post = mu_meas, var_post = var_meas;
      }
      const float res = mu_meas - mu_pred;
      const float dist = res * res / var_pred;
      if (!(dist > P.outlier_sigma_thresh * P.outlier_sigma_thresh)) {
        mu_post = (mu_post <= 0) ? 0.0f : mu_post;
        if (isnan(mu_post) || isnan(var_post) || !(var_post >= 0)) {
          if (leader) atomicMin(&stats[kStatAssert], i);
          return 0;
        }
        if (P.do_meas_fusion) f.idepth_mu = mu_post, f.idepth_var = var_post;
        else f.idepth_mu = mu_meas, f.idepth_var = var_meas;
        f.valid = 1;
        f.num_updates++;
        f.num_dropouts = 0;
        bits |= 1u;
        updated = true;
      }
    }
  }
  if (!updated) bits |= fail_feature(P, f);
  if (leader) feats[i] = f;
  return bits;
}

// G lanes per feature: 16 (a DPP row shares the feature: every lane runs the short scalar parts redundantly -- they cost
// the wave the same whether 1 or 16 lanes are live -- and the row splits the epipolar walk, line_match_row), or 1 (the
// whole body in one lane, line_match).  Lane 0 of a group stores; the six statistics counters are summed over the wave
// (ballot + popcount) and added with one atomic each into one of kStatSlots copies of the counter block: every feature
// adding to the same six words made the kernel run at the speed of same-address atomics (2.9 ns per feature: 24 us at
// 8.4 k features, 171 us at 57 k, whatever the rest of the kernel did).
template <int G>
__global__ __launch_bounds__(64) void k_update_feature_idepths(
    const StereoParams P, const StereoCamera cam, const int n_poses, const StereoPoseEntry* __restrict__ poses,
    const uint8_t* __restrict__ new_img, const float* __restrict__ new_gx, const float* __restrict__ new_gy,
    const uint32_t curr_pf_id, const int n, StereoFeature* __restrict__ feats, int* __restrict__ stats) {
  const int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) / G);
  const bool leader = (threadIdx.x % G) == 0;
  uint32_t bits = 0;
  if (i < n) bits = update_one_feature<G>(P, cam, n_poses, poses, new_img, new_gx, new_gy, curr_pf_id, i, leader, feats, stats);
  if (!leader) bits = 0;
  int* slot = stats + kStatCount + (blockIdx.x % kStatSlots) * kStatSlotStride;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const unsigned long long m = __ballot((bits >> c) & 1u);
    if (threadIdx.x == 0 && m) atomicAdd(&slot[c], __popcll(m));
  }
}



// utils::Frame::create level 0: img_pad = copyMakeBorder(REFLECT_101), grad*_pad = copyMakeBorder(
// getCentralGradient(img), CONSTANT 0).  The differences of two bytes (and their halves) are exact in float.
__global__ __launch_bounds__(256) void k_frame_pad_gradient(const uint8_t* __restrict__ img, int width, int height,
                                                            int border, uint8_t* __restrict__ img_pad,
                                                            float* __restrict__ gx_pad, float* __restrict__ gy_pad) {
  const int pw = width + 2 * border, ph = height + 2 * border;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= pw || y >= ph) return;
  int sx = x - border, sy = y - border;
  const bool inside = sx >= 0 && sx < width && sy >= 0 && sy < height;
  float gx = 0.0f, gy = 0.0f;
  if (inside) {
    const uint8_t* row = img + (size_t)sy * width;
    if (sx == 0) gx = (float)row[1] - (float)row[0];
    else if (sx == width - 1) gx = (float)row[width - 1] - (float)row[width - 2];
    else gx = 0.5f * ((float)row[sx + 1] - (float)row[sx - 1]);
    if (sy == 0) gy = (float)img[width + sx] - (float)img[sx];
    else if (sy == height - 1) gy = (float)img[(size_t)(height - 1) * width + sx] - (float)img[(size_t)(height - 2) * width + sx];
    else gy = 0.5f * ((float)img[(size_t)(sy + 1) * width + sx] - (float)img[(size_t)(sy - 1) * width + sx]);
  }
  if (sx < 0) sx = -sx;
  if (sx >= width) sx = 2 * (width - 1) - sx;
  if (sy < 0) sy = -sy;
  if (sy >= height) sy = 2 * (height - 1) - sy;
  const size_t o = (size_t)y * pw + x;
  img_pad[o] = img[(size_t)sy * width + sx];
  gx_pad[o] = gx;
  gy_pad[o] = gy;
}

}  // namespace

hipError_t launch_update_feature_idepths(const StereoParams& P, const StereoCamera& cam, int n_poses,
                                         const StereoPoseEntry* poses, const uint8_t* new_img, const float* new_gx,
                                         const float* new_gy, uint32_t curr_pf_id, int n, StereoFeature* feats, int* stats,
                                         int lanes_per_feature, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int block = 64;  // one wave per workgroup, spread over the CUs: 8.5 k features are 2100 waves (133 at one lane each)
  if (lanes_per_feature == 16) {
    hipLaunchKernelGGL(k_update_feature_idepths<16>, dim3((n + 3) / 4), dim3(block), 0, stream, P, cam, n_poses, poses, new_img,
                       new_gx, new_gy, curr_pf_id, n, feats, stats);
  } else {
    hipLaunchKernelGGL(k_update_feature_idepths<1>, dim3((n + block - 1) / block), dim3(block), 0, stream, P, cam, n_poses,
                       poses, new_img, new_gx, new_gy, curr_pf_id, n, feats, stats);
  }
  return hipGetLastError();
}

hipError_t launch_frame_pad_gradient(const uint8_t* img, int width, int height, int border, uint8_t* img_pad,
                                     float* gx_pad, float* gy_pad, hipStream_t stream) {
  const int pw = width + 2 * border, ph = height + 2 * border;
  hipLaunchKernelGGL(k_frame_pad_gradient, dim3((pw + 255) / 256, ph), dim3(256), 0, stream, img, width, height, border,
                     img_pad, gx_pad, gy_pad);
  return hipGetLastError();
}

}  // namespace flame_hip
