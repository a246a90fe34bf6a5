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

struct V2 {
  float x, y;
};
struct V3 {
  float x, y, z;
};

// ---- Eigen semantics (Quaternionf * Vector3f, toRotationMatrix, fixed 3x3 products) ------------------------
__device__ __forceinline__ V3 rotate(const float* q, V3 v) {  // q = (w, x, y, z); Eigen _transformVector
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

__device__ __forceinline__ void mul3(const float* a, const float* b, float* c) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}

__device__ void load_geometry(Geo& g, const StereoCamera& cam, const float* q, const float* t) {
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

// cv::Rect::contains(Point2f): the point becomes a Point2i through cvRound
__device__ __forceinline__ bool rect_contains(int rx, int ry, int rw, int rh, V2 p) {
  const int ix = __float2int_rn(p.x), iy = __float2int_rn(p.y);
  return rx <= ix && ix < rx + rw && ry <= iy && iy < ry + rh;
}

__device__ __forceinline__ void fail_feature(const StereoParams& P, StereoFeature& f, int* __restrict__ stats) {
  f.idepth_var *= P.process_fail_var_factor;
  if (f.idepth_var > P.idepth_var_max) {
    f.valid = 0;
    atomicAdd(&stats[1], 1);
  }
  f.num_dropouts++;
  if (f.num_dropouts > (uint32_t)P.max_dropouts) {
    f.valid = 0;
    atomicAdd(&stats[2], 1);
  }
}

__global__ __launch_bounds__(64) void k_update_feature_idepths(
    const StereoParams P, const StereoCamera cam, const int n_poses, const StereoPoseEntry* __restrict__ poses,
    const uint8_t* __restrict__ new_img, const float* __restrict__ new_gx, const float* __restrict__ new_gy,
    const uint32_t curr_pf_id, const int n, StereoFeature* __restrict__ feats, int* __restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  StereoFeature f = feats[i];
  // pfs.at(fii.frame_id)
  int slot = -1;
  for (int k = 0; k < n_poses; ++k)
    if (poses[k].frame_id == f.frame_id) {
      slot = k;
      break;
    }
  if (slot < 0) {
    atomicMin(&stats[kStatBadFrame], i);
    return;
  }
  const StereoPoseEntry& pe = poses[slot];
  Geo geo;
  load_geometry(geo, cam, pe.q_ref_to_new, pe.t_ref_to_new);
  {
    const float* t = pe.t_ref_to_new;
    const float baseline = sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
    if (baseline < P.min_baseline) return;
  }
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
      Geo gpf;
      load_geometry(gpf, cam, pe.q_ref_to_pf, pe.t_ref_to_pf);
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
      const int r = line_match(P, rescale, patch, new_img, rows, cols, {u_start.x + off, u_start.y + off},
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
    atomicMin(&stats[kStatAssert], i);
    return;
  }
  // failure-type counters read the status field whatever wrote it last (flame.cc:1337-1345)
  if (f.search_status == 1) atomicAdd(&stats[3], 1);
  else if (f.search_status == 2) atomicAdd(&stats[4], 1);
  else if (f.search_status == 3) atomicAdd(&stats[5], 1);
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
      atomicMin(&stats[kStatAssert], i);
      return;
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
          atomicMin(&stats[kStatAssert], i);
          return;
        }
        if (P.do_meas_fusion) f.idepth_mu = mu_post, f.idepth_var = var_post;
        else f.idepth_mu = mu_meas, f.idepth_var = var_meas;
        f.valid = 1;
        f.num_updates++;
        f.num_dropouts = 0;
        atomicAdd(&stats[0], 1);
        updated = true;
      }
    }
  }
  if (!updated) fail_feature(P, f, stats);
  feats[i] = f;
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
                                         hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int block = 64;  // one wave per workgroup: 8.5 k features are only 133 waves, spread them over the CUs
  hipLaunchKernelGGL(k_update_feature_idepths, dim3((n + block - 1) / block), dim3(block), 0, stream, P, cam, n_poses,
                     poses, new_img, new_gx, new_gy, curr_pf_id, n, feats, stats);
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
