/*
 * oracle/stereo_oracle.c -- TEST INFRASTRUCTURE (CPU checker) for the per-feature epipolar inverse-depth
 * update.  Nothing under flame_amd/ may link, load or call this file.
 *
 * Restates, in plain C with the reference's float evaluation order (compiled -ffp-contract=off):
 *   Flame::updateFeatureIDepths          /root/reference/src/flame/flame.cc:1280-1536
 *   Flame::trackFeature                  flame.cc:1538-1752
 *   stereo::EpipolarGeometry<float>      src/flame/stereo/epipolar_geometry.h:84-417
 *   stereo::inverse_depth_filter::*      src/flame/stereo/inverse_depth_filter.cc:36-307
 *   stereo::line_stereo::match           src/flame/stereo/line_stereo.h:73-385
 *   stereo::InverseDepthMeasModel::idepth  src/flame/stereo/inverse_depth_meas_model.cc:48-154
 *   utils::bilinearInterp, bilinearWeights  src/flame/utils/image_utils.h:199-255
 *   utils::clipLineLiangBarsky           src/flame/utils/image_utils.cc:269-372
 *   utils::Frame::create (level 0)       src/flame/utils/frame.cc:33-71  (+ getCentralGradient,
 *                                        image_utils.h:425-470; cv::copyMakeBorder REFLECT_101 / CONSTANT)
 *
 * PINNING.  The EpipolarGeometry functions are pinned by the reference's own known-answer tests
 * (test/stereo/epipolar_geometry_test.cc, 21 tests) -- tests/test_stereo.py replays them against this
 * file; bilinearInterp by test/utils/image_utils_test.cc:150-166, clipLineLiangBarsky by :644-752 (5 tests),
 * getCentralGradient by :171-326 (ramp images).  Everything else here (predict, search
 * region, line search, measurement model, fusion, the per-feature driver) has NO reference test and the
 * reference translation units cannot be built in this image (Eigen, Sophus and OpenCV are absent and
 * stand-in headers are not allowed): PARITY UNPINNED for those parts.
 *
 * Third-party semantics restated (dependencies of the reference that are not vendored):
 *   Eigen 3 (find_package(Eigen3) in the reference's cmake/setup):
 *     Quaternion::toRotationMatrix, Quaternion * Vector3 (_transformVector), Quaternion::inverse()
 *     (= conjugate / squaredNorm, the 4-float squaredNorm reduced in SSE packet order (x2+z2)+(y2+w2)),
 *     fixed-size 3x3 products with coefficient (a0*b0 + a1*b1) + a2*b2.
 *   OpenCV core (types.hpp): Point_<float> arithmetic (float op float), `double * Point2f` computed in
 *     double then narrowed, Point2f -> Point2i conversion by cvRound (round-half-even) inside
 *     Rect::contains(Point2f), Rect::contains as x <= px < x + width.
 *   Unqualified sqrt()/fabs() on floats resolve to the double overloads in these headers (<cmath> only):
 *     `1.0f / sqrt(norm2)` is a double division narrowed to float (epipolar_geometry.h:279,312).
 *
 * Reference asserts (FLAME_ASSERT -> exit(1)) become the return value -(1 + feature index) of
 * stereo_update_feature_idepths; features before the failing one have been updated.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---- parameter block: the members of flame::Params read by this path (params.h:36-126) ---------- */
typedef struct stereo_params {
  float min_baseline;           /* params.h:72   0.01  */
  int32_t do_letterbox;         /* params.h:47   false */
  float rescale_factor_min;     /* params.h:64   0.7   */
  float rescale_factor_max;     /* params.h:65   1.4   */
  float idepth_var_max;         /* params.h:68   0.25  */
  int32_t max_dropouts;         /* params.h:69   5     */
  float outlier_sigma_thresh;   /* params.h:70   3.0   */
  int32_t do_meas_fusion;       /* params.h:73   true  */
  /* fparams (inverse_depth_filter.h:50-71) */
  int32_t win_size;             /* 5 */
  float search_sigma;           /* 2 */
  float min_grad_mag;           /* 5 */
  float idepth_min;             /* 1e-3 */
  float idepth_max;             /* 2 */
  float epilength_min;          /* 3 */
  float epilength_max;          /* 32 */
  float process_var_factor;     /* 1.01 */
  float process_fail_var_factor;/* 1.1 */
  /* fparams.sparams (line_stereo.h:47-60) */
  float max_cost;               /* 1300 */
  int32_t do_subpixel;          /* true */
  float sample_dist;            /* 1 */
  float second_best_factor;     /* 1.5 */
  /* zparams (inverse_depth_meas_model.h:43-51) */
  int32_t z_win_size;           /* 5 */
  float pixel_var;              /* 16 */
  float epipolar_line_var;      /* 1 */
} stereo_params;

/* flame.h:88-99 */
typedef struct stereo_feature {
  uint32_t id, frame_id;
  float x, y;
  float idepth_mu, idepth_var;
  uint8_t valid;
  uint8_t pad_[3];
  uint32_t num_updates, num_dropouts;
  int32_t search_status; /* inverse_depth_filter::Status: 0 ok, 1 ref patch gradient, 2 ambiguous, 3 max cost */
} stereo_feature;

typedef struct stereo_geometry {
  float K[9], Kinv[9];
  float q[4]; /* w, x, y, z */
  float t[3];
  float tcr[3]; /* t_cmp_to_ref */
  float KRKinv[9];
  float Kt[3];
  float epx, epy;
} stereo_geometry;

static void quat_rotate(const float* q, const float* v, float* out) {
  const float w = q[0], ux = q[1], uy = q[2], uz = q[3];
  float uvx = uy * v[2] - uz * v[1];
  float uvy = uz * v[0] - ux * v[2];
  float uvz = ux * v[1] - uy * v[0];
  uvx += uvx, uvy += uvy, uvz += uvz;
  const float cx = uy * uvz - uz * uvy;
  const float cy = uz * uvx - ux * uvz;
  const float cz = ux * uvy - uy * uvx;
  out[0] = (v[0] + w * uvx) + cx;
  out[1] = (v[1] + w * uvy) + cy;
  out[2] = (v[2] + w * uvz) + cz;
}

static void mat3_mul(const float* a, const float* b, float* c) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}

/* epipolar_geometry.h:84-102.  K, Kinv row-major; q = (w,x,y,z). */
void stereo_load_geometry(stereo_geometry* g, const float* K, const float* Kinv, const float* q, const float* t) {
  memcpy(g->K, K, sizeof g->K);
  memcpy(g->Kinv, Kinv, sizeof g->Kinv);
  memcpy(g->q, q, sizeof g->q);
  memcpy(g->t, t, sizeof g->t);
  /* q.inverse(): conjugate / squaredNorm */
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  const float n2 = (x * x + z * z) + (y * y + w * w);
  float qi[4] = {w, -x, -y, -z};
  if (n2 > 0.0f) {
    qi[0] = w / n2, qi[1] = -x / n2, qi[2] = -y / n2, qi[3] = -z / n2;
  } else {
    qi[0] = qi[1] = qi[2] = qi[3] = 0.0f;
  }
  float r[3];
  quat_rotate(qi, t, r);
  g->tcr[0] = -r[0], g->tcr[1] = -r[1], g->tcr[2] = -r[2];
  /* toRotationMatrix */
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const float R[9] = {1.0f - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0f - (txx + tzz),
                      tyz - twx,          txz - twy, tyz + twx, 1.0f - (txx + tyy)};
  float KR[9];
  mat3_mul(K, R, KR);
  mat3_mul(KR, Kinv, g->KRKinv);
  for (int i = 0; i < 3; ++i) g->Kt[i] = (K[3 * i] * t[0] + K[3 * i + 1] * t[1]) + K[3 * i + 2] * t[2];
  g->epx = g->epy = 0.0f;
  if (t[2] > 0) {
    g->epx = (K[0] * t[0] + K[2] * t[2]) / t[2];
    g->epy = (K[4] * t[1] + K[5] * t[2]) / t[2];
  }
}

/* h:191-201 */
void stereo_max_depth_projection(const stereo_geometry* g, float ux, float uy, float* ox, float* oy) {
  const float* M = g->KRKinv;
  const float h0 = (M[0] * ux + M[1] * uy) + M[2] * 1.0f;
  const float h1 = (M[3] * ux + M[4] * uy) + M[5] * 1.0f;
  const float h2 = (M[6] * ux + M[7] * uy) + M[8] * 1.0f;
  const float inv = 1.0f / h2;
  *ox = h0 * inv;
  *oy = h1 * inv;
}

/* h:127-143; returns 0, or 1 when the reference would assert (idepth < 0) */
int stereo_project(const stereo_geometry* g, float ux, float uy, float idepth, float* ox, float* oy) {
  if (!(idepth >= 0.0f)) return 1;
  if (idepth == 0.0f) {
    stereo_max_depth_projection(g, ux, uy, ox, oy);
    return 0;
  }
  const float* M = g->KRKinv;
  const float depth = 1.0f / idepth;
  const float a = ux * depth, b = uy * depth, c = depth;
  const float h0 = ((M[0] * a + M[1] * b) + M[2] * c) + g->Kt[0];
  const float h1 = ((M[3] * a + M[4] * b) + M[5] * c) + g->Kt[1];
  const float h2 = ((M[6] * a + M[7] * b) + M[8] * c) + g->Kt[2];
  if (!(fabsf(h2) > 0.0f)) return 1;
  const float inv = 1.0f / h2;
  *ox = h0 * inv;
  *oy = h1 * inv;
  return 0;
}

/* h:152-180 */
int stereo_project_idepth(const stereo_geometry* g, float ux, float uy, float idepth, float* ox, float* oy,
                          float* new_idepth) {
  if (!(idepth >= 0.0f)) return 1;
  if (idepth == 0.0f) {
    stereo_max_depth_projection(g, ux, uy, ox, oy);
    *new_idepth = 0.0f;
    return 0;
  }
  const float depth = 1.0f / idepth;
  float p[3] = {g->Kinv[0] * ux + g->Kinv[2], g->Kinv[4] * uy + g->Kinv[5], 1.0f};
  p[0] *= depth, p[1] *= depth, p[2] *= depth;
  float r[3];
  quat_rotate(g->q, p, r);
  const float pc0 = r[0] + g->t[0], pc1 = r[1] + g->t[1], pc2 = r[2] + g->t[2];
  const float u0 = g->K[0] * pc0 + g->K[2] * pc2, u1 = g->K[4] * pc1 + g->K[5] * pc2;
  if (!(fabsf(pc2) > 0.0f)) return 1;
  *new_idepth = 1.0f / pc2;
  *ox = u0 * (*new_idepth);
  *oy = u1 * (*new_idepth);
  return 0;
}

/* h:239-264; returns 1 on the reference's assert (p_cmp(2) > 0) */
int stereo_min_depth_projection(const stereo_geometry* g, float ux, float uy, float* ox, float* oy) {
  if (g->t[2] > 0) {
    *ox = g->epx, *oy = g->epy;
  } else if (g->t[2] == 0) {
    const float ex = g->K[0] * g->t[0], ey = g->K[4] * g->t[1];
    float ix, iy;
    stereo_max_depth_projection(g, ux, uy, &ix, &iy);
    *ox = ix + (float)((double)ex * 1e6);
    *oy = iy + (float)((double)ey * 1e6);
  } else {
    const float p[3] = {g->Kinv[0] * ux + g->Kinv[2], g->Kinv[4] * uy + g->Kinv[5], 1.0f};
    float qp[3];
    quat_rotate(g->q, p, qp);
    const float min_depth = (1.0f - g->t[2]) / qp[2];
    const float c0 = min_depth * qp[0] + g->t[0], c1 = min_depth * qp[1] + g->t[1], c2 = min_depth * qp[2] + g->t[2];
    if (!(c2 > 0.0f)) return 1;
    *ox = (g->K[0] * c0 + g->K[2] * c2) / c2;
    *oy = (g->K[4] * c1 + g->K[5] * c2) / c2;
  }
  return 0;
}

/* h:271-292 */
int stereo_epiline(const stereo_geometry* g, float ux, float uy, float* ix, float* iy, float* ex, float* ey) {
  float zx, zy;
  if (stereo_min_depth_projection(g, ux, uy, &zx, &zy)) return 1;
  stereo_max_depth_projection(g, ux, uy, ix, iy);
  float dx = zx - *ix, dy = zy - *iy;
  const float norm2 = dx * dx + dy * dy;
  if ((double)norm2 > 1e-10) {
    const float inv = (float)(1.0 / sqrt((double)norm2));
    dx *= inv, dy *= inv;
  } else {
    dx = 0.0f, dy = 0.0f;
  }
  *ex = dx, *ey = dy;
  return 0;
}

/* h:303-325; returns 1 on the reference's assert (norm2 > 0) */
int stereo_reference_epiline(const stereo_geometry* g, float ux, float uy, float* ex, float* ey) {
  float ax = -g->K[0] * g->tcr[0] + g->tcr[2] * (ux - g->K[2]);
  float ay = -g->K[4] * g->tcr[1] + g->tcr[2] * (uy - g->K[5]);
  const float n2 = ax * ax + ay * ay;
  if (!(n2 > 0)) return 1;
  const float inv = (float)(1.0 / sqrt((double)n2));
  ax *= inv, ay *= inv;
  *ex = ax, *ey = ay;
  return 0;
}

/* h:336-350 */
int stereo_disparity(const stereo_geometry* g, float ux, float uy, float cx, float cy, float* ix, float* iy,
                     float* ex, float* ey, float* disp) {
  if (stereo_epiline(g, ux, uy, ix, iy, ex, ey)) return 1;
  *disp = *ex * (cx - *ix) + *ey * (cy - *iy);
  return 0;
}

/* h:361-376 */
float stereo_disparity_to_depth(const stereo_geometry* g, float ux, float uy, float ix, float iy, float ex, float ey,
                                float disparity) {
  const float w = g->KRKinv[6] * ux + g->KRKinv[7] * uy + g->KRKinv[8];
  const float wd = w * disparity;
  const float Ax = ex * wd, Ay = ey * wd;
  const float bx = g->Kt[0] - g->Kt[2] * (ix + disparity * ex);
  const float by = g->Kt[1] - g->Kt[2] * (iy + disparity * ey);
  const float ATA = Ax * Ax + Ay * Ay, ATb = Ax * bx + Ay * by;
  return ATb / ATA;
}

/* h:389-405 */
float stereo_disparity_to_idepth(const stereo_geometry* g, float ux, float uy, float ix, float iy, float ex,
                                 float ey, float disparity) {
  const float w = g->KRKinv[6] * ux + g->KRKinv[7] * uy + g->KRKinv[8];
  const float Ax = g->Kt[0] - g->Kt[2] * (ix + disparity * ex);
  const float Ay = g->Kt[1] - g->Kt[2] * (iy + disparity * ey);
  const float wd = w * disparity;
  const float bx = ex * wd, by = ey * wd;
  const float ATA = Ax * Ax + Ay * Ay, ATb = Ax * bx + Ay * by;
  return ATb / ATA;
}

/* image_utils.h:199-255 for ChannelType = uint8_t / float */
static int bil_ok(int rows, int cols, float x, float y) {
  return x >= 0 && y >= 0 && x < (float)(uint32_t)(cols - 1) && y < (float)(uint32_t)(rows - 1);
}
static float bil_u8(const uint8_t* data, int step, float x, float y) {
  const int xf = (int)x, yf = (int)y;
  const float dx = x - xf, dy = y - yf;
  const float w11 = dx * dy, w01 = dx - w11, w10 = dy - w11, w00 = 1.0f - dx - dy + w11;
  const uint8_t* p = data + (long)yf * step + xf;
  return w00 * p[0] + w01 * p[1] + w10 * p[step] + w11 * p[1 + step];
}
static float bil_f32(const float* data, int step, float x, float y) {
  const int xf = (int)x, yf = (int)y;
  const float dx = x - xf, dy = y - yf;
  const float w11 = dx * dy, w01 = dx - w11, w10 = dy - w11, w00 = 1.0f - dx - dy + w11;
  const float* p = data + (long)yf * step + xf;
  return w00 * p[0] + w01 * p[1] + w10 * p[step] + w11 * p[1 + step];
}

/* image_utils.cc:269-372; 1 = visible */
int stereo_clip_liang_barsky(float xmin, float xmax, float ymin, float ymax, float x0, float y0, float x1, float y1,
                             float* ox0, float* oy0, float* ox1, float* oy1) {
  float t0 = 0.0f, t1 = 1.0f;
  const float xd = x1 - x0, yd = y1 - y0;
  for (int edge = 0; edge < 4; ++edge) {
    float p = 1.0f, q = 0.0f;
    if (edge == 0) p = -xd, q = -(xmin - x0);
    else if (edge == 1) p = xd, q = (xmax - x0);
    else if (edge == 2) p = -yd, q = -(ymin - y0);
    else p = yd, q = (ymax - y0);
    const float r = q / p;
    if (p == 0 && q < 0) return 0;
    if (p < 0) {
      if (r > t1) return 0;
      else if (r > t0) t0 = r;
    } else if (p > 0) {
      if (r < t0) return 0;
      else if (r < t1) t1 = r;
    }
  }
  float a = x0 + t0 * xd, b = y0 + t0 * yd, c = x0 + t1 * xd, d = y0 + t1 * yd;
  if (a < xmin) a = xmin;
  if (a > xmax) a = xmax;
  if (b < ymin) b = ymin;
  if (b > ymax) b = ymax;
  if (c < xmin) c = xmin;
  if (c > xmax) c = xmax;
  if (d < ymin) d = ymin;
  if (d > ymax) d = ymax;
  *ox0 = a, *oy0 = b, *ox1 = c, *oy1 = d;
  return 1;
}

/* inverse_depth_filter.cc:36-62; 0 ok, 1 behind camera, -1 reference assert */
int stereo_predict(const stereo_geometry* g, float process_var_factor, float ux, float uy, float mu, float var,
                   float* cx, float* cy, float* mu_pred, float* var_pred) {
  if (stereo_project_idepth(g, ux, uy, mu, cx, cy, mu_pred)) return -1;
  if (*mu_pred < 0.0f) {
    *mu_pred = 0.0f;
    *var_pred = 1e10f;
    return 1;
  }
  float f = *mu_pred / mu;
  f *= f;
  f *= f;
  if ((double)mu < 1e-6) f = 1;
  *var_pred = process_var_factor * f * var;
  return 0;
}

/* inverse_depth_filter.cc:64-176; 1 = region found, 0 = none, -1 = reference assert */
int stereo_search_region(const stereo_params* P, const stereo_geometry* g, int width, int height, float ux, float uy,
                         float mu, float var, float* sx, float* sy, float* ex, float* ey, float* epx, float* epy) {
  float id_min = P->idepth_min, id_max = P->idepth_max;
  if (!isnan(mu) && !isnan(var)) {
    const float sigma = sqrtf(var);
    id_min = mu - P->search_sigma * sigma;
    id_max = mu + P->search_sigma * sigma;
  }
  id_min = (id_min < P->idepth_min) ? P->idepth_min : id_min;
  id_max = (id_max > P->idepth_max) ? P->idepth_max : id_max;
  if (id_max < id_min) return 0;
  float ax, ay, bx, by;
  if (stereo_project(g, ux, uy, id_min, &ax, &ay)) return -1;
  if (stereo_project(g, ux, uy, id_max, &bx, &by)) return -1;
  float dx = bx - ax, dy = by - ay;
  float epilength = sqrtf(dx * dx + dy * dy);
  if (epilength <= 0) return 0;
  const float epix = dx / epilength, epiy = dy / epilength;
  *epx = epix, *epy = epiy;
  const float xmin = 1.0f, ymin = 1.0f, xmax = (float)(1 + width - 2), ymax = (float)(1 + height - 2);
  if (isnan(ax) || isnan(ay) || isnan(bx) || isnan(by)) return -1;
  float cax, cay, cbx, cby;
  if (!stereo_clip_liang_barsky(xmin, xmax, ymin, ymax, ax, ay, bx, by, &cax, &cay, &cbx, &cby)) return 0;
  ax = cax, ay = cay, bx = cbx, by = cby;
  dx = bx - ax, dy = by - ay;
  epilength = sqrtf(dx * dx + dy * dy);
  if (epilength <= 0) return 0;
  if (epilength < P->epilength_min) {
    const float pad = (P->epilength_min - epilength) / 2.0f;
    ax -= epix * pad, ay -= epiy * pad;
    bx += epix * pad, by += epiy * pad;
  }
  if (epilength > P->epilength_max) {
    epilength = P->epilength_max;
    bx = ax + epix * epilength, by = ay + epiy * epilength;
  }
  if (isnan(ax) || isnan(ay) || isnan(bx) || isnan(by)) return -1;
  if (!stereo_clip_liang_barsky(xmin, xmax, ymin, ymax, ax, ay, bx, by, &cax, &cay, &cbx, &cby)) return 0;
  *sx = cax, *sy = cay, *ex = cbx, *ey = cby;
  return 1;
}

/* line_stereo.h:73-385.  Returns 0 success, 1 ambiguous, 2 max cost, -1 reference assert (sample outside image). */
int stereo_line_match(const stereo_params* P, float rescale_factor, const float* ref_patch, const uint8_t* img,
                      int rows, int cols, int step, float sx, float sy, float ex, float ey, float* mx, float* my,
                      float* residual) {
  const float rm2 = ref_patch[0], rm1 = ref_patch[1], r0 = ref_patch[2], rp1 = ref_patch[3], rp2 = ref_patch[4];
  float incx = ex - sx, incy = ey - sy;
  const float epl = sqrtf(incx * incx + incy * incy);
  incx *= P->sample_dist / epl;
  incy *= P->sample_dist / epl;
  float cpx = sx, cpy = sy;
#define SAMPLE(dst, X, Y)                      \
  do {                                         \
    const float sx_ = (X), sy_ = (Y);          \
    if (!bil_ok(rows, cols, sx_, sy_)) return -1; \
    dst = bil_u8(img, step, sx_, sy_);         \
  } while (0)
  float vm2, vm1, v0, vp1, vp2;
  SAMPLE(vm2, cpx - 2.0f * incx, cpy - 2.0f * incy);
  SAMPLE(vm1, cpx - incx, cpy - incy);
  SAMPLE(v0, cpx, cpy);
  SAMPLE(vp1, cpx + incx, cpy + incy);
  int loop = 0;
  float best_x = -1, best_y = -1, best_err = 3.402823466e+38f, second_err = 3.402823466e+38f;
  float errPre = NAN, errPost = NAN, diffPre = NAN, diffPost = NAN;
  int best_was_last = 0;
  float eeLast = -1;
  float eA[5] = {NAN, NAN, NAN, NAN, NAN}, eB[5] = {NAN, NAN, NAN, NAN, NAN};
  int cBest = -1, cSecond = -1;
  while ((((incx < 0) == (cpx > ex)) && ((incy < 0) == (cpy > ey))) || loop == 0) {
    if (loop >= 65536) return -1; /* degenerate segment: the reference would spin; both sides report an assert */
    SAMPLE(vp2, cpx + 2 * incx, cpy + 2 * incy);
    float ee = 0.0f;
    float* e = (loop % 2 == 0) ? eA : eB;
    e[0] = vp2 - rp2, ee += e[0] * e[0];
    e[1] = vp1 - rp1, ee += e[1] * e[1];
    e[2] = v0 - r0, ee += e[2] * e[2];
    e[3] = vm1 - rm1, ee += e[3] * e[3];
    e[4] = vm2 - rm2, ee += e[4] * e[4];
    if (ee < best_err) {
      second_err = best_err;
      cSecond = cBest;
      best_err = ee;
      cBest = loop;
      errPre = eeLast;
      diffPre = eA[0] * eB[0] + eA[1] * eB[1] + eA[2] * eB[2] + eA[3] * eB[3] + eA[4] * eB[4];
      errPost = -1;
      diffPost = -1;
      best_x = cpx, best_y = cpy;
      best_was_last = 1;
    } else {
      if (best_was_last) {
        errPost = ee;
        diffPost = eA[0] * eB[0] + eA[1] * eB[1] + eA[2] * eB[2] + eA[3] * eB[3] + eA[4] * eB[4];
        best_was_last = 0;
      }
      if (ee < second_err) {
        second_err = ee;
        cSecond = loop;
      }
    }
    eeLast = ee;
    vm2 = vm1, vm1 = v0, v0 = vp1, vp1 = vp2;
    cpx += incx, cpy += incy;
    ++loop;
  }
#undef SAMPLE
  *residual = best_err;
  if (best_err > 4.0f * P->max_cost) return 2;
  {
    int d = cBest - cSecond;
    d = d > 0 ? d : -d;
    if (((float)d > 1.0f) && (P->second_best_factor * best_err > second_err)) return 1;
  }
  if (P->do_subpixel) {
    const float gPre_pre = -(errPre - diffPre), gPre_this = +(best_err - diffPre);
    const float gPost_this = -(best_err - diffPost), gPost_post = +(errPost - diffPost);
    int interpPost = 0, interpPre = 0;
    if (errPre < 0 || errPost < 0) {
    } else if ((gPost_this < 0) ^ (gPre_this < 0)) {
    } else if ((gPre_pre < 0) ^ (gPre_this < 0)) {
      if ((gPost_post < 0) ^ (gPost_this < 0)) {
      } else {
        interpPre = 1;
      }
    } else if ((gPost_post < 0) ^ (gPost_this < 0)) {
      interpPost = 1;
    }
    if (interpPre) {
      const float d = gPre_this / (gPre_this - gPre_pre);
      best_x -= d * incx;
      best_y -= d * incy;
      best_err = best_err - 2 * d * gPre_this - (gPre_pre - gPre_this) * d * d;
    } else if (interpPost) {
      const float d = gPost_this / (gPost_this - gPost_post);
      best_x += d * incx;
      best_y += d * incy;
      best_err = best_err + 2 * d * gPost_this + (gPost_post - gPost_this) * d * d;
    }
  }
  *residual = best_err;
  const float sampleDist = P->sample_dist * rescale_factor;
  float grad = 0, tmp = rp2 - rp1;
  grad += tmp * tmp;
  tmp = rp1 - r0;
  grad += tmp * tmp;
  tmp = r0 - rm1;
  grad += tmp * tmp;
  tmp = rm1 - rm2;
  grad += tmp * tmp;
  grad /= sampleDist * sampleDist;
  if (best_err > P->max_cost + sqrtf(grad) * 20) return 2;
  *mx = best_x, *my = best_y;
  return 0;
}

/* inverse_depth_filter.cc:178-263.  Status 0..3, or -1 on a reference assert.  u_* in padded coordinates. */
int stereo_search(const stereo_params* P, const stereo_geometry* g, float rescale_factor, const uint8_t* img_ref,
                  const uint8_t* img_cmp, int rows, int cols, int step, float ux, float uy, float sx, float sy,
                  float ex, float ey, float* mx, float* my) {
  float rx, ry;
  if (stereo_reference_epiline(g, ux, uy, &rx, &ry)) return -1;
  if (P->win_size != 5) return -1;
  if (!((ux - 2 * rx * rescale_factor) >= 0) || !((ux + 2 * rx * rescale_factor) < cols - 1) ||
      !((uy - 2 * ry * rescale_factor) >= 0) || !((uy + 2 * ry * rescale_factor) < rows - 1))
    return -1;
  float patch[5];
  patch[0] = bil_u8(img_ref, step, ux - 2 * rx * rescale_factor, uy - 2 * ry * rescale_factor);
  patch[1] = bil_u8(img_ref, step, ux - rx * rescale_factor, uy - ry * rescale_factor);
  patch[2] = bil_u8(img_ref, step, ux, uy);
  patch[3] = bil_u8(img_ref, step, ux + rx * rescale_factor, uy + ry * rescale_factor);
  patch[4] = bil_u8(img_ref, step, ux + 2 * rx * rescale_factor, uy + 2 * ry * rescale_factor);
  float gmax = 0.0f;
  for (int i = 1; i < 5; ++i) {
    float a = patch[i] - patch[i - 1];
    a = (a > 0) ? a : -a;
    if (a > gmax) gmax = a;
  }
  if (gmax < P->min_grad_mag) return 1;
  float residual = 3.402823466e+38f;
  const int r = stereo_line_match(P, rescale_factor, patch, img_cmp, rows, cols, step, sx, sy, ex, ey, mx, my, &residual);
  if (r < 0) return -1;
  if (r == 1) return 2;
  if (r == 2) return 3;
  return 0;
}

/* inverse_depth_filter.cc:265-303; 1 = fused, 0 = rejected */
int stereo_fuse(float mu_pred, float var_pred, float mu_meas, float var_meas, float* mu_post, float* var_post,
                float outlier_sigma_thresh) {
  if (!isnan(mu_pred) && (mu_pred > 0.0f)) {
    const float w = var_pred + var_meas;
    *mu_post = (var_meas * mu_pred + var_pred * mu_meas) / w;
    *var_post = (var_pred * var_meas) / w;
  } else {
    *mu_post = mu_meas;
    *var_post = var_meas;
  }
  const float res = mu_meas - mu_pred;
  const float dist = res * res / var_pred;
  if (dist > outlier_sigma_thresh * outlier_sigma_thresh) return 0;
  *mu_post = (*mu_post <= 0) ? 0.0f : *mu_post;
  return 1;
}

/* inverse_depth_meas_model.cc:48-154; 1 ok, 0 no measurement, -1 reference assert.
 * gradx/grady: padded gradient images of the comparison frame, `gstep` floats per row, grows x gcols. */
int stereo_meas_idepth(const stereo_params* P, const stereo_geometry* g, const float* gradx, const float* grady,
                       int grows, int gcols, int gstep, float ux, float uy, float cx, float cy, float* mu, float* var) {
  float ix, iy, ex, ey, disp;
  if (stereo_disparity(g, ux, uy, cx, cy, &ix, &iy, &ex, &ey, &disp)) return -1;
  if ((double)disp < 1e-3) {
    *mu = 0.0f, *var = 1e10f;
    return 0;
  }
  *mu = stereo_disparity_to_idepth(g, ux, uy, ix, iy, ex, ey, disp);
  if (*mu < 0.0f) {
    *mu = 0.0f, *var = 1e10f;
    return 0;
  }
  const float off = (float)(P->z_win_size / 2 + 1);
  if (!bil_ok(grows, gcols, cx + off, cy + off)) return -1;
  const float gx = bil_f32(gradx, gstep, cx + off, cy + off);
  const float gy = bil_f32(grady, gstep, cx + off, cy + off);
  const float gnorm = sqrtf(gx * gx + gy * gy);
  if ((double)gnorm < 1e-3) {
    *mu = 0.0f, *var = 1e10f;
    return 0;
  }
  const float ngx = gx / gnorm, ngy = gy / gnorm;
  const float edn = ngx * ex + ngy * ey;
  const float geo_var = P->epipolar_line_var / (edn * edn);
  if ((double)((edn > 0) ? edn : -edn) < 1e-3) {
    *mu = 0.0f, *var = 1e10f;
    return 0;
  }
  const float edg = gx * ex + gy * ey;
  const float photo_var = 2 * P->pixel_var / (edg * edg);
  const float dmin = disp - disp / 10, dmax = disp + disp / 10;
  const float idmin = stereo_disparity_to_idepth(g, ux, uy, ix, iy, ex, ey, dmin);
  const float idmax = stereo_disparity_to_idepth(g, ux, uy, ix, iy, ex, ey, dmax);
  const float alpha = (idmax - idmin) / (dmax - dmin);
  const float meas_var = alpha * alpha * (geo_var + photo_var);
  if (isnan(meas_var) || isinf(meas_var)) return -1;
  *var = meas_var;
  return 1;
}

/* cv::Rect(x, y, w, h).contains(Point2f): the point converts to Point2i by cvRound (round half to even) */
static int rect_contains(int rx, int ry, int rw, int rh, float px, float py) {
  const long ix = lrintf(px), iy = lrintf(py);
  return rx <= ix && ix < rx + rw && ry <= iy && iy < ry + rh;
}

typedef struct stereo_frame_ref {
  uint32_t id;
  const uint8_t* img_pad;   /* (height + 2 pad) x (width + 2 pad), contiguous */
  float q_to_new[4], t_to_new[3];   /* T_ref_to_new = fnew.pose^-1 * pf.pose   (flame.cc:1315) */
  float q_to_pf[4], t_to_pf[3];     /* T_old_to_new = curr_pf.pose^-1 * pf.pose (flame.cc:1614) */
} stereo_frame_ref;

/* flame.cc:1538-1752.  1 tracked, 0 not, -1 reference assert */
static int track_feature(const stereo_params* P, const float* K, const float* Kinv, const stereo_frame_ref* fr,
                         const stereo_geometry* epigeo, int width, int height, int pad, const uint8_t* new_img_pad,
                         uint32_t curr_pf_id, stereo_feature* f, float* flow_x, float* flow_y) {
  float ucx, ucy, idepth_cmp, var_cmp;
  const int pr = stereo_predict(epigeo, P->process_var_factor, f->x, f->y, f->idepth_mu, f->idepth_var, &ucx, &ucy,
                                &idepth_cmp, &var_cmp);
  if (pr < 0) return -1;
  if (pr != 0) return 0;
  int row_offset = 0;
  if (P->do_letterbox) row_offset = height / 3;
  const int border = (int)(P->rescale_factor_max * P->win_size / 2 + 1);
  const int vx = border, vy = border + row_offset, vw = width - 2 * border, vh = height - 2 * border - 2 * row_offset;
  float rescale = 1.0f;
  if ((f->idepth_mu > 0.0f) && (idepth_cmp > 0.0f)) rescale = idepth_cmp / f->idepth_mu;
  if (isnan(rescale) || !(rescale > 0)) return -1;
  if ((rescale <= P->rescale_factor_min) || (rescale >= P->rescale_factor_max)) {
    stereo_geometry epipf;
    stereo_load_geometry(&epipf, K, Kinv, fr->q_to_pf, fr->t_to_pf);
    float upx, upy, idepth_pf, var_pf;
    const int mr = stereo_predict(&epipf, P->process_var_factor, f->x, f->y, f->idepth_mu, f->idepth_var, &upx, &upy,
                                  &idepth_pf, &var_pf);
    if (mr < 0) return -1;
    if (mr != 0 || !rect_contains(vx, vy, vw, vh, upx, upy)) {
      f->valid = 0;
      return 0;
    }
    f->frame_id = curr_pf_id;
    f->x = upx, f->y = upy;
    const float old = f->idepth_mu;
    f->idepth_mu = idepth_pf;
    float v4 = idepth_pf / old;
    v4 *= v4;
    v4 *= v4;
    if ((double)idepth_pf < 1e-6) v4 = 1;
    f->idepth_var *= v4;
    return 0;
  }
  float sx, sy, ex, ey, epx, epy;
  const int sr = stereo_search_region(P, epigeo, width, height, f->x, f->y, f->idepth_mu, f->idepth_var, &sx, &sy, &ex,
                                      &ey, &epx, &epy);
  if (sr < 0) return -1;
  if (sr == 0) return 0;
  const float off = (float)pad;
  if (!rect_contains(vx, vy, vw, vh, f->x, f->y)) return 0;
  float mx = ucx, my = ucy;
  const int rows = height + 2 * pad, cols = width + 2 * pad;
  const int st = stereo_search(P, epigeo, rescale, fr->img_pad, new_img_pad, rows, cols, cols, f->x + off, f->y + off,
                               sx + off, sy + off, ex + off, ey + off, &mx, &my);
  if (st < 0) return -1;
  f->search_status = st;
  if (st != 0) return 0;
  *flow_x = mx - off, *flow_y = my - off;
  return 1;
}

static void fail_feature(const stereo_params* P, stereo_feature* f, int32_t* stats) {
  f->idepth_var *= P->process_fail_var_factor;
  if (f->idepth_var > P->idepth_var_max) {
    f->valid = 0;
    ++stats[1];
  }
  f->num_dropouts++;
  if (f->num_dropouts > (uint32_t)P->max_dropouts) {
    f->valid = 0;
    ++stats[2];
  }
}

/* flame.cc:1280-1536.  stats[0..5] = num_idepth_updates, num_fail_max_var, num_fail_max_dropouts,
 * num_fail_ref_patch_grad, num_fail_ambiguous_match, num_fail_max_cost; stats[6] = the function's bool.
 * Returns 0, -(1 + i) when feature i hits a reference assert, or 1 + i when its frame id is unknown
 * (pfs.at() would throw). */
long stereo_update_feature_idepths(const stereo_params* P, const float* K, const float* Kinv, int width, int height,
                                   int pad, int n_frames, const stereo_frame_ref* frames, const uint8_t* new_img_pad,
                                   const float* new_gradx_pad, const float* new_grady_pad, uint32_t curr_pf_id, int n,
                                   stereo_feature* feats, int32_t* stats) {
  for (int k = 0; k < 7; ++k) stats[k] = 0;
  const int rows = height + 2 * pad, cols = width + 2 * pad;
  for (int i = 0; i < n; ++i) {
    stereo_feature* f = &feats[i];
    const stereo_frame_ref* fr = 0;
    for (int k = 0; k < n_frames; ++k)
      if (frames[k].id == f->frame_id) {
        fr = &frames[k];
        break;
      }
    if (!fr) return 1 + i;
    stereo_geometry epigeo;
    stereo_load_geometry(&epigeo, K, Kinv, fr->q_to_new, fr->t_to_new);
    const float* t = fr->t_to_new;
    const float baseline = sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
    if (baseline < P->min_baseline) continue;
    float flow_x = 0, flow_y = 0;
    const int tr = track_feature(P, K, Kinv, fr, &epigeo, width, height, pad, new_img_pad, curr_pf_id, f, &flow_x, &flow_y);
    if (tr < 0) return -(1 + (long)i);
    if (f->search_status == 1) ++stats[3];
    else if (f->search_status == 2) ++stats[4];
    else if (f->search_status == 3) ++stats[5];
    if (tr == 0) {
      fail_feature(P, f, stats);
      continue;
    }
    /* the measurement model reloads the same geometry from the frame the feature belongs to (flame.cc:1381-1383) */
    float mu_meas, var_meas;
    const int sr = stereo_meas_idepth(P, &epigeo, new_gradx_pad, new_grady_pad, rows, cols, cols, f->x, f->y, flow_x,
                                      flow_y, &mu_meas, &var_meas);
    if (sr < 0) return -(1 + (long)i);
    if (sr == 0) {
      fail_feature(P, f, stats);
      continue;
    }
    float mu_post, var_post;
    if (!stereo_fuse(f->idepth_mu, f->idepth_var, mu_meas, var_meas, &mu_post, &var_post, P->outlier_sigma_thresh)) {
      fail_feature(P, f, stats);
      continue;
    }
    if (isnan(mu_post) || isnan(var_post) || !(var_post >= 0)) return -(1 + (long)i);
    if (P->do_meas_fusion) {
      f->idepth_mu = mu_post, f->idepth_var = var_post;
    } else {
      f->idepth_mu = mu_meas, f->idepth_var = var_meas;
    }
    f->valid = 1;
    f->num_updates++;
    f->num_dropouts = 0;
    ++stats[0];
    stats[6] = 1;
  }
  return 0;
}

/* utils::Frame::create, level 0 (frame.cc:33-71): padded image (BORDER_REFLECT_101) and padded central
 * gradients (BORDER_CONSTANT 0).  getCentralGradient<uint8_t,float>: image_utils.h:425-470. */
void stereo_make_frame(const uint8_t* img, int width, int height, int border, uint8_t* img_pad, float* gradx_pad,
                       float* grady_pad) {
  const int pw = width + 2 * border, ph = height + 2 * border;
  for (int y = 0; y < ph; ++y) {
    int sy = y - border;
    if (sy < 0) sy = -sy;
    if (sy >= height) sy = 2 * (height - 1) - sy;
    for (int x = 0; x < pw; ++x) {
      int sx = x - border;
      if (sx < 0) sx = -sx;
      if (sx >= width) sx = 2 * (width - 1) - sx;
      img_pad[(long)y * pw + x] = img[(long)sy * width + sx];
    }
  }
  memset(gradx_pad, 0, sizeof(float) * (size_t)pw * ph);
  memset(grady_pad, 0, sizeof(float) * (size_t)pw * ph);
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      float gx, gy;
      if (x == 0) gx = (float)img[(long)y * width + 1] - (float)img[(long)y * width];
      else if (x == width - 1) gx = (float)img[(long)y * width + width - 1] - (float)img[(long)y * width + width - 2];
      else gx = (float)(0.5 * (double)((float)img[(long)y * width + x + 1] - (float)img[(long)y * width + x - 1]));
      if (y == 0) gy = (float)img[width + x] - (float)img[x];
      else if (y == height - 1) gy = (float)img[(long)(height - 1) * width + x] - (float)img[(long)(height - 2) * width + x];
      else gy = (float)(0.5 * (double)((float)img[(long)(y + 1) * width + x] - (float)img[(long)(y - 1) * width + x]));
      gradx_pad[(long)(y + border) * pw + x + border] = gx;
      grady_pad[(long)(y + border) * pw + x + border] = gy;
    }
}
