/*
 * oracle/photometric_oracle.c -- TEST INFRASTRUCTURE (CPU checker) for the per-vertex photometric
 * residual of BASELINE config 5.
 *
 * The residual itself has NO live reference code: the only occurrence is the commented-out block
 * /root/reference/src/flame/flame.cc:854-893 ("Compute photo error").  Its two building blocks ARE
 * live and ARE pinned by the reference's own known-answer tests, which tests/test_photometric.py
 * replays against this file:
 *   EpipolarGeometry::project(u_ref, idepth)   src/flame/stereo/epipolar_geometry.h:127-143
 *     (+ maxDepthProjection h:191-201)         KAT: test/stereo/epipolar_geometry_test.cc:773-806
 *   utils::bilinearInterp<uint8_t,float>       src/flame/utils/image_utils.h:199-214, 230-255
 *                                              KAT: test/utils/image_utils_test.cc:150-166
 * Matrix-vector products follow Eigen's fixed-size evaluation ((m0*v0 + m1*v1) + m2*v2, then + Kt).
 *
 * Residual definition (restating flame.cc:868-893 for mesh vertices instead of pixels):
 *   idepth = x[v] * graph_scale (flame.cc:377);  u_cmp = project(pos[v], idepth);
 *   err[v] = | bilinear(I_cmp, u_cmp) - bilinear(I_ref, pos[v]) |, NaN when idepth is NaN or < 0,
 *   or when pos[v] / u_cmp fall outside [border, cols-border) x [border, rows-border).
 *   (The dead code tests u_cmp with cv::Rect::contains after rounding to int; here the comparison is
 *   done on the float coordinates -- stated deviation, OpenCV is not available to check against.)
 * The residual never feeds back into x: solver parity of config 5 is unaffected by it.
 */
#include <math.h>
#include <stdint.h>

/* KRKinv row-major 3x3, Kt 3-vector (EpipolarGeometry::loadGeometry, h:88-93). */
void photo_project(const float* KRKinv, const float* Kt, float ux, float uy, float idepth, float* cx,
                   float* cy) {
  float h0, h1, h2;
  if (idepth == 0.0f) { /* maxDepthProjection, h:191-201 */
    h0 = (KRKinv[0] * ux + KRKinv[1] * uy) + KRKinv[2] * 1.0f;
    h1 = (KRKinv[3] * ux + KRKinv[4] * uy) + KRKinv[5] * 1.0f;
    h2 = (KRKinv[6] * ux + KRKinv[7] * uy) + KRKinv[8] * 1.0f;
  } else { /* h:135-142 */
    const float depth = 1.0f / idepth;
    const float a = ux * depth, b = uy * depth, c = depth;
    h0 = ((KRKinv[0] * a + KRKinv[1] * b) + KRKinv[2] * c) + Kt[0];
    h1 = ((KRKinv[3] * a + KRKinv[4] * b) + KRKinv[5] * c) + Kt[1];
    h2 = ((KRKinv[6] * a + KRKinv[7] * b) + KRKinv[8] * c) + Kt[2];
  }
  const float inv = 1.0f / h2;
  *cx = h0 * inv;
  *cy = h1 * inv;
}

/* image_utils.h:199-214 and :230-255 */
float photo_bilinear_u8(const uint8_t* data, int step, float x, float y) {
  const int x_floor = (int)x, y_floor = (int)y;
  const float dx = x - x_floor, dy = y - y_floor;
  const float w11 = dx * dy;
  const float w01 = dx - w11;
  const float w10 = dy - w11;
  const float w00 = 1.0f - dx - dy + w11;
  const uint8_t* p = data + (long)y_floor * step + x_floor;
  return w00 * p[0] + w01 * p[1] + w10 * p[step] + w11 * p[1 + step];
}

static int inside(float x, float y, int rows, int cols, int border) {
  return x >= (float)border && y >= (float)border && x < (float)(cols - border) && y < (float)(rows - border);
}

void photo_residual(int V, const float* pos, const float* x, float graph_scale, const float* KRKinv,
                    const float* Kt, const uint8_t* ref, const uint8_t* cmp, int rows, int cols, int step,
                    int border, float* err) {
  for (int v = 0; v < V; ++v) {
    err[v] = NAN;
    const float idepth = x[v] * graph_scale;
    const float ux = pos[2 * v], uy = pos[2 * v + 1];
    if (isnan(idepth) || idepth < 0.0f || !inside(ux, uy, rows, cols, border)) continue;
    float cx, cy;
    photo_project(KRKinv, Kt, ux, uy, idepth, &cx, &cy);
    if (!(cx == cx) || !(cy == cy) || !inside(cx, cy, rows, cols, border)) continue;
    const float a = photo_bilinear_u8(cmp, step, cx, cy);
    const float b = photo_bilinear_u8(ref, step, ux, uy);
    const float d = a - b;
    err[v] = (d > 0) ? d : -d; /* utils::fast_abs */
  }
}
