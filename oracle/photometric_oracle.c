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

/* ------------------------------------------------------------------------------------------------------
 * Graph maintenance pieces of SURVEY.md 8(f) rank 1 (test infrastructure; UNPINNED: Eigen is not
 * available to check the quaternion product against, and the reference has no test for these):
 *   Flame::projectGraph            /root/reference/src/flame/flame.cc:1862-1938 (re-projection + the
 *                                   keep/remove decision; do_grad_check_after_projection = false)
 *   EpipolarGeometry::project(u_ref, idepth, &u_cmp, &new_idepth)   stereo/epipolar_geometry.h:152-180
 *   Eigen::Quaternion * Vector3 (_transformVector): uv = q.vec x v; uv += uv; v + q.w*uv + q.vec x uv
 *   rescale_data block             flame.cc:328-351
 * q = (w, x, y, z) of q_ref_to_cmp; K, Kinv row-major 3x3 (only the entries the reference reads are used).
 * ------------------------------------------------------------------------------------------------------ */
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

void graph_project(int V, float* pos, float* x, float graph_scale, const float* K, const float* Kinv, const float* q,
                   const float* t, const float* KRKinv, float rx, float ry, float rw, float rh, uint8_t* keep) {
  for (int v = 0; v < V; ++v) {
    const float ux = pos[2 * v], uy = pos[2 * v + 1];
    const float idepth = x[v] * graph_scale;
    float nx, ny, nid;
    if (idepth == 0.0f) { /* h:155-161: maxDepthProjection */
      const float h0 = (KRKinv[0] * ux + KRKinv[1] * uy) + KRKinv[2] * 1.0f;
      const float h1 = (KRKinv[3] * ux + KRKinv[4] * uy) + KRKinv[5] * 1.0f;
      const float h2 = (KRKinv[6] * ux + KRKinv[7] * uy) + KRKinv[8] * 1.0f;
      const float inv = 1.0f / h2;
      nx = h0 * inv, ny = h1 * inv, nid = 0.0f;
    } else {
      const float depth = 1.0f / idepth;
      float p_ref[3] = {Kinv[0] * ux + Kinv[2], Kinv[4] * uy + Kinv[5], 1.0f};
      p_ref[0] *= depth, p_ref[1] *= depth, p_ref[2] *= depth;
      float r[3];
      quat_rotate(q, p_ref, r);
      const float pc0 = r[0] + t[0], pc1 = r[1] + t[1], pc2 = r[2] + t[2];
      const float u0 = K[0] * pc0 + K[2] * pc2, u1 = K[4] * pc1 + K[5] * pc2;
      nid = 1.0f / pc2;
      nx = u0 * nid, ny = u1 * nid;
    }
    pos[2 * v] = nx, pos[2 * v + 1] = ny;
    x[v] = nid / graph_scale; /* flame.cc:1899-1900 */
    /* cv::Rect_<float>::contains: x <= pt.x < x + width, same for y (flame.cc:1902) */
    const int inside = rx <= nx && nx < rx + rw && ry <= ny && ny < ry + rh;
    keep[v] = (uint8_t)(inside && !(nid < 0.0f));
  }
}

/* The fixed summation order of the device's k_block_sum (flame_amd/csrc/nltgv2_kernels.hip): 1024 partial sums over the elements
 * t, t + 1024, ..., each sequential in float, combined pairwise p[t] += p[t + s], s = 512 ... 1.  Used where the reference's own
 * order is unspecified (it walks a hash set). */
float strided_tree_sum(int n, const float* in, float scale) {
  float p[1024];
  for (int t = 0; t < 1024; ++t) {
    float sum = 0.0f;
    for (int i = t; i < n; i += 1024) sum += in[i] * scale;
    p[t] = sum;
  }
  for (int s = 512; s > 0; s >>= 1)
    for (int t = 0; t < s; ++t) p[t] += p[t + s];
  return p[0];
}

/* flame.cc:328-351; the reference sums over BGL's hash order (unspecified): here in the order of strided_tree_sum. */
float graph_rescale(int V, float* x, float* x_bar, float* x_prev, float* data_term, float graph_scale,
                    float* data_factor) {
  const float idepth_sum = strided_tree_sum(V, data_term, graph_scale);
  const float new_scale = idepth_sum / V;
  for (int v = 0; v < V; ++v) {
    x[v] = x[v] * graph_scale / new_scale;
    x_bar[v] = x_bar[v] * graph_scale / new_scale;
    x_prev[v] = x_prev[v] * graph_scale / new_scale;
    data_term[v] = data_term[v] * graph_scale / new_scale;
  }
  *data_factor *= new_scale / graph_scale;
  return new_scale;
}
