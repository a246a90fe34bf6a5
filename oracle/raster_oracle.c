/*
 * oracle/raster_oracle.c -- TEST INFRASTRUCTURE (CPU checker) for the mesh -> dense inverse-depth
 * rasterisation that follows the solver every frame:
 *   utils::interpolateMesh                    /root/reference/src/flame/utils/image_utils.cc:373-396
 *   utils::DrawShadedTriangleBarycentric      /root/reference/src/flame/utils/rasterization.cc:164-246
 *   utils::Edge::init                         /root/reference/src/flame/utils/rasterization.h:120-154
 *   call site                                 /root/reference/src/flame/flame.cc:409-437 (+ coverage)
 *
 * PINNED by the reference's own fixtures (tests/test_raster.py):
 *   test/data/RasterizationTest_DrawShadedTriangleBarycentric{1,2}.png, compared byte for byte as
 *   test/utils/rasterization_test.cc:514-598 does (after the test's normalize + 8-bit conversion);
 *   ImageUtilsTest.interpolateMeshTest, test/utils/image_utils_test.cc:754-785.
 *
 * The reference rasteriser is 4-wide SSE: per 4-pixel block it evaluates the three edge functions,
 * masks (w1>=0 && w2>=0 && w3>=0) and blends.  All edge values are integers held in floats (exact
 * below 2^24), so evaluating A*x + B*y + C per pixel gives the same floats as the reference's
 * incremental stepping; the value uses the reference's association:
 *   (v1*w1 + (v2*w2 + v3*w3)) / (w1 + (w2 + w3)).
 * Pixels are only ever written where the mask is true (the SSE code rewrites masked-out lanes with
 * the value they already had).
 */
#include <math.h>
#include <stdint.h>

static int imin3(int x, int y, int z) { return x < y ? (x < z ? x : z) : (y < z ? y : z); }
static int imax3(int x, int y, int z) { return x > y ? (x > z ? x : z) : (y > z ? y : z); }

/* Edge::init, h:128-153: A = v1.y - v0.y, B = v0.x - v1.x, C = v1.x*v0.y - v0.x*v1.y (int -> float). */
static float edge_eval(int v0x, int v0y, int v1x, int v1y, int px, int py) {
  const float A = (float)(v1y - v0y);
  const float B = (float)(v0x - v1x);
  const float C = (float)(v1x * v0y - v0x * v1y);
  return (A * (float)px + B * (float)py) + C;
}

/* rasterization.cc:164-246.  img: rows x cols floats, row-major, contiguous. */
void raster_triangle_barycentric(int p1x, int p1y, int p2x, int p2y, int p3x, int p3y, float v1, float v2,
                                 float v3, float* img, int rows, int cols) {
  const int xmin = imin3(p1x, p2x, p3x), ymin = imin3(p1y, p2y, p3y);
  const int xmax = imax3(p1x, p2x, p3x), ymax = imax3(p1y, p2y, p3y);
  for (int y = ymin; y <= ymax; ++y) {
    /* the reference walks x in blocks of 4 from xmin: lanes beyond xmax are evaluated too */
    const int xlast = xmin + ((xmax - xmin) / 4) * 4 + 3;
    for (int x = xmin; x <= xlast; ++x) {
      const float w1 = edge_eval(p2x, p2y, p3x, p3y, x, y); /* e23 */
      const float w2 = edge_eval(p3x, p3y, p1x, p1y, x, y); /* e31 */
      const float w3 = edge_eval(p1x, p1y, p2x, p2y, x, y); /* e12 */
      if (w1 >= 0.0f && w2 >= 0.0f && w3 >= 0.0f) {
        const float norm = w1 + (w2 + w3);
        const float val = (v1 * w1 + (v2 * w2 + v3 * w3)) / norm;
        if (x >= 0 && y >= 0 && x < cols && y < rows) img[(long)y * cols + x] = val;
      }
    }
  }
}

/* cv::Point2f -> cv::Point: saturate_cast<int>(float) == cvRound == round half to even. */
static int cv_round(float v) { return (int)lrintf(v); }

/* image_utils.cc:373-396: triangles in order, later triangles overwrite shared pixels. */
void raster_interpolate_mesh(int T, const int32_t* tris, const float* vtx_xy, const float* values,
                             const uint8_t* vtx_valid, const uint8_t* tri_valid, float* img, int rows, int cols) {
  for (int t = 0; t < T; ++t) {
    const int a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    if ((tri_valid && !tri_valid[t]) || (vtx_valid && !(vtx_valid[a] && vtx_valid[b] && vtx_valid[c]))) continue;
    /* "Triangle spits out points in clockwise order, but drawing function expects CCW": (2,1,0) */
    raster_triangle_barycentric(cv_round(vtx_xy[2 * c]), cv_round(vtx_xy[2 * c + 1]), cv_round(vtx_xy[2 * b]),
                                cv_round(vtx_xy[2 * b + 1]), cv_round(vtx_xy[2 * a]), cv_round(vtx_xy[2 * a + 1]),
                                values[c], values[b], values[a], img, rows, cols);
  }
}

/* flame.cc:428-437 */
int raster_coverage(const float* img, int rows, int cols) {
  int n = 0;
  for (long i = 0; i < (long)rows * cols; ++i) n += !isnan(img[i]);
  return n;
}
