// oracle/ref_delaunay.cc -- TEST INFRASTRUCTURE (fixture generation only; never shipped logic).
//
// Thin extern "C" driver around the reference's vendored Triangle
// (/root/reference/src/flame/external/triangle/triangle.{h,cpp}), which is compiled IN PLACE
// from its own single source file by oracle/Makefile into oracle/_ref/ (it needs nothing the
// image lacks).  The call below is the one the reference makes in
// src/flame/utils/delaunay.cc:33-68 (switches "zneQB"); the edge list is read as
// delaunay.cc:125-133 does, so edge k = (edgelist[2k], edgelist[2k+1]) has exactly the order and
// (source,target) orientation that flame.cc:2085-2096 hands to boost::add_edge.
//
// Triangle is NOT on the solver hot path; this exists so that the golden fixtures under
// tests/golden/ carry reference-faithful Delaunay edge lists.
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "flame/external/triangle/triangle.h"

extern "C" {

// pts: 2*n floats (x,y interleaved).  Returns number of edges, or -1 on error.
// *edges_out / *tris_out are malloc'ed (2*E / 3*T ints); free with ref_delaunay_free().
int ref_delaunay(const float* pts, int n, int** edges_out, int* n_tris, int** tris_out) {
  if (n < 3 || !pts || !edges_out) return -1;
  struct triangulateio in, out;
  std::memset(&in, 0, sizeof(in));
  std::memset(&out, 0, sizeof(out));

  in.numberofpoints = n;
  in.pointlist = static_cast<float*>(std::malloc(sizeof(float) * 2 * static_cast<size_t>(n)));
  std::memcpy(in.pointlist, pts, sizeof(float) * 2 * static_cast<size_t>(n));

  char parameters[] = "zneQB";
  ::triangulate(parameters, &in, &out, NULL);
  std::free(in.pointlist);

  const int E = out.numberofedges;
  *edges_out = static_cast<int*>(std::malloc(sizeof(int) * 2 * static_cast<size_t>(E > 0 ? E : 1)));
  std::memcpy(*edges_out, out.edgelist, sizeof(int) * 2 * static_cast<size_t>(E));
  if (n_tris) *n_tris = out.numberoftriangles;
  if (tris_out) {
    const int T = out.numberoftriangles;
    *tris_out = static_cast<int*>(std::malloc(sizeof(int) * 3 * static_cast<size_t>(T > 0 ? T : 1)));
    std::memcpy(*tris_out, out.trianglelist, sizeof(int) * 3 * static_cast<size_t>(T));
  }
  std::free(out.pointlist);
  std::free(out.trianglelist);
  std::free(out.edgelist);
  std::free(out.neighborlist);
  return E;
}

void ref_delaunay_free(int* p) { std::free(p); }

}  // extern "C"
