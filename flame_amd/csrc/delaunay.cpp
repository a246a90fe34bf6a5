// delaunay.cpp -- 2-D Delaunay triangulation of float32 points (host code, like the reference's):
// the counterpart of utils::Delaunay (/root/reference/src/flame/utils/delaunay.{h,cc}), which wraps the
// vendored Shewchuk Triangle with switches "zneQB" (delaunay.cc:66-68) and hands Flame::syncGraph the edge
// list the solver's graph is built from (flame.cc:2073-2104) and the triangles interpolateMesh draws
// (flame.cc:409-415).  SURVEY.md section 8(f) rank 3.
//
// Own implementation, not a port: incremental Bowyer-Watson with ghost triangles (no super-triangle, so
// the convex hull is exact), points inserted along a Morton curve and located by a visibility walk from
// the previous insertion.  Predicates are EXACT: every float32 coordinate is an integer multiple of a
// common power of two, so coordinates are rescaled to int64 once; orient2d is then exact in __int128 and
// incircle in 256-bit integer arithmetic, behind a double-precision filter with a forward error bound.
// A point set in general position has ONE Delaunay triangulation, so the result equals Triangle's as a
// set of triangles (pinned in tests/test_delaunay.py against triangulations produced by the reference's
// Triangle, oracle/_ref); order and rotation of the output triangles / edges are this implementation's
// own, which is fine because the solver consumes explicit edge lists.
//
// Output conventions (matching Triangle's): triangles counter-clockwise in x-right / y-up coordinates;
// edges unique and undirected, listed as first met walking the triangles (v0,v1),(v1,v2),(v2,v0).
// Exact duplicates of an earlier point are skipped (Triangle ignores them as well).
//
// Large inputs (>= 4096 points) are triangulated in parallel (round 3: the triangulation had become 60 % of a frame):
// the points are cut into vertical strips of equal counts; every strip triangulates, on its own thread and with the same
// exact algorithm, the points of its x range widened by a halo plus the points near the top / bottom of the bounding box
// over the whole width, and reports the triangles whose leftmost vertex lies in its own range -- each after a CERTIFICATE
// that it is a triangle of the global Delaunay triangulation: its circumdisk (for a hull edge: the outer half-plane) does
// not reach into the part of the bounding box whose points the strip did not look at.  A strip that cannot certify a
// triangle retries with a three times wider halo; if that fails too, or if the strips' counts do not add up to a
// triangulation (Euler), the whole input is triangulated sequentially.  The number of strips depends on the number of points
// only, never on the machine: the output is the same whatever the thread count.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

#include "flame_nltgv2.h"
#include "host_workers.hpp"
#include "roctx_ranges.hpp"

namespace {

typedef __int128 i128;

// ---- 256-bit signed accumulation of products of two i128 (incircle needs |value| < 2^240) --------------
struct I256 {
  uint64_t w[4];  // little endian, two's complement
};

inline I256 from_product(i128 a, i128 b) {
  const bool neg = (a < 0) != (b < 0);
  unsigned __int128 ua = a < 0 ? (unsigned __int128)(-a) : (unsigned __int128)a;
  unsigned __int128 ub = b < 0 ? (unsigned __int128)(-b) : (unsigned __int128)b;
  const uint64_t a0 = (uint64_t)ua, a1 = (uint64_t)(ua >> 64), b0 = (uint64_t)ub, b1 = (uint64_t)(ub >> 64);
  unsigned __int128 p00 = (unsigned __int128)a0 * b0, p01 = (unsigned __int128)a0 * b1;
  unsigned __int128 p10 = (unsigned __int128)a1 * b0, p11 = (unsigned __int128)a1 * b1;
  I256 r;
  r.w[0] = (uint64_t)p00;
  unsigned __int128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
  r.w[1] = (uint64_t)mid;
  unsigned __int128 hi = (mid >> 64) + (p01 >> 64) + (p10 >> 64) + (uint64_t)p11;
  r.w[2] = (uint64_t)hi;
  r.w[3] = (uint64_t)((hi >> 64) + (p11 >> 64));
  if (neg) {  // two's complement negate
    unsigned __int128 c = 1;
    for (int i = 0; i < 4; ++i) {
      c += (uint64_t)~r.w[i];
      r.w[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}

inline void add_to(I256& a, const I256& b) {
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (unsigned __int128)a.w[i] + b.w[i];
    a.w[i] = (uint64_t)c;
    c >>= 64;
  }
}

inline int sign_of(const I256& a) {
  if (a.w[3] >> 63) return -1;
  return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) ? 1 : 0;
}

struct Pt {
  int64_t x, y;  // exact integer image of the float32 coordinates (common power-of-two scale)
  double fx, fy; // the same values as doubles, for the incircle filter
};

const int GHOST = -1;

struct alignas(32) Tri {
  int v[3];
  int n[3];         // neighbour across the edge opposite v[i]
  uint8_t alive;
  uint8_t ghost;    // one of v[] is GHOST
  uint8_t in_cavity;  // scratch of insert(), reset after each insertion
  uint8_t pad_[5];
};

class Triangulator {
 public:
  std::vector<Pt> p;
  std::vector<Tri> t;
  std::vector<int> cavity, stack_;
  bool filter_ok = false;       // differences of coordinates are exact in double
  // the filters' error bounds with every coordinate difference replaced by the extent D of the point set:
  // orient: 4e-16 (|l| + |r|) <= 4e-16 * 2 D^2;  incircle: 1.2e-15 * permanent <= 1.2e-15 * 12 D^4
  double ori_static = 0.0, icc_static = 0.0;

  int orient(int a, int b, int c) const {  // > 0: c to the left of a->b (counter-clockwise), exact
    if (filter_ok) {  // the differences are exact in double; only the two products and their difference round
      const double l = (p[b].fx - p[a].fx) * (p[c].fy - p[a].fy), r = (p[b].fy - p[a].fy) * (p[c].fx - p[a].fx);
      const double det = l - r;
      if (det > ori_static) return 1;  // (the a-priori bound over the whole point set: no need to form this call's own)
      if (det < -ori_static) return -1;
      const double bound = 4.0e-16 * (std::fabs(l) + std::fabs(r));  // > (3 + 16 eps) eps, Shewchuk's ccwerrboundA
      if (det > bound) return 1;
      if (det < -bound) return -1;
    }
    const i128 d = (i128)(p[b].x - p[a].x) * (p[c].y - p[a].y) - (i128)(p[b].y - p[a].y) * (p[c].x - p[a].x);
    return d > 0 ? 1 : (d < 0 ? -1 : 0);
  }

  int incircle(int a, int b, int c, int d) const {  // > 0: d strictly inside the circle through ccw a,b,c
    if (filter_ok) {
      const double adx = p[a].fx - p[d].fx, ady = p[a].fy - p[d].fy;
      const double bdx = p[b].fx - p[d].fx, bdy = p[b].fy - p[d].fy;
      const double cdx = p[c].fx - p[d].fx, cdy = p[c].fy - p[d].fy;
      const double bdxcdy = bdx * cdy, cdxbdy = cdx * bdy, cdxady = cdx * ady, adxcdy = adx * cdy;
      const double adxbdy = adx * bdy, bdxady = bdx * ady;
      const double alift = adx * adx + ady * ady, blift = bdx * bdx + bdy * bdy, clift = cdx * cdx + cdy * cdy;
      const double det = alift * (bdxcdy - cdxbdy) + blift * (cdxady - adxcdy) + clift * (adxbdy - bdxady);
      if (det > icc_static) return 1;
      if (det < -icc_static) return -1;
      const double permanent = (std::fabs(bdxcdy) + std::fabs(cdxbdy)) * alift +
                               (std::fabs(cdxady) + std::fabs(adxcdy)) * blift +
                               (std::fabs(adxbdy) + std::fabs(bdxady)) * clift;
      const double errbound = 1.2e-15 * permanent;  // > (10 + 96 eps) eps, Shewchuk's iccerrboundA
      if (det > errbound) return 1;
      if (det < -errbound) return -1;
    }
    const i128 adx = (i128)p[a].x - p[d].x, ady = (i128)p[a].y - p[d].y;
    const i128 bdx = (i128)p[b].x - p[d].x, bdy = (i128)p[b].y - p[d].y;
    const i128 cdx = (i128)p[c].x - p[d].x, cdy = (i128)p[c].y - p[d].y;
    I256 s = from_product(adx * adx + ady * ady, bdx * cdy - cdx * bdy);
    add_to(s, from_product(bdx * bdx + bdy * bdy, cdx * ady - adx * cdy));
    add_to(s, from_product(cdx * cdx + cdy * cdy, adx * bdy - bdx * ady));
    return sign_of(s);
  }

  // Bowyer-Watson membership: is point d inside the (open) circumdisk of triangle ti?  For a ghost triangle
  // (a, b, GHOST) the "disk" is the open half-plane to the left of a->b plus the open segment ab.
  bool in_disk(int ti, int d) const {
    const Tri& T = t[ti];
    if (!T.ghost) return incircle(T.v[0], T.v[1], T.v[2], d) > 0;
    for (int i = 0; i < 3; ++i) {
      if (T.v[i] == GHOST) {
        const int a = T.v[(i + 1) % 3], b = T.v[(i + 2) % 3];
        const int o = orient(a, b, d);
        if (o > 0) return true;
        if (o < 0) return false;
        // collinear: inside iff strictly between a and b
        const i128 dot1 = (i128)(p[d].x - p[a].x) * (p[b].x - p[a].x) + (i128)(p[d].y - p[a].y) * (p[b].y - p[a].y);
        const i128 dot2 = (i128)(p[d].x - p[b].x) * (p[a].x - p[b].x) + (i128)(p[d].y - p[b].y) * (p[a].y - p[b].y);
        return dot1 > 0 && dot2 > 0;
      }
    }
    return incircle(T.v[0], T.v[1], T.v[2], d) > 0;
  }

  int new_tri(int a, int b, int c) {
    Tri T;
    T.v[0] = a, T.v[1] = b, T.v[2] = c;
    T.n[0] = T.n[1] = T.n[2] = -1;
    T.alive = 1;
    T.ghost = (a == GHOST || b == GHOST || c == GHOST) ? 1 : 0;
    T.in_cavity = 0;
    if (!free_.empty()) {  // reuse the slot of a removed triangle: the working set stays ~2n triangles
      const int i = free_.back();
      free_.pop_back();
      t[(size_t)i] = T;
      return i;
    }
    t.push_back(T);
    return (int)t.size() - 1;
  }

  static bool is_ghost(const Tri& T) { return T.ghost != 0; }

  // Visibility walk from a real triangle; returns a triangle (real or ghost) whose disk contains d, or -1
  // when d coincides with an existing vertex.
  int locate(int start, int d) const {
    int cur = start;
    if (is_ghost(t[cur])) {
      for (int i = 0; i < 3; ++i)
        if (t[cur].v[i] == GHOST) cur = t[cur].n[i];
    }
    int from = -1;  // the triangle the walk came from: d is on this side of the shared edge, no need to test it again
    for (size_t guard = 0; guard < t.size() + 8; ++guard) {
      const Tri& T = t[cur];
      bool moved = false;
      for (int i = 0; i < 3; ++i) {
        if (T.n[i] == from) continue;
        const int a = T.v[(i + 1) % 3], b = T.v[(i + 2) % 3];
        if (orient(a, b, d) < 0) {
          const int nb = T.n[i];
          if (is_ghost(t[nb])) return nb;
          from = cur;
          cur = nb;
          moved = true;
          break;
        }
      }
      if (!moved) return cur;
    }
    return cur;
  }

  // Inserts point d; `seed` is a triangle whose disk contains d.  Returns one of the new real triangles.
  int insert(int seed, int d) {
    cavity.clear();
    stack_.clear();
    stack_.push_back(seed);
    t[seed].in_cavity = 1;
    while (!stack_.empty()) {
      const int ti = stack_.back();
      stack_.pop_back();
      cavity.push_back(ti);
      for (int i = 0; i < 3; ++i) {
        const int nb = t[ti].n[i];
        if (nb >= 0 && !t[nb].in_cavity && in_disk(nb, d)) {
          t[nb].in_cavity = 1;
          stack_.push_back(nb);
        }
      }
    }
    // boundary edges (a -> b as seen from inside the cavity, cavity on the left) and their outer neighbours
    be.clear();
    for (int ti : cavity) {
      for (int i = 0; i < 3; ++i) {
        const int nb = t[ti].n[i];
        if (nb < 0 || !t[nb].in_cavity) be.push_back(BE{t[ti].v[(i + 1) % 3], t[ti].v[(i + 2) % 3], nb, -1});
      }
    }
    for (int ti : cavity) {  // retire the cavity (their slots are reused by the fan below)
      t[ti].alive = 0;
      t[ti].in_cavity = 0;
      free_.push_back(ti);
    }
    for (BE& e : be) {
      e.tri = new_tri(e.a, e.b, d);  // orientation inherited from the removed triangle: (a, b, apex) ccw
      Tri& T = t[e.tri];
      T.n[2] = e.outer;  // across a-b (opposite d)
      if (e.outer >= 0) {
        Tri& O = t[e.outer];
        for (int i = 0; i < 3; ++i) {
          const int oa = O.v[(i + 1) % 3], ob = O.v[(i + 2) % 3];
          if (oa == e.b && ob == e.a) O.n[i] = e.tri;
        }
      }
    }
    // link the fan: triangle (a, b, d) meets, across edge b-d (opposite a, index 0), the triangle whose a == b;
    // across edge d-a (opposite b, index 1), the triangle whose b == a.
    // Vertex ids include GHOST (-1): index by id + 1.
    if (fan_next.size() < p.size() + 1) fan_next.assign(p.size() + 1, -1);
    for (const BE& e : be) fan_next[e.a + 1] = e.tri;
    for (const BE& e : be) {
      const int nxt = fan_next[e.b + 1];  // triangle starting at b
      t[e.tri].n[0] = nxt;
      t[nxt].n[1] = e.tri;
    }
    for (const BE& e : be) fan_next[e.a + 1] = -1;
    for (const BE& e : be)
      if (!is_ghost(t[e.tri])) return e.tri;
    return be.empty() ? seed : be[0].tri;
  }

  struct BE { int a, b, outer, tri; };
  std::vector<BE> be;
  std::vector<int> fan_next, free_;
};

inline uint32_t spread16(uint32_t v) {
  v &= 0xFFFFu;
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}


// What every (sub)triangulation of one input shares: the exact integer image of the coordinates and the bounding box.
struct Input {
  const float* xy = nullptr;
  int32_t n = 0;
  double to_int = 1.0;
  bool filter_ok = false;
  float minx = 0, maxx = 0, miny = 0, maxy = 0;
  double ori_static = 0.0, icc_static = 0.0;
};

// Triangulates the points ids[0..m) of `in` (global ids, ascending within equal Morton keys): T's vertex k is ids[k].
// Returns false when the subset has no triangle (fewer than three distinct points, or all collinear).
bool triangulate_subset(const Input& in, const int* ids, int m, Triangulator& T) {
  const float* xy = in.xy;
  T.filter_ok = in.filter_ok, T.ori_static = in.ori_static, T.icc_static = in.icc_static;
  T.p.resize((size_t)m);
  for (int k = 0; k < m; ++k) {
    const int i = ids[k];
    T.p[(size_t)k].x = (int64_t)((double)xy[2 * i] * in.to_int);  // exact: a power-of-two scale, |result| < 2^58
    T.p[(size_t)k].y = (int64_t)((double)xy[2 * i + 1] * in.to_int);
    T.p[(size_t)k].fx = (double)xy[2 * i];
    T.p[(size_t)k].fy = (double)xy[2 * i + 1];
  }
  // ---- insertion order: Morton curve (consecutive points are close: short walks).  The keys are quantised over the
  // bounding box of the WHOLE input, and ties keep the order of `ids`: two points are inserted in the same relative order
  // in every subset that holds both (what breaks co-circular ties the same way in neighbouring strips)
  // (scratch kept per thread between calls: a strip's ~150 KB of fresh vectors every frame were an mmap, its page faults and a munmap each)
  static thread_local std::vector<int> order, tmp;
  static thread_local std::vector<uint32_t> key, count;
  static thread_local std::vector<char> used;
  order.resize((size_t)m);
  std::iota(order.begin(), order.end(), 0);
  {
    const double sx = in.maxx > in.minx ? 65535.0 / ((double)in.maxx - in.minx) : 0.0;
    const double sy = in.maxy > in.miny ? 65535.0 / ((double)in.maxy - in.miny) : 0.0;
    key.resize((size_t)m);
    for (int k = 0; k < m; ++k) {
      const int i = ids[k];
      const uint32_t qx = (uint32_t)(((double)xy[2 * i] - in.minx) * sx), qy = (uint32_t)(((double)xy[2 * i + 1] - in.miny) * sy);
      key[(size_t)k] = spread16(qx) | (spread16(qy) << 1);
    }
    // stable LSD radix sort by key (3 passes of 11 bits)
    tmp.resize((size_t)m);
    count.resize(2049);
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = 11 * pass;
      std::fill(count.begin(), count.end(), 0u);
      for (int k = 0; k < m; ++k) count[((key[(size_t)order[(size_t)k]] >> shift) & 2047u) + 1]++;
      for (int b = 0; b < 2048; ++b) count[(size_t)b + 1] += count[(size_t)b];
      for (int k = 0; k < m; ++k) tmp[count[(key[(size_t)order[(size_t)k]] >> shift) & 2047u]++] = order[(size_t)k];
      order.swap(tmp);
    }
  }

  // ---- first non-degenerate triangle --------------------------------------------------------------------
  const int a = order[0];
  size_t ib = 1;
  while (ib < order.size() && T.p[(size_t)order[ib]].x == T.p[(size_t)a].x && T.p[(size_t)order[ib]].y == T.p[(size_t)a].y) ++ib;
  if (ib >= order.size()) return false;  // all points identical
  const int b = order[ib];
  size_t ic = ib + 1;
  while (ic < order.size() && T.orient(a, b, order[ic]) == 0) ++ic;
  if (ic >= order.size()) return false;  // all points collinear: no triangles
  int c = order[ic];
  int v0 = a, v1 = b, v2 = c;
  if (T.orient(v0, v1, v2) < 0) std::swap(v1, v2);
  T.t.clear();
  T.free_.clear();
  T.t.reserve((size_t)m * 4 + 16);
  const int t0 = T.new_tri(v0, v1, v2);
  // ghosts: across edge opposite v[i] of t0, i.e. edge (v[i+1], v[i+2]); the ghost holds it reversed
  int g[3];
  for (int i = 0; i < 3; ++i) g[i] = T.new_tri(T.t[t0].v[(i + 2) % 3], T.t[t0].v[(i + 1) % 3], GHOST);
  for (int i = 0; i < 3; ++i) {
    T.t[t0].n[i] = g[i];
    T.t[g[i]].n[2] = t0;  // across its real edge (opposite GHOST)
  }
  // ghost i = (u, w, G) with u = v[i+2], w = v[i+1]: edge w-G (opposite u, index 0) is shared with the ghost
  // that starts at w, i.e. ghost (i+2) = (v[i+1], v[i], G); edge G-u (opposite w, index 1) with ghost (i+1).
  for (int i = 0; i < 3; ++i) {
    T.t[g[i]].n[0] = g[(i + 2) % 3];
    T.t[g[i]].n[1] = g[(i + 1) % 3];
  }

  used.assign((size_t)m, 0);
  used[(size_t)v0] = used[(size_t)v1] = used[(size_t)v2] = 1;
  int last = t0;
  for (size_t k = 0; k < order.size(); ++k) {
    const int d = order[k];
    if (used[(size_t)d]) continue;
    used[(size_t)d] = 1;
    int seed = T.locate(last, d);
    // duplicate of an existing vertex: the located triangle has it as a corner
    bool dup = false;
    for (int i = 0; i < 3; ++i) {
      const int v = T.t[seed].v[i];
      if (v != GHOST && T.p[(size_t)v].x == T.p[(size_t)d].x && T.p[(size_t)v].y == T.p[(size_t)d].y) dup = true;
    }
    if (dup) continue;
    if (!T.in_disk(seed, d)) {
      // d lies on the boundary of the located triangle's disk only when it is ON an edge / hull line:
      // one of the neighbours then contains it in its open disk
      int alt = -1;
      for (int i = 0; i < 3 && alt < 0; ++i) {
        const int nb = T.t[seed].n[i];
        if (nb >= 0 && T.in_disk(nb, d)) alt = nb;
      }
      if (alt < 0) continue;  // cannot happen for a point not equal to a vertex; skip defensively
      seed = alt;
    }
    last = T.insert(seed, d);
  }
  return true;
}

struct Rect {
  double x0, x1, y0, y1;  // empty when x0 >= x1 or y0 >= y1
  bool empty() const { return !(x0 < x1 && y0 < y1); }
};

// ---- one strip of the parallel build ------------------------------------------------------------------------------------
struct StripResult {
  std::vector<int32_t> tris, edges;  // global vertex ids
  int64_t hull_edges = 0, vertices = 0;  // hull edges among `edges`; distinct points of the strip's own x range in its triangulation
  bool certified = true;
};

// Is the closed disk through the (counter-clockwise) triangle a, b, c clear of rectangle r?  In double: the coordinates are
// integer multiples of a common power of two with <= 25 significant bits here (filter_ok), so the orientation determinant
// is exact and centre / radius are good to ~1e-15 relative; the comparison keeps a margin far above that.
inline bool disk_clear_of(const Pt& a, const Pt& b, const Pt& c, const Rect& r) {
  if (r.empty()) return true;
  const double bx = b.fx - a.fx, by = b.fy - a.fy, cx = c.fx - a.fx, cy = c.fy - a.fy;
  const double d = 2.0 * (bx * cy - by * cx);
  if (!(d > 0.0)) return false;
  const double bl = bx * bx + by * by, cl = cx * cx + cy * cy;
  const double ux = (cy * bl - by * cl) / d, uy = (bx * cl - cx * bl) / d;  // centre relative to a
  const double r2 = ux * ux + uy * uy;
  const double ox = a.fx + ux, oy = a.fy + uy;
  const double dx = std::max(std::max(r.x0 - ox, 0.0), ox - r.x1), dy = std::max(std::max(r.y0 - oy, 0.0), oy - r.y1);
  const double dist2 = dx * dx + dy * dy;
  return dist2 > r2 * (1.0 + 1e-9) + 1e-9 * (1.0 + std::fabs(ox) + std::fabs(oy));
}

// Is the open half-plane to the RIGHT of u -> v (the outer side of a hull edge whose triangle lies on the left) clear of r?
inline bool outer_side_clear_of(const Pt& u, const Pt& v, const Rect& r) {
  if (r.empty()) return true;
  const double ex = v.fx - u.fx, ey = v.fy - u.fy;
  const double tol = 1e-9 * (std::fabs(ex) + std::fabs(ey)) * (1.0 + std::fabs(r.x1 - r.x0) + std::fabs(r.y1 - r.y0));
  const double xs[2] = {r.x0, r.x1}, ys[2] = {r.y0, r.y1};
  for (double x : xs)
    for (double y : ys)
      if (ex * (y - u.fy) - ey * (x - u.fx) < tol) return false;  // a corner on the line or right of it
  return true;
}

constexpr int kBins = 1024;

// Strip s owns the points of x bins [core0, core1); it triangulates the bins [core0 - halo, core1 + halo) plus the band points
// (y within `band` of the bounding box's top / bottom) of all other bins.
void run_strip(const Input& in, const std::vector<int>& by_bin, const std::vector<int>& bin_start, const std::vector<int>& band_pts,
               const std::vector<uint16_t>& bin_of, int core0, int core1, int halo, double band, StripResult* out) {
  const int lo = std::max(0, core0 - halo), hi = std::min(kBins, core1 + halo);
  static thread_local std::vector<int> ids;
  ids.assign(by_bin.begin() + bin_start[(size_t)lo], by_bin.begin() + bin_start[(size_t)hi]);
  const size_t n_block = ids.size();
  for (int i : band_pts)
    if (bin_of[(size_t)i] < lo || bin_of[(size_t)i] >= hi) ids.push_back(i);
  // ascending ids inside a bin, bins ascending; the band points of other bins come last (their Morton keys differ from every
  // block point's: another x bin) -- the relative order of equal keys is that of the whole input
  (void)n_block;
  out->tris.clear(), out->edges.clear(), out->hull_edges = 0, out->vertices = 0, out->certified = true;
  static thread_local Triangulator T;
  if (!triangulate_subset(in, ids.data(), (int)ids.size(), T)) {
    out->certified = false;
    return;
  }
  // Bin b holds the points whose quantised x, floor((x - minx) * 65535 / W), lies in [64 b, 64 b + 63] (triangulate_strips): the
  // points this strip has NOT looked at have x < minx + 64 lo / sx on the left and x >= minx + 64 hi / sx on the right.  Both box
  // edges come from that same quantisation, widened by one quantum (the product is rounded before the floor) -- W / 1024 per bin,
  // as used before, is smaller than a bin by 1 / 65536 of its width per bin index and left points of bin lo - 1 outside the box.
  const double inv_sx = ((double)in.maxx - (double)in.minx) / 65535.0;
  const double xlo = lo == 0 ? -1e300 : (double)in.minx + (64.0 * lo + 1.0) * inv_sx;
  const double xhi = hi == kBins ? 1e300 : (double)in.minx + (64.0 * hi - 1.0) * inv_sx;
  // what this strip has NOT looked at: the boxes left and right of its x range, between the bands (the boxes are closed and one
  // quantum wider than the excluded bins -- conservative)
  const Rect left = {(double)in.minx, xlo, (double)in.miny + band, (double)in.maxy - band};
  const Rect right = {xhi, (double)in.maxx, (double)in.miny + band, (double)in.maxy - band};
  static thread_local std::vector<char> in_tri;
  in_tri.assign(ids.size(), 0);
  auto own = [&](int v) { return bin_of[(size_t)ids[(size_t)v]] >= core0 && bin_of[(size_t)ids[(size_t)v]] < core1; };
  for (size_t ti = 0; ti < T.t.size(); ++ti) {
    const Tri& tr = T.t[ti];
    if (!tr.alive || Triangulator::is_ghost(tr)) continue;
    for (int i = 0; i < 3; ++i) in_tri[(size_t)tr.v[i]] = 1;
    // the leftmost vertex (smallest x, then smallest global id) decides whose triangle this is
    int lm = 0;
    for (int i = 1; i < 3; ++i) {
      const Pt &q = T.p[(size_t)tr.v[i]], &m = T.p[(size_t)tr.v[lm]];
      if (q.x < m.x || (q.x == m.x && ids[(size_t)tr.v[i]] < ids[(size_t)tr.v[lm]])) lm = i;
    }
    if (!own(tr.v[lm])) continue;
    const Pt &a = T.p[(size_t)tr.v[0]], &b = T.p[(size_t)tr.v[1]], &c = T.p[(size_t)tr.v[2]];
    if (!disk_clear_of(a, b, c, left) || !disk_clear_of(a, b, c, right)) {
      if (std::getenv("FLAME_DELAUNAY_TRACE"))
        std::fprintf(stderr, "[delaunay] strip bins [%d,%d) halo %d: disk of (%g,%g) (%g,%g) (%g,%g) reaches an unseen box [%g,%g] / [%g,%g] x [%g,%g]\n", core0, core1, halo,
                     a.fx, a.fy, b.fx, b.fy, c.fx, c.fy, left.x0, left.x1, right.x0, right.x1, left.y0, left.y1);
      out->certified = false;
      return;
    }
    out->tris.push_back(ids[(size_t)tr.v[0]]), out->tris.push_back(ids[(size_t)tr.v[1]]), out->tris.push_back(ids[(size_t)tr.v[2]]);
    for (int i = 0; i < 3; ++i) {
      const int u = tr.v[i], v = tr.v[(i + 1) % 3];  // the triangle is on the left of u -> v
      const int nb = tr.n[(i + 2) % 3];
      const bool hull = nb < 0 || Triangulator::is_ghost(T.t[(size_t)nb]);
      if (hull) {
        if (!outer_side_clear_of(T.p[(size_t)u], T.p[(size_t)v], left) || !outer_side_clear_of(T.p[(size_t)u], T.p[(size_t)v], right)) {
          if (std::getenv("FLAME_DELAUNAY_TRACE"))
            std::fprintf(stderr, "[delaunay] strip bins [%d,%d) halo %d: hull edge (%g,%g)->(%g,%g): its outer side reaches an unseen box\n", core0, core1, halo,
                         T.p[(size_t)u].fx, T.p[(size_t)u].fy, T.p[(size_t)v].fx, T.p[(size_t)v].fy);
          out->certified = false;
          return;
        }
        out->hull_edges++;
      }
      // an inner edge is listed by the triangle that has it as lower id -> higher id, a hull edge by its only triangle
      if (hull || ids[(size_t)u] < ids[(size_t)v]) out->edges.push_back(ids[(size_t)u]), out->edges.push_back(ids[(size_t)v]);
    }
  }
  for (size_t k = 0; k < ids.size(); ++k)
    if (in_tri[k] && own((int)k)) out->vertices++;
}

// The parallel build; false = not certified (the caller triangulates sequentially).
// The strips' triangles and edges go straight into the caller's arrays (every strip copies its own share, in strip order); with
// null arrays only the counts are returned.  *fits = false: an array was too small (counts are still reported).
bool triangulate_strips(const Input& in, int n_strips, int32_t* tris, int32_t tri_capacity, int32_t* n_tris, int32_t* edges,
                        int32_t edge_capacity, int32_t* n_edges, bool* fits) {
  const int32_t n = in.n;
  const bool prof = std::getenv("FLAME_DELAUNAY_PROFILE") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  if (!in.filter_ok || !(in.maxx > in.minx) || !(in.maxy > in.miny)) return false;
  // x bins (the Morton key's x quantisation >> 6: equal keys share a bin), counting sort by bin
  std::vector<uint16_t> bin_of((size_t)n);
  std::vector<int> bin_start(kBins + 1, 0), by_bin((size_t)n), band_pts;
  const double sx = 65535.0 / ((double)in.maxx - in.minx);
  for (int32_t i = 0; i < n; ++i) {
    bin_of[(size_t)i] = (uint16_t)((uint32_t)(((double)in.xy[2 * i] - in.minx) * sx) >> 6);
    bin_start[(size_t)bin_of[(size_t)i] + 1]++;
  }
  for (int b = 0; b < kBins; ++b) bin_start[(size_t)b + 1] += bin_start[(size_t)b];
  {
    std::vector<int> at(bin_start.begin(), bin_start.end() - 1);
    for (int32_t i = 0; i < n; ++i) by_bin[(size_t)at[bin_of[(size_t)i]]++] = i;
  }
  // mean spacing of the points -> halo (in bins) and band height
  const double W = (double)in.maxx - in.minx, H = (double)in.maxy - in.miny;
  const double spacing = std::sqrt(W * H / std::max(1, n));
  const double band = std::min(0.25 * H, 2.0 * spacing);
  const int halo0 = std::max(1, (int)std::ceil(4.0 * spacing / (W / kBins)));
  for (int32_t i = 0; i < n; ++i) {
    const double y = in.xy[2 * i + 1];
    if (y <= (double)in.miny + band || y >= (double)in.maxy - band) band_pts.push_back(i);
  }
  // strips of equal point counts, cut at bin edges
  std::vector<int> cut((size_t)n_strips + 1, 0);
  cut[(size_t)n_strips] = kBins;
  for (int s = 1; s < n_strips; ++s) {
    const int want = (int)((int64_t)n * s / n_strips);
    cut[(size_t)s] = (int)(std::upper_bound(bin_start.begin(), bin_start.end(), want) - bin_start.begin()) - 1;
    cut[(size_t)s] = std::max(cut[(size_t)s], cut[(size_t)s - 1] + 1);
    if (cut[(size_t)s] >= kBins) return false;
  }
  std::vector<StripResult> res((size_t)n_strips);
  const double t1 = now();
  flame_hip::Workers::get().run(n_strips, [&](int s) {
    run_strip(in, by_bin, bin_start, band_pts, bin_of, cut[(size_t)s], cut[(size_t)s + 1], halo0, band, &res[(size_t)s]);
    if (!res[(size_t)s].certified)  // once more, looking three times as far
      run_strip(in, by_bin, bin_start, band_pts, bin_of, cut[(size_t)s], cut[(size_t)s + 1], 3 * halo0, band, &res[(size_t)s]);
  });
  const double t2 = now();
  int64_t nt = 0, ne = 0, nh = 0, nv = 0;
  for (const StripResult& r : res) {
    if (!r.certified) return false;
    nt += (int64_t)r.tris.size() / 3, ne += (int64_t)r.edges.size() / 2, nh += r.hull_edges, nv += r.vertices;
  }
  // Euler: a triangulation of nv points with nh hull edges has 2 nv - 2 - nh triangles and 3 nv - 3 - nh edges
  if (nv < 3 || nt != 2 * nv - 2 - nh || ne != 3 * nv - 3 - nh) {
    if (std::getenv("FLAME_DELAUNAY_TRACE"))
      std::fprintf(stderr, "[delaunay] strips do not add up: %lld vertices, %lld hull edges, %lld triangles, %lld edges\n", (long long)nv, (long long)nh,
                   (long long)nt, (long long)ne);
    return false;
  }
  *n_tris = (int32_t)nt, *n_edges = (int32_t)ne;
  *fits = (!tris || tri_capacity >= nt) && (!edges || edge_capacity >= ne);
  if (*fits && (tris || edges)) {
    std::vector<int64_t> t_at((size_t)n_strips), e_at((size_t)n_strips);
    int64_t t = 0, e = 0;
    for (int s = 0; s < n_strips; ++s) t_at[(size_t)s] = t, e_at[(size_t)s] = e, t += (int64_t)res[(size_t)s].tris.size(), e += (int64_t)res[(size_t)s].edges.size();
    flame_hip::Workers::get().run(n_strips, [&](int s) {
      const StripResult& r = res[(size_t)s];
      if (tris && !r.tris.empty()) std::memcpy(tris + t_at[(size_t)s], r.tris.data(), sizeof(int32_t) * r.tris.size());
      if (edges && !r.edges.empty()) std::memcpy(edges + e_at[(size_t)s], r.edges.data(), sizeof(int32_t) * r.edges.size());
    });
  }
  if (prof) std::fprintf(stderr, "[delaunay] %d points, %d strips: bins %.3f ms, strips %.3f ms, copy-out %.3f ms\n", n, n_strips, t1 - t0, t2 - t1, now() - t2);
  return true;
}


// ---- divide and conquer (round 5) -----------------------------------------------------------------------------------------
// The certified strips above make every strip triangulate ~5 x its own share (halo + bands).  Here a strip triangulates its OWN
// points only, and neighbouring parts are merged along their seam, pair by pair, in a tree:
//   * the lower common tangent of the two convex hulls (the parts are separated by a vertical line: their x ranges are disjoint bins);
//   * a zipper up the two hull chains that face each other: every hull edge of the chains has a ghost triangle (edge + the vertex at
//     infinity) -- it BECOMES the real triangle (edge + the opposite end of the current base) by getting that vertex for its ghost
//     vertex, so nothing is allocated but the two ghosts of the tangent edges;
//   * Lawson flips, started from the zipper's triangles, with the exact in-circle predicate: the result is the Delaunay triangulation
//     of the union (unique in general position).
// Everything lives in one arena of triangles with GLOBAL vertex ids, a slot range per strip; merges of one level touch disjoint parts.
// The team of threads goes through all phases behind spin barriers (host_workers.hpp).  Any surprise -- a strip without a triangle,
// a degenerate tangent, a guard that trips -- gives up and the caller takes the certified strips.
struct Merger {
  Triangulator G;  // p: all points by global id; t: the arena
  bool ok = true;

  int ghost_from(int g) const { return G.t[(size_t)g].v[0]; }
  int ghost_to(int g) const { return G.t[(size_t)g].v[1]; }
  // hull vertex handles are their OUTGOING ghosts (v, cw-next, GHOST): n[0] = the ghost that starts at cw-next, n[1] = the one that ends at v
  int cw(int go) const { return G.t[(size_t)go].n[0]; }

  void flip(int A, int i, int B, int j, std::vector<int>& stack) {
    Tri& ta = G.t[(size_t)A];
    Tri& tb = G.t[(size_t)B];
    const int a = ta.v[i], b = ta.v[(i + 1) % 3], c = ta.v[(i + 2) % 3], d = tb.v[j];
    const int n_ab = ta.n[(i + 2) % 3], n_ca = ta.n[(i + 1) % 3];  // across a-b (opposite c), across c-a (opposite b)
    // in B = (d, c, b) rotated: across b-d is opposite c, across d-c is opposite b
    int jc = -1, jb = -1;
    for (int k = 0; k < 3; ++k) {
      if (tb.v[k] == c) jc = k;
      if (tb.v[k] == b) jb = k;
    }
    const int n_bd = tb.n[jc], n_dc = tb.n[jb];
    // A' = (a, b, d), B' = (a, d, c)
    ta.v[0] = a, ta.v[1] = b, ta.v[2] = d;
    ta.n[0] = n_bd, ta.n[1] = B, ta.n[2] = n_ab;
    tb.v[0] = a, tb.v[1] = d, tb.v[2] = c;
    tb.n[0] = n_dc, tb.n[1] = n_ca, tb.n[2] = A;
    auto relink = [&](int outer, int was, int now) {
      if (outer < 0) return;
      Tri& o = G.t[(size_t)outer];
      for (int k = 0; k < 3; ++k)
        if (o.n[k] == was) o.n[k] = now;
    };
    relink(n_bd, B, A);   // (n_ab stays with A, n_dc with B)
    relink(n_ca, A, B);
    // the four outer edges of the pair are to be looked at again (entries are triangle * 4 + edge; the new diagonal is Delaunay)
    stack.push_back(4 * A + 0), stack.push_back(4 * A + 2), stack.push_back(4 * B + 0), stack.push_back(4 * B + 1);
  }

  // Merges the parts whose hulls hold the ghosts gl (left part) and gr (right part); ghost slots s0, s1 are free for the two tangent
  // edges.  Returns a ghost of the merged hull, or -1.
  int merge(int gl, int gr, int s0, int s1, std::vector<int>& stack) {
    const size_t guard_hull = G.t.size() + 16;
    // rightmost vertex of the left hull, leftmost of the right hull (by their outgoing ghosts)
    int a = gl, b = gr;
    {
      size_t steps = 0;
      for (int g = cw(gl); g != gl; g = cw(g)) {
        if (G.p[(size_t)ghost_from(g)].x > G.p[(size_t)ghost_from(a)].x) a = g;
        if (++steps > guard_hull) return -1;
      }
      steps = 0;
      for (int g = cw(gr); g != gr; g = cw(g)) {
        if (G.p[(size_t)ghost_from(g)].x < G.p[(size_t)ghost_from(b)].x) b = g;
        if (++steps > guard_hull) return -1;
      }
    }
    // lower common tangent: a walks down the left hull (clockwise), b down the right hull (counter-clockwise)
    for (size_t steps = 0;; ++steps) {
      if (steps > 2 * guard_hull) return -1;
      const int va = ghost_from(a), vb = ghost_from(b);
      const int pa = cw(a), vp = ghost_from(pa);
      int o = G.orient(va, vb, vp);
      if (o < 0 || (o == 0 && G.p[(size_t)vp].x > G.p[(size_t)va].x)) {
        a = pa;
        continue;
      }
      const int qb = G.t[(size_t)b].n[1];  // the ghost that ends at vb = the outgoing ghost of its ccw-next
      const int vq = ghost_from(qb);
      o = G.orient(va, vb, vq);
      if (o < 0 || (o == 0 && G.p[(size_t)vq].x < G.p[(size_t)vb].x)) {
        b = qb;
        continue;
      }
      break;
    }
    int bl = ghost_from(a), br = ghost_from(b);
    int gL = G.t[(size_t)a].n[1];  // (nl, bl, GHOST): the left hull's edge that goes UP from bl
    int gR = b;                    // (br, nr, GHOST): the right hull's edge that goes UP from br
    // the ghost of the lower tangent (hull edge bl -> br counter-clockwise: the ghost holds it reversed)
    const int GB = s0, GT = s1;
    {
      Tri& t = G.t[(size_t)GB];
      t.v[0] = br, t.v[1] = bl, t.v[2] = GHOST;
      t.alive = 1, t.ghost = 1, t.in_cavity = 0;
      t.n[0] = a;                       // starts at bl
      t.n[1] = G.t[(size_t)b].n[1];     // ends at br
      t.n[2] = -1;
      G.t[(size_t)a].n[1] = GB;
      G.t[(size_t)t.n[1]].n[0] = GB;
    }
    const size_t first_new = stack.size();
    int prev = GB, prev_slot = 2, made = 0;
    for (size_t steps = 0;; ++steps) {
      if (steps > 2 * guard_hull) return -1;
      const int nl = ghost_from(gL), nr = ghost_to(gR);
      if (ghost_to(gL) != bl || ghost_from(gR) != br) return -1;
      const bool lvis = G.orient(bl, nl, br) < 0, rvis = G.orient(br, nr, bl) > 0;
      if (!lvis && !rvis) break;
      bool take_left = lvis;
      if (lvis && rvis) {
        take_left = !(G.incircle(bl, br, nl, nr) > 0);
        if (take_left && !(G.orient(br, nr, nl) > 0)) take_left = false;       // (its new base must not cut the other chain)
        else if (!take_left && !(G.orient(bl, nl, nr) < 0)) take_left = true;
      }
      if (take_left) {
        if (!(G.orient(bl, br, nl) > 0)) return -1;
        Tri& t = G.t[(size_t)gL];  // (nl, bl, GHOST) -> (nl, bl, br)
        const int next_gL = t.n[1];
        t.v[2] = br, t.ghost = 0;
        t.n[0] = prev, G.t[(size_t)prev].n[prev_slot] = gL;  // across bl-br: the old base
        t.n[1] = -1;
        stack.push_back(4 * gL), stack.push_back(4 * gL + 1), stack.push_back(4 * gL + 2);
        prev = gL, prev_slot = 1;  // across br-nl: the new base
        bl = nl, gL = next_gL;
      } else {
        if (!(G.orient(bl, br, nr) > 0)) return -1;
        Tri& t = G.t[(size_t)gR];  // (br, nr, GHOST) -> (br, nr, bl)
        const int next_gR = t.n[0];
        t.v[2] = bl, t.ghost = 0;
        t.n[1] = prev, G.t[(size_t)prev].n[prev_slot] = gR;  // across bl-br
        t.n[0] = -1;
        stack.push_back(4 * gR), stack.push_back(4 * gR + 1), stack.push_back(4 * gR + 2);
        prev = gR, prev_slot = 0;  // across nr-bl
        br = nr, gR = next_gR;
      }
      ++made;
    }
    if (made == 0) return -1;
    {  // the ghost of the upper tangent (hull edge br -> bl counter-clockwise)
      Tri& t = G.t[(size_t)GT];
      t.v[0] = bl, t.v[1] = br, t.v[2] = GHOST;
      t.alive = 1, t.ghost = 1, t.in_cavity = 0;
      t.n[0] = gR, t.n[1] = gL, t.n[2] = prev;
      G.t[(size_t)prev].n[prev_slot] = GT;
      G.t[(size_t)gR].n[1] = GT;
      G.t[(size_t)gL].n[0] = GT;
    }
    // Lawson flips from the zipper's triangles
    size_t flips = 0;
    const size_t guard_flips = 64 * G.t.size() + 1024;
    while (stack.size() > first_new) {
      const int A = stack.back() >> 2, i = stack.back() & 3;
      stack.pop_back();
      const Tri& ta = G.t[(size_t)A];
      if (!ta.alive || ta.ghost) continue;
      const int B = ta.n[i];
      if (B < 0) return -1;
      const Tri& tb = G.t[(size_t)B];
      if (tb.ghost) continue;
      int j = -1;
      for (int k = 0; k < 3; ++k)
        if (tb.n[k] == A) j = k;
      if (j < 0) return -1;
      if (G.incircle(ta.v[0], ta.v[1], ta.v[2], tb.v[j]) > 0) {
        if (++flips > guard_flips) return -1;
        flip(A, i, B, j, stack);
      }
    }
    if (std::getenv("FLAME_DELAUNAY_TRACE")) std::fprintf(stderr, "[delaunay] merge: %d zipper triangles, %zu flips\n", made, flips);
    return GB;
  }
};

bool triangulate_merge(const Input& in, int n_strips, int32_t* tris, int32_t tri_capacity, int32_t* n_tris, int32_t* edges,
                       int32_t edge_capacity, int32_t* n_edges, bool* fits) {
  const int32_t n = in.n;
  const bool prof = std::getenv("FLAME_DELAUNAY_PROFILE") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  if (!in.filter_ok || !(in.maxx > in.minx) || !(in.maxy > in.miny) || n_strips < 2) return false;
  flame_hip::Workers& W = flame_hip::Workers::get();
  const int team = std::max(1, std::min(W.threads(), n_strips));  // (one thread does the same work in the same order: the output does not
                                                                  //  depend on the machine)
  // x bins as in triangulate_strips (equal keys share a bin), counting sort by bin
  // (scratch of the CALLING thread, kept between calls; the team reaches it through these references -- a thread_local name inside
  //  the team's lambda would be the worker's own, empty, copy)
  static thread_local std::vector<uint16_t> tl_bin_of;
  static thread_local std::vector<int> tl_bin_start, tl_by_bin;
  std::vector<uint16_t>& bin_of = tl_bin_of;
  std::vector<int>&bin_start = tl_bin_start, &by_bin = tl_by_bin;
  bin_of.resize((size_t)n), by_bin.resize((size_t)n);
  bin_start.assign(kBins + 1, 0);
  const double sx = 65535.0 / ((double)in.maxx - in.minx);
  for (int32_t i = 0; i < n; ++i) {
    bin_of[(size_t)i] = (uint16_t)((uint32_t)(((double)in.xy[2 * i] - in.minx) * sx) >> 6);
    bin_start[(size_t)bin_of[(size_t)i] + 1]++;
  }
  for (int b = 0; b < kBins; ++b) bin_start[(size_t)b + 1] += bin_start[(size_t)b];
  {
    std::vector<int> at(bin_start.begin(), bin_start.end() - 1);
    for (int32_t i = 0; i < n; ++i) by_bin[(size_t)at[bin_of[(size_t)i]]++] = i;
  }
  std::vector<int> cut((size_t)n_strips + 1, 0);
  cut[(size_t)n_strips] = kBins;
  for (int s = 1; s < n_strips; ++s) {
    const int want = (int)((int64_t)n * s / n_strips);
    cut[(size_t)s] = (int)(std::upper_bound(bin_start.begin(), bin_start.end(), want) - bin_start.begin()) - 1;
    cut[(size_t)s] = std::max(cut[(size_t)s], cut[(size_t)s - 1] + 1);
    if (cut[(size_t)s] >= kBins) return false;
  }
  // the arena: a slot range per strip (a triangulation of m points never holds more than 2 m + 2 triangles, ghosts included), then two
  // ghosts per merge
  static thread_local Merger tl_merger;
  Merger& M = tl_merger;
  M.ok = true;
  Triangulator& G = M.G;
  G.filter_ok = in.filter_ok, G.ori_static = in.ori_static, G.icc_static = in.icc_static;
  G.p.resize((size_t)n);
  std::vector<int> base((size_t)n_strips + 1, 0), comp((size_t)n_strips, -1);
  for (int s = 0; s < n_strips; ++s) {
    const int m = bin_start[(size_t)cut[(size_t)s + 1]] - bin_start[(size_t)cut[(size_t)s]];
    if (m < 3) return false;
    base[(size_t)s + 1] = base[(size_t)s] + 2 * m + 8;
  }
  const int ghost_base = base[(size_t)n_strips];
  G.t.resize((size_t)ghost_base + 2 * (size_t)n_strips);
  for (size_t i = (size_t)ghost_base; i < G.t.size(); ++i) G.t[i].alive = 0, G.t[i].ghost = 0;
  // output ranges: the strips' slot ranges and the tangent ghosts' (a tangent edge of one merge may face the seam of a later one and
  // become a real triangle there)
  const int n_ranges = n_strips + 1;
  base.push_back((int)G.t.size());
  std::atomic<int> failed{0};
  // The phases are separated by WORK counters, not by thread barriers: a phase is over when its items are done, whoever did them.  A
  // thread that wakes late (a pool asleep for a frame: the slowest of 31 wake-ups used to decide when phase 0 ended) finds the items
  // taken, the counters full, and walks through.
  struct Phase {
    std::atomic<int> next{0}, done{0};
    int total = 0;
    int take() { return next.fetch_add(1, std::memory_order_relaxed); }
    void finished() { done.fetch_add(1, std::memory_order_release); }
    void wait() const {
      for (int spins = 0; done.load(std::memory_order_acquire) < total; ++spins)
        if (spins > 4000) std::this_thread::yield();
    }
  };
  // The merge tree: parts [i, i + w) and [i + w, i + 2 w), w = 1, 2, 4 ...  A merge needs its two parts and nothing else, so it runs as
  // soon as the second of them stands -- done by the thread that has just finished that part (its triangles are in that core's
  // cache) -- and not when a whole level of the tree has been merged (round 5: five level barriers at ~45 us each were half of the
  // triangulation's time, the seams themselves 12 us apiece).  The thread that arrives first at a merge leaves it to the other.
  struct Pair { int left, right, slot; };
  std::vector<Pair> pairs;
  std::vector<std::vector<int>> as_left((size_t)n_strips);  // the merges in which part r is the left side, in tree order
  std::vector<int> as_right((size_t)n_strips, -1);           // ... and the one that absorbs it
  for (int w = 1; w < n_strips; w *= 2)
    for (int i = 0; i + w < n_strips; i += 2 * w) {
      as_left[(size_t)i].push_back((int)pairs.size()), as_right[(size_t)(i + w)] = (int)pairs.size();
      pairs.push_back(Pair{i, i + w, ghost_base + 2 * (int)pairs.size()});
    }
  std::vector<std::atomic<int>> arrived(pairs.size());
  for (auto& a : arrived) a.store(0, std::memory_order_relaxed);
  std::atomic<int> merged{pairs.empty() ? 1 : 0};
  // (FLAME_DELAUNAY_PROFILE: when every strip and every merge began and ended, and on which member of the team)
  std::vector<double> pf_strip(prof ? 3 * (size_t)n_strips : 0, 0.0), pf_merge(prof ? 3 * pairs.size() : 0, 0.0);
  Phase strips_phase, count_phase, write_phase;
  strips_phase.total = n_strips, count_phase.total = n_ranges, write_phase.total = n_ranges;
  // per output range (the strips' slot ranges + the tangent ghosts'): counts, then offsets
  std::vector<int64_t> cnt_t((size_t)n_ranges, 0), cnt_e((size_t)n_ranges, 0);
  double t_strips = 0, t_merged = 0;
  const double t_call = now();
  W.run(team, [&](int member) {
    static thread_local Triangulator T;
    static thread_local std::vector<int> ids;
    static thread_local std::vector<int> stack;
    // part `r` (a strip, or what has been merged into it so far) stands: go up the tree as long as this thread is the second to arrive
    auto climb = [&](int r, int member) {
      size_t stage = 0;  // merges of part r as the left side that are done
      for (;;) {
        const int pi = stage < as_left[(size_t)r].size() ? as_left[(size_t)r][stage] : as_right[(size_t)r];
        if (pi < 0) {  // part 0 after its last merge: the triangulation stands
          merged.store(1, std::memory_order_release);
          return;
        }
        if (arrived[(size_t)pi].fetch_add(1, std::memory_order_acq_rel) == 0) return;  // the other side is not there yet: its thread merges
        const Pair& pr = pairs[(size_t)pi];
        if (!failed.load()) {
          stack.clear();
          if (prof) pf_merge[3 * (size_t)pi] = now(), pf_merge[3 * (size_t)pi + 2] = member;
          const int g = M.merge(comp[(size_t)pr.left], comp[(size_t)pr.right], pr.slot, pr.slot + 1, stack);
          if (prof) pf_merge[3 * (size_t)pi + 1] = now();
          if (g < 0) failed.store(1);
          comp[(size_t)pr.left] = g;
        }
        // the merged part is rooted at pr.left and has done one more merge as the left side than pr.left had before
        if (r != pr.left) {
          r = pr.left;
          stage = 0;
          while (stage < as_left[(size_t)r].size() && as_left[(size_t)r][stage] != pi) ++stage;
        }
        ++stage;
      }
    };
    // ---- phase 0: the strips, each into its slot range ----------------------------------------------------------------------
    for (int s; (s = strips_phase.take()) < n_strips; strips_phase.finished()) {
      if (prof) pf_strip[3 * (size_t)s] = now(), pf_strip[3 * (size_t)s + 2] = member;
      ids.assign(by_bin.begin() + bin_start[(size_t)cut[(size_t)s]], by_bin.begin() + bin_start[(size_t)cut[(size_t)s + 1]]);
      if (!triangulate_subset(in, ids.data(), (int)ids.size(), T) || (int)T.t.size() > base[(size_t)s + 1] - base[(size_t)s]) {
        failed.store(1);
        climb(s, member);  // (nothing is merged any more, but every merge still hears from both sides: nobody waits for ever)
        continue;
      }
      const int b0 = base[(size_t)s];
      for (size_t k = 0; k < ids.size(); ++k) G.p[(size_t)ids[k]] = T.p[k];
      int ghost = -1;
      for (size_t i = 0; i < T.t.size(); ++i) {
        Tri t = T.t[i];
        for (int k = 0; k < 3; ++k) {
          if (t.v[k] != GHOST) t.v[k] = ids[(size_t)t.v[k]];
          if (t.n[k] >= 0) t.n[k] += b0;
        }
        if (t.alive && t.ghost) {  // (rotated so that the ghost vertex is the last: what the merge reads)
          while (t.v[2] != GHOST) {
            const int v0 = t.v[0], n0 = t.n[0];
            t.v[0] = t.v[1], t.v[1] = t.v[2], t.v[2] = v0;
            t.n[0] = t.n[1], t.n[1] = t.n[2], t.n[2] = n0;
          }
          ghost = b0 + (int)i;
        }
        G.t[(size_t)b0 + i] = t;
      }
      for (int i = b0 + (int)T.t.size(); i < base[(size_t)s + 1]; ++i) G.t[(size_t)i].alive = 0, G.t[(size_t)i].ghost = 0;
      comp[(size_t)s] = ghost;
      if (ghost < 0) failed.store(1);
      if (prof) pf_strip[3 * (size_t)s + 1] = now();
      climb(s, member);
    }
    if (prof) {
      strips_phase.wait();
      if (t_strips == 0) t_strips = now();
    }
    // ---- the merge tree: see above; entered from phase 0 by whoever finishes a strip -----------------------------------------------
    for (int spins = 0; merged.load(std::memory_order_acquire) == 0; ++spins)
      if (spins > 4000) std::this_thread::yield();
    if (prof && t_merged == 0) t_merged = now();
    if (failed.load()) return;
    // ---- output: real triangles in slot order; an edge by the triangle with the smaller slot (or its only one) -------------------
    for (int s; (s = count_phase.take()) < n_ranges; count_phase.finished()) {
      int64_t ct = 0, ce = 0;
      for (int ti = base[(size_t)s]; ti < base[(size_t)s + 1]; ++ti) {
        const Tri& tr = G.t[(size_t)ti];
        if (!tr.alive || tr.ghost) continue;
        ++ct;
        for (int i = 0; i < 3; ++i) {
          const int nb = tr.n[(i + 2) % 3];
          if (nb < 0 || G.t[(size_t)nb].ghost || nb > ti) ++ce;
        }
      }
      cnt_t[(size_t)s] = ct, cnt_e[(size_t)s] = ce;
    }
    count_phase.wait();
    int64_t sum_t = 0, sum_e = 0;  // (every thread forms the same offsets for itself: 33 additions)
    for (int s = 0; s < n_ranges; ++s) sum_t += cnt_t[(size_t)s], sum_e += cnt_e[(size_t)s];
    if (!((!tris || tri_capacity >= sum_t) && (!edges || edge_capacity >= sum_e)) || (!tris && !edges)) return;
    for (int s; (s = write_phase.take()) < n_ranges; write_phase.finished()) {
      int64_t kt = 0, ke = 0;
      for (int r = 0; r < s; ++r) kt += cnt_t[(size_t)r], ke += cnt_e[(size_t)r];
      for (int ti = base[(size_t)s]; ti < base[(size_t)s + 1]; ++ti) {
        const Tri& tr = G.t[(size_t)ti];
        if (!tr.alive || tr.ghost) continue;
        if (tris) tris[3 * kt] = tr.v[0], tris[3 * kt + 1] = tr.v[1], tris[3 * kt + 2] = tr.v[2];
        ++kt;
        if (!edges) continue;
        for (int i = 0; i < 3; ++i) {
          const int nb = tr.n[(i + 2) % 3];
          if (nb < 0 || G.t[(size_t)nb].ghost || nb > ti) edges[2 * ke] = tr.v[i], edges[2 * ke + 1] = tr.v[(i + 1) % 3], ++ke;
        }
      }
    }
  });
  int64_t total_t = 0, total_e = 0;
  for (int s = 0; s < n_ranges; ++s) total_t += cnt_t[(size_t)s], total_e += cnt_e[(size_t)s];
  if (!failed.load()) {
    *n_tris = (int32_t)total_t, *n_edges = (int32_t)total_e;
    *fits = (!tris || tri_capacity >= total_t) && (!edges || edge_capacity >= total_e);
  }
  if (failed.load()) {
    if (std::getenv("FLAME_DELAUNAY_TRACE")) std::fprintf(stderr, "[delaunay] merge of %d strips gave up\n", n_strips);
    return false;
  }
  // Euler, over what went in: every point of the input is a vertex unless it duplicates another (those are not counted: the check is
  // triangles against edges, 3 T = 2 E - H with the hull's H from the ghosts -- cheap enough to keep)
  {
    int64_t hull = 0;
    const int g0 = comp[0];
    int g = g0;
    do {
      ++hull, g = G.t[(size_t)g].n[0];
    } while (g != g0 && hull <= (int64_t)G.t.size());
    if (3 * total_t != 2 * total_e - hull) {
      if (std::getenv("FLAME_DELAUNAY_TRACE"))
        std::fprintf(stderr, "[delaunay] merged strips do not add up: %lld triangles, %lld edges, %lld hull edges\n", (long long)total_t, (long long)total_e, (long long)hull);
      return false;
    }
  }
  // FLAME_DELAUNAY_VERIFY=1 (advisor, round 5): the Euler count above accepts any triangulation of the hull; this checks that the merged
  // one is THE Delaunay triangulation -- the exact in-circle predicate across every interior edge (what tests/test_delaunay.py does from
  // outside on integer-pixel inputs).  ~0.5 ms at 8 480 points, one thread: a switch for soaks and for a pipeline that wants the proof.
  if (std::getenv("FLAME_DELAUNAY_VERIFY")) {
    int64_t bad = 0;
    for (size_t ti = 0; ti < G.t.size(); ++ti) {
      const Tri& tr = G.t[ti];
      if (!tr.alive || tr.ghost) continue;
      for (int k = 0; k < 3; ++k) {
        const int nb = tr.n[k];  // across the edge opposite v[k]
        if (nb < 0 || (size_t)nb < ti || G.t[(size_t)nb].ghost || !G.t[(size_t)nb].alive) continue;
        const Tri& o = G.t[(size_t)nb];
        for (int j = 0; j < 3; ++j)
          if (o.v[j] != tr.v[(k + 1) % 3] && o.v[j] != tr.v[(k + 2) % 3]) bad += G.incircle(tr.v[0], tr.v[1], tr.v[2], o.v[j]) > 0;
      }
    }
    if (bad) {
      std::fprintf(stderr, "[delaunay] FLAME_DELAUNAY_VERIFY: %lld interior edges of the merged triangulation are not locally Delaunay: falling back\n", (long long)bad);
      return false;
    }
  }
  // the arena and the bins are kept by the calling thread between calls; a call far smaller than the largest one before gives memory back
  auto trim = [](auto& v) {
    if (v.capacity() > 4 * v.size() + 4096) v.shrink_to_fit();
  };
  trim(G.t), trim(G.p), trim(bin_of), trim(by_bin);
  if (prof) {
    double first = 1e300, last_start = 0, last_end = 0, dur_max = 0, dur_sum = 0;
    for (int k = 0; k < n_strips; ++k) {
      first = std::min(first, pf_strip[3 * (size_t)k]), last_start = std::max(last_start, pf_strip[3 * (size_t)k]);
      last_end = std::max(last_end, pf_strip[3 * (size_t)k + 1]);
      dur_max = std::max(dur_max, pf_strip[3 * (size_t)k + 1] - pf_strip[3 * (size_t)k]), dur_sum += pf_strip[3 * (size_t)k + 1] - pf_strip[3 * (size_t)k];
    }
    std::fprintf(stderr, "[delaunay]   bins + set-up until the team is called %.3f ms; strips: first starts at %.3f, last starts at %.3f, last ends at %.3f (a strip: mean %.3f, longest %.3f ms);"
                 " merges (start, duration us, member):", t_call - t0, first - t0, last_start - t0, last_end - t0, dur_sum / n_strips, dur_max);
    for (size_t k = 0; k < pairs.size(); ++k)
      if (pairs[k].right - pairs[k].left >= n_strips / 8 || k + 1 == pairs.size())  // (the upper levels: the critical path)
        std::fprintf(stderr, " [%d+%d: %.3f, %.0f, %d]", pairs[k].left, pairs[k].right, pf_merge[3 * k] - t0, 1e3 * (pf_merge[3 * k + 1] - pf_merge[3 * k]), (int)pf_merge[3 * k + 2]);
    std::fprintf(stderr, "\n");
  }
  if (prof) std::fprintf(stderr, "[delaunay] %d points, %d strips merged: all strips triangulated after %.3f ms, the tree merged after %.3f ms (merges start as their "
                         "parts stand), output %.3f ms\n", n, n_strips, t_strips - t0, t_merged - t0, now() - t_merged);
  return true;
}

}  // namespace

extern "C" int flame_delaunay_triangulate(const float* xy, int32_t n, int32_t* triangles, int32_t tri_capacity,
                                          int32_t* n_triangles, int32_t* edges, int32_t edge_capacity,
                                          int32_t* n_edges) {
  flame_hip::RoctxRange roctx_range_("flame_delaunay_triangulate");
  if (n < 0 || (n > 0 && !xy) || !n_triangles || !n_edges) return FLAME_NLTGV2_ERR_INVALID_ARG;
  *n_triangles = 0;
  *n_edges = 0;
  if (n < 3) return FLAME_NLTGV2_OK;
  const double t_enter = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();

  // ---- exact integer image of the coordinates --------------------------------------------------------
  int emin = 1000, emax = -1000;
  {  // one branch-free pass over the bit patterns (it vectorises); subnormal inputs take the general loop below
    uint32_t bmin = 255u, bmax = 0u, any_sub = 0u;
    for (int32_t i = 0; i < 2 * n; ++i) {
      uint32_t bits;
      std::memcpy(&bits, xy + i, sizeof bits);
      const uint32_t a = bits & 0x7fffffffu, b = a >> 23;
      bmax = std::max(bmax, b);
      bmin = std::min(bmin, a == 0u ? 255u : b);
      any_sub |= (b == 0u && a != 0u) ? 1u : 0u;
    }
    if (bmax == 255u) return FLAME_NLTGV2_ERR_INVALID_ARG;  // NaN / Inf
    if (!any_sub) {
      if (bmin != 255u) emin = (int)bmin - 126 - 24, emax = (int)bmax - 126;  // v = m * 2^e, m in [0.5, 1): e = biased - 126; ulp = 2^(e-24)
    } else {
      for (int32_t i = 0; i < 2 * n; ++i) {
        const float v = xy[i];
        if (v == 0.0f) continue;
        int e;
        std::frexp(v, &e);  // |v| in [2^(e-1), 2^e); ulp(v) = 2^(e-24)
        emin = std::min(emin, e - 24);
        emax = std::max(emax, e);
      }
    }
  }
  if (emin == 1000) return FLAME_NLTGV2_OK;  // all points at the origin
  if (emax - emin > 58) return FLAME_NLTGV2_ERR_INVALID_ARG;  // dynamic range beyond the exact predicates
  Input in;
  in.xy = xy, in.n = n;
  in.filter_ok = (emax - emin) <= 50;  // differences exact in double, products well inside the error bound
  in.to_int = std::ldexp(1.0, -emin);
  in.minx = in.maxx = xy[0], in.miny = in.maxy = xy[1];
  for (int32_t i = 0; i < n; ++i) {
    in.minx = std::min(in.minx, xy[2 * i]), in.maxx = std::max(in.maxx, xy[2 * i]);
    in.miny = std::min(in.miny, xy[2 * i + 1]), in.maxy = std::max(in.maxy, xy[2 * i + 1]);
  }
  {
    const double D = std::max((double)in.maxx - (double)in.minx, (double)in.maxy - (double)in.miny) * 1.0000001;
    in.ori_static = 4.0e-16 * 2.0 * D * D;
    in.icc_static = 1.2e-15 * 12.0 * D * D * D * D;
  }

  // ---- large inputs: strips in parallel (same triangulation; the order of the output is the strips') ------------------
  // (measured on a 256-core host: 8 480 points 0.83 / 0.65 / 0.64 ms with 8 / 16 / 32 strips, 57 600 points 5.1 / 3.9 / 3.5 ms.  The count fixes
  // the ORDER of the output edges: flame_amd/synth.py pins the former min(32, n / 1024) for the synthetic graphs the fixtures hold)
  int n_strips = n < 4096 ? 1 : std::min(32, n / 256);  // (round 5: merged strips are cheapest at 32 from 8 k points; 16 until round 4)
  if (const char* e = std::getenv("FLAME_DELAUNAY_STRIPS")) n_strips = std::max(1, std::min(std::atoi(e), std::min(64, n / 64 + 1)));  // (tests)
  if (std::getenv("FLAME_DELAUNAY_PROFILE"))
    std::fprintf(stderr, "[delaunay] prep %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_enter);
  if (n_strips > 1) {
    bool fits = true;
    // round 5: strips of their own points merged along their seams; the certified strips (whose output ORDER the fixtures' synthetic
    // graphs pin: FLAME_DELAUNAY_MERGE=0, flame_amd/synth.py) when that gives up
    const char* merge_env = std::getenv("FLAME_DELAUNAY_MERGE");
    if (!(merge_env && std::atoi(merge_env) == 0)) {
      if (triangulate_merge(in, n_strips, triangles, tri_capacity, n_triangles, edges, edge_capacity, n_edges, &fits))
        return fits ? FLAME_NLTGV2_OK : FLAME_NLTGV2_ERR_INVALID_ARG;
      *n_triangles = 0, *n_edges = 0, fits = true;
    }
    if (triangulate_strips(in, n_strips, triangles, tri_capacity, n_triangles, edges, edge_capacity, n_edges, &fits))
      return fits ? FLAME_NLTGV2_OK : FLAME_NLTGV2_ERR_INVALID_ARG;
    *n_triangles = 0, *n_edges = 0;
  }

  // ---- sequential: one triangulation of everything ------------------------------------------------------------------
  static thread_local Triangulator T;
  std::vector<int> all((size_t)n);
  std::iota(all.begin(), all.end(), 0);
  if (!triangulate_subset(in, all.data(), n, T)) return FLAME_NLTGV2_OK;

  // ---- output ---------------------------------------------------------------------------------------------
  int32_t nt = 0;
  for (const Tri& tr : T.t)
    if (tr.alive && !Triangulator::is_ghost(tr)) ++nt;
  *n_triangles = nt;
  if (triangles) {
    if (tri_capacity < nt) return FLAME_NLTGV2_ERR_INVALID_ARG;
    int32_t k = 0;
    for (const Tri& tr : T.t)
      if (tr.alive && !Triangulator::is_ghost(tr)) {
        triangles[3 * k] = tr.v[0], triangles[3 * k + 1] = tr.v[1], triangles[3 * k + 2] = tr.v[2];
        ++k;
      }
  }
  // edges: every real triangle contributes the edges whose opposite neighbour is a ghost or has a larger
  // index (each undirected edge exactly once), in triangle order
  int32_t ne = 0;
  for (size_t ti = 0; ti < T.t.size(); ++ti) {
    const Tri& tr = T.t[ti];
    if (!tr.alive || Triangulator::is_ghost(tr)) continue;
    for (int i = 0; i < 3; ++i) {
      const int nb = tr.n[(i + 2) % 3];  // neighbour across edge (v[i], v[i+1]) = opposite v[i+2]
      if (nb < 0 || Triangulator::is_ghost(T.t[(size_t)nb]) || (size_t)nb > ti) {
        if (edges) {
          if (ne >= edge_capacity) return FLAME_NLTGV2_ERR_INVALID_ARG;
          edges[2 * ne] = tr.v[i], edges[2 * ne + 1] = tr.v[(i + 1) % 3];
        }
        ++ne;
      }
    }
  }
  *n_edges = ne;
  return FLAME_NLTGV2_OK;
}
