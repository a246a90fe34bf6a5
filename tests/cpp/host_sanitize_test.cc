// tests/cpp/host_sanitize_test.cc -- the HOST code of the library under AddressSanitizer + UndefinedBehaviorSanitizer
// (the optional sanitizer build of the CPU-side code).  No device is needed: what runs here is what
// the library runs on the host for every upload / frame --
//   * flame_amd/csrc/delaunay.cpp          flame_delaunay_triangulate: exact-predicate Bowyer-Watson (utils/delaunay.cc:31-77's
//                                          counterpart), on random, co-circular (grid), duplicated, collinear and invalid inputs;
//   * flame_amd/csrc/nltgv2_pack.hpp       build_layout: CSR, components, Morton walk, SELL-64, the patch / half-edge / vertex
//                                          rows, with and without row packing, host-expanded or not, on ragged,
//                                          empty, star-shaped and batched graphs;
//   * include/flame_hip/*.hpp              the facade's FlatGraph <-> flat array packing (GraphAccess).
// Built and run by `make -C flame_amd/csrc sanitize` and tests/test_sanitizers.py with
// -fsanitize=address,undefined -fno-sanitize-recover=all: any out-of-bounds access, use after free, signed overflow,
// misaligned or invalid shift aborts the program.  Exit code 0 = clean.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <utility>
#include <vector>

#include "flame_hip/nltgv2_l1_graph_regularizer.hpp"
#include "host_workers.hpp"
#include "nltgv2_pack.hpp"

using namespace flame_hip;

static unsigned long long sm(unsigned long long& s) {
  unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static float u01(unsigned long long& s) { return (float)(sm(s) >> 40) * (1.0f / 16777216.0f); }

static int fails = 0;
#define EXPECT(cond, what)                                   \
  do {                                                       \
    if (!(cond)) {                                           \
      std::printf("FAIL: %s (line %d)\n", what, __LINE__);   \
      ++fails;                                               \
    }                                                        \
  } while (0)

// Delaunay of `xy`; returns the status, fills tris / edges
static int triangulate(const std::vector<float>& xy, std::vector<int32_t>* tris, std::vector<int32_t>* edges) {
  const int32_t n = (int32_t)(xy.size() / 2);
  tris->assign((size_t)std::max(1, 2 * n) * 3, -1);
  edges->assign((size_t)std::max(1, 3 * n) * 2, -1);
  int32_t nt = 0, ne = 0;
  const int rc = flame_delaunay_triangulate(xy.data(), n, tris->data(), (int32_t)(tris->size() / 3), &nt, edges->data(),
                                            (int32_t)(edges->size() / 2), &ne);
  tris->resize((size_t)nt * 3), edges->resize((size_t)ne * 2);
  return rc;
}

static void check_triangulation(const std::vector<float>& xy, const char* name, bool expect_euler) {
  std::vector<int32_t> t, e;
  const int rc = triangulate(xy, &t, &e);
  EXPECT(rc == 0, name);
  const int32_t n = (int32_t)(xy.size() / 2);
  std::set<std::pair<int32_t, int32_t>> es;
  for (size_t i = 0; i + 1 < e.size(); i += 2) {
    EXPECT(e[i] >= 0 && e[i] < n && e[i + 1] >= 0 && e[i + 1] < n && e[i] != e[i + 1], "edge endpoints in range");
    es.insert({std::min(e[i], e[i + 1]), std::max(e[i], e[i + 1])});
  }
  EXPECT(es.size() * 2 == e.size(), "edges unique");
  for (size_t i = 0; i + 2 < t.size(); i += 3) {
    const double ax = xy[2 * t[i]], ay = xy[2 * t[i] + 1], bx = xy[2 * t[i + 1]], by = xy[2 * t[i + 1] + 1], cx = xy[2 * t[i + 2]],
                 cy = xy[2 * t[i + 2] + 1];
    EXPECT((bx - ax) * (cy - ay) - (by - ay) * (cx - ax) > 0, "triangles counter-clockwise");
    for (int k = 0; k < 3; ++k) {
      const int32_t u = t[i + k], v = t[i + (k + 1) % 3];
      EXPECT(es.count({std::min(u, v), std::max(u, v)}) == 1, "triangle sides are edges");
    }
  }
  if (expect_euler) EXPECT((long)n - (long)es.size() + (long)(t.size() / 3) == 1, "Euler: V - E + T = 1");
  std::printf("delaunay %-28s n=%6d  T=%6zu E=%6zu  %s\n", name, n, t.size() / 3, es.size(), rc == 0 ? "ok" : "FAIL");
}

struct HostGraph {
  std::vector<float> pos, x, w1, w2, xb, w1b, w2b, data, weight, alpha, beta, q1, q2, q3;
  std::vector<int32_t> src, dst;
  flame_nltgv2_graph view() {
    flame_nltgv2_graph g;
    std::memset(&g, 0, sizeof g);
    g.V = (int32_t)x.size(), g.E = (int32_t)src.size();
    g.pos = pos.data(), g.x = x.data(), g.w1 = w1.data(), g.w2 = w2.data(), g.x_bar = xb.data(), g.w1_bar = w1b.data(), g.w2_bar = w2b.data();
    g.data_term = data.data(), g.data_weight = weight.data(), g.src = src.data(), g.dst = dst.data();
    g.alpha = alpha.data(), g.beta = beta.data(), g.q1 = q1.data(), g.q2 = q2.data(), g.q3 = q3.data();
    return g;
  }
  void add_vertex(float px, float py) {
    pos.push_back(px), pos.push_back(py);
    x.push_back(1.f), w1.push_back(0.f), w2.push_back(0.f), xb.push_back(1.f), w1b.push_back(0.f), w2b.push_back(0.f);
    data.push_back(1.f), weight.push_back(1.f);
  }
  void add_edge(int32_t a, int32_t b) {
    src.push_back(a), dst.push_back(b);
    alpha.push_back(0.2f), beta.push_back(1.f), q1.push_back(0.f), q2.push_back(0.f), q3.push_back(0.f);
  }
};

static HostGraph delaunay_graph(int nx, int ny, unsigned long long seed, float ox = 0.f) {
  HostGraph h;
  std::vector<float> xy;
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      const float px = ox + 6.f * x + 5.f * u01(seed), py = 6.f * y + 5.f * u01(seed);
      xy.push_back(px), xy.push_back(py);
      h.add_vertex(px, py);
    }
  std::vector<int32_t> t, e;
  triangulate(xy, &t, &e);
  for (size_t i = 0; i + 1 < e.size(); i += 2) {
    if (sm(seed) & 1) h.add_edge(e[i], e[i + 1]); else h.add_edge(e[i + 1], e[i]);
  }
  return h;
}

static void check_layout(HostGraph& h, const char* name) {
  flame_nltgv2_graph g = h.view();
  for (int rowpack = 0; rowpack <= 1; ++rowpack)
      for (int host_expand = 0; host_expand <= 1; ++host_expand) {
        PackedLayout L;
        const int rc = build_layout(&g, &L, host_expand != 0, rowpack != 0, 0x7fffffff);
        EXPECT(rc == 0, name);
        if (rc != 0) continue;
        if (host_expand) {
          build_tv_rows(&L);
          build_patch_rows2(&L);  // (E2): two half-edges per lane (not for a graph with a vertex of more than 32 edges)
          EXPECT(L.wg2_walked, "the second patch walk ran");
          if (L.wg2_ok) {
            EXPECT(L.wg2_meta.size() == (size_t)L.wg2_count * kWave && L.wg2_slot.size() == 2 * L.wg2_meta.size(), "two-half-edge rows sized");
            EXPECT(L.wg2_count <= L.wg_count || !L.wg_rowpack, "two half-edges per lane need no more waves than one");
          } else {
            EXPECT(L.max_degree > 32 || g.V == 0, "the two-half-edge form refuses only hubs of more than 32 edges (or nothing)");
          }
          // every edge occupies exactly two slots of the SELL rows (one per endpoint)
          std::vector<int> seen((size_t)g.E, 0);
          for (size_t i = 0; i < L.rec_edge.size(); ++i)
            if (L.rec_edge[i] >= 0) {
              EXPECT(L.rec_edge[i] < g.E, "slot edge id in range");
              if (L.rec_edge[i] < g.E) seen[(size_t)L.rec_edge[i]]++;
            }
          bool two = true;
          for (int v : seen) two = two && v == 2;
          EXPECT(two, "every edge in exactly two slots");
          if (L.wg_ok) {
            EXPECT(L.wg_meta.size() == (size_t)L.wg_count * kWave, "patch rows sized");
            EXPECT(L.wg_info.size() >= (size_t)L.wg_count * 4, "patch info sized");
          }
        }
      }
  std::printf("layout   %-28s V=%6d E=%6d                 ok\n", name, g.V, g.E);
}

int main() {
  unsigned long long seed = 42;
  // ---- worker pool: regions of different sizes back to back (a worker that wakes up late must never run a job of the
  // NEXT region, and no job may be counted twice: every region has its own state, host_workers.hpp) ---------------------
  {
    long long total = 0, expect = 0;
    bool once = true;
    for (int r = 0; r < 4000; ++r) {
      const int n = 1 + (int)(sm(seed) % 33);
      std::vector<int> hit((size_t)n, 0);  // (dies at the end of the iteration: a stale worker writing here is a use after free)
      flame_hip::Workers::get().run(n, [&](int i) { __atomic_add_fetch(&hit[(size_t)i], 1, __ATOMIC_RELAXED); });
      for (int i = 0; i < n; ++i) once = once && hit[(size_t)i] == 1, total += hit[(size_t)i];
      expect += n;
    }
    EXPECT(once && total == expect, "worker pool: every job of every region ran exactly once");
    std::printf("workers  %d threads, 4000 regions of 1..33 jobs                  %s\n", flame_hip::Workers::get().threads(), once && total == expect ? "ok" : "FAIL");
  }
  // ---- Delaunay ---------------------------------------------------------------------------------------------
  {
    std::vector<float> xy;
    for (int i = 0; i < 4000; ++i) xy.push_back(640.f * u01(seed)), xy.push_back(480.f * u01(seed));
    check_triangulation(xy, "random 4000", true);
  }
  {
    std::vector<float> xy;  // exact grid: every cell co-circular (the incircle ties must be broken consistently)
    for (int y = 0; y < 40; ++y)
      for (int x = 0; x < 50; ++x) xy.push_back(8.f * x), xy.push_back(8.f * y);
    check_triangulation(xy, "co-circular grid 50x40", true);
  }
  {
    std::vector<float> xy;  // duplicates and points on edges / the hull line
    for (int i = 0; i < 300; ++i) {
      const float px = (float)(int)(40.f * u01(seed)), py = (float)(int)(30.f * u01(seed));
      xy.push_back(px), xy.push_back(py);
      if (i % 7 == 0) xy.push_back(px), xy.push_back(py);
    }
    check_triangulation(xy, "integer points + duplicates", false);
  }
  {
    std::vector<float> xy;  // all collinear: no triangle
    for (int i = 0; i < 50; ++i) xy.push_back(3.f * i), xy.push_back(1.5f * i);
    std::vector<int32_t> t, e;
    const int rc = triangulate(xy, &t, &e);
    EXPECT(rc == 0 && t.empty(), "collinear input: no triangles, no error");
    std::printf("delaunay %-28s n=%6d  T=%6zu                 %s\n", "collinear", 50, t.size() / 3, rc == 0 ? "ok" : "FAIL");
  }
  {
    std::vector<float> xy = {0.f, 0.f, 1.f, NAN, 2.f, 2.f, 3.f, 0.f};
    std::vector<int32_t> t, e;
    EXPECT(triangulate(xy, &t, &e) != 0, "NaN coordinate is refused");
    std::vector<float> tiny = {0.f, 0.f, 1.f, 0.f};
    EXPECT(triangulate(tiny, &t, &e) == 0 && t.empty(), "two points: empty result");
    int32_t nt = 0, ne = 0;
    EXPECT(flame_delaunay_triangulate(nullptr, 5, nullptr, 0, &nt, nullptr, 0, &ne) != 0, "null input is refused");
    // capacity too small: refused, nothing written past the capacity
    std::vector<float> sq = {0.f, 0.f, 4.f, 0.f, 4.f, 4.f, 0.f, 4.f, 2.f, 1.f};
    int32_t tb[3] = {-7, -7, -7}, eb[2] = {-7, -7};
    const int rc = flame_delaunay_triangulate(sq.data(), 5, tb, 1, &nt, eb, 1, &ne);
    EXPECT(rc != 0, "too small an output capacity is refused");
    std::printf("delaunay %-28s                                 ok\n", "invalid / tiny inputs");
  }
  // ---- layouts ------------------------------------------------------------------------------------------------
  {
    HostGraph g = delaunay_graph(40, 30, 7);
    check_layout(g, "delaunay 40x30");
    HostGraph b = delaunay_graph(20, 15, 11);  // a batch: three disjoint frames in one graph
    for (int k = 1; k < 3; ++k) {
      HostGraph f = delaunay_graph(20, 15, 11 + k, 200.f * k);
      const int32_t off = (int32_t)b.x.size();
      for (size_t i = 0; i < f.x.size(); ++i) b.add_vertex(f.pos[2 * i], f.pos[2 * i + 1]);
      for (size_t i = 0; i < f.src.size(); ++i) b.add_edge(f.src[i] + off, f.dst[i] + off);
    }
    check_layout(b, "batch of 3 frames");
    HostGraph star;  // one hub of 70 edges (> 64: no patch rows), a hub of 20, isolated vertices, parallel edges
    for (int i = 0; i < 120; ++i) star.add_vertex(10.f * u01(seed), 10.f * u01(seed));
    for (int i = 1; i <= 70; ++i) star.add_edge(0, i);
    for (int i = 71; i <= 90; ++i) star.add_edge(i, 71 == i ? 72 : 71);
    star.add_edge(100, 101), star.add_edge(101, 100), star.add_edge(100, 101);
    check_layout(star, "hubs / isolated / parallel");
    HostGraph empty;
    check_layout(empty, "empty graph");
    HostGraph one;
    one.add_vertex(1.f, 2.f);
    check_layout(one, "single vertex");
  }
  // ---- facade packing -----------------------------------------------------------------------------------------
  {
    FlatGraph fg;
    fg.vertices.resize(5);
    for (size_t i = 0; i < fg.vertices.size(); ++i) fg.vertices[i].pos_x = (float)i, fg.vertices[i].x = 0.5f * i;
    EdgeData e;
    e.source = 0, e.target = 3, e.q1 = 0.25f;
    fg.edges.push_back(e);
    FlatArrays a;
    GraphAccess<FlatGraph>::pack(fg, &a);
    a.x[2] = 9.f, a.q1[0] = -1.f;
    GraphAccess<FlatGraph>::unpack(a, &fg);
    EXPECT(fg.vertices[2].x == 9.f && fg.edges[0].q1 == -1.f, "FlatGraph pack / unpack round trip");
    std::printf("facade   %-28s                                 ok\n", "FlatGraph <-> flat arrays");
  }
  std::printf(fails ? "%d check(s) FAILED\n" : "sanitized host code: all ok\n", fails);
  return fails ? 1 : 0;
}
