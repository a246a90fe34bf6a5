// tests/cpp/integration_switch_test.cc -- INTEGRATION.md section 0, compiled VERBATIM.
//
// The five fenced blocks of INTEGRATION.md section 0 (tagged `<!-- edit:1a -->` ... `<!-- edit:4 -->`) are extracted from the markdown
// by tests/test_cpp_facade.py into edit_1a.inc ... edit_4.inc and #included below, at the places of a mock `Flame` that correspond
// to the places of the reference the blocks name.  The mock's members have exactly the reference's types
// (/root/reference/src/flame/flame.h):
//     std::recursive_mutex update_mtx_;   flame.h:512
//     std::mutex pfs_mtx_;                flame.h:516
//     Graph graph_;                       flame.h:536   (boost::adjacency_list<hash_setS, hash_setS, undirectedS, VertexData, EdgeData>,
//     float graph_scale_;                 flame.h:537    nltgv2_l1_graph_regularizer.h:107-112; here over tests/cpp/mock_boost)
//     std::mutex graph_mtx_;              flame.h:539   <- a PLAIN mutex: SolverLoop<Graph> must take it as its default GraphMutex
//     std::mutex triangulator_mtx_;       flame.h:546
// and projectGraph / syncGraph are static members with the reference's parameter lists (flame.h:374-396), so the lock sites and calls
// of flame.cc:99-124 and 296-381 compile as the reference writes them.  Built with -std=c++11 -Wall -Wextra -Werror.
//
// On a GPU the program also runs: four update() calls against a free-running solver, then ~Flame with graph_mtx_ held (the
// reference's own destructor order, flame.cc:120-124) must return; exit code 77 = no usable device.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

#include <boost/graph/adjacency_list.hpp>

namespace cv { struct Point2f { float x, y; }; }

// ---- what flame.h sees through #include "flame/optimizers/nltgv2_l1_graph_regularizer.h" (h:74-129) -----------------------------
namespace flame {
namespace optimizers {
namespace nltgv2_l1_graph_regularizer {
struct VertexData {
  cv::Point2f pos;
  float x = 0.0f, w1 = 0.0f, w2 = 0.0f;
  float x_bar = 0.0f, w1_bar = 0.0f, w2_bar = 0.0f;
  float x_prev = 0.0f, w1_prev = 0.0f, w2_prev = 0.0f;
  float data_term = 0.0f, data_weight = 1.0f;
};
struct EdgeData {
  float alpha = 1.0f, beta = 1.0f;
  float q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
  bool valid = true;
};
using Graph = boost::adjacency_list<boost::hash_setS, boost::hash_setS, boost::undirectedS, VertexData, EdgeData>;
using VertexHandle = boost::graph_traits<Graph>::vertex_descriptor;
struct Params {
  float data_factor = 0.1f, step_x = 0.001f, step_q = 125.0f, theta = 0.25f, x_min = 0.0f, x_max = 10.0f;
};
}  // namespace nltgv2_l1_graph_regularizer
}  // namespace optimizers
}  // namespace flame

// ===== edit 1a: flame.h, below its includes ========================================================================================
#include "edit_1a.inc"

namespace flame {
namespace dgraph = optimizers::nltgv2_l1_graph_regularizer;  // flame.h:52
using Graph = dgraph::Graph;                                 // flame.h:54-60
using VertexHandle = dgraph::VertexHandle;

struct Params {  // params.h: only what the edited lines read
  bool do_nltgv2 = true;
  dgraph::Params rparams;
};
struct Opaque {};  // stands for EpipolarGeometry, Frame, maps, stats: passed through, never looked at

class Flame {
 public:
  Flame() : graph_scale_(1.0f), fnew_(new Opaque) {
    // ===== edit 2: flame.cc:99-112 =================================================================================================
#include "edit_2.inc"
  }
  ~Flame() {
    const auto t0 = std::chrono::steady_clock::now();
    {
      // ===== edit 3: flame.cc:120-124 ==============================================================================================
#include "edit_3.inc"
    }
    destructor_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    delete fnew_;
  }
  bool update() {
    std::lock_guard<std::recursive_mutex> lock(update_mtx_);  // flame.cc:136
    // ===== edit 4: flame.cc:296-381 ================================================================================================
#include "edit_4.inc"
    return sync_success;
  }

  static double destructor_ms;
  const std::vector<float>& idepths() const { return vtx_idepths_; }
  flame_hip::SolverLoop<Graph>* solver() { return solver_.get(); }
  size_t vertices() {
    std::lock_guard<std::mutex> lock(graph_mtx_);  // flame.cc:329 locks it this way
    return boost::num_vertices(graph_);
  }

 private:
  // flame.h:374-396, bodies: a stand-in that edits the graph the way the reference's do (add_vertex / add_edge, new data terms)
  static void projectGraph(const Params&, const Opaque&, const Opaque&, Graph* graph, float, Opaque*, Opaque*, Opaque*) {
    Graph::vertex_iterator vit, end;
    boost::tie(vit, end) = boost::vertices(*graph);
    for (; vit != end; ++vit) (*graph)[*vit].pos.x += 0.125f;  // flame.cc:1898-1900 moves the vertices
  }
  static bool syncGraph(const Params&, const Opaque&, const Opaque&, const Opaque&, const Opaque&, const Opaque&, Opaque*, Graph* graph,
                        float, Opaque*, Opaque*, Opaque*) {
    static unsigned long long s = 11;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffff) / 65536.0f; };
    if (boost::num_vertices(*graph) == 0) {  // first frame: a 20 x 15 grid
      const int nx = 20, ny = 15;
      std::vector<VertexHandle> vh((size_t)nx * ny);
      for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
          dgraph::VertexData v;
          v.pos = {6.0f * x + 5.0f * rnd(), 6.0f * y + 5.0f * rnd()};
          v.data_term = 0.8f + 0.02f * x + 0.2f * (rnd() - 0.5f);
          v.x = v.x_bar = v.x_prev = v.data_term;  // flame.cc:2046-2048
          vh[(size_t)y * nx + x] = boost::add_vertex(v, *graph);
        }
      auto add = [&](int a, int b) {
        dgraph::EdgeData e;
        const float dx = (*graph)[vh[a]].pos.x - (*graph)[vh[b]].pos.x, dy = (*graph)[vh[a]].pos.y - (*graph)[vh[b]].pos.y;
        e.alpha = 1.0f / std::sqrt(dx * dx + dy * dy);  // flame.cc:2102
        boost::add_edge(vh[a], vh[b], e, *graph);
      };
      for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
          const int v = y * nx + x;
          if (x + 1 < nx) add(v, v + 1);
          if (y + 1 < ny) add(v, v + nx);
          if (x + 1 < nx && y + 1 < ny) add(v, v + nx + 1);
        }
    } else {  // later frames: new data terms, one more vertex tied to an old one
      Graph::vertex_iterator vit, end;
      boost::tie(vit, end) = boost::vertices(*graph);
      VertexHandle first = *vit;
      for (; vit != end; ++vit) (*graph)[*vit].data_term += 0.01f * (rnd() - 0.5f);
      dgraph::VertexData v;
      v.pos = {130.0f + 3.0f * rnd(), 40.0f * rnd()};
      v.data_term = v.x = v.x_bar = v.x_prev = 1.0f;
      VertexHandle nv = boost::add_vertex(v, *graph);
      dgraph::EdgeData e;
      e.alpha = 0.05f;
      boost::add_edge(nv, first, e, *graph);
    }
    return true;
  }

  Params params_;
  std::recursive_mutex update_mtx_;  // flame.h:512
  std::mutex pfs_mtx_;               // flame.h:516
  Opaque epigeo_, Kinv_, pfs_, idepthmap_, feats_, feats_in_curr_, triangulator_, feat_to_vtx_, vtx_to_feat_, stats_;
  std::vector<float> vtx_idepths_;
  Graph graph_;                      // flame.h:536
  float graph_scale_;                // flame.h:537
  // ===== edit 1b: flame.h:538 ======================================================================================================
#include "edit_1b.inc"
  std::mutex graph_mtx_;             // flame.h:539
  std::mutex triangulator_mtx_;      // flame.h:546
  Opaque* fnew_;
};
double Flame::destructor_ms = -1.0;
}  // namespace flame

int main() {
  int fails = 0;
  try {
    flame::Flame* f = new flame::Flame;
    size_t want_v = 300;
    for (int frame = 0; frame < 4; ++frame, ++want_v) {
      const bool ok = f->update();
      const size_t V = f->vertices();
      // what update() extracted: the values readBack() brought home at the third lock site (frame 0: the graph went up a moment ago)
      bool finite = f->idepths().size() == V;
      for (float x : f->idepths()) finite = finite && std::isfinite(x) && x >= 0.0f && x <= 10.0f;
      const uint64_t before = f->solver()->iterations();
      const auto t0 = std::chrono::steady_clock::now();
      while (f->solver()->iterations() < before + 200 && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5))
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      const bool moving = f->solver()->iterations() >= before + 200 && f->solver()->error().empty();
      std::printf("frame %d: update %s, V = %zu (%s), extracted idepths %s, solver iterating on the new graph %s (%llu uploads)\n", frame,
                  ok ? "ok" : "FAIL", V, V == want_v ? "ok" : "FAIL", finite ? "ok" : "FAIL", moving ? "ok" : "FAIL",
                  (unsigned long long)f->solver()->uploads());
      fails += !ok + (V != want_v) + !finite + !moving;
    }
    delete f;  // edit 3: reset() under std::lock_guard<std::mutex>(graph_mtx_)
    const bool joined = flame::Flame::destructor_ms >= 0.0 && flame::Flame::destructor_ms < 1000.0;
    std::printf("~Flame with graph_mtx_ (std::mutex) held: %s (%.1f ms)\n", joined ? "ok" : "FAIL", flame::Flame::destructor_ms);
    fails += !joined;
  } catch (const flame_hip::Error& e) {
    std::printf("%s\n", e.what());
    return e.status == FLAME_NLTGV2_ERR_NO_DEVICE ? 77 : 1;
  }
  return fails ? 1 : 0;
}
