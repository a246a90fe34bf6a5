// tests/cpp/solver_loop_test.cc -- flame_hip::SolverLoop (include/flame_hip/solver_loop.hpp): the reference's solver
// thread (flame.cc:99-112) with a stop flag and a joinable destructor, over the device image of a host graph.
//   * three frames of  lock -> readBack -> edit graph (data terms; then vertices and edges) -> markDirty -> unlock,
//     with a fixed iteration budget per frame so that the result can be compared with the CPU checker bit for bit;
//   * a graph edited WITHOUT markDirty() is refused by readBack() (and by DeviceGraph::download), not written to;
//   * free-running mode iterates until stop(); stop() / the destructor join in well under a second.
// Build+run: tests/test_cpp_facade.py.  Exit code 0 = pass, 77 = no usable HIP device.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "flame_hip/solver_loop.hpp"

namespace dgraph = flame::optimizers::nltgv2_l1_graph_regularizer::hip;

extern "C" {
struct nltgv2_params { float data_factor, step_x, step_q, theta, x_min, x_max; };
int nltgv2_oracle_run(const nltgv2_params*, flame_nltgv2_graph*, int);
}

static unsigned long long sm(unsigned long long& s) {
  unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static float u01(unsigned long long& s) { return (float)(sm(s) >> 40) * (1.0f / 16777216.0f); }

static void add_edge(flame_hip::FlatGraph* g, int a, int b, unsigned long long& seed) {
  flame_hip::EdgeData e;
  if (sm(seed) & 1) { e.source = a, e.target = b; } else { e.source = b, e.target = a; }
  const float dx = g->vertices[a].pos_x - g->vertices[b].pos_x, dy = g->vertices[a].pos_y - g->vertices[b].pos_y;
  e.alpha = 1.0f / std::sqrt(dx * dx + dy * dy);
  g->edges.push_back(e);
}
static flame_hip::FlatGraph make_graph(int nx, int ny, unsigned long long seed) {
  flame_hip::FlatGraph g;
  g.vertices.resize((size_t)nx * ny);
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      flame_hip::VertexData& v = g.vertices[(size_t)y * nx + x];
      v.pos_x = 6.0f * x + 5.0f * u01(seed), v.pos_y = 6.0f * y + 5.0f * u01(seed);
      v.data_term = (x < nx / 2 ? 0.6f + 0.01f * x : 1.4f - 0.005f * y) + 0.05f * (u01(seed) - 0.5f);
      v.x = v.x_bar = v.x_prev = v.data_term;
    }
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      const int v = y * nx + x;
      if (x + 1 < nx) add_edge(&g, v, v + 1, seed);
      if (y + 1 < ny) add_edge(&g, v, v + nx, seed);
      if (x + 1 < nx && y + 1 < ny) add_edge(&g, v, v + nx + 1, seed);
    }
  return g;
}

static int compare(const flame_hip::FlatGraph& g, const flame_hip::FlatGraph& ref, const char* what) {
  int bad = g.vertices.size() != ref.vertices.size() || g.edges.size() != ref.edges.size();
  for (size_t v = 0; !bad && v < g.vertices.size(); ++v)
    bad += std::memcmp(&g.vertices[v].x, &ref.vertices[v].x, 9 * sizeof(float)) != 0;  // x .. w2_prev
  for (size_t e = 0; !bad && e < g.edges.size(); ++e) bad += std::memcmp(&g.edges[e].q1, &ref.edges[e].q1, 3 * sizeof(float)) != 0;
  std::printf("%-44s %s\n", what, bad ? "FAIL" : "ok");
  return bad ? 1 : 0;
}
static void oracle_run(flame_hip::FlatGraph* g, const nltgv2_params& cp, int n) {
  flame_hip::FlatArrays a;
  flame_hip::GraphAccess<flame_hip::FlatGraph>::pack(*g, &a);
  flame_nltgv2_graph v = a.view();
  nltgv2_oracle_run(&cp, &v, n);
  flame_hip::GraphAccess<flame_hip::FlatGraph>::unpack(a, g);
}
template <class F>
static bool wait_for(F cond, int ms) {
  for (int i = 0; i < ms; ++i) {
    if (cond()) return true;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  return cond();
}

int main() {
  {
    flame_nltgv2_ctx* probe = nullptr;
    const int rc = flame_nltgv2_create(&probe, 0);
    if (rc != 0) {
      std::printf("%s\n", flame_nltgv2_status_string(rc));
      return 77;
    }
    flame_nltgv2_destroy(probe);
  }
  int fails = 0;
  const dgraph::Params params;
  const nltgv2_params cp = {params.data_factor, params.step_x, params.step_q, params.theta, params.x_min, params.x_max};
  unsigned long long seed = 99;
  flame_hip::FlatGraph graph = make_graph(30, 22, 7), ref = graph;
  std::mutex graph_mtx;

  {  // ---- (1) three frames with a fixed budget: 5 rounds of 20 iterations per upload ----------------------------------
    flame_hip::SolverLoop<flame_hip::FlatGraph> loop(&graph, &graph_mtx, params, 20, 5);
    loop.start();
    for (int frame = 0; frame < 3; ++frame) {
      const uint64_t want_iters = 100ull * (frame + 1);
      if (!wait_for([&] { return loop.iterations() >= want_iters; }, 3000)) {
        std::printf("frame %d: the loop did not reach %llu iterations (%llu) %s\n", frame, (unsigned long long)want_iters,
                    (unsigned long long)loop.iterations(), loop.error().c_str());
        return 1;
      }
      std::lock_guard<std::mutex> lock(graph_mtx);
      fails += !loop.readBack();
      oracle_run(&ref, cp, 100);
      char what[64];
      std::snprintf(what, sizeof what, "frame %d: readBack == checker after 100", frame);
      fails += compare(graph, ref, what);
      // edit under the lock, as Flame::update does: new data terms; in frame 1 also new vertices and edges
      for (size_t v = 0; v < graph.vertices.size(); v += 3) ref.vertices[v].data_term = graph.vertices[v].data_term += 0.02f * (u01(seed) - 0.5f);
      if (frame == 1) {
        for (int k = 0; k < 4; ++k) {
          flame_hip::VertexData nv;
          nv.pos_x = 200.0f + 6.0f * k, nv.pos_y = 3.0f + u01(seed);
          nv.data_term = nv.x = nv.x_bar = nv.x_prev = 1.0f + 0.1f * k;
          graph.vertices.push_back(nv);
          add_edge(&graph, (int)graph.vertices.size() - 1, 29 - k, seed);
          if (k) add_edge(&graph, (int)graph.vertices.size() - 1, (int)graph.vertices.size() - 2, seed);
        }
        ref = graph;
        // edited, not yet marked dirty: the device image is of the old graph -- nothing may be written into the new one
        const bool refused = !loop.readBack();
        std::printf("%-44s %s\n", "edited graph, no markDirty: readBack refused", refused ? "ok" : "FAIL");
        fails += !refused;
      }
      loop.markDirty();
    }
    fails += !wait_for([&] { return loop.iterations() >= 400; }, 3000);
    {
      std::lock_guard<std::mutex> lock(graph_mtx);
      fails += !loop.readBack();
      oracle_run(&ref, cp, 100);
      fails += compare(graph, ref, "after the last frame");
    }
    const auto t0 = std::chrono::steady_clock::now();
    loop.stop();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%-44s %s (%.1f ms, %llu uploads)\n", "stop() joins", ms < 1000.0 ? "ok" : "FAIL", ms, (unsigned long long)loop.uploads());
    fails += !(ms < 1000.0) || loop.uploads() != 4 || !loop.error().empty();
  }
  {  // ---- (2) free-running, as the reference's thread; the destructor stops and joins -----------------------------------
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t its = 0;
    {
      flame_hip::SolverLoop<flame_hip::FlatGraph> loop(&graph, &graph_mtx, params, 50);
      loop.start();
      fails += !wait_for([&] { return loop.iterations() >= 2000; }, 3000);
      std::lock_guard<std::mutex> lock(graph_mtx);
      fails += !loop.readBack();
      its = loop.iterations();
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    float mx = 0.0f;
    for (size_t v = 0; v < graph.vertices.size(); ++v) mx = std::fmax(mx, std::fabs(graph.vertices[v].x - graph.vertices[v].data_term));
    const bool ok = its >= 2000 && ms < 4000.0 && mx < 0.5f && mx > 0.0f;
    std::printf("%-44s %s (%llu iterations, whole block %.0f ms)\n", "free-running loop, destructor joins", ok ? "ok" : "FAIL",
                (unsigned long long)its, ms);
    fails += !ok;
  }
  {  // ---- (2b) stop() under the graph mutex, restart, state queries (advisor r02: the join used to deadlock against a loop
     //      blocked on that mutex; running() raced with start(); a loop that had ended could not be started again) ---------
    flame_hip::SolverLoop<flame_hip::FlatGraph> loop(&graph, &graph_mtx, params, 10);
    bool ok = !loop.running();
    loop.start();
    ok = ok && loop.running() && wait_for([&] { return loop.iterations() >= 50; }, 3000);
    const auto t0 = std::chrono::steady_clock::now();
    {
      std::lock_guard<std::mutex> lock(graph_mtx);  // the caller edits ...
      graph.vertices[0].data_term += 0.01f;
      loop.markDirty();                                        // ... the loop now wants the mutex for its re-upload ...
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
      loop.stop();                                             // ... and is stopped while the caller still holds it
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ok = ok && ms < 1000.0 && !loop.running();
    const uint64_t before = loop.iterations();
    loop.start();  // a stopped loop starts again
    ok = ok && loop.running() && wait_for([&] { return loop.iterations() >= before + 50; }, 3000);
    loop.stop();
    loop.stop();   // idempotent
    ok = ok && !loop.running() && loop.error().empty();
    std::printf("%-44s %s (stop under the mutex: %.1f ms)\n", "stop() with the graph mutex held, restart", ok ? "ok" : "FAIL", ms);
    fails += !ok;
  }
  {  // ---- (3) DeviceGraph::download refuses a graph that is not the uploaded one -----------------------------------------
    dgraph::DeviceGraph dev;
    dev.upload(graph, 5);
    flame_hip::FlatGraph other = graph;
    other.vertices.pop_back();
    bool threw1 = false, threw2 = false;
    try { dev.download(&other, 5); } catch (const flame_hip::Error& e) { threw1 = e.status == FLAME_NLTGV2_ERR_INVALID_ARG; }
    try { dev.download(&graph, 6); } catch (const flame_hip::Error& e) { threw2 = e.status == FLAME_NLTGV2_ERR_INVALID_ARG; }
    dev.download(&graph, 5);
    std::printf("%-44s %s\n", "download(): size / generation guard", threw1 && threw2 ? "ok" : "FAIL");
    fails += !(threw1 && threw2);
  }
  return fails ? 1 : 0;
}
