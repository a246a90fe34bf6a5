// tests/cpp/facade_test.cc -- the C++ facade (reference call surface) end to end on the GPU, written
// the way a reference test for optimizers/ would read: build a Graph, call step(), compare with the
// CPU checker (oracle/liboracle_nltgv2.so, linked here as test infrastructure).
// Build+run: tests/test_cpp_facade.py.   Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "flame_hip/nltgv2_l1_graph_regularizer.hpp"

namespace dgraph = flame::optimizers::nltgv2_l1_graph_regularizer::hip;

extern "C" {
// oracle/nltgv2_oracle.c (test infrastructure)
struct nltgv2_params { float data_factor, step_x, step_q, theta, x_min, x_max; };
int nltgv2_oracle_run(const nltgv2_params*, flame_nltgv2_graph*, int);
int nltgv2_oracle_dual_step(const nltgv2_params*, flame_nltgv2_graph*);
float nltgv2_oracle_smoothness_cost(const nltgv2_params*, const flame_nltgv2_graph*);
float nltgv2_oracle_data_cost(const nltgv2_params*, const flame_nltgv2_graph*);
}

static unsigned long long sm(unsigned long long& s) {
  unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static float u01(unsigned long long& s) { return (float)(sm(s) >> 40) * (1.0f / 16777216.0f); }

// Jittered grid + the two diagonals' worth of edges (a planar triangulation), initialised with the
// conventions of Flame::syncGraph: x = x_bar = x_prev = data (flame.cc:2040-2048), alpha = 1/len,
// beta = 1, q = 0 (flame.cc:2087-2104).
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
  auto add = [&](int a, int b) {
    flame_hip::EdgeData e;
    if (sm(seed) & 1) { e.source = a, e.target = b; } else { e.source = b, e.target = a; }
    const float dx = g.vertices[a].pos_x - g.vertices[b].pos_x, dy = g.vertices[a].pos_y - g.vertices[b].pos_y;
    e.alpha = 1.0f / std::sqrt(dx * dx + dy * dy);
    g.edges.push_back(e);
  };
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      const int v = y * nx + x;
      if (x + 1 < nx) add(v, v + 1);
      if (y + 1 < ny) add(v, v + nx);
      if (x + 1 < nx && y + 1 < ny) add(v, v + nx + 1);
    }
  return g;
}

static int compare(const flame_hip::FlatGraph& g, flame_hip::FlatArrays& ref, const char* what) {
  int bad = 0;
  for (size_t v = 0; v < g.vertices.size(); ++v) {
    const flame_hip::VertexData& d = g.vertices[v];
    bad += std::memcmp(&d.x, &ref.x[v], 4) != 0 || std::memcmp(&d.w1, &ref.w1[v], 4) != 0 ||
           std::memcmp(&d.w2, &ref.w2[v], 4) != 0 || std::memcmp(&d.x_bar, &ref.x_bar[v], 4) != 0;
  }
  for (size_t e = 0; e < g.edges.size(); ++e) bad += std::memcmp(&g.edges[e].q1, &ref.q1[e], 4) != 0;
  std::printf("%-28s %s (%d mismatching elements)\n", what, bad ? "FAIL" : "ok", bad);
  return bad;
}

int main() {
  int fails = 0;
  dgraph::Params params;  // reference defaults
  const nltgv2_params cp = {params.data_factor, params.step_x, params.step_q, params.theta, params.x_min, params.x_max};
  flame_hip::FlatGraph graph = make_graph(40, 30, 42);
  flame_hip::FlatArrays ref;
  flame_hip::GraphAccess<flame_hip::FlatGraph>::pack(graph, &ref);
  flame_nltgv2_graph rv = ref.view();

  // (1) the reference's free function: step(params, &graph)
  dgraph::step(params, &graph);
  nltgv2_oracle_run(&cp, &rv, 1);
  fails += compare(graph, ref, "step(params,&graph)");

  // (2) pipeline form: DeviceGraph kept next to the Graph, many steps between host syncs
  dgraph::DeviceGraph dev;
  dev.upload(graph);
  dev.run(params, 150);
  dev.download(&graph);
  nltgv2_oracle_run(&cp, &rv, 150);
  fails += compare(graph, ref, "DeviceGraph::run(150)");

  // (3) costs
  const float sc = dgraph::smoothnessCost(params, graph), dc = dgraph::dataCost(params, graph);
  const float rsc = nltgv2_oracle_smoothness_cost(&cp, &rv), rdc = nltgv2_oracle_data_cost(&cp, &rv);
  const bool cost_ok = sc == rsc && dc == rdc;  // sequential float sums in edge / vertex order: exact
  std::printf("%-28s %s (%g vs %g, %g vs %g)\n", "smoothnessCost/dataCost", cost_ok ? "ok" : "FAIL", sc, rsc, dc, rdc);
  fails += !cost_ok;
  const float c = dgraph::cost(params, graph);
  fails += !(std::fabs(c - (sc + dc)) <= 1e-5f * std::fabs(c));

  // (4) internal::dualStep
  dgraph::internal::dualStep(params, &graph);
  nltgv2_oracle_dual_step(&cp, &rv);
  fails += compare(graph, ref, "internal::dualStep");

  // (5) errors are exceptions, not exit(1)
  bool threw = false;
  try {
    flame_hip::FlatGraph bad = graph;
    bad.edges[0].target = bad.edges[0].source;  // self loop
    dgraph::step(params, &bad);
  } catch (const flame_hip::Error& e) {
    threw = e.status == FLAME_NLTGV2_ERR_INVALID_ARG;
  }
  std::printf("%-28s %s\n", "invalid graph -> Error", threw ? "ok" : "FAIL");
  fails += !threw;
  return fails ? 1 : 0;
}
