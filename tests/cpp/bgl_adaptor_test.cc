// tests/cpp/bgl_adaptor_test.cc -- include/flame_hip/bgl_adaptor.hpp end to end: a Graph with the
// reference's VertexData/EdgeData field names (nltgv2_l1_graph_regularizer.h:74-102) is built the way
// Flame::syncGraph builds it (add_vertex / add_edge, flame.cc:2035, 2096), stepped through the facade's
// reference-style free function step(params, &graph), and compared with the CPU checker run on the flat
// image the adaptor itself produced.  Compiled against tests/cpp/mock_boost (Boost is not installed).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace cv { struct Point2f { float x, y; }; }
// the reference's vertex / edge payloads (field names are what the adaptor reads)
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

#include "flame_hip/bgl_adaptor.hpp"

using Graph = boost::adjacency_list<boost::hash_setS, boost::hash_setS, boost::undirectedS, VertexData, EdgeData>;
namespace dgraph = flame::optimizers::nltgv2_l1_graph_regularizer::hip;

extern "C" {
struct nltgv2_params { float data_factor, step_x, step_q, theta, x_min, x_max; };
int nltgv2_oracle_run(const nltgv2_params*, flame_nltgv2_graph*, int);
}

int main() {
  Graph graph;
  const int nx = 24, ny = 18;
  std::vector<void*> vh((size_t)nx * ny);
  unsigned long long s = 7;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffff) / 65536.0f; };
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      VertexData v;
      v.pos = {6.0f * x + 5.0f * rnd(), 6.0f * y + 5.0f * rnd()};
      v.data_term = 0.8f + 0.02f * x + 0.1f * (rnd() - 0.5f);
      v.x = v.x_bar = v.x_prev = v.data_term;
      vh[(size_t)y * nx + x] = boost::add_vertex(v, graph);
    }
  auto add = [&](int a, int b) {
    EdgeData e;
    const float dx = graph[vh[a]].pos.x - graph[vh[b]].pos.x, dy = graph[vh[a]].pos.y - graph[vh[b]].pos.y;
    e.alpha = 1.0f / std::sqrt(dx * dx + dy * dy);
    if (rnd() < 0.5f) boost::add_edge(vh[a], vh[b], e, graph); else boost::add_edge(vh[b], vh[a], e, graph);
  };
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      const int v = y * nx + x;
      if (x + 1 < nx) add(v, v + 1);
      if (y + 1 < ny) add(v, v + nx);
      if (x + 1 < nx && y + 1 < ny) add(v, v + nx + 1);
    }

  // checker on the flat image the adaptor produces (vertices() order = hash order, edges() = list order)
  flame_hip::FlatArrays ref;
  flame_hip::GraphAccess<Graph>::pack(graph, &ref);
  flame_nltgv2_graph rv = ref.view();
  dgraph::Params params;
  const nltgv2_params cp = {params.data_factor, params.step_x, params.step_q, params.theta, params.x_min, params.x_max};
  nltgv2_oracle_run(&cp, &rv, 40);

  // the reference call, on the BGL graph:  step(params, &graph)  x 3, then a resident run of 37
  for (int i = 0; i < 3; ++i) dgraph::step(params, &graph);
  dgraph::DeviceGraph dev;
  dev.upload(graph);
  dev.run(params, 37);
  dev.download(&graph);

  flame_hip::FlatArrays got;
  flame_hip::GraphAccess<Graph>::pack(graph, &got);  // same walk order as `ref`
  int bad = 0;
  for (size_t v = 0; v < got.x.size(); ++v)
    bad += std::memcmp(&got.x[v], &ref.x[v], 4) != 0 || std::memcmp(&got.w1[v], &ref.w1[v], 4) != 0 ||
           std::memcmp(&got.w2_bar[v], &ref.w2_bar[v], 4) != 0 || std::memcmp(&got.x_prev[v], &ref.x_prev[v], 4) != 0;
  for (size_t e = 0; e < got.q1.size(); ++e) bad += std::memcmp(&got.q1[e], &ref.q1[e], 4) != 0 || std::memcmp(&got.q3[e], &ref.q3[e], 4) != 0;
  std::printf("bgl adaptor: V=%zu E=%zu, %d mismatching elements -> %s\n", got.x.size(), got.q1.size(), bad, bad ? "FAIL" : "ok");
  return bad ? 1 : 0;
}
