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

  // ---- edits between upload() and download(): values go back BY POSITION, so every edit must be refused -------------------
  // (a) an edge removed: the count differs; (b) removed and added again: both counts as uploaded, but the edge now stands at
  // the END of edges() -- the adaptor's identity hash sees it; (c) a vertex removed the way Flame::syncGraph does it
  // (clear_vertex + remove_vertex, flame.cc:2021-2023) and another added; (d) the untouched graph is accepted again after a
  // fresh upload.  Also: graph_traits / edge(u, v) / add_edge on an existing pair behave as Boost documents.
  int refuse_bad = 0;
  auto refused = [&](const char* what) {
    bool threw = false;
    try {
      dev.download(&graph);
    } catch (const flame_hip::Error& e) {
      threw = e.status == FLAME_NLTGV2_ERR_INVALID_ARG;
    }
    if (!threw) std::printf("FAIL: download() accepted a graph after %s\n", what), ++refuse_bad;
  };
  typedef boost::graph_traits<Graph>::edge_descriptor Edge;
  typedef boost::graph_traits<Graph>::vertex_descriptor Vertex;
  dev.upload(graph);
  Graph::edge_iterator e0, e1;
  boost::tie(e0, e1) = boost::edges(graph);
  ++e0, ++e0;
  const Edge victim = *e0;
  const Vertex vs = boost::source(victim, graph), vt = boost::target(victim, graph);
  const EdgeData kept = graph[victim];
  if (!boost::edge(vt, vs, graph).second || boost::add_edge(vt, vs, EdgeData(), graph).second) std::printf("FAIL: edge() / add_edge on an existing pair\n"), ++refuse_bad;
  boost::remove_edge(victim, graph);
  refused("remove_edge");
  boost::add_edge(vs, vt, kept, graph);
  if (boost::num_edges(graph) != got.q1.size()) ++refuse_bad;
  refused("remove_edge + add_edge (same counts, the edge moved to the end of edges())");
  dev.upload(graph);
  dev.download(&graph);  // a fresh upload is accepted
  const Vertex gone = vh[5];
  boost::clear_vertex(gone, graph);
  boost::remove_vertex(gone, graph);
  VertexData nv;
  nv.pos = {1.5f, 2.5f};
  const Vertex fresh = boost::add_vertex(nv, graph);
  EdgeData ne;
  boost::add_edge(fresh, vh[0], ne, graph), boost::add_edge(fresh, vh[1], ne, graph);
  refused("clear_vertex + remove_vertex + add_vertex");
  dev.upload(graph);
  dev.run(params, 5);
  dev.download(&graph);
  size_t deg_sum = 0;
  Graph::vertex_iterator v0, v1;
  for (boost::tie(v0, v1) = boost::vertices(graph); v0 != v1; ++v0) {
    Graph::adjacency_iterator a0, a1;
    for (boost::tie(a0, a1) = boost::adjacent_vertices(*v0, graph); a0 != a1; ++a0) ++deg_sum;
  }
  if (deg_sum != 2 * boost::num_edges(graph)) std::printf("FAIL: adjacent_vertices\n"), ++refuse_bad;
  std::printf("bgl adaptor: edits between upload and download refused, fresh uploads accepted -> %s\n", refuse_bad ? "FAIL" : "ok");
  return (bad || refuse_bad) ? 1 : 0;
}
