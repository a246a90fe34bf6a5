// flame_hip/nltgv2_l1_graph_regularizer.hpp -- C++11 facade over the C-ABI (flame_nltgv2.h) that
// keeps the reference's call surface:
//
//   reference  /root/reference/src/flame/optimizers/nltgv2_l1_graph_regularizer.h
//     namespace flame::optimizers::nltgv2_l1_graph_regularizer
//       struct Params                                   h:121-129
//       void  step(const Params&, Graph*)               h:134
//       float smoothnessCost / dataCost / cost          h:139-151
//       internal::dualStep / primalStep / extraGradientStep   h:158-168
//
//   here       namespace flame::optimizers::nltgv2_l1_graph_regularizer::hip  -- same names, same
//              argument meaning, same Params fields and defaults.  A maintainer switches a call site
//              by adding `::hip` (or a namespace alias).
//
// Host side stays ordinary C++: the facade is header-only, needs no HIP headers and works with any
// graph container for which flame_hip::GraphAccess<Graph> is specialised:
//   * flame_hip/bgl_adaptor.hpp      the reference's boost::adjacency_list Graph (needs Boost)
//   * flame_hip::FlatGraph (below)   a dependency-free container with the reference's field names
//
// Errors: the reference aborts the process (FLAME_ASSERT -> exit(1), assert.h:111); the facade throws
// flame_hip::Error carrying the C-ABI status instead.  Nothing is ever computed on the CPU here.
#ifndef FLAME_HIP_NLTGV2_L1_GRAPH_REGULARIZER_HPP_
#define FLAME_HIP_NLTGV2_L1_GRAPH_REGULARIZER_HPP_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "flame_nltgv2.h"

namespace flame_hip {

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string& what)
      : std::runtime_error(what + ": " + flame_nltgv2_status_string(s)), status(s) {}
};

// Flat image of a Graph in boost::vertices()/boost::edges() order; what upload/download move.
struct FlatArrays {
  std::vector<float> pos, x, w1, w2, x_bar, w1_bar, w2_bar, x_prev, w1_prev, w2_prev, data_term, data_weight;
  std::vector<int32_t> src, dst;
  std::vector<float> alpha, beta, q1, q2, q3;
  void resize(size_t V, size_t E) {
    pos.resize(2 * V);
    for (auto* a : {&x, &w1, &w2, &x_bar, &w1_bar, &w2_bar, &x_prev, &w1_prev, &w2_prev, &data_term, &data_weight})
      a->resize(V);
    src.resize(E), dst.resize(E);
    for (auto* a : {&alpha, &beta, &q1, &q2, &q3}) a->resize(E);
  }
  flame_nltgv2_graph view() {
    flame_nltgv2_graph g;
    g.V = static_cast<int32_t>(x.size()), g.E = static_cast<int32_t>(src.size());
    g.pos = pos.data();
    g.x = x.data(), g.w1 = w1.data(), g.w2 = w2.data();
    g.x_bar = x_bar.data(), g.w1_bar = w1_bar.data(), g.w2_bar = w2_bar.data();
    g.x_prev = x_prev.data(), g.w1_prev = w1_prev.data(), g.w2_prev = w2_prev.data();
    g.data_term = data_term.data(), g.data_weight = data_weight.data();
    g.src = src.data(), g.dst = dst.data();
    g.alpha = alpha.data(), g.beta = beta.data();
    g.q1 = q1.data(), g.q2 = q2.data(), g.q3 = q3.data();
    return g;
  }
};

// Specialise for a graph container:
//   static void pack(const Graph&, FlatArrays*)      vertices()/edges() order; (src,dst) = (source,target)
//   static void unpack(const FlatArrays&, Graph*)    writes x,w,x_bar,w_bar,x_prev,w_prev,q back
//   static void size(const Graph&, size_t* V, size_t* E)
//   static uint64_t identity(const Graph&)           a hash of WHICH objects stand at which position of vertices()/edges()
//                                                    (0 where positions are stable by construction): download() refuses a graph
//                                                    whose identity differs from the uploaded one's -- an edit that leaves the
//                                                    counts alone (remove_edge + add_edge) still moves objects to other positions
template <class Graph>
struct GraphAccess;

inline uint64_t mix_identity(uint64_t h, uint64_t v) {  // (splitmix64 step: order-sensitive running hash)
  h += 0x9e3779b97f4a7c15ull + v;
  h = (h ^ (h >> 30)) * 0xbf58476d1ce4e5b9ull;
  h = (h ^ (h >> 27)) * 0x94d049bb133111ebull;
  return h ^ (h >> 31);
}

// Dependency-free graph container with the reference's VertexData/EdgeData field names (h:74-102).
struct VertexData {
  float pos_x = 0.0f, pos_y = 0.0f;  // cv::Point2f pos
  float x = 0.0f, w1 = 0.0f, w2 = 0.0f;
  float x_bar = 0.0f, w1_bar = 0.0f, w2_bar = 0.0f;
  float x_prev = 0.0f, w1_prev = 0.0f, w2_prev = 0.0f;
  float data_term = 0.0f, data_weight = 1.0f;
};
struct EdgeData {
  int32_t source = 0, target = 0;  // boost::source / boost::target
  float alpha = 1.0f, beta = 1.0f;
  float q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
  bool valid = true;
};
struct FlatGraph {
  std::vector<VertexData> vertices;
  std::vector<EdgeData> edges;
};

template <>
struct GraphAccess<FlatGraph> {
  static void pack(const FlatGraph& g, FlatArrays* f) {
    f->resize(g.vertices.size(), g.edges.size());
    for (size_t v = 0; v < g.vertices.size(); ++v) {
      const VertexData& d = g.vertices[v];
      f->pos[2 * v] = d.pos_x, f->pos[2 * v + 1] = d.pos_y;
      f->x[v] = d.x, f->w1[v] = d.w1, f->w2[v] = d.w2;
      f->x_bar[v] = d.x_bar, f->w1_bar[v] = d.w1_bar, f->w2_bar[v] = d.w2_bar;
      f->x_prev[v] = d.x_prev, f->w1_prev[v] = d.w1_prev, f->w2_prev[v] = d.w2_prev;
      f->data_term[v] = d.data_term, f->data_weight[v] = d.data_weight;
    }
    for (size_t e = 0; e < g.edges.size(); ++e) {
      const EdgeData& d = g.edges[e];
      f->src[e] = d.source, f->dst[e] = d.target;
      f->alpha[e] = d.alpha, f->beta[e] = d.beta;
      f->q1[e] = d.q1, f->q2[e] = d.q2, f->q3[e] = d.q3;
    }
  }
  static void size(const FlatGraph& g, size_t* V, size_t* E) { *V = g.vertices.size(), *E = g.edges.size(); }
  static uint64_t identity(const FlatGraph&) { return 0; }  // (vector positions: stable by construction)
  static void unpack(const FlatArrays& f, FlatGraph* g) {
    for (size_t v = 0; v < g->vertices.size(); ++v) {
      VertexData& d = g->vertices[v];
      d.x = f.x[v], d.w1 = f.w1[v], d.w2 = f.w2[v];
      d.x_bar = f.x_bar[v], d.w1_bar = f.w1_bar[v], d.w2_bar = f.w2_bar[v];
      d.x_prev = f.x_prev[v], d.w1_prev = f.w1_prev[v], d.w2_prev = f.w2_prev[v];
    }
    for (size_t e = 0; e < g->edges.size(); ++e) {
      EdgeData& d = g->edges[e];
      d.q1 = f.q1[e], d.q2 = f.q2[e], d.q3 = f.q3[e];
    }
  }
};

// utils::Delaunay counterpart (src/flame/utils/delaunay.{h,cc}): triangles()/edges() of the Delaunay
// triangulation of `vertices_xy` (x0,y0,x1,y1,...), exact predicates, host code.
inline void delaunayTriangulate(const std::vector<float>& vertices_xy, std::vector<int32_t>* triangles,
                                std::vector<int32_t>* edges) {
  const int32_t n = static_cast<int32_t>(vertices_xy.size() / 2);
  int32_t nt = 0, ne = 0;
  std::vector<int32_t> t(static_cast<size_t>(n > 0 ? 6 * n : 3)), e(static_cast<size_t>(n > 0 ? 6 * n : 2));
  const int rc = flame_delaunay_triangulate(vertices_xy.data(), n, t.data(), static_cast<int32_t>(t.size() / 3), &nt,
                                            e.data(), static_cast<int32_t>(e.size() / 2), &ne);
  if (rc != 0) throw Error(rc, "flame_delaunay_triangulate");
  if (triangles) triangles->assign(t.begin(), t.begin() + 3 * nt);
  if (edges) edges->assign(e.begin(), e.begin() + 2 * ne);
}

}  // namespace flame_hip

namespace flame {
namespace optimizers {
namespace nltgv2_l1_graph_regularizer {
namespace hip {

// == struct Params h:121-129: layout-identical to flame_nltgv2_params, same defaults.
struct Params {
  float data_factor = 0.1f;  // lambda in the TV literature.
  float step_x = 0.001f;     // Primal step size.
  float step_q = 125.0f;     // Dual step size.
  float theta = 0.25f;       // Extra gradient step size.
  float x_min = 0.0f;        // Feasible set.
  float x_max = 10.0f;
};
static_assert(sizeof(Params) == sizeof(flame_nltgv2_params), "Params must mirror the C-ABI struct");

inline flame_nltgv2_params to_c(const Params& p) {
  flame_nltgv2_params c;
  c.data_factor = p.data_factor, c.step_x = p.step_x, c.step_q = p.step_q;
  c.theta = p.theta, c.x_min = p.x_min, c.x_max = p.x_max;
  return c;
}

// Device image of ONE Graph: what the pipeline keeps next to `Graph graph_` (flame.h:536).  Not
// thread-safe -- hold graph_mtx_ (flame.h:539) around every call exactly as the reference does around
// step() and the graph edits (flame.cc:103, 302, 309, 329, 365).
class DeviceGraph {
 public:
  explicit DeviceGraph(int device = 0) : ctx_(nullptr) {
    const int rc = flame_nltgv2_create(&ctx_, device);
    if (rc != 0) throw flame_hip::Error(rc, "flame_nltgv2_create");
  }
  ~DeviceGraph() { flame_nltgv2_destroy(ctx_); }
  DeviceGraph(const DeviceGraph&) = delete;
  DeviceGraph& operator=(const DeviceGraph&) = delete;

  // After Flame::syncGraph changed vertices / edges / data (flame.cc:1940-2188).
  // `generation`: any caller-side counter of graph edits; download() of another generation is refused.
  template <class Graph>
  void upload(const Graph& graph, uint64_t generation = 0) {
    flame_hip::GraphAccess<Graph>::pack(graph, &flat_);
    flame_nltgv2_graph v = flat_.view();
    check(flame_nltgv2_upload_graph(ctx_, &v), "upload_graph");
    uploaded_v_ = flat_.x.size(), uploaded_e_ = flat_.src.size(), generation_ = generation, uploaded_ = true;
    identity_ = flame_hip::GraphAccess<Graph>::identity(graph);
  }
  // Before Flame::update reads x, w1, w2 (flame.cc:372-380) or edits the graph.  Values go back to the objects they
  // came from BY POSITION in vertices()/edges() order, so the graph must be the one that was uploaded: a graph whose
  // vertex or edge count differs, or (when the caller tracks edits) whose generation differs, is refused
  // (FLAME_NLTGV2_ERR_INVALID_ARG) instead of receiving values that belong to other vertices.
  template <class Graph>
  void download(Graph* graph, uint64_t generation = 0) {
    if (!matches(*graph, generation)) throw flame_hip::Error(FLAME_NLTGV2_ERR_INVALID_ARG, "download: the graph changed since upload()");
    flame_nltgv2_graph v = flat_.view();
    check(flame_nltgv2_download_state(ctx_, &v), "download_state");
    flame_hip::GraphAccess<Graph>::unpack(flat_, graph);
  }
  template <class Graph>
  bool matches(const Graph& graph, uint64_t generation = 0) const {
    size_t V = 0, E = 0;
    flame_hip::GraphAccess<Graph>::size(graph, &V, &E);
    return uploaded_ && V == uploaded_v_ && E == uploaded_e_ && generation == generation_ &&
           flame_hip::GraphAccess<Graph>::identity(graph) == identity_;
  }
  uint64_t generation() const { return generation_; }

  // Per-frame warm-start synchronisation: Flame::syncGraph's graph edits (flame.cc:1985-2121) applied to the
  // device image; see flame_nltgv2_sync_graph.  `edges` = triangulator->edges() as index pairs.
  void sync(const std::vector<int32_t>& feat_id, const std::vector<float>& pos_xy, const std::vector<float>& data_term,
            const std::vector<float>& data_weight, const std::vector<int32_t>& edges, bool check_sticky_obstacles = false,
            const float* init_x = nullptr, float init_graph_scale = 0.0f, bool edges_unique = false) {
    flame_nltgv2_sync_input in{};
    in.init_graph_scale = init_graph_scale;  // > 0: NaN entries of init_x -> neighbours' mean (flame.cc:2133-2158)
    in.edges_unique = edges_unique ? 1 : 0;  // the caller vouches (a triangulator's edge list): no search for repeated pairs
    in.V = static_cast<int32_t>(feat_id.size());
    in.feat_id = feat_id.data(), in.pos = pos_xy.data();
    in.data_term = data_term.data(), in.data_weight = data_weight.data();
    in.init_x = init_x;
    in.E = static_cast<int32_t>(edges.size() / 2);
    in.edges = edges.data();
    in.check_sticky_obstacles = check_sticky_obstacles ? 1 : 0;
    in.sticky_threshold = 0.25f;  // flame.cc:2011
    check(flame_nltgv2_sync_graph(ctx_, &in), "sync_graph");
  }

  // The same sync in two halves (flame_nltgv2_sync_prepare / _commit): prepare under graph_mtx_ right after the triangulation --
  // it only reads the device image, the solver thread goes on stepping --, commit under graph_mtx_ when Flame::update would have
  // left syncGraph.  init_from_map: new vertices start at the prediction of the dense map the last interpolateMesh left on the
  // device (init_with_prediction, flame.cc:2131) instead of a host gather into init_x.
  void syncPrepare(const std::vector<int32_t>& feat_id, const std::vector<float>& pos_xy, const std::vector<float>& data_term,
                   const std::vector<float>& data_weight, const std::vector<int32_t>& edges, bool check_sticky_obstacles = false,
                   const float* init_x = nullptr, float init_graph_scale = 0.0f, bool edges_unique = false, bool init_from_map = false) {
    flame_nltgv2_sync_input in{};
    in.init_graph_scale = init_graph_scale, in.edges_unique = edges_unique ? 1 : 0, in.init_from_map = init_from_map ? 1 : 0;
    in.V = static_cast<int32_t>(feat_id.size());
    in.feat_id = feat_id.data(), in.pos = pos_xy.data();
    in.data_term = data_term.data(), in.data_weight = data_weight.data();
    in.init_x = init_x;
    in.E = static_cast<int32_t>(edges.size() / 2);
    in.edges = edges.data();
    in.check_sticky_obstacles = check_sticky_obstacles ? 1 : 0;
    in.sticky_threshold = 0.25f;  // flame.cc:2011
    check(flame_nltgv2_sync_prepare(ctx_, &in), "sync_prepare");
  }
  void syncCommit() { check(flame_nltgv2_sync_commit(ctx_), "sync_commit"); }
  // interpolateMesh in two halves: the rasteriser and the copy-out run on a side stream while the solver steps again.
  void interpolateMeshBegin(const std::vector<int32_t>& triangles, int rows, int cols, float graph_scale, const uint8_t* tri_validity = nullptr) {
    check(flame_nltgv2_interpolate_mesh_begin(ctx_, triangles.data(), static_cast<int32_t>(triangles.size() / 3), tri_validity, rows, cols,
                                              graph_scale), "interpolate_mesh_begin");
  }
  int interpolateMeshEnd(const float** idepthmap) {
    int32_t coverage = 0;
    check(flame_nltgv2_interpolate_mesh_end(ctx_, idepthmap, &coverage), "interpolate_mesh_end");
    return coverage;
  }

  // utils::interpolateMesh (utils/image_utils.cc:373-396) at its call site flame.cc:409-415: rasterises
  // x*graph_scale of the device state over `triangles` (triangulator->triangles(), 3 ints each) into a
  // rows*cols float image (NaN = uncovered); returns the coverage count of flame.cc:428-437.
  int interpolateMesh(const std::vector<int32_t>& triangles, int rows, int cols, float graph_scale, float* idepthmap,
                      const uint8_t* tri_validity = nullptr) {
    int32_t coverage = 0;
    check(flame_nltgv2_interpolate_mesh(ctx_, triangles.data(), static_cast<int32_t>(triangles.size() / 3), tri_validity,
                                        rows, cols, graph_scale, idepthmap, &coverage),
          "interpolate_mesh");
    return coverage;
  }

  void step(const Params& p) { run(p, 1); }
  void run(const Params& p, int n_iters) {
    const flame_nltgv2_params c = to_c(p);
    check(flame_nltgv2_run(ctx_, &c, n_iters), "run");
  }
  void dualStep(const Params& p) { const flame_nltgv2_params c = to_c(p); check(flame_nltgv2_dual_step(ctx_, &c), "dualStep"); }
  void primalStep(const Params& p) { const flame_nltgv2_params c = to_c(p); check(flame_nltgv2_primal_step(ctx_, &c), "primalStep"); }
  void extraGradientStep(const Params& p) {
    const flame_nltgv2_params c = to_c(p);
    check(flame_nltgv2_extragradient_step(ctx_, &c), "extraGradientStep");
  }
  void savePrev() { check(flame_nltgv2_save_prev(ctx_), "savePrev"); }
  float smoothnessCost(const Params& p) { float s = 0, d = 0; costs(p, &s, &d); return s; }
  float dataCost(const Params& p) { float s = 0, d = 0; costs(p, &s, &d); return d; }
  float cost(const Params& p) { float s = 0, d = 0; costs(p, &s, &d); return s + d; }
  void costs(const Params& p, float* smooth, float* data) {
    const flame_nltgv2_params c = to_c(p);
    check(flame_nltgv2_costs(ctx_, &c, smooth, data), "costs");
  }
  // Asynchronous use (multi-GPU hosts, frame_gather.hpp): the solver's stream, a standing device target that every run
  // leaves x * scale in (the read-back of flame.cc:372-380, on the device), enqueue-only runs and the matching wait.
  void setStream(void* hip_stream) { check(flame_nltgv2_set_stream(ctx_, hip_stream), "set_stream"); }
  void setExportTarget(void* dst_device, float scale) { check(flame_nltgv2_set_export_target(ctx_, dst_device, scale), "set_export_target"); }
  void runAsync(const Params& p, int n_iters) {
    const flame_nltgv2_params c = to_c(p);
    check(flame_nltgv2_run_async(ctx_, &c, n_iters), "run_async");
  }
  void sync() { check(flame_nltgv2_sync(ctx_), "sync"); }
  // A run that goes on until the next call that needs the solver settled asks it to stop (flame_nltgv2_run_open: the reference's solver
  // thread is `while (true) step()`, flame.cc:99-112); at most max_iters (even) iterations.  false: not applicable to this graph or
  // configuration -- nothing was enqueued, use runAsync.
  bool runOpen(const Params& p, int max_iters) {
    const flame_nltgv2_params c = to_c(p);
    int32_t opened = 0;
    check(flame_nltgv2_run_open(ctx_, &c, max_iters, &opened), "run_open");
    return opened != 0;
  }
  // Iterations applied to the state by all runs so far; an open run is counted when it has been settled (*open_in_flight: one is not yet).
  // Never waits: right after a call that settled the solver this is the exact number the state has seen.
  uint64_t iterations(bool* open_in_flight = nullptr) {
    int64_t total = 0;
    int32_t open_ = 0;
    check(flame_nltgv2_iterations(ctx_, &total, &open_), "iterations");
    if (open_in_flight) *open_in_flight = open_ != 0;
    return static_cast<uint64_t>(total);
  }
  // Another stream of the caller waits for the runs enqueued so far (a consumer of the export row: a collective, a copy); right behind
  // runAsync() it costs the solver's stream nothing -- the launch carries the event.
  void streamWaitRun(void* other_hip_stream) { check(flame_nltgv2_stream_wait_run(ctx_, other_hip_stream), "stream_wait_run"); }
  // How many of the last two runAsync() runs the device has not finished yet (0..2; no wait, no check of their results).
  int runsInFlight() {
    int32_t n = 0;
    check(flame_nltgv2_runs_in_flight(ctx_, &n), "runs_in_flight");
    return static_cast<int>(n);
  }
  // Asynchronous runs that sync() had to take back and redo (an expired neighbour wait or a torn record: the state is
  // right again when sync() returns, but anything that consumed the export row BEFORE sync() -- a gather enqueued right
  // behind runAsync() -- read the previous frame's values).  A caller compares this before and after its sync().
  int replays() {
    flame_nltgv2_info info;
    check(flame_nltgv2_get_info(ctx_, &info), "get_info");
    return info.timeouts_recovered + info.torn_records_detected;
  }
  flame_nltgv2_ctx* handle() { return ctx_; }

 private:
  void check(int rc, const char* what) {
    if (rc != 0) throw flame_hip::Error(rc, what);
  }
  flame_nltgv2_ctx* ctx_;
  flame_hip::FlatArrays flat_;
  size_t uploaded_v_ = 0, uploaded_e_ = 0;
  uint64_t generation_ = 0, identity_ = 0;
  bool uploaded_ = false;
};

// ---- the reference's free functions, same signatures (h:134-168) -----------------------------------
// Stateless convenience forms: upload + compute + download on every call.  Exact, but they pay the
// transfer each time; a pipeline should own a DeviceGraph instead.
template <class Graph>
inline void step(const Params& params, Graph* graph) {
  DeviceGraph d;
  d.upload(*graph);
  d.step(params);
  d.download(graph);
}
template <class Graph>
inline float smoothnessCost(const Params& params, const Graph& graph) {
  DeviceGraph d;
  d.upload(graph);
  return d.smoothnessCost(params);
}
template <class Graph>
inline float dataCost(const Params& params, const Graph& graph) {
  DeviceGraph d;
  d.upload(graph);
  return d.dataCost(params);
}
template <class Graph>
inline float cost(const Params& params, const Graph& graph) {
  return smoothnessCost(params, graph) + dataCost(params, graph);
}

namespace internal {
template <class Graph>
inline void dualStep(const Params& params, Graph* graph) {
  DeviceGraph d;
  d.upload(*graph);
  d.dualStep(params);
  d.download(graph);
}
template <class Graph>
inline void primalStep(const Params& params, Graph* graph) {
  DeviceGraph d;
  d.upload(*graph);
  d.primalStep(params);
  d.download(graph);
}
template <class Graph>
inline void extraGradientStep(const Params& params, Graph* graph) {
  DeviceGraph d;
  d.upload(*graph);
  d.extraGradientStep(params);
  d.download(graph);
}
}  // namespace internal

}  // namespace hip
}  // namespace nltgv2_l1_graph_regularizer
}  // namespace optimizers
}  // namespace flame

#endif  // FLAME_HIP_NLTGV2_L1_GRAPH_REGULARIZER_HPP_
