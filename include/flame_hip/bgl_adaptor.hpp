// flame_hip/bgl_adaptor.hpp -- GraphAccess for the reference's Boost.Graph container
//   using Graph = boost::adjacency_list<hash_setS, hash_setS, undirectedS, VertexData, EdgeData>
//   (/root/reference/src/flame/optimizers/nltgv2_l1_graph_regularizer.h:107-112).
//
// Include this header AFTER the reference's nltgv2_l1_graph_regularizer.h inside the FLaME tree.  It needs
// Boost.Graph and the reference's VertexData/EdgeData; Boost is absent from the build image of this repository, so here
// it is compiled and run only against tests/cpp/mock_boost/ -- a test-only model of the interface Boost 1.58 documents for
// adjacency_list<hash_setS, hash_setS, undirectedS>: void* vertex descriptors, value-type edge descriptors, graph_traits,
// node-based edge list, no parallel edges, clear_vertex / remove_vertex (not the reference, not Boost: a check of the call
// shapes and of the behaviour the adaptor relies on, stated as such in DESIGN.md) -- and against real Boost only inside
// the FLaME tree.
//
// Order and orientation are taken exactly as the reference's loops see them:
//   vertices: boost::vertices(graph) order               (cc:35-42, 145-151, 158-171)
//   edges:    boost::edges(graph) order, src = boost::source, dst = boost::target   (cc:91-96, 118-123)
// unpack() walks the same ranges, so values return to the objects they came from even though BGL's
// hash_setS containers give no stable numbering.
#ifndef FLAME_HIP_BGL_ADAPTOR_HPP_
#define FLAME_HIP_BGL_ADAPTOR_HPP_

#include <boost/graph/adjacency_list.hpp>
#include <cstdint>
#include <unordered_map>

#include "flame_hip/nltgv2_l1_graph_regularizer.hpp"

namespace flame_hip {

template <class OutEdgeS, class VertexS, class DirS, class VP, class EP, class GP, class EdgeS>
struct GraphAccess<boost::adjacency_list<OutEdgeS, VertexS, DirS, VP, EP, GP, EdgeS> > {
  typedef boost::adjacency_list<OutEdgeS, VertexS, DirS, VP, EP, GP, EdgeS> Graph;
  typedef typename boost::graph_traits<Graph>::vertex_descriptor Vertex;

  static void pack(const Graph& g, FlatArrays* f) {
    const size_t V = boost::num_vertices(g), E = boost::num_edges(g);
    f->resize(V, E);
    std::unordered_map<Vertex, int32_t> index;
    index.reserve(V);
    typename Graph::vertex_iterator vit, vend;
    boost::tie(vit, vend) = boost::vertices(g);
    for (int32_t v = 0; vit != vend; ++vit, ++v) {
      index[*vit] = v;
      const VP& d = g[*vit];
      f->pos[2 * v] = d.pos.x, f->pos[2 * v + 1] = d.pos.y;
      f->x[v] = d.x, f->w1[v] = d.w1, f->w2[v] = d.w2;
      f->x_bar[v] = d.x_bar, f->w1_bar[v] = d.w1_bar, f->w2_bar[v] = d.w2_bar;
      f->x_prev[v] = d.x_prev, f->w1_prev[v] = d.w1_prev, f->w2_prev[v] = d.w2_prev;
      f->data_term[v] = d.data_term, f->data_weight[v] = d.data_weight;
    }
    typename Graph::edge_iterator eit, eend;
    boost::tie(eit, eend) = boost::edges(g);
    for (int32_t e = 0; eit != eend; ++eit, ++e) {
      const EP& d = g[*eit];
      f->src[e] = index[boost::source(*eit, g)];
      f->dst[e] = index[boost::target(*eit, g)];
      f->alpha[e] = d.alpha, f->beta[e] = d.beta;
      f->q1[e] = d.q1, f->q2[e] = d.q2, f->q3[e] = d.q3;
    }
  }

  static void size(const Graph& g, size_t* V, size_t* E) { *V = boost::num_vertices(g), *E = boost::num_edges(g); }

  // Which vertex and edge objects stand at which position of vertices() / edges(): the descriptors of a hash_setS graph are the
  // addresses of its nodes (void*; an edge descriptor carries the address of its property), so a running hash over them in walk
  // order changes with every add_ / remove_ that moves an object to another position -- also one that leaves both counts alone.
  static uint64_t identity(const Graph& g) {
    uint64_t h = 0x243f6a8885a308d3ull;
    typename Graph::vertex_iterator vit, vend;
    boost::tie(vit, vend) = boost::vertices(g);
    for (; vit != vend; ++vit) h = mix_identity(h, (uint64_t)(uintptr_t)(const void*)*vit);
    typename Graph::edge_iterator eit, eend;
    boost::tie(eit, eend) = boost::edges(g);
    for (; eit != eend; ++eit) {
      h = mix_identity(h, (uint64_t)(uintptr_t)(const void*)boost::source(*eit, g));
      h = mix_identity(h, (uint64_t)(uintptr_t)(const void*)boost::target(*eit, g));
      h = mix_identity(h, (uint64_t)(uintptr_t)(const void*)&g[*eit]);
    }
    return h;
  }

  // By position in vertices()/edges() order: DeviceGraph::download() only calls this for a graph with the vertex and
  // edge counts (and, where the caller tracks it, the edit generation) of the one that was packed.
  static void unpack(const FlatArrays& f, Graph* g) {
    if (boost::num_vertices(*g) != f.x.size() || boost::num_edges(*g) != f.src.size()) return;
    typename Graph::vertex_iterator vit, vend;
    boost::tie(vit, vend) = boost::vertices(*g);
    for (int32_t v = 0; vit != vend; ++vit, ++v) {
      VP& d = (*g)[*vit];
      d.x = f.x[v], d.w1 = f.w1[v], d.w2 = f.w2[v];
      d.x_bar = f.x_bar[v], d.w1_bar = f.w1_bar[v], d.w2_bar = f.w2_bar[v];
      d.x_prev = f.x_prev[v], d.w1_prev = f.w1_prev[v], d.w2_prev = f.w2_prev[v];
    }
    typename Graph::edge_iterator eit, eend;
    boost::tie(eit, eend) = boost::edges(*g);
    for (int32_t e = 0; eit != eend; ++eit, ++e) {
      EP& d = (*g)[*eit];
      d.q1 = f.q1[e], d.q2 = f.q2[e], d.q3 = f.q3[e];
    }
  }
};

}  // namespace flame_hip

#endif  // FLAME_HIP_BGL_ADAPTOR_HPP_
