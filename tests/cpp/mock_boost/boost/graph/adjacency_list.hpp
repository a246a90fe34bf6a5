// tests/cpp/mock_boost/boost/graph/adjacency_list.hpp -- TEST-ONLY mock of the small part of Boost.Graph
// that include/flame_hip/bgl_adaptor.hpp (OUR code) touches.  Boost is not installed in the build image;
// this mock exists solely so that the adaptor -- the piece a FLaME maintainer would actually include --
// can be compiled and exercised end to end (tests/cpp/bgl_adaptor_test.cc).  It is NOT used to build any
// reference source.  It mimics what the real container guarantees and the adaptor relies on:
//   * vertices are heap nodes, descriptors are opaque pointers, boost::vertices() walks a hash set
//     (unspecified order);
//   * edges live in a std::list: boost::edges() = insertion order; boost::source/target return the
//     vertices in add_edge's argument order.
#pragma once
#include <cstddef>
#include <list>
#include <tuple>
#include <unordered_set>
#include <utility>

namespace boost {
struct hash_setS {};
struct vecS {};
struct listS {};
struct undirectedS {};
struct no_property {};

template <class OutEdgeS = vecS, class VertexS = vecS, class DirS = undirectedS, class VP = no_property,
          class EP = no_property, class GP = no_property, class EdgeS = listS>
class adjacency_list {
 public:
  typedef void* vertex_descriptor;
  struct edge_node {
    vertex_descriptor s, t;
    EP prop;
  };
  typedef edge_node* edge_descriptor;
  struct vertex_node {
    VP prop;
  };
  typedef typename std::unordered_set<void*>::const_iterator vertex_iterator;
  struct edge_iterator {
    typename std::list<edge_node>::iterator it;
    edge_descriptor operator*() const { return &*it; }
    edge_iterator& operator++() { ++it; return *this; }
    bool operator!=(const edge_iterator& o) const { return it != o.it; }
    bool operator==(const edge_iterator& o) const { return it == o.it; }
  };
  ~adjacency_list() { for (void* v : verts_) delete static_cast<vertex_node*>(v); }
  adjacency_list() = default;
  adjacency_list(const adjacency_list&) = delete;
  VP& operator[](vertex_descriptor v) { return static_cast<vertex_node*>(v)->prop; }
  const VP& operator[](vertex_descriptor v) const { return static_cast<vertex_node*>(v)->prop; }
  EP& operator[](edge_descriptor e) { return e->prop; }
  const EP& operator[](edge_descriptor e) const { return e->prop; }
  std::unordered_set<void*> verts_;
  mutable std::list<edge_node> edges_;
};

template <class G>
struct graph_traits {
  typedef typename G::vertex_descriptor vertex_descriptor;
  typedef typename G::edge_descriptor edge_descriptor;
};

using std::tie;

template <class A, class B, class C, class D, class E, class F, class H>
std::pair<typename adjacency_list<A, B, C, D, E, F, H>::vertex_iterator, typename adjacency_list<A, B, C, D, E, F, H>::vertex_iterator>
vertices(const adjacency_list<A, B, C, D, E, F, H>& g) { return {g.verts_.begin(), g.verts_.end()}; }

template <class A, class B, class C, class D, class E, class F, class H>
std::pair<typename adjacency_list<A, B, C, D, E, F, H>::edge_iterator, typename adjacency_list<A, B, C, D, E, F, H>::edge_iterator>
edges(const adjacency_list<A, B, C, D, E, F, H>& g) {
  typedef typename adjacency_list<A, B, C, D, E, F, H>::edge_iterator It;
  return {It{g.edges_.begin()}, It{g.edges_.end()}};
}

template <class A, class B, class C, class D, class E, class F, class H>
size_t num_vertices(const adjacency_list<A, B, C, D, E, F, H>& g) { return g.verts_.size(); }
template <class A, class B, class C, class D, class E, class F, class H>
size_t num_edges(const adjacency_list<A, B, C, D, E, F, H>& g) { return g.edges_.size(); }

template <class G>
typename G::vertex_descriptor source(typename G::edge_descriptor e, const G&) { return e->s; }
template <class G>
typename G::vertex_descriptor target(typename G::edge_descriptor e, const G&) { return e->t; }

template <class VPin, class A, class B, class C, class D, class E, class F, class H>
typename adjacency_list<A, B, C, D, E, F, H>::vertex_descriptor add_vertex(const VPin& p, adjacency_list<A, B, C, D, E, F, H>& g) {
  typedef typename adjacency_list<A, B, C, D, E, F, H>::vertex_node N;
  N* n = new N{p};
  g.verts_.insert(n);
  return n;
}
template <class EPin, class A, class B, class C, class D, class E, class F, class H>
std::pair<typename adjacency_list<A, B, C, D, E, F, H>::edge_descriptor, bool> add_edge(
    void* u, void* v, const EPin& p, adjacency_list<A, B, C, D, E, F, H>& g) {
  g.edges_.push_back({u, v, p});
  return {&g.edges_.back(), true};
}
}  // namespace boost
