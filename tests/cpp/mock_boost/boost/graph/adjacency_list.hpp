// tests/cpp/mock_boost/boost/graph/adjacency_list.hpp -- TEST-ONLY model of the part of Boost.Graph (1.58, the version the reference
// builds against: README.md:32, CMakeLists.txt:31) that include/flame_hip/bgl_adaptor.hpp -- OUR code -- and a FLaME-style caller
// touch.  Boost is not installed in the build image; this file exists solely so that the adaptor, the piece a FLaME maintainer would
// actually include, is compiled against the SIGNATURES Boost documents and exercised end to end (tests/cpp/bgl_adaptor_test.cc).  It
// is NOT used to build any reference source, and it is not Boost: a look-alike written here from the documented interface of
//   boost::adjacency_list<hash_setS, hash_setS, undirectedS, VertexProperty, EdgeProperty>     (nltgv2_l1_graph_regularizer.h:107-112)
// What it models, because the adaptor or the pipeline's contract relies on it:
//   * VertexList = hash_setS: vertices are individually allocated nodes, `vertex_descriptor` is `void*`, vertices(g) walks a hash
//     set -- an unspecified order that changes when vertices are added or removed, but is stable between two walks of an unchanged
//     graph (what pack() / unpack() by position need);
//   * `edge_descriptor` is a VALUE type (boost::detail::edge_desc_impl<undirected_tag, void*>): source, target and a pointer to the
//     edge's property; edge iterators return it by value; g[e] is the bundled EdgeProperty;
//   * the edge list is a std::list: edges(g) = insertion order, remove_edge keeps the order of the rest, add_edge appends;
//     source(e, g) / target(e, g) = add_edge's argument order;
//   * OutEdgeList = hash_setS: no parallel edges -- add_edge(u, v) on an existing pair returns {existing, false}; edge(u, v, g)
//     finds an edge whichever way it was added (undirected);
//   * remove_vertex(v, g) after clear_vertex(v, g), as Flame::syncGraph does (flame.cc:2021-2023); descriptors and iterators of
//     other vertices stay valid (node-based containers), positions in vertices(g) do not;
//   * graph_traits<G> with vertex_descriptor, edge_descriptor, vertex_iterator, edge_iterator, adjacency_iterator and the size types;
//     num_vertices / num_edges; adjacent_vertices (flame.cc:2138); boost::tie on the iterator pairs.
#pragma once
#include <cstddef>
#include <iterator>
#include <list>
#include <tuple>
#include <unordered_set>
#include <utility>
#include <vector>

namespace boost {
struct hash_setS {};
struct vecS {};
struct listS {};
struct undirectedS {};
struct undirected_tag {};
struct no_property {};

namespace detail {
template <class Directed, class Vertex>
struct edge_desc_impl {  // == boost/graph/detail/edge.hpp
  edge_desc_impl() : m_source(), m_target(), m_eproperty(nullptr) {}
  edge_desc_impl(Vertex s, Vertex t, const void* p) : m_source(s), m_target(t), m_eproperty(const_cast<void*>(p)) {}
  Vertex m_source, m_target;
  void* m_eproperty;
  void* get_property() const { return m_eproperty; }
  bool operator==(const edge_desc_impl& o) const { return m_eproperty == o.m_eproperty; }
  bool operator!=(const edge_desc_impl& o) const { return m_eproperty != o.m_eproperty; }
};
}  // namespace detail

template <class OutEdgeS = vecS, class VertexS = vecS, class DirS = undirectedS, class VP = no_property,
          class EP = no_property, class GP = no_property, class EdgeS = listS>
class adjacency_list {
 public:
  typedef void* vertex_descriptor;
  typedef detail::edge_desc_impl<undirected_tag, void*> edge_descriptor;
  typedef std::size_t vertices_size_type;
  typedef std::size_t edges_size_type;
  typedef std::size_t degree_size_type;
  typedef VP vertex_bundled;
  typedef EP edge_bundled;
  struct stored_edge {
    vertex_descriptor s, t;
    EP prop;
  };
  struct stored_vertex {
    VP prop;
    std::vector<typename std::list<stored_edge>::iterator> out;  // (incident edges; Boost keeps a hash set of them)
  };
  typedef typename std::unordered_set<void*>::const_iterator vertex_iterator;
  struct edge_iterator {
    typedef std::forward_iterator_tag iterator_category;
    typedef edge_descriptor value_type;
    typedef std::ptrdiff_t difference_type;
    typedef const edge_descriptor* pointer;
    typedef edge_descriptor reference;
    typename std::list<stored_edge>::iterator it;
    edge_descriptor operator*() const { return edge_descriptor(it->s, it->t, &it->prop); }
    edge_iterator& operator++() { ++it; return *this; }
    edge_iterator operator++(int) { edge_iterator c = *this; ++it; return c; }
    bool operator!=(const edge_iterator& o) const { return it != o.it; }
    bool operator==(const edge_iterator& o) const { return it == o.it; }
  };
  struct adjacency_iterator {
    const stored_vertex* v;
    void* self;
    std::size_t i;
    vertex_descriptor operator*() const { return v->out[i]->s == self ? v->out[i]->t : v->out[i]->s; }
    adjacency_iterator& operator++() { ++i; return *this; }
    bool operator!=(const adjacency_iterator& o) const { return i != o.i; }
    bool operator==(const adjacency_iterator& o) const { return i == o.i; }
  };
  ~adjacency_list() { for (void* v : verts_) delete static_cast<stored_vertex*>(v); }
  adjacency_list() = default;
  adjacency_list(const adjacency_list&) = delete;
  adjacency_list& operator=(const adjacency_list&) = delete;
  VP& operator[](vertex_descriptor v) { return static_cast<stored_vertex*>(v)->prop; }
  const VP& operator[](vertex_descriptor v) const { return static_cast<stored_vertex*>(v)->prop; }
  EP& operator[](const edge_descriptor& e) { return *static_cast<EP*>(e.get_property()); }
  const EP& operator[](const edge_descriptor& e) const { return *static_cast<const EP*>(e.get_property()); }
  // (implementation detail of the mock, not part of the modelled interface)
  std::unordered_set<void*> verts_;
  mutable std::list<stored_edge> edges_;
};

template <class G>
struct graph_traits {
  typedef typename G::vertex_descriptor vertex_descriptor;
  typedef typename G::edge_descriptor edge_descriptor;
  typedef typename G::vertex_iterator vertex_iterator;
  typedef typename G::edge_iterator edge_iterator;
  typedef typename G::adjacency_iterator adjacency_iterator;
  typedef typename G::vertices_size_type vertices_size_type;
  typedef typename G::edges_size_type edges_size_type;
  typedef typename G::degree_size_type degree_size_type;
  static vertex_descriptor null_vertex() { return nullptr; }
};

using std::tie;  // (boost::tie assigns from a std::pair the same way)

#define FLAME_MOCK_AL template <class A, class B, class C, class D, class E, class F, class H>
#define FLAME_MOCK_G adjacency_list<A, B, C, D, E, F, H>

FLAME_MOCK_AL std::pair<typename FLAME_MOCK_G::vertex_iterator, typename FLAME_MOCK_G::vertex_iterator> vertices(const FLAME_MOCK_G& g) {
  return {g.verts_.begin(), g.verts_.end()};
}
FLAME_MOCK_AL std::pair<typename FLAME_MOCK_G::edge_iterator, typename FLAME_MOCK_G::edge_iterator> edges(const FLAME_MOCK_G& g) {
  typedef typename FLAME_MOCK_G::edge_iterator It;
  return {It{g.edges_.begin()}, It{g.edges_.end()}};
}
FLAME_MOCK_AL typename FLAME_MOCK_G::vertices_size_type num_vertices(const FLAME_MOCK_G& g) { return g.verts_.size(); }
FLAME_MOCK_AL typename FLAME_MOCK_G::edges_size_type num_edges(const FLAME_MOCK_G& g) { return g.edges_.size(); }

FLAME_MOCK_AL typename FLAME_MOCK_G::vertex_descriptor source(const typename FLAME_MOCK_G::edge_descriptor& e, const FLAME_MOCK_G&) { return e.m_source; }
FLAME_MOCK_AL typename FLAME_MOCK_G::vertex_descriptor target(const typename FLAME_MOCK_G::edge_descriptor& e, const FLAME_MOCK_G&) { return e.m_target; }

FLAME_MOCK_AL std::pair<typename FLAME_MOCK_G::adjacency_iterator, typename FLAME_MOCK_G::adjacency_iterator> adjacent_vertices(
    typename FLAME_MOCK_G::vertex_descriptor v, const FLAME_MOCK_G&) {
  typedef typename FLAME_MOCK_G::adjacency_iterator It;
  const typename FLAME_MOCK_G::stored_vertex* sv = static_cast<const typename FLAME_MOCK_G::stored_vertex*>(v);
  return {It{sv, v, 0}, It{sv, v, sv->out.size()}};
}
FLAME_MOCK_AL typename FLAME_MOCK_G::degree_size_type out_degree(typename FLAME_MOCK_G::vertex_descriptor v, const FLAME_MOCK_G&) {
  return static_cast<const typename FLAME_MOCK_G::stored_vertex*>(v)->out.size();
}

FLAME_MOCK_AL std::pair<typename FLAME_MOCK_G::edge_descriptor, bool> edge(typename FLAME_MOCK_G::vertex_descriptor u,
                                                                            typename FLAME_MOCK_G::vertex_descriptor v, const FLAME_MOCK_G&) {
  typedef typename FLAME_MOCK_G::edge_descriptor Ed;
  const typename FLAME_MOCK_G::stored_vertex* su = static_cast<const typename FLAME_MOCK_G::stored_vertex*>(u);
  for (const auto& it : su->out)
    if ((it->s == u && it->t == v) || (it->s == v && it->t == u)) return {Ed(it->s, it->t, &it->prop), true};
  return {Ed(), false};
}

template <class VPin, class A, class B, class C, class D, class E, class F, class H>
typename FLAME_MOCK_G::vertex_descriptor add_vertex(const VPin& p, FLAME_MOCK_G& g) {
  typedef typename FLAME_MOCK_G::stored_vertex N;
  N* n = new N{p, {}};
  g.verts_.insert(n);
  return n;
}
template <class EPin, class A, class B, class C, class D, class E, class F, class H>
std::pair<typename FLAME_MOCK_G::edge_descriptor, bool> add_edge(typename FLAME_MOCK_G::vertex_descriptor u,
                                                                   typename FLAME_MOCK_G::vertex_descriptor v, const EPin& p, FLAME_MOCK_G& g) {
  typedef typename FLAME_MOCK_G::edge_descriptor Ed;
  const std::pair<Ed, bool> have = edge(u, v, g);
  if (have.second) return {have.first, false};  // OutEdgeList = hash_setS: no parallel edges
  g.edges_.push_back({u, v, p});
  auto it = std::prev(g.edges_.end());
  static_cast<typename FLAME_MOCK_G::stored_vertex*>(u)->out.push_back(it);
  static_cast<typename FLAME_MOCK_G::stored_vertex*>(v)->out.push_back(it);
  return {Ed(u, v, &it->prop), true};
}

namespace detail {
template <class G, class It>
void mock_unlink(typename G::stored_vertex* sv, It it) {
  for (std::size_t i = 0; i < sv->out.size(); ++i)
    if (sv->out[i] == it) {
      sv->out.erase(sv->out.begin() + (std::ptrdiff_t)i);
      return;
    }
}
}  // namespace detail

FLAME_MOCK_AL void remove_edge(const typename FLAME_MOCK_G::edge_descriptor& e, FLAME_MOCK_G& g) {
  for (auto it = g.edges_.begin(); it != g.edges_.end(); ++it)
    if (&it->prop == e.get_property()) {
      detail::mock_unlink<FLAME_MOCK_G>(static_cast<typename FLAME_MOCK_G::stored_vertex*>(it->s), it);
      detail::mock_unlink<FLAME_MOCK_G>(static_cast<typename FLAME_MOCK_G::stored_vertex*>(it->t), it);
      g.edges_.erase(it);
      return;
    }
}
FLAME_MOCK_AL void remove_edge(typename FLAME_MOCK_G::vertex_descriptor u, typename FLAME_MOCK_G::vertex_descriptor v, FLAME_MOCK_G& g) {
  const auto have = edge(u, v, g);
  if (have.second) remove_edge(have.first, g);
}
FLAME_MOCK_AL void clear_vertex(typename FLAME_MOCK_G::vertex_descriptor v, FLAME_MOCK_G& g) {
  typename FLAME_MOCK_G::stored_vertex* sv = static_cast<typename FLAME_MOCK_G::stored_vertex*>(v);
  while (!sv->out.empty()) {
    auto it = sv->out.back();
    remove_edge(typename FLAME_MOCK_G::edge_descriptor(it->s, it->t, &it->prop), g);
  }
}
FLAME_MOCK_AL void remove_vertex(typename FLAME_MOCK_G::vertex_descriptor v, FLAME_MOCK_G& g) {  // (after clear_vertex, as Boost requires)
  g.verts_.erase(v);
  delete static_cast<typename FLAME_MOCK_G::stored_vertex*>(v);
}

#undef FLAME_MOCK_AL
#undef FLAME_MOCK_G
}  // namespace boost
