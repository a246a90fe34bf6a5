// nltgv2_regions.hpp -- layout (R) of the region-per-workgroup persistent run (k_persistent_rg, nltgv2_persistent_rg.hip).
//
// The patch-per-wave forms hand EVERY neighbour record over through L2 once per step: a step costs one wave's dependent
// instructions plus one L2 hand-off (~1100-1500 cycles), whatever the graph (docs/LAB_NOTES.md, "what the period is made of").
// Layout (R) pays that hand-off once per k steps instead:
//
//   * the vertices are cut into compact REGIONS by recursive coordinate bisection of `pos` (balanced to one vertex, one
//     region per workgroup, one workgroup per CU at the headline size);
//   * a region also carries a GHOST RING of depth k around its owned vertices (breadth-first over the graph).  Inside a
//     block of k steps nothing leaves the workgroup: neighbours meet in LDS.  Ghost vertices run the SAME arithmetic in the
//     same order as their owners, so whatever a region computes about a vertex is bit-identical to its owner's value for as
//     long as the inputs were exact: after sub-step s of a block the vertices of depth <= k - s are exact, and after the
//     k-th only the owned ones are -- then the block ends and the ring is refreshed from the owners (16-byte tagged
//     records through L2, as in the other persistent forms): full state {x, w1, w2}, {x_bar, w1_bar, w2_bar} for depth
//     1 .. k-1, the bar record alone for depth k, and {q1, q2, q3} of the edges whose farther endpoint lies at depth >= 2
//     (an edge both of whose endpoints are at depth <= 1 stays exact by itself);
//   * a lane per EDGE for the dual update (one copy of q per edge and region: a quarter of the vector instructions of the
//     lane-per-half-edge forms, which is what makes the redundant ring affordable on four SIMDs) and a lane per VERTEX for
//     the ordered accumulation (cc:120-142: ascending edge id), proxL1 and the extragradient; edge lanes leave their
//     contributions in per-(vertex, rank) LDS slots, rank = the edge's position in the vertex's ascending edge list.
//
// Reference arithmetic: /root/reference/src/flame/optimizers/nltgv2_l1_graph_regularizer.cc:89-174 (untouched: the kernel
// evaluates the same expressions in the same order; tests/cpp/rg_layout_test.cc replays this layout and its block schedule on
// the CPU against the checker before any GPU time is spent).
//
// Pure C++17, no HIP.  Needs the host image of PackedLayout (A) + (B)'s header: row_ptr, half, half_nbr, iperm, slice_row, rid_of.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

#include "nltgv2_pack.hpp"

namespace flame_hip {

constexpr int kRgMaxThreads = 1024;  // lanes of one region's workgroup
constexpr int kRgMaxDegree = 32;     // LDS contribution slots per vertex (a graph with a larger vertex runs in another form)
constexpr int kRgMaxDepth = 6;       // steps per block (ghost ring depth)
constexpr int kRgProbeWords = 16;    // per region and block, FLAME_NLTGV2_OPT_PROBE (nltgv2_persistent_rg.hip)
constexpr int kRgInfoWords = 8;      // per region: thread offset, threads, computed vertices, local vertices, edges, fetch entries per thread,
                                     //             fetch table offset, largest degree of a computed vertex
// v_meta
constexpr uint32_t kRgDepthMask = 15u, kRgOwned = 1u << 4, kRgExportA = 1u << 5, kRgExportB = 1u << 6;
constexpr int kRgDegShift = 8;
// e_meta: rank at the source | rank at the target << 8 | flags | level << 20
constexpr uint32_t kRgSrcComputed = 1u << 16, kRgDstComputed = 1u << 17, kRgHome = 1u << 18, kRgExportQ = 1u << 19;
constexpr int kRgLevelShift = 20;
constexpr int kRgMaxFetch = 2;       // fetch duties per lane (a region with more records to fetch gets more lanes)

struct RegionLayout {
  bool ok = false;
  int32_t n_regions = 0, depth = 0;
  int32_t block_threads = 0;   // launch block size = most threads of any region
  int32_t nb_cap = 0;          // most local vertices (LDS bar entries)
  int32_t nc_cap = 0;          // most computed vertices, rounded up to 64 (stride of the contribution slots)
  int32_t deg_cap = 0;         // most contribution slots per vertex
  int32_t n_packed = 0;        // Vp: records {x,w} at [0, Vp), bar records at [Vp, 2 Vp), edge records at [2 Vp, 2 Vp + E)
  int32_t n_rec = 0;           // records per copy
  int64_t total_threads = 0, total_fetch = 0;
  std::vector<int32_t> info;        // [n_regions * kRgInfoWords]
  std::vector<int32_t> v_pv;        // [T] packed vertex of the lane's vertex role, -1 none
  std::vector<uint32_t> v_meta;     // [T]
  std::vector<int32_t> e_slot_src;  // [T] SELL slot of the edge's source half-edge, -1: no edge role
  std::vector<int32_t> e_slot_dst;  // [T] ... of its target half-edge
  std::vector<int32_t> e_id;        // [T] edge id (record 2 Vp + id)
  std::vector<uint32_t> e_li;       // [T] local index of the source | of the target << 16
  std::vector<uint32_t> e_meta;     // [T]
  // Fetch duties: entry j of lane t is at [f_off + j * threads + t]; the record it fetches lands in POLL SLOT j * threads + t of the
  // region's LDS, where its consumers read it: a ring vertex's {x, w1, w2} (v_fa), an outer edge's {q1, q2, q3} (e_fq), and -- in the
  // first step of a block only -- the bar record of a ring endpoint (e_fbs / e_fbd; later steps read what the region computed itself).
  std::vector<int32_t> f_src;       // [total_fetch] record index, -1 none
  std::vector<int32_t> f_prod;      // [total_fetch] the region that publishes the record
  std::vector<int32_t> v_fa;        // [T] poll slot of this vertex lane's {x, w1, w2} record, -1: owned (or not computed)
  std::vector<int32_t> e_fq;        // [T] poll slot of this edge lane's q record, -1: level <= 1
  std::vector<int32_t> e_fbs;       // [T] poll slot of the bar record of the edge's source, -1: the source is owned
  std::vector<int32_t> e_fbd;       // [T] ... of its target
  int32_t f_cap = 0;                // most fetch duties of a lane (<= kRgMaxFetch)
  std::vector<int32_t> region_of;   // [V] owner region of every vertex (host only)
  // statistics (tools / DESIGN.md)
  int64_t sum_owned = 0, sum_computed = 0, sum_local = 0, sum_edges = 0, sum_fetch = 0;
};

namespace rg_detail {

struct Rcb {
  const float* pos;
  std::vector<int32_t>* idx;
  std::vector<int32_t>* region_of;
  std::vector<int32_t>* first;  // [n_regions + 1] ranges of idx
  const float* weight = nullptr;  // per vertex: a cut leaves `parts / 2` of the parts' share of the WEIGHT on its lower side (nullptr: of the count)
  int32_t next = 0;
  void cut(int32_t lo, int32_t hi, int32_t parts) {
    if (parts <= 1) {
      (*first)[static_cast<size_t>(next)] = lo;
      for (int32_t i = lo; i < hi; ++i) (*region_of)[static_cast<size_t>((*idx)[static_cast<size_t>(i)])] = next;
      ++next;
      return;
    }
    float mn[2] = {0, 0}, mx[2] = {0, 0};
    for (int32_t i = lo; i < hi; ++i) {
      const int32_t v = (*idx)[static_cast<size_t>(i)];
      for (int a = 0; a < 2; ++a) {
        const float c = pos[2 * v + a];
        if (i == lo || c < mn[a]) mn[a] = c;
        if (i == lo || c > mx[a]) mx[a] = c;
      }
    }
    const int axis = (mx[1] - mn[1] > mx[0] - mn[0]) ? 1 : 0;
    const int32_t pl = parts / 2;
    int32_t mid = lo + static_cast<int32_t>((static_cast<int64_t>(hi - lo) * pl) / parts);
    int32_t* b = idx->data();
    const float* p = pos;
    auto less = [p, axis](int32_t u, int32_t v) {
      const float cu = p[2 * u + axis], cv = p[2 * v + axis];
      return cu < cv || (cu == cv && u < v);
    };
    if (weight == nullptr) {
      std::nth_element(b + lo, b + mid, b + hi, less);
    } else {
      std::sort(b + lo, b + hi, less);
      double total = 0.0, acc = 0.0;
      for (int32_t i = lo; i < hi; ++i) total += weight[b[i]];
      const double want = total * pl / parts;
      mid = lo;
      while (mid < hi && acc + 0.5 * weight[b[mid]] < want) acc += weight[b[mid++]];
      // (every part keeps at least one vertex)
      mid = std::max(lo + pl, std::min(mid, hi - (parts - pl)));
    }
    cut(lo, mid, pl);
    cut(mid, hi, parts - pl);
  }
};

}  // namespace rg_detail

// How many regions a graph of V vertices is cut into for `target` workgroups (one per CU): never regions of fewer than 16 vertices.
inline int32_t rg_region_count(int32_t V, int32_t target) {
  return std::max<int32_t>(1, std::min<int32_t>(target, V / 16));
}

// Returns FLAME_NLTGV2_OK; R->ok says whether the form can run the graph (degree, size of a region's workgroup).
inline int build_regions(const flame_nltgv2_graph* g, const PackedLayout& L, int32_t n_regions_target, int32_t depth, RegionLayout* R,
                         int balance_rounds = 3) {
  *R = RegionLayout{};
  const int32_t V = L.V, E = L.E;
  if (!g || V <= 0 || depth < 1 || depth > kRgMaxDepth || n_regions_target < 1) return FLAME_NLTGV2_OK;
  if (L.max_degree > kRgMaxDegree) return FLAME_NLTGV2_OK;
  if (L.row_ptr.size() != static_cast<size_t>(V) + 1 || L.half_nbr.size() != static_cast<size_t>(2) * E || L.iperm.size() != static_cast<size_t>(V))
    return FLAME_NLTGV2_ERR_INVALID_ARG;
  const int32_t NR = rg_region_count(V, n_regions_target);
  const int32_t k = depth;
  R->n_regions = NR, R->depth = k;
  R->n_packed = L.n_slices * kWave;
  R->n_rec = 2 * R->n_packed + E;
  // ---- regions: recursive coordinate bisection, balanced to one vertex ------------------------------------------------
  std::vector<int32_t> idx(static_cast<size_t>(V)), first(static_cast<size_t>(NR) + 1, 0);
  for (int32_t v = 0; v < V; ++v) idx[static_cast<size_t>(v)] = v;
  R->region_of.assign(static_cast<size_t>(V), 0);
  {
    rg_detail::Rcb rcb{g->pos, &idx, &R->region_of, &first};
    rcb.cut(0, V, NR);
    first[static_cast<size_t>(NR)] = V;
  }
  // ---- balance what a region COSTS, not what it owns: the lock-step network runs at the pace of its slowest workgroup, and a region
  // on the hull of the triangulation (long edges, vertices of 10-19 edges) drags a ring in that is half again as large as an inner
  // region's.  Three rounds of: the cost of every region (lanes of its ring: edges + 2 x computed vertices) -> the weight of its
  // vertices -> the bisection again, by weight.
  if (NR > 1 && balance_rounds > 0) {
    std::vector<float> wgt(static_cast<size_t>(V), 1.0f);
    std::vector<int32_t> stamp(static_cast<size_t>(V), -1), dep(static_cast<size_t>(V), 0), estamp2(static_cast<size_t>(E), -1), fr, nx;
    for (int round = 0; round < balance_rounds; ++round) {
      std::vector<double> cost(static_cast<size_t>(NR), 0.0);
      double sum = 0.0;
      for (int32_t r = 0; r < NR; ++r) {
        fr.assign(idx.begin() + first[static_cast<size_t>(r)], idx.begin() + first[static_cast<size_t>(r) + 1]);
        for (int32_t v : fr) stamp[static_cast<size_t>(v)] = r + NR * round, dep[static_cast<size_t>(v)] = 0;
        int64_t n_vc = 0, n_e = 0;
        for (int32_t d = 0; d < k; ++d) {  // the computed vertices: depth 0 .. k - 1
          n_vc += static_cast<int64_t>(fr.size());
          nx.clear();
          for (int32_t v : fr)
            for (int32_t h = L.row_ptr[v]; h < L.row_ptr[v + 1]; ++h) {
              const int32_t e = static_cast<int32_t>(L.half[static_cast<size_t>(h)] & ~kRoleBit);
              if (estamp2[static_cast<size_t>(e)] != r + NR * round) estamp2[static_cast<size_t>(e)] = r + NR * round, ++n_e;
              const int32_t u = L.half_nbr[static_cast<size_t>(h)];
              if (stamp[static_cast<size_t>(u)] != r + NR * round) stamp[static_cast<size_t>(u)] = r + NR * round, nx.push_back(u);
            }
          fr.swap(nx);
        }
        cost[static_cast<size_t>(r)] = static_cast<double>(n_e) + 2.0 * static_cast<double>(n_vc);
        sum += cost[static_cast<size_t>(r)];
      }
      const double mean = sum / NR;
      for (int32_t r = 0; r < NR; ++r) {
        const float f = static_cast<float>(std::min(2.0, std::max(0.5, cost[static_cast<size_t>(r)] / mean)));
        for (int32_t i = first[static_cast<size_t>(r)]; i < first[static_cast<size_t>(r) + 1]; ++i) wgt[static_cast<size_t>(idx[static_cast<size_t>(i)])] *= f;
      }
      for (int32_t v = 0; v < V; ++v) idx[static_cast<size_t>(v)] = v;
      rg_detail::Rcb rcb{g->pos, &idx, &R->region_of, &first, wgt.data()};
      rcb.cut(0, V, NR);
      first[static_cast<size_t>(NR)] = V;
    }
  }
  // ---- rank of every edge in its endpoints' ascending edge lists (the order of the reference's scatter, cc:120-142) ---
  std::vector<uint8_t> rank_src(static_cast<size_t>(E)), rank_dst(static_cast<size_t>(E));
  for (int32_t v = 0; v < V; ++v)
    for (int32_t h = L.row_ptr[v]; h < L.row_ptr[v + 1]; ++h) {
      const uint32_t he = L.half[static_cast<size_t>(h)];
      const int32_t e = static_cast<int32_t>(he & ~kRoleBit);
      ((he & kRoleBit) ? rank_dst : rank_src)[static_cast<size_t>(e)] = static_cast<uint8_t>(h - L.row_ptr[v]);
    }
  auto sell_slot = [&](int32_t v, int32_t rank) {
    const int32_t s = L.iperm[static_cast<size_t>(v)];
    return static_cast<int32_t>((static_cast<int64_t>(L.slice_row[static_cast<size_t>(s / kWave)]) + rank) * kWave + (s % kWave));
  };
  // ---- per region: ghost ring, lanes, fetch list -----------------------------------------------------------------------
  std::vector<int32_t> vstamp(static_cast<size_t>(V), -1), vdepth(static_cast<size_t>(V), 0), vli(static_cast<size_t>(V), 0);
  std::vector<int32_t> estamp(static_cast<size_t>(E), -1);
  std::vector<int32_t> own_thread(static_cast<size_t>(V), -1), home_thread(static_cast<size_t>(E), -1);  // lane (global) of the owner's / home copy
  std::vector<uint8_t> needA(static_cast<size_t>(V), 0), needB(static_cast<size_t>(V), 0), needQ(static_cast<size_t>(E), 0);
  std::vector<int32_t> local, frontier, next, edges;
  struct Fetch {
    int32_t prod, src, kind, who;
  };
  std::vector<Fetch> fetch;
  std::vector<int32_t> fb_of;  // poll slot of a local vertex's bar record
  R->info.assign(static_cast<size_t>(NR) * kRgInfoWords, 0);
  {  // (the lane tables grow region by region: room for a typical layout up front, so that they are not copied at every region)
    const size_t guess = static_cast<size_t>(NR) * 64 + static_cast<size_t>(8 * k) * (static_cast<size_t>(V) + E);
    for (std::vector<int32_t>* a : {&R->v_pv, &R->v_fa, &R->e_slot_src, &R->e_slot_dst, &R->e_id, &R->e_fq, &R->e_fbs, &R->e_fbd}) a->reserve(guess);
    for (std::vector<uint32_t>* a : {&R->v_meta, &R->e_li, &R->e_meta}) a->reserve(guess);
    R->f_src.reserve(2 * guess), R->f_prod.reserve(2 * guess);
  }
  int64_t T = 0, FT = 0;
  bool fits = true;
  for (int32_t r = 0; r < NR; ++r) {
    local.clear(), edges.clear(), fetch.clear();
    frontier.assign(idx.begin() + first[static_cast<size_t>(r)], idx.begin() + first[static_cast<size_t>(r) + 1]);
    const int32_t n_owned = static_cast<int32_t>(frontier.size());
    for (int32_t v : frontier) vstamp[static_cast<size_t>(v)] = r, vdepth[static_cast<size_t>(v)] = 0;
    int32_t n_vc = 0;
    for (int32_t d = 0; d <= k; ++d) {
      std::sort(frontier.begin(), frontier.end(), [&](int32_t a, int32_t b) { return L.rid_of[static_cast<size_t>(a)] < L.rid_of[static_cast<size_t>(b)]; });
      for (int32_t v : frontier) vli[static_cast<size_t>(v)] = static_cast<int32_t>(local.size()), local.push_back(v);
      if (d == k - 1) n_vc = static_cast<int32_t>(local.size());
      if (d == k) break;
      next.clear();
      for (int32_t v : frontier)
        for (int32_t h = L.row_ptr[v]; h < L.row_ptr[v + 1]; ++h) {
          const int32_t u = L.half_nbr[static_cast<size_t>(h)];
          if (vstamp[static_cast<size_t>(u)] != r) vstamp[static_cast<size_t>(u)] = r, vdepth[static_cast<size_t>(u)] = d + 1, next.push_back(u);
        }
      frontier.swap(next);
    }
    const int32_t n_vall = static_cast<int32_t>(local.size());
    // edges: every edge of a computed vertex (depth <= k - 1), once; level = the larger depth of its endpoints
    int32_t maxdeg = 0;
    for (int32_t i = 0; i < n_vc; ++i) {
      const int32_t v = local[static_cast<size_t>(i)];
      maxdeg = std::max(maxdeg, L.row_ptr[v + 1] - L.row_ptr[v]);
      for (int32_t h = L.row_ptr[v]; h < L.row_ptr[v + 1]; ++h) {
        const int32_t e = static_cast<int32_t>(L.half[static_cast<size_t>(h)] & ~kRoleBit);
        if (estamp[static_cast<size_t>(e)] != r) estamp[static_cast<size_t>(e)] = r, edges.push_back(e);
      }
    }
    auto level_of = [&](int32_t e) { return std::max(vdepth[static_cast<size_t>(g->src[e])], vdepth[static_cast<size_t>(g->dst[e])]); };
    std::sort(edges.begin(), edges.end(), [&](int32_t a, int32_t b) {
      const int32_t la = level_of(a), lb = level_of(b);
      return la < lb || (la == lb && a < b);
    });
    const int32_t n_e = static_cast<int32_t>(edges.size());
    // fetch list: {x, w} and bar records of the ring's computed vertices, bar records of the outermost ring, q of the outer edges
    // -- kind 0: bar of local vertex `who`, 1: {x, w} of local vertex `who`, 2: q of edge lane `who`
    for (int32_t i = n_owned; i < n_vall; ++i) {
      const int32_t v = local[static_cast<size_t>(i)], pv = L.iperm[static_cast<size_t>(v)], d = vdepth[static_cast<size_t>(v)];
      const int32_t prod = R->region_of[static_cast<size_t>(v)];
      if (d < k) fetch.push_back(Fetch{prod, pv, 1, i}), needA[static_cast<size_t>(v)] = 1;
      fetch.push_back(Fetch{prod, R->n_packed + pv, 0, i}), needB[static_cast<size_t>(v)] = 1;
    }
    for (int32_t t = 0; t < n_e; ++t) {
      const int32_t e = edges[static_cast<size_t>(t)];
      if (level_of(e) >= 2) fetch.push_back(Fetch{R->region_of[static_cast<size_t>(g->src[e])], 2 * R->n_packed + e, 2, t}), needQ[static_cast<size_t>(e)] = 1;
    }
    std::sort(fetch.begin(), fetch.end(), [](const Fetch& a, const Fetch& b) { return a.prod < b.prod || (a.prod == b.prod && a.src < b.src); });
    const int32_t n_f = static_cast<int32_t>(fetch.size());
    // lanes: one per local vertex, one per edge, and enough of them for at most kRgMaxFetch fetch duties each
    const int32_t n_lanes = std::max(std::max(std::max(n_vall, n_e), (n_f + kRgMaxFetch - 1) / kRgMaxFetch), 1);
    const int32_t n_threads = (n_lanes + kWave - 1) / kWave * kWave;
    if (n_threads > kRgMaxThreads || n_vall > 0xffff) fits = false;
    const int32_t F = (n_f + n_threads - 1) / n_threads;
    int32_t* inf = &R->info[static_cast<size_t>(r) * kRgInfoWords];
    inf[0] = static_cast<int32_t>(T), inf[1] = n_threads, inf[2] = n_vc, inf[3] = n_vall, inf[4] = n_e, inf[5] = F, inf[6] = static_cast<int32_t>(FT), inf[7] = maxdeg;
    // lanes
    const size_t T1 = static_cast<size_t>(T + n_threads);
    R->v_pv.resize(T1, -1), R->v_meta.resize(T1, 0u), R->v_fa.resize(T1, -1);
    R->e_slot_src.resize(T1, -1), R->e_slot_dst.resize(T1, -1), R->e_id.resize(T1, -1), R->e_li.resize(T1, 0u), R->e_meta.resize(T1, 0u);
    R->e_fq.resize(T1, -1), R->e_fbs.resize(T1, -1), R->e_fbd.resize(T1, -1);
    R->f_src.resize(static_cast<size_t>(FT + static_cast<int64_t>(F) * n_threads), -1);
    R->f_prod.resize(R->f_src.size(), 0);
    fb_of.assign(static_cast<size_t>(n_vall), -1);
    for (int32_t i = 0; i < n_f; ++i) {  // entry i -> lane i % threads, round i / threads: a round's loads are consecutive records
      const Fetch& f = fetch[static_cast<size_t>(i)];
      const int32_t slot = (i / n_threads) * n_threads + i % n_threads;  // (== i: the poll slot is the entry's index)
      R->f_src[static_cast<size_t>(FT + slot)] = f.src, R->f_prod[static_cast<size_t>(FT + slot)] = f.prod;
      if (f.kind == 0) fb_of[static_cast<size_t>(f.who)] = slot;
      else if (f.kind == 1) R->v_fa[static_cast<size_t>(T + f.who)] = slot;
      else R->e_fq[static_cast<size_t>(T + f.who)] = slot;
    }
    for (int32_t i = 0; i < n_vall; ++i) {
      const int32_t v = local[static_cast<size_t>(i)], d = vdepth[static_cast<size_t>(v)];
      R->v_pv[static_cast<size_t>(T + i)] = L.iperm[static_cast<size_t>(v)];
      R->v_meta[static_cast<size_t>(T + i)] = static_cast<uint32_t>(d) | (d == 0 ? kRgOwned : 0u) | (static_cast<uint32_t>(L.row_ptr[v + 1] - L.row_ptr[v]) << kRgDegShift);
      if (d == 0) own_thread[static_cast<size_t>(v)] = static_cast<int32_t>(T + i);
    }
    for (int32_t t = 0; t < n_e; ++t) {
      const int32_t e = edges[static_cast<size_t>(t)], a = g->src[e], b = g->dst[e];
      const int32_t da = vdepth[static_cast<size_t>(a)], db = vdepth[static_cast<size_t>(b)];
      // (both endpoints are local: a computed vertex's neighbours are at most one ring further out)
      uint32_t m = static_cast<uint32_t>(rank_src[static_cast<size_t>(e)]) | (static_cast<uint32_t>(rank_dst[static_cast<size_t>(e)]) << 8) |
                   (static_cast<uint32_t>(std::max(da, db)) << kRgLevelShift);
      if (da < k) m |= kRgSrcComputed;
      if (db < k) m |= kRgDstComputed;
      if (da == 0) m |= kRgHome, home_thread[static_cast<size_t>(e)] = static_cast<int32_t>(T + t);
      R->e_slot_src[static_cast<size_t>(T + t)] = sell_slot(a, rank_src[static_cast<size_t>(e)]);
      R->e_slot_dst[static_cast<size_t>(T + t)] = sell_slot(b, rank_dst[static_cast<size_t>(e)]);
      R->e_id[static_cast<size_t>(T + t)] = e;
      R->e_li[static_cast<size_t>(T + t)] = static_cast<uint32_t>(vli[static_cast<size_t>(a)]) | (static_cast<uint32_t>(vli[static_cast<size_t>(b)]) << 16);
      R->e_meta[static_cast<size_t>(T + t)] = m;
      R->e_fbs[static_cast<size_t>(T + t)] = fb_of[static_cast<size_t>(vli[static_cast<size_t>(a)])];
      R->e_fbd[static_cast<size_t>(T + t)] = fb_of[static_cast<size_t>(vli[static_cast<size_t>(b)])];
    }
    R->f_cap = std::max(R->f_cap, F);
    R->block_threads = std::max(R->block_threads, n_threads);
    R->nb_cap = std::max(R->nb_cap, n_vall);
    R->nc_cap = std::max(R->nc_cap, (n_vc + kWave - 1) / kWave * kWave);
    R->deg_cap = std::max(R->deg_cap, maxdeg);
    R->sum_owned += n_owned, R->sum_computed += n_vc, R->sum_local += n_vall, R->sum_edges += n_e, R->sum_fetch += n_f;
    T += n_threads, FT += static_cast<int64_t>(F) * n_threads;
  }
  R->total_threads = T, R->total_fetch = FT;
  // ---- who publishes what: a record somebody fetches is published by the lane that owns it -------------------------------
  for (int32_t v = 0; v < V; ++v) {
    const int32_t t = own_thread[static_cast<size_t>(v)];
    if (t < 0) return FLAME_NLTGV2_ERR_INVALID_ARG;  // (every vertex is owned by exactly one region)
    if (needA[static_cast<size_t>(v)]) R->v_meta[static_cast<size_t>(t)] |= kRgExportA;
    if (needB[static_cast<size_t>(v)]) R->v_meta[static_cast<size_t>(t)] |= kRgExportB;
  }
  for (int32_t e = 0; e < E; ++e) {
    const int32_t t = home_thread[static_cast<size_t>(e)];
    if (t < 0) return FLAME_NLTGV2_ERR_INVALID_ARG;  // (every edge has its home in the region that owns its source)
    if (needQ[static_cast<size_t>(e)]) R->e_meta[static_cast<size_t>(t)] |= kRgExportQ;
  }
  R->ok = fits && T < (int64_t(1) << 30);
  return FLAME_NLTGV2_OK;
}

// Contribution slots per vertex: the largest degree rounded up to 8 (the V phase reads eight slots per round of loads), + 1 (an odd
// stride keeps the lanes of a wave on different LDS banks).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int rg_slot_stride(int deg_cap) { return ((deg_cap > 0 ? deg_cap : 1) + 7) / 8 * 8 + 1; }

// LDS of one workgroup, in bytes: [bar: nb | poll slots: f_cap x threads | spare: threads | contributions {cx, a1, a2, b1}: nc x SD]
// as float4, then [b2: nc x SD | spare: threads] as float.
inline size_t rg_lds_bytes(const RegionLayout& R) {
  const size_t sd = static_cast<size_t>(rg_slot_stride(R.deg_cap));
  const size_t f4 = static_cast<size_t>(R.nb_cap) + static_cast<size_t>(R.f_cap > 0 ? R.f_cap : 1) * R.block_threads + R.block_threads + sd * R.nc_cap;
  return 16 * f4 + 4 * (sd * R.nc_cap + R.block_threads) + 16;
}

}  // namespace flame_hip
