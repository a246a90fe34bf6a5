// nltgv2_graph_capi.hip -- graph in / out over the C-ABI: upload, per-frame warm-start sync, projection, rescale, state download,
// costs, export of x * graph_scale (see nltgv2_context.hpp).
#include "nltgv2_context.hpp"

#include <atomic>
#include <unordered_set>

#include "host_workers.hpp"

extern "C" {

int flame_nltgv2_upload_graph(flame_nltgv2_ctx* ctx, const flame_nltgv2_graph* g) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_upload_graph");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!g) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  const int32_t V = g->V, E = g->E;
  if (V < 0 || E < 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (V > 0 && (!g->pos || !g->x || !g->w1 || !g->w2 || !g->x_bar || !g->w1_bar || !g->w2_bar ||
                !g->data_term || !g->data_weight))
    return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (E > 0 && (!g->src || !g->dst || !g->alpha || !g->beta || !g->q1 || !g->q2 || !g->q3))
    return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  ctx->have_graph = false;
  const bool trace = std::getenv("FLAME_NLTGV2_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  // Make sure nothing in flight still uses buffers we may reallocate (or the staging buffer).
  HIPCHK(ctx, wait_solver_stream(ctx));
  const size_t fV = sizeof(float) * (size_t)V, fE = sizeof(float) * (size_t)E;
  struct { DevBuf* b; size_t bytes; } req[] = {
      {&ctx->x, fV}, {&ctx->w1, fV}, {&ctx->w2, fV}, {&ctx->xb, fV}, {&ctx->w1b, fV}, {&ctx->w2b, fV}, {&ctx->xp, fV},
      {&ctx->w1p, fV}, {&ctx->w2p, fV}, {&ctx->data, fV}, {&ctx->weight, fV}, {&ctx->alpha, fE}, {&ctx->beta, fE},
      {&ctx->q1, fE}, {&ctx->q2, fE}, {&ctx->q3, fE}};
  for (auto& r : req) {
    rc = ensure(ctx, *r.b, r.bytes);
    if (rc) return rc;
  }
  const StageCopy state[] = {
      {&ctx->x, g->x, fV}, {&ctx->w1, g->w1, fV}, {&ctx->w2, g->w2, fV}, {&ctx->xb, g->x_bar, fV},
      {&ctx->w1b, g->w1_bar, fV}, {&ctx->w2b, g->w2_bar, fV},
      {&ctx->xp, g->x_prev ? g->x_prev : g->x, fV}, {&ctx->w1p, g->w1_prev ? g->w1_prev : g->w1, fV},
      {&ctx->w2p, g->w2_prev ? g->w2_prev : g->w2, fV}, {&ctx->data, g->data_term, fV},
      {&ctx->weight, g->data_weight, fV}, {&ctx->alpha, g->alpha, fE}, {&ctx->beta, g->beta, fE}, {&ctx->q1, g->q1, fE},
      {&ctx->q2, g->q2, fE}, {&ctx->q3, g->q3, fE}};
  bool on_device = false;  // the per-vertex layout tables by kernels (nltgv2_topo.hip) where that applies, else by the host builders
  rc = topo_upload(ctx, g, state, sizeof(state) / sizeof(state[0]), &on_device);
  if (!rc && !on_device) rc = upload_topology(ctx, g, state, sizeof(state) / sizeof(state[0]), /*long_lived=*/true);
  if (rc) return rc;
  const auto t_packed = std::chrono::steady_clock::now();
  LAUNCHCHK(ctx, launch_pack_static(ctx->c, ctx->f, ctx->stream));
  // No wait here: the caller's arrays were copied into the staging buffer, the device work is ordered on the stream in
  // front of whatever comes next (a run, an export), and the next upload synchronises before it reuses the staging buffer.
  if (trace) {
    HIPCHK(ctx, wait_solver_stream(ctx));
    const auto t_end = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[flame_nltgv2] upload_graph V=%d E=%d: host tables + enqueue %.3f ms, device %.3f ms\n", V, E,
                 std::chrono::duration<double, std::milli>(t_packed - t_begin).count(),
                 std::chrono::duration<double, std::milli>(t_end - t_packed).count());
  }
  ctx->h_feat.resize((size_t)V);
  for (int32_t v = 0; v < V; ++v) ctx->h_feat[(size_t)v] = v;  // default feature id = vertex index
  ctx->feat_map_valid = false, ctx->feat_tab_valid = false;
  ctx->canon_valid = true;
  ctx->fused_valid = false;
  ctx->have_prev = false;
  ctx->parity = 0;
  ctx->have_graph = true;
  ctx->last_error = 0;
  return FLAME_NLTGV2_OK;
}

// ---- per-frame graph synchronisation: the graph-edit part of Flame::syncGraph / projectGraph --------
// (flame.cc:1985-2121 and 1862-1938), on explicit orders instead of BGL's hash-set iteration order:
//   * a vertex whose feature id was in the previous graph keeps x,w1,w2,x_bar,w_bar,x_prev,w_prev
//     (warm start); pos / data_term / data_weight are replaced (flame.cc:1996-2001); with
//     check_sticky_obstacles, x is reset to data_term where x - data_term > threshold (flame.cc:2011-2014);
//     vertices absent from the new list disappear together with their edges (flame.cc:2020-2028,
//     1923-1931);
//   * a new vertex starts at x = x_bar = x_prev = init_x (or data_term), w = 0 (flame.cc:2035-2048,
//     2160-2162);
//   * an edge of the new triangulation that already connected the same two features keeps its dual
//     (q1,q2,q3) AND its old (source,target) orientation -- boost::edge(u,v) finds it either way
//     (flame.cc:2094-2100); other old edges are dropped (flame.cc:2108-2119); new edges get q = 0 and the
//     orientation (edges[2k], edges[2k+1]) = add_edge(v[e0], v[e1]) (flame.cc:2085-2096);
//   * every edge gets alpha = 1/||pos_a - pos_b|| from the NEW positions and beta = 1 (flame.cc:2087-2103);
//   * resulting edge order = surviving edges in their previous relative order, then the new edges in
//     triangulator order (boost::edges() walks a std::list: erase keeps order, add_edge appends).
}  // extern "C"

namespace flame_hip {
namespace host {

// init_with_prediction on the device: the gather reads the dense map the last interpolate_mesh left in r_img (behind the
// rasteriser if that is still running on its side stream); no map yet = a 0 x 0 map: every look-up is outside, i.e. NaN.
void set_init_map(flame_nltgv2_ctx* ctx, bool on, SyncArgs* sa) {
  if (!on) return;
  if (ctx->raster_inflight) (void)hipStreamWaitEvent(ctx->stream, ctx->ev_raster_done, 0);
  const bool have = ctx->map_rows > 0 && ctx->r_img.p;
  sa->init_map = have ? (const float*)ctx->r_img.p : (const float*)ctx->data.p;
  sa->map_rows = have ? ctx->map_rows : 0, sa->map_cols = have ? ctx->map_cols : 0;
}

static int sync_input_ok(flame_nltgv2_ctx* ctx, const flame_nltgv2_sync_input* in) {
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!in || in->V < 0 || in->E < 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (in->V > 0 && (!in->feat_id || !in->pos || !in->data_term || !in->data_weight)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (in->E > 0 && !in->edges) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (in->init_from_map && (in->init_x || !(in->init_graph_scale > 0.0f))) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  return 0;
}

// The host path of the per-frame sync (rounds 1-3): index maps from the host image of the previous topology, the new graph's
// per-vertex tables by the host builders; the state moves on the device.
int sync_graph_host(flame_nltgv2_ctx* ctx, const flame_nltgv2_sync_input* in) {
  const int32_t V = in->V, E = in->E;
  const bool trace = std::getenv("FLAME_NLTGV2_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  // The previous state stays on the device: the canonical arrays are brought up to date (a kernel, enqueued) while the host works
  // out the index maps below, from its image of the previous topology.
  int rc = cancel_prepared(ctx);
  if (!rc) rc = ensure_canon(ctx);
  if (!rc) rc = ensure_host_layout(ctx);
  if (rc) return rc;
  const int32_t Vo = ctx->L.V, Eo = ctx->L.E;

  // (the look-ups below -- ~V hash probes, ~E walks over a previous vertex's ~6 incident edges -- are independent of each
  //  other and read-only: they run in chunks on the library's worker threads, host_workers.hpp; what depends on order, the
  //  duplicate checks and "first of equal pairs wins", is done afterwards in one cheap sequential pass)
  const int parts = V + E >= 8192 ? std::min(Workers::get().threads(), 8) : 1;
  std::chrono::steady_clock::time_point tp[6];
  tp[0] = std::chrono::steady_clock::now();

  // vertices: new vertex -> its index in the previous graph (-1: new).  Feature ids are small non-negative counters in the
  // reference (flame.cc feature ids grow by one per detection): while the largest id stays below kFeatDirectMax the
  // id -> vertex map is a plain table (a probe is one load; ids arrive roughly in order, so it streams), beyond that a hash map.
  std::vector<int32_t>& old_of_new = ctx->h_old_of_new;
  old_of_new.assign((size_t)V, -1);
  std::atomic<int> bad{0};
  int32_t max_id = -1;
  for (int32_t v = 0; v < V; ++v) {
    if (in->feat_id[v] < 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
    max_id = std::max(max_id, in->feat_id[v]);
  }
  for (int32_t id : ctx->h_feat) max_id = std::max(max_id, id);
  const bool direct = max_id < kFeatDirectMax;
  if (direct) {
    if (ctx->feat_tab.size() <= (size_t)max_id) {
      ctx->feat_tab.resize((size_t)max_id + 1 + (size_t)max_id / 2, -1);
      ctx->feat_stamp.resize(ctx->feat_tab.size(), 0u);
      ctx->feat_tab_valid = false;
    }
    if (!ctx->feat_tab_valid) {  // (after an upload / set_feature_ids / a frame that went through the hash map)
      std::fill(ctx->feat_tab.begin(), ctx->feat_tab.end(), -1);
      for (int32_t v = 0; v < Vo; ++v) ctx->feat_tab[(size_t)ctx->h_feat[(size_t)v]] = v;
      ctx->feat_tab_valid = true;
    }
    const uint32_t stamp = ++ctx->feat_stamp_now;
    if (stamp == 0u) std::fill(ctx->feat_stamp.begin(), ctx->feat_stamp.end(), 0u), ctx->feat_stamp_now = 1u;
    for (int32_t v = 0; v < V; ++v) {
      const size_t id = (size_t)in->feat_id[v];
      if (ctx->feat_stamp[id] == ctx->feat_stamp_now) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);  // duplicate id
      ctx->feat_stamp[id] = ctx->feat_stamp_now;
      old_of_new[(size_t)v] = ctx->feat_tab[id];
    }
  } else {
    if (!ctx->feat_map_valid) {  // (after an upload / set_feature_ids; otherwise the map the previous sync built)
      FlatMap& m = ctx->feat_maps[ctx->feat_cur];
      m.reset((size_t)Vo);
      for (int32_t v = 0; v < Vo; ++v) m.emplace((uint64_t)(uint32_t)ctx->h_feat[(size_t)v], v);
      ctx->feat_map_valid = true;
    }
    const FlatMap& old_of_feat = ctx->feat_maps[ctx->feat_cur];
    parallel_chunks(V, parts, [&](int64_t v0, int64_t v1) {
      for (int64_t v = v0; v < v1; ++v) {
        const int32_t* it = old_of_feat.find((uint64_t)(uint32_t)in->feat_id[v]);
        if (it) old_of_new[(size_t)v] = *it;
      }
    });
    FlatMap& seen = ctx->feat_maps[ctx->feat_cur ^ 1];  // duplicate ids, and the next sync's old_of_feat
    seen.reset((size_t)V);
    for (int32_t v = 0; v < V; ++v)
      if (!seen.emplace((uint64_t)(uint32_t)in->feat_id[v], v).second) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);  // duplicate id
  }
  tp[1] = std::chrono::steady_clock::now();
  // edges.  A surviving edge joins two surviving vertices: it is looked up in the previous graph's adjacency (the
  // host copy of the packed layout: ~6 incident edges per vertex, one cache line) instead of a hash table over all
  // edges -- in chunks on the worker threads (host_workers.hpp): the look-ups are independent and read-only.  Of several
  // triangulator edges that join the same two features the FIRST keeps the old edge (boost::edge() finds it, nothing is
  // added for the others): an atomic minimum over the triangulator index per old edge.  Survivors must come out in their
  // PREVIOUS relative order (boost::edges() walks a std::list: erase keeps the order of the rest): read back by old edge id
  // in one pass -- no sort.
  std::vector<std::pair<int32_t, int32_t>> fresh;
  fresh.reserve((size_t)E);
  const std::vector<int32_t>& orow = ctx->L.row_ptr;
  const std::vector<uint32_t>& ohalf = ctx->L.half;
  const std::vector<int32_t>& onbr = ctx->L.half_nbr;
  std::vector<int32_t>& old_edge = ctx->h_old_edge_of_pair;  // per triangulator edge: the previous edge | orientation bit 31, -1 = none
  std::vector<int32_t>& first_k = ctx->h_first_pair_of_old;  // per previous edge: the first triangulator edge that keeps it
  old_edge.assign((size_t)E, -1);
  first_k.assign((size_t)Eo, 0x7fffffff);
  parallel_chunks(E, parts, [&](int64_t k0, int64_t k1) {
    for (int64_t k = k0; k < k1; ++k) {
      const int32_t a = in->edges[2 * k], b = in->edges[2 * k + 1];
      if (a < 0 || a >= V || b < 0 || b >= V || a == b) {
        bad.store(1);
        return;
      }
      const int32_t oa = old_of_new[(size_t)a], ob = old_of_new[(size_t)b];
      if (oa < 0 || ob < 0) continue;
      for (int32_t h = orow[(size_t)oa]; h < orow[(size_t)oa + 1]; ++h) {
        if (onbr[(size_t)h] == ob) {
          // the first (lowest id) of possible parallel edges, as boost::edge() on the list would find; bit 31: the old edge
          // runs ob -> oa
          const int32_t e = (int32_t)(ohalf[(size_t)h] & ~kRoleBit);
          old_edge[(size_t)k] = e | ((ohalf[(size_t)h] & kRoleBit) ? (int32_t)0x80000000u : 0);
          int32_t cur = __atomic_load_n(&first_k[(size_t)e], __ATOMIC_RELAXED);
          while ((int32_t)k < cur && !__atomic_compare_exchange_n(&first_k[(size_t)e], &cur, (int32_t)k, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
          }
          break;
        }
      }
    }
  });
  tp[2] = std::chrono::steady_clock::now();
  if (bad.load()) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  for (int32_t k = 0; k < E; ++k)
    if (old_edge[(size_t)k] == -1) fresh.emplace_back(in->edges[2 * k], in->edges[2 * k + 1]);
  int32_t n_keep = 0;
  for (int32_t o = 0; o < Eo; ++o) n_keep += first_k[(size_t)o] != 0x7fffffff;
  tp[3] = std::chrono::steady_clock::now();
  if (!fresh.empty() && !in->edges_unique) {
    // No parallel edges among the new ones either (boost::edge() finds the one just added): of several new edges between the
    // same two vertices the first stays.  The edges are bucketed by their lower vertex (a counting sort: two streaming passes;
    // a vertex has ~3 of them), equal pairs found inside a bucket, the later ones dropped in place -- no hashing.
    std::vector<int32_t>& start = ctx->h_bucket_start;
    std::vector<int32_t>& item = ctx->h_bucket_item;  // (higher vertex, index into fresh) pairs, bucket by bucket
    start.assign((size_t)V + 1, 0);
    item.resize(2 * fresh.size());
    for (const auto& f : fresh) start[(size_t)std::min(f.first, f.second) + 1]++;
    for (int32_t v = 0; v < V; ++v) start[(size_t)v + 1] += start[(size_t)v];
    {
      std::vector<int32_t>& at = ctx->h_bucket_at;
      at.assign(start.begin(), start.end() - 1);
      for (size_t i = 0; i < fresh.size(); ++i) {
        const int32_t lo = std::min(fresh[i].first, fresh[i].second), hi = std::max(fresh[i].first, fresh[i].second);
        const int32_t p = at[(size_t)lo]++;
        item[2 * (size_t)p] = hi, item[2 * (size_t)p + 1] = (int32_t)i;
      }
    }
    size_t dropped = 0;
    for (int32_t v = 0; v < V; ++v)
      for (int32_t i = start[(size_t)v] + 1; i < start[(size_t)v + 1]; ++i)      // (a bucket holds its edges in list order)
        for (int32_t j = start[(size_t)v]; j < i; ++j)
          if (item[2 * (size_t)j] == item[2 * (size_t)i] && fresh[(size_t)item[2 * (size_t)j + 1]].first >= 0) {
            fresh[(size_t)item[2 * (size_t)i + 1]].first = -1;  // the same pair as an earlier new edge
            ++dropped;
            break;
          }
    if (dropped) {
      size_t n = 0;
      for (const auto& f : fresh)
        if (f.first >= 0) fresh[n++] = f;
      fresh.resize(n);
    }
  }
  const int32_t En = (int32_t)((size_t)n_keep + fresh.size());
  std::vector<int32_t> src((size_t)En), dst((size_t)En);
  std::vector<int32_t>& old_of_new_edge = ctx->h_old_of_new_edge;
  old_of_new_edge.assign((size_t)En, -1);
  int32_t e = 0;
  for (int32_t o = 0; o < Eo; ++o) {
    const int32_t k = first_k[(size_t)o];
    if (k == 0x7fffffff) continue;
    const int32_t a = in->edges[2 * k], b = in->edges[2 * k + 1];
    const bool same = old_edge[(size_t)k] >= 0;  // the old edge runs old(a) -> old(b): it keeps that orientation
    src[(size_t)e] = same ? a : b, dst[(size_t)e] = same ? b : a;
    old_of_new_edge[(size_t)e] = o;
    ++e;
  }
  for (const auto& f : fresh) {
    src[(size_t)e] = f.first, dst[(size_t)e] = f.second;
    ++e;
  }
  const auto t1 = std::chrono::steady_clock::now();
  if (trace) {
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[flame_nltgv2] sync_graph maps: settle %.3f, vertices %.3f, edge look-ups %.3f (%d chunks), new-edge list %.3f, order + duplicates %.3f ms\n", ms(t0, tp[0]), ms(tp[0], tp[1]), ms(tp[1], tp[2]), parts, ms(tp[2], tp[3]), ms(tp[3], t1));
  }

  // New topology + the frame's inputs up, state gathered on the device out of the previous arrays into spare ones,
  // which then take their place.
  ctx->have_graph = false;  // (until the new graph stands)
  HIPCHK(ctx, wait_solver_stream(ctx));  // nothing in flight uses buffers that may be reallocated below
  const size_t fV = sizeof(float) * (size_t)V, fE = sizeof(float) * (size_t)En;
  DevBuf* cur_v[9] = {&ctx->x, &ctx->w1, &ctx->w2, &ctx->xb, &ctx->w1b, &ctx->w2b, &ctx->xp, &ctx->w1p, &ctx->w2p};
  DevBuf* cur_q[3] = {&ctx->q1, &ctx->q2, &ctx->q3};
  for (int i = 0; i < 9 && !rc; ++i) rc = ensure(ctx, ctx->sp_v[i], fV);
  for (int i = 0; i < 3 && !rc; ++i) rc = ensure(ctx, ctx->sp_q[i], fE);
  if (!rc) rc = ensure(ctx, ctx->data, fV);
  if (!rc) rc = ensure(ctx, ctx->weight, fV);
  if (!rc) rc = ensure(ctx, ctx->alpha, fE);
  if (!rc) rc = ensure(ctx, ctx->beta, fE);
  if (!rc) rc = ensure(ctx, ctx->sync_init, fV);
  if (!rc) rc = ensure(ctx, ctx->sync_vmap, sizeof(int32_t) * (size_t)V);
  if (!rc) rc = ensure(ctx, ctx->sync_emap, sizeof(int32_t) * (size_t)En);
  if (!rc) rc = ensure(ctx, ctx->sync_need, (size_t)V);
  if (rc) return rc;
  flame_nltgv2_graph g{};
  g.V = V, g.E = En;
  g.pos = const_cast<float*>(in->pos);
  g.src = src.data(), g.dst = dst.data();
  const StageCopy extra[] = {
      {&ctx->data, in->data_term, fV}, {&ctx->weight, in->data_weight, fV},
      {&ctx->sync_init, in->init_x, in->init_x ? fV : 0},
      {&ctx->sync_vmap, old_of_new.data(), sizeof(int32_t) * (size_t)V},
      {&ctx->sync_emap, old_of_new_edge.data(), sizeof(int32_t) * (size_t)En}};
  rc = upload_topology(ctx, &g, extra, sizeof(extra) / sizeof(extra[0]), /*long_lived=*/false);
  if (rc) return rc;
  SyncArgs sa;
  sa.V = V, sa.E = En;
  sa.old_of_new = (const int32_t*)ctx->sync_vmap.p, sa.old_of_new_edge = (const int32_t*)ctx->sync_emap.p;
  sa.data = (const float*)ctx->data.p, sa.weight = (const float*)ctx->weight.p;
  sa.init_x = in->init_x ? (const float*)ctx->sync_init.p : nullptr;
  set_init_map(ctx, in->init_from_map != 0, &sa);
  sa.check_sticky = in->check_sticky_obstacles ? 1 : 0, sa.sticky_threshold = in->sticky_threshold;
  sa.graph_scale = in->init_graph_scale;
  for (int i = 0; i < 9; ++i) sa.o[i] = (const float*)cur_v[i]->p, sa.n[i] = (float*)ctx->sp_v[i].p;
  for (int i = 0; i < 3; ++i) sa.oq[i] = (const float*)cur_q[i]->p, sa.nq[i] = (float*)ctx->sp_q[i].p;
  sa.src = (const int32_t*)ctx->src.p, sa.dst = (const int32_t*)ctx->dst.p, sa.row_ptr = (const int32_t*)ctx->row_ptr.p;
  sa.half = (const uint32_t*)ctx->half.p, sa.pos = (const float2*)ctx->pos.p;
  sa.alpha = (float*)ctx->alpha.p, sa.beta = (float*)ctx->beta.p, sa.need_nbr = (uint8_t*)ctx->sync_need.p;
  LAUNCHCHK(ctx, launch_sync_state(sa, ctx->stream));
  for (int i = 0; i < 9; ++i) std::swap(*cur_v[i], ctx->sp_v[i]);
  for (int i = 0; i < 3; ++i) std::swap(*cur_q[i], ctx->sp_q[i]);
  refresh_args(ctx);
  LAUNCHCHK(ctx, launch_pack_static(ctx->c, ctx->f, ctx->stream));
  if (trace) {  // (no wait otherwise: see flame_nltgv2_upload_graph)
    HIPCHK(ctx, wait_solver_stream(ctx));
    const auto t2 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[flame_nltgv2] sync_graph: index maps %.3f ms, tables + upload + device gather %.3f ms\n", ms(t0, t1), ms(t1, t2));
  }
  if (direct) {  // the table follows the graph: the previous graph's ids out, this one's in
    for (int32_t id : ctx->h_feat) ctx->feat_tab[(size_t)id] = -1;
    for (int32_t v = 0; v < V; ++v) ctx->feat_tab[(size_t)in->feat_id[v]] = v;
    ctx->feat_map_valid = false;  // (the hash maps no longer describe the current graph)
  } else {
    ctx->feat_cur ^= 1;  // (the map filled above is of the graph that stands now)
    ctx->feat_tab_valid = false;
  }
  ctx->h_feat.assign(in->feat_id, in->feat_id + V);
  ctx->canon_valid = true;
  ctx->fused_valid = false;
  ctx->have_prev = false;
  ctx->parity = 0;
  ctx->have_graph = true;
  ctx->last_error = 0;
  ctx->last_sync_path = 1;
  return FLAME_NLTGV2_OK;
}

}  // namespace host
}  // namespace flame_hip

extern "C" {

// The sync in one call: the device path (round 4, nltgv2_topo_capi.hip) -- index maps AND the new graph's layout tables by kernels
// over the resident previous topology -- takes the pipeline's case, the triangulator's duplicate-free edge list (edges_unique) with
// small feature ids, and declines the rest (a hub of more than 64 edges, ...), which goes the host way.
int flame_nltgv2_sync_graph(flame_nltgv2_ctx* ctx, const flame_nltgv2_sync_input* in) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_sync_graph");
  int rc = enter(ctx);
  if (!rc) rc = sync_input_ok(ctx, in);
  if (rc) return rc;
  if (ctx->opt_sync_path != 1) {
    bool applicable = false, done = false;
    rc = topo_prepare(ctx, in, &applicable);
    if (!rc && applicable) rc = topo_commit(ctx, &done);
    if (rc) return rc;
    if (done) return FLAME_NLTGV2_OK;
    if (ctx->opt_sync_path == 2) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);  // (asked for by name and not applicable)
  }
  return sync_graph_host(ctx, in);
}

// The same in two halves, for a caller whose solver keeps iterating (run_async) while the frame's graph is prepared -- the reference
// holds graph_mtx_ through all of Flame::syncGraph, its solver thread stands still (flame.cc:103, 309-318).  prepare: checks, one
// staged copy, the builder enqueued on a side stream; returns at once.  Between the two only runs (run / run_async / sync) and
// read-outs are allowed; anything that changes the topology cancels the prepared sync.  commit: waits for the builder, settles the
// runs, swaps the new topology in, gathers the state.  Where the device path does not apply the inputs are kept and commit does the
// whole sync the host way: the caller's code is the same either way.
int flame_nltgv2_sync_prepare(flame_nltgv2_ctx* ctx, const flame_nltgv2_sync_input* in) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_sync_prepare");
  int rc = enter(ctx);
  if (!rc) rc = sync_input_ok(ctx, in);
  if (rc) return rc;
  bool applicable = false;
  if (ctx->opt_sync_path != 1) {
    rc = topo_prepare(ctx, in, &applicable);
    if (rc) return rc;
    if (!applicable && ctx->opt_sync_path == 2) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  }
  if (!applicable) {  // keep the inputs (pinned, like the staged ones) for the host path at commit
    // ... after the value checks sync_graph's host path makes (the header: prepare reports the same errors as sync_graph; advisor,
    // round 4: a bad id or edge used to surface at commit, indistinguishable from "nothing prepared")
    {
      std::unordered_set<int32_t> ids;
      ids.reserve((size_t)in->V * 2);
      for (int32_t v = 0; v < in->V; ++v)
        if (in->feat_id[v] < 0 || !ids.insert(in->feat_id[v]).second) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      for (int32_t k = 0; k < in->E; ++k) {
        const int32_t a = in->edges[2 * k], b = in->edges[2 * k + 1];
        if (a < 0 || a >= in->V || b < 0 || b >= in->V || a == b) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      }
    }
    rc = cancel_prepared(ctx);
    if (rc) return rc;
    flame_nltgv2_ctx::PreparedSync& P = ctx->prepared;
    const size_t fV = sizeof(float) * (size_t)in->V;
    DevBuf none{};
    const StageCopy cp[] = {{&none, in->feat_id, sizeof(int32_t) * (size_t)in->V}, {&none, in->edges, sizeof(int32_t) * 2 * (size_t)in->E},
                            {&none, in->pos, 2 * fV}, {&none, in->data_term, fV}, {&none, in->data_weight, fV}, {&none, in->init_x, in->init_x ? fV : 0}};
    size_t total = 0;
    for (const StageCopy& c : cp) total += (c.bytes + 255) & ~size_t(255);
    ctx->prep_host.resize(total + 256);
    size_t off = 0;
    for (int i = 0; i < 6; ++i) {
      P.off[i] = off;
      if (cp[i].bytes) std::memcpy(ctx->prep_host.data() + off, cp[i].src, cp[i].bytes);
      off += (cp[i].bytes + 255) & ~size_t(255);
    }
    P.active = true, P.device = false, P.topo = ctx->topo, P.V = in->V, P.E = in->E, P.has_init = in->init_x != nullptr;
    P.check_sticky = in->check_sticky_obstacles, P.sticky_threshold = in->sticky_threshold, P.init_graph_scale = in->init_graph_scale;
    P.edges_unique = in->edges_unique, P.init_from_map = in->init_from_map;
  }
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_sync_commit(flame_nltgv2_ctx* ctx) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_sync_commit");
  int rc = enter(ctx);
  if (rc) return rc;
  flame_nltgv2_ctx::PreparedSync& P = ctx->prepared;
  if (!P.active || P.topo != ctx->topo || !ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);  // nothing prepared (or cancelled)
  const char* base = P.device ? static_cast<const char*>(ctx->stage[1].h) : ctx->prep_host.data();
  if (P.device) {
    bool done = false;
    rc = topo_commit(ctx, &done);
    if (rc || done) return rc;
  }
  P.active = false;
  flame_nltgv2_sync_input in{};  // the host way, from the kept inputs
  in.V = P.V, in.E = P.E;
  in.feat_id = reinterpret_cast<const int32_t*>(base + P.off[0]), in.edges = reinterpret_cast<const int32_t*>(base + P.off[1]);
  in.pos = reinterpret_cast<const float*>(base + P.off[2]), in.data_term = reinterpret_cast<const float*>(base + P.off[3]);
  in.data_weight = reinterpret_cast<const float*>(base + P.off[4]), in.init_x = P.has_init ? reinterpret_cast<const float*>(base + P.off[5]) : nullptr;
  in.check_sticky_obstacles = P.check_sticky, in.sticky_threshold = P.sticky_threshold, in.init_graph_scale = P.init_graph_scale;
  in.edges_unique = P.edges_unique, in.init_from_map = P.init_from_map;
  return sync_graph_host(ctx, &in);
}

int flame_nltgv2_project_graph(flame_nltgv2_ctx* ctx, const flame_nltgv2_projection* pr, float graph_scale,
                               uint8_t* keep_out, float* pos_out) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_project_graph");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!pr || !(graph_scale > 0.0f) || (ctx->L.V > 0 && !keep_out)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  const size_t V = (size_t)ctx->L.V;
  // the keep mask is written by the kernel straight into pinned host memory, the positions as they stood are kept by the same kernel
  // (round 6: as a device-to-device copy, the kernel and a copy out to the caller's pageable array this call was three more trips
  // through the runtime, 10-16 us each, with the solver standing still: profiles/r06_holds.txt)
  if (ctx->h_keep_cap < V + 16) {
    request_open_stop(ctx);  // (the pinned allocator waits for the device)
    if (ctx->h_keep) (void)hipHostFree(ctx->h_keep);
    ctx->h_keep = nullptr, ctx->h_keep_cap = 0;
    const size_t want = (V + 16) + (V + 16) / 2;
    if (hipHostMalloc((void**)&ctx->h_keep, want, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return fail(ctx, FLAME_NLTGV2_ERR_OOM);
    }
    ctx->h_keep_cap = want;
  }
  ProjectGeometry geo;
  std::memcpy(geo.K, pr->K, sizeof(geo.K));
  std::memcpy(geo.Kinv, pr->Kinv, sizeof(geo.Kinv));
  std::memcpy(geo.KRKinv, pr->KRKinv, sizeof(geo.KRKinv));
  std::memcpy(geo.q, pr->q_ref_to_cmp, sizeof(geo.q));
  std::memcpy(geo.t, pr->t_ref_to_cmp, sizeof(geo.t));
  geo.rx = pr->region_x, geo.ry = pr->region_y, geo.rw = pr->region_w, geo.rh = pr->region_h;
  // the old positions go to layout_pos if the layout was built from them (nltgv2_context.hpp: kept for its host image), to pos_undo
  // otherwise: either way the projection can be taken back
  const bool first_since_layout = !ctx->layout_pos_saved;
  DevBuf& save = first_since_layout ? ctx->layout_pos : ctx->pos_undo;
  rc = ensure(ctx, save, sizeof(float) * 2 * V);
  if (rc) return rc;
  const std::function<int()> project = [&]() -> int {
    LAUNCHCHK(ctx, launch_project_graph(ctx->c, graph_scale, geo, ctx->h_keep, (float2*)save.p, ctx->stream));
    return 0;
  };
  // The kernel goes out BEHIND the rounds in flight and the unpack of their state, before the host has seen how they ended: one wait of
  // the host with the solver standing still instead of two (~29 us each).  If the rounds expired the projection worked on rubbish: the
  // state has been unpacked again by the time ensure_canon returns, the positions come back from `save`, the kernel runs once more.
  int behind = kBehindNotLaunched;
  rc = ensure_canon(ctx, &project, &behind);
  if (behind == kBehindSpoiled && V) HIPCHK(ctx, hipMemcpyAsync(ctx->pos.p, save.p, sizeof(float) * 2 * V, hipMemcpyDeviceToDevice, ctx->stream));
  if (rc) return rc;
  if (behind != kBehindDone) {
    rc = project();
    if (rc) return rc;
  }
  if (first_since_layout && V) ctx->layout_pos_saved = true;
  if (V && pos_out) HIPCHK(ctx, hipMemcpyAsync(pos_out, ctx->pos.p, sizeof(float) * 2 * V, hipMemcpyDeviceToHost, ctx->stream));
  if (behind != kBehindDone || (V && pos_out)) HIPCHK(ctx, wait_solver_stream(ctx));
  if (V) std::memcpy(keep_out, ctx->h_keep, V);
  ctx->fused_valid = false;  // x changed; pos changed: alpha/dx/dy of the packed records are stale until the next
                             // sync_graph / upload_graph re-derives them (the reference re-triangulates right after)
  ctx->static_stale = true;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_rescale_data(flame_nltgv2_ctx* ctx, float graph_scale, float* new_graph_scale, flame_nltgv2_params* p) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_rescale_data");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!new_graph_scale || !p || ctx->L.V <= 0 || !(graph_scale > 0.0f)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  rc = ensure_canon(ctx);
  if (rc) return rc;
  LAUNCHCHK(ctx, launch_rescale(ctx->c, graph_scale, (float*)ctx->cost_out.p, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_cost, ctx->cost_out.p, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));
  const float ns = ctx->h_cost[0];
  p->data_factor *= ns / graph_scale;  // flame.cc:349
  *new_graph_scale = ns;
  ctx->fused_valid = false;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_set_feature_ids(flame_nltgv2_ctx* ctx, const int32_t* feat_id) {
  if (!ctx) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!feat_id && ctx->L.V > 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  FlatMap seen((size_t)ctx->L.V);
  for (int32_t v = 0; v < ctx->L.V; ++v)
    if (feat_id[v] < 0 || !seen.emplace((uint64_t)(uint32_t)feat_id[v], v).second) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (cancel_prepared(ctx) != 0) return ctx->last_error;  // (a prepared sync looked the old ids up)
  ctx->h_feat.assign(feat_id, feat_id + ctx->L.V);
  ctx->feat_map_valid = false, ctx->feat_tab_valid = false, ctx->feat_dev_valid = false;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_get_topology(flame_nltgv2_ctx* ctx, int32_t* src, int32_t* dst, int32_t* feat_id) {
  if (!ctx) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if ((src || dst) && !ctx->host_layout_valid) {  // (after a device-side build the edge list comes down on demand)
    int rc = enter(ctx);
    if (!rc) rc = ensure_host_layout(ctx);
    if (rc) return rc;
  }
  if (src) std::copy(ctx->h_src.begin(), ctx->h_src.end(), src);
  if (dst) std::copy(ctx->h_dst.begin(), ctx->h_dst.end(), dst);
  if (feat_id) std::copy(ctx->h_feat.begin(), ctx->h_feat.end(), feat_id);
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_graph_size(flame_nltgv2_ctx* ctx, int32_t* V, int32_t* E) {
  if (!ctx) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (V) *V = ctx->L.V;
  if (E) *E = ctx->L.E;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_update_data(flame_nltgv2_ctx* ctx, const float* data_term, const float* data_weight) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_update_data");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (ctx->L.V > 0 && (!data_term || !data_weight)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  rc = ensure_canon(ctx);
  if (rc) return rc;
  const size_t fV = sizeof(float) * (size_t)ctx->L.V;
  rc = h2d(ctx, ctx->data, data_term, fV);
  if (!rc) rc = h2d(ctx, ctx->weight, data_weight, fV);
  if (rc) return rc;
  HIPCHK(ctx, wait_solver_stream(ctx));
  ctx->fused_valid = false;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_upload_state(flame_nltgv2_ctx* ctx, const flame_nltgv2_graph* s) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!s) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  rc = ensure_canon(ctx);
  if (rc) return rc;
  const size_t fV = sizeof(float) * (size_t)ctx->L.V, fE = sizeof(float) * (size_t)ctx->L.E;
  const struct { DevBuf* b; const float* src; size_t bytes; } cp[] = {
      {&ctx->x, s->x, fV}, {&ctx->w1, s->w1, fV}, {&ctx->w2, s->w2, fV}, {&ctx->xb, s->x_bar, fV},
      {&ctx->w1b, s->w1_bar, fV}, {&ctx->w2b, s->w2_bar, fV}, {&ctx->xp, s->x_prev, fV},
      {&ctx->w1p, s->w1_prev, fV}, {&ctx->w2p, s->w2_prev, fV}, {&ctx->q1, s->q1, fE}, {&ctx->q2, s->q2, fE},
      {&ctx->q3, s->q3, fE}};
  for (auto& c : cp) {
    if (!c.src) continue;
    rc = h2d(ctx, *c.b, c.src, c.bytes);
    if (rc) return rc;
  }
  HIPCHK(ctx, hipMemsetAsync(ctx->err.p, 0, kErrBytes, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));
  ctx->fused_valid = false;
  ctx->last_error = 0;
  return FLAME_NLTGV2_OK;
}


int flame_nltgv2_costs(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, float* smoothness, float* data) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_costs");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!params_ok(p)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  rc = ensure_canon(ctx);
  if (rc) return rc;
  // The reference adds the terms up sequentially in float (edge order, then vertex order): the device forms the
  // addends, the host adds them in that order -- the result is the reference's to the last bit.  (A statistics call,
  // flame.cc:2172-2173: 2E + V dependent additions, ~60 us at 640x480.)
  const size_t E = (size_t)ctx->L.E, V = (size_t)ctx->L.V, n = 2 * E + V;
  rc = ensure(ctx, ctx->cost_terms, sizeof(float) * n);
  if (rc) return rc;
  LAUNCHCHK(ctx, launch_cost_terms(ctx->c, (float*)ctx->cost_terms.p, ctx->stream));
  if (ctx->opt_cost_sum == 1) {
    // FLAME_NLTGV2_OPT_COST_SUM = 1: both sums by k_block_sum on the device, 8 bytes come back (the statistics of a frame loop,
    // flame.cc:2172-2173: ~10 us instead of a 2E + V float copy and as many dependent additions on the host)
    LAUNCHCHK(ctx, launch_cost_sums(ctx->c, (const float*)ctx->cost_terms.p, (float*)ctx->cost_out.p, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_cost, ctx->cost_out.p, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, wait_solver_stream(ctx));
    if (smoothness) *smoothness = p->data_factor * ctx->h_cost[0];
    if (data) *data = ctx->h_cost[1];
    return FLAME_NLTGV2_OK;
  }
  ctx->h_terms.resize(n);
  if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->h_terms.data(), ctx->cost_terms.p, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));
  float cost = 0.0f;
  for (size_t k = 0; k < 2 * E; ++k) cost += ctx->h_terms[k];
  float dcost = 0.0f;
  for (size_t k = 0; k < V; ++k) dcost += ctx->h_terms[2 * E + k];
  if (smoothness) *smoothness = p->data_factor * cost;
  if (data) *data = dcost;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_download_state(flame_nltgv2_ctx* ctx, flame_nltgv2_graph* out) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_download_state");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!out) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  rc = ensure_canon(ctx);
  if (rc) return rc;
  const size_t fV = sizeof(float) * (size_t)ctx->L.V, fE = sizeof(float) * (size_t)ctx->L.E;
  const struct { float* dst; DevBuf* b; size_t bytes; } cp[] = {
      {out->x, &ctx->x, fV}, {out->w1, &ctx->w1, fV}, {out->w2, &ctx->w2, fV}, {out->x_bar, &ctx->xb, fV},
      {out->w1_bar, &ctx->w1b, fV}, {out->w2_bar, &ctx->w2b, fV}, {out->x_prev, &ctx->xp, fV},
      {out->w1_prev, &ctx->w1p, fV}, {out->w2_prev, &ctx->w2p, fV}, {out->q1, &ctx->q1, fE},
      {out->q2, &ctx->q2, fE}, {out->q3, &ctx->q3, fE}};
  for (auto& c : cp) {
    if (!c.dst || c.bytes == 0) continue;
    HIPCHK(ctx, hipMemcpyAsync(c.dst, c.b->p, c.bytes, hipMemcpyDeviceToHost, ctx->stream));
  }
  out->V = ctx->L.V, out->E = ctx->L.E;
  return finish(ctx);
}

static int export_idepth(flame_nltgv2_ctx* ctx, void* dst_device, float scale, bool wait) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!dst_device && ctx->L.V > 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (ctx->pending.active) {
    // an unchecked persistent run is in flight: the waiting form settles it first (so that what it copies out is the
    // checked result), the asynchronous form joins the chain and is redone should the chain have to be replayed
    if (wait || ctx->open_inflight) {  // (nothing is chained behind an open run)
      rc = finish(ctx);
    } else {
      rc = snapshot_chain_start(ctx);
      if (!rc) {
        flame_nltgv2_ctx::PendingOp op;
        op.kind = 1, op.dst = (float*)dst_device, op.scale = scale;
        ctx->pending.ops.push_back(op);
      }
    }
    if (rc) return rc;
  }
  const bool packed = !ctx->canon_valid;
  LAUNCHCHK(ctx, launch_export(ctx->c, ctx->f, packed, scale, (float*)dst_device, ctx->stream));
  if (wait) HIPCHK(ctx, wait_solver_stream(ctx));
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_set_export_target(flame_nltgv2_ctx* ctx, void* dst_device, float scale) {
  if (!ctx) return FLAME_NLTGV2_ERR_INVALID_ARG;
  ctx->export_ptr = (float*)dst_device;
  ctx->export_scale = scale;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_export_idepth_device(flame_nltgv2_ctx* ctx, void* dst_device, float scale) {
  return export_idepth(ctx, dst_device, scale, true);
}

int flame_nltgv2_export_idepth_device_async(flame_nltgv2_ctx* ctx, void* dst_device, float scale) {
  return export_idepth(ctx, dst_device, scale, false);
}


}  // extern "C"
