// nltgv2_topo_capi.hip -- host side of the device-built topology (nltgv2_topo.hip): the per-frame warm-start sync
// (flame_nltgv2_sync_graph; Flame::syncGraph's graph edits, /root/reference/src/flame/flame.cc:1985-2121) without host index maps and
// without host layout tables, and the host image of a device-built topology on demand.
//
// What the host still does per frame: check the inputs the way the C-ABI promises (ids unique and >= 0, edges in range, positions
// finite: three streaming passes), stage the frame's arrays into ONE pinned blob, enqueue the builder on a side stream (prepare);
// then (commit) read 64 bytes of dimensions back (rows, patches, largest degree), stop the solver, swap the next topology in, size
// the slot / lane arrays and enqueue their expansion and the state gather.  The reference holds graph_mtx_ -- the solver stands
// still -- through all of Flame::syncGraph including the triangulation (flame.cc:309-318, 2052-2071); here it stands still for
// the commit only.
#include "nltgv2_context.hpp"

namespace flame_hip {
namespace host {

namespace {

struct Carve {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) & ~size_t(255);
    return at;
  }
};

constexpr size_t kWalkPad = 256 * 64;  // k_topo_walk stages whole workgroups of 64 segments

// Where the builder's scratch arrays lie in ctx->topo_scratch (sync: + the frame's ids / edges / init values and the index maps).
struct ScratchMap {
  size_t fid, edges, init, old_edge, first_k, scan, deg, cur, parent, minid, morton, key_out, width, wflag, vf0, vf1, seg0, seg1, counters, wg2_v0, vmap,
      emap, old_feat, sort, total, sort_bytes, vpad;
  int cc_bits, n_slices;
};
ScratchMap make_scratch(int32_t V, int32_t E, int32_t Vo, int32_t Eo, bool sync) {
  ScratchMap m{};
  const size_t fV = sizeof(float) * (size_t)V, fE = sizeof(float) * (size_t)E, iV = sizeof(int32_t) * (size_t)V;
  m.n_slices = (V + kWave - 1) / kWave;
  m.vpad = ((size_t)V + kWalkPad - 1) / kWalkPad * kWalkPad;
  const int n_seg = (V + 255) / 256;
  const int n_scan = (sync ? Eo + E : 0) + V + 1;
  m.sort_bytes = topo_sort_temp_bytes(V, n_scan);
  m.cc_bits = 1;
  while ((1 << m.cc_bits) < V) ++m.cc_bits;
  const size_t iM = sizeof(int32_t) << m.cc_bits;
  Carve cv;
  m.fid = cv.take(sync ? iV : 0), m.edges = cv.take(sync ? 2 * fE : 0), m.init = cv.take(sync ? fV : 0), m.old_edge = cv.take(sync ? fE : 0);
  m.first_k = cv.take(sizeof(int32_t) * (size_t)std::max(sync ? Eo : 0, 1));
  m.scan = cv.take(sizeof(int32_t) * (size_t)n_scan), m.deg = cv.take(iV), m.cur = cv.take(iV), m.parent = cv.take(iM), m.minid = cv.take(iM), m.morton = cv.take(iV);
  m.key_out = cv.take(8 * (size_t)V), m.width = cv.take(sizeof(int32_t) * (size_t)m.n_slices);
  m.wflag = cv.take(m.vpad), m.vf0 = cv.take(m.vpad), m.vf1 = cv.take(m.vpad), m.seg0 = cv.take(sizeof(int32_t) * (size_t)n_seg), m.seg1 = cv.take(sizeof(int32_t) * (size_t)n_seg);
  m.counters = cv.take(64), m.wg2_v0 = cv.take(iV), m.vmap = cv.take(sync ? iV : 0), m.emap = cv.take(sync ? fE : 0);
  m.old_feat = cv.take(sizeof(int32_t) * (size_t)std::max(sync ? Vo : 0, 1)), m.sort = cv.take(m.sort_bytes);
  m.total = cv.off;
  return m;
}
void bind_scratch(TopoBuild* t, char* sc, const ScratchMap& m) {
  t->n_slices = m.n_slices;
  t->old_edge = (int32_t*)(sc + m.old_edge), t->first_k = (int32_t*)(sc + m.first_k), t->scan = (int32_t*)(sc + m.scan);
  t->deg = (int32_t*)(sc + m.deg), t->cur = (int32_t*)(sc + m.cur), t->parent = (int32_t*)(sc + m.parent), t->morton = (uint32_t*)(sc + m.morton);
  t->cc_bits = m.cc_bits, t->minid = (int32_t*)(sc + m.minid);
  t->key_out = (uint64_t*)(sc + m.key_out), t->width = (int32_t*)(sc + m.width);
  t->wflag = (uint8_t*)(sc + m.wflag), t->vf[0] = (uint8_t*)(sc + m.vf0), t->vf[1] = (uint8_t*)(sc + m.vf1);
  t->seg_count[0] = (int32_t*)(sc + m.seg0), t->seg_count[1] = (int32_t*)(sc + m.seg1);
  t->sort_tmp = sc + m.sort, t->sort_tmp_bytes = m.sort_bytes, t->counters = (int*)(sc + m.counters);
  t->wg2_v0 = (int32_t*)(sc + m.wg2_v0);
}

}  // namespace

// Brings the host image of the current topology up to date after a device-side build: the edge list comes down, the host
// builders redo the tables (they are the reference the device tables are compared with: flame_nltgv2_layout_selftest).
int ensure_host_layout(flame_nltgv2_ctx* ctx) {
  if (ctx->host_layout_valid) return 0;
  const int32_t V = ctx->L.V, E = ctx->L.E;
  std::vector<float> pos(2 * (size_t)V);
  ctx->h_src.resize((size_t)E), ctx->h_dst.resize((size_t)E);
  if (V) HIPCHK(ctx, hipMemcpyAsync(pos.data(), ctx->layout_pos_saved ? ctx->layout_pos.p : ctx->pos.p, sizeof(float) * pos.size(), hipMemcpyDeviceToHost, ctx->stream));
  if (E) HIPCHK(ctx, hipMemcpyAsync(ctx->h_src.data(), ctx->src.p, sizeof(int32_t) * (size_t)E, hipMemcpyDeviceToHost, ctx->stream));
  if (E) HIPCHK(ctx, hipMemcpyAsync(ctx->h_dst.data(), ctx->dst.p, sizeof(int32_t) * (size_t)E, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));
  flame_nltgv2_graph g{};
  g.V = V, g.E = E, g.pos = pos.data(), g.src = ctx->h_src.data(), g.dst = ctx->h_dst.data();
  const PackedLayout& dev = ctx->L;  // (scalars as the device reported them; its vectors are stale)
  // built aside and committed only once it agrees with the device's tables: a mismatch leaves the context as it was (advisor, round 4)
  PackedLayout H;
  int rc = build_layout(&g, &H, /*host_expand=*/false, /*rowpack=*/dev.wg_rowpack);
  if (rc) return fail(ctx, rc);
  if (dev.wg2_walked) build_patch_walk2(&H);
  H.tv_ok = dev.tv_ok, H.tv_waves = dev.tv_waves;
  const bool same = H.rows == dev.rows && H.n_slices == dev.n_slices && H.max_degree == dev.max_degree && H.wg_count == dev.wg_count &&
                    H.wg_lcap == dev.wg_lcap && H.wg_rowpack == dev.wg_rowpack &&
                    (!dev.wg2_ok || (H.wg2_ok && H.wg2_count == dev.wg2_count && H.wg2_lcap == dev.wg2_lcap));  // (the device walks (E2) always, the
                                                                                                                 //  host only up to 32 edges per vertex)
  if (!same) {  // the device tables and the host builders disagree: a bug, never silent
    std::fprintf(stderr, "[flame_nltgv2] host image of a device-built layout differs: rows %ld / %ld, slices %d / %d, max degree %d / %d, patches %d / %d, "
                 "lcap %d / %d, rowpack %d / %d, (E2) ok %d / %d patches %d / %d lcap %d / %d (host / device)\n", (long)H.rows, (long)dev.rows, H.n_slices, dev.n_slices,
                 H.max_degree, dev.max_degree, H.wg_count, dev.wg_count, H.wg_lcap, dev.wg_lcap, (int)H.wg_rowpack, (int)dev.wg_rowpack, (int)H.wg2_ok,
                 (int)dev.wg2_ok, H.wg2_count, dev.wg2_count, H.wg2_lcap, dev.wg2_lcap);
    return fail(ctx, FLAME_NLTGV2_ERR_HIP);
  }
  ctx->L = std::move(H);
  ctx->host_layout_valid = true;
  return 0;
}

int cancel_prepared(flame_nltgv2_ctx* ctx) {
  if (!ctx->prepared.active) return 0;
  if (ctx->prepared.device) HIPCHK(ctx, hipStreamSynchronize(ctx->topo_stream));
  ctx->prepared.active = false;
  return 0;
}

// The per-frame sync with the topology built on the device, in two halves.
//
// topo_prepare: the checks the C-ABI promises, the frame's arrays staged into ONE pinned blob, the builder enqueued on the SIDE
// stream -- it reads the live topology (CSR, edge list) and the feature table and writes the next topology into buffers of its own
// (ctx->nx), so the solver may keep iterating on the context's stream meanwhile (run_async); nothing the solver reads is touched.
// *applicable = false: a case the device path does not take (no duplicate-free edge list vouched for, ids beyond the direct table,
// fewer than two vertices or no edge); nothing has been done.
//
// topo_commit: waits for the builder's 64 bytes of dimensions, stops the solver (settles the chain of runs, state to the canonical
// arrays), swaps the next topology in, expands the slot / lane arrays and gathers the state.  *done = false: the builder declined
// (a hub of more than 64 edges, a pair listed twice, a graph beyond the row-packed patch form): the previous graph is whole, the
// caller goes the host way with the staged inputs.
int topo_prepare(flame_nltgv2_ctx* ctx, const flame_nltgv2_sync_input* in, bool* applicable) {
  *applicable = false;
  const int32_t V = in->V, E = in->E;
  const int32_t Vo = ctx->L.V, Eo = ctx->L.E;
  if (!in->edges_unique || V < 2 || E < 1 || V > (1 << 22) || (int64_t)Eo + E + V + 1 > 0x7fff0000ll) return 0;
  flame_nltgv2_ctx::PreparedSync& P = ctx->prepared;
  P.t_begin = std::chrono::steady_clock::now();
  // ---- the checks the C-ABI promises (INVALID_ARG before anything is changed) -------------------------------------------------
  for (int32_t v = 0; v < V; ++v)
    if (in->feat_id[v] < 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  {  // duplicate ids: a stamped open-addressing set of 4 V slots (the ids themselves may be anything up to 2^31 - 1)
    size_t cap = 1024;
    while (cap < 4 * (size_t)V) cap <<= 1;
    if (ctx->dup_key.size() != cap) ctx->dup_key.assign(cap, 0), ctx->dup_stamp.assign(cap, 0u), ctx->dup_now = 0u;
    if (++ctx->dup_now == 0u) std::fill(ctx->dup_stamp.begin(), ctx->dup_stamp.end(), 0u), ctx->dup_now = 1u;
    const size_t mask = cap - 1;
    for (int32_t v = 0; v < V; ++v) {
      const int32_t id = in->feat_id[v];
      size_t h = ((uint32_t)id * 0x9E3779B1u) & mask;
      while (ctx->dup_stamp[h] == ctx->dup_now) {
        if (ctx->dup_key[h] == id) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);  // duplicate id
        h = (h + 1) & mask;
      }
      ctx->dup_stamp[h] = ctx->dup_now, ctx->dup_key[h] = id;
    }
  }
  {
    uint32_t bad = 0;
    for (int32_t k = 0; k < E; ++k) {
      const uint32_t a = (uint32_t)in->edges[2 * k], b = (uint32_t)in->edges[2 * k + 1];
      bad |= (a >= (uint32_t)V) | (b >= (uint32_t)V) | (a == b);
    }
    if (bad) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  }
  float minx = in->pos[0], maxx = minx, miny = in->pos[1], maxy = miny;
  for (int32_t v = 0; v < V; ++v) {
    const float px = in->pos[2 * v], py = in->pos[2 * v + 1];
    if (!std::isfinite(px) || !std::isfinite(py)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
    minx = std::min(minx, px), maxx = std::max(maxx, px);
    miny = std::min(miny, py), maxy = std::max(maxy, py);
  }
  int rc = cancel_prepared(ctx);  // (a sync prepared earlier and never committed)
  if (rc) return rc;

  // ---- buffers: scratch and the next topology.  Nothing here belongs to the live graph: growing one of them frees memory the
  // solver does not use (a hipFree still waits for the device: buffers grow geometrically, so a steady stream of frames stops growing)
  hipStream_t ts = ctx->topo_stream;
  const ScratchMap sm = make_scratch(V, E, Vo, Eo, /*sync=*/true);
  const int n_slices = sm.n_slices;
  const size_t n_packed = (size_t)n_slices * kWave;
  const size_t fV = sizeof(float) * (size_t)V, fE = sizeof(float) * (size_t)E, iV = sizeof(int32_t) * (size_t)V;
  const size_t vpad = sm.vpad;
  const size_t o_fid = sm.fid, o_edges = sm.edges, o_init = sm.init, o_vmap = sm.vmap, o_emap = sm.emap, o_old_feat = sm.old_feat;
  struct { size_t off; } cv{sm.total};
  const size_t nx_bytes[flame_nltgv2_ctx::NX_COUNT] = {
      2 * fV, fE, fE, sizeof(int32_t) * ((size_t)V + 1), 2 * fE, iV, iV, sizeof(int32_t) * n_packed, iV, sizeof(int32_t) * n_packed,
      sizeof(int32_t) * ((size_t)n_slices + 1), 4 * iV, iV, vpad + 16, 4 * iV, vpad + 16, fV, fV};  // (patch tables by their upper bound: a patch holds at least one vertex)
  // (the previous commit's kernels may still read the scratch maps and write what this builder reads: it starts behind them)
  HIPCHK(ctx, hipStreamWaitEvent(ts, ctx->ev_topo_ready, 0));
  if (ctx->raster_inflight) HIPCHK(ctx, hipStreamWaitEvent(ts, ctx->ev_raster_done, 0));  // (it may read a position buffer that was swapped out)
  if (ctx->topo_scratch.cap < cv.off) HIPCHK(ctx, wait_solver_stream(ctx));  // (growing the scratch under the previous commit's state gather)
  rc = ensure(ctx, ctx->topo_scratch, cv.off);
  for (int i = 0; i < flame_nltgv2_ctx::NX_COUNT && !rc; ++i) rc = ensure(ctx, ctx->nx[i], nx_bytes[i]);
  if (!rc) rc = ensure(ctx, ctx->topo_dims, sizeof(TopoDims));
  if (rc) return rc;
  if (!ctx->h_dims) request_open_stop(ctx);
  if (!ctx->h_dims && hipHostMalloc((void**)&ctx->h_dims, sizeof(TopoDims), hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return fail(ctx, FLAME_NLTGV2_ERR_OOM);
  }
  char* const sc = static_cast<char*>(ctx->topo_scratch.p);
  // the device's feature tables: two of 2^bits slots, at least four per vertex (grown: rebuilt from the current graph's ids)
  {
    int bits = 12;
    while (((size_t)1 << bits) < 4 * (size_t)std::max(V, Vo)) ++bits;
    if (bits > ctx->feat_tab_bits_d) {
      const size_t want = (size_t)2 << bits;
      rc = ensure(ctx, ctx->feat_stamp_d, sizeof(uint32_t) * want);
      if (!rc) rc = ensure(ctx, ctx->feat_key_d, sizeof(int32_t) * want);
      if (!rc) rc = ensure(ctx, ctx->feat_val_d, sizeof(int32_t) * want);
      if (rc) return rc;
      ctx->feat_tab_bits_d = bits;
      ctx->feat_dev_valid = false;
    }
  }
  const bool rebuild_table = !ctx->feat_dev_valid || ctx->feat_gen >= 0xfffffff0u;
  if (rebuild_table) {
    HIPCHK(ctx, hipMemsetAsync(ctx->feat_stamp_d.p, 0, sizeof(uint32_t) * ((size_t)2 << ctx->feat_tab_bits_d), ts));
    ctx->feat_gen = 1;
  }

  // ---- one blob up: the frame's inputs (and, when the table is rebuilt, the live graph's ids) -----------------------------------------
  DevBuf b_fid{sc + o_fid, iV}, b_edges{sc + o_edges, 2 * fE}, b_init{sc + o_init, fV}, b_old{sc + o_old_feat, sizeof(int32_t) * (size_t)std::max(Vo, 1)};
  const StageCopy cp[] = {{&b_fid, in->feat_id, iV}, {&b_edges, in->edges, 2 * fE}, {&ctx->nx[flame_nltgv2_ctx::NX_POS], in->pos, 2 * fV},
                          {&ctx->nx[flame_nltgv2_ctx::NX_DATA], in->data_term, fV}, {&ctx->nx[flame_nltgv2_ctx::NX_WEIGHT], in->data_weight, fV},
                          {&b_init, in->init_x, in->init_x ? fV : 0},
                          {&b_old, ctx->h_feat.data(), rebuild_table ? sizeof(int32_t) * (size_t)Vo : 0}};
  size_t hoff[7];
  rc = staged_h2d(ctx, cp, sizeof(cp) / sizeof(cp[0]), nullptr, 0, /*slot=*/1, ts, hoff);
  if (rc) return rc;
  if (rebuild_table && Vo > 0)
    LAUNCHCHK(ctx, launch_topo_feat_build((const int32_t*)(sc + o_old_feat), Vo, (uint32_t*)ctx->feat_stamp_d.p, (int32_t*)ctx->feat_key_d.p,
                                          (int32_t*)ctx->feat_val_d.p, ctx->feat_tab_bits_d, ctx->feat_gen, ts));

  // ---- the builder ------------------------------------------------------------------------------------------------------------------
  using C = flame_nltgv2_ctx;
  TopoBuild t;
  t.V = V, t.E = E, t.Vo = Vo, t.Eo = Eo;
  t.fid = (const int32_t*)(sc + o_fid), t.pos = (const float2*)ctx->nx[C::NX_POS].p, t.tri_edges = (const int32_t*)(sc + o_edges);
  t.minx = minx, t.miny = miny;
  t.sx = (maxx > minx) ? 65535.0f / (maxx - minx) : 0.0f, t.sy = (maxy > miny) ? 65535.0f / (maxy - miny) : 0.0f;
  t.o_row_ptr = (const int32_t*)ctx->row_ptr.p, t.o_half = (const uint32_t*)ctx->half.p;
  t.o_src = (const int32_t*)ctx->src.p, t.o_dst = (const int32_t*)ctx->dst.p;
  t.feat_stamp = (uint32_t*)ctx->feat_stamp_d.p, t.feat_key = (int32_t*)ctx->feat_key_d.p, t.feat_val = (int32_t*)ctx->feat_val_d.p, t.tab_bits = ctx->feat_tab_bits_d;
  t.gen_prev = ctx->feat_gen, t.gen_new = ctx->feat_gen + 1;
  bind_scratch(&t, sc, sm);
  t.old_of_new = (int32_t*)(sc + o_vmap), t.old_of_new_edge = (int32_t*)(sc + o_emap);
  t.src = (int32_t*)ctx->nx[C::NX_SRC].p, t.dst = (int32_t*)ctx->nx[C::NX_DST].p;
  t.row_ptr = (int32_t*)ctx->nx[C::NX_ROW_PTR].p, t.half = (uint32_t*)ctx->nx[C::NX_HALF].p;
  t.order_m = (int32_t*)ctx->nx[C::NX_ORDER_M].p, t.rid_of = (int32_t*)ctx->nx[C::NX_RID_OF].p, t.perm = (int32_t*)ctx->nx[C::NX_PERM].p;
  t.iperm = (int32_t*)ctx->nx[C::NX_IPERM].p, t.pdeg = (int32_t*)ctx->nx[C::NX_PDEG].p, t.slice_row = (int32_t*)ctx->nx[C::NX_SLICE_ROW].p;
  t.wg_info = (int32_t*)ctx->nx[C::NX_WG_INFO].p, t.wg_v0 = (int32_t*)ctx->nx[C::NX_WG_V0].p, t.wg_vfirst = (uint8_t*)ctx->nx[C::NX_WG_VFIRST].p;
  t.wg2_info = (int32_t*)ctx->nx[C::NX_WG2_INFO].p, t.wg2_vfirst = (uint8_t*)ctx->nx[C::NX_WG2_VFIRST].p;
  t.dims = (TopoDims*)ctx->topo_dims.p;
  ctx->feat_dev_valid = false;  // (the table moves on to the new graph: valid again once that graph stands)
  LAUNCHCHK(ctx, launch_topo_sync_front(t, ts));
  LAUNCHCHK(ctx, launch_topo_back(t, ts));
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_dims, ctx->topo_dims.p, sizeof(TopoDims), hipMemcpyDeviceToHost, ts));
  P.active = true, P.device = true, P.topo = ctx->topo, P.V = V, P.E = E, P.has_init = in->init_x != nullptr;
  P.check_sticky = in->check_sticky_obstacles ? 1 : 0, P.sticky_threshold = in->sticky_threshold, P.init_graph_scale = in->init_graph_scale;
  P.edges_unique = in->edges_unique, P.init_from_map = in->init_from_map;
  for (int i = 0; i < 6; ++i) P.off[i] = hoff[i];
  ctx->prep_vmap = sc + o_vmap, ctx->prep_emap = sc + o_emap, ctx->prep_init = sc + o_init;
  P.t_enqueued = std::chrono::steady_clock::now();
  *applicable = true;
  return 0;
}

// upload_graph with the per-vertex tables built on the device: the caller's (pos, src, dst) go up in the one staged copy together
// with the state, the builder runs on the context's stream in front of the slot / lane expansion.  *done = false: not a case for
// the device builder (fewer than two vertices, no edge, a hub of more than 64 edges, a graph beyond the row-packed patch form,
// FLAME_NLTGV2_OPT_SYNC_PATH = 1): the caller goes on with upload_topology, the host builders.
int topo_upload(flame_nltgv2_ctx* ctx, const flame_nltgv2_graph* g, const StageCopy* extra, size_t n_extra, bool* done) {
  *done = false;
  const int32_t V = g->V, E = g->E;
  const int cus = ctx->prop.multiProcessorCount;
  const bool rowpack = ctx->opt_persistent == 4 || (static_cast<int64_t>(2) * E + V / 32) / 54 + 1 <= (int64_t)kPvDensePerCu * cus;
  if (ctx->opt_sync_path == 1 || !rowpack || V < 2 || E < 1 || V > (1 << 22) || (int64_t)E + V + 1 > 0x7fff0000ll) return 0;
  {
    uint32_t bad = 0;
    for (int32_t k = 0; k < E; ++k) {
      const uint32_t a = (uint32_t)g->src[k], b = (uint32_t)g->dst[k];
      bad |= (a >= (uint32_t)V) | (b >= (uint32_t)V) | (a == b);
    }
    if (bad) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  }
  float minx = g->pos[0], maxx = minx, miny = g->pos[1], maxy = miny;
  for (int32_t v = 0; v < V; ++v) {
    const float px = g->pos[2 * v], py = g->pos[2 * v + 1];
    if (!std::isfinite(px) || !std::isfinite(py)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
    minx = std::min(minx, px), maxx = std::max(maxx, px);
    miny = std::min(miny, py), maxy = std::max(maxy, py);
  }
  int rc = cancel_prepared(ctx);
  if (rc) return rc;
  const ScratchMap sm = make_scratch(V, E, 0, 0, /*sync=*/false);
  const size_t n_packed = (size_t)sm.n_slices * kWave;
  const size_t fV = sizeof(float) * (size_t)V, fE = sizeof(float) * (size_t)E, iV = sizeof(int32_t) * (size_t)V;
  struct { DevBuf* b; size_t bytes; } req[] = {
      {&ctx->topo_scratch, sm.total}, {&ctx->topo_dims, sizeof(TopoDims)}, {&ctx->pos, 2 * fV}, {&ctx->src, fE}, {&ctx->dst, fE},
      {&ctx->row_ptr, sizeof(int32_t) * ((size_t)V + 1)}, {&ctx->half, 2 * fE}, {&ctx->slice_row, sizeof(int32_t) * ((size_t)sm.n_slices + 1)},
      {&ctx->perm, sizeof(int32_t) * n_packed}, {&ctx->pdeg, sizeof(int32_t) * n_packed}, {&ctx->iperm, iV}, {&ctx->order_m, iV}, {&ctx->rid_of, iV},
      {&ctx->wg_info, 4 * iV}, {&ctx->wg_v0, iV}, {&ctx->wg_vfirst, sm.vpad + 16}, {&ctx->wg2_info, 4 * iV}, {&ctx->wg2_vfirst, sm.vpad + 16}};
  for (auto& r : req) {
    rc = ensure(ctx, *r.b, r.bytes);
    if (rc) return rc;
  }
  if (!ctx->h_dims) request_open_stop(ctx);
  if (!ctx->h_dims && hipHostMalloc((void**)&ctx->h_dims, sizeof(TopoDims), hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return fail(ctx, FLAME_NLTGV2_ERR_OOM);
  }
  if (ctx->raster_inflight) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_raster_done, 0));  // (it reads the positions this replaces)
  std::vector<StageCopy> cp = {{&ctx->pos, g->pos, 2 * fV}, {&ctx->src, g->src, fE}, {&ctx->dst, g->dst, fE}};
  cp.insert(cp.end(), extra, extra + n_extra);
  rc = staged_h2d(ctx, cp.data(), cp.size());
  if (rc) return rc;
  char* const sc = static_cast<char*>(ctx->topo_scratch.p);
  TopoBuild t;
  t.V = V, t.E = E;
  t.pos = (const float2*)ctx->pos.p;
  t.minx = minx, t.miny = miny;
  t.sx = (maxx > minx) ? 65535.0f / (maxx - minx) : 0.0f, t.sy = (maxy > miny) ? 65535.0f / (maxy - miny) : 0.0f;
  bind_scratch(&t, sc, sm);
  t.src = (int32_t*)ctx->src.p, t.dst = (int32_t*)ctx->dst.p, t.row_ptr = (int32_t*)ctx->row_ptr.p, t.half = (uint32_t*)ctx->half.p;
  t.order_m = (int32_t*)ctx->order_m.p, t.rid_of = (int32_t*)ctx->rid_of.p, t.perm = (int32_t*)ctx->perm.p, t.iperm = (int32_t*)ctx->iperm.p;
  t.pdeg = (int32_t*)ctx->pdeg.p, t.slice_row = (int32_t*)ctx->slice_row.p;
  t.wg_info = (int32_t*)ctx->wg_info.p, t.wg_v0 = (int32_t*)ctx->wg_v0.p, t.wg_vfirst = (uint8_t*)ctx->wg_vfirst.p;
  t.wg2_info = (int32_t*)ctx->wg2_info.p, t.wg2_vfirst = (uint8_t*)ctx->wg2_vfirst.p;
  t.dims = (TopoDims*)ctx->topo_dims.p;
  LAUNCHCHK(ctx, launch_topo_upload_front(t, ctx->stream));
  LAUNCHCHK(ctx, launch_topo_back(t, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_dims, ctx->topo_dims.p, sizeof(TopoDims), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));
  const TopoDims dm = *ctx->h_dims;
  if (dm.flags != 0 || (ctx->opt_persistent != 4 && dm.wg_count > kPvDensePerCu * cus)) return 0;  // the host builders take over
  PackedLayout& L = ctx->L;
  L.V = V, L.E = E, L.n_slices = sm.n_slices, L.rows = dm.rows, L.max_degree = dm.max_degree;
  L.wg_ok = true, L.wg_rowpack = true, L.wg_count = dm.wg_count, L.wg_lcap = dm.wg_lcap, L.wg_slab_slots = 0, L.n_rec = V;
  L.wg2_walked = true, L.wg2_ok = dm.max_degree <= 32 && dm.wg2_count > 0, L.wg2_count = dm.wg2_count, L.wg2_lcap = dm.wg2_lcap;
  L.tv_ok = false, L.tv_waves = 0;
  ctx->host_layout_valid = false;
  const bool want_e2 = wants_e2(ctx) && L.wg2_ok && L.wg2_count <= kPv2WavesPerCu * cus * 4;
  rc = topology_buffers(ctx, want_e2, 4 * (size_t)L.wg_count, (size_t)L.wg_count, (size_t)V, 4 * (size_t)L.wg2_count, (size_t)V);
  if (rc) return rc;
  std::vector<StageFill> fills;
  topology_fills(ctx, want_e2, &fills);
  rc = staged_h2d(ctx, nullptr, 0, fills.data(), fills.size());
  if (rc) return rc;
  rc = topology_expand(ctx, want_e2);
  if (rc) return rc;
  ctx->feat_dev_valid = false;  // (the device's feature table describes the previous graph)
  HIPCHK(ctx, hipEventRecord(ctx->ev_topo_ready, ctx->stream));
  *done = true;
  return 0;
}

int topo_commit(flame_nltgv2_ctx* ctx, bool* done) {
  *done = false;
  using C = flame_nltgv2_ctx;
  C::PreparedSync& P = ctx->prepared;
  if (!P.active || !P.device || P.topo != ctx->topo) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  hipStream_t ts = ctx->topo_stream;
  const bool trace = std::getenv("FLAME_NLTGV2_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  HIPCHK(ctx, hipStreamSynchronize(ctx->topo_stream));  // the solver is still iterating on the context's stream meanwhile
  const auto t1 = std::chrono::steady_clock::now();
  P.active = false;
  const int32_t V = P.V, E = P.E;
  const TopoDims dm = *ctx->h_dims;
  const int cus = ctx->prop.multiProcessorCount;
  const bool rowpack = ctx->opt_persistent == 4 || (static_cast<int64_t>(2) * E + V / 32) / 54 + 1 <= (int64_t)kPvDensePerCu * cus;
  if (dm.flags != 0 || dm.n_edges != E || !rowpack || (ctx->opt_persistent != 4 && dm.wg_count > kPvDensePerCu * cus)) {
    // A case for the host builders.  The live graph is whole -- its topology and state were only read -- except for the feature
    // table, which has moved on: it is rebuilt from the live graph's ids when the device path is next taken.
    if (trace) std::fprintf(stderr, "[flame_nltgv2] sync_graph: device build declined (flags %d, edges %d of %d, patches %d), host path\n", dm.flags, dm.n_edges, E, dm.wg_count);
    return 0;
  }
  // ---- the next topology's expansion: into the spare tables, on the side stream, while the solver's last rounds still run ---------------
  // (round 6: these six kernels and their launches stood between the settled solver and its next round, ~60 us of a 130 us stop)
  if (std::getenv("FLAME_NLTGV2_LATE_EXPAND")) {  // (for comparison: the solver is settled first, as through round 5)
    const int rc0 = ensure_canon(ctx);
    if (rc0) return rc0;
  }
  const int n_slices = (V + kWave - 1) / kWave;
  PackedLayout Ln;  // (scalars only: the tables stand on the device)
  Ln.V = V, Ln.E = E, Ln.n_slices = n_slices, Ln.rows = dm.rows, Ln.max_degree = dm.max_degree;
  Ln.wg_ok = true, Ln.wg_rowpack = true, Ln.wg_count = dm.wg_count, Ln.wg_lcap = dm.wg_lcap, Ln.wg_slab_slots = 0, Ln.n_rec = V;
  Ln.wg2_walked = true, Ln.wg2_ok = dm.max_degree <= 32 && dm.wg2_count > 0, Ln.wg2_count = dm.wg2_count, Ln.wg2_lcap = dm.wg2_lcap;
  Ln.tv_ok = false, Ln.tv_waves = 0;
  const bool want_e2 = wants_e2(ctx, Ln) && Ln.wg2_ok && Ln.wg2_count <= kPv2WavesPerCu * cus * 4;
  const bool placed = placement_applies(ctx, Ln);
  int rc = 0;
  {
    const size_t n_slots = (size_t)(Ln.rows + kRowPad) * kWave, lanes = (size_t)Ln.wg_count * kWave, lanes2 = (size_t)Ln.wg2_count * kWave;
    if (n_slots > (size_t)0x7fffffff) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
    const size_t stride = records_capacity(Ln);
    struct { int i; size_t bytes; } req[] = {
        {C::EX_REC_NBR, sizeof(uint32_t) * n_slots}, {C::EX_REC_EDGE, sizeof(int32_t) * n_slots}, {C::EX_EDGE_SRC_SLOT, sizeof(int32_t) * (size_t)E},
        {C::EX_WG_SLOT, sizeof(int32_t) * lanes}, {C::EX_WG_VID, sizeof(int32_t) * lanes}, {C::EX_WG_META, sizeof(uint32_t) * lanes},
        {C::EX_WG_NBR, sizeof(int32_t) * lanes}, {C::EX_WG_FETCH, sizeof(int32_t) * lanes},
        {C::EX_WG2_SLOT, want_e2 ? sizeof(int32_t) * 2 * lanes2 : 0}, {C::EX_WG2_NBR, want_e2 ? sizeof(int32_t) * 2 * lanes2 : 0},
        {C::EX_WG2_VID, want_e2 ? sizeof(int32_t) * lanes2 : 0}, {C::EX_WG2_META, want_e2 ? sizeof(uint32_t) * lanes2 : 0},
        {C::EX_WG2_FETCH, want_e2 ? sizeof(int32_t) * lanes2 : 0}, {C::EX_WG2_RMAX, want_e2 ? (size_t)64 : 0},
        {C::EX_PLACE_REC_OFF, placed ? sizeof(int32_t) * 2 * stride : 0}};
    for (auto& r : req)
      if (r.bytes && (rc = ensure(ctx, ctx->ex[r.i], r.bytes)) != 0) return rc;
    if (placed) {
      rc = ensure(ctx, ctx->place_patch_nx, sizeof(int32_t) * stride + stride);
      if (!rc && ctx->place_fill_nx.cap == 0) {
        rc = ensure(ctx, ctx->place_fill_nx, sizeof(int) * (2 * kPlacePages + 16 + 128));
        if (!rc) HIPCHK(ctx, hipMemsetAsync(ctx->place_fill_nx.p, 0, sizeof(int) * (2 * kPlacePages + 16 + 128), ts));
      }
      if (rc) return rc;
    }
    ExpandTables t;
    t.c = ctx->c, t.f = ctx->f;  // (every pointer the expansion touches is replaced below; the rest is not read)
    t.c.V = V, t.c.E = E;
    t.c.src = (int32_t*)ctx->nx[C::NX_SRC].p, t.c.dst = (int32_t*)ctx->nx[C::NX_DST].p;
    t.c.row_ptr = (int32_t*)ctx->nx[C::NX_ROW_PTR].p, t.c.half = (uint32_t*)ctx->nx[C::NX_HALF].p;
    t.f.n_slices = n_slices, t.f.n_slots = (int)n_slots;
    t.f.slice_row = (int32_t*)ctx->nx[C::NX_SLICE_ROW].p, t.f.perm = (int32_t*)ctx->nx[C::NX_PERM].p, t.f.pdeg = (int32_t*)ctx->nx[C::NX_PDEG].p;
    t.f.rec_nbr = (uint32_t*)ctx->ex[C::EX_REC_NBR].p, t.f.rec_edge = (int32_t*)ctx->ex[C::EX_REC_EDGE].p;
    t.f.edge_src_slot = (int32_t*)ctx->ex[C::EX_EDGE_SRC_SLOT].p;
    t.f.wg_count = Ln.wg_count, t.f.n_rec = Ln.n_rec, t.f.wg_lcap = Ln.wg_lcap, t.f.wg_slab_slots = 0, t.f.wg_rowpack = 1;
    t.f.wg_slot = (int32_t*)ctx->ex[C::EX_WG_SLOT].p, t.f.wg_vid = (int32_t*)ctx->ex[C::EX_WG_VID].p, t.f.wg_meta = (uint32_t*)ctx->ex[C::EX_WG_META].p;
    t.f.wg_nbr = (int32_t*)ctx->ex[C::EX_WG_NBR].p, t.f.wg_fetch = (int32_t*)ctx->ex[C::EX_WG_FETCH].p;
    t.f.wg_info = (int32_t*)ctx->nx[C::NX_WG_INFO].p;
    t.iperm = (const int32_t*)ctx->nx[C::NX_IPERM].p, t.wg_v0 = (const int32_t*)ctx->nx[C::NX_WG_V0].p;
    t.order_m = (const int32_t*)ctx->nx[C::NX_ORDER_M].p, t.rid_of = (const int32_t*)ctx->nx[C::NX_RID_OF].p;
    t.wg_vfirst = (const uint8_t*)ctx->nx[C::NX_WG_VFIRST].p;
    t.wg2_info = (int32_t*)ctx->nx[C::NX_WG2_INFO].p, t.wg2_vfirst = (const uint8_t*)ctx->nx[C::NX_WG2_VFIRST].p;
    t.wg2_slot = (int32_t*)ctx->ex[C::EX_WG2_SLOT].p, t.wg2_vid = (int32_t*)ctx->ex[C::EX_WG2_VID].p, t.wg2_nbr = (int32_t*)ctx->ex[C::EX_WG2_NBR].p;
    t.wg2_fetch = (int32_t*)ctx->ex[C::EX_WG2_FETCH].p, t.wg2_meta = (uint32_t*)ctx->ex[C::EX_WG2_META].p, t.wg2_rmax = (int*)ctx->ex[C::EX_WG2_RMAX].p;
    if (placed) {
      t.rec_off = (int32_t*)ctx->ex[C::EX_PLACE_REC_OFF].p, t.place_patch = (int32_t*)ctx->place_patch_nx.p, t.place_fill = (int*)ctx->place_fill_nx.p;
      t.per_xcd = (Ln.wg_count + 7) / 8, t.stride = stride;
    }
    std::vector<StageFill> fills;
    expansion_fills(Ln, t.f.rec_edge, t.f.rec_nbr, want_e2 ? t.wg2_rmax : nullptr, &fills);
    rc = staged_h2d(ctx, nullptr, 0, fills.data(), fills.size(), 0, ts);
    if (!rc) rc = expand_launches(ctx, Ln, t, want_e2, ts);
    if (rc) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->ev_expanded, ts));
  }
  // ---- the state gather: old state -> the new graph's arrays through the index maps, new vertices initialised.  It reads the canonical
  // arrays of the OLD graph and the per-vertex tables of the new one and writes spare arrays only: it can be redone, so it goes out
  // behind the rounds in flight and the unpack of their state (ensure_canon's `behind`), before the host has seen how they ended.
  const size_t fV = sizeof(float) * (size_t)V, fE = sizeof(float) * (size_t)E;
  DevBuf* cur_v[9] = {&ctx->x, &ctx->w1, &ctx->w2, &ctx->xb, &ctx->w1b, &ctx->w2b, &ctx->xp, &ctx->w1p, &ctx->w2p};
  DevBuf* cur_q[3] = {&ctx->q1, &ctx->q2, &ctx->q3};
  for (int i = 0; i < 9 && !rc; ++i) rc = ensure(ctx, ctx->sp_v[i], fV);  // (spares: nothing in flight uses them)
  for (int i = 0; i < 3 && !rc; ++i) rc = ensure(ctx, ctx->sp_q[i], fE);
  for (int i = 0; i < 2 && !rc; ++i) rc = ensure(ctx, ctx->sp_ab[i], fE);
  if (!rc) rc = ensure(ctx, ctx->sync_need, (size_t)V);
  if (rc) return rc;
  SyncArgs sa;
  sa.V = V, sa.E = E;
  sa.old_of_new = (const int32_t*)ctx->prep_vmap, sa.old_of_new_edge = (const int32_t*)ctx->prep_emap;
  sa.data = (const float*)ctx->nx[C::NX_DATA].p, sa.weight = (const float*)ctx->nx[C::NX_WEIGHT].p;
  sa.init_x = P.has_init ? (const float*)ctx->prep_init : nullptr;
  set_init_map(ctx, P.init_from_map != 0, &sa);
  if (P.init_from_map != 0 && sa.map_rows == 0) sa.init_map = sa.data;  // (no resident map: set_init_map named the LIVE data terms)
  sa.check_sticky = P.check_sticky, sa.sticky_threshold = P.sticky_threshold;
  sa.graph_scale = P.init_graph_scale;
  for (int i = 0; i < 9; ++i) sa.o[i] = (const float*)cur_v[i]->p, sa.n[i] = (float*)ctx->sp_v[i].p;
  for (int i = 0; i < 3; ++i) sa.oq[i] = (const float*)cur_q[i]->p, sa.nq[i] = (float*)ctx->sp_q[i].p;
  sa.src = (const int32_t*)ctx->nx[C::NX_SRC].p, sa.dst = (const int32_t*)ctx->nx[C::NX_DST].p, sa.row_ptr = (const int32_t*)ctx->nx[C::NX_ROW_PTR].p;
  sa.half = (const uint32_t*)ctx->nx[C::NX_HALF].p, sa.pos = (const float2*)ctx->nx[C::NX_POS].p;
  sa.alpha = (float*)ctx->sp_ab[0].p, sa.beta = (float*)ctx->sp_ab[1].p, sa.need_nbr = (uint8_t*)ctx->sync_need.p;
  const std::function<int()> gather = [&]() -> int {
    LAUNCHCHK(ctx, launch_sync_state(sa, ctx->stream));
    return 0;
  };
  const auto t1b = std::chrono::steady_clock::now();
  // ---- the solver stops here: the chain of runs is settled, the state goes to its canonical arrays ---------------------------------------
  int behind = kBehindNotLaunched;
  rc = ensure_canon(ctx, &gather, &behind);
  if (rc) return rc;  // (what the gather wrote, if it went out: spare arrays)
  if (behind != kBehindDone) {  // (nothing was in flight, or the rounds expired and the state has been unpacked again)
    rc = gather();
    if (rc) return rc;
  }
  const auto t2 = std::chrono::steady_clock::now();
  ctx->have_graph = false;  // (until the new graph stands)
  PackedLayout& L = ctx->L;
  L.V = V, L.E = E, L.n_slices = n_slices, L.rows = dm.rows, L.max_degree = dm.max_degree;
  L.wg_ok = true, L.wg_rowpack = true, L.wg_count = dm.wg_count, L.wg_lcap = dm.wg_lcap, L.wg_slab_slots = 0, L.n_rec = V;
  L.wg2_walked = true, L.wg2_ok = Ln.wg2_ok, L.wg2_count = dm.wg2_count, L.wg2_lcap = dm.wg2_lcap;
  L.tv_ok = false, L.tv_waves = 0;
  ctx->host_layout_valid = false;
  for (int i = 0; i < C::NX_COUNT; ++i) std::swap(*ctx->nx_live(i), ctx->nx[i]);  // the next topology becomes the live one
  for (int i = 0; i < C::EX_COUNT; ++i) std::swap(*ctx->ex_live(i), ctx->ex[i]);  // ... with its expansion
  rc = topology_buffers(ctx, want_e2, 4 * (size_t)L.wg_count, (size_t)L.wg_count, (size_t)V, 4 * (size_t)L.wg2_count, (size_t)V);
  if (rc) return rc;
  std::vector<StageFill> fills;
  topology_fills(ctx, want_e2, &fills, /*early=*/true);
  rc = staged_h2d(ctx, nullptr, 0, fills.data(), fills.size());
  if (rc) return rc;
  rc = topology_expand(ctx, want_e2, /*launched=*/true);
  if (rc) return rc;
  if (placed) ctx->place_topo = ctx->topo, ctx->place_per_xcd = (L.wg_count + 7) / 8;
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_expanded, 0));  // (what follows reads the expansion)
  for (int i = 0; i < 9; ++i) std::swap(*cur_v[i], ctx->sp_v[i]);
  for (int i = 0; i < 3; ++i) std::swap(*cur_q[i], ctx->sp_q[i]);
  std::swap(ctx->alpha, ctx->sp_ab[0]), std::swap(ctx->beta, ctx->sp_ab[1]);
  refresh_args(ctx);
  ctx->static_stale = true;  // (the records of the packed form: with the state, in the one launch of the next run's ensure_fused)
  HIPCHK(ctx, hipEventRecord(ctx->ev_topo_ready, ctx->stream));
  ctx->feat_gen += 1, ctx->feat_dev_valid = true;
  const int32_t* const fid = reinterpret_cast<const int32_t*>(static_cast<const char*>(ctx->stage[1].h) + P.off[0]);
  ctx->h_feat.assign(fid, fid + V);
  ctx->feat_map_valid = false, ctx->feat_tab_valid = false;  // (the host path's maps describe an earlier graph)
  ctx->canon_valid = true, ctx->fused_valid = false, ctx->have_prev = false, ctx->parity = 0;
  ctx->have_graph = true, ctx->last_error = 0, ctx->last_sync_path = 2;
  *done = true;
  if (trace) {
    const auto t2b = std::chrono::steady_clock::now();
    HIPCHK(ctx, wait_solver_stream(ctx));
    const auto t3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[flame_nltgv2] sync_graph (device): prepare (checks, staging, builder enqueued) %.3f ms; commit: builder awaited %.3f, "
                 "expansion enqueued beside the solver %.3f, solver settled (unpack + state gather behind it) %.3f, swap + clears enqueued %.3f, device done %.3f ms later "
                 "(V=%d E=%d: %d kept, %d patches, %d two-half-edge patches)\n",
                 ms(P.t_begin, P.t_enqueued), ms(t0, t1), ms(t1, t1b), ms(t1b, t2), ms(t2, t2b), ms(t2b, t3), V, E, dm.n_keep, dm.wg_count, dm.wg2_count);
  }
  return 0;
}

}  // namespace host
}  // namespace flame_hip
