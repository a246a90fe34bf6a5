// nltgv2_topo_capi.hip -- host side of the device-built topology (nltgv2_topo.hip): the per-frame warm-start sync
// (flame_nltgv2_sync_graph; Flame::syncGraph's graph edits, /root/reference/src/flame/flame.cc:1985-2121) without host index maps and
// without host layout tables, and the host image of a device-built topology on demand.
//
// What the host still does per frame: check the inputs the way the C-ABI promises (ids unique and >= 0, edges in range, positions
// finite: three streaming passes), stage the frame's arrays into ONE pinned blob, enqueue the builder, read 64 bytes of dimensions
// back (rows, patches, largest degree), size the slot / lane arrays and enqueue their expansion and the state gather.
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

}  // namespace

// Brings the host image of the current topology up to date after a device-side build: the edge list comes down, the host
// builders redo the tables (they are the reference the device tables are compared with: flame_nltgv2_layout_selftest).
int ensure_host_layout(flame_nltgv2_ctx* ctx) {
  if (ctx->host_layout_valid) return 0;
  const int32_t V = ctx->L.V, E = ctx->L.E;
  std::vector<float> pos(2 * (size_t)V);
  ctx->h_src.resize((size_t)E), ctx->h_dst.resize((size_t)E);
  if (V) HIPCHK(ctx, hipMemcpyAsync(pos.data(), ctx->pos.p, sizeof(float) * pos.size(), hipMemcpyDeviceToHost, ctx->stream));
  if (E) HIPCHK(ctx, hipMemcpyAsync(ctx->h_src.data(), ctx->src.p, sizeof(int32_t) * (size_t)E, hipMemcpyDeviceToHost, ctx->stream));
  if (E) HIPCHK(ctx, hipMemcpyAsync(ctx->h_dst.data(), ctx->dst.p, sizeof(int32_t) * (size_t)E, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  flame_nltgv2_graph g{};
  g.V = V, g.E = E, g.pos = pos.data(), g.src = ctx->h_src.data(), g.dst = ctx->h_dst.data();
  const PackedLayout dev = ctx->L;  // (scalars as the device reported them; its vectors are stale)
  int rc = build_layout(&g, &ctx->L, /*host_expand=*/false, /*rowpack=*/dev.wg_rowpack);
  if (rc) return fail(ctx, rc);
  if (dev.wg2_walked) build_patch_walk2(&ctx->L);
  PackedLayout& L = ctx->L;
  L.tv_ok = dev.tv_ok, L.tv_waves = dev.tv_waves;
  const bool same = L.rows == dev.rows && L.n_slices == dev.n_slices && L.max_degree == dev.max_degree && L.wg_count == dev.wg_count &&
                    L.wg_lcap == dev.wg_lcap && L.wg_rowpack == dev.wg_rowpack &&
                    (!dev.wg2_walked || (L.wg2_count == dev.wg2_count && L.wg2_lcap == dev.wg2_lcap));
  if (!same) return fail(ctx, FLAME_NLTGV2_ERR_HIP);  // the device tables and the host builders disagree: a bug, never silent
  ctx->host_layout_valid = true;
  return 0;
}

// The per-frame sync with the topology built on the device.  *done = false: the case is one the device path does not take
// (the caller goes on with the host path; nothing has been changed).  Preconditions checked here: the caller vouches for a
// duplicate-free edge list (edges_unique), ids fit the direct table, at least one edge.
int sync_graph_device(flame_nltgv2_ctx* ctx, const flame_nltgv2_sync_input* in, bool* done) {
  *done = false;
  const int32_t V = in->V, E = in->E;
  const int32_t Vo = ctx->L.V, Eo = ctx->L.E;
  if (!in->edges_unique || V < 2 || E < 1 || V > (1 << 22) || (int64_t)Eo + E + V + 1 > 0x7fff0000ll) return 0;
  const bool trace = std::getenv("FLAME_NLTGV2_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  // ---- the checks the C-ABI promises (INVALID_ARG before anything is changed) -------------------------------------------------
  int32_t max_id = -1;
  for (int32_t v = 0; v < V; ++v) {
    if (in->feat_id[v] < 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
    max_id = std::max(max_id, in->feat_id[v]);
  }
  for (int32_t id : ctx->h_feat) max_id = std::max(max_id, id);
  if (max_id >= kFeatDirectMax) return 0;
  if (ctx->feat_stamp.size() <= (size_t)max_id) ctx->feat_stamp.resize((size_t)max_id + 1 + (size_t)max_id / 2, 0u);
  {
    const uint32_t stamp = ++ctx->feat_stamp_now;
    if (stamp == 0u) std::fill(ctx->feat_stamp.begin(), ctx->feat_stamp.end(), 0u), ctx->feat_stamp_now = 1u;
    for (int32_t v = 0; v < V; ++v) {
      uint32_t& st = ctx->feat_stamp[(size_t)in->feat_id[v]];
      if (st == ctx->feat_stamp_now) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);  // duplicate id
      st = ctx->feat_stamp_now;
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
  const auto t1 = std::chrono::steady_clock::now();

  // ---- previous state to its canonical arrays (a kernel, enqueued); buffers ------------------------------------------------------
  int rc = ensure_canon(ctx);
  if (rc) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // nothing in flight uses buffers that may be reallocated below (or the staging blob)
  const int n_slices = (V + kWave - 1) / kWave;
  const size_t n_packed = (size_t)n_slices * kWave;
  const size_t fV = sizeof(float) * (size_t)V, fE = sizeof(float) * (size_t)E, iV = sizeof(int32_t) * (size_t)V;
  const size_t vpad = ((size_t)V + kWalkPad - 1) / kWalkPad * kWalkPad;
  const int n_seg = (V + 255) / 256;
  const int n_scan = Eo + E + V + 1;
  const size_t sort_bytes = topo_sort_temp_bytes(V, n_scan);
  Carve cv;
  const size_t o_fid = cv.take(iV), o_edges = cv.take(2 * fE), o_old_edge = cv.take(fE), o_first_k = cv.take(sizeof(int32_t) * (size_t)std::max(Eo, 1));
  int cc_bits = 1;
  while ((1 << cc_bits) < V) ++cc_bits;
  const size_t iM = sizeof(int32_t) << cc_bits;
  const size_t o_scan = cv.take(sizeof(int32_t) * (size_t)n_scan), o_deg = cv.take(iV), o_cur = cv.take(iV), o_parent = cv.take(iM), o_minid = cv.take(iM), o_morton = cv.take(iV);
  const size_t o_key_out = cv.take(8 * (size_t)V), o_width = cv.take(sizeof(int32_t) * (size_t)n_slices);
  const size_t o_wflag = cv.take(vpad), o_vf0 = cv.take(vpad), o_vf1 = cv.take(vpad), o_seg0 = cv.take(sizeof(int32_t) * (size_t)n_seg), o_seg1 = cv.take(sizeof(int32_t) * (size_t)n_seg);
  const size_t o_counters = cv.take(64), o_wg2_v0 = cv.take(iV), o_sort = cv.take(sort_bytes);
  rc = ensure(ctx, ctx->topo_scratch, cv.off);
  DevBuf* cur_v[9] = {&ctx->x, &ctx->w1, &ctx->w2, &ctx->xb, &ctx->w1b, &ctx->w2b, &ctx->xp, &ctx->w1p, &ctx->w2p};
  DevBuf* cur_q[3] = {&ctx->q1, &ctx->q2, &ctx->q3};
  for (int i = 0; i < 9 && !rc; ++i) rc = ensure(ctx, ctx->sp_v[i], fV);
  for (int i = 0; i < 3 && !rc; ++i) rc = ensure(ctx, ctx->sp_q[i], fE);
  struct { DevBuf* b; size_t bytes; } req[] = {
      {&ctx->data, fV}, {&ctx->weight, fV}, {&ctx->alpha, fE}, {&ctx->beta, fE}, {&ctx->sync_init, fV}, {&ctx->sync_vmap, iV},
      {&ctx->sync_emap, fE}, {&ctx->sync_need, (size_t)V}, {&ctx->nx_pos, 2 * fV}, {&ctx->nx_src, fE}, {&ctx->nx_dst, fE},
      {&ctx->nx_row_ptr, sizeof(int32_t) * ((size_t)V + 1)}, {&ctx->nx_half, 2 * fE}, {&ctx->topo_dims, sizeof(TopoDims)},
      // the tables the builder writes, sized by their upper bounds (a patch holds at least one vertex)
      {&ctx->slice_row, sizeof(int32_t) * ((size_t)n_slices + 1)}, {&ctx->perm, sizeof(int32_t) * n_packed}, {&ctx->pdeg, sizeof(int32_t) * n_packed},
      {&ctx->iperm, iV}, {&ctx->order_m, iV}, {&ctx->rid_of, iV}, {&ctx->wg_info, 4 * iV}, {&ctx->wg_v0, iV}, {&ctx->wg_vfirst, vpad},
      {&ctx->wg2_info, 4 * iV}, {&ctx->wg2_vfirst, vpad}};
  for (auto& r : req)
    if (!rc) rc = ensure(ctx, *r.b, r.bytes);
  if (rc) return rc;
  // the device's feature table follows the ids (grown: its content is rebuilt from the current graph's ids)
  if (ctx->feat_tab_size_d <= max_id) {
    const size_t want = (size_t)max_id + 1 + (size_t)max_id / 2;
    rc = ensure(ctx, ctx->feat_stamp_d, sizeof(uint32_t) * want);
    if (!rc) rc = ensure(ctx, ctx->feat_val_d, sizeof(int32_t) * want);
    if (rc) return rc;
    ctx->feat_tab_size_d = (int)std::min(ctx->feat_stamp_d.cap / sizeof(uint32_t), ctx->feat_val_d.cap / sizeof(int32_t));
    ctx->feat_dev_valid = false;
  }
  char* const sc = static_cast<char*>(ctx->topo_scratch.p);
  if (!ctx->feat_dev_valid || ctx->feat_gen >= 0xfffffff0u) {
    HIPCHK(ctx, hipMemsetAsync(ctx->feat_stamp_d.p, 0, sizeof(uint32_t) * (size_t)ctx->feat_tab_size_d, ctx->stream));
    ctx->feat_gen = 1;
    if (Vo > 0) {  // (the previous graph's ids travel through the scratch area the frame's ids will use next; ordered on the stream)
      rc = ensure(ctx, ctx->sync_need, std::max((size_t)V, sizeof(int32_t) * (size_t)Vo));
      if (rc) return rc;
      HIPCHK(ctx, hipMemcpyAsync(ctx->sync_need.p, ctx->h_feat.data(), sizeof(int32_t) * (size_t)Vo, hipMemcpyHostToDevice, ctx->stream));
      LAUNCHCHK(ctx, launch_topo_feat_build((const int32_t*)ctx->sync_need.p, Vo, (uint32_t*)ctx->feat_stamp_d.p, (int32_t*)ctx->feat_val_d.p,
                                            ctx->feat_tab_size_d, ctx->feat_gen, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (h_feat is pageable and about to change)
    }
    ctx->feat_dev_valid = true;
  }

  // ---- one blob up: the frame's inputs ----------------------------------------------------------------------------------------------
  DevBuf b_fid{sc + o_fid, iV}, b_edges{sc + o_edges, 2 * fE};
  const StageCopy cp[] = {{&b_fid, in->feat_id, iV}, {&b_edges, in->edges, 2 * fE}, {&ctx->nx_pos, in->pos, 2 * fV},
                          {&ctx->data, in->data_term, fV}, {&ctx->weight, in->data_weight, fV},
                          {&ctx->sync_init, in->init_x, in->init_x ? fV : 0}};
  rc = staged_h2d(ctx, cp, sizeof(cp) / sizeof(cp[0]));
  if (rc) return rc;
  const auto t2 = std::chrono::steady_clock::now();

  // ---- the builder ------------------------------------------------------------------------------------------------------------------
  TopoBuild t;
  t.V = V, t.E = E, t.Vo = Vo, t.Eo = Eo, t.n_slices = n_slices;
  t.fid = (const int32_t*)(sc + o_fid), t.pos = (const float2*)ctx->nx_pos.p, t.tri_edges = (const int32_t*)(sc + o_edges);
  t.minx = minx, t.miny = miny;
  t.sx = (maxx > minx) ? 65535.0f / (maxx - minx) : 0.0f, t.sy = (maxy > miny) ? 65535.0f / (maxy - miny) : 0.0f;
  t.o_row_ptr = (const int32_t*)ctx->row_ptr.p, t.o_half = (const uint32_t*)ctx->half.p;
  t.o_src = (const int32_t*)ctx->src.p, t.o_dst = (const int32_t*)ctx->dst.p;
  t.feat_stamp = (uint32_t*)ctx->feat_stamp_d.p, t.feat_val = (int32_t*)ctx->feat_val_d.p, t.tab_size = ctx->feat_tab_size_d;
  t.gen_prev = ctx->feat_gen, t.gen_new = ctx->feat_gen + 1;
  t.old_edge = (int32_t*)(sc + o_old_edge), t.first_k = (int32_t*)(sc + o_first_k), t.scan = (int32_t*)(sc + o_scan);
  t.deg = (int32_t*)(sc + o_deg), t.cur = (int32_t*)(sc + o_cur), t.parent = (int32_t*)(sc + o_parent), t.morton = (uint32_t*)(sc + o_morton);
  t.cc_bits = cc_bits, t.minid = (int32_t*)(sc + o_minid);
  t.key_out = (uint64_t*)(sc + o_key_out), t.width = (int32_t*)(sc + o_width);
  t.wflag = (uint8_t*)(sc + o_wflag), t.vf[0] = (uint8_t*)(sc + o_vf0), t.vf[1] = (uint8_t*)(sc + o_vf1);
  t.seg_count[0] = (int32_t*)(sc + o_seg0), t.seg_count[1] = (int32_t*)(sc + o_seg1);
  t.sort_tmp = sc + o_sort, t.sort_tmp_bytes = sort_bytes, t.counters = (int*)(sc + o_counters);
  t.old_of_new = (int32_t*)ctx->sync_vmap.p, t.old_of_new_edge = (int32_t*)ctx->sync_emap.p;
  t.src = (int32_t*)ctx->nx_src.p, t.dst = (int32_t*)ctx->nx_dst.p, t.row_ptr = (int32_t*)ctx->nx_row_ptr.p, t.half = (uint32_t*)ctx->nx_half.p;
  t.order_m = (int32_t*)ctx->order_m.p, t.rid_of = (int32_t*)ctx->rid_of.p, t.perm = (int32_t*)ctx->perm.p, t.iperm = (int32_t*)ctx->iperm.p;
  t.pdeg = (int32_t*)ctx->pdeg.p, t.slice_row = (int32_t*)ctx->slice_row.p;
  t.wg_info = (int32_t*)ctx->wg_info.p, t.wg_v0 = (int32_t*)ctx->wg_v0.p, t.wg_vfirst = (uint8_t*)ctx->wg_vfirst.p;
  t.wg2_info = (int32_t*)ctx->wg2_info.p, t.wg2_v0 = (int32_t*)(sc + o_wg2_v0), t.wg2_vfirst = (uint8_t*)ctx->wg2_vfirst.p;
  t.dims = (TopoDims*)ctx->topo_dims.p;
  ctx->have_graph = false;  // (until the new graph stands)
  ctx->feat_dev_valid = false;  // (the table is being moved on to the new graph: valid again once that graph stands)
  LAUNCHCHK(ctx, launch_topo_sync_front(t, ctx->stream));
  LAUNCHCHK(ctx, launch_topo_back(t, ctx->stream));
  if (!ctx->h_dims && hipHostMalloc((void**)&ctx->h_dims, sizeof(TopoDims), hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return fail(ctx, FLAME_NLTGV2_ERR_OOM);
  }
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_dims, ctx->topo_dims.p, sizeof(TopoDims), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const auto t3 = std::chrono::steady_clock::now();
  const TopoDims dm = *ctx->h_dims;
  const int cus = ctx->prop.multiProcessorCount;
  const bool rowpack = ctx->opt_persistent == 4 || (static_cast<int64_t>(2) * E + V / 32) / 54 + 1 <= (int64_t)kPvDensePerCu * cus;
  if (dm.flags != 0 || dm.n_edges != E || !rowpack || (ctx->opt_persistent != 4 && dm.wg_count > kPvDensePerCu * cus)) {
    // A case for the host builders (a hub of more than 64 edges, a pair listed twice, a graph beyond the row-packed patch form).
    // The previous graph is still whole -- its topology and state were only read -- except for the feature table, which has moved
    // on: it is rebuilt from the previous graph's ids when the device path is next taken.
    ctx->have_graph = true;
    if (trace) std::fprintf(stderr, "[flame_nltgv2] sync_graph: device build declined (flags %d, edges %d of %d, patches %d), host path\n", dm.flags, dm.n_edges, E, dm.wg_count);
    return 0;
  }
  // ---- the new graph stands: sizes, slot / lane arrays, state ------------------------------------------------------------------------
  PackedLayout& L = ctx->L;
  L.V = V, L.E = E, L.n_slices = n_slices, L.rows = dm.rows, L.max_degree = dm.max_degree;
  L.wg_ok = true, L.wg_rowpack = true, L.wg_count = dm.wg_count, L.wg_lcap = dm.wg_lcap, L.wg_slab_slots = 0, L.n_rec = V;
  L.wg2_walked = true, L.wg2_ok = dm.max_degree <= 32 && dm.wg2_count > 0, L.wg2_count = dm.wg2_count, L.wg2_lcap = dm.wg2_lcap;
  L.tv_ok = false, L.tv_waves = 0;
  ctx->host_layout_valid = false;
  bool want_e2 = wants_e2(ctx) && L.wg2_ok && L.wg2_count <= kPv2WavesPerCu * cus * 4;
  std::swap(ctx->pos, ctx->nx_pos), std::swap(ctx->src, ctx->nx_src), std::swap(ctx->dst, ctx->nx_dst), std::swap(ctx->row_ptr, ctx->nx_row_ptr), std::swap(ctx->half, ctx->nx_half);
  rc = topology_buffers(ctx, want_e2, 4 * (size_t)L.wg_count, (size_t)L.wg_count, (size_t)V, 4 * (size_t)L.wg2_count, (size_t)V);
  if (rc) return rc;
  std::vector<StageFill> fills;
  topology_fills(ctx, want_e2, &fills);
  rc = staged_h2d(ctx, nullptr, 0, fills.data(), fills.size());
  if (rc) return rc;
  rc = topology_expand(ctx, want_e2);
  if (rc) return rc;
  SyncArgs sa;
  sa.V = V, sa.E = E;
  sa.old_of_new = (const int32_t*)ctx->sync_vmap.p, sa.old_of_new_edge = (const int32_t*)ctx->sync_emap.p;
  sa.data = (const float*)ctx->data.p, sa.weight = (const float*)ctx->weight.p;
  sa.init_x = in->init_x ? (const float*)ctx->sync_init.p : nullptr;
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
  ctx->feat_gen += 1, ctx->feat_dev_valid = true;
  ctx->h_feat.assign(in->feat_id, in->feat_id + V);
  ctx->feat_map_valid = false, ctx->feat_tab_valid = false;  // (the host path's maps describe an earlier graph)
  ctx->canon_valid = true, ctx->fused_valid = false, ctx->have_prev = false, ctx->parity = 0;
  ctx->have_graph = true, ctx->last_error = 0, ctx->last_sync_path = 2;
  *done = true;
  if (trace) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const auto t4 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[flame_nltgv2] sync_graph (device): checks %.3f, settle + buffers + staging %.3f, builder until the dimensions are back %.3f, "
                 "expansion + state gather %.3f ms (V=%d E=%d: %d kept, %d patches, %d two-half-edge patches)\n",
                 ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), V, E, dm.n_keep, dm.wg_count, dm.wg2_count);
  }
  return 0;
}

}  // namespace host
}  // namespace flame_hip
