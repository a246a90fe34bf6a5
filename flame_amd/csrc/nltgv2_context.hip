// nltgv2_context.hip -- solver context: buffers, uploads of a topology, life cycle, options, info, self-tests (see nltgv2_context.hpp).
#include "nltgv2_context.hpp"

namespace flame_hip {
namespace host {

int fail(flame_nltgv2_ctx* ctx, int status) {
  if (ctx) ctx->last_error = status;
  return status;
}

// An open run in flight ends when it is asked to.  Whoever is about to wait for the device without going through finish() -- a buffer
// that grows: hipFree and the allocators wait for every stream -- asks first (the request is sent once; finish() does the checking later).
void request_open_stop(flame_nltgv2_ctx* ctx) {
  if (!ctx->open_inflight || ctx->open_stop_sent || !ctx->stop_dev.p || !ctx->ctl_stream) return;
  *ctx->h_stop = ctx->open_tag0;  // (the run it is for; a copy of an earlier request still under way would carry this one early: the same request)
  if (hipMemcpyAsync(ctx->stop_dev.p, ctx->h_stop, sizeof(unsigned), hipMemcpyHostToDevice, ctx->ctl_stream) != hipSuccess) (void)hipGetLastError();
  ctx->open_stop_sent = true;
}

// The host waits for the solver's stream: an open run in flight is asked to stop first (it would go on to its bound).
hipError_t wait_solver_stream(flame_nltgv2_ctx* ctx) {
  request_open_stop(ctx);
  return hipStreamSynchronize(ctx->stream);
}

int ensure(flame_nltgv2_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.cap >= bytes) return 0;
  request_open_stop(ctx);
  size_t want = bytes + bytes / 2;  // geometric growth, reused across frames
  want = (want + 255) & ~size_t(255);
  if (b.p) {
    HIPCHK(ctx, hipFree(b.p));
    ctx->device_bytes -= b.cap;
    b.p = nullptr, b.cap = 0;
  }
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    ctx->last_hip = (int)e;
    b.p = nullptr;
    return fail(ctx, e == hipErrorOutOfMemory ? FLAME_NLTGV2_ERR_OOM : FLAME_NLTGV2_ERR_HIP);
  }
  b.cap = want;
  ctx->device_bytes += want;
  return 0;
}

void drop_graphs(flame_nltgv2_ctx* ctx) {
  for (auto& g : ctx->graphs)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  ctx->graphs.clear();
}

SolverParams to_sp(const flame_nltgv2_params* p) {
  return SolverParams{p->data_factor, p->step_x, p->step_q, p->theta, p->x_min, p->x_max};
}

int enter(flame_nltgv2_ctx* ctx) {
  if (!ctx) return FLAME_NLTGV2_ERR_INVALID_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ++ctx->call_seq;
  return 0;
}

void refresh_args(flame_nltgv2_ctx* ctx) {
  CanonArgs& c = ctx->c;
  c.V = ctx->L.V, c.E = ctx->L.E;
  c.pos = (float2*)ctx->pos.p;
  c.x = (float*)ctx->x.p, c.w1 = (float*)ctx->w1.p, c.w2 = (float*)ctx->w2.p;
  c.xb = (float*)ctx->xb.p, c.w1b = (float*)ctx->w1b.p, c.w2b = (float*)ctx->w2b.p;
  c.xp = (float*)ctx->xp.p, c.w1p = (float*)ctx->w1p.p, c.w2p = (float*)ctx->w2p.p;
  c.data = (float*)ctx->data.p, c.weight = (float*)ctx->weight.p;
  c.src = (int32_t*)ctx->src.p, c.dst = (int32_t*)ctx->dst.p;
  c.alpha = (float*)ctx->alpha.p, c.beta = (float*)ctx->beta.p;
  c.q1 = (float*)ctx->q1.p, c.q2 = (float*)ctx->q2.p, c.q3 = (float*)ctx->q3.p;
  c.row_ptr = (int32_t*)ctx->row_ptr.p, c.half = (uint32_t*)ctx->half.p;
  c.err = (int*)ctx->err.p;
  FusedArgs& f = ctx->f;
  f.n_slices = ctx->L.n_slices;
  f.n_slots = (ctx->L.rows + kRowPad) * kWave;
  f.slice_row = (int32_t*)ctx->slice_row.p, f.perm = (int32_t*)ctx->perm.p, f.pdeg = (int32_t*)ctx->pdeg.p;
  f.rec_nbr = (uint32_t*)ctx->rec_nbr.p, f.rec_edge = (int32_t*)ctx->rec_edge.p;
  f.edge_src_slot = (int32_t*)ctx->edge_src_slot.p;
  f.hrec = (int4*)ctx->hrec.p, f.hq = (float4*)ctx->hq.p;
  f.vstate = (float4*)ctx->vstate.p, f.vaux = (float2*)ctx->vaux.p;
  f.hq_out = (float4*)ctx->hq_alt.p, f.vstate_out = (float4*)ctx->vstate_alt.p;
  f.bar[0] = (float4*)ctx->bar0.p, f.bar[1] = (float4*)ctx->bar1.p;
  f.vprev = (float4*)ctx->vprev.p;
  f.xbuf = ctx->xbuf.p;
  f.tv_waves = (ctx->tv_built && ctx->L.tv_ok) ? ctx->L.tv_waves : 0;
  f.tv_slot = (int32_t*)ctx->tv_slot.p, f.tv_vid = (int32_t*)ctx->tv_vid.p;
  f.tv_meta = (uint32_t*)ctx->tv_meta.p, f.tv_wave = (uint32_t*)ctx->tv_wave.p;
  f.wg_count = ctx->L.wg_ok ? ctx->L.wg_count : 0;
  f.n_rec = ctx->L.n_rec;
  f.wg_lcap = ctx->L.wg_lcap, f.wg_slab_slots = ctx->L.wg_slab_slots;
  f.wg_rowpack = ctx->L.wg_rowpack ? 1 : 0;
  f.wg_slot = (int32_t*)ctx->wg_slot.p, f.wg_vid = (int32_t*)ctx->wg_vid.p, f.wg_meta = (uint32_t*)ctx->wg_meta.p;
  f.wg_nbr = (int32_t*)ctx->wg_nbr.p, f.wg_fetch = (int32_t*)ctx->wg_fetch.p, f.wg_info = (int32_t*)ctx->wg_info.p;
  f.abort_flag = (int*)ctx->abort_flag.p;
  f.err = (int*)ctx->err.p;
}

int h2d(flame_nltgv2_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  HIPCHK(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

// records the exchange buffers hold: one per packed vertex slot (the he / tv forms index by packed vertex)
size_t records_capacity(const PackedLayout& L) {
  const size_t n_packed = (size_t)L.n_slices * kWave, n_rec = ((size_t)L.n_rec + kWave - 1) / kWave * kWave;
  return std::max(n_packed, n_rec);
}

int wait_raster(flame_nltgv2_ctx* ctx) {
  if (ctx->raster_inflight) {
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_raster_done, 0));
    ctx->raster_inflight = false;
  }
  return 0;
}

int ensure_canon(flame_nltgv2_ctx* ctx, const std::function<int()>* behind, int* behind_state) {
  bool unpacked = false, launched = false;
  if (behind_state) *behind_state = kBehindNotLaunched;
  if (ctx->pending.active) {  // a persistent run is still unchecked: settle it before anything reads or edits the state
    const int rc = finish(ctx, /*unpack_behind=*/true, &unpacked, behind, &launched);
    if (behind_state && launched) *behind_state = unpacked ? kBehindDone : kBehindSpoiled;
    if (rc) {
      if (unpacked) ctx->canon_valid = true;
      return rc;
    }
  }
  // interpolate_mesh_begin's side stream may still read the canonical pos / x: whoever comes through here is about to rewrite them (the
  // unpack below, or the caller: project_graph, rescale_data, update_data, upload_state) -- also when the arrays are already current
  // (advisor, round 4: with canon_valid set the early return skipped the wait and the caller wrote under the rasteriser)
  int rc0 = wait_raster(ctx);
  if (rc0) return rc0;
  if (ctx->canon_valid) return 0;
  if (!unpacked) LAUNCHCHK(ctx, launch_unpack_state(ctx->c, ctx->f, ctx->parity, ctx->have_prev, ctx->stream));
  ctx->canon_valid = true;
  return 0;
}

int ensure_fused(flame_nltgv2_ctx* ctx) {
  if (ctx->fused_valid) return 0;
  // (static_stale: the positions moved, or the topology is new: dx, dy, alpha of the packed records follow in the same launch)
  LAUNCHCHK(ctx, launch_pack_state(ctx->c, ctx->f, ctx->parity, ctx->static_stale, ctx->stream));
  ctx->static_stale = false;
  ctx->fused_valid = true;
  ctx->have_prev = false;
  return 0;
}


// ---- uploading a topology ---------------------------------------------------------------------------------------------
// The host computes only the per-vertex tables (nltgv2_pack.hpp with host_expand = false); every array goes through ONE
// pinned staging buffer (the copies out of it are asynchronous and cost a few microseconds each; out of pageable memory
// each of the ~35 copies of round 1 was a synchronous staging round trip); the per-slot and per-lane arrays are expanded
// on the device (nltgv2_layout.hip).
// All of `cp` through the pinned staging buffer as ONE host-to-device copy into a device-side blob, then one kernel that
// distributes the pieces to their buffers and does the clears of `fills` (k_scatter).  The caller's arrays are free when
// this returns (they were copied into the staging buffer); the staging buffer itself is reused by the next upload, which
// synchronises the stream first.
int staged_h2d(flame_nltgv2_ctx* ctx, const StageCopy* cp, size_t n, const StageFill* fills, size_t n_fills, int slot, hipStream_t stream,
               size_t* host_off) {
  if (!stream) stream = ctx->stream;
  flame_nltgv2_ctx::StageSlot& st = ctx->stage[slot];
  size_t total = 0;
  for (size_t i = 0; i < n; ++i) total += (cp[i].bytes + 255) & ~size_t(255);
  if (total >= (size_t)0xffffff00u) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (total > st.cap) {
    request_open_stop(ctx);  // (the pinned allocator waits for the device)
    if (st.h) (void)hipHostFree(st.h);
    st.h = nullptr, st.cap = 0;
    const size_t want = total + total / 2;
    if (hipHostMalloc(&st.h, want, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return fail(ctx, FLAME_NLTGV2_ERR_OOM);
    }
    st.cap = want;
  }
  int rc = ensure(ctx, st.d, st.cap);
  if (rc) return rc;
  std::vector<ScatterTable> tables(1);
  auto push = [&](const ScatterEntry& e) {
    if (tables.back().n == kScatterMax) tables.emplace_back();
    ScatterTable& t = tables.back();
    t.e[t.n++] = e;
  };
  size_t off = 0;
  for (size_t i = 0; i < n; ++i) {
    if (host_off) host_off[i] = off;
    if (cp[i].bytes == 0) continue;
    std::memcpy(static_cast<char*>(st.h) + off, cp[i].src, cp[i].bytes);
    push(ScatterEntry{cp[i].b->p, (uint32_t)off, 0u, cp[i].bytes});
    off += (cp[i].bytes + 255) & ~size_t(255);
  }
  for (size_t i = 0; i < n_fills; ++i)
    if (fills[i].bytes) push(ScatterEntry{fills[i].dst, kScatterFill, fills[i].word, fills[i].bytes});
  if (off) HIPCHK(ctx, hipMemcpyAsync(st.d.p, st.h, off, hipMemcpyHostToDevice, stream));
  for (const ScatterTable& t : tables) LAUNCHCHK(ctx, launch_scatter(t, st.d.p, stream));
  return 0;
}

// Sizes every buffer of the packed forms for the layout whose scalars stand in ctx->L (V, E, n_slices, rows, wg_count, wg2_count)
// and starts the bookkeeping of a new topology.  Table sizes are passed explicitly: the host builders know them from their
// vectors, the device builder (nltgv2_topo_capi.hip) sizes them by their upper bounds before it runs.
int topology_buffers(flame_nltgv2_ctx* ctx, bool want_e2, size_t n_wg_info, size_t n_wg_v0, size_t n_wg_vfirst, size_t n_wg2_info,
                     size_t n_wg2_vfirst) {
  const PackedLayout& L = ctx->L;
  const int32_t V = L.V, E = L.E;
  const size_t n_slots = (size_t)(L.rows + kRowPad) * kWave;
  if (n_slots > (size_t)0x7fffffff) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  const size_t n_packed = (size_t)L.n_slices * kWave;
  const size_t lanes = (size_t)L.wg_count * kWave;
  const size_t fV = sizeof(float) * (size_t)V, fE = sizeof(float) * (size_t)E, iV = sizeof(int32_t) * (size_t)V;
  struct { DevBuf* b; size_t bytes; } req[] = {
      {&ctx->pos, 2 * fV}, {&ctx->src, fE}, {&ctx->dst, fE},
      {&ctx->row_ptr, sizeof(int32_t) * ((size_t)V + 1)}, {&ctx->half, 2 * fE},
      {&ctx->slice_row, sizeof(int32_t) * ((size_t)L.n_slices + 1)}, {&ctx->perm, sizeof(int32_t) * n_packed},
      {&ctx->pdeg, sizeof(int32_t) * n_packed}, {&ctx->iperm, iV}, {&ctx->order_m, iV}, {&ctx->rid_of, iV},
      {&ctx->rec_nbr, sizeof(uint32_t) * n_slots}, {&ctx->rec_edge, sizeof(int32_t) * n_slots}, {&ctx->edge_src_slot, fE},
      {&ctx->hrec, sizeof(int4) * n_slots}, {&ctx->hq, sizeof(float4) * n_slots},
      {&ctx->vstate, sizeof(float4) * n_packed}, {&ctx->vaux, sizeof(float2) * n_packed},
      {&ctx->hq_alt, sizeof(float4) * n_slots}, {&ctx->vstate_alt, sizeof(float4) * n_packed}, {&ctx->photo_err, fV},
      {&ctx->bar0, sizeof(float4) * n_packed}, {&ctx->bar1, sizeof(float4) * n_packed},
      {&ctx->vprev, sizeof(float4) * n_packed}, {&ctx->xbuf, kXbufBytesPerVertex * records_capacity(L) + 64},
      {&ctx->wg_v0, sizeof(int32_t) * n_wg_v0},
      {&ctx->wg_vfirst, n_wg_vfirst + 16}, {&ctx->abort_flag, sizeof(int)}, {&ctx->err, kErrBytes}, {&ctx->cost_out, 2 * sizeof(float)},
      {&ctx->wg_slot, sizeof(int32_t) * lanes}, {&ctx->wg_vid, sizeof(int32_t) * lanes}, {&ctx->wg_meta, sizeof(uint32_t) * lanes},
      {&ctx->wg_nbr, sizeof(int32_t) * lanes}, {&ctx->wg_fetch, sizeof(int32_t) * lanes},
      {&ctx->wg_info, sizeof(int32_t) * n_wg_info}};
  for (auto& r : req) {
    const int rc = ensure(ctx, *r.b, r.bytes);
    if (rc) return rc;
  }
  if (want_e2) {
    const size_t lanes2 = (size_t)L.wg2_count * kWave;
    struct { DevBuf* b; size_t bytes; } req2[] = {
        {&ctx->wg2_info, sizeof(int32_t) * n_wg2_info}, {&ctx->wg2_vfirst, n_wg2_vfirst + 16}, {&ctx->wg2_rmax, 64},
        {&ctx->wg2_slot, sizeof(int32_t) * 2 * lanes2}, {&ctx->wg2_nbr, sizeof(int32_t) * 2 * lanes2}, {&ctx->wg2_vid, sizeof(int32_t) * lanes2},
        {&ctx->wg2_meta, sizeof(uint32_t) * lanes2}, {&ctx->wg2_fetch, sizeof(int32_t) * lanes2}};
    for (auto& r : req2) {
      const int rc = ensure(ctx, *r.b, r.bytes);
      if (rc) return rc;
    }
  }
  ctx->topo++;
  ctx->layout_pos_saved = false;  // (the positions that are about to stand ARE the layout's)
  ctx->tv_built = ctx->wg2_built = false;
  drop_graphs(ctx);
  refresh_args(ctx);
  return 0;
}

// The clears every new topology needs (no record or flag of an earlier topology may survive).
void expansion_fills(const PackedLayout& L, void* rec_edge, void* rec_nbr, void* wg2_rmax, std::vector<StageFill>* fills) {
  // the spare rows behind the last slice: no edge, neighbour 0 (what the unrolled sweeps may read past a slice's end)
  fills->push_back(StageFill{(char*)rec_edge + sizeof(int32_t) * (size_t)L.rows * kWave, sizeof(int32_t) * kRowPad * kWave, 0xffffffffu});
  fills->push_back(StageFill{(char*)rec_nbr + sizeof(uint32_t) * (size_t)L.rows * kWave, sizeof(uint32_t) * kRowPad * kWave, 0u});
  if (wg2_rmax) fills->push_back(StageFill{wg2_rmax, 64, 0u});
}

void topology_fills(flame_nltgv2_ctx* ctx, bool want_e2, std::vector<StageFill>* fills, bool early) {
  const PackedLayout& L = ctx->L;
  const size_t n_slots = (size_t)(L.rows + kRowPad) * kWave, n_packed = (size_t)L.n_slices * kWave;
  const StageFill base[] = {
      {ctx->err.p, kErrBytes, 0u}, {ctx->abort_flag.p, sizeof(int), 0u}, {ctx->xbuf.p, kXbufBytesPerVertex * records_capacity(L) + 64, 0u},
      // empty slots / padding vertices of the second copies: zero, as the packing kernels write them in the first
      {ctx->hq_alt.p, sizeof(float4) * n_slots, 0u}, {ctx->vstate_alt.p, sizeof(float4) * n_packed, 0u}};
  fills->insert(fills->end(), base, base + sizeof(base) / sizeof(base[0]));
  if (!early) expansion_fills(L, ctx->rec_edge.p, ctx->rec_nbr.p, want_e2 ? ctx->wg2_rmax.p : nullptr, fills);
  // (the tags start over: no record of an earlier topology may survive, in the placement pool either)
  if (ctx->place_base) {
    fills->push_back(StageFill{ctx->place_base, (size_t)2 * kPlacePages * 4096, 0u});
    fills->push_back(StageFill{(int*)ctx->place_fill.p + 2 * kPlacePages, 64, 0u});
  }
}

// Where the records that cross XCDs go (if the pages have been timed already; otherwise the first run does both).
bool placement_applies(const flame_nltgv2_ctx* ctx, const PackedLayout& L) {
  return L.wg_ok && ctx->place_state == 1 && ctx->opt_place && L.wg_count > 2 * (ctx->prop.multiProcessorCount / 8) &&
         (ctx->opt_xcds == 0 || ctx->opt_xcds == 8);
}

// The live tables as an expansion's operands (after refresh_args).
ExpandTables live_tables(flame_nltgv2_ctx* ctx) {
  ExpandTables t;
  t.c = ctx->c, t.f = ctx->f;
  t.iperm = (const int32_t*)ctx->iperm.p, t.wg_v0 = (const int32_t*)ctx->wg_v0.p, t.order_m = (const int32_t*)ctx->order_m.p;
  t.rid_of = (const int32_t*)ctx->rid_of.p, t.wg_vfirst = (const uint8_t*)ctx->wg_vfirst.p;
  t.wg2_info = (int32_t*)ctx->wg2_info.p, t.wg2_vfirst = (const uint8_t*)ctx->wg2_vfirst.p;
  t.wg2_slot = (int32_t*)ctx->wg2_slot.p, t.wg2_vid = (int32_t*)ctx->wg2_vid.p, t.wg2_nbr = (int32_t*)ctx->wg2_nbr.p;
  t.wg2_fetch = (int32_t*)ctx->wg2_fetch.p, t.wg2_meta = (uint32_t*)ctx->wg2_meta.p, t.wg2_rmax = (int*)ctx->wg2_rmax.p;
  return t;
}

// Per-slot and per-lane arrays from the per-vertex tables that stand on the device (nltgv2_layout.hip) and the records' places: the
// launches, on the tables `t` names.
int expand_launches(flame_nltgv2_ctx* ctx, const PackedLayout& L, const ExpandTables& t, bool want_e2, hipStream_t stream) {
  LAUNCHCHK(ctx, launch_build_sell(t.c, t.f, t.iperm, stream));
  if (L.wg_ok) LAUNCHCHK(ctx, launch_build_patches(t.c, t.f, t.wg_v0, t.order_m, t.rid_of, t.wg_vfirst, t.iperm, stream));
  if (want_e2)
    LAUNCHCHK(ctx, launch_build_patches2(t.c, t.f, L.wg2_count, t.wg2_info, t.order_m, t.rid_of, t.wg2_vfirst, t.iperm, t.wg2_slot, t.wg2_vid,
                                         t.wg2_meta, t.wg2_nbr, t.wg2_fetch, t.wg2_rmax, stream));
  if (t.rec_off)
    LAUNCHCHK(ctx, launch_place_records(t.c, t.f, t.per_xcd, t.order_m, t.rid_of, t.place_patch, (int8_t*)(t.place_patch + t.stride),
                                        (const uint16_t*)ctx->place_rank.p, kPlacePages, t.place_fill, t.rec_off, (int)t.stride, stream));
  return 0;
}

// The expansion of the live tables (unless topo_commit has had it done on them while they were the spare set) and the run-side
// bookkeeping of a new topology.
int topology_expand(flame_nltgv2_ctx* ctx, bool want_e2, bool launched) {
  const PackedLayout& L = ctx->L;
  if (!launched) {
    const int rc = expand_launches(ctx, L, live_tables(ctx), want_e2, ctx->stream);
    if (rc) return rc;
  }
  ctx->pv2_args = Pv2Args{};
  ctx->wg2_usable = false;
  if (want_e2) {
    ctx->pv2_args.slot = (const int32_t*)ctx->wg2_slot.p, ctx->pv2_args.vid = (const int32_t*)ctx->wg2_vid.p;
    ctx->pv2_args.meta = (const uint32_t*)ctx->wg2_meta.p, ctx->pv2_args.nbr = (const int32_t*)ctx->wg2_nbr.p;
    ctx->pv2_args.fetch = (const int32_t*)ctx->wg2_fetch.p, ctx->pv2_args.info = (const int32_t*)ctx->wg2_info.p;
    ctx->pv2_args.count = L.wg2_count, ctx->pv2_args.lcap = L.wg2_lcap;
    if (ctx->pv2_occ == 0 || ctx->pv2_occ_lcap != L.wg2_lcap)  // (the kernel's LDS grows with lcap: asked once per value, not once per context)
      ctx->pv2_occ = pv2_patches_per_cu(L.wg2_lcap, false), ctx->pv2_occ_verify = pv2_patches_per_cu(L.wg2_lcap, true), ctx->pv2_occ_lcap = L.wg2_lcap;
    ctx->wg2_built = true;  // (on the device; whether every patch can fetch its records the first plan reads from wg2_rmax)
  }
  // the records' places: on the stream behind the layout kernels, nobody waits for it
  if (!launched && placement_applies(ctx, L)) {
    refresh_args(ctx);
    const int rc = place_records(ctx, (L.wg_count + 7) / 8);
    if (rc) return rc;
  }
  ctx->pending = flame_nltgv2_ctx::PendingRun{};
  ctx->tag_next = 1;
  ctx->xbuf_form = 0;
  ctx->static_stale = false;
  return 0;
}

// Whether the new topology also gets layout (E2), two half-edges per lane: where the one-half-edge patches would fill the CUs
// (kPv2FromPerCu per CU and more), or where it is asked for by name.
bool wants_e2(const flame_nltgv2_ctx* ctx) { return wants_e2(ctx, ctx->L); }
bool wants_e2(const flame_nltgv2_ctx* ctx, const PackedLayout& L) {
  const int cus = ctx->prop.multiProcessorCount;
  const int64_t est_patches = L.wg_rowpack ? (int64_t)L.wg_count : ((static_cast<int64_t>(2) * L.E + L.V / 32) / 54 + 1);
  return L.wg_ok && L.max_degree <= 32 && ctx->opt_probe == 0 &&
         (ctx->opt_persistent == 6 || (ctx->opt_persistent == 1 && est_patches > (int64_t)kPv2FromPerCu * cus));
}

// Layout + topology arrays of `g` (V, E, pos, src, dst) onto the device, the per-vertex tables by the HOST builders
// (nltgv2_pack.hpp); the caller adds the state.  On return the stream still holds the copies: the caller synchronises before the
// staging buffer or `g`'s arrays may change.
int upload_topology(flame_nltgv2_ctx* ctx, const flame_nltgv2_graph* g, const StageCopy* extra, size_t n_extra, bool long_lived) {
  const int32_t V = g->V, E = g->E;
  (void)long_lived;
  int rc = cancel_prepared(ctx);  // (a builder on the side stream reads the topology this is about to replace)
  if (!rc) rc = wait_raster(ctx);  // (... and the rasteriser's side stream the positions)
  if (rc) return rc;
  rc = build_layout(g, &ctx->L, /*host_expand=*/false, /*rowpack=*/true,
                        /*rowpack_max_patches=*/ctx->opt_persistent == 4 ? 0x7fffffff : kPvDensePerCu * ctx->prop.multiProcessorCount);
  if (rc) return fail(ctx, rc);
  // Row packing costs ~15 % more waves than lanes back to back.  It pays where the patch-per-wave kernel runs them; a layout
  // that turns out too large for that kernel (more patches than the estimate) is better off back to back, for the
  // vertex-per-lane form.
  // (FLAME_NLTGV2_OPT_PERSISTENT 4 -- the patch-per-wave form asked for by name -- keeps the row-packed layout whatever the
  //  size: the kernel then runs it as groups of whole components)
  if (ctx->L.wg_ok && ctx->L.wg_rowpack && ctx->opt_persistent != 4 && ctx->L.wg_count > kPvDensePerCu * ctx->prop.multiProcessorCount) {
    rc = build_layout(g, &ctx->L, /*host_expand=*/false, /*rowpack=*/false, 0);
    if (rc) return fail(ctx, rc);
  }
  // Layout (E2): pass 1 (per vertex) here, the lanes on the device below.
  bool want_e2 = wants_e2(ctx);
  if (want_e2) {
    build_patch_walk2(&ctx->L);
    want_e2 = ctx->L.wg2_ok && ctx->L.wg2_count <= kPv2WavesPerCu * ctx->prop.multiProcessorCount * 4;  // (beyond four groups the vertex-per-lane form it is)
  }
  const PackedLayout& L = ctx->L;
  rc = topology_buffers(ctx, want_e2, L.wg_info.size(), L.wg_v0.size(), L.wg_vfirst.size(), L.wg2_info.size(), L.wg2_vfirst.size());
  if (rc) return rc;
  const size_t n_packed = (size_t)L.n_slices * kWave;
  const size_t fV = sizeof(float) * (size_t)V, fE = sizeof(float) * (size_t)E, iV = sizeof(int32_t) * (size_t)V;
  std::vector<StageCopy> cp = {
      {&ctx->pos, g->pos, 2 * fV}, {&ctx->src, g->src, fE}, {&ctx->dst, g->dst, fE},
      {&ctx->row_ptr, L.row_ptr.data(), sizeof(int32_t) * ((size_t)V + 1)}, {&ctx->half, L.half.data(), 2 * fE},
      {&ctx->slice_row, L.slice_row.data(), sizeof(int32_t) * ((size_t)L.n_slices + 1)},
      {&ctx->perm, L.perm.data(), sizeof(int32_t) * n_packed}, {&ctx->pdeg, L.pdeg.data(), sizeof(int32_t) * n_packed},
      {&ctx->iperm, L.iperm.data(), iV}, {&ctx->order_m, L.order_m.data(), iV}, {&ctx->rid_of, L.rid_of.data(), iV},
      {&ctx->wg_info, L.wg_info.data(), sizeof(int32_t) * L.wg_info.size()},
      {&ctx->wg_v0, L.wg_v0.data(), sizeof(int32_t) * L.wg_v0.size()},
      {&ctx->wg_vfirst, L.wg_vfirst.data(), L.wg_vfirst.size()}};
  if (want_e2) {
    cp.push_back(StageCopy{&ctx->wg2_info, L.wg2_info.data(), sizeof(int32_t) * L.wg2_info.size()});
    cp.push_back(StageCopy{&ctx->wg2_vfirst, L.wg2_vfirst.data(), L.wg2_vfirst.size()});
  }
  cp.insert(cp.end(), extra, extra + n_extra);
  std::vector<StageFill> fills;
  topology_fills(ctx, want_e2, &fills);
  rc = staged_h2d(ctx, cp.data(), cp.size(), fills.data(), fills.size());
  if (rc) return rc;
  rc = topology_expand(ctx, want_e2);
  if (rc) return rc;
  ctx->h_src.assign(g->src, g->src + E);
  ctx->h_dst.assign(g->dst, g->dst + E);
  ctx->host_layout_valid = true;
  ctx->feat_dev_valid = false;  // (the device's feature table describes the previous graph)
  HIPCHK(ctx, hipEventRecord(ctx->ev_topo_ready, ctx->stream));
  return 0;
}

// (D) rows: built and uploaded when the vertex-per-lane persistent form is first wanted for the current topology (single
// frames run in the patch-per-wave form and never need them).
int ensure_form_rows(flame_nltgv2_ctx* ctx, int form) {
  PackedLayout& L = ctx->L;
  if ((form == 4 && !ctx->wg2_built) || (form == 2 && !ctx->tv_built)) {  // (the host builders read the host image of the tables; form 5 below does the same)
    const int rc = ensure_host_layout(ctx);
    if (rc) return rc;
  }
  if (form == 4 && !ctx->wg2_built) {
    build_patch_rows2(&L);
    ctx->pv2_args = Pv2Args{};
    if (L.wg2_ok) {
      struct { DevBuf* b; const void* src; size_t bytes; } cp[] = {
          {&ctx->wg2_slot, L.wg2_slot.data(), sizeof(int32_t) * L.wg2_slot.size()}, {&ctx->wg2_vid, L.wg2_vid.data(), sizeof(int32_t) * L.wg2_vid.size()},
          {&ctx->wg2_meta, L.wg2_meta.data(), sizeof(uint32_t) * L.wg2_meta.size()}, {&ctx->wg2_nbr, L.wg2_nbr.data(), sizeof(int32_t) * L.wg2_nbr.size()},
          {&ctx->wg2_fetch, L.wg2_fetch.data(), sizeof(int32_t) * L.wg2_fetch.size()}, {&ctx->wg2_info, L.wg2_info.data(), sizeof(int32_t) * L.wg2_info.size()}};
      HIPCHK(ctx, wait_solver_stream(ctx));
      for (auto& c : cp) {
        int rc = ensure(ctx, *c.b, c.bytes);
        if (!rc) rc = h2d(ctx, *c.b, c.src, c.bytes);
        if (rc) return rc;
      }
      HIPCHK(ctx, wait_solver_stream(ctx));
      ctx->pv2_args.slot = (const int32_t*)ctx->wg2_slot.p, ctx->pv2_args.vid = (const int32_t*)ctx->wg2_vid.p, ctx->pv2_args.meta = (const uint32_t*)ctx->wg2_meta.p;
      ctx->pv2_args.nbr = (const int32_t*)ctx->wg2_nbr.p, ctx->pv2_args.fetch = (const int32_t*)ctx->wg2_fetch.p, ctx->pv2_args.info = (const int32_t*)ctx->wg2_info.p;
      ctx->pv2_args.count = L.wg2_count, ctx->pv2_args.lcap = L.wg2_lcap;
      ctx->pv2_occ = pv2_patches_per_cu(L.wg2_lcap, false), ctx->pv2_occ_verify = pv2_patches_per_cu(L.wg2_lcap, true), ctx->pv2_occ_lcap = L.wg2_lcap;
    }
    ctx->wg2_built = true;
    ctx->wg2_usable = L.wg2_ok, ctx->wg2_checked_topo = ctx->topo;  // (the host builder knows)
  }
  if (form == 2 && !ctx->tv_built) {
    build_tv_rows(&L);
    struct { DevBuf* b; const void* src; size_t bytes; } cp[] = {
        {&ctx->tv_slot, L.tv_slot.data(), sizeof(int32_t) * L.tv_slot.size()}, {&ctx->tv_vid, L.tv_vid.data(), sizeof(int32_t) * L.tv_vid.size()},
        {&ctx->tv_meta, L.tv_meta.data(), sizeof(uint32_t) * L.tv_meta.size()}, {&ctx->tv_wave, L.tv_wave.data(), sizeof(uint32_t) * L.tv_wave.size()}};
    HIPCHK(ctx, wait_solver_stream(ctx));
    for (auto& c : cp) {
      int rc = ensure(ctx, *c.b, c.bytes);
      if (!rc) rc = h2d(ctx, *c.b, c.src, c.bytes);
      if (rc) return rc;
    }
    HIPCHK(ctx, wait_solver_stream(ctx));
    ctx->tv_built = true;
    refresh_args(ctx);
  }
  return 0;
}

bool params_ok(const flame_nltgv2_params* p) { return p != nullptr; }

}  // namespace host
}  // namespace flame_hip

extern "C" {

int flame_nltgv2_abi_version(void) { return FLAME_NLTGV2_ABI_VERSION; }

void flame_nltgv2_default_params(flame_nltgv2_params* p) {
  if (!p) return;
  p->data_factor = 0.1f, p->step_x = 0.001f, p->step_q = 125.0f;
  p->theta = 0.25f, p->x_min = 0.0f, p->x_max = 10.0f;
}

const char* flame_nltgv2_status_string(int status) {
  switch (status) {
    case FLAME_NLTGV2_OK: return "ok";
    case FLAME_NLTGV2_ERR_INVALID_ARG: return "invalid argument";
    case FLAME_NLTGV2_ERR_NO_DEVICE: return "no usable HIP device";
    case FLAME_NLTGV2_ERR_HIP: return "HIP runtime error";
    case FLAME_NLTGV2_ERR_NO_GRAPH: return "no graph uploaded";
    case FLAME_NLTGV2_ERR_NAN: return "dual variable became NaN/Inf (reference FLAME_ASSERT, h:174)";
    case FLAME_NLTGV2_ERR_OOM: return "out of device memory";
    case FLAME_NLTGV2_ERR_TIMEOUT: return "persistent run: neighbour wait timed out";
    case FLAME_NLTGV2_ERR_ASSERT: return "input on which the reference asserts (FLAME_ASSERT)";
    default: return "unknown status";
  }
}

int flame_nltgv2_create(flame_nltgv2_ctx** out, int device) {
  if (!out) return FLAME_NLTGV2_ERR_INVALID_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return FLAME_NLTGV2_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return FLAME_NLTGV2_ERR_NO_DEVICE;
  flame_nltgv2_ctx* ctx = new (std::nothrow) flame_nltgv2_ctx();
  if (!ctx) return FLAME_NLTGV2_ERR_OOM;
  ctx->device = device;
  bool ok = hipSetDevice(device) == hipSuccess;
  ok = ok && hipGetDeviceProperties(&ctx->prop, device) == hipSuccess;
  // The context's own stream is a HIGH-priority stream.  Every stream that submits work takes one of the process's in-order hardware
  // queues (GPU_MAX_HW_QUEUES, 4 by default, per priority level), and the device's cooperative queue -- every new topology's first run
  // goes through it -- is created by the first cooperative launch of the process, which since round 5 is the warm-up in
  // flame_nltgv2_create, on this stream.  Measured on the pipelined frame loop (tools/frame_loop.py, solver on a high-priority stream of
  // the caller's): with a normal-priority stream here the solver stood still 0.15 instead of 0.09 ms per commit at 640x480 (0.2-0.25
  // instead of 0.11-0.12 at 1920x1080) -- builder, rasteriser, tracker and this stream then share the four normal queues.
  int prio_least = 0, prio_greatest = 0;
  ok = ok && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess;
  ok = ok && hipStreamCreateWithPriority(&ctx->own_stream, hipStreamNonBlocking, prio_greatest) == hipSuccess;
  ok = ok && hipEventCreate(&ctx->ev0) == hipSuccess && hipEventCreate(&ctx->ev1) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&ctx->topo_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&ctx->raster_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->ev_canon, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->ev_snap, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->ev_raster_done, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->ev_topo_ready, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->ev_expanded, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->ev_run[0], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->ev_run[1], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&ctx->h_err, kErrBytes, hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&ctx->h_cost, 2 * sizeof(float), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&ctx->h_stop, 64, hipHostMallocDefault) == hipSuccess;
  if (ctx->h_stop) *ctx->h_stop = 0u;
  ok = ok && hipStreamCreateWithPriority(&ctx->ctl_stream, hipStreamNonBlocking, -1) == hipSuccess;
  if (!ok) {
    flame_nltgv2_destroy(ctx);
    return FLAME_NLTGV2_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  *ctx->h_err = 0;
  {
    // The first context of a process pays what would otherwise land on its first frame (tools/cold_start.py: 16 ms of code-object
    // loading in the first upload_graph, 4 ms of record-placement calibration in the first run): the kernels' code objects are
    // loaded here, and the page ranking of this device is measured here, once per process (later contexts take it from the cache).
    static std::once_flag warm;
    if (!std::getenv("FLAME_NLTGV2_NO_WARM"))
    std::call_once(warm, [ctx] {
      warm_module_kernels(), warm_module_persistent(ctx->own_stream, cooperative_allowed()), warm_module_persistent_tv(), warm_module_persistent_pv2();
      warm_module_layout(), warm_module_topo();
    });
    if (!std::getenv("FLAME_NLTGV2_LAZY_CALIBRATION") && ctx->prop.multiProcessorCount >= 64 && place_calibrate_at_create(device) && place_calibrate(ctx) != 0) {
      ctx->last_error = 0, ctx->last_hip = 0;
      (void)hipGetLastError();
    }
    if (ctx->place_state < 0) ctx->place_state = 0;  // (not now: the first run that can use it tries again)
  }
  ctx->all = {&ctx->pos, &ctx->x, &ctx->w1, &ctx->w2, &ctx->xb, &ctx->w1b, &ctx->w2b, &ctx->xp, &ctx->w1p,
              &ctx->w2p, &ctx->data, &ctx->weight, &ctx->src, &ctx->dst, &ctx->alpha, &ctx->beta, &ctx->q1,
              &ctx->q2, &ctx->q3, &ctx->row_ptr, &ctx->half, &ctx->slice_row, &ctx->perm, &ctx->pdeg,
              &ctx->rec_nbr, &ctx->rec_edge, &ctx->edge_src_slot, &ctx->hrec, &ctx->hq, &ctx->vstate, &ctx->hq_alt, &ctx->vstate_alt, &ctx->cost_terms, &ctx->run_tail,
              &ctx->vaux, &ctx->bar0, &ctx->bar1, &ctx->vprev, &ctx->xbuf, &ctx->abort_flag, &ctx->tv_slot, &ctx->tv_vid, &ctx->tv_meta, &ctx->tv_wave, &ctx->wg2_slot, &ctx->wg2_vid, &ctx->wg2_meta, &ctx->wg2_nbr, &ctx->wg2_fetch, &ctx->wg2_info, &ctx->wg2_vfirst, &ctx->wg2_rmax, &ctx->err,
              &ctx->cost_out, &ctx->img_ref, &ctx->img_cmp, &ctx->photo_err, &ctx->r_tris, &ctx->r_valid, &ctx->r_tvalid, &ctx->r_keys,
              &ctx->r_img, &ctx->r_cov, &ctx->r_vtx, &ctx->r_val, &ctx->wg_slot, &ctx->wg_vid, &ctx->wg_meta, &ctx->wg_nbr,
              &ctx->wg_fetch, &ctx->wg_info, &ctx->wg_v0, &ctx->probe, &ctx->snap_hq, &ctx->snap_vstate, &ctx->snap_bar, &ctx->iperm,
              &ctx->order_m, &ctx->rid_of, &ctx->stage[0].d, &ctx->stage[1].d, &ctx->sync_init, &ctx->sync_vmap, &ctx->sync_emap, &ctx->sync_need,
              &ctx->wg_vfirst, &ctx->place_pool, &ctx->place_rank, &ctx->place_fill, &ctx->place_rec_off, &ctx->place_patch, &ctx->place_meas, &ctx->progress};
  for (DevBuf* b : {&ctx->feat_stamp_d, &ctx->feat_key_d, &ctx->feat_val_d, &ctx->topo_scratch, &ctx->topo_dims, &ctx->layout_pos}) ctx->all.push_back(b);
  for (auto& b : ctx->nx) ctx->all.push_back(&b);
  for (auto& b : ctx->ex) ctx->all.push_back(&b);
  ctx->all.push_back(&ctx->pos_undo);
  ctx->all.push_back(&ctx->stop_dev);
  ctx->all.push_back(&ctx->place_patch_nx), ctx->all.push_back(&ctx->place_fill_nx);
  for (auto& b : ctx->sp_v) ctx->all.push_back(&b);
  for (auto& b : ctx->sp_q) ctx->all.push_back(&b);
  for (auto& b : ctx->sp_ab) ctx->all.push_back(&b);
  *out = ctx;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_destroy(flame_nltgv2_ctx* ctx) {
  if (!ctx) return FLAME_NLTGV2_OK;
  (void)hipSetDevice(ctx->device);
  request_open_stop(ctx);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  place_pool_release(ctx);
  drop_graphs(ctx);
  for (DevBuf* b : ctx->all)
    if (b->p) (void)hipFree(b->p);
  if (ctx->h_err) (void)hipHostFree(ctx->h_err);
  if (ctx->h_cost) (void)hipHostFree(ctx->h_cost);
  if (ctx->h_keep) (void)hipHostFree(ctx->h_keep);
  if (ctx->h_stop) (void)hipHostFree(ctx->h_stop);
  if (ctx->ctl_stream) (void)hipStreamSynchronize(ctx->ctl_stream), (void)hipStreamDestroy(ctx->ctl_stream);
  for (auto& st : ctx->stage)
    if (st.h) (void)hipHostFree(st.h);
  if (ctx->topo_stream) (void)hipStreamSynchronize(ctx->topo_stream), (void)hipStreamDestroy(ctx->topo_stream);
  if (ctx->ev_topo_ready) (void)hipEventDestroy(ctx->ev_topo_ready);
  if (ctx->ev_expanded) (void)hipEventDestroy(ctx->ev_expanded);
  if (ctx->raster_stream) (void)hipStreamSynchronize(ctx->raster_stream), (void)hipStreamDestroy(ctx->raster_stream);
  if (ctx->ev_canon) (void)hipEventDestroy(ctx->ev_canon);
  if (ctx->ev_snap) (void)hipEventDestroy(ctx->ev_snap);
  if (ctx->ev_raster_done) (void)hipEventDestroy(ctx->ev_raster_done);
  for (hipEvent_t e : ctx->ev_run)
    if (e) (void)hipEventDestroy(e);
  if (ctx->h_img) (void)hipHostFree(ctx->h_img);
  if (ctx->h_dims) (void)hipHostFree(ctx->h_dims);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_set_stream(flame_nltgv2_ctx* ctx, void* hip_stream) {
  int rc = enter(ctx);
  if (rc) return rc;
  HIPCHK(ctx, wait_solver_stream(ctx));
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_set_option(flame_nltgv2_ctx* ctx, int option, int value) {
  if (!ctx) return FLAME_NLTGV2_ERR_INVALID_ARG;
  switch (option) {
    case FLAME_NLTGV2_OPT_SOLVER:
      if (value != 0 && value != 1) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_solver = value;
      return 0;
    case FLAME_NLTGV2_OPT_USE_HIPGRAPH:
      ctx->opt_use_graph = value ? 1 : 0;
      return 0;
    case FLAME_NLTGV2_OPT_BLOCK_WAVES:
      if (value != 0 && value != 1 && value != 2 && value != 4) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_block_waves = value;
      return 0;
    case FLAME_NLTGV2_OPT_PERSISTENT:
      if (value < 0 || value > 6 || value == 2 || value == 5) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);  // (2: the retired lane-per-half-edge form; 7, the
      // region-per-workgroup form of round 5, left the library in round 6: measured 15 % slower at every size, profiles/r05_wg_region.txt)
      ctx->opt_persistent = value;
      return 0;
    case FLAME_NLTGV2_OPT_PLACEMENT:
      if (value < 0 || value > 1) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_place = value;
      return 0;
    case FLAME_NLTGV2_OPT_POLL_GAP:
      if (value < 0 || value > 256) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_poll_gap = value;
      return 0;
    case FLAME_NLTGV2_OPT_VERIFY_RECORDS:
      if (value < 0 || value > 2) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_verify = value;
      if (value != 2) ctx->persist_refused_topo = ctx->persist_backoff_topo = ~0ull;  // hook off: let the persistent path be tried again
      return 0;
    case FLAME_NLTGV2_OPT_PROBE:
      ctx->opt_probe = value ? 1 : 0;
      return 0;
    case FLAME_NLTGV2_OPT_DUAL_PUBLISH:
      if (value < 0 || value > 2) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_dual = value;
      return 0;
    case FLAME_NLTGV2_OPT_FAULT_INJECT:
      if (value < 0 || value > (1 << 24)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_fault = value;
      if (value == 0) ctx->persist_refused_topo = ctx->persist_backoff_topo = ~0ull;  // let the persistent path be tried again
      return 0;
    case FLAME_NLTGV2_OPT_XCDS:
      if (value < 0 || value > 8) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_xcds = value;
      ctx->coop_checked_key = ~0ull;
      return 0;
    case FLAME_NLTGV2_OPT_PRESLEEP:
      if (value < 0 || value > 256) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_presleep = value;
      return 0;
    case FLAME_NLTGV2_OPT_MESH_STATE:
      if (value < 0 || value > 1) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_mesh_state = value;
      return 0;
    case FLAME_NLTGV2_OPT_COST_SUM:
      if (value < 0 || value > 1) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_cost_sum = value;
      return 0;
    case FLAME_NLTGV2_OPT_SYNC_PATH:
      if (value < 0 || value > 2) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_sync_path = value;
      return 0;
    case FLAME_NLTGV2_OPT_UNROLL:
      if (value != 0 && value != 4 && value != 8 && value != 16) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_unroll = value;
      return 0;
    default:
      return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  }
}


int flame_nltgv2_get_info(flame_nltgv2_ctx* ctx, flame_nltgv2_info* info) {
  if (!ctx || !info) return FLAME_NLTGV2_ERR_INVALID_ARG;
  std::memset(info, 0, sizeof(*info));
  info->abi_version = FLAME_NLTGV2_ABI_VERSION;
  info->device = ctx->device;
  info->V = ctx->L.V, info->E = ctx->L.E;
  info->n_slices = ctx->L.n_slices;
  info->max_degree = ctx->L.max_degree;
  info->padded_half_edges = ctx->L.rows * kWave;
  info->device_bytes = (int64_t)ctx->device_bytes;
  info->algorithmic_bytes_per_iter = 64ll * ctx->L.V + 40ll * ctx->L.E;
  info->compute_units = ctx->prop.multiProcessorCount;
  std::snprintf(info->device_name, sizeof(info->device_name), "%s", ctx->prop.name);
  std::snprintf(info->gcn_arch, sizeof(info->gcn_arch), "%s", ctx->prop.gcnArchName);
  info->last_run_path = ctx->last_run_path;
  info->he_waves = 0;  // (the lane-per-half-edge form was retired in round 3; the field stays for the ABI)
  if (ctx->have_graph && !ctx->tv_built && ctx->tv_counted_topo != ctx->topo) {
    // the vertex-per-lane rows are built on demand; a caller sizing a batch asks here (host table only, once per topology; the
    // upload happens when the form is first used)
    if (ensure_host_layout(ctx) == 0) {
      ctx->L.tv_waves = 0;
      build_tv_rows(&ctx->L);
      ctx->tv_counted_topo = ctx->topo;
    }
  }
  info->tv_waves = ctx->L.tv_ok ? ctx->L.tv_waves : 0;
  info->patches = ctx->L.wg_ok ? ctx->L.wg_count : 0;
  info->tv_wave_capacity = kTvLdsWavesPerCu * ctx->prop.multiProcessorCount;
  info->last_run_groups = ctx->last_run_groups;
  info->timeouts_recovered = ctx->timeouts_recovered;
  info->torn_records_detected = ctx->torn_records_detected;
  info->last_sync_path = ctx->last_sync_path;
  info->last_run_waves_per_cu = ctx->last_run_waves_per_cu;
  info->reserved0 = 0, info->reserved1 = 0;
  info->replays_per_step = ctx->replays_per_step;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_last_error(flame_nltgv2_ctx* ctx) { return ctx ? ctx->last_error : FLAME_NLTGV2_ERR_INVALID_ARG; }
int flame_nltgv2_last_hip_error(flame_nltgv2_ctx* ctx) { return ctx ? ctx->last_hip : 0; }

int flame_nltgv2_read_probe(flame_nltgv2_ctx* ctx, uint32_t* out, int64_t max_words, int64_t* n_words) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (ctx->pending.active) {
    rc = finish(ctx);
    if (rc) return rc;
  }
  HIPCHK(ctx, wait_solver_stream(ctx));
  const int64_t have = (int64_t)ctx->probe_words;
  if (n_words) *n_words = have;
  if (out && ctx->probe.p) {
    const int64_t n = std::min(have, max_words);
    if (n > 0) HIPCHK(ctx, hipMemcpy(out, ctx->probe.p, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
  }
  return FLAME_NLTGV2_OK;
}

// The per-slot / per-lane layout arrays as the device expanded them (nltgv2_layout.hip) against the host builders of
// nltgv2_pack.hpp on the same topology: number of differing words (0 = identical).
int flame_nltgv2_layout_selftest(flame_nltgv2_ctx* ctx, int64_t* mismatches) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!mismatches) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  rc = ensure_canon(ctx);
  if (!rc) rc = ensure_host_layout(ctx);
  if (rc) return rc;
  const int32_t V = ctx->L.V, E = ctx->L.E;
  std::vector<float> pos(2 * (size_t)V);
  HIPCHK(ctx, hipMemcpyAsync(pos.data(), ctx->layout_pos_saved ? ctx->layout_pos.p : ctx->pos.p, sizeof(float) * pos.size(), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));
  flame_nltgv2_graph g{};
  g.V = V, g.E = E, g.pos = pos.data(), g.src = ctx->h_src.data(), g.dst = ctx->h_dst.data();
  PackedLayout H;
  rc = build_layout(&g, &H, /*host_expand=*/true, ctx->L.wg_rowpack);  // (the modes the layout was actually built with)
  if (rc) return fail(ctx, rc);
  int64_t bad = 0;
  auto cmp = [&](const DevBuf& b, const void* host, size_t bytes) -> int {
    if (bytes == 0) return 0;
    std::vector<uint32_t> d(bytes / 4);
    if (hipMemcpy(d.data(), b.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const uint32_t* h = static_cast<const uint32_t*>(host);
    for (size_t i = 0; i < d.size(); ++i) bad += d[i] != h[i];
    return 0;
  };
  const PackedLayout& L = ctx->L;
  bad += (H.rows != L.rows) + (H.n_slices != L.n_slices) + (H.wg_ok != L.wg_ok) + (H.wg_count != L.wg_count) +
         (H.wg_lcap != L.wg_lcap) + (H.wg_slab_slots != L.wg_slab_slots) + (H.n_rec != L.n_rec) +
         (H.wg_v0 != L.wg_v0) + (H.wg_vfirst != L.wg_vfirst) + (H.wg_rowpack != L.wg_rowpack);
  if (bad == 0) {  // the per-vertex tables as they stand on the device (sent up by the host path, or built there: nltgv2_topo.hip)
    const size_t iV = 4 * (size_t)V;
    int e = cmp(ctx->row_ptr, H.row_ptr.data(), iV + 4) | cmp(ctx->half, H.half.data(), 8 * (size_t)E) | cmp(ctx->order_m, H.order_m.data(), iV) |
            cmp(ctx->rid_of, H.rid_of.data(), iV) | cmp(ctx->iperm, H.iperm.data(), iV) | cmp(ctx->pdeg, H.pdeg.data(), 4 * H.pdeg.size()) |
            cmp(ctx->wg_v0, H.wg_v0.data(), 4 * H.wg_v0.size()) | cmp(ctx->wg_vfirst, H.wg_vfirst.data(), H.wg_vfirst.size() & ~size_t(3));
    if (e) return fail(ctx, FLAME_NLTGV2_ERR_HIP);
    std::vector<int32_t> hs(ctx->h_src.size()), hd(ctx->h_dst.size());  // (and the host image of the edge list is the device's)
    if (E && (hipMemcpy(hs.data(), ctx->src.p, 4 * (size_t)E, hipMemcpyDeviceToHost) != hipSuccess ||
              hipMemcpy(hd.data(), ctx->dst.p, 4 * (size_t)E, hipMemcpyDeviceToHost) != hipSuccess))
      return fail(ctx, FLAME_NLTGV2_ERR_HIP);
    bad += (hs != ctx->h_src) + (hd != ctx->h_dst);
  }
  if (bad == 0) {
    const size_t n = (size_t)L.rows * kWave, lanes = (size_t)L.wg_count * kWave;
    int e = cmp(ctx->rec_nbr, H.rec_nbr.data(), 4 * n) | cmp(ctx->rec_edge, H.rec_edge.data(), 4 * n) |
            cmp(ctx->edge_src_slot, H.edge_src_slot.data(), 4 * (size_t)E) | cmp(ctx->perm, H.perm.data(), 4 * H.perm.size()) |
            cmp(ctx->slice_row, H.slice_row.data(), 4 * H.slice_row.size());
    if (L.wg_ok)
      e |= cmp(ctx->wg_slot, H.wg_slot.data(), 4 * lanes) | cmp(ctx->wg_vid, H.wg_vid.data(), 4 * lanes) |
           cmp(ctx->wg_meta, H.wg_meta.data(), 4 * lanes) | cmp(ctx->wg_nbr, H.wg_nbr.data(), 4 * lanes) |
           cmp(ctx->wg_fetch, H.wg_fetch.data(), 4 * lanes) | cmp(ctx->wg_info, H.wg_info.data(), 4 * H.wg_info.size());
    if (ctx->wg2_built) {  // (E2), two half-edges per lane: the device expansion against the host builder
      build_patch_rows2(&H);
      bad += (H.wg2_ok != L.wg2_ok) + (H.wg2_count != L.wg2_count) + (H.wg2_lcap != L.wg2_lcap) + (H.wg2_vfirst != L.wg2_vfirst);
      if (bad == 0 && H.wg2_ok) {
        const size_t lanes2 = (size_t)H.wg2_count * kWave;
        e |= cmp(ctx->wg2_slot, H.wg2_slot.data(), 8 * lanes2) | cmp(ctx->wg2_nbr, H.wg2_nbr.data(), 8 * lanes2) |
             cmp(ctx->wg2_vid, H.wg2_vid.data(), 4 * lanes2) | cmp(ctx->wg2_meta, H.wg2_meta.data(), 4 * lanes2) |
             cmp(ctx->wg2_fetch, H.wg2_fetch.data(), 4 * lanes2) | cmp(ctx->wg2_info, H.wg2_info.data(), 4 * H.wg2_info.size());
      }
    }
    if (e) return fail(ctx, FLAME_NLTGV2_ERR_HIP);
  }
  if (bad == 0 && ctx->place_state == 1 && ctx->place_topo == ctx->topo && L.wg_ok) {
    // placed records: aligned, inside their parity's half of the pool, no slot given out twice -- and exactly the records a
    // patch on another XCD reads (host: the same rule as k_place_assign, from the host's own patch walk)
    const size_t stride = records_capacity(L);
    std::vector<int32_t> off(2 * stride);
    if (hipMemcpy(off.data(), ctx->place_rec_off.p, sizeof(int32_t) * off.size(), hipMemcpyDeviceToHost) != hipSuccess)
      return fail(ctx, FLAME_NLTGV2_ERR_HIP);
    std::vector<int32_t> patch_of_rec((size_t)V, -1);
    for (int32_t q = 0; q < H.wg_count; ++q)
      for (int32_t i = 0; i < (H.wg_info[(size_t)q * 4 + 2] & 0xffff); ++i) patch_of_rec[(size_t)H.wg_info[(size_t)q * 4] + i] = q;
    const int32_t per = ctx->place_per_xcd;
    for (int par = 0; par < 2; ++par) {
      std::vector<int32_t> used;
      for (int32_t u = 0; u < V; ++u) {
        const int32_t r = H.rid_of[(size_t)u], a = patch_of_rec[(size_t)r] / per;
        bool crosses = false;
        for (int32_t h = H.row_ptr[(size_t)u]; h < H.row_ptr[(size_t)u + 1] && !crosses; ++h)
          crosses = patch_of_rec[(size_t)H.rid_of[(size_t)H.half_nbr[(size_t)h]]] / per != a;
        const int32_t o = off[(size_t)par * stride + r];
        bad += crosses != (o >= 0);
        if (o < 0) continue;
        bad += (o & 15) != 0 || o < par * kPlacePages * 4096 || o >= (par + 1) * kPlacePages * 4096;
        used.push_back(o);
      }
      std::sort(used.begin(), used.end());
      for (size_t i = 1; i < used.size(); ++i) bad += used[i] == used[i - 1];
    }
  }
  *mismatches = bad;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_placement_info(flame_nltgv2_ctx* ctx, int32_t* state, int32_t* placed_records, float* us) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (state) *state = ctx->place_state;
  if (us) us[0] = ctx->place_best_us, us[1] = ctx->place_mean_us, us[2] = ctx->place_worst_us;
  if (placed_records) {
    *placed_records = 0;
    if (ctx->have_graph && ctx->place_state == 1 && ctx->place_topo == ctx->topo) {
      std::vector<int32_t> off((size_t)ctx->L.V);  // (parity 0; the walk's records)
      HIPCHK(ctx, hipMemcpyAsync(off.data(), ctx->place_rec_off.p, sizeof(int32_t) * off.size(), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, wait_solver_stream(ctx));
      for (int32_t o : off) *placed_records += o >= 0;
    }
  }
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_pack_probe(const flame_nltgv2_graph* g, int32_t* perm, int32_t* slice_row, int32_t* rec_nbr,
                            int32_t* rec_edge, int64_t capacity_rows, int64_t* rows_out) {
  PackedLayout L;
  int rc = build_layout(g, &L);
  if (rc) return rc;
  if (rows_out) *rows_out = L.rows;
  if (perm) std::memcpy(perm, L.perm.data(), sizeof(int32_t) * L.perm.size());
  if (slice_row) std::memcpy(slice_row, L.slice_row.data(), sizeof(int32_t) * L.slice_row.size());
  if ((rec_nbr || rec_edge) && capacity_rows < L.rows) return FLAME_NLTGV2_ERR_INVALID_ARG;
  const size_t n = (size_t)L.rows * kWave;
  if (rec_nbr) std::memcpy(rec_nbr, L.rec_nbr.data(), sizeof(int32_t) * n);
  if (rec_edge) std::memcpy(rec_edge, L.rec_edge.data(), sizeof(int32_t) * n);
  return L.n_slices;
}

}  // extern "C"
