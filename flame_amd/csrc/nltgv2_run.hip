// nltgv2_run.hip -- running n steps: which kernels (planner), enqueue, settle, roll back and redo; the run entry points of the C-ABI
// (see nltgv2_context.hpp).
#include "nltgv2_context.hpp"

namespace flame_hip {
namespace host {

void pick_config(const flame_nltgv2_ctx* ctx, int* unroll, int* wpb) {
  // Small graphs are latency bound: one wave per workgroup spreads the slices over as many CUs as
  // possible and a deep chunk (all slots of a slice in one round of loads) shortens the dependent
  // chain.  Large graphs / batches are bandwidth bound: 4-wave workgroups, shallow chunks keep the
  // register footprint (and thus occupancy) reasonable.
  const bool small = ctx->L.n_slices <= 8 * ctx->prop.multiProcessorCount;
  *unroll = ctx->opt_unroll ? ctx->opt_unroll : (small ? 8 : 4);
  *wpb = ctx->opt_block_waves ? ctx->opt_block_waves : (small ? 1 : 4);
}

// A persistent launch covers a contiguous range of waves of one form.
struct WaveGroup {
  int begin, count;
};

// Which persistent form (if any) runs n steps -- 0 none (one launch per step), 1 lane-per-half-edge
// (lowest latency), 2 vertex-per-lane (fewest instructions) -- and over which wave groups.  A graph that
// is resident as a whole is one group.  A disjoint union too large for that (a big batch of frames) is run
// group of connected components by group, each group resident on its own: the components are independent,
// so running them one after the other for all n steps is exactly the same computation.
int ensure_form_rows(flame_nltgv2_ctx* ctx, int form);
int plan_groups(flame_nltgv2_ctx* ctx, int form, int total, int cap, const std::vector<int32_t>& cw, std::vector<WaveGroup>* groups);

// Cooperative launches and rocprofiler-sdk do not end well together (ROCm 7.2): a process that runs under rocprofv3 and has made ONE
// cooperative launch -- of any kernel: tools/coop_exit_repro.hip is 30 lines without this library -- dies with SIGSEGV inside
// libhsa-runtime64 when the HIP runtime's static destructor (amd::Runtime::tearDown) takes the device's cooperative queue down after the
// tool has finalised (profiles/r06_segv.txt: the backtrace and the bisection; rounds 2-5 logged it as "every profiled process ends with a
// segmentation fault").  The cooperative launch is only the first launch of a (topology, form): the runtime's check that the grid is
// resident as a whole, which the planner derives itself anyway (pv_patches_per_cu: the query over-reports), and the launch without
// it is the one every later run of the topology makes.  So under a profiler the first launch is a plain one too.
// FLAME_NLTGV2_NO_COOPERATIVE=1 does the same by hand, FLAME_NLTGV2_COOPERATIVE=1 keeps the cooperative launches whatever is loaded.
bool cooperative_allowed() {
  static const bool ok = [] {
    if (std::getenv("FLAME_NLTGV2_NO_COOPERATIVE")) return false;
    if (std::getenv("FLAME_NLTGV2_COOPERATIVE")) return true;
    if (std::getenv("ROCP_TOOL_LIBRARIES")) return false;  // (what rocprofv3 exports for the process it starts)
    if (void* h = dlopen("librocprofiler-sdk.so.1", RTLD_NOLOAD | RTLD_LAZY)) {  // a tool library is in the process
      dlclose(h);
      return false;
    }
    return true;
  }();
  return ok;
}

// `consume`: the call plans a run that is about to be enqueued (enqueue_run) -- only then does a planned-around run count against
// the back-off after an expired wait; a query (persistent_eligible, prepare_run) leaves the bookkeeping alone.
int plan_persistent(flame_nltgv2_ctx* ctx, int n, std::vector<WaveGroup>* groups, int* use_tv_lds = nullptr, bool consume = false) {
  groups->clear();
  if (use_tv_lds) *use_tv_lds = 0;
  if (!ctx->opt_persistent || n < 4 || n > (1 << 24) || !ctx->prop.cooperativeLaunch) return 0;
  if (ctx->persist_refused_topo == ctx->topo || ctx->replaying == 2) return 0;
  const bool retry = ctx->replaying == 1;  // the first replay of an expired chain: persistent once more, with room left on every CU
  if (!retry && ctx->persist_backoff_topo == ctx->topo && (ctx->persist_backoff_left > 0 || ctx->opt_fault > 0)) {  // (the test hook's fault does not pass)
    if (consume && ctx->persist_backoff_left > 0) --ctx->persist_backoff_left;
    return 0;
  }
  PackedLayout& L = ctx->L;
  const int cus = ctx->prop.multiProcessorCount;
  // ask once per (topology, kernel instance): the LDS use varies with the layout, the registers with the instance
  const uint64_t occ_key = ctx->topo * 4 + (ctx->opt_verify != 0 ? 1 : 0) + (ctx->opt_probe != 0 ? 2 : 0);
  if (L.wg_ok && ctx->pv_occ_topo != occ_key) {
    // The REAL residency (pv_patches_per_cu: the runtime's query over-reports, tools/residency_probe.hip), and of that at most
    // kPvDensePerCu: beyond it the lane-per-half-edge / vertex-per-lane forms are as fast or faster (tools/pv_big.py,
    // profiles/r03_pv_dense.txt: the hand-off itself gets slower with the number of polling waves)
    ctx->pv_occ = std::min(kPvDensePerCu, pv_patches_per_cu(ctx->f, ctx->opt_verify != 0 || ctx->opt_probe != 0));
    if (std::getenv("FLAME_NLTGV2_TRACE"))
      std::fprintf(stderr, "[flame_nltgv2] pv: %d patches, row-packed %d, slab slots %d, local records %d -> %d resident per CU\n", L.wg_count,
                   (int)L.wg_rowpack, ctx->f.wg_slab_slots, ctx->f.wg_lcap, ctx->pv_occ);
    ctx->pv_occ_topo = occ_key;
  }
  const bool crowded = (ctx->topo < ctx->crowded_until_topo && ctx->opt_persistent == 1) || retry;  // (a form asked for by name is run as asked)
  const int wg_cap = (crowded ? std::min(ctx->pv_occ, kCrowdedWavesPerCu) : ctx->pv_occ) * cus;  // patch-per-wave form, in patches
  // The vertex-per-lane rows (D) are built only when that form is actually chosen.
  const bool pv_fits = L.wg_ok && L.wg_rowpack && L.wg_count > 0 && L.wg_count <= wg_cap;  // (the kernel runs row-packed patches)
  int form = 0;
  // The two-half-edges-per-lane form: half the waves for the same graph.  Its lanes were expanded on the device at upload where
  // the graph is large enough (upload_topology); whether every patch can fetch its records (<= 64 distinct foreign ones) the
  // expansion left in a word that is read here, once per topology.
  auto pv2_usable = [&]() -> bool {
    if (!ctx->wg2_built) return false;
    if (ctx->wg2_checked_topo != ctx->topo) {
      int rmax = 0;
      if (hipMemcpyAsync(&rmax, ctx->wg2_rmax.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        return false;
      }
      ctx->wg2_usable = rmax <= 64 && L.wg2_ok && ctx->pv2_occ > 1 && ctx->pv2_occ_verify > 1;
      ctx->wg2_checked_topo = ctx->topo;
    }
    return ctx->wg2_usable;
  };
  const int wg2_cap = std::min(crowded ? kCrowdedWavesPerCu : kPv2WavesPerCu, std::max(1, (ctx->opt_verify ? ctx->pv2_occ_verify : ctx->pv2_occ) - 1)) * cus;
  if (ctx->opt_persistent == 4) form = (L.wg_ok && L.wg_rowpack) ? 3 : 0;
  else if (ctx->opt_persistent == 6 && ctx->opt_probe == 0) {  // ... by name (it has no cycle probe: with that on, the one-half-edge form or the sweep)
    if (!ctx->wg2_built && ensure_form_rows(ctx, 4) != 0) return 0;
    form = pv2_usable() ? 4 : 0;
  } else if (ctx->opt_persistent == 3) form = 2;
  else if (ctx->opt_probe == 0 && ctx->wg2_built && L.wg2_count <= kPv2MaxGroups * wg2_cap &&
           (!pv_fits || L.wg_count > kPv2FromPerCu * cus) && pv2_usable()) form = 4;  // (beyond one launch: as two groups of whole frames, below)
  else if (pv_fits) form = 3;  // lowest latency wherever all patches are resident: 320x240 ... 1920x1080 single frames, 2-7 frames of 640x480
  else form = 2;               // too big for that: vertex-per-lane, in groups of whole components if need be
  if (form == 2) {
    if (ensure_form_rows(ctx, form) != 0) return 0;
    if (!L.tv_ok) {  // (a vertex of more than 512 edges: the per-step sweep, unless the patch form can run it in groups)
      form = (L.wg_ok && L.wg_rowpack && L.wg_count > 0) ? 3 : 0;
    }
  }
  // vertex-per-lane form: the per-slot constants in LDS (16 waves per CU: 30 frames of 640x480 resident in one launch).  (The
  // instance that kept them in registers -- 8 waves per CU, 9 % faster per wave -- was retired in round 4: the batches it ran, 11 to
  // 15 frames, are faster as two groups of the two-half-edges patch form: profiles/r04_large_batches.txt.)
  const bool tv_lds = form == 2;
  const int tv_cap = kTvLdsWavesPerCu * cus;
  if (use_tv_lds) *use_tv_lds = tv_lds ? 1 : 0;
  if (form == 0) return 0;
  const int total = form == 4 ? L.wg2_count : form == 3 ? L.wg_count : L.tv_waves;
  const int cap = form == 4 ? wg2_cap : form == 3 ? wg_cap : tv_cap;
  if (total <= 0) return 0;
  if (total <= cap) {
    groups->push_back(WaveGroup{0, total});
    return form;
  }
  if (ensure_host_layout(ctx) != 0) return 0;  // (the component tables live in the host image of the layout)
  if (form == 4 && ctx->opt_persistent == 1) {  // groups of the two-half-edges form only where whole frames make them up
    const std::vector<int32_t>& c2 = L.comp_wg2;
    bool ok2 = c2.size() >= 3;
    for (size_t c = 0; ok2 && c + 1 < c2.size(); ++c) ok2 = c2[c + 1] - c2[c] <= cap;
    if (!ok2) {
      if (ensure_form_rows(ctx, 2) != 0 || !L.tv_ok) return 0;
      if (use_tv_lds) *use_tv_lds = 1;
      groups->clear();
      return plan_groups(ctx, 2, L.tv_waves, kTvLdsWavesPerCu * cus, L.comp_tv_wave, groups);
    }
  }
  const std::vector<int32_t>& cw = form == 4 ? L.comp_wg2 : form == 3 ? L.comp_wg : L.comp_tv_wave;
  return plan_groups(ctx, form, total, cap, cw, groups);
}

// `total` waves of `form` over groups of whole components (cw: first wave of every component), each at most `cap` waves.
int plan_groups(flame_nltgv2_ctx* ctx, int form, int total, int cap, const std::vector<int32_t>& cw, std::vector<WaveGroup>* groups) {
  (void)ctx;
  if (total <= cap) {
    groups->push_back(WaveGroup{0, total});
    return form;
  }
  if (cw.size() < 3) return 0;  // one component that does not fit: stream it
  // Groups of about equal size (the per-step time of a group grows with its waves, and a small last group would run
  // at low occupancy): cut at the component boundaries nearest to k * total / n_groups, never beyond what the chip
  // holds; if the components are too uneven for that, fall back to filling each group greedily.
  for (size_t c = 0; c + 1 < cw.size(); ++c)
    if (cw[c + 1] - cw[c] > cap) return 0;  // a single component larger than the chip
  const int n_groups = (total + cap - 1) / cap;
  bool ok = true;
  {
    size_t c = 0;
    int begin = cw[0];
    for (int gi = 1; gi <= n_groups && ok; ++gi) {
      const long ideal = cw[0] + (long)gi * total / n_groups;
      size_t e = c + 1;  // at least one component per group
      while (e + 1 < cw.size() && cw[e] < ideal) ++e;
      if (gi == n_groups) e = cw.size() - 1;
      while (e > c + 1 && cw[e] - begin > cap) --e;
      if (cw[e] - begin > cap) ok = false;
      groups->push_back(WaveGroup{begin, cw[e] - begin});
      begin = cw[e], c = e;
      if (c + 1 >= cw.size() && gi < n_groups) break;
    }
    if (ok && begin != cw.back()) ok = false;  // balanced cuts did not cover everything within n_groups groups
  }
  if (!ok) {
    groups->clear();
    int begin = cw[0];
    for (size_t c = 0; c + 1 < cw.size(); ++c) {
      if (cw[c + 1] - begin > cap) {
        groups->push_back(WaveGroup{begin, cw[c] - begin});
        begin = cw[c];
      }
    }
    groups->push_back(WaveGroup{begin, cw.back() - begin});
  }
  return form;
}
bool persistent_eligible(flame_nltgv2_ctx* ctx, int n) {
  std::vector<WaveGroup> g;
  return plan_persistent(ctx, n, &g) != 0;
}

bool same_params(const flame_nltgv2_params& a, const flame_nltgv2_params& b) {
  return std::memcmp(&a, &b, sizeof(a)) == 0;
}

int enqueue_fused_eager(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n, int parity, int unroll,
                        int wpb, bool prev_on_last) {
  const SolverParams sp = to_sp(p);
  for (int it = 0; it < n; ++it) {
    LAUNCHCHK(ctx, launch_fused_step(ctx->f, sp, parity ^ (it & 1), prev_on_last && it == n - 1, unroll, wpb,
                                     ctx->stream));
  }
  return 0;
}

int get_graph(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n, int parity, int unroll, int wpb,
              hipGraphExec_t* out) {
  for (auto& g : ctx->graphs) {
    if (g.n == n && g.parity == parity && g.unroll == unroll && g.wpb == wpb && g.topo == ctx->topo &&
        g.gen == ctx->buf_gen && same_params(g.params, *p)) {
      g.stamp = ++ctx->stamp;
      *out = g.exec;
      return 0;
    }
  }
  hipGraph_t graph = nullptr;
  HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
  int rc = enqueue_fused_eager(ctx, p, n, parity, unroll, wpb, true);
  hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
  if (rc != 0) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  HIPCHK(ctx, e);
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  HIPCHK(ctx, e);
  if ((int)ctx->graphs.size() >= kMaxCachedGraphs) {
    size_t victim = 0;
    for (size_t i = 1; i < ctx->graphs.size(); ++i)
      if (ctx->graphs[i].stamp < ctx->graphs[victim].stamp) victim = i;
    (void)hipGraphExecDestroy(ctx->graphs[victim].exec);
    ctx->graphs.erase(ctx->graphs.begin() + (long)victim);
  }
  CachedGraph cg;
  cg.exec = exec, cg.n = n, cg.parity = parity, cg.unroll = unroll, cg.wpb = wpb, cg.topo = ctx->topo;
  cg.gen = ctx->buf_gen;
  cg.params = *p, cg.stamp = ++ctx->stamp;
  ctx->graphs.push_back(cg);
  *out = exec;
  return 0;
}

// Builds (if needed) every hipGraph a run of n steps will replay, without running anything: keeps
// graph instantiation out of timed regions.
int prepare_run(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n) {
  if (ctx->opt_solver != 0 || !ctx->opt_use_graph || persistent_eligible(ctx, n)) return 0;
  int unroll, wpb;
  pick_config(ctx, &unroll, &wpb);
  int parity = ctx->parity;
  hipGraphExec_t exec;
  if (n >= kGraphChunk) {
    int rc = get_graph(ctx, p, kGraphChunk, parity, unroll, wpb, &exec);
    if (rc) return rc;
  }
  const int rem = n % kGraphChunk;
  if (rem >= 4) {
    int rc = get_graph(ctx, p, rem, parity, unroll, wpb, &exec);
    if (rc) return rc;
  }
  return 0;
}

PhotoFuse photo_target(const flame_nltgv2_ctx* ctx) {
  PhotoFuse f;
  std::memset(static_cast<void*>(&f), 0, sizeof f);  // padding too: the block is compared bytewise before it is re-sent
  f.graph_scale = 1.0f;
  if (!ctx->photo_fused || ctx->img_rows == 0) return f;
  f.pos = (const float2*)ctx->pos.p;
  f.ref = (const uint8_t*)ctx->img_ref.p, f.cmp = (const uint8_t*)ctx->img_cmp.p;
  f.err = (float*)ctx->photo_err.p;
  f.geo = ctx->photo_geo;
  f.graph_scale = ctx->photo_scale;
  f.rows = ctx->img_rows, f.cols = ctx->img_cols, f.step = ctx->img_step, f.border = ctx->photo_border;
  return f;
}

// after a run on a path whose kernels have no photometric epilogue
int enqueue_photo_sweep(flame_nltgv2_ctx* ctx, bool packed_current) {
  const PhotoFuse f = photo_target(ctx);
  if (!f.err) return 0;
  if (packed_current) {
    LAUNCHCHK(ctx, launch_photo_residual_packed(ctx->f, f, ctx->stream));
  } else {
    LAUNCHCHK(ctx, launch_photo_residual(ctx->c, f.graph_scale, f.geo, f.ref, f.cmp, f.rows, f.cols, f.step, f.border, f.err,
                                         ctx->stream));
  }
  return 0;
}

// Record placement, once per context: every page of the pool is timed for all 28 pairs of XCDs (k_place_calibrate) and
// ranked per pair.  ~3 ms, at the first run that can use it.  Anything unexpected (a pair missing because two blocks
// shared an XCD, a wait that expired because the GPU is busy with someone else's work) switches placement off for this
// context: the records then keep their linear places.
// The page ranking belongs to the POOL it was measured on (which page is close to which pair of XCDs is a property of the pages'
// physical addresses: a ranking applied to another allocation is a random page choice -- 1.35 instead of 1.16 us per iteration at
// 1280x720).  Pools therefore outlive their contexts: a context that ends hands its pool, with its ranking, to a per-device list, the
// next context of the device takes it from there (a fresh context's first run: 4.6 -> 0.6 ms; eight contexts alive at once measure
// eight pools, once).  A pool is cleared when it changes hands.
namespace {
struct PlacePool {
  int device = -1;
  void* mem = nullptr;  // (2 * kPlacePages + 1) pages
  std::vector<uint16_t> rank;
  float best = 0.f, mean = 0.f, worst = 0.f;
};
// (never destroyed: a context may still be released while the process exits, after the static destructors have run)
std::mutex& g_place_mu = *new std::mutex;
std::vector<PlacePool>& g_place_free = *new std::vector<PlacePool>;
std::vector<int>& g_place_seen = *new std::vector<int>;  // devices whose first context has been created
}  // namespace

// flame_nltgv2_create: is NOW the time to rank this context's pages?  Yes for the first context of a device (the process's warm-up is
// paid there anyway) and whenever a ranked pool waits to be taken over (no measurement at all).  A later context beside live ones
// ranks lazily, at its first run that can use placement: the measurement is ~3 ms of cross-XCD spin kernels that would compete for
// wave slots with another context's free-running solver (a SolverLoop in device mode) -- and a context that never runs a placed form
// never pays it.  (Pools, 512 KB of device memory each, are kept until the process exits: at most as many as contexts were alive at once.)
bool place_calibrate_at_create(int device) {
  std::lock_guard<std::mutex> lock(g_place_mu);
  bool first = true;
  for (int d : g_place_seen) first = first && d != device;
  if (first) g_place_seen.push_back(device);
  for (const PlacePool& e : g_place_free)
    if (e.device == device) return true;
  return first;
}

// flame_nltgv2_destroy: the context's pool goes back to the list (its stream has been synchronised: nobody writes it any more)
void place_pool_release(flame_nltgv2_ctx* ctx) {
  if (!ctx->place_pool.p || ctx->place_state != 1 || ctx->place_rank_host.empty()) return;  // (an unranked pool is freed with the context)
  PlacePool e;
  e.device = ctx->device, e.mem = ctx->place_pool.p, e.rank.swap(ctx->place_rank_host);
  e.best = ctx->place_best_us, e.mean = ctx->place_mean_us, e.worst = ctx->place_worst_us;
  ctx->device_bytes -= ctx->place_pool.cap;
  ctx->place_pool = DevBuf{};
  ctx->place_base = nullptr, ctx->place_state = 0;
  std::lock_guard<std::mutex> lock(g_place_mu);
  g_place_free.push_back(std::move(e));
}

int place_calibrate(flame_nltgv2_ctx* ctx) {
  ctx->place_state = -1;
  constexpr int P = kPlacePages, kIters = 12;
  const size_t pool_bytes = (size_t)2 * P * 4096;
  if (!ctx->place_pool.p && !std::getenv("FLAME_NLTGV2_RECALIBRATE")) {
    PlacePool e;
    {
      std::lock_guard<std::mutex> lock(g_place_mu);
      for (size_t i = 0; i < g_place_free.size(); ++i)
        if (g_place_free[i].device == ctx->device) {
          e = std::move(g_place_free[i]);
          g_place_free.erase(g_place_free.begin() + (long)i);
          break;
        }
    }
    if (e.mem) {
      ctx->place_pool.p = e.mem, ctx->place_pool.cap = pool_bytes + 4096;
      ctx->device_bytes += ctx->place_pool.cap;
      ctx->place_best_us = e.best, ctx->place_mean_us = e.mean, ctx->place_worst_us = e.worst;
      ctx->place_rank_host.swap(e.rank);
      int rc = ensure(ctx, ctx->place_rank, sizeof(uint16_t) * 2 * 64 * P);
      if (!rc) rc = ensure(ctx, ctx->place_fill, sizeof(int) * (2 * P + 16 + 128));
      if (rc) return rc;
      ctx->place_base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ctx->place_pool.p) + 4095) & ~uintptr_t(4095));
      HIPCHK(ctx, hipMemsetAsync(ctx->place_base, 0, pool_bytes, ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(ctx->place_fill.p, 0, sizeof(int) * (2 * P + 16 + 128), ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(ctx->place_rank.p, ctx->place_rank_host.data(), sizeof(uint16_t) * ctx->place_rank_host.size(), hipMemcpyHostToDevice,
                                 ctx->stream));
      HIPCHK(ctx, wait_solver_stream(ctx));
      ctx->place_state = 1;
      return 0;
    }
  }
  int rc = ensure(ctx, ctx->place_pool, pool_bytes + 4096);
  if (!rc) rc = ensure(ctx, ctx->place_meas, sizeof(unsigned) * 64 * 2 * P + sizeof(int) * 64 + sizeof(int));
  if (!rc) rc = ensure(ctx, ctx->place_rank, sizeof(uint16_t) * 2 * 64 * P);
  if (!rc) rc = ensure(ctx, ctx->place_fill, sizeof(int) * (2 * P + 16 + 128));  // (+ the rotation word of the launches, + k_place_assign's cursors)
  if (rc) return rc;
  ctx->place_base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ctx->place_pool.p) + 4095) & ~uintptr_t(4095));
  unsigned* d_out = (unsigned*)ctx->place_meas.p;
  int* d_xcc = (int*)(d_out + (size_t)64 * 2 * P);
  int* d_fail = d_xcc + 64;
  HIPCHK(ctx, hipMemsetAsync(ctx->place_base, 0, pool_bytes, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->place_meas.p, 0, sizeof(unsigned) * 64 * 2 * P + sizeof(int) * 65, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->place_fill.p, 0, sizeof(int) * (2 * P + 16 + 128), ctx->stream));
  LAUNCHCHK(ctx, launch_place_calibrate(ctx->place_base, 2 * P, kIters, d_out, d_xcc, d_fail, ctx->stream));
  std::vector<unsigned> out((size_t)64 * 2 * P);
  int xcc[65];
  HIPCHK(ctx, hipMemcpyAsync(out.data(), d_out, sizeof(unsigned) * out.size(), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(xcc, d_xcc, sizeof(int) * 65, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));
  HIPCHK(ctx, hipMemsetAsync(ctx->place_base, 0, pool_bytes, ctx->stream));  // (tags of the calibration: gone)
  if (xcc[64] != 0) return 0;  // a wait expired
  std::vector<unsigned> lat((size_t)64 * 2 * P, 0u);  // [8 a + b][page]: a record written on XCD a, seen on XCD b
  bool have[64] = {false};
  for (int b = 8; b < 64; ++b) {
    const int x = b & 7, j = b >> 3, pb = ((8 - j) & 7) * 8 + ((x + j) & 7);
    const int from = xcc[pb], to = xcc[b];  // block b timed what its partner sent
    if (from < 0 || from > 7 || to < 0 || to > 7 || from == to) return 0;
    for (int pg = 0; pg < 2 * P; ++pg) {
      const unsigned t = out[(size_t)b * 2 * P + pg];
      if (t == 0u || t > (1u << 24)) return 0;  // (a clock that ran backwards between two XCDs would show up here)
      lat[(size_t)(from * 8 + to) * 2 * P + pg] = t;
    }
    have[from * 8 + to] = true;
  }
  for (int a = 0; a < 8; ++a)
    for (int b = 0; b < 8; ++b)
      if (a != b && !have[a * 8 + b]) return 0;
  std::vector<uint16_t> rank((size_t)2 * 64 * P);
  double best = 0.0, mean = 0.0, worst = 0.0;
  for (int par = 0; par < 2; ++par)
    for (int c = 0; c < 64; ++c) {
      uint16_t* r = &rank[((size_t)par * 64 + c) * P];
      for (int t = 0; t < P; ++t) r[t] = (uint16_t)t;
      if (c / 8 == c % 8) continue;
      const unsigned* l = &lat[(size_t)c * 2 * P + (size_t)par * P];
      std::stable_sort(r, r + P, [&](uint16_t u, uint16_t v) { return l[u] < l[v]; });
      double m = 0.0;
      for (int t = 0; t < P; ++t) m += l[t];
      best += l[r[0]], worst += l[r[P - 1]], mean += m / P;
    }
  if (std::getenv("FLAME_NLTGV2_TRACE")) {  // the best page of every ordered pair of XCDs, in us (parity 0)
    for (int a = 0; a < 8; ++a) {
      std::fprintf(stderr, "[flame_nltgv2] hand-off from XCD %d to 0..7, best page (us):", a);
      for (int b = 0; b < 8; ++b) {
        if (a == b) { std::fprintf(stderr, "   -  "); continue; }
        const unsigned* l = &lat[(size_t)(a * 8 + b) * 2 * P];
        unsigned best_t = ~0u, worst_t = 0u;
        for (int t = 0; t < P; ++t) best_t = std::min(best_t, l[t]), worst_t = std::max(worst_t, l[t]);
        std::fprintf(stderr, " %.3f/%.3f", best_t / (100.0 * kIters), worst_t / (100.0 * kIters));
      }
      std::fprintf(stderr, "\n");
    }
  }
  const double to_us = 1.0 / (100.0 * kIters) / (2.0 * 56.0);  // 100 MHz ticks of kIters hand-offs; mean of 2 x 56 classes
  ctx->place_best_us = (float)(best * to_us), ctx->place_mean_us = (float)(mean * to_us), ctx->place_worst_us = (float)(worst * to_us);
  HIPCHK(ctx, hipMemcpyAsync(ctx->place_rank.p, rank.data(), sizeof(uint16_t) * rank.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));  // (`rank` is pageable and leaves scope)
  ctx->place_state = 1;
  ctx->place_rank_host.swap(rank);  // (kept: the pool and its ranking go to the next context of this device, place_pool_release)
  return 0;
}

// ... and once per topology: the records that are read across XCDs get their places (k_place_assign)
int place_records(flame_nltgv2_ctx* ctx, int per_xcd) {
  const size_t stride = records_capacity(ctx->L);
  int rc = ensure(ctx, ctx->place_rec_off, sizeof(int32_t) * 2 * stride);
  if (!rc) rc = ensure(ctx, ctx->place_patch, sizeof(int32_t) * stride + stride);  // [patch of a record | its class (1 byte)]
  if (rc) return rc;
  // (records beyond the walk -- the exchange buffers are sized for the packed vertex count -- are never read through this table)
  LAUNCHCHK(ctx, launch_place_records(ctx->c, ctx->f, per_xcd, (const int32_t*)ctx->order_m.p, (const int32_t*)ctx->rid_of.p,
                                      (int32_t*)ctx->place_patch.p, (int8_t*)((int32_t*)ctx->place_patch.p + stride),
                                      (const uint16_t*)ctx->place_rank.p, kPlacePages, (int*)ctx->place_fill.p,
                                      (int32_t*)ctx->place_rec_off.p, (int)stride, ctx->stream));
  ctx->place_topo = ctx->topo, ctx->place_per_xcd = per_xcd;
  return 0;
}

// An open run: ONE launch of the patch-per-wave form or of the two-half-edges form.  Their OPEN instances keep fewer patches resident than
// the plain ones (24 instead of 28 per CU; 16 instead of 20: tests/test_abi.py): graphs of at most 20 / 14 patches per CU.
static bool open_run_applies(const flame_nltgv2_ctx* ctx, int form, const std::vector<WaveGroup>& groups) {
  if (groups.size() != 1) return false;
  const int cus = ctx->prop.multiProcessorCount;
  return (form == 3 && groups[0].count <= 20 * cus) || (form == 4 && groups[0].count <= 14 * cus);
}

int enqueue_run(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n) {
  if (n <= 0) return 0;
  if (ctx->open_inflight) {  // nothing is chained behind an open run: it would wait for the run's upper bound
    const int rc0 = finish(ctx);
    if (rc0) return rc0;
  }
  if (ctx->opt_solver == 1) {  // canonical 4-sweep path
    int rc = ensure_canon(ctx);  // (settles a pending persistent run first)
    if (rc) return rc;
    const SolverParams sp = to_sp(p);
    for (int it = 0; it < n; ++it) {
      LAUNCHCHK(ctx, launch_save_prev(ctx->c, ctx->stream));
      LAUNCHCHK(ctx, launch_dual(ctx->c, sp, ctx->stream));
      LAUNCHCHK(ctx, launch_primal(ctx->c, sp, ctx->stream));
      LAUNCHCHK(ctx, launch_extragradient(ctx->c, sp, ctx->stream));
    }
    ctx->fused_valid = false;
    ctx->last_run_path = 4;
    if (ctx->replaying == 0) ctx->iters_total += n;
    if (ctx->export_ptr) LAUNCHCHK(ctx, launch_export(ctx->c, ctx->f, false, ctx->export_scale, ctx->export_ptr, ctx->stream));
    return enqueue_photo_sweep(ctx, false);
  }
  if (ctx->opt_mesh_state == 1 && ctx->canon_valid) {
    // FLAME_NLTGV2_OPT_MESH_STATE = 1: the canonical arrays stand as they are until the next unpack (the runs work on the packed
    // copies); interpolate_mesh_begin may read them beside the runs once everything enqueued so far -- their writers -- is through
    HIPCHK(ctx, hipEventRecord(ctx->ev_snap, ctx->stream));
    ctx->snap_topo = ctx->topo;
  }
  int rc = ensure_fused(ctx);
  if (rc) return rc;
  if (ctx->pending.active) {  // chaining onto an unchecked persistent run
    if (ctx->pending.ops.size() >= kMaxChain) {
      rc = finish(ctx);
    } else {
      rc = snapshot_chain_start(ctx);
    }
    if (rc) return rc;
  }
  int unroll, wpb;
  pick_config(ctx, &unroll, &wpb);
  std::vector<WaveGroup> groups;
  int tv_lds = 0;
  const int form = plan_persistent(ctx, n, &groups, &tv_lds, /*consume=*/true);
  // an open run: the patch-per-wave form as ONE launch of its plain instance, the first run of a chain -- anything else is run as asked
  // for (n iterations), the caller sees that from flame_nltgv2_run_open's `opened`
  const bool open_run = ctx->want_open != 0 && open_run_applies(ctx, form, groups) && !ctx->pending.active && ctx->opt_probe == 0 &&
                        ctx->opt_verify == 0 && ctx->replaying == 0 && (n & 1) == 0 && ctx->h_stop != nullptr && ctx->ctl_stream != nullptr;
  if (ctx->want_open != 0 && !open_run) {  // flame_nltgv2_run_open on a graph / a configuration it does not apply to: nothing is run
    ctx->want_open = 0;
    return 0;
  }
  if (form != 0) {
    // tags must stay unique: clear the record buffers long before the 28-bit tag of the XCC table wraps -- and when the
    // form changes (the forms lay the buffers out differently: one's XCC table is another's record area)
    if ((uint64_t)ctx->tag_next + (uint64_t)n >= 0x07ff0000ull || (ctx->xbuf_form != 0 && ctx->xbuf_form != form)) {
      const size_t bytes = kXbufBytesPerVertex * records_capacity(ctx->L);
      HIPCHK(ctx, hipMemsetAsync(ctx->xbuf.p, 0, bytes, ctx->stream));
      if (ctx->place_base) {
        HIPCHK(ctx, hipMemsetAsync(ctx->place_base, 0, (size_t)2 * kPlacePages * 4096, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync((int*)ctx->place_fill.p + 2 * kPlacePages, 0, 64, ctx->stream));
      }
      HIPCHK(ctx, hipMemsetAsync((int*)ctx->err.p + 12, 0, 2 * sizeof(int), ctx->stream));  // (an open run's words count from its tag0)
      ctx->tag_next = 1;
    }
    ctx->xbuf_form = form;
    // a fresh first tag per run: records left by earlier runs (whose state may since have been changed
    // by per-step launches or host uploads) can never satisfy a wait of this one
    const uint32_t tag0 = ctx->tag_next + 2;
    // (topology, form, kernel instance): a new instance -- other registers, other LDS -- gets a cooperative first launch
    const uint64_t key = ctx->topo * 1024 + (uint64_t)form * 64 + (uint64_t)tv_lds * 32 + (ctx->opt_verify != 0 ? 16 : 0) + (ctx->opt_probe != 0 ? 8 : 0) +
                         (ctx->opt_dual == 2 ? 4 : ctx->opt_dual == 1 ? 2 : 0) + (ctx->opt_xcds > 0 ? 1 : 0);
    const RunTail* tail_dev = nullptr;
    {  // standing outputs: the small block the kernels read in their epilogue, (re)sent when no slot holds it
      RunTail want;
      std::memset(static_cast<void*>(&want), 0, sizeof want);
      want.export_out = ctx->export_ptr, want.export_scale = ctx->export_scale;
      if (open_run) {  // the request word: device memory; a request names the run it is for (its tag0), so the word is never cleared
        const bool fresh = ctx->stop_dev.p == nullptr;
        rc = ensure(ctx, ctx->stop_dev, 64);
        if (rc) return rc;
        if (fresh) HIPCHK(ctx, hipMemsetAsync(ctx->stop_dev.p, 0, 64, ctx->stream));
        ctx->open_tag0 = tag0;
      }
      want.stop_req = open_run ? (const unsigned*)ctx->stop_dev.p : nullptr;
      const PhotoFuse pf = photo_target(ctx);
      std::memcpy(static_cast<void*>(&want.photo), &pf, sizeof pf);
      if (form == 3 && std::getenv("FLAME_NLTGV2_TRACE")) {
        rc = ensure(ctx, ctx->progress, 2 * sizeof(unsigned) * (size_t)ctx->L.wg_count);  // [how far | started when]
        if (rc) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->progress.p, 0, 2 * sizeof(unsigned) * (size_t)ctx->L.wg_count, ctx->stream));
        want.progress = (unsigned*)ctx->progress.p;
      }
      constexpr size_t kTailStride = (sizeof(RunTail) + 255) / 256 * 256;
      rc = ensure(ctx, ctx->run_tail, kTailStride * flame_nltgv2_ctx::kTailSlots);
      if (rc) return rc;
      int slot = -1;
      for (int i = 0; i < flame_nltgv2_ctx::kTailSlots && slot < 0; ++i)
        if (ctx->tail_valid[i] && std::memcmp(&want, &ctx->tail_sent[i], sizeof(RunTail)) == 0) slot = i;
      if (slot < 0) {  // (ordered on the solver's stream behind every launch that still reads the slot)
        slot = ctx->tail_next, ctx->tail_next = (ctx->tail_next + 1) % flame_nltgv2_ctx::kTailSlots;
        // pageable source: the runtime stages it during the call, so `want` may go out of scope
        HIPCHK(ctx, hipMemcpyAsync((char*)ctx->run_tail.p + kTailStride * (size_t)slot, &want, sizeof(RunTail), hipMemcpyHostToDevice, ctx->stream));
        ctx->tail_sent[slot] = want;
        ctx->tail_valid[slot] = true;
      }
      tail_dev = (const RunTail*)((const char*)ctx->run_tail.p + kTailStride * (size_t)slot);
    }
    int e = 0;
    for (const WaveGroup& gr : groups) {
      // the run's last launch carries an event as its own completion signal (flame_nltgv2_stream_wait_run, _runs_in_flight); a cooperative launch
      // (the first of a topology) cannot: the wait then records the event behind it
      // (only a run enqueued by flame_nltgv2_run_async binds one: run_ev_pick < 0 for the blocking run(), finish()'s replays, the warm-up --
      //  their launches must not re-arm an event that still stands for an earlier run in flight)
      const bool carries = &gr == &groups.back() && ctx->coop_checked_key == key && ctx->run_ev_pick >= 0 && ctx->ev_run[ctx->run_ev_pick] != nullptr;
      ctx->f.stop_event = carries ? ctx->ev_run[ctx->run_ev_pick] : nullptr;
      const int pw = gr.count <= 4 * ctx->prop.multiProcessorCount ? 1 : 4;  // waves per workgroup
      // same-XCD exchange through L2: with the waves laid out along the Morton curve it wins at every size
      // (measured per step: 640x480 -16 %, 1280x720 -23 %, 1080p -18 %, 7-frame batch -20 %, 15 frames -19 %)
      const int dual = (ctx->opt_dual == 2 || (ctx->opt_dual == 1 && gr.count > kDualMinWavesPerCu * ctx->prop.multiProcessorCount) ? 1 : 0) |
                       (ctx->opt_verify == 2 ? 6 : ctx->opt_verify == 1 ? 2 : 0);  // bits 1, 2: record verification, its test hook
      // A graph small enough runs on ONE XCD (32 CUs) entirely: every exchange stays in that XCD's L2 -- while its CUs get at
      // most two patches each (measured: 48 patches 0.98 against 1.21 us per step on all eight, 208 patches 1.53 against 1.33)
      const int cus_per_xcd = ctx->prop.multiProcessorCount / 8;
      const bool one_xcd = form == 3 && gr.count <= 2 * cus_per_xcd;
      const int xcds = ctx->opt_xcds > 0 ? ctx->opt_xcds : one_xcd ? 1 : 8;
      const int presleep = ctx->opt_presleep > 0 ? ctx->opt_presleep - 1 : kPreSleepTv;  // (the patch form has its own, below)
      // (test hook: a fault of 2^22 + n also hits the persistent replay of the chain, so that the per-step rung below it is exercised)
      const unsigned spins_arg = (ctx->opt_fault > 0 && (ctx->replaying == 0 || ctx->opt_fault >= (1 << 22)))
                                     ? (0x80000000u | (unsigned)(ctx->opt_fault & ((1 << 22) - 1))) : kMaxSpins;
      if (form == 3) {
        // pacing: none where a CU holds few patches (a poll costs nothing there and a pause only delays the hand-off); at
        // high residency the polls of ~20 waves per CU saturate the L2s' request ports and the fabric and it is the hand-off
        // itself that slows down (probe: 0.72 us at 4 patches per CU, 1.03 at 15; 26 per CU unpaced: 50 us per step) -- a
        // pause before the first poll and between rounds then wins (tools/pv_big.py sweeps, profiles/r03_pv_dense.txt)
        const bool dense = gr.count > kPvPaceAbovePerCu * ctx->prop.multiProcessorCount;
        const bool denser = gr.count > kPvPaceMoreAbovePerCu * ctx->prop.multiProcessorCount;
        const int dense_gap = denser ? kPvDenserGap : kPvDenseGap, dense_pre = denser ? kPvDenserPreSleep : kPvDensePreSleep;
        const int gap = ctx->opt_poll_gap > 0 ? ctx->opt_poll_gap - 1 : dense ? (3 | ((dense_gap - 1) << 4)) : kPvPollGap;
        ctx->f.wg_poll_gap = gap | ((ctx->opt_presleep > 0 ? ctx->opt_presleep - 1 : dense ? dense_pre : kPvPreSleep) << 8);
        ctx->f.rec_off = nullptr, ctx->f.place_pool = nullptr;
        if (ctx->opt_place && groups.size() == 1 && gr.begin == 0 && xcds == 8 && (dual & 1)) {
          if (ctx->place_state == 0) {
            rc = place_calibrate(ctx);
            if (rc) return rc;
          }
          const int per_xcd = (gr.count + xcds - 1) / xcds;
          if (ctx->place_state == 1 && (ctx->place_topo != ctx->topo || ctx->place_per_xcd != per_xcd)) {
            rc = place_records(ctx, per_xcd);
            if (rc) return rc;
          }
          if (ctx->place_state == 1) {
            ctx->f.place_pool = ctx->place_base, ctx->f.rec_off = (const int32_t*)ctx->place_rec_off.p;
            ctx->f.rec_off_stride = (int)records_capacity(ctx->L);
            ctx->f.rot_word = (unsigned*)ctx->place_fill.p + 2 * kPlacePages;
          }
        }
        ctx->f.probe = nullptr;
        if (ctx->opt_probe) {  // [patch][step][8 words]
          const size_t words = (size_t)ctx->L.wg_count * (size_t)n * 8;
          rc = ensure(ctx, ctx->probe, words * sizeof(unsigned));
          if (rc) return rc;
          ctx->f.probe = (unsigned*)ctx->probe.p;
          ctx->probe_words = words;
          HIPCHK(ctx, hipMemsetAsync(ctx->probe.p, 0, words * sizeof(unsigned), ctx->stream));  // (idle instances write nothing)
        }
      }
      if (form == 4) {  // two half-edges per lane: its own pacing (swept: profiles/r03_pv2.txt)
        const bool dense = gr.count > kPv2PaceAbovePerCu * ctx->prop.multiProcessorCount;
        const int gap = ctx->opt_poll_gap > 0 ? ctx->opt_poll_gap - 1 : dense ? (3 | ((kPv2DenseGap - 1) << 4)) : kPvPollGap;
        const int poll_gap = gap | ((ctx->opt_presleep > 0 ? ctx->opt_presleep - 1 : dense ? kPv2DensePreSleep : kPvPreSleep) << 8);
        ctx->f.open_run = open_run ? 1 : 0;
        e = launch_persistent_pv2(ctx->f, ctx->pv2_args, to_sp(p), gr.begin, gr.count, ctx->parity, tag0, n, spins_arg, poll_gap, dual,
                                  tail_dev, cooperative_allowed() && ctx->coop_checked_key != key, ctx->stream);
        ctx->f.open_run = 0;
        if (e != 0) break;
        continue;
      }
      ctx->f.open_run = open_run ? 1 : 0;
      e = launch_persistent_run(ctx->f, to_sp(p), form, gr.begin, gr.count, ctx->parity, tag0, n, pw, spins_arg, presleep, dual,
                                tv_lds, xcds, tail_dev, cooperative_allowed() && ctx->coop_checked_key != key, ctx->stream);
      ctx->f.open_run = 0;
      if (e != 0) break;
    }
    ctx->run_event_bound = e == 0 && ctx->f.stop_event != nullptr;
    ctx->f.stop_event = nullptr;
    ctx->tag_next = tag0 + (uint32_t)n;
    if (e == 0) {
      // The kernels wrote (will write) hq / vstate / bar into the other copies: make those current.  finish() takes
      // this back if the run reports a timeout.
      if (!ctx->pending.active) {
        ctx->pending.active = true, ctx->pending.snapshotted = false;
        ctx->pending.ops.clear();
        ctx->pending.parity_before = ctx->parity, ctx->pending.have_prev_before = ctx->have_prev;
      }
      {
        flame_nltgv2_ctx::PendingOp op;
        op.kind = 0, op.params = *p, op.n = n;
        op.open = open_run, op.tag0 = tag0;
        op.dst = ctx->export_ptr, op.scale = ctx->export_scale;  // the standing export target THIS run was enqueued with
        ctx->pending.ops.push_back(op);
      }
      ctx->open_inflight = open_run, ctx->open_stop_sent = false;
      if (open_run) ctx->want_open = 2;
      if (!open_run && ctx->replaying == 0) ctx->iters_total += n;  // (an open run: counted by finish(), which learns how far it went)
      std::swap(ctx->hq, ctx->hq_alt);
      std::swap(ctx->vstate, ctx->vstate_alt);
      ctx->buf_gen ^= 1;
      refresh_args(ctx);
      ctx->coop_checked_key = key;
      ctx->last_run_path = form == 4 ? 7 : form == 3 ? 6 : 5;
      ctx->last_run_groups = (int)groups.size();
      ctx->last_run_waves_per_cu = 0;
      for (const WaveGroup& gr : groups) ctx->last_run_waves_per_cu = std::max(ctx->last_run_waves_per_cu, (gr.count + ctx->prop.multiProcessorCount - 1) / ctx->prop.multiProcessorCount);
      ctx->parity ^= 1;
      ctx->have_prev = true;
      ctx->canon_valid = false;
      return 0;
    }
    if (std::getenv("FLAME_NLTGV2_TRACE")) std::fprintf(stderr, "[flame_nltgv2] persistent launch refused: hip error %d (%s), form %d, pv_occ %d\n", e, hipGetErrorString((hipError_t)e), form, ctx->pv_occ);
    // e.g. cooperative launch too large.  Groups already enqueued write into the other copies only: the current
    // state is intact, the steps are done on the one-launch-per-step path below.  Let those groups drain first and
    // forget what they reported (their waits expire without the missing groups): that is not a failure of a run.
    (void)hipGetLastError();
    ctx->persist_refused_topo = ctx->topo;  // do not try again for this topology
    if (groups.size() > 1 && !ctx->pending.active) {
      HIPCHK(ctx, hipMemsetAsync(ctx->abort_flag.p, 0xff, sizeof(int), ctx->stream));  // tells them to leave at once
      HIPCHK(ctx, wait_solver_stream(ctx));
      HIPCHK(ctx, hipMemsetAsync(ctx->err.p, 0, kErrBytes, ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(ctx->abort_flag.p, 0, sizeof(int), ctx->stream));
    }
  }
  if (ctx->pending.active) {
    flame_nltgv2_ctx::PendingOp op;
    op.kind = 0, op.params = *p, op.n = n;
    op.dst = ctx->export_ptr, op.scale = ctx->export_scale;
    ctx->pending.ops.push_back(op);
  }
  int left = n;
  while (left > 0) {
    const int chunk = left >= kGraphChunk ? kGraphChunk : left;
    if (ctx->opt_use_graph && chunk >= 4) {
      hipGraphExec_t exec = nullptr;
      rc = get_graph(ctx, p, chunk, ctx->parity, unroll, wpb, &exec);
      if (rc) return rc;
      HIPCHK(ctx, hipGraphLaunch(exec, ctx->stream));
      ctx->last_run_path = 2;
    } else {
      rc = enqueue_fused_eager(ctx, p, chunk, ctx->parity, unroll, wpb, true);
      if (rc) return rc;
      ctx->last_run_path = 3;
    }
    ctx->parity ^= (chunk & 1);
    left -= chunk;
  }
  ctx->have_prev = true;
  ctx->canon_valid = false;
  if (ctx->replaying == 0) ctx->iters_total += n;
  if (ctx->export_ptr) LAUNCHCHK(ctx, launch_export(ctx->c, ctx->f, true, ctx->export_scale, ctx->export_ptr, ctx->stream));
  return enqueue_photo_sweep(ctx, true);
}

// FLAME_NLTGV2_TRACE: which wait of the persistent run expired first, and how far every patch had got
void trace_expired_wait(flame_nltgv2_ctx* ctx) {
  const int* e = ctx->h_err;
  static const char* const kWhich[] = {"?", "rotation word", "XCC table", "step records"};
  std::fprintf(stderr, "[flame_nltgv2] persistent run taken back (flags %d): first expired wait = %s, patch %d, step %d, lanes waiting %08x%08x, "
               "first of them for record %d (tag seen %u, wanted %u), on XCC %d hw_id %08x\n", e[0], kWhich[e[1] & 3], e[2], e[3], (unsigned)e[5],
               (unsigned)e[4], e[6], (unsigned)e[7], (unsigned)e[8], e[9], (unsigned)e[10]);
  if (!ctx->progress.p || ctx->L.wg_count <= 0) return;
  std::vector<unsigned> pg((size_t)2 * ctx->L.wg_count);
  if (hipMemcpy(pg.data(), ctx->progress.p, sizeof(unsigned) * pg.size(), hipMemcpyDeviceToHost) != hipSuccess) return;
  size_t silent = 0, left = 0;
  unsigned lo = ~0u, hi = 0;
  int first_silent = -1;
  const size_t n_p = (size_t)ctx->L.wg_count;
  {  // when the patches started: all within microseconds of each other if they were co-resident
    unsigned t_lo = ~0u, t_hi = 0;
    size_t started = 0, late = 0;
    for (size_t i = 0; i < n_p; ++i)
      if (pg[n_p + i]) ++started, t_lo = std::min(t_lo, pg[n_p + i]), t_hi = std::max(t_hi, pg[n_p + i]);
    int first_late = -1;
    for (size_t i = 0; i < n_p; ++i)
      if (pg[n_p + i] && pg[n_p + i] - t_lo > 10000u) {
        if (first_late < 0) first_late = (int)i;
        ++late;
      }
    std::fprintf(stderr, "[flame_nltgv2]   %zu of %zu patches started, over %u us; %zu of them more than 10 ms after the first (first such patch: %d)\n",
                 started, n_p, started ? t_hi - t_lo : 0u, late, first_late);
  }
  for (size_t i = 0; i < n_p; ++i) {
    if (pg[i] == 0) {
      if (first_silent < 0) first_silent = (int)i;
      ++silent;
    } else {
      ++left, lo = std::min(lo, pg[i] & 0x7fffffffu), hi = std::max(hi, pg[i] & 0x7fffffffu);
    }
  }
  std::fprintf(stderr, "[flame_nltgv2]   %zu patches left through an expired wait (in steps %u..%u), %zu wrote nothing (finished, or never ran; first: %d)\n",
               left, left ? lo - 1 : 0, left ? hi - 1 : 0, silent, first_silent);
}

int finish(flame_nltgv2_ctx* ctx, bool unpack_behind, bool* unpacked, const std::function<int()>* behind_fn, bool* behind_launched) {
  const bool open_run = ctx->open_inflight;
  if (open_run) {  // an open run ends when it is asked to: its deciding patch looks at this word every kOpenCheck iterations
    request_open_stop(ctx);
    ctx->open_inflight = false, ctx->open_stop_sent = false;
  }
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_err, ctx->err.p, kErrBytes, hipMemcpyDeviceToHost, ctx->stream));
  bool behind = false;
  if (unpack_behind && ctx->pending.active && !ctx->canon_valid && !ctx->replaying) {
    // The caller wants the state in its canonical arrays (ensure_canon): the unpack goes out BEHIND the runs, before the host has seen
    // how they ended -- between the error word's arrival and a kernel launched after it the solver's stream stood empty for 25-30 us,
    // in every call of a frame loop that settles the solver (profiles/r06_cpp_frame_loop.txt).  If the runs did expire the canonical
    // arrays now hold rubbish: the chain is redone below, canon_valid stays false and the caller unpacks once more.
    const int rc = wait_raster(ctx);
    if (rc) return rc;
    LAUNCHCHK(ctx, launch_unpack_state(ctx->c, ctx->f, ctx->parity, ctx->have_prev, ctx->stream));
    behind = true;
    if (behind_fn) {  // (the caller's own launches on the canonical arrays: see ensure_canon)
      const int rc2 = (*behind_fn)();
      if (rc2) return rc2;
      if (behind_launched) *behind_launched = true;
    }
  }
  HIPCHK(ctx, wait_solver_stream(ctx));
  flame_nltgv2_ctx::PendingRun run;
  std::swap(run, ctx->pending);  // (ctx->pending is now inactive and empty)
  if (open_run && !run.ops.empty() && run.ops.back().open) {
    // How far the open run went: its deciding patch left tag0 + n in err[13] on its way out (n = the iteration all patches left at, or
    // the upper bound).  A run that expired has no such number: its replay does the iterations up to the decision if one was made
    // (err[12]), the smallest chunk otherwise -- any number is a legal one for a solver that runs until it is stopped, as long as the
    // caller is told the truth (flame_nltgv2_iterations).
    flame_nltgv2_ctx::PendingOp& op = run.ops.back();
    const uint32_t done = (uint32_t)ctx->h_err[13], decided = (uint32_t)ctx->h_err[12];
    int n_done = op.n;
    if (!(*ctx->h_err & 6)) {
      if (done > op.tag0 && done - op.tag0 <= (uint32_t)op.n) n_done = (int)(done - op.tag0);
    } else {
      n_done = (decided > op.tag0 && decided - op.tag0 <= (uint32_t)op.n) ? (int)(decided - op.tag0) : std::min(op.n, 256);
    }
    op.n = n_done, op.open = false;
    ctx->iters_total += n_done;
    ctx->last_open_iters = n_done;
  }
  if (*ctx->h_err & 6) {
    if (*ctx->h_err & 4) ctx->torn_records_detected++;  // the record verification found a second read that differed
    std::memcpy(ctx->last_expired, ctx->h_err, kErrBytes);
    if (std::getenv("FLAME_NLTGV2_TRACE")) trace_expired_wait(ctx);
    // A neighbour wait of a persistent run expired (its waves were not all resident: the GPU is shared with
    // something that keeps CUs full).  Go back to the state the run -- or the chain of runs enqueued behind it --
    // started from, and do the same steps on the one-launch-per-step path, which needs no co-residency.
    if (!run.active) {  // nothing recorded to go back to
      ctx->have_graph = false;
      return fail(ctx, FLAME_NLTGV2_ERR_TIMEOUT);
    }
    if (run.snapshotted) {
      const size_t n_slots = (size_t)(ctx->L.rows + kRowPad) * kWave, n_packed = (size_t)ctx->L.n_slices * kWave;
      HIPCHK(ctx, hipMemcpyAsync(ctx->hq.p, ctx->snap_hq.p, sizeof(float4) * n_slots, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(ctx->vstate.p, ctx->snap_vstate.p, sizeof(float4) * n_packed, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(ctx->f.bar[run.parity_before], ctx->snap_bar.p, sizeof(float4) * n_packed, hipMemcpyDeviceToDevice,
                                 ctx->stream));
    } else {  // a single run: it wrote the other copies, its input is intact
      std::swap(ctx->hq, ctx->hq_alt);
      std::swap(ctx->vstate, ctx->vstate_alt);
      ctx->buf_gen ^= 1;
      refresh_args(ctx);
    }
    ctx->parity = run.parity_before;
    ctx->have_prev = run.have_prev_before;
    ctx->fused_valid = true, ctx->canon_valid = false;
    const int attempt = ctx->replay_attempt;  // 0: the chain as enqueued expired; 1: so did its persistent replay
    if (attempt == 0) {
      ctx->persist_timeout_streak = (ctx->persist_backoff_topo == ctx->topo) ? std::min(ctx->persist_timeout_streak + 1, 9) : 1;
      ctx->persist_backoff_topo = ctx->topo;
      ctx->persist_backoff_left = 4 << (ctx->persist_timeout_streak - 1);
      if (!(*ctx->h_err & 4)) {
        ctx->timeouts_recovered++;
        if (ctx->last_run_waves_per_cu > kCrowdedWavesPerCu) ctx->crowded_until_topo = ctx->topo + 1 + kCrowdedTopologies;
      }
    } else {
      ctx->replays_per_step++;
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->err.p, 0, kErrBytes, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->abort_flag.p, 0, sizeof(int), ctx->stream));
    // The rung between the persistent forms and the one-launch-per-step path: the chain is first redone persistently, planned for at
    // most kCrowdedWavesPerCu waves per CU (a 1080p frame: the two-half-edges form at 12.6 per CU instead of 25 patches per CU).
    // A torn record (verification on) goes to the per-step path at once: that is not a question of residency.
    ctx->replaying = (attempt == 0 && !(*ctx->h_err & 4) && ctx->opt_persistent != 0 && !std::getenv("FLAME_NLTGV2_REPLAY_PER_STEP")) ? 1 : 2;
    ctx->replay_attempt = attempt + 1;
    int replay_rc = 0;
    // every run is redone with the standing export target it was enqueued with (the caller may have switched the target in the
    // middle of the chain -- the double-buffered rows of a result gather: run k exports to row A, run k + 1 to row B)
    float* const export_now = ctx->export_ptr;
    const float export_scale_now = ctx->export_scale;
    for (const flame_nltgv2_ctx::PendingOp& op : run.ops) {
      if (op.kind == 0) {
        ctx->export_ptr = op.dst, ctx->export_scale = op.scale;
        replay_rc = enqueue_run(ctx, &op.params, op.n);
        ctx->export_ptr = export_now, ctx->export_scale = export_scale_now;
      } else {
        const int e = launch_export(ctx->c, ctx->f, true, op.scale, op.dst, ctx->stream);
        if (e) ctx->last_hip = e, ctx->last_error = FLAME_NLTGV2_ERR_HIP, replay_rc = FLAME_NLTGV2_ERR_HIP;
      }
      if (replay_rc) break;
    }
    ctx->replaying = 0;
    if (replay_rc) {
      ctx->replay_attempt = 0;
      return replay_rc;
    }
    const int rc2 = finish(ctx);  // (a persistent replay that expired as well comes back here once more, for the per-step path)
    ctx->replay_attempt = 0;
    return rc2;
  }
  if (run.active && ctx->last_run_path >= 5) ctx->persist_timeout_streak = 0;  // (a persistent run went through: the next expired one starts over)
  if (*ctx->h_err != 0) {
    // NaN/Inf in a dual variable (the reference's FLAME_ASSERT h:174): reported once; the state stays readable
    // (download_state, costs) and the solve can go on or be re-initialised -- q was clamped to +-1 where it happened
    HIPCHK(ctx, hipMemsetAsync(ctx->err.p, 0, kErrBytes, ctx->stream));
    // (the state stays readable; launches of the caller that went out behind the unpack count as spoiled: the call fails, they are taken back)
    if (unpacked) *unpacked = behind && !(behind_launched && *behind_launched);
    return fail(ctx, FLAME_NLTGV2_ERR_NAN);
  }
  if (unpacked) *unpacked = behind;
  return 0;
}

// Copies the state a chain of asynchronous runs started from aside, once per chain, before anything behind the first
// (still unchecked) persistent run is enqueued: that run read (hq, vstate, bar[parity_before]) and wrote the other
// copies, so the stream-ordered copies below still see its input.
int snapshot_chain_start(flame_nltgv2_ctx* ctx) {
  if (!ctx->pending.active || ctx->pending.snapshotted) return 0;
  const size_t n_slots = (size_t)(ctx->L.rows + kRowPad) * kWave, n_packed = (size_t)ctx->L.n_slices * kWave;
  int rc = ensure(ctx, ctx->snap_hq, sizeof(float4) * n_slots);
  if (!rc) rc = ensure(ctx, ctx->snap_vstate, sizeof(float4) * n_packed);
  if (!rc) rc = ensure(ctx, ctx->snap_bar, sizeof(float4) * n_packed);
  if (rc) return rc;
  CopyTable t;  // (one launch: as three memcpy calls this was 19 us between the first and the second round of every chain)
  t.n = 3;
  t.e[0] = {ctx->snap_hq.p, ctx->hq_alt.p, sizeof(float4) * n_slots};
  t.e[1] = {ctx->snap_vstate.p, ctx->vstate_alt.p, sizeof(float4) * n_packed};
  t.e[2] = {ctx->snap_bar.p, ctx->f.bar[ctx->pending.parity_before], sizeof(float4) * n_packed};
  LAUNCHCHK(ctx, launch_copy_arrays(t, ctx->stream));
  ctx->pending.snapshotted = true;
  return 0;
}

}  // namespace host
}  // namespace flame_hip

extern "C" {

int flame_nltgv2_run_async(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n_iters) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_run_async");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!params_ok(p) || n_iters < 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (n_iters == 0) return 0;  // nothing is enqueued: the two events keep standing for the runs they stand for
  const int idx = ctx->run_ev_last ^ 1;  // the two events take turns: the last two runs can be told apart
  ctx->run_ev_pick = idx;
  ctx->run_event_bound = false;
  rc = enqueue_run(ctx, p, n_iters);
  ctx->run_ev_pick = -1;
  if (rc) return rc;
  bool stands = ctx->run_event_bound;
  if (!stands && ctx->track_runs) {  // (a cooperative first launch, the per-step path: recorded behind it)
    HIPCHK(ctx, hipEventRecord(ctx->ev_run[idx], ctx->stream));
    stands = true;
  }
  ctx->run_ev_valid[idx] = stands;
  ctx->run_ev_last = idx;
  ctx->run_event_seq = ctx->call_seq;
  return 0;
}

int flame_nltgv2_run_open(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int max_iters, int32_t* opened) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_run_open");
  int rc = enter(ctx);
  if (rc) return rc;
  if (opened) *opened = 0;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!params_ok(p) || max_iters < 0 || (max_iters & 1)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (max_iters == 0) return 0;
  if (ctx->pending.active) {  // applicable at all?  Asked BEFORE anything is settled: a loop that falls back to rounds must not have them waited for here
    std::vector<WaveGroup> groups;
    int tv_lds = 0;
    const int form = plan_persistent(ctx, max_iters, &groups, &tv_lds, /*consume=*/false);
    if (!open_run_applies(ctx, form, groups) || ctx->opt_probe != 0 || ctx->opt_verify != 0 || !ctx->h_stop || !ctx->ctl_stream) return 0;
  }
  if (ctx->pending.active) {  // an open run is the first of its chain
    rc = finish(ctx);
    if (rc) return rc;
  }
  const int idx = ctx->run_ev_last ^ 1;
  ctx->run_ev_pick = idx;
  ctx->run_event_bound = false;
  ctx->want_open = 1;
  rc = enqueue_run(ctx, p, max_iters);
  const bool was_open = ctx->want_open == 2;
  ctx->want_open = 0;
  ctx->run_ev_pick = -1;
  if (rc) return rc;
  bool stands = ctx->run_event_bound;
  if (!stands && ctx->track_runs) {
    HIPCHK(ctx, hipEventRecord(ctx->ev_run[idx], ctx->stream));
    stands = true;
  }
  ctx->run_ev_valid[idx] = stands;
  ctx->run_ev_last = idx;
  ctx->run_event_seq = ctx->call_seq;
  if (opened) *opened = was_open ? 1 : 0;
  return 0;
}

int flame_nltgv2_iterations(flame_nltgv2_ctx* ctx, int64_t* total, int32_t* open_in_flight) {
  if (!ctx || !total) return FLAME_NLTGV2_ERR_INVALID_ARG;
  *total = ctx->iters_total;
  if (open_in_flight) *open_in_flight = ctx->open_inflight ? 1 : 0;
  return 0;
}

int flame_nltgv2_stream_wait_run(flame_nltgv2_ctx* ctx, void* hip_stream) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_stream_wait_run");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!hip_stream || (hipStream_t)hip_stream == ctx->stream) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  // the run enqueued by the call right before this one carries (or was followed by) its event; anything else is recorded now
  const int idx = ctx->run_ev_last;
  if (!(ctx->run_ev_valid[idx] && ctx->call_seq == ctx->run_event_seq + 1)) {
    HIPCHK(ctx, hipEventRecord(ctx->ev_run[idx], ctx->stream));
    ctx->run_ev_valid[idx] = true;
  }
  HIPCHK(ctx, hipStreamWaitEvent((hipStream_t)hip_stream, ctx->ev_run[idx], 0));
  return 0;
}

int flame_nltgv2_runs_in_flight(flame_nltgv2_ctx* ctx, int32_t* n_out) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!n_out) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  ctx->track_runs = true;
  if (ctx->pending.active && !ctx->run_ev_valid[ctx->run_ev_last]) {  // a run enqueued before anybody asked: marked now
    HIPCHK(ctx, hipEventRecord(ctx->ev_run[ctx->run_ev_last], ctx->stream));
    ctx->run_ev_valid[ctx->run_ev_last] = true;
  }
  int n = 0;
  for (int i = 0; i < 2; ++i) {
    if (!ctx->run_ev_valid[i]) continue;
    const hipError_t q = hipEventQuery(ctx->ev_run[i]);
    if (q == hipErrorNotReady) {
      (void)hipGetLastError();
      ++n;
    } else if (q != hipSuccess) {
      HIPCHK(ctx, q);
    }
  }
  *n_out = n;
  return 0;
}

int flame_nltgv2_sync(flame_nltgv2_ctx* ctx) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_sync");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) {
    HIPCHK(ctx, wait_solver_stream(ctx));
    return 0;
  }
  return finish(ctx);
}

int flame_nltgv2_run(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n_iters) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_run");
  int rc = flame_nltgv2_run_async(ctx, p, n_iters);
  if (rc) return rc;
  return finish(ctx);
}

int flame_nltgv2_run_timed(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n_iters, float* elapsed_ms) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!params_ok(p) || n_iters < 0 || !elapsed_ms) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  // layout conversion and graph instantiation are not part of the steady-state step
  if (ctx->opt_solver == 0) rc = ensure_fused(ctx); else rc = ensure_canon(ctx);
  if (rc) return rc;
  rc = prepare_run(ctx, p, n_iters);
  if (rc) return rc;
  HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  rc = enqueue_run(ctx, p, n_iters);
  if (rc) return rc;
  HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  rc = finish(ctx);
  float ms = 0.f;
  HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *elapsed_ms = ms;
  return rc;
}

int flame_nltgv2_step(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p) { return flame_nltgv2_run(ctx, p, 1); }

#define CANON_OP(NAME, LAUNCH)                                                  \
  int NAME(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p) {               \
    int rc = enter(ctx);                                                        \
    if (rc) return rc;                                                          \
    if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);          \
    if (!params_ok(p)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);          \
    rc = ensure_canon(ctx);                                                     \
    if (rc) return rc;                                                          \
    const SolverParams sp = to_sp(p);                                           \
    LAUNCHCHK(ctx, LAUNCH(ctx->c, sp, ctx->stream));                            \
    ctx->fused_valid = false;                                                   \
    return finish(ctx);                                                         \
  }
CANON_OP(flame_nltgv2_dual_step, launch_dual)
CANON_OP(flame_nltgv2_primal_step, launch_primal)
CANON_OP(flame_nltgv2_extragradient_step, launch_extragradient)
#undef CANON_OP

int flame_nltgv2_save_prev(flame_nltgv2_ctx* ctx) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  rc = ensure_canon(ctx);
  if (rc) return rc;
  LAUNCHCHK(ctx, launch_save_prev(ctx->c, ctx->stream));
  ctx->fused_valid = false;
  return finish(ctx);
}


}  // extern "C"
