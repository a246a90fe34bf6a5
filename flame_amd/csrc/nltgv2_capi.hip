// nltgv2_capi.hip -- solver context and the C-ABI declared in include/flame_nltgv2.h.
//
// Host side of the drop-in boundary for flame::optimizers::nltgv2_l1_graph_regularizer
// (/root/reference/src/flame/optimizers/nltgv2_l1_graph_regularizer.h:134-168).  The context owns
// the device image of one reference Graph (h:107-112) in two forms:
//   canonical  SoA arrays in the caller's vertex/edge order  (upload/download, the individually
//              callable dual/primal/extragradient sweeps, costs)
//   packed     SELL-64 layout of the fused one-kernel-per-step sweep (run)
// and converts between them on the device only when the other form is asked for.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <new>
#include <memory>
#include <vector>

#include "flame_nltgv2.h"
#include "nltgv2_kernels.h"
#include "nltgv2_pack.hpp"
#include "roctx_ranges.hpp"

using namespace flame_hip;

namespace {

constexpr int kGraphChunk = 256;      // steps per captured hipGraph (even: keeps ping-pong parity)
constexpr int kMaxCachedGraphs = 6;
constexpr int kHeWavesPerCu = 24;      // residency cap of k_persistent_he (<= 64 VGPRs: the hardware admits 32)
constexpr int kTvLdsWavesPerCu = 16;   // ... with the slot constants in LDS (<= 128 VGPRs -> 4 waves per SIMD; 10 KB LDS per wave = all 160 KB)
constexpr size_t kXbufBytesPerVertex = 8 * 16 + 4;  // exchange buffers: up to four step buffers x (remote + same-XCD copy) of
                                                    // 16-byte records (the patch-per-wave form; the others use two) + the XCC table
constexpr int kPvDensePerCu = 23;      // k_persistent_pv is used up to this many patches per CU (24 are resident: 6 waves per SIMD)
constexpr int kPvPaceAbovePerCu = 17;  // ... and above this many its polls are paced (kPvDensePreSleep, kPvDenseGap)
constexpr int kPvDensePreSleep = 8, kPvDenseGap = 4;  // x64 cycles before the first poll of a step / between poll rounds
constexpr int kPvPreSleep = 0;         // k_persistent_pv: x64 cycles between a step's start and its first poll
constexpr int kPvPollGap = 2;          // k_persistent_pv polls: re-loading only the fetch entries still waiting, no pause between
                                       // rounds (with the round-2 first form of the kernel an s_sleep between rounds won by 1-3 %;
                                       // with the shorter hand-off path of its final form no pause wins by 3-4 % at 640x480)
constexpr int kTvWavesPerCu = 8;       // residency of k_persistent_tv (<= 256 VGPRs -> 2 waves per SIMD)
constexpr int kDualMinWavesPerCu = 0;  // auto: exchange through the XCD's L2 when more waves than this share a CU
// x64-cycle sleep between publishing and the first neighbour poll (measured optimum, r01 sweep: he 6 at
// <= 12 waves/CU, 10 above; tv is insensitive, shortest wins)
constexpr int kPreSleepHe = 6, kPreSleepHeDense = 10, kPreSleepHeOneXcd = 4, kPreSleepTv = 2;
constexpr size_t kErrBytes = 16 * sizeof(int);  // the flag word + what the first expired wait reports (report_expired)
constexpr unsigned kMaxSpins = 1u << 20;  // bound of every neighbour wait in the persistent run (~1 s of polling bursts)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct CachedGraph {
  hipGraphExec_t exec = nullptr;
  int n = 0, parity = 0, unroll = 0, wpb = 0, gen = 0;
  uint64_t topo = 0;
  flame_nltgv2_params params{};
  uint64_t stamp = 0;
};

// Open-addressing hash map u64 -> i32 (linear probing, power-of-two capacity, no erase): the per-frame
// bookkeeping of sync_graph looks up ~V feature ids and ~E feature pairs; std::unordered_map made that the
// most expensive part of a frame (2.0-2.5 ms at 640x480), this table does it in a fraction.
class FlatMap {
 public:
  FlatMap() = default;
  explicit FlatMap(size_t n) { reset(n); }
  // Empties the table for up to n keys.  A slot is live only if it carries the current generation, so a table that is
  // big enough is emptied by counting the generation up -- nothing is cleared (the per-frame sync empties two of these).
  void reset(size_t n) {
    size_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    if (cap > slots_.size() || gen_ == 0xffffffffu) {
      slots_.assign(std::max(cap, slots_.size()), Slot{0, 0, 0});
      gen_ = 0;
    }
    mask_ = slots_.size() - 1;
    ++gen_;
  }
  // inserts (k,v) if k is absent; returns the slot's value pointer and whether it was inserted
  std::pair<int32_t*, bool> emplace(uint64_t k, int32_t v) {
    size_t i = hash(k) & mask_;
    for (;; i = (i + 1) & mask_) {
      Slot& s = slots_[i];
      if (s.gen != gen_) {
        s = Slot{k, v, gen_};
        return {&s.val, true};
      }
      if (s.key == k) return {&s.val, false};
    }
  }
  const int32_t* find(uint64_t k) const {
    size_t i = hash(k) & mask_;
    for (;; i = (i + 1) & mask_) {
      const Slot& s = slots_[i];
      if (s.gen != gen_) return nullptr;
      if (s.key == k) return &s.val;
    }
  }

 private:
  struct Slot {
    uint64_t key;
    int32_t val;
    uint32_t gen;
  };
  static size_t hash(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return (size_t)(z ^ (z >> 31));
  }
  size_t mask_ = 0;
  uint32_t gen_ = 0;
  std::vector<Slot> slots_;
};

}  // namespace

struct flame_nltgv2_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int last_error = 0;
  int last_hip = 0;
  hipDeviceProp_t prop{};

  bool have_graph = false;
  bool canon_valid = false, fused_valid = false, have_prev = false;
  DevBuf sp_v[9], sp_q[3], sync_init, sync_vmap, sync_emap, sync_need;  // sync_graph: spare state arrays, inputs, index maps
  std::vector<int32_t> h_old_of_new, h_old_of_new_edge;
  FlatMap feat_maps[3];     // [cur]: feature id -> vertex of the CURRENT graph (h_feat), kept from one sync to the next; the
  int feat_cur = 0;         // next sync fills the other one; [2]: scratch of a sync (new edges' duplicates)
  bool feat_map_valid = false;
  int parity = 0;
  uint64_t topo = 0, stamp = 0;

  int opt_solver = 0, opt_use_graph = 1, opt_block_waves = 0, opt_unroll = 0, opt_persistent = 1, opt_dual = 1;
  int xbuf_form = 0;     // the persistent form whose records the exchange buffers hold (0: cleared)
  int opt_xcds = 0;      // XCDs a persistent launch spreads over: 0 = auto, 1..8
  bool photo_fused = false;     // flame_nltgv2_photo_fuse: every run also leaves the photometric residual in photo_err
  PhotoGeometry photo_geo{};
  float photo_scale = 1.0f;
  int photo_border = 3;
  float* export_ptr = nullptr;  // flame_nltgv2_set_export_target: every run also leaves x * scale there
  float export_scale = 1.0f;
  int opt_fault = 0;     // test hook: > 0 = the next persistent runs time out after this many spins
  int opt_presleep = 0;  // 0: auto (kPreSleep*); n > 0: (n - 1) x 64 cycles
  int opt_poll_gap = 0;  // patch-per-wave form: 0 = default (kPvPollGap), 1 = no sleep between polls, 2 = one s_sleep, 3 / 4 = the same, narrowed
  mutable int pv_occ = 0;            // patches of k_persistent_pv the runtime keeps resident per CU for the current layout
  mutable uint64_t pv_occ_topo = ~0ull;
  int opt_probe = 0;     // > 0: k_persistent_pv records a per-patch, per-step cycle probe (flame_nltgv2_read_probe)
  size_t probe_words = 0;
  int opt_tv_lds = 1;  // 0 registers, 1 auto (LDS when the register form is not resident in one launch), 2 LDS
  uint32_t tag_next = 1;  // persistent run: tag of the current bar values (monotonic)
  int last_run_path = 0, last_run_groups = 0;
  uint64_t persist_refused_topo = ~0ull;  // topology for which the runtime refused the persistent grid
  bool static_stale = false;  // pos changed on the device (project_graph): packed alpha/dx/dy need a re-pack
  uint64_t coop_checked_key = 0;  // (topology, form) whose persistent grid the runtime has verified as resident
  int buf_gen = 0;                // which of the two (hq, vstate) copies is current; part of the hipGraph cache key
  int timeouts_recovered = 0;     // persistent runs that timed out and were redone on the per-step path
  int torn_records_detected = 0;  // ... that the record verification (opt_verify) stopped, redone the same way
  int opt_verify = 0;             // 1: persistent kernels re-read every record after its tag matched; 2: + test hook
  // Record placement of the patch-per-wave form (nltgv2_layout.hip): a pool of pages measured once per context, the
  // records read across XCDs assigned to them once per topology
  int opt_place = 1;              // 1 (default) on, 0 off
  int place_state = 0;            // 0 not calibrated yet, 1 page ranking on the device, -1 unavailable (calibration failed)
  uint64_t place_topo = ~0ull;    // topology / patches per XCD the record offsets are valid for
  int place_per_xcd = 0;
  char* place_base = nullptr;     // the pool, 4 KB aligned inside place_pool
  float place_best_us = 0.0f, place_mean_us = 0.0f, place_worst_us = 0.0f;  // one-way hand-off by page choice, mean over XCD pairs
  DevBuf place_pool, place_rank, place_fill, place_rec_off, place_patch, place_meas;
  // The persistent run(s) in flight, until finish() has seen the error word: what is needed to take them back.  One
  // run is taken back by swapping the buffer roles (it wrote the other copies).  When more work is enqueued before the
  // first run has been checked (run_async back to back: the frame loop, bench.py), the state the chain started from is
  // copied aside first (three device-to-device copies, once per chain), and the chain is kept as a list of operations:
  // a wait that expires anywhere in it restores that state and replays the list on the one-launch-per-step path.
  struct PendingOp {
    int kind = 0;  // 0 run, 1 explicit export of x * scale
    flame_nltgv2_params params{};
    int n = 0;
    float* dst = nullptr;
    float scale = 1.0f;
  };
  struct PendingRun {
    bool active = false;
    bool snapshotted = false;  // the pre-chain state is in snap_hq / snap_vstate / snap_bar
    int parity_before = 0;
    bool have_prev_before = false;
    std::vector<PendingOp> ops;
  } pending;
  DevBuf snap_hq, snap_vstate, snap_bar;
  DevBuf iperm, order_m, rid_of;   // per-vertex tables the device-side layout expansion reads (nltgv2_layout.hip)
  bool he_built = false, tv_built = false;  // layouts (C) / (D) exist for the current topology (built on demand)
  void* h_stage = nullptr;         // pinned staging buffer of the uploads
  DevBuf d_stage;                 // ... and its device-side landing area (one copy; k_scatter distributes)
  size_t stage_cap = 0;

  PackedLayout L;
  std::vector<int32_t> h_src, h_dst, h_feat;  // host image of the current topology (for sync_graph)
  CanonArgs c;
  FusedArgs f;
  std::vector<DevBuf*> all;
  // canonical
  DevBuf pos, x, w1, w2, xb, w1b, w2b, xp, w1p, w2p, data, weight, src, dst, alpha, beta, q1, q2, q3, row_ptr, half;
  // packed
  DevBuf slice_row, perm, pdeg, rec_nbr, rec_edge, edge_src_slot, hrec, hq, vstate, vaux, bar0, bar1, vprev;
  DevBuf cost_terms;          // addends of smoothnessCost / dataCost
  DevBuf run_tail;            // RunTail of the persistent kernels (standing export / photometric targets)
  RunTail tail_sent{};        // what run_tail currently holds
  bool tail_valid = false;
  std::vector<float> h_terms;
  DevBuf hq_alt, vstate_alt;  // the other copies of hq / vstate: a persistent run writes there, success swaps the roles
  DevBuf xbuf, abort_flag, he_slot, he_vid, he_meta, he_wave_chain, tv_slot, tv_vid, tv_meta, tv_wave;
  DevBuf wg_slot, wg_vid, wg_meta, wg_nbr, wg_fetch, wg_info, wg_v0, wg_vfirst, probe, progress;
  // misc
  DevBuf err, cost_out, img_ref, img_cmp, photo_err, r_tris, r_valid, r_keys, r_img, r_cov, r_vtx, r_val;
  int img_rows = 0, img_cols = 0, img_step = 0;
  int* h_err = nullptr;    // pinned, kErrBytes
  int last_expired[16] = {0};  // what the most recent expired wait reported (report_expired)
  float* h_cost = nullptr; // pinned
  std::vector<CachedGraph> graphs;
  size_t device_bytes = 0;
};

namespace {

#define HIPCHK(ctx, expr)                              \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) {                            \
      (ctx)->last_hip = (int)_e;                       \
      (ctx)->last_error = FLAME_NLTGV2_ERR_HIP;        \
      return FLAME_NLTGV2_ERR_HIP;                     \
    }                                                  \
  } while (0)

#define LAUNCHCHK(ctx, expr)                           \
  do {                                                 \
    int _e = (expr);                                   \
    if (_e != 0) {                                     \
      (ctx)->last_hip = _e;                            \
      (ctx)->last_error = FLAME_NLTGV2_ERR_HIP;        \
      return FLAME_NLTGV2_ERR_HIP;                     \
    }                                                  \
  } while (0)

int fail(flame_nltgv2_ctx* ctx, int status) {
  if (ctx) ctx->last_error = status;
  return status;
}

int ensure(flame_nltgv2_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.cap >= bytes) return 0;
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
  f.he_waves = (ctx->he_built && ctx->L.he_ok) ? ctx->L.he_waves : 0;
  f.he_slot = (int32_t*)ctx->he_slot.p, f.he_vid = (int32_t*)ctx->he_vid.p;
  f.he_meta = (uint32_t*)ctx->he_meta.p, f.he_wave_chain = (int32_t*)ctx->he_wave_chain.p;
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

int finish(flame_nltgv2_ctx* ctx);
int snapshot_chain_start(flame_nltgv2_ctx* ctx);
constexpr size_t kMaxChain = 256;  // operations enqueued behind an unchecked persistent run before the host settles it

// records the exchange buffers hold: one per packed vertex slot (the he / tv forms index by packed vertex)
size_t records_capacity(const PackedLayout& L) {
  const size_t n_packed = (size_t)L.n_slices * kWave, n_rec = ((size_t)L.n_rec + kWave - 1) / kWave * kWave;
  return std::max(n_packed, n_rec);
}

int ensure_canon(flame_nltgv2_ctx* ctx) {
  if (ctx->pending.active) {  // a persistent run is still unchecked: settle it before anything reads or edits the state
    const int rc = finish(ctx);
    if (rc) return rc;
  }
  if (ctx->canon_valid) return 0;
  LAUNCHCHK(ctx, launch_unpack_state(ctx->c, ctx->f, ctx->parity, ctx->have_prev, ctx->stream));
  ctx->canon_valid = true;
  return 0;
}

int ensure_fused(flame_nltgv2_ctx* ctx) {
  if (ctx->fused_valid) return 0;
  if (ctx->static_stale) {  // positions moved: dx, dy of the packed records follow (alpha stays the caller's)
    LAUNCHCHK(ctx, launch_pack_static(ctx->c, ctx->f, ctx->stream));
    ctx->static_stale = false;
  }
  LAUNCHCHK(ctx, launch_pack_state(ctx->c, ctx->f, ctx->parity, ctx->stream));
  ctx->fused_valid = true;
  ctx->have_prev = false;
  return 0;
}

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

int plan_persistent(flame_nltgv2_ctx* ctx, int n, std::vector<WaveGroup>* groups, int* use_tv_lds = nullptr) {
  groups->clear();
  if (use_tv_lds) *use_tv_lds = 0;
  if (!ctx->opt_persistent || n < 4 || n > (1 << 24) || !ctx->prop.cooperativeLaunch) return 0;
  if (ctx->persist_refused_topo == ctx->topo) return 0;
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
  const int wg_cap = ctx->pv_occ * cus;  // patch-per-wave form, in patches
  const int he_cap = kHeWavesPerCu * cus;
  // The lane-per-half-edge rows (C) come from the same greedy walk as the patches (E): as many waves, possible under the
  // same condition (no vertex of more than 64 incident edges).  They, and the vertex-per-lane rows (D), are built only
  // when their form is actually chosen.
  const bool pv_fits = L.wg_ok && L.wg_rowpack && L.wg_count > 0 && L.wg_count <= wg_cap;  // (the kernel runs row-packed patches)
  const bool he_possible = L.wg_ok && L.wg_count > 0;
  int form = 0;
  if (ctx->opt_persistent == 4) form = (L.wg_ok && L.wg_rowpack) ? 3 : 0;
  else if (ctx->opt_persistent == 2) form = he_possible ? 1 : 0;
  else if (ctx->opt_persistent == 3) form = 2;
  else if (pv_fits) form = 3;  // lowest latency wherever all patches are resident: 320x240 ... 1280x720 single frames
  else if (he_possible && L.wg_count <= he_cap) form = 1;
  else form = 2;               // too big for that: vertex-per-lane, in groups of whole components if need be
  if (form == 1 || form == 2) {
    if (ensure_form_rows(ctx, form) != 0) return 0;
    if (form == 2 && !L.tv_ok) {
      form = he_possible ? 1 : 0;
      if (form == 1 && ensure_form_rows(ctx, 1) != 0) return 0;
    }
  }
  // vertex-per-lane form: slot constants in registers (8 waves/CU, fastest per wave) while the graph is resident
  // that way, else in LDS (16 waves/CU: 30 frames of 640x480 resident in one launch)
  const bool tv_lds = form == 2 && (ctx->opt_tv_lds == 2 || (ctx->opt_tv_lds == 1 && L.tv_waves > kTvWavesPerCu * cus));
  const int tv_cap = (tv_lds ? kTvLdsWavesPerCu : kTvWavesPerCu) * cus;
  if (use_tv_lds) *use_tv_lds = tv_lds ? 1 : 0;
  if (form == 0) return 0;
  const int total = form == 3 ? L.wg_count : form == 2 ? L.tv_waves : L.he_waves;
  const int cap = form == 3 ? wg_cap : form == 2 ? tv_cap : he_cap;
  if (total <= 0) return 0;
  if (total <= cap) {
    groups->push_back(WaveGroup{0, total});
    return form;
  }
  const std::vector<int32_t>& cw = form == 3 ? L.comp_wg : form == 2 ? L.comp_tv_wave : L.comp_he_wave;
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
int place_calibrate(flame_nltgv2_ctx* ctx) {
  ctx->place_state = -1;
  constexpr int P = kPlacePages, kIters = 12;
  const size_t pool_bytes = (size_t)2 * P * 4096;
  int rc = ensure(ctx, ctx->place_pool, pool_bytes + 4096);
  if (!rc) rc = ensure(ctx, ctx->place_meas, sizeof(unsigned) * 64 * 2 * P + sizeof(int) * 64 + sizeof(int));
  if (!rc) rc = ensure(ctx, ctx->place_rank, sizeof(uint16_t) * 2 * 64 * P);
  if (!rc) rc = ensure(ctx, ctx->place_fill, sizeof(int) * (2 * P + 16));  // (+ the rotation word of the launches)
  if (rc) return rc;
  ctx->place_base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ctx->place_pool.p) + 4095) & ~uintptr_t(4095));
  unsigned* d_out = (unsigned*)ctx->place_meas.p;
  int* d_xcc = (int*)(d_out + (size_t)64 * 2 * P);
  int* d_fail = d_xcc + 64;
  HIPCHK(ctx, hipMemsetAsync(ctx->place_base, 0, pool_bytes, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->place_meas.p, 0, sizeof(unsigned) * 64 * 2 * P + sizeof(int) * 65, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->place_fill.p, 0, sizeof(int) * (2 * P + 16), ctx->stream));
  LAUNCHCHK(ctx, launch_place_calibrate(ctx->place_base, 2 * P, kIters, d_out, d_xcc, d_fail, ctx->stream));
  std::vector<unsigned> out((size_t)64 * 2 * P);
  int xcc[65];
  HIPCHK(ctx, hipMemcpyAsync(out.data(), d_out, sizeof(unsigned) * out.size(), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(xcc, d_xcc, sizeof(int) * 65, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  const double to_us = 1.0 / (100.0 * kIters) / (2.0 * 56.0);  // 100 MHz ticks of kIters hand-offs; mean of 2 x 56 classes
  ctx->place_best_us = (float)(best * to_us), ctx->place_mean_us = (float)(mean * to_us), ctx->place_worst_us = (float)(worst * to_us);
  HIPCHK(ctx, hipMemcpyAsync(ctx->place_rank.p, rank.data(), sizeof(uint16_t) * rank.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (`rank` is pageable and leaves scope)
  ctx->place_state = 1;
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

int enqueue_run(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n) {
  if (n <= 0) return 0;
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
    if (ctx->export_ptr) LAUNCHCHK(ctx, launch_export(ctx->c, ctx->f, false, ctx->export_scale, ctx->export_ptr, ctx->stream));
    return enqueue_photo_sweep(ctx, false);
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
  const int form = plan_persistent(ctx, n, &groups, &tv_lds);
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
      ctx->tag_next = 1;
    }
    ctx->xbuf_form = form;
    // a fresh first tag per run: records left by earlier runs (whose state may since have been changed
    // by per-step launches or host uploads) can never satisfy a wait of this one
    const uint32_t tag0 = ctx->tag_next + 2;
    // (topology, form, kernel instance): a new instance -- other registers, other LDS -- gets a cooperative first launch
    const uint64_t key = ctx->topo * 256 + (uint64_t)form * 64 + (uint64_t)tv_lds * 32 + (ctx->opt_verify != 0 ? 16 : 0) + (ctx->opt_probe != 0 ? 8 : 0) +
                         (ctx->opt_dual == 2 ? 4 : ctx->opt_dual == 1 ? 2 : 0) + (ctx->opt_xcds > 0 ? 1 : 0);
    {  // standing outputs: (re)send the small block the kernels read in their epilogue when it changed
      RunTail want;
      std::memset(static_cast<void*>(&want), 0, sizeof want);
      want.export_out = ctx->export_ptr, want.export_scale = ctx->export_scale;
      const PhotoFuse pf = photo_target(ctx);
      std::memcpy(static_cast<void*>(&want.photo), &pf, sizeof pf);
      if (form == 3 && std::getenv("FLAME_NLTGV2_TRACE")) {
        rc = ensure(ctx, ctx->progress, 2 * sizeof(unsigned) * (size_t)ctx->L.wg_count);  // [how far | started when]
        if (rc) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->progress.p, 0, 2 * sizeof(unsigned) * (size_t)ctx->L.wg_count, ctx->stream));
        want.progress = (unsigned*)ctx->progress.p;
      }
      rc = ensure(ctx, ctx->run_tail, sizeof(RunTail));
      if (rc) return rc;
      if (!ctx->tail_valid || std::memcmp(&want, &ctx->tail_sent, sizeof(RunTail)) != 0) {
        // pageable source: the runtime stages it during the call, so `want` may go out of scope
        HIPCHK(ctx, hipMemcpyAsync(ctx->run_tail.p, &want, sizeof(RunTail), hipMemcpyHostToDevice, ctx->stream));
        ctx->tail_sent = want;
        ctx->tail_valid = true;
      }
    }
    int e = 0;
    for (const WaveGroup& gr : groups) {
      const int pw = gr.count <= 4 * ctx->prop.multiProcessorCount ? 1 : 4;  // waves per workgroup
      // same-XCD exchange through L2: with the waves laid out along the Morton curve it wins at every size
      // (measured per step: 640x480 -16 %, 1280x720 -23 %, 1080p -18 %, 7-frame batch -20 %, 15 frames -19 %)
      const int dual = (ctx->opt_dual == 2 || (ctx->opt_dual == 1 && gr.count > kDualMinWavesPerCu * ctx->prop.multiProcessorCount) ? 1 : 0) |
                       (ctx->opt_verify == 2 ? 6 : ctx->opt_verify == 1 ? 2 : 0);  // bits 1, 2: record verification, its test hook
      // A graph of <= 8 lane-per-half-edge waves per CU of ONE XCD (32 CUs) runs there entirely: every exchange
      // stays in that XCD's L2 (measured 320x240: 1.23 instead of 1.47 us per step; at 640x480 the 26 waves per CU
      // this would need cost more than the shorter hop saves).
      const int cus_per_xcd = ctx->prop.multiProcessorCount / 8;
      // (patch-per-wave form: one XCD while its CUs get at most two patches each -- measured: 48 patches 0.98 against
      // 1.21 us per step on all eight, 208 patches 1.53 against 1.33)
      const bool one_xcd = form == 3 ? gr.count <= 2 * cus_per_xcd : (form == 1 && gr.count <= 8 * cus_per_xcd);
      const int xcds = ctx->opt_xcds > 0 ? ctx->opt_xcds : one_xcd ? 1 : 8;
      const int presleep = ctx->opt_presleep > 0 ? ctx->opt_presleep - 1
                           : form == 2                ? kPreSleepTv
                           : xcds == 1                ? kPreSleepHeOneXcd
                           : gr.count > 12 * ctx->prop.multiProcessorCount ? kPreSleepHeDense
                                                                            : kPreSleepHe;
      const unsigned spins_arg = ctx->opt_fault > 0 ? (0x80000000u | (unsigned)ctx->opt_fault) : kMaxSpins;
      if (form == 3) {
        // pacing: none where a CU holds few patches (a poll costs nothing there and a pause only delays the hand-off); at
        // high residency the polls of ~20 waves per CU saturate the L2s' request ports and the fabric and it is the hand-off
        // itself that slows down (probe: 0.72 us at 4 patches per CU, 1.03 at 15; 26 per CU unpaced: 50 us per step) -- a
        // pause before the first poll and between rounds then wins (tools/pv_big.py sweeps, profiles/r03_pv_dense.txt)
        const bool dense = gr.count > kPvPaceAbovePerCu * ctx->prop.multiProcessorCount;
        const int gap = ctx->opt_poll_gap > 0 ? ctx->opt_poll_gap - 1 : dense ? (3 | ((kPvDenseGap - 1) << 4)) : kPvPollGap;
        ctx->f.wg_poll_gap = gap |
                             ((ctx->opt_presleep > 0 ? ctx->opt_presleep - 1 : dense ? kPvDensePreSleep : kPvPreSleep) << 8) |
                             0;
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
      e = launch_persistent_run(ctx->f, to_sp(p), form, gr.begin, gr.count, ctx->parity, tag0, n, pw, spins_arg, presleep, dual,
                                tv_lds, xcds, (const RunTail*)ctx->run_tail.p, ctx->coop_checked_key != key, ctx->stream);
      if (e != 0) break;
    }
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
        ctx->pending.ops.push_back(op);
      }
      std::swap(ctx->hq, ctx->hq_alt);
      std::swap(ctx->vstate, ctx->vstate_alt);
      ctx->buf_gen ^= 1;
      refresh_args(ctx);
      ctx->coop_checked_key = key;
      ctx->last_run_path = form == 3 ? 6 : form == 2 ? 5 : 1;
      ctx->last_run_groups = (int)groups.size();
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
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(ctx->err.p, 0, kErrBytes, ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(ctx->abort_flag.p, 0, sizeof(int), ctx->stream));
    }
  }
  if (ctx->pending.active) {
    flame_nltgv2_ctx::PendingOp op;
    op.kind = 0, op.params = *p, op.n = n;
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

int finish(flame_nltgv2_ctx* ctx) {
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_err, ctx->err.p, kErrBytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  flame_nltgv2_ctx::PendingRun run;
  std::swap(run, ctx->pending);  // (ctx->pending is now inactive and empty)
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
    ctx->persist_refused_topo = ctx->topo;
    if (!(*ctx->h_err & 4)) ctx->timeouts_recovered++;
    HIPCHK(ctx, hipMemsetAsync(ctx->err.p, 0, kErrBytes, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->abort_flag.p, 0, sizeof(int), ctx->stream));
    for (const flame_nltgv2_ctx::PendingOp& op : run.ops) {
      if (op.kind == 0) {
        const int rc = enqueue_run(ctx, &op.params, op.n);
        if (rc) return rc;
      } else {
        LAUNCHCHK(ctx, launch_export(ctx->c, ctx->f, true, op.scale, op.dst, ctx->stream));
      }
    }
    return finish(ctx);
  }
  if (*ctx->h_err != 0) {
    // NaN/Inf in a dual variable (the reference's FLAME_ASSERT h:174): reported once; the state stays readable
    // (download_state, costs) and the solve can go on or be re-initialised -- q was clamped to +-1 where it happened
    HIPCHK(ctx, hipMemsetAsync(ctx->err.p, 0, kErrBytes, ctx->stream));
    return fail(ctx, FLAME_NLTGV2_ERR_NAN);
  }
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
  HIPCHK(ctx, hipMemcpyAsync(ctx->snap_hq.p, ctx->hq_alt.p, sizeof(float4) * n_slots, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->snap_vstate.p, ctx->vstate_alt.p, sizeof(float4) * n_packed, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->snap_bar.p, ctx->f.bar[ctx->pending.parity_before], sizeof(float4) * n_packed, hipMemcpyDeviceToDevice,
                             ctx->stream));
  ctx->pending.snapshotted = true;
  return 0;
}

// ---- uploading a topology ---------------------------------------------------------------------------------------------
// The host computes only the per-vertex tables (nltgv2_pack.hpp with host_expand = false); every array goes through ONE
// pinned staging buffer (the copies out of it are asynchronous and cost a few microseconds each; out of pageable memory
// each of the ~35 copies of round 1 was a synchronous staging round trip); the per-slot and per-lane arrays are expanded
// on the device (nltgv2_layout.hip).
struct StageCopy {
  DevBuf* b;
  const void* src;
  size_t bytes;
};

struct StageFill {
  void* dst;
  size_t bytes;
  uint32_t word;
};

// All of `cp` through the pinned staging buffer as ONE host-to-device copy into a device-side blob, then one kernel that
// distributes the pieces to their buffers and does the clears of `fills` (k_scatter).  The caller's arrays are free when
// this returns (they were copied into the staging buffer); the staging buffer itself is reused by the next upload, which
// synchronises the stream first.
int staged_h2d(flame_nltgv2_ctx* ctx, const StageCopy* cp, size_t n, const StageFill* fills = nullptr, size_t n_fills = 0) {
  size_t total = 0;
  for (size_t i = 0; i < n; ++i) total += (cp[i].bytes + 255) & ~size_t(255);
  if (total >= (size_t)0xffffff00u) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (total > ctx->stage_cap) {
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr, ctx->stage_cap = 0;
    const size_t want = total + total / 2;
    if (hipHostMalloc(&ctx->h_stage, want, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return fail(ctx, FLAME_NLTGV2_ERR_OOM);
    }
    ctx->stage_cap = want;
  }
  int rc = ensure(ctx, ctx->d_stage, ctx->stage_cap);
  if (rc) return rc;
  std::vector<ScatterTable> tables(1);
  auto push = [&](const ScatterEntry& e) {
    if (tables.back().n == kScatterMax) tables.emplace_back();
    ScatterTable& t = tables.back();
    t.e[t.n++] = e;
  };
  size_t off = 0;
  for (size_t i = 0; i < n; ++i) {
    if (cp[i].bytes == 0) continue;
    std::memcpy(static_cast<char*>(ctx->h_stage) + off, cp[i].src, cp[i].bytes);
    push(ScatterEntry{cp[i].b->p, (uint32_t)off, 0u, cp[i].bytes});
    off += (cp[i].bytes + 255) & ~size_t(255);
  }
  for (size_t i = 0; i < n_fills; ++i)
    if (fills[i].bytes) push(ScatterEntry{fills[i].dst, kScatterFill, fills[i].word, fills[i].bytes});
  if (off) HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage.p, ctx->h_stage, off, hipMemcpyHostToDevice, ctx->stream));
  for (const ScatterTable& t : tables) LAUNCHCHK(ctx, launch_scatter(t, ctx->d_stage.p, ctx->stream));
  return 0;
}

// Layout + topology arrays of `g` (V, E, pos, src, dst) onto the device; the caller adds the state.  On return the
// stream still holds the copies: the caller synchronises before the staging buffer or `g`'s arrays may change.
int upload_topology(flame_nltgv2_ctx* ctx, const flame_nltgv2_graph* g, const StageCopy* extra, size_t n_extra, bool long_lived) {
  const int32_t V = g->V, E = g->E;
  (void)long_lived;
  int rc = build_layout(g, &ctx->L, /*host_expand=*/false, /*rowpack=*/true,
                        /*rowpack_max_patches=*/ctx->opt_persistent == 4 ? 0x7fffffff : kPvDensePerCu * ctx->prop.multiProcessorCount);
  if (rc) return fail(ctx, rc);
  // Row packing costs ~15 % more waves than lanes back to back.  It pays where the patch-per-wave kernel runs them; a layout
  // that turns out too large for that kernel (more patches than the estimate, or a vertex of more than 16 edges, whose
  // instance of the kernel keeps 12 patches per CU) is better off back to back, for the lane-per-half-edge form.
  // (FLAME_NLTGV2_OPT_PERSISTENT 4 -- the patch-per-wave form asked for by name -- keeps the row-packed layout whatever the
  //  size: the kernel then runs it as groups of whole components)
  if (ctx->L.wg_ok && ctx->L.wg_rowpack && ctx->opt_persistent != 4 &&
      ctx->L.wg_count > (ctx->L.wg_slab_slots > 0 ? 4 * pv_real_waves_per_simd(2, false) : kPvDensePerCu) * ctx->prop.multiProcessorCount) {
    rc = build_layout(g, &ctx->L, /*host_expand=*/false, /*rowpack=*/false, 0);
    if (rc) return fail(ctx, rc);
  }
  const PackedLayout& L = ctx->L;
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
      {&ctx->wg_v0, sizeof(int32_t) * L.wg_v0.size()},
      {&ctx->wg_vfirst, L.wg_vfirst.size() + 16}, {&ctx->abort_flag, sizeof(int)}, {&ctx->err, kErrBytes}, {&ctx->cost_out, 2 * sizeof(float)},
      {&ctx->wg_slot, sizeof(int32_t) * lanes}, {&ctx->wg_vid, sizeof(int32_t) * lanes}, {&ctx->wg_meta, sizeof(uint32_t) * lanes},
      {&ctx->wg_nbr, sizeof(int32_t) * lanes}, {&ctx->wg_fetch, sizeof(int32_t) * lanes},
      {&ctx->wg_info, sizeof(int32_t) * L.wg_info.size()}};
  for (auto& r : req) {
    rc = ensure(ctx, *r.b, r.bytes);
    if (rc) return rc;
  }
  ctx->topo++;
  ctx->he_built = ctx->tv_built = false;
  drop_graphs(ctx);
  refresh_args(ctx);

  std::vector<StageCopy> cp = {
      {&ctx->pos, g->pos, 2 * fV}, {&ctx->src, g->src, fE}, {&ctx->dst, g->dst, fE},
      {&ctx->row_ptr, L.row_ptr.data(), sizeof(int32_t) * ((size_t)V + 1)}, {&ctx->half, L.half.data(), 2 * fE},
      {&ctx->slice_row, L.slice_row.data(), sizeof(int32_t) * ((size_t)L.n_slices + 1)},
      {&ctx->perm, L.perm.data(), sizeof(int32_t) * n_packed}, {&ctx->pdeg, L.pdeg.data(), sizeof(int32_t) * n_packed},
      {&ctx->iperm, L.iperm.data(), iV}, {&ctx->order_m, L.order_m.data(), iV}, {&ctx->rid_of, L.rid_of.data(), iV},
      {&ctx->wg_info, L.wg_info.data(), sizeof(int32_t) * L.wg_info.size()},
      {&ctx->wg_v0, L.wg_v0.data(), sizeof(int32_t) * L.wg_v0.size()},
      {&ctx->wg_vfirst, L.wg_vfirst.data(), L.wg_vfirst.size()}};
  cp.insert(cp.end(), extra, extra + n_extra);
  std::vector<StageFill> fills = {
      {ctx->err.p, kErrBytes, 0u}, {ctx->abort_flag.p, sizeof(int), 0u}, {ctx->xbuf.p, kXbufBytesPerVertex * records_capacity(L) + 64, 0u},
      // empty slots / padding vertices of the second copies: zero, as the packing kernels write them in the first
      {ctx->hq_alt.p, sizeof(float4) * n_slots, 0u}, {ctx->vstate_alt.p, sizeof(float4) * n_packed, 0u},
      // the spare rows behind the last slice: no edge, neighbour 0 (what the unrolled sweeps may read past a slice's end)
      {(char*)ctx->rec_edge.p + sizeof(int32_t) * (size_t)L.rows * kWave, sizeof(int32_t) * kRowPad * kWave, 0xffffffffu},
      {(char*)ctx->rec_nbr.p + sizeof(uint32_t) * (size_t)L.rows * kWave, sizeof(uint32_t) * kRowPad * kWave, 0u}};
  // (the tags start over below: no record of an earlier topology may survive, in the placement pool either)
  if (ctx->place_base) {
    fills.push_back(StageFill{ctx->place_base, (size_t)2 * kPlacePages * 4096, 0u});
    fills.push_back(StageFill{(int*)ctx->place_fill.p + 2 * kPlacePages, 64, 0u});
  }
  rc = staged_h2d(ctx, cp.data(), cp.size(), fills.data(), fills.size());
  if (rc) return rc;
  LAUNCHCHK(ctx, launch_build_sell(ctx->c, ctx->f, (const int32_t*)ctx->iperm.p, ctx->stream));
  if (L.wg_ok)
    LAUNCHCHK(ctx, launch_build_patches(ctx->c, ctx->f, (const int32_t*)ctx->wg_v0.p, (const int32_t*)ctx->order_m.p,
                                        (const int32_t*)ctx->rid_of.p, (const uint8_t*)ctx->wg_vfirst.p,
                                        (const int32_t*)ctx->iperm.p, ctx->stream));
  // where the records that cross XCDs go (if the pages have been timed already; otherwise the first run does both): on
  // the stream behind the layout kernels, nobody waits for it
  if (L.wg_ok && ctx->place_state == 1 && ctx->opt_place && L.wg_count > 2 * (ctx->prop.multiProcessorCount / 8) &&
      (ctx->opt_xcds == 0 || ctx->opt_xcds == 8)) {
    refresh_args(ctx);
    rc = place_records(ctx, (L.wg_count + 7) / 8);
    if (rc) return rc;
  }
  ctx->pending = flame_nltgv2_ctx::PendingRun{};
  ctx->tag_next = 1;
  ctx->xbuf_form = 0;
  ctx->static_stale = false;
  ctx->h_src.assign(g->src, g->src + E);
  ctx->h_dst.assign(g->dst, g->dst + E);
  return 0;
}

// (C) / (D) rows: built and uploaded when the lane-per-half-edge / vertex-per-lane persistent form is first wanted for
// the current topology (single frames run in the patch-per-wave form and never need them).
int ensure_form_rows(flame_nltgv2_ctx* ctx, int form) {
  PackedLayout& L = ctx->L;
  if (form == 1 && !ctx->he_built) {
    // (C) is (E) lane for lane (the same greedy walk): converted on the device from the patch rows, no host work
    const size_t lanes = (size_t)L.wg_count * kWave;
    L.he_ok = L.wg_ok, L.he_waves = L.wg_count, L.he_max_chain = std::max(L.max_degree, 1), L.comp_he_wave = L.comp_wg;
    struct { DevBuf* b; size_t bytes; } req[] = {{&ctx->he_slot, sizeof(int32_t) * lanes}, {&ctx->he_vid, sizeof(int32_t) * lanes},
                                                 {&ctx->he_meta, sizeof(uint32_t) * lanes},
                                                 {&ctx->he_wave_chain, sizeof(int32_t) * (size_t)L.wg_count}};
    bool grow = false;
    for (auto& r : req) grow = grow || r.bytes > r.b->cap;
    if (grow) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (a buffer is about to be reallocated)
    for (auto& r : req) {
      const int rc = ensure(ctx, *r.b, r.bytes);
      if (rc) return rc;
    }
    LAUNCHCHK(ctx, launch_he_from_patches(ctx->f, (int32_t*)ctx->he_slot.p, (int32_t*)ctx->he_vid.p, (uint32_t*)ctx->he_meta.p,
                                          (int32_t*)ctx->he_wave_chain.p, ctx->stream));
    ctx->he_built = true;
    refresh_args(ctx);
  }
  if (form == 2 && !ctx->tv_built) {
    build_tv_rows(&L);
    struct { DevBuf* b; const void* src; size_t bytes; } cp[] = {
        {&ctx->tv_slot, L.tv_slot.data(), sizeof(int32_t) * L.tv_slot.size()}, {&ctx->tv_vid, L.tv_vid.data(), sizeof(int32_t) * L.tv_vid.size()},
        {&ctx->tv_meta, L.tv_meta.data(), sizeof(uint32_t) * L.tv_meta.size()}, {&ctx->tv_wave, L.tv_wave.data(), sizeof(uint32_t) * L.tv_wave.size()}};
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& c : cp) {
      int rc = ensure(ctx, *c.b, c.bytes);
      if (!rc) rc = h2d(ctx, *c.b, c.src, c.bytes);
      if (rc) return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tv_built = true;
    refresh_args(ctx);
  }
  return 0;
}

bool params_ok(const flame_nltgv2_params* p) { return p != nullptr; }

}  // namespace

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
  ok = ok && hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreate(&ctx->ev0) == hipSuccess && hipEventCreate(&ctx->ev1) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&ctx->h_err, kErrBytes, hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&ctx->h_cost, 2 * sizeof(float), hipHostMallocDefault) == hipSuccess;
  if (!ok) {
    flame_nltgv2_destroy(ctx);
    return FLAME_NLTGV2_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  *ctx->h_err = 0;
  ctx->all = {&ctx->pos, &ctx->x, &ctx->w1, &ctx->w2, &ctx->xb, &ctx->w1b, &ctx->w2b, &ctx->xp, &ctx->w1p,
              &ctx->w2p, &ctx->data, &ctx->weight, &ctx->src, &ctx->dst, &ctx->alpha, &ctx->beta, &ctx->q1,
              &ctx->q2, &ctx->q3, &ctx->row_ptr, &ctx->half, &ctx->slice_row, &ctx->perm, &ctx->pdeg,
              &ctx->rec_nbr, &ctx->rec_edge, &ctx->edge_src_slot, &ctx->hrec, &ctx->hq, &ctx->vstate, &ctx->hq_alt, &ctx->vstate_alt, &ctx->cost_terms, &ctx->run_tail,
              &ctx->vaux, &ctx->bar0, &ctx->bar1, &ctx->vprev, &ctx->xbuf, &ctx->abort_flag, &ctx->he_slot, &ctx->he_vid, &ctx->he_meta, &ctx->he_wave_chain, &ctx->tv_slot, &ctx->tv_vid, &ctx->tv_meta, &ctx->tv_wave, &ctx->err,
              &ctx->cost_out, &ctx->img_ref, &ctx->img_cmp, &ctx->photo_err, &ctx->r_tris, &ctx->r_valid, &ctx->r_keys,
              &ctx->r_img, &ctx->r_cov, &ctx->r_vtx, &ctx->r_val, &ctx->wg_slot, &ctx->wg_vid, &ctx->wg_meta, &ctx->wg_nbr,
              &ctx->wg_fetch, &ctx->wg_info, &ctx->wg_v0, &ctx->probe, &ctx->snap_hq, &ctx->snap_vstate, &ctx->snap_bar, &ctx->iperm,
              &ctx->order_m, &ctx->rid_of, &ctx->d_stage, &ctx->sync_init, &ctx->sync_vmap, &ctx->sync_emap, &ctx->sync_need,
              &ctx->wg_vfirst, &ctx->place_pool, &ctx->place_rank, &ctx->place_fill, &ctx->place_rec_off, &ctx->place_patch, &ctx->place_meas, &ctx->progress};
  for (auto& b : ctx->sp_v) ctx->all.push_back(&b);
  for (auto& b : ctx->sp_q) ctx->all.push_back(&b);
  *out = ctx;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_destroy(flame_nltgv2_ctx* ctx) {
  if (!ctx) return FLAME_NLTGV2_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  drop_graphs(ctx);
  for (DevBuf* b : ctx->all)
    if (b->p) (void)hipFree(b->p);
  if (ctx->h_err) (void)hipHostFree(ctx->h_err);
  if (ctx->h_cost) (void)hipHostFree(ctx->h_cost);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_set_stream(flame_nltgv2_ctx* ctx, void* hip_stream) {
  int rc = enter(ctx);
  if (rc) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
      if (value < 0 || value > 4) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_persistent = value;
      return 0;
    case FLAME_NLTGV2_OPT_PLACEMENT:
      if (value < 0 || value > 1) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_place = value;
      return 0;
    case FLAME_NLTGV2_OPT_POLL_GAP:
      if (value < 0 || value > 4) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_poll_gap = value;
      return 0;
    case FLAME_NLTGV2_OPT_VERIFY_RECORDS:
      if (value < 0 || value > 2) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_verify = value;
      if (value != 2) ctx->persist_refused_topo = ~0ull;  // hook off: let the persistent path be tried again
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
      if (value == 0) ctx->persist_refused_topo = ~0ull;  // let the persistent path be tried again
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
    case FLAME_NLTGV2_OPT_TV_LDS:
      if (value < 0 || value > 2) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_tv_lds = value;
      return 0;
    case FLAME_NLTGV2_OPT_UNROLL:
      if (value != 0 && value != 4 && value != 8 && value != 16) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
      ctx->opt_unroll = value;
      return 0;
    default:
      return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  }
}

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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  rc = upload_topology(ctx, g, state, sizeof(state) / sizeof(state[0]), /*long_lived=*/true);
  if (rc) return rc;
  const auto t_packed = std::chrono::steady_clock::now();
  LAUNCHCHK(ctx, launch_pack_static(ctx->c, ctx->f, ctx->stream));
  // No wait here: the caller's arrays were copied into the staging buffer, the device work is ordered on the stream in
  // front of whatever comes next (a run, an export), and the next upload synchronises before it reuses the staging buffer.
  if (trace) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const auto t_end = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[flame_nltgv2] upload_graph V=%d E=%d: host tables + enqueue %.3f ms, device %.3f ms\n", V, E,
                 std::chrono::duration<double, std::milli>(t_packed - t_begin).count(),
                 std::chrono::duration<double, std::milli>(t_end - t_packed).count());
  }
  ctx->h_feat.resize((size_t)V);
  for (int32_t v = 0; v < V; ++v) ctx->h_feat[(size_t)v] = v;  // default feature id = vertex index
  ctx->feat_map_valid = false;
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
int flame_nltgv2_sync_graph(flame_nltgv2_ctx* ctx, const flame_nltgv2_sync_input* in) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_sync_graph");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!in || in->V < 0 || in->E < 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  const int32_t V = in->V, E = in->E;
  if (V > 0 && (!in->feat_id || !in->pos || !in->data_term || !in->data_weight)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (E > 0 && !in->edges) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);

  const bool trace = std::getenv("FLAME_NLTGV2_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  // The previous state stays on the device: the canonical arrays are brought up to date (a kernel, enqueued) while the
  // host works out the index maps below.
  rc = ensure_canon(ctx);
  if (rc) return rc;
  const int32_t Vo = ctx->L.V, Eo = ctx->L.E;

  for (int32_t v = 0; v < V; ++v)
    if (in->feat_id[v] < 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (!ctx->feat_map_valid) {  // (after an upload / set_feature_ids; otherwise the map the previous sync built)
    FlatMap& m = ctx->feat_maps[ctx->feat_cur];
    m.reset((size_t)Vo);
    for (int32_t v = 0; v < Vo; ++v) m.emplace((uint64_t)(uint32_t)ctx->h_feat[(size_t)v], v);
    ctx->feat_map_valid = true;
  }
  const FlatMap& old_of_feat = ctx->feat_maps[ctx->feat_cur];
  auto key = [](int32_t a, int32_t b) {
    const uint32_t lo = (uint32_t)std::min(a, b), hi = (uint32_t)std::max(a, b);
    return ((uint64_t)hi << 32) | lo;
  };

  // vertices: new vertex -> its index in the previous graph (-1: new)
  std::vector<int32_t>& old_of_new = ctx->h_old_of_new;
  old_of_new.assign((size_t)V, -1);
  FlatMap& seen = ctx->feat_maps[ctx->feat_cur ^ 1];  // ... and the next sync's old_of_feat
  seen.reset((size_t)V);
  for (int32_t v = 0; v < V; ++v) {
    if (!seen.emplace((uint64_t)(uint32_t)in->feat_id[v], v).second) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);  // duplicate id
    const int32_t* it = old_of_feat.find((uint64_t)(uint32_t)in->feat_id[v]);
    if (it) old_of_new[(size_t)v] = *it;
  }
  // edges.  A surviving edge joins two surviving vertices: it is looked up in the previous graph's adjacency (the
  // host copy of the packed layout: ~6 incident edges per vertex, one cache line) instead of a hash table over all
  // edges.  Survivors must come out in their PREVIOUS relative order (boost::edges() walks a std::list: erase keeps
  // the order of the rest): they are met in triangulator order, so they are parked in a table indexed by the old
  // edge id and read back in one pass -- no sort.
  struct Keep { int32_t a, b; };
  std::vector<Keep> keep_of_old((size_t)Eo, Keep{-1, -1});
  std::vector<std::pair<int32_t, int32_t>> fresh;
  fresh.reserve((size_t)E / 4 + 16);
  int32_t n_keep = 0;
  const std::vector<int32_t>& orow = ctx->L.row_ptr;
  const std::vector<uint32_t>& ohalf = ctx->L.half;
  const std::vector<int32_t>& onbr = ctx->L.half_nbr;
  for (int32_t k = 0; k < E; ++k) {
    const int32_t a = in->edges[2 * k], b = in->edges[2 * k + 1];
    if (a < 0 || a >= V || b < 0 || b >= V || a == b) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
    const int32_t oa = old_of_new[(size_t)a], ob = old_of_new[(size_t)b];
    int32_t e = -1;
    bool same = true;  // the old edge runs oa -> ob
    if (oa >= 0 && ob >= 0) {
      for (int32_t h = orow[(size_t)oa]; h < orow[(size_t)oa + 1]; ++h) {
        if (onbr[(size_t)h] == ob) {
          e = (int32_t)(ohalf[(size_t)h] & ~kRoleBit);  // the first (lowest id) of possible parallel edges, as
          same = (ohalf[(size_t)h] & kRoleBit) == 0u;   // boost::edge() on the list would find
          break;
        }
      }
    }
    if (e >= 0) {
      if (keep_of_old[(size_t)e].a >= 0) continue;  // the same pair again: boost::edge() finds the edge, nothing is added
      keep_of_old[(size_t)e] = Keep{same ? a : b, same ? b : a};
      ++n_keep;
    } else {
      fresh.emplace_back(a, b);
    }
  }
  if (!fresh.empty()) {  // no parallel edges among the new ones either (boost::edge() finds the one just added)
    FlatMap& dup = ctx->feat_maps[2];
    dup.reset(fresh.size());
    size_t n = 0;
    for (const auto& f : fresh)
      if (dup.emplace(key(in->feat_id[f.first], in->feat_id[f.second]), 1).second) fresh[n++] = f;
    fresh.resize(n);
  }
  const int32_t En = (int32_t)((size_t)n_keep + fresh.size());
  std::vector<int32_t> src((size_t)En), dst((size_t)En);
  std::vector<int32_t>& old_of_new_edge = ctx->h_old_of_new_edge;
  old_of_new_edge.assign((size_t)En, -1);
  int32_t e = 0;
  for (int32_t o = 0; o < Eo; ++o) {
    const Keep& kp = keep_of_old[(size_t)o];
    if (kp.a < 0) continue;
    src[(size_t)e] = kp.a, dst[(size_t)e] = kp.b;
    old_of_new_edge[(size_t)e] = o;
    ++e;
  }
  for (const auto& f : fresh) {
    src[(size_t)e] = f.first, dst[(size_t)e] = f.second;
    ++e;
  }
  const auto t1 = std::chrono::steady_clock::now();

  // New topology + the frame's inputs up, state gathered on the device out of the previous arrays into spare ones,
  // which then take their place.
  ctx->have_graph = false;  // (until the new graph stands)
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // nothing in flight uses buffers that may be reallocated below
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
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const auto t2 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[flame_nltgv2] sync_graph: index maps %.3f ms, tables + upload + device gather %.3f ms\n", ms(t0, t1), ms(t1, t2));
  }
  ctx->h_feat.assign(in->feat_id, in->feat_id + V);
  ctx->feat_cur ^= 1;  // (the map filled above is of the graph that stands now)
  ctx->canon_valid = true;
  ctx->fused_valid = false;
  ctx->have_prev = false;
  ctx->parity = 0;
  ctx->have_graph = true;
  ctx->last_error = 0;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_project_graph(flame_nltgv2_ctx* ctx, const flame_nltgv2_projection* pr, float graph_scale,
                               uint8_t* keep_out, float* pos_out) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_project_graph");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!pr || !(graph_scale > 0.0f) || (ctx->L.V > 0 && !keep_out)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  rc = ensure_canon(ctx);
  if (rc) return rc;
  const size_t V = (size_t)ctx->L.V;
  rc = ensure(ctx, ctx->r_valid, V + 16);
  if (rc) return rc;
  ProjectGeometry geo;
  std::memcpy(geo.K, pr->K, sizeof(geo.K));
  std::memcpy(geo.Kinv, pr->Kinv, sizeof(geo.Kinv));
  std::memcpy(geo.KRKinv, pr->KRKinv, sizeof(geo.KRKinv));
  std::memcpy(geo.q, pr->q_ref_to_cmp, sizeof(geo.q));
  std::memcpy(geo.t, pr->t_ref_to_cmp, sizeof(geo.t));
  geo.rx = pr->region_x, geo.ry = pr->region_y, geo.rw = pr->region_w, geo.rh = pr->region_h;
  LAUNCHCHK(ctx, launch_project_graph(ctx->c, graph_scale, geo, (uint8_t*)ctx->r_valid.p, ctx->stream));
  if (V) HIPCHK(ctx, hipMemcpyAsync(keep_out, ctx->r_valid.p, V, hipMemcpyDeviceToHost, ctx->stream));
  if (V && pos_out) HIPCHK(ctx, hipMemcpyAsync(pos_out, ctx->pos.p, sizeof(float) * 2 * V, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  ctx->h_feat.assign(feat_id, feat_id + ctx->L.V);
  ctx->feat_map_valid = false;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_get_topology(flame_nltgv2_ctx* ctx, int32_t* src, int32_t* dst, int32_t* feat_id) {
  if (!ctx) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (src) std::copy(ctx->h_src.begin(), ctx->h_src.end(), src);
  if (dst) std::copy(ctx->h_dst.begin(), ctx->h_dst.end(), dst);
  if (feat_id) std::copy(ctx->h_feat.begin(), ctx->h_feat.end(), feat_id);
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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->fused_valid = false;
  ctx->last_error = 0;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_run_async(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n_iters) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_run_async");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!params_ok(p) || n_iters < 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  return enqueue_run(ctx, p, n_iters);
}

int flame_nltgv2_sync(flame_nltgv2_ctx* ctx) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_sync");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  ctx->h_terms.resize(n);
  LAUNCHCHK(ctx, launch_cost_terms(ctx->c, (float*)ctx->cost_terms.p, ctx->stream));
  if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->h_terms.data(), ctx->cost_terms.p, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
    if (wait) {
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
  if (wait) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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

// Shared tail of the two interpolate_mesh entry points: triangles/validity -> device, rasterise, copy back.
static int interpolate_common(flame_nltgv2_ctx* ctx, const int32_t* triangles, int32_t T, int32_t V,
                              const uint8_t* vtx_valid, const uint8_t* tri_valid, const float2* d_vtx,
                              const float* d_val, float value_scale, int rows, int cols, float* out, int32_t* coverage) {
  if (T < 0 || rows <= 0 || cols <= 0 || !out || (T > 0 && !triangles)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  for (int32_t t = 0; t < 3 * T; ++t)
    if (triangles[t] < 0 || triangles[t] >= V) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  const size_t n = (size_t)rows * (size_t)cols;
  int rc = ensure(ctx, ctx->r_tris, sizeof(int32_t) * 3 * (size_t)T);
  if (!rc) rc = ensure(ctx, ctx->r_valid, (size_t)T + (size_t)V + 16);
  if (!rc) rc = ensure(ctx, ctx->r_keys, sizeof(unsigned long long) * n);
  if (!rc) rc = ensure(ctx, ctx->r_img, sizeof(float) * n);
  if (!rc) rc = ensure(ctx, ctx->r_cov, sizeof(int));
  if (rc) return rc;
  if (T > 0) HIPCHK(ctx, hipMemcpyAsync(ctx->r_tris.p, triangles, sizeof(int32_t) * 3 * (size_t)T, hipMemcpyHostToDevice, ctx->stream));
  uint8_t* d_tv = nullptr;
  uint8_t* d_vv = nullptr;
  if (tri_valid && T > 0) {
    d_tv = (uint8_t*)ctx->r_valid.p;
    HIPCHK(ctx, hipMemcpyAsync(d_tv, tri_valid, (size_t)T, hipMemcpyHostToDevice, ctx->stream));
  }
  if (vtx_valid && V > 0) {
    d_vv = (uint8_t*)ctx->r_valid.p + (size_t)T;
    HIPCHK(ctx, hipMemcpyAsync(d_vv, vtx_valid, (size_t)V, hipMemcpyHostToDevice, ctx->stream));
  }
  LAUNCHCHK(ctx, launch_interpolate_mesh(T, (const int32_t*)ctx->r_tris.p, d_vtx, d_val, value_scale, d_vv, d_tv,
                                         (unsigned long long*)ctx->r_keys.p, (float*)ctx->r_img.p, (int*)ctx->r_cov.p,
                                         rows, cols, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->r_img.p, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  int cov = 0;
  HIPCHK(ctx, hipMemcpyAsync(&cov, ctx->r_cov.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (coverage) *coverage = cov;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_interpolate_mesh(flame_nltgv2_ctx* ctx, const int32_t* triangles, int32_t T, const uint8_t* tri_valid,
                                  int rows, int cols, float graph_scale, float* idepthmap_out, int32_t* coverage_out) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_interpolate_mesh");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  rc = ensure_canon(ctx);
  if (rc) return rc;
  return interpolate_common(ctx, triangles, T, ctx->L.V, nullptr, tri_valid, ctx->c.pos, ctx->c.x, graph_scale, rows,
                            cols, idepthmap_out, coverage_out);
}

int flame_nltgv2_interpolate_mesh_arrays(flame_nltgv2_ctx* ctx, const int32_t* triangles, int32_t T, const float* vertices_xy,
                                         const float* values, int32_t V, const uint8_t* vtx_valid,
                                         const uint8_t* tri_valid, int rows, int cols, float* img_out,
                                         int32_t* coverage_out) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_interpolate_mesh_arrays");
  int rc = enter(ctx);
  if (rc) return rc;
  if (V < 0 || (V > 0 && (!vertices_xy || !values))) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  rc = ensure(ctx, ctx->r_vtx, sizeof(float) * 2 * (size_t)V);
  if (!rc) rc = ensure(ctx, ctx->r_val, sizeof(float) * (size_t)V);
  if (rc) return rc;
  if (V > 0) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->r_vtx.p, vertices_xy, sizeof(float) * 2 * (size_t)V, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->r_val.p, values, sizeof(float) * (size_t)V, hipMemcpyHostToDevice, ctx->stream));
  }
  return interpolate_common(ctx, triangles, T, V, vtx_valid, tri_valid, (const float2*)ctx->r_vtx.p,
                            (const float*)ctx->r_val.p, 1.0f, rows, cols, img_out, coverage_out);
}

int flame_nltgv2_photo_set_images(flame_nltgv2_ctx* ctx, const uint8_t* ref, const uint8_t* cmp, int rows, int cols,
                                  int step_bytes) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ref || !cmp || rows < 2 || cols < 2 || step_bytes < cols) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  const size_t bytes = (size_t)rows * (size_t)step_bytes;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  rc = ensure(ctx, ctx->img_ref, bytes + 16);
  if (!rc) rc = ensure(ctx, ctx->img_cmp, bytes + 16);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->img_ref.p, ref, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->img_cmp.p, cmp, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->img_rows = rows, ctx->img_cols = cols, ctx->img_step = step_bytes;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_photo_residual(flame_nltgv2_ctx* ctx, const float* KRKinv, const float* Kt, float graph_scale,
                                int border, float* err_out) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!KRKinv || !Kt || !err_out || border < 1 || ctx->img_rows == 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  rc = ensure_canon(ctx);
  if (rc) return rc;
  const size_t fV = sizeof(float) * (size_t)ctx->L.V;
  rc = ensure(ctx, ctx->photo_err, fV);
  if (rc) return rc;
  PhotoGeometry geo;
  std::memcpy(geo.KRKinv, KRKinv, sizeof(geo.KRKinv));
  std::memcpy(geo.Kt, Kt, sizeof(geo.Kt));
  LAUNCHCHK(ctx, launch_photo_residual(ctx->c, graph_scale, geo, (const uint8_t*)ctx->img_ref.p,
                                       (const uint8_t*)ctx->img_cmp.p, ctx->img_rows, ctx->img_cols, ctx->img_step,
                                       border, (float*)ctx->photo_err.p, ctx->stream));
  if (fV) HIPCHK(ctx, hipMemcpyAsync(err_out, ctx->photo_err.p, fV, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_photo_fuse(flame_nltgv2_ctx* ctx, const float* KRKinv, const float* Kt, float graph_scale, int border,
                            int enable) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!enable) {
    ctx->photo_fused = false;
    return FLAME_NLTGV2_OK;
  }
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!KRKinv || !Kt || border < 1 || ctx->img_rows == 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  rc = ensure(ctx, ctx->photo_err, sizeof(float) * (size_t)ctx->L.V);
  if (rc) return rc;
  std::memcpy(ctx->photo_geo.KRKinv, KRKinv, sizeof(ctx->photo_geo.KRKinv));
  std::memcpy(ctx->photo_geo.Kt, Kt, sizeof(ctx->photo_geo.Kt));
  ctx->photo_scale = graph_scale, ctx->photo_border = border;
  ctx->photo_fused = true;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_photo_residual_last(flame_nltgv2_ctx* ctx, float* err_out) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!ctx->photo_fused || !err_out) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (ctx->pending.active) {
    rc = finish(ctx);
    if (rc) return rc;
  }
  const size_t fV = sizeof(float) * (size_t)ctx->L.V;
  if (fV) HIPCHK(ctx, hipMemcpyAsync(err_out, ctx->photo_err.p, fV, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLAME_NLTGV2_OK;
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
  info->he_waves = ctx->L.wg_ok ? ctx->L.wg_count : 0;  // (the same greedy walk as the patches)
  if (ctx->have_graph && !ctx->tv_built) {  // the vertex-per-lane rows are built on demand; a caller sizing a batch asks here
    ctx->L.tv_waves = 0;
    build_tv_rows(&ctx->L);  // host table only; the upload happens when the form is first used
  }
  info->tv_waves = ctx->L.tv_ok ? ctx->L.tv_waves : 0;
  info->patches = ctx->L.wg_ok ? ctx->L.wg_count : 0;
  info->tv_wave_capacity = (ctx->opt_tv_lds ? kTvLdsWavesPerCu : kTvWavesPerCu) * ctx->prop.multiProcessorCount;
  info->last_run_groups = ctx->last_run_groups;
  info->timeouts_recovered = ctx->timeouts_recovered;
  info->torn_records_detected = ctx->torn_records_detected;
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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  if (rc) return rc;
  const int32_t V = ctx->L.V, E = ctx->L.E;
  std::vector<float> pos(2 * (size_t)V);
  HIPCHK(ctx, hipMemcpyAsync(pos.data(), ctx->pos.p, sizeof(float) * pos.size(), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  if (bad == 0) {
    const size_t n = (size_t)L.rows * kWave, lanes = (size_t)L.wg_count * kWave;
    int e = cmp(ctx->rec_nbr, H.rec_nbr.data(), 4 * n) | cmp(ctx->rec_edge, H.rec_edge.data(), 4 * n) |
            cmp(ctx->edge_src_slot, H.edge_src_slot.data(), 4 * (size_t)E) | cmp(ctx->perm, H.perm.data(), 4 * H.perm.size()) |
            cmp(ctx->slice_row, H.slice_row.data(), 4 * H.slice_row.size());
    if (L.wg_ok)
      e |= cmp(ctx->wg_slot, H.wg_slot.data(), 4 * lanes) | cmp(ctx->wg_vid, H.wg_vid.data(), 4 * lanes) |
           cmp(ctx->wg_meta, H.wg_meta.data(), 4 * lanes) | cmp(ctx->wg_nbr, H.wg_nbr.data(), 4 * lanes) |
           cmp(ctx->wg_fetch, H.wg_fetch.data(), 4 * lanes) | cmp(ctx->wg_info, H.wg_info.data(), 4 * H.wg_info.size());
    if (ctx->he_built)  // (C), converted on the device from (E), against the host's own walk
      e |= cmp(ctx->he_slot, H.he_slot.data(), 4 * H.he_slot.size()) | cmp(ctx->he_vid, H.he_vid.data(), 4 * H.he_vid.size()) |
           cmp(ctx->he_meta, H.he_meta.data(), 4 * H.he_meta.size()) |
           cmp(ctx->he_wave_chain, H.he_wave_chain.data(), 4 * H.he_wave_chain.size());
    if (ctx->he_built) bad += (H.he_waves != L.he_waves) + (H.he_max_chain != L.he_max_chain) + (H.comp_he_wave != L.comp_he_wave);
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
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
