// nltgv2_context.hpp -- internal to libflame_nltgv2_hip.so: the solver context behind flame_nltgv2_ctx (include/flame_nltgv2.h) and the
// host-side functions the C-ABI translation units share.  Host side of the drop-in boundary for
// flame::optimizers::nltgv2_l1_graph_regularizer (/root/reference/src/flame/optimizers/nltgv2_l1_graph_regularizer.h:134-168).
// The context owns the device image of one reference Graph (h:107-112) in two forms:
//   canonical  SoA arrays in the caller's vertex/edge order  (upload/download, the individually callable
//              dual/primal/extragradient sweeps, costs)
//   packed     SELL-64 layout of the fused one-kernel-per-step sweep and the row layouts of the persistent kernels (run)
// and converts between them on the device only when the other form is asked for.
//   nltgv2_context.hip     buffers, uploads of a topology, context life cycle, options, info, self-tests
//   nltgv2_run.hip         which kernels run n steps (planner), enqueue / settle / roll back, the run entry points
//   nltgv2_graph_capi.hip  graph in / out: upload, per-frame sync, projection, rescale, state download, costs, export
//   nltgv2_frame_capi.hip  the rows around the solver that work on its device state: mesh rasteriser, photometric residual
#ifndef FLAME_AMD_NLTGV2_CONTEXT_HPP_
#define FLAME_AMD_NLTGV2_CONTEXT_HPP_

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <new>
#include <memory>
#include <mutex>
#include <functional>
#include <vector>

#include "flame_nltgv2.h"
#include "flame_nltgv2_test_options.h"
#include "nltgv2_kernels.h"
#include "nltgv2_pack.hpp"
#include "roctx_ranges.hpp"

namespace flame_hip {
namespace host {

constexpr int kGraphChunk = 256;      // steps per captured hipGraph (even: keeps ping-pong parity)
constexpr int kMaxCachedGraphs = 6;
constexpr int kTvLdsWavesPerCu = 16;   // ... with the slot constants in LDS (<= 128 VGPRs -> 4 waves per SIMD; 10 KB LDS per wave = all 160 KB)
constexpr size_t kXbufBytesPerVertex = 8 * 16 + 4;  // exchange buffers: up to four step buffers x (remote + same-XCD copy) of
                                                    // 16-byte records (the patch-per-wave form; the others use two) + the XCC table
constexpr int kPv2FromPerCu = 14;       // from this many one-half-edge patches per CU on, the two-half-edges-per-lane form (k_persistent_pv2) runs the graph
constexpr int kPv2PaceAbovePerCu = 10, kPv2DensePreSleep = 2, kPv2DenseGap = 2;  // its polls are paced from this many of its waves per CU (x64 cycles before the first poll / between rounds)
constexpr int kPv2MaxGroups = 2;        // ... in at most this many launch groups of whole frames (11-20 frames of 640x480: 9-18 % faster than the vertex-per-lane form)
constexpr int kPv2WavesPerCu = 19;     // ... up to this many of ITS waves per CU (20 really resident: 91 VGPRs)
constexpr int kCrowdedWavesPerCu = 16, kCrowdedTopologies = 64;  // (see flame_nltgv2_ctx::crowded_until_topo)
constexpr int kPvDensePerCu = 27;      // k_persistent_pv is used up to this many patches per CU (28 are resident: 7 waves per SIMD at <= 96 SGPRs)
constexpr int kPvPaceAbovePerCu = 13;  // ... and above this many its polls are paced (kPvDensePreSleep, kPvDenseGap)
constexpr int kPvDensePreSleep = 3, kPvDenseGap = 2;  // x64 cycles before the first poll of a step / between poll rounds (re-swept
                                                      // with the issue priorities in: 8 / 4 before them; profiles/r03_priority.txt)
constexpr int kPvPaceMoreAbovePerCu = 23;  // ... and above this many more slowly (a 1080p frame: 25 per CU)
constexpr int kPvDenserPreSleep = 5, kPvDenserGap = 4;
constexpr int kPvPreSleep = 0;         // k_persistent_pv: x64 cycles between a step's start and its first poll
constexpr int kPvPollGap = 2;          // k_persistent_pv polls: re-loading only the fetch entries still waiting, no pause between
                                       // rounds (with the round-2 first form of the kernel an s_sleep between rounds won by 1-3 %;
                                       // with the shorter hand-off path of its final form no pause wins by 3-4 % at 640x480)
constexpr int kDualMinWavesPerCu = 0;  // auto: exchange through the XCD's L2 when more waves than this share a CU
// x64-cycle sleep between publishing and the first neighbour poll of k_persistent_tv (insensitive, shortest wins)
constexpr int kPreSleepTv = 2;
constexpr int32_t kFeatDirectMax = 1 << 22;  // sync_graph: feature ids below this are looked up in a plain table (16 MB at most), others hashed
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

}  // namespace host
}  // namespace flame_hip

using namespace flame_hip;
using namespace flame_hip::host;

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
  DevBuf sp_ab[2];  // ... and spare alpha / beta (the device path's commit writes them before the solver has stopped)
  std::vector<int32_t> h_old_of_new, h_old_of_new_edge, h_old_edge_of_pair, h_first_pair_of_old, h_bucket_start, h_bucket_item, h_bucket_at;
  FlatMap feat_maps[3];     // [cur]: feature id -> vertex of the CURRENT graph (h_feat), kept from one sync to the next; the
  int feat_cur = 0;         // next sync fills the other one; [2]: scratch of a sync (new edges' duplicates)
  bool feat_map_valid = false;
  std::vector<int32_t> feat_tab;     // the same map as a plain table id -> vertex (-1: absent) while the ids stay below kFeatDirectMax
  std::vector<uint32_t> feat_stamp;  // duplicate check of a sync: the sync (feat_stamp_now) that last saw the id
  uint32_t feat_stamp_now = 0;
  bool feat_tab_valid = false;
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
  uint32_t tag_next = 1;  // persistent run: tag of the current bar values (monotonic)
  int last_run_path = 0, last_run_groups = 0;
  uint64_t persist_refused_topo = ~0ull;  // topology for which the runtime refused the persistent grid
  // ... and after an EXPIRED run of a topology the persistent path is not tried for its next `persist_backoff_left` runs: 4 after the
  // first expired run in a row, doubling up to 1024 (a stall that passes does not leave a static graph on the per-step path for good;
  // a GPU that stays full costs one expired run per 1024).  flame_nltgv2_info::timeouts_recovered counts them.
  uint64_t persist_backoff_topo = ~0ull;
  int persist_backoff_left = 0, persist_timeout_streak = 0;
  // finish() is redoing an expired chain: 1 = once more in a persistent form, at reduced residency (at most kCrowdedWavesPerCu waves per
  // CU, the test hook's fault off) -- an expired wait mostly means that somebody else's kernels held wave slots for a moment, and the
  // per-step path is 5x slower --, 2 = one launch per step, whatever the planner would say (the second attempt, which cannot expire)
  int replaying = 0;
  // A persistent run that needs most of the chip's wave slots (a 1080p frame: 25 of the 28 a CU really holds) only starts
  // whole when nothing else keeps slots busy; when such a run expires beside other kernels (the tracker's, the rasteriser's:
  // tools/soak_pipeline.py at 1080p), the next kCrowdedTopologies topologies are planned for at most kCrowdedWavesPerCu waves per
  // CU -- for a 1080p frame that is the vertex-per-lane form (4 waves per CU) -- then the full residency is tried again.
  uint64_t crowded_until_topo = 0;
  int last_run_waves_per_cu = 0;
  bool static_stale = false;  // pos changed on the device (project_graph): packed alpha/dx/dy need a re-pack
  uint64_t coop_checked_key = 0;  // (topology, form) whose persistent grid the runtime has verified as resident
  int buf_gen = 0;                // which of the two (hq, vstate) copies is current; part of the hipGraph cache key
  int replay_attempt = 0;         // finish(): how often the chain being settled has already been taken back (0 outside finish)
  int replays_per_step = 0;       // ... chains whose persistent replay expired as well and that went to the per-step path
  int timeouts_recovered = 0;     // persistent runs that timed out and were redone (persistently at reduced residency, then per step)
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
  std::vector<uint16_t> place_rank_host;  // the ranking of place_pool's pages (travels with the pool: nltgv2_run.hip, place_pool_release)
  // The persistent run(s) in flight, until finish() has seen the error word: what is needed to take them back.  One
  // run is taken back by swapping the buffer roles (it wrote the other copies).  When more work is enqueued before the
  // first run has been checked (run_async back to back: the frame loop, bench.py), the state the chain started from is
  // copied aside first (three device-to-device copies, once per chain), and the chain is kept as a list of operations:
  // a wait that expires anywhere in it restores that state and replays the list on the one-launch-per-step path.
  struct PendingOp {
    int kind = 0;  // 0 run, 1 explicit export of x * scale
    flame_nltgv2_params params{};
    int n = 0;
    bool open = false;      // an open run (flame_nltgv2_run_open): n is its upper bound until finish() has read how far it went
    uint32_t tag0 = 0;      // ... its first tag: err[12] / err[13] of the run count from it
    float* dst = nullptr;  // kind 1: where to; kind 0: the standing export target the run was enqueued with (NULL: none)
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
  bool tv_built = false;  // layout (D) exists for the current topology (built on demand)
  bool wg2_built = false; // ... and layout (E2) (two half-edges per lane; experimental)
  Pv2Args pv2_args;
  int pv2_occ_lcap = -1;  // the wg2_lcap (LDS sizing) the two numbers below were derived for
  int pv2_occ = 0, pv2_occ_verify = 0;  // patches of k_persistent_pv2 really co-resident per CU (plain / record-verifying instance)
  uint64_t wg2_checked_topo = ~0ull;  // the device expansion's verdict (no patch with more than 64 foreign records) was read for this topology
  bool wg2_usable = false;
  // pinned staging buffers of the uploads and their device-side landing areas (one copy; k_scatter distributes): [0] the context's
  // stream, [1] the side stream on which a frame sync is prepared while the solver runs (nltgv2_topo_capi.hip)
  struct StageSlot {
    void* h = nullptr;
    size_t cap = 0;
    DevBuf d;
  } stage[2];

  PackedLayout L;
  std::vector<int32_t> h_src, h_dst, h_feat;  // host image of the current topology (for sync_graph)
  // After a device-side build (nltgv2_topo_capi.hip) only L's scalars and h_feat are current; the vectors of L and h_src / h_dst are
  // brought up to date on demand (ensure_host_layout: the host sync path, layouts built on the host on demand, get_topology, the
  // self-test).
  // The layout is a function of the positions it was built from (Morton walk).  Flame::projectGraph moves the positions between a
  // build and the next sync: it then first copies them aside (layout_pos), so that the host image of the layout -- and the self-test --
  // can be formed from what the device builder saw.
  DevBuf layout_pos;
  bool layout_pos_saved = false;
  bool host_layout_valid = true;
  uint64_t tv_counted_topo = ~0ull;  // get_info counted the vertex-per-lane waves of this topology
  int opt_mesh_state = 0;         // interpolate_mesh_begin beside runs in flight: 0 settles them, 1 reads the canonical arrays as the last settle left them
  uint64_t snap_topo = ~0ull;     // ev_snap was recorded for this topology (enqueue_run, when the first run after a settle goes out)
  hipEvent_t ev_snap = nullptr;
  int opt_cost_sum = 0;           // flame_nltgv2_costs: 0 sequential sums in the reference's edge / the caller's vertex order (host), 1 k_block_sum
  int opt_sync_path = 0;          // 0 auto (device where it applies), 1 host index maps + host tables, 2 device or error
  int last_sync_path = 0;         // 1 host, 2 device
  // the device's feature tables: id -> vertex of the graph of generation feat_gen in table feat_gen & 1 (nltgv2_topo.hip: feat_insert / feat_lookup)
  DevBuf feat_stamp_d, feat_key_d, feat_val_d, topo_scratch, topo_dims;
  std::vector<int32_t> dup_key;      // prepared sync: duplicate check of the caller's feature ids (a stamped hash set)
  std::vector<uint32_t> dup_stamp;
  uint32_t dup_now = 0;
  // A frame sync is PREPARED beside the running solver (flame_nltgv2_sync_prepare): the builder reads the live topology and writes
  // the next one into these; flame_nltgv2_sync_commit swaps them with their live counterparts (nx_live) once the solver has stopped.
  enum { NX_POS, NX_SRC, NX_DST, NX_ROW_PTR, NX_HALF, NX_ORDER_M, NX_RID_OF, NX_PERM, NX_IPERM, NX_PDEG, NX_SLICE_ROW, NX_WG_INFO, NX_WG_V0,
         NX_WG_VFIRST, NX_WG2_INFO, NX_WG2_VFIRST, NX_DATA, NX_WEIGHT, NX_COUNT };
  DevBuf nx[NX_COUNT];
  DevBuf* nx_live(int i) {
    DevBuf* const live[NX_COUNT] = {&pos, &src, &dst, &row_ptr, &half, &order_m, &rid_of, &perm, &iperm, &pdeg, &slice_row, &wg_info, &wg_v0,
                                    &wg_vfirst, &wg2_info, &wg2_vfirst, &data, &weight};
    return live[i];
  }
  // ... and the tables the EXPANSION of that topology writes (per slot, per lane, the records' places): filled on the side stream at
  // commit while the solver's last rounds still run on the live ones, swapped (ex_live) together with nx once the solver has stopped.
  enum { EX_REC_NBR, EX_REC_EDGE, EX_EDGE_SRC_SLOT, EX_WG_SLOT, EX_WG_VID, EX_WG_META, EX_WG_NBR, EX_WG_FETCH, EX_WG2_SLOT, EX_WG2_NBR,
         EX_WG2_VID, EX_WG2_META, EX_WG2_FETCH, EX_WG2_RMAX, EX_PLACE_REC_OFF, EX_COUNT };
  DevBuf ex[EX_COUNT];
  DevBuf* ex_live(int i) {
    DevBuf* const live[EX_COUNT] = {&rec_nbr, &rec_edge, &edge_src_slot, &wg_slot, &wg_vid, &wg_meta, &wg_nbr, &wg_fetch, &wg2_slot, &wg2_nbr,
                                    &wg2_vid, &wg2_meta, &wg2_fetch, &wg2_rmax, &place_rec_off};
    return live[i];
  }
  DevBuf place_patch_nx, place_fill_nx;  // placement scratch of that expansion (place_patch / place_fill's counters may be a run's own placement's)
  hipEvent_t ev_expanded = nullptr;      // recorded on the side stream behind it: the context's stream waits for it after the swap
  std::vector<char> prep_host;      // inputs of a prepared sync that will go the host way at commit
  const void *prep_vmap = nullptr, *prep_emap = nullptr, *prep_init = nullptr;  // (in topo_scratch) index maps / init values of the prepared sync
  hipStream_t raster_stream = nullptr;  // the side stream of interpolate_mesh_begin / _end
  hipEvent_t ev_canon = nullptr, ev_raster_done = nullptr;
  bool raster_inflight = false;       // the side stream still reads the canonical pos / x: the next unpack waits for ev_raster_done
  float* h_img = nullptr;             // pinned: the map of the last interpolate_mesh_begin (+ the coverage count behind it)
  size_t h_img_cap = 0;
  int map_rows = 0, map_cols = 0;     // the dense map resident in r_img (0: none)
  int img_pending_rows = 0, img_pending_cols = 0;  // the map an interpolate_mesh_begin left in h_img for its _end (0: none pending)
  hipStream_t topo_stream = nullptr;  // the side stream of a prepared sync
  hipEvent_t ev_topo_ready = nullptr; // recorded on the context's stream when a topology stands (upload, commit): the next builder waits for it
  struct PreparedSync {
    bool active = false, device = false;
    uint64_t topo = 0;               // the topology it was prepared against
    int32_t V = 0, E = 0;
    bool has_init = false;
    int32_t check_sticky = 0, edges_unique = 0, init_from_map = 0;
    float sticky_threshold = 0.0f, init_graph_scale = 0.0f;
    size_t off[6] = {};              // the inputs in stage[1].h: feat_id, edges, pos, data_term, data_weight, init_x
    std::chrono::steady_clock::time_point t_begin, t_enqueued;
  } prepared;
  int feat_tab_bits_d = 0;
  uint32_t feat_gen = 0;
  bool feat_dev_valid = false;
  TopoDims* h_dims = nullptr;     // pinned
  CanonArgs c;
  FusedArgs f;
  std::vector<DevBuf*> all;
  // canonical
  DevBuf pos, x, w1, w2, xb, w1b, w2b, xp, w1p, w2p, data, weight, src, dst, alpha, beta, q1, q2, q3, row_ptr, half;
  // packed
  DevBuf slice_row, perm, pdeg, rec_nbr, rec_edge, edge_src_slot, hrec, hq, vstate, vaux, bar0, bar1, vprev;
  DevBuf cost_terms;          // addends of smoothnessCost / dataCost
  // RunTail of the persistent kernels (standing export / photometric targets): kTailSlots copies on the device, a launch takes the one
  // that holds what it needs -- a target alternating between two rows (the double-buffered gather) costs no copy per step
  static constexpr int kTailSlots = 4;
  DevBuf run_tail;
  RunTail tail_sent[kTailSlots]{};
  bool tail_valid[kTailSlots] = {false, false, false, false};
  int tail_next = 0;          // the slot the next unseen RunTail overwrites
  // flame_nltgv2_stream_wait_run / _runs_in_flight: the last plain persistent launch of a run carries one of these two events (in turn) as
  // its own completion signal (hipExtLaunchKernel's stop event: no operation of its own on the solver's in-order queue)
  hipEvent_t ev_run[2] = {nullptr, nullptr};
  int run_ev_pick = -1, run_ev_last = 0;          // the event the run being enqueued may bind / the one of the last enqueued run
  bool run_ev_valid[2] = {false, false};         // the event stands for a run (bound to its launch, or recorded behind it)
  bool track_runs = false;                       // somebody asks runs_in_flight: a run whose launch cannot carry the event gets it recorded
  bool run_event_bound = false;  // ev_run[run_ev_last] was carried by the last enqueued run's own launch ...
  uint64_t call_seq = 0, run_event_seq = 0;  // ... as long as no other call into the context followed it (enter() counts the calls)
  std::vector<float> h_terms;
  DevBuf hq_alt, vstate_alt;  // the other copies of hq / vstate: a persistent run writes there, success swaps the roles
  DevBuf xbuf, abort_flag, tv_slot, tv_vid, tv_meta, tv_wave, wg2_slot, wg2_vid, wg2_meta, wg2_nbr, wg2_fetch, wg2_info, wg2_vfirst, wg2_rmax;
  DevBuf wg_slot, wg_vid, wg_meta, wg_nbr, wg_fetch, wg_info, wg_v0, wg_vfirst, probe, progress;
  // misc
  DevBuf err, cost_out, img_ref, img_cmp, photo_err, r_tris, r_valid, r_tvalid, r_keys, r_img, r_cov, r_vtx, r_val;
  int img_rows = 0, img_cols = 0, img_step = 0;
  int* h_err = nullptr;    // pinned, kErrBytes
  int last_expired[16] = {0};  // what the most recent expired wait reported (report_expired)
  float* h_cost = nullptr; // pinned
  uint8_t* h_keep = nullptr;  // pinned: project_graph's keep mask, written by its kernel
  unsigned* h_stop = nullptr; // pinned: the open run's tag0, which request_open_stop copies into stop_dev when the host wants the state
  DevBuf stop_dev;            // the word an open run's deciding patch looks at (device memory: a word in host memory cost that patch -- and with it
                              // the whole lock-step network -- a PCIe round trip per two steps: 1.38 against 0.95 us per iteration)
  hipStream_t ctl_stream = nullptr;  // ... the stream of that 4-byte copy
  bool open_inflight = false; // an open run is enqueued and unchecked (the last op of ctx->pending)
  bool open_stop_sent = false; // ... and has been asked to stop (request_open_stop)
  uint32_t open_tag0 = 0;      // ... its first tag: what a request to stop it says
  int want_open = 0;          // enqueue_run: the run being enqueued is to be an open one (flame_nltgv2_run_open sets it for its call)
  int last_open_iters = 0;    // how far the last open run went
  int64_t iters_total = 0;    // iterations applied to the state by every run since create (open runs: counted when they are settled)
  DevBuf pos_undo;            // project_graph: the positions as they stood (when layout_pos already holds older ones), to take a projection back
  size_t h_keep_cap = 0;
  std::vector<CachedGraph> graphs;
  size_t device_bytes = 0;
};

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

namespace flame_hip {
namespace host {

// ---- nltgv2_context.hip ------------------------------------------------------------------------------------------------
int fail(flame_nltgv2_ctx* ctx, int status);
int ensure(flame_nltgv2_ctx* ctx, DevBuf& b, size_t bytes);   // grows a device buffer geometrically, reused across frames
void drop_graphs(flame_nltgv2_ctx* ctx);
SolverParams to_sp(const flame_nltgv2_params* p);
int enter(flame_nltgv2_ctx* ctx);
void refresh_args(flame_nltgv2_ctx* ctx);                     // kernel argument blocks from the current buffers
int h2d(flame_nltgv2_ctx* ctx, DevBuf& b, const void* src, size_t bytes);
size_t records_capacity(const PackedLayout& L);
int wait_raster(flame_nltgv2_ctx* ctx);                       // the context's stream waits for an interpolate_mesh_begin still reading pos / x
// settles a pending run, unpacks the state if needed.  `behind` (optional): launches of the caller that only read / edit the canonical arrays
// and can be REDONE -- they go out right behind the unpack, before the host has seen how the runs ended (one wait of the host instead of
// two with the solver standing still).  *behind_state: 0 = not launched (nothing was in flight: the caller launches them now), 1 = launched
// and the runs went through (done), 2 = launched but the runs expired and were redone: the canonical STATE has been unpacked again, what
// else the launches wrote the caller puts right before it launches them once more.
enum { kBehindNotLaunched = 0, kBehindDone = 1, kBehindSpoiled = 2 };
int ensure_canon(flame_nltgv2_ctx* ctx, const std::function<int()>* behind = nullptr, int* behind_state = nullptr);
void request_open_stop(flame_nltgv2_ctx* ctx);  // an open run in flight is asked to stop (once); see ensure()
hipError_t wait_solver_stream(flame_nltgv2_ctx* ctx);  // hipStreamSynchronize(ctx->stream) behind such a request
int ensure_fused(flame_nltgv2_ctx* ctx);
bool params_ok(const flame_nltgv2_params* p);
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
// slot 0 on the context's stream by default; host_off (optional): where each piece lies in the slot's pinned buffer
int staged_h2d(flame_nltgv2_ctx* ctx, const StageCopy* cp, size_t n, const StageFill* fills = nullptr, size_t n_fills = 0, int slot = 0,
               hipStream_t stream = nullptr, size_t* host_off = nullptr);
int upload_topology(flame_nltgv2_ctx* ctx, const flame_nltgv2_graph* g, const StageCopy* extra, size_t n_extra, bool long_lived);
int topology_buffers(flame_nltgv2_ctx* ctx, bool want_e2, size_t n_wg_info, size_t n_wg_v0, size_t n_wg_vfirst, size_t n_wg2_info,
                     size_t n_wg2_vfirst);
// The tables an expansion reads (per vertex, per patch) and writes (per slot, per lane, the records' places): the live ones, or the spare
// sets of a prepared sync (nx / ex).
struct ExpandTables {
  CanonArgs c;   // V, E, row_ptr, half, src, dst
  FusedArgs f;   // n_slices, slice_row, perm, rec_nbr, rec_edge, edge_src_slot, wg_count, wg_info, wg_slot .. wg_fetch
  const int32_t *iperm = nullptr, *wg_v0 = nullptr, *order_m = nullptr, *rid_of = nullptr;
  const uint8_t* wg_vfirst = nullptr;
  int32_t* wg2_info = nullptr;
  const uint8_t* wg2_vfirst = nullptr;
  int32_t *wg2_slot = nullptr, *wg2_vid = nullptr, *wg2_nbr = nullptr, *wg2_fetch = nullptr;
  uint32_t* wg2_meta = nullptr;
  int* wg2_rmax = nullptr;
  int32_t* rec_off = nullptr;      // placement: [2 * stride] record offsets (nullptr: no placement)
  int32_t* place_patch = nullptr;  // ... its scratch: [stride] patch of a record + [stride] class bytes
  int* place_fill = nullptr;       // ... the page counters and cursors
  int per_xcd = 0;
  size_t stride = 0;
};
ExpandTables live_tables(flame_nltgv2_ctx* ctx);
bool placement_applies(const flame_nltgv2_ctx* ctx, const PackedLayout& L);
// early = true: the clears of the tables an expansion writes (the spare rows of rec_*, the patches' fetch maximum) are left out -- they
// were done on the spare set in front of the expansion (topo_commit)
void topology_fills(flame_nltgv2_ctx* ctx, bool want_e2, std::vector<StageFill>* fills, bool early = false);
void expansion_fills(const PackedLayout& L, void* rec_edge, void* rec_nbr, void* wg2_rmax, std::vector<StageFill>* fills);
int expand_launches(flame_nltgv2_ctx* ctx, const PackedLayout& L, const ExpandTables& t, bool want_e2, hipStream_t stream);
int topology_expand(flame_nltgv2_ctx* ctx, bool want_e2, bool launched = false);  // launched: by expand_launches on the tables now live
bool wants_e2(const flame_nltgv2_ctx* ctx);
bool wants_e2(const flame_nltgv2_ctx* ctx, const PackedLayout& L);
int ensure_host_layout(flame_nltgv2_ctx* ctx);  // the host image of the topology (ctx->L's vectors, h_src, h_dst) after a device build
// ---- nltgv2_topo_capi.hip: the per-frame sync with the topology built on the device
int topo_prepare(flame_nltgv2_ctx* ctx, const flame_nltgv2_sync_input* in, bool* applicable);  // builder enqueued on the side stream
int topo_upload(flame_nltgv2_ctx* ctx, const flame_nltgv2_graph* g, const StageCopy* extra, size_t n_extra, bool* done);  // upload_graph that way
int topo_commit(flame_nltgv2_ctx* ctx, bool* done);                                             // solver stopped, sets swapped, state gathered
int cancel_prepared(flame_nltgv2_ctx* ctx);  // before anything else changes the topology: waits for an in-flight builder, forgets it
int sync_graph_host(flame_nltgv2_ctx* ctx, const flame_nltgv2_sync_input* in);
void set_init_map(flame_nltgv2_ctx* ctx, bool on, SyncArgs* sa);  // nltgv2_graph_capi.hip: index maps + tables on the host
int ensure_form_rows(flame_nltgv2_ctx* ctx, int form);

// ---- nltgv2_run.hip ----------------------------------------------------------------------------------------------------
constexpr size_t kMaxChain = 256;  // operations enqueued behind an unchecked persistent run before the host settles it
bool persistent_eligible(flame_nltgv2_ctx* ctx, int n);
int prepare_run(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n);
PhotoFuse photo_target(const flame_nltgv2_ctx* ctx);
int enqueue_photo_sweep(flame_nltgv2_ctx* ctx, bool packed_current);
int enqueue_run(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n);
// reads the error word; rolls a failed persistent run back and redoes it.  unpack_behind: the unpack of the state is enqueued before the
// host waits (*unpacked: it was, and the runs went through -- the canonical arrays are current)
int finish(flame_nltgv2_ctx* ctx, bool unpack_behind = false, bool* unpacked = nullptr, const std::function<int()>* behind = nullptr,
           bool* behind_launched = nullptr);
int snapshot_chain_start(flame_nltgv2_ctx* ctx);
int place_records(flame_nltgv2_ctx* ctx, int per_xcd);        // record placement, once per topology (k_place_assign)
int place_calibrate(flame_nltgv2_ctx* ctx);                   // ... and the page ranking of the context's pool (measured, or taken over with a pool)
void place_pool_release(flame_nltgv2_ctx* ctx);               // the pool and its ranking to the next context of the device
bool cooperative_allowed();                                   // false under rocprofiler-sdk (nltgv2_run.hip): first launches are plain ones then
bool place_calibrate_at_create(int device);                    // flame_nltgv2_create: rank now (first context of the device, or a ranked pool is free) or at the first placed run

}  // namespace host
}  // namespace flame_hip

#endif  // FLAME_AMD_NLTGV2_CONTEXT_HPP_
