// nltgv2_pack.hpp -- host-side packing of a flat reference Graph image into the device layouts.
//
// Two layouts are derived from the caller's edge list (src[k] -> dst[k], k = boost::edges() order,
// /root/reference/src/flame/optimizers/nltgv2_l1_graph_regularizer.cc:91-96):
//
//  (A) canonical CSR of incident half-edges, per ORIGINAL vertex, ascending edge id.  Ascending
//      edge id is the order in which the reference's primalStep edge scatter (cc:120-142) touches a
//      given vertex, so a sequential per-vertex gather over this list reproduces the reference's
//      float accumulation order exactly.
//
//  (B) SELL-64 ("sliced ELLPACK, 64 = one gfx950 wavefront") layout for the fused sweep:
//      vertices are renumbered (connected component, then Morton order of pos for locality, then
//      by degree inside windows of 512 to equalise slice widths) and cut into slices of 64; slice s
//      owns rows [slice_row[s], slice_row[s+1]) of 64-lane-wide half-edge arrays, row k holding the
//      k-th incident half-edge (again ascending edge id) of each of its 64 vertices.  Lane l of a
//      wave therefore streams rec[(row+k)*64 + l]: fully coalesced, one vertex per lane.
//
// Pure C++17, no HIP: also used by flame_nltgv2_pack_probe for the CPU test-suite.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "flame_nltgv2.h"

#ifdef FLAME_PACK_PROFILE
#include <chrono>
double g_prof[8];
#define PROF_T(i) do { auto _n = std::chrono::steady_clock::now(); g_prof[i] += std::chrono::duration<double, std::milli>(_n - _t).count(); _t = _n; } while (0)
#define PROF_BEGIN auto _t = std::chrono::steady_clock::now();
#else
#define PROF_T(i)
#define PROF_BEGIN
#endif

namespace flame_hip {

constexpr int kWave = 64;           // gfx950 wavefront
constexpr int kDegreeWindow = 512;  // vertices per degree-sorting window (8 slices): a fixed grid over the walk positions, cut at
                                    // component borders -- a window is then one workgroup of the device builder (nltgv2_topo.hip)
constexpr int kWalkSegment = 256;   // the greedy patch walks of (E) / (E2) start a new patch at every multiple of this many walk
                                    // positions (and at every component): segments are walked independently -- one lane each on the
                                    // device (k_topo_walk) -- at the price of half a patch per segment (~1.8 % more patches)
constexpr int kRowPad = 16;         // spare rows at the end of the half-edge arrays (largest unroll)
constexpr uint32_t kRoleBit = 0x80000000u;  // set: the owning vertex is the TARGET (jj) of the edge

struct PackedLayout {
  int32_t V = 0, E = 0, n_slices = 0, max_degree = 0;
  int64_t rows = 0;                    // 64-wide rows actually used (excluding kRowPad)
  std::vector<int32_t> order_m;        // [V] the vertices in (component, Morton) order: the walk of (D) and (E)
  std::vector<int32_t> rid_of;         // [V] inverse of order_m: a vertex's position in the walk = its record id in (E)
  std::vector<int32_t> perm;           // [n_slices*64] packed slot -> original vertex (-1 padding)
  std::vector<int32_t> iperm;          // [V] original vertex -> packed slot
  std::vector<int32_t> pdeg;           // [n_slices*64] degree of packed vertex
  std::vector<int32_t> slice_row;      // [n_slices+1]
  std::vector<uint32_t> rec_nbr;       // [(rows+pad)*64] packed neighbour | role bit; empty: self
  std::vector<int32_t> rec_edge;       // [(rows+pad)*64] edge id, -1 empty
  std::vector<int32_t> edge_src_slot;  // [E] slot of the source-side copy of edge e
  std::vector<int32_t> row_ptr;        // [V+1]  (A)
  std::vector<uint32_t> half;          // [2E]   (A) edge id | role bit, ascending edge id per vertex
  std::vector<int32_t> half_nbr;       // [2E]   (A) the vertex at the other end of that half-edge (host only)
  // (D) one-vertex-per-lane rows of the register-resident persistent run (throughput form): a lane
  // holds up to kTvSlots half-edges; a vertex of higher degree occupies ceil(deg/kTvSlots) ADJACENT
  // lanes of one wave ("chain"), the last of which owns the vertex.
  bool tv_ok = false;
  int32_t tv_waves = 0;
  std::vector<int32_t> tv_slot;   // [tv_waves*kTvSlots*64] (wave, k, lane) -> SELL slot, -1 none
  std::vector<int32_t> tv_vid;    // [tv_waves*64] packed vertex the lane belongs to, -1 unused
  std::vector<uint32_t> tv_meta;  // [tv_waves*64] nslots | chain_idx<<4 | owner_lane<<10 | owner<<16 | valid<<17
  std::vector<uint32_t> tv_wave;  // [tv_waves] passes | has_chain<<8 | slots used in pass 0 <<16 | in later passes <<20
  // connected components (= frames of a batch) are contiguous in packed order and never share a wave of
  // (D) / a patch of (E): a batch too large to be resident at once is run group of components by group
  std::vector<int32_t> comp_start;     // [n_comp+1] first packed vertex of each component
  std::vector<int32_t> comp_tv_wave;   // [n_comp+1] first (D) wave of each component
  // (E) patch-per-wave rows of the persistent run (k_persistent_pv): one LANE per half-edge, the lanes of a vertex contiguous (ascending edge id) inside ONE wave
  // (isolated vertices get one idle lane), one wave = one compact
  // Morton patch of vertices.  Inside the patch the exchange goes through LDS; only the DISTINCT vertices of other
  // patches that the patch touches are fetched from memory (one lane each, sorted by record id), and only vertices
  // with a neighbour in another patch publish.  Records are numbered in walk order (rid), so a patch's records are
  // contiguous.
  bool wg_ok = false;
  int32_t wg_count = 0;                // patches
  int32_t wg_lcap = 0, wg_rcap = 0;    // most local vertices / fetched records of any patch (LDS sizing; wg_ok needs rcap <= 64)
  int32_t wg_slab_slots = 0;           // most (local vertices x slab stride) of any patch (LDS sizing)
  std::vector<int32_t> wg_slot;        // [wg_count*64] SELL slot of the lane's half-edge, -1 idle
  std::vector<int32_t> wg_vid;         // [wg_count*64] packed vertex owning the lane, -1 unused lane
  std::vector<uint32_t> wg_meta;       // [wg_count*64] first lane | deg<<6 | local index<<13 | flags<<24
  std::vector<int32_t> wg_nbr;         // [wg_count*64] neighbour: local index, or 0x80000000 | fetch index
  std::vector<int32_t> wg_fetch;       // [wg_count*64] record id fetched by this lane, -1 none
  std::vector<int32_t> wg_info;        // [wg_count*4] first record id, fetched records, local vertices (| kWgSlab), slab stride
                                       //               (row-packed patches: the patch's largest degree instead)
  // Row-packed patches (wg_rowpack): a vertex's lanes lie inside one 16-lane row of the wave (the walk fits each vertex into
  // the first of the wave's four rows that has room; one of more than 8 edges gets a row to itself), which is what lets the
  // kernel add a vertex's contributions up across lanes with DPP row shifts instead of through LDS.  A vertex of more than 16
  // edges starts a patch at lane 0 and fills whole rows (the kernel continues its sum row after row).  wg_vfirst[i] = first
  // lane, within its patch, of the i-th vertex of the walk (either way).
  bool wg_rowpack = false;
  std::vector<uint8_t> wg_vfirst;
  std::vector<int32_t> comp_wg;        // [n_comp+1] first patch of each component
  // (E2) patch-per-wave rows with TWO half-edges per lane (k_persistent_pv2): the same walk and record ids, a vertex of d edges
  // takes max(1, ceil(d / 2)) consecutive lanes of one 16-lane row (more than 8 lanes: a row to itself; the form is not
  // built for a graph with a vertex of more than 32 edges); lane first + j holds half-edges 2j (slot 0) and 2j + 1 (slot 1)
  // of the vertex in ascending edge id.  Half as many waves as (E) for the same graph.
  bool wg2_ok = false;
  int32_t wg2_count = 0, wg2_lcap = 0, wg2_rcap = 0;
  std::vector<int32_t> wg2_slot;       // [wg2_count*2*64] (patch, slot s, lane) -> SELL slot of the half-edge, -1 idle
  std::vector<int32_t> wg2_nbr;        // [wg2_count*2*64] neighbour of that half-edge: local index, or 0x80000000 | fetch index
  std::vector<int32_t> wg2_vid;        // [wg2_count*64] packed vertex owning the lane, -1 unused lane
  std::vector<uint32_t> wg2_meta;      // [wg2_count*64] first lane | lanes of the vertex<<6 | local index<<13 | flags<<24 (kWgActive: slot 0 is a half-edge)
  std::vector<int32_t> wg2_fetch;      // [wg2_count*64] record id fetched by this lane, -1 none
  std::vector<int32_t> wg2_info;       // [wg2_count*4] first record id, fetched records, local vertices, most lanes of a vertex
  std::vector<int32_t> comp_wg2;       // [n_comp+1] first patch of each component
  std::vector<uint8_t> wg2_vfirst;     // [V] first lane, within its patch, of the i-th vertex of the walk
  std::vector<int32_t> wg2_v0;         // [wg2_count] walk position of the patch's first vertex
  bool wg2_walked = false;             // the walk (pass 1) was done for this topology
  int32_t n_rec = 0;                   // record ids in use (= V: a record's id is its vertex's position in the walk)
  std::vector<int32_t> wg_v0;          // [wg_count] walk position of the patch's first vertex (= its first record id)
};
constexpr int32_t kWgSlab = 1 << 17;    // in wg_info[4p+2], row-packed layouts: this patch is not (it holds a vertex of > 16 edges)
constexpr uint32_t kWgTail = 1u << 24, kWgActive = 1u << 25, kWgValid = 1u << 26, kWgPublish = 1u << 27, kWgHead = 1u << 28;
constexpr int32_t kWgRow = 16;  // lanes of a DPP row
constexpr int kTvSlots = 8;
constexpr uint32_t kTvOwner = 1u << 16, kTvValid = 1u << 17;

constexpr uint32_t kHeTail = 1u << 12, kHeActive = 1u << 13, kHeValid = 1u << 14;

inline uint32_t morton_spread16(uint32_t v) {
  v &= 0xFFFFu;
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

// Where the next vertex of the walk goes in the current wave: lanes back to back, or -- row-packed (see PackedLayout::
// wg_rowpack) -- first fit into the wave's four 16-lane rows, a vertex's lanes contiguous inside one row.
struct WaveFit {
  bool rowpack = false;                                  // the layout is row-packed wherever the degrees allow
  bool rows = false;                                     // ... and so is the current wave
  int32_t fill = kWave;                                  // back to back: lanes used
  int32_t row[4] = {16, 16, 16, 16};                      // row-packed: lanes used per row
  // A new wave, for a vertex of `need` lanes first.
  void open(int32_t need) {
    (void)need;
    rows = rowpack;
    fill = 0;
    row[0] = row[1] = row[2] = row[3] = 0;
  }
  // first lane for a vertex of `need` lanes, or -1: the wave is full for it (or cannot take it)
  int32_t place(int32_t need) {
    if (!rows) {
      if (fill + need > kWave) return -1;
      const int32_t f = fill;
      fill += need;
      return f;
    }
    if (need > 16) {
      // a vertex of more than 16 edges starts a wave of its own at lane 0 and fills ceil(need / 16) whole rows: its sum runs row
      // after row (k_persistent_pv: the head of row r takes the running sums over from row r - 1); the other rows are packed as usual
      if (row[0] | row[1] | row[2] | row[3]) return -1;
      for (int32_t r = 0; r < (need + 15) / 16; ++r) row[r] = 16;
      return 0;
    }
    for (int32_t r = 0; r < 4; ++r) {
      if (need > 8) {  // a vertex of more than 8 edges gets a row to itself: its shifts 8.. then run over lanes of its own or
        if (row[r] != 0) continue;  // idle ones only, and need no mask of their own (k_persistent_pv)
        row[r] = 16;
        return 16 * r;
      }
      if (row[r] + need <= 16) {
        const int32_t f = 16 * r + row[r];
        row[r] += need;
        return f;
      }
    }
    return -1;
  }
};

// Returns FLAME_NLTGV2_OK or FLAME_NLTGV2_ERR_INVALID_ARG.
// ---- (D): needs (B)'s header (iperm, pdeg, slice_row), comp_start and order_m.  Built on demand: only the vertex-per-lane persistent
// form reads it.
inline void build_tv_rows(PackedLayout* L) {
  const int32_t V = L->V, maxdeg = L->max_degree;
  const std::vector<int32_t>& order_m = L->order_m;
  // ---- (D) one-vertex-per-lane rows, kTvSlots register slots per lane ------------------------------
  L->tv_ok = (maxdeg <= kTvSlots * kWave);
  L->tv_waves = 0;
  L->tv_slot.clear(), L->tv_vid.clear(), L->tv_meta.clear(), L->tv_wave.clear();
  L->comp_tv_wave.clear();
  if (L->tv_ok && V > 0) {
    int32_t fill = kWave;
    size_t next_comp = 0;
    for (int32_t i = 0; i < V; ++i) {
      const int32_t s = L->iperm[order_m[i]];
      const int32_t d = L->pdeg[s];
      const int32_t lanes = std::max(1, (d + kTvSlots - 1) / kTvSlots);
      const bool comp_begin = next_comp < L->comp_start.size() && L->comp_start[next_comp] == i;
      if (comp_begin) {
        L->comp_tv_wave.push_back(L->tv_waves);
        ++next_comp;
      }
      if (fill + lanes > kWave || comp_begin) {
        L->tv_slot.resize(L->tv_slot.size() + static_cast<size_t>(kTvSlots) * kWave, -1);
        L->tv_vid.resize(L->tv_vid.size() + kWave, -1);
        L->tv_meta.resize(L->tv_meta.size() + kWave, 0u);
        L->tv_wave.push_back(1u);
        L->tv_waves++;
        fill = 0;
      }
      const size_t w = static_cast<size_t>(L->tv_waves - 1);
      const int64_t row0 = L->slice_row[s / kWave];
      const int32_t owner_lane = fill + lanes - 1;
      for (int32_t c = 0; c < lanes; ++c) {
        const int32_t lane = fill + c;
        const int32_t k0 = c * kTvSlots;
        const int32_t ns = std::max(0, std::min(kTvSlots, d - k0));
        uint32_t m = static_cast<uint32_t>(ns) | (static_cast<uint32_t>(c) << 4) |
                     (static_cast<uint32_t>(owner_lane) << 10) | kTvValid;
        if (c == lanes - 1) m |= kTvOwner;
        L->tv_meta[w * kWave + lane] = m;
        L->tv_vid[w * kWave + lane] = s;
        for (int32_t k = 0; k < ns; ++k) {
          L->tv_slot[(w * kTvSlots + k) * kWave + lane] =
              static_cast<int32_t>((row0 + k0 + k) * kWave + (s % kWave));
        }
      }
      uint32_t& wi = L->tv_wave[w];
      const uint32_t passes = std::max<uint32_t>(wi & 0xffu, static_cast<uint32_t>(lanes));
      uint32_t k_first = (wi >> 16) & 15u, k_later = (wi >> 20) & 15u;
      k_first = std::max<uint32_t>(k_first, static_cast<uint32_t>(std::min(d, kTvSlots)));
      if (lanes > 1) k_later = std::max<uint32_t>(k_later, lanes > 2 ? kTvSlots : static_cast<uint32_t>(d - kTvSlots));
      wi = passes | ((passes > 1u || (wi & 0x100u)) ? 0x100u : 0u) | (k_first << 16) | (k_later << 20);
      fill += lanes;
    }
    L->comp_tv_wave.push_back(L->tv_waves);
  }
}

// ---- (E) patch-per-wave rows ------------------------------------------------------------------------
// order_m = the vertices in (component, Morton) order; needs (B) (iperm, pdeg, slice_row, rec_nbr) and comp_start.
inline void build_patch_rows(const flame_nltgv2_graph* g, PackedLayout* L, const std::vector<int32_t>& order_m, bool host_expand,
                             bool rowpack = true, int rowpack_max_patches = 0x7fffffff) {
  (void)g;
  const int32_t V = L->V;
  constexpr int32_t T = kWave;
  L->wg_ok = false;
  L->wg_count = 0, L->wg_lcap = 0, L->wg_rcap = 0, L->wg_slab_slots = 0;
  L->wg_slot.clear(), L->wg_vid.clear(), L->wg_meta.clear(), L->wg_nbr.clear(), L->wg_fetch.clear();
  L->wg_info.clear(), L->comp_wg.clear();
  L->n_rec = V;
  L->wg_v0.clear();
  L->wg_vfirst.clear();
  // (row packing fills a wave to ~54 of its 64 lanes: more, smaller patches.  That pays where a patch has a SIMD to itself
  //  or nearly; a graph too big for the patch-per-wave form keeps its lanes back to back, for the forms that then run it)
  L->wg_rowpack = rowpack && (static_cast<int64_t>(2) * L->E + V / 32) / 54 + 1 <= static_cast<int64_t>(rowpack_max_patches);
  if (L->max_degree > kWave || V <= 0) return;
  L->wg_vfirst.resize(static_cast<size_t>(V));
  L->wg_info.reserve(((static_cast<size_t>(2) * L->E + V) * 9 / 8 / T + L->comp_start.size() + 2) * 4);
  // pass 1 (host, per vertex): the greedy walk -- a vertex's lanes never straddle two waves, a component begins a
  // new wave.  A vertex's record id is its position in the walk (L->rid_of).  Per patch: first record id, vertex count,
  // slab stride (its largest degree rounded up to 4, at least 8).
  int32_t n_local = 0, max_deg = 1;
  WaveFit fit;
  fit.rowpack = L->wg_rowpack;
  bool slab_patch = !L->wg_rowpack;
  size_t next_comp = 0;
  auto close_patch = [&]() {
    if (L->wg_count == 0) return;
    if (!slab_patch) {
      L->wg_info[static_cast<size_t>(L->wg_count - 1) * 4 + 3] = max_deg;
      return;
    }
    if (L->wg_rowpack) L->wg_info[static_cast<size_t>(L->wg_count - 1) * 4 + 2] |= kWgSlab;
    const int32_t stride = std::max(8, (max_deg + 3) & ~3);
    L->wg_info[static_cast<size_t>(L->wg_count - 1) * 4 + 3] = stride;
    L->wg_slab_slots = std::max(L->wg_slab_slots, (stride + 1) * n_local);  // (+1: the kernel pads a vertex's slab, see there)
  };
  for (int32_t i = 0; i < V; ++i) {
    const int32_t o = order_m[i];
    const int32_t need = std::max(L->row_ptr[o + 1] - L->row_ptr[o], 1);
    const bool comp_begin = next_comp < L->comp_start.size() && L->comp_start[next_comp] == i;
    if (comp_begin) ++next_comp;
    int32_t fill = (comp_begin || i % kWalkSegment == 0) ? -1 : fit.place(need);
    if (fill < 0) {
      close_patch();
      if (comp_begin) L->comp_wg.push_back(L->wg_count);
      L->wg_info.resize(L->wg_info.size() + 4, 0);
      L->wg_info[static_cast<size_t>(L->wg_count) * 4] = i;
      L->wg_count++;
      n_local = 0, max_deg = 1;
      fit.open(need);
      slab_patch = !fit.rows;
      fill = fit.place(need);
    }
    max_deg = std::max(max_deg, need);
    L->wg_info[static_cast<size_t>(L->wg_count - 1) * 4 + 2] = ++n_local;  // (| kWgSlab when the patch is closed)
    L->wg_lcap = std::max(L->wg_lcap, n_local);
    L->wg_vfirst[static_cast<size_t>(i)] = static_cast<uint8_t>(fill);
  }
  close_patch();
  L->comp_wg.push_back(L->wg_count);
  L->wg_v0.resize(static_cast<size_t>(L->wg_count));
  for (int32_t q = 0; q < L->wg_count; ++q) L->wg_v0[static_cast<size_t>(q)] = L->wg_info[static_cast<size_t>(q) * 4];
  if (host_expand) {
    // pass 2 (the device does this in k_build_patch, nltgv2_layout.hip): the 64 lanes of every patch and its fetch list.
    // A neighbour is local iff its record id lies in the patch's range; the others are fetched -- one lane per DISTINCT
    // record, sorted by record id.
    const size_t lanes = static_cast<size_t>(L->wg_count) * T;
    L->wg_slot.assign(lanes, -1), L->wg_vid.assign(lanes, -1), L->wg_meta.assign(lanes, 0u), L->wg_nbr.assign(lanes, 0);
    L->wg_fetch.assign(lanes, -1);
    int32_t want[kWave];
    for (int32_t wg = 0; wg < L->wg_count; ++wg) {
      const size_t b = static_cast<size_t>(wg) * T;
      const int32_t r0 = L->wg_info[static_cast<size_t>(wg) * 4];
      const int32_t n_loc = L->wg_info[static_cast<size_t>(wg) * 4 + 2] & 0xffff, r1 = r0 + n_loc;
      const int32_t v0 = L->wg_v0[static_cast<size_t>(wg)];
      const int32_t* rid = L->rid_of.data();
      int32_t n_want = 0;
      for (int32_t i = 0; i < n_loc; ++i) {
        const int32_t o = order_m[v0 + i];
        for (int32_t h = L->row_ptr[o]; h < L->row_ptr[o + 1]; ++h) {
          const int32_t r = rid[L->half_nbr[h]];
          if (r < r0 || r >= r1) want[n_want++] = r;
        }
      }
      std::sort(want, want + n_want);
      n_want = static_cast<int32_t>(std::unique(want, want + n_want) - want);
      for (int32_t k = 0; k < n_want; ++k) L->wg_fetch[b + k] = want[k];
      L->wg_info[static_cast<size_t>(wg) * 4 + 1] = n_want;
      L->wg_rcap = std::max(L->wg_rcap, n_want);
      for (int32_t i = 0; i < n_loc; ++i) {
        const int32_t o = order_m[v0 + i];
        const int32_t s = L->iperm[o];
        const int32_t d = L->row_ptr[o + 1] - L->row_ptr[o], need = std::max(d, 1);
        const int32_t lane = L->wg_vfirst[static_cast<size_t>(v0 + i)];
        const int64_t row0 = L->slice_row[s / kWave];
        bool publishes = false;
        for (int32_t k = 0; k < need; ++k) {
          uint32_t m = static_cast<uint32_t>(lane) | (static_cast<uint32_t>(d) << 6) | (static_cast<uint32_t>(i) << 13) | kWgValid;
          if (k == need - 1) m |= kWgTail;
          if (k == 0) m |= kWgHead;
          if (k < d) {
            m |= kWgActive;
            L->wg_slot[b + lane + k] = static_cast<int32_t>((row0 + k) * kWave + (s % kWave));
            const int32_t r = rid[L->half_nbr[L->row_ptr[o] + k]];
            if (r >= r0 && r < r1) {
              L->wg_nbr[b + lane + k] = r - r0;
            } else {
              const int32_t fi = static_cast<int32_t>(std::lower_bound(want, want + n_want, r) - want);
              L->wg_nbr[b + lane + k] = static_cast<int32_t>(0x80000000u | static_cast<uint32_t>(fi));
              publishes = true;  // a neighbour outside the patch reads this vertex
            }
          }
          L->wg_vid[b + lane + k] = s;
          L->wg_meta[b + lane + k] = m;
        }
        if (publishes)
          for (int32_t k = 0; k < need; ++k) L->wg_meta[b + lane + k] |= kWgPublish;
      }
    }
  }
  L->wg_ok = true;  // (a patch has at most 64 half-edges, hence at most 64 distinct foreign records: one per lane)
}

// ---- (E2) two half-edges per lane: pass 1 (host, per vertex) -- the greedy walk; the lanes are expanded on the device
// (nltgv2_layout.hip: k_build_patch2) or, for the CPU tests and the selftest, by build_patch_rows2 below
inline void build_patch_walk2(PackedLayout* L) {
  const int32_t V = L->V;
  const std::vector<int32_t>& order_m = L->order_m;
  L->wg2_ok = false;
  L->wg2_count = 0, L->wg2_lcap = 0, L->wg2_rcap = 0;
  L->wg2_slot.clear(), L->wg2_nbr.clear(), L->wg2_vid.clear(), L->wg2_meta.clear(), L->wg2_fetch.clear(), L->wg2_info.clear();
  L->comp_wg2.clear();
  L->wg2_vfirst.clear(), L->wg2_v0.clear();
  L->wg2_walked = true;
  if (V <= 0 || L->max_degree > 32) return;
  std::vector<uint8_t>& vfirst = L->wg2_vfirst;
  std::vector<int32_t>& v0s = L->wg2_v0;
  vfirst.assign(static_cast<size_t>(V), 0);
  int32_t n_local = 0, max_l = 1;
  WaveFit fit;
  fit.rowpack = true;
  size_t next_comp = 0;
  for (int32_t i = 0; i < V; ++i) {
    const int32_t o = order_m[i];
    const int32_t d = L->row_ptr[o + 1] - L->row_ptr[o];
    const int32_t need = std::max((d + 1) / 2, 1);
    const bool comp_begin = next_comp < L->comp_start.size() && L->comp_start[next_comp] == i;
    if (comp_begin) ++next_comp;
    int32_t fill = (comp_begin || i % kWalkSegment == 0) ? -1 : fit.place(need);
    if (fill < 0) {
      if (L->wg2_count > 0) L->wg2_info[static_cast<size_t>(L->wg2_count - 1) * 4 + 3] = max_l;
      if (comp_begin) L->comp_wg2.push_back(L->wg2_count);
      L->wg2_info.resize(L->wg2_info.size() + 4, 0);
      L->wg2_info[static_cast<size_t>(L->wg2_count) * 4] = i;
      v0s.push_back(i);
      L->wg2_count++;
      n_local = 0, max_l = 1;
      fit.open(need);
      fill = fit.place(need);
    }
    max_l = std::max(max_l, need);
    L->wg2_info[static_cast<size_t>(L->wg2_count - 1) * 4 + 2] = ++n_local;
    L->wg2_lcap = std::max(L->wg2_lcap, n_local);
    vfirst[static_cast<size_t>(i)] = static_cast<uint8_t>(fill);
  }
  if (L->wg2_count > 0) L->wg2_info[static_cast<size_t>(L->wg2_count - 1) * 4 + 3] = max_l;
  L->comp_wg2.push_back(L->wg2_count);
  L->wg2_ok = L->wg2_count > 0;  // (unless a patch turns out to read more than 64 distinct foreign records: the expansion says)
}

// pass 2 on the host: the 64 lanes of every patch and its fetch list (the reference of the device expansion)
inline void build_patch_rows2(PackedLayout* L) {
  constexpr int32_t T = kWave;
  const std::vector<int32_t>& order_m = L->order_m;
  build_patch_walk2(L);
  if (!L->wg2_ok) return;
  L->wg2_ok = false;
  const std::vector<uint8_t>& vfirst = L->wg2_vfirst;
  const std::vector<int32_t>& v0s = L->wg2_v0;
  const size_t lanes = static_cast<size_t>(L->wg2_count) * T;
  L->wg2_slot.assign(2 * lanes, -1), L->wg2_nbr.assign(2 * lanes, 0), L->wg2_vid.assign(lanes, -1), L->wg2_meta.assign(lanes, 0u);
  L->wg2_fetch.assign(lanes, -1);
  std::vector<int32_t> want;
  const int32_t* rid = L->rid_of.data();
  for (int32_t wg = 0; wg < L->wg2_count; ++wg) {
    const size_t b = static_cast<size_t>(wg) * T, b2 = static_cast<size_t>(wg) * 2 * T;
    const int32_t r0 = L->wg2_info[static_cast<size_t>(wg) * 4];
    const int32_t n_loc = L->wg2_info[static_cast<size_t>(wg) * 4 + 2], r1 = r0 + n_loc;
    const int32_t v0 = v0s[static_cast<size_t>(wg)];
    want.clear();
    for (int32_t i = 0; i < n_loc; ++i) {
      const int32_t o = order_m[v0 + i];
      for (int32_t h = L->row_ptr[o]; h < L->row_ptr[o + 1]; ++h) {
        const int32_t r = rid[L->half_nbr[h]];
        if (r < r0 || r >= r1) want.push_back(r);
      }
    }
    std::sort(want.begin(), want.end());
    want.erase(std::unique(want.begin(), want.end()), want.end());
    const int32_t n_want = static_cast<int32_t>(want.size());
    if (n_want > T) return;  // (more distinct foreign records than lanes: the form cannot run this graph; wg2_ok stays false)
    for (int32_t k = 0; k < n_want; ++k) L->wg2_fetch[b + k] = want[static_cast<size_t>(k)];
    L->wg2_info[static_cast<size_t>(wg) * 4 + 1] = n_want;
    L->wg2_rcap = std::max(L->wg2_rcap, n_want);
    for (int32_t i = 0; i < n_loc; ++i) {
      const int32_t o = order_m[v0 + i];
      const int32_t s = L->iperm[o];
      const int32_t d = L->row_ptr[o + 1] - L->row_ptr[o], need = std::max((d + 1) / 2, 1);
      const int32_t lane = vfirst[static_cast<size_t>(v0 + i)];
      const int64_t row0 = L->slice_row[s / kWave];
      bool publishes = false;
      for (int32_t j = 0; j < need; ++j) {
        uint32_t m = static_cast<uint32_t>(lane) | (static_cast<uint32_t>(need) << 6) | (static_cast<uint32_t>(i) << 13) | kWgValid;
        if (j == need - 1) m |= kWgTail;
        if (j == 0) m |= kWgHead;
        for (int32_t sl = 0; sl < 2; ++sl) {
          const int32_t k = 2 * j + sl;
          if (k >= d) continue;
          if (sl == 0) m |= kWgActive;
          L->wg2_slot[b2 + static_cast<size_t>(sl) * T + lane + j] = static_cast<int32_t>((row0 + k) * kWave + (s % kWave));
          const int32_t r = rid[L->half_nbr[L->row_ptr[o] + k]];
          if (r >= r0 && r < r1) {
            L->wg2_nbr[b2 + static_cast<size_t>(sl) * T + lane + j] = r - r0;
          } else {
            const int32_t fi = static_cast<int32_t>(std::lower_bound(want.begin(), want.end(), r) - want.begin());
            L->wg2_nbr[b2 + static_cast<size_t>(sl) * T + lane + j] = static_cast<int32_t>(0x80000000u | static_cast<uint32_t>(fi));
            publishes = true;
          }
        }
        L->wg2_vid[b + lane + j] = s;
        L->wg2_meta[b + lane + j] = m;
      }
      if (publishes)
        for (int32_t j = 0; j < need; ++j) L->wg2_meta[b + lane + j] |= kWgPublish;
    }
  }
  L->wg2_ok = true;
}

// host_expand = false: only what needs the host (per-vertex tables: (A), the walk order, (B)'s slice table, (E)'s patch
// walk); the per-slot / per-lane arrays of (B) and (E) are then produced on the device (nltgv2_layout.hip) and (D)
// on demand.  host_expand = true: everything here -- the reference the device expansion is checked against, and what the
// CPU test-suite looks at.
inline int build_layout(const flame_nltgv2_graph* g, PackedLayout* L, bool host_expand = true, bool rowpack = true,
                        int rowpack_max_patches = 0x7fffffff) {
  if (!g || g->V < 0 || g->E < 0) return FLAME_NLTGV2_ERR_INVALID_ARG;
  const int32_t V = g->V, E = g->E;
  if (V > 0 && !g->pos) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (E > 0 && (!g->src || !g->dst)) return FLAME_NLTGV2_ERR_INVALID_ARG;
  if (V >= (1 << 30)) return FLAME_NLTGV2_ERR_INVALID_ARG;
  L->V = V;
  L->E = E;
  PROF_BEGIN

  // ---- (A) canonical CSR, ascending edge id -------------------------------------------------
  L->row_ptr.assign(static_cast<size_t>(V) + 1, 0);
  for (int32_t k = 0; k < E; ++k) {
    const int32_t i = g->src[k], j = g->dst[k];
    if (i < 0 || i >= V || j < 0 || j >= V || i == j) return FLAME_NLTGV2_ERR_INVALID_ARG;
    L->row_ptr[i + 1]++;
    L->row_ptr[j + 1]++;
  }
  for (int32_t v = 0; v < V; ++v) L->row_ptr[v + 1] += L->row_ptr[v];
  L->half.assign(static_cast<size_t>(2) * E, 0);
  L->half_nbr.assign(static_cast<size_t>(2) * E, 0);
  {
    std::vector<int32_t> cur(L->row_ptr.begin(), L->row_ptr.end() - 1);
    for (int32_t k = 0; k < E; ++k) {
      const int32_t i = g->src[k], j = g->dst[k];
      L->half_nbr[cur[i]] = j;
      L->half[cur[i]++] = static_cast<uint32_t>(k);
      L->half_nbr[cur[j]] = i;
      L->half[cur[j]++] = static_cast<uint32_t>(k) | kRoleBit;
    }
  }
  int32_t maxdeg = 0;
  for (int32_t v = 0; v < V; ++v) maxdeg = std::max(maxdeg, L->row_ptr[v + 1] - L->row_ptr[v]);
  L->max_degree = maxdeg;

  PROF_T(0);
  // ---- vertex renumbering ---------------------------------------------------------------------
  // connected components (union-find): a batch of independent frames is a disjoint union and the
  // frames overlap in image coordinates, so Morton order alone would interleave them.
  std::vector<int32_t> parent(V);
  std::iota(parent.begin(), parent.end(), 0);
  auto find = [&](int32_t a) {
    while (parent[a] != a) {
      parent[a] = parent[parent[a]];
      a = parent[a];
    }
    return a;
  };
  for (int32_t k = 0; k < E; ++k) {
    int32_t a = find(g->src[k]), b = find(g->dst[k]);
    if (a != b) parent[std::max(a, b)] = std::min(a, b);  // root = smallest id of the component
  }
  float minx = 0, miny = 0, maxx = 1, maxy = 1;
  if (V > 0) {
    minx = maxx = g->pos[0];
    miny = maxy = g->pos[1];
    for (int32_t v = 0; v < V; ++v) {
      const float px = g->pos[2 * v], py = g->pos[2 * v + 1];
      if (!std::isfinite(px) || !std::isfinite(py)) return FLAME_NLTGV2_ERR_INVALID_ARG;
      minx = std::min(minx, px), maxx = std::max(maxx, px);
      miny = std::min(miny, py), maxy = std::max(maxy, py);
    }
  }
  const float sx = (maxx > minx) ? 65535.0f / (maxx - minx) : 0.0f;
  const float sy = (maxy > miny) ? 65535.0f / (maxy - miny) : 0.0f;
  std::vector<uint64_t> key(V);
  for (int32_t v = 0; v < V; ++v) {
    const uint32_t qx = static_cast<uint32_t>((g->pos[2 * v] - minx) * sx);
    const uint32_t qy = static_cast<uint32_t>((g->pos[2 * v + 1] - miny) * sy);
    const uint32_t m = morton_spread16(qx) | (morton_spread16(qy) << 1);
    key[v] = (static_cast<uint64_t>(static_cast<uint32_t>(find(v))) << 32) | m;
  }
  PROF_T(1);
  // stable LSD radix sort of the vertices by key (6 passes of 11 bits, constant digits skipped): O(V)
  std::vector<int32_t> order(V), tmp(V);
  std::iota(order.begin(), order.end(), 0);
  {
    constexpr int kBits = 11, kBuckets = 1 << kBits;
    std::vector<uint32_t> count(kBuckets + 1);
    for (int pass = 0; pass * kBits < 64; ++pass) {
      const int shift = kBits * pass;
      bool trivial = true;  // skip a pass whose digit is the same for all keys
      const uint32_t d0 = V ? static_cast<uint32_t>((key[0] >> shift) & (kBuckets - 1)) : 0u;
      for (int32_t v = 0; v < V && trivial; ++v) trivial = static_cast<uint32_t>((key[v] >> shift) & (kBuckets - 1)) == d0;
      if (trivial) continue;
      std::fill(count.begin(), count.end(), 0u);
      for (int32_t i = 0; i < V; ++i) count[((key[order[i]] >> shift) & (kBuckets - 1)) + 1]++;
      for (int d = 0; d < kBuckets; ++d) count[d + 1] += count[d];
      for (int32_t i = 0; i < V; ++i) tmp[count[(key[order[i]] >> shift) & (kBuckets - 1)]++] = order[i];
      order.swap(tmp);
    }
  }
  // The persistent layouts (D)/(E) walk the vertices in this pure Morton order: a wave then holds a compact
  // patch, so its graph neighbours sit in few other waves (fewer producers to wait for, better L2 locality).
  // Only the SELL-64 slices of the per-step sweep want the degree-sorted order below (uniform slice widths).
  L->order_m = order;
  const std::vector<int32_t>& order_m = L->order_m;
  L->rid_of.assign(static_cast<size_t>(V), 0);
  for (int32_t i = 0; i < V; ++i) L->rid_of[static_cast<size_t>(order_m[i])] = i;
  auto degree = [&](int32_t v) { return L->row_ptr[v + 1] - L->row_ptr[v]; };
  // stable counting sort by descending degree inside windows that never straddle two components
  // (frames of a batch)
  {
    std::vector<int32_t> bucket(static_cast<size_t>(maxdeg) + 2);
    for (int32_t c0 = 0; c0 < V;) {
      int32_t c1 = c0 + 1;
      while (c1 < V && (key[order[c1]] >> 32) == (key[order[c0]] >> 32)) ++c1;
      for (int32_t w0 = c0; w0 < c1;) {
        const int32_t w1 = std::min<int32_t>(c1, (w0 / kDegreeWindow + 1) * kDegreeWindow);
        std::fill(bucket.begin(), bucket.end(), 0);
        for (int32_t i = w0; i < w1; ++i) bucket[maxdeg - degree(order[i]) + 1]++;  // slot 0 = highest degree
        for (int32_t d = 0; d <= maxdeg; ++d) bucket[d + 1] += bucket[d];
        for (int32_t i = w0; i < w1; ++i) tmp[w0 + bucket[maxdeg - degree(order[i])]++] = order[i];
        std::copy(tmp.begin() + w0, tmp.begin() + w1, order.begin() + w0);
        w0 = w1;
      }
      c0 = c1;
    }
  }

  PROF_T(2);
  L->comp_start.clear();
  for (int32_t s = 0; s < V; ++s)
    if (s == 0 || (key[order[s]] >> 32) != (key[order[s - 1]] >> 32)) L->comp_start.push_back(s);
  L->comp_start.push_back(V);
  // ---- (B) SELL-64 ----------------------------------------------------------------------------
  const int32_t n_slices = (V + kWave - 1) / kWave;
  L->n_slices = n_slices;
  const size_t n_packed = static_cast<size_t>(n_slices) * kWave;
  L->perm.assign(n_packed, -1);
  L->pdeg.assign(n_packed, 0);
  L->iperm.assign(V, -1);
  for (int32_t s = 0; s < V; ++s) {
    L->perm[s] = order[s];
    L->iperm[order[s]] = s;
    L->pdeg[s] = degree(order[s]);
  }
  L->slice_row.assign(static_cast<size_t>(n_slices) + 1, 0);
  for (int32_t s = 0; s < n_slices; ++s) {
    int32_t width = 0;
    for (int l = 0; l < kWave; ++l) width = std::max(width, L->pdeg[static_cast<size_t>(s) * kWave + l]);
    L->slice_row[s + 1] = L->slice_row[s] + width;
  }
  L->rows = L->slice_row[n_slices];
  const size_t n_slots = static_cast<size_t>(L->rows + kRowPad) * kWave;
  L->rec_edge.clear(), L->rec_nbr.clear(), L->edge_src_slot.clear();
  if (host_expand) {  // (else: k_build_sell, nltgv2_layout.hip)
  L->rec_edge.assign(n_slots, -1);
  L->rec_nbr.assign(n_slots, 0);
  // empty slots point at a harmless, in-range vertex: the lane's own packed index where there is
  // one, else 0 (spare rows).
  for (int32_t s = 0; s < n_slices; ++s) {
    for (int64_t r = L->slice_row[s]; r < L->slice_row[s + 1]; ++r) {
      for (int l = 0; l < kWave; ++l) {
        L->rec_nbr[static_cast<size_t>(r) * kWave + l] = static_cast<uint32_t>(s * kWave + l);
      }
    }
  }
  L->edge_src_slot.assign(E, -1);
  for (size_t p = 0; p < n_packed; ++p) {
    const int32_t o = L->perm[p];
    if (o < 0) continue;
    const int32_t s = static_cast<int32_t>(p / kWave), l = static_cast<int32_t>(p % kWave);
    const int64_t row0 = L->slice_row[s];
    for (int32_t k = 0; k < degree(o); ++k) {
      const uint32_t h = L->half[L->row_ptr[o] + k];
      const int32_t e = static_cast<int32_t>(h & ~kRoleBit);
      const bool is_target = (h & kRoleBit) != 0;
      const int32_t other = is_target ? g->src[e] : g->dst[e];
      const size_t slot = static_cast<size_t>(row0 + k) * kWave + l;
      L->rec_edge[slot] = e;
      L->rec_nbr[slot] = static_cast<uint32_t>(L->iperm[other]) | (is_target ? kRoleBit : 0u);
      if (!is_target) L->edge_src_slot[e] = static_cast<int32_t>(slot);
    }
  }
  }

  PROF_T(3);
  build_patch_rows(g, L, order_m, host_expand, rowpack, rowpack_max_patches);
  PROF_T(4);
  PROF_T(5);
  if (host_expand) build_tv_rows(L);
  PROF_T(6);
  return FLAME_NLTGV2_OK;
}

}  // namespace flame_hip
