// nltgv2_layout.hip -- the slot- and lane-level layout arrays of nltgv2_pack.hpp, expanded ON THE DEVICE.
//
// A topology change (upload_graph, sync_graph: every frame of the real pipeline, flame.cc:1985-2121) used to cost
// ~1 ms of host packing and ~35 host-to-device copies.  What genuinely needs the host is per-VERTEX: the CSR of
// incident half-edges in ascending edge id (A), the (component, Morton) order, the slice table of (B) and the greedy
// patch walk of (E) -- a few tens of KB.  Everything per SLOT and per LANE (rec_nbr, rec_edge, edge_src_slot of (B);
// wg_slot, wg_vid, wg_meta, wg_nbr, wg_fetch of (E): ~2 MB at 640x480) follows from those and is produced here, by the
// same rules as the host builders in nltgv2_pack.hpp (which remain the reference: flame_nltgv2_layout_selftest and
// tests/test_pack.py compare the two bit for bit).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nltgv2_kernels.h"

namespace flame_hip {

namespace {

constexpr uint32_t kRole = 0x80000000u;
constexpr uint32_t kWgTail = 1u << 24, kWgActive = 1u << 25, kWgValid = 1u << 26, kWgPublish = 1u << 27, kWgHead = 1u << 28;

// (B) body: one thread per packed vertex slot p = slice*64 + lane (padding lanes included): its column of the slice's
// rows.  Row k of a real vertex = its k-th incident half-edge (ascending edge id, from the CSR); every other slot of the
// slice is "empty": rec_edge -1, rec_nbr = the lane's own packed index (a harmless, in-range gather target).
__global__ void __launch_bounds__(256)
k_build_sell(const int n_packed, const int32_t* __restrict__ perm, const int32_t* __restrict__ slice_row,
             const int32_t* __restrict__ row_ptr, const uint32_t* __restrict__ half, const int32_t* __restrict__ src,
             const int32_t* __restrict__ dst, const int32_t* __restrict__ iperm, uint32_t* __restrict__ rec_nbr,
             int32_t* __restrict__ rec_edge, int32_t* __restrict__ edge_src_slot) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_packed) return;
  const int s = p >> 6, l = p & 63;
  const int row0 = slice_row[s], row1 = slice_row[s + 1];
  const int o = perm[p];
  int base = 0, deg = 0;
  if (o >= 0) {
    base = row_ptr[o];
    deg = row_ptr[o + 1] - base;
  }
  for (int r = row0; r < row1; ++r) {
    const int k = r - row0;
    const size_t slot = (size_t)r * 64 + l;
    if (k < deg) {
      const uint32_t h = half[base + k];
      const int e = (int)(h & ~kRole);
      const bool is_target = (h & kRole) != 0u;
      const int other = is_target ? src[e] : dst[e];
      rec_edge[slot] = e;
      rec_nbr[slot] = (uint32_t)iperm[other] | (is_target ? kRole : 0u);
      if (!is_target) edge_src_slot[e] = (int)slot;
    } else {
      rec_edge[slot] = -1;
      rec_nbr[slot] = (uint32_t)p;
    }
  }
}

// (E) lanes: one wave per patch.  Host input per patch: wg_info[4p] = its first record id, wg_info[4p+2] = its vertex count
// (| kWgSlab), wg_v0[p] = position in the walk of its first vertex (= that record id); order_m = the walk (caller's vertex
// ids); rid_tab = rid_of, the inverse of the walk: the record of a vertex.
// Output: the 64 lanes of the patch (wg_slot, wg_vid, wg_meta, wg_nbr), its fetch list (wg_fetch: the DISTINCT records
// of other patches it reads, ascending) and wg_info[4p+1] = their number.
__global__ void __launch_bounds__(64)
k_build_patch(const int n_patches, int32_t* __restrict__ wg_info, const int32_t* __restrict__ wg_v0, const int32_t* __restrict__ order_m,
              const int32_t* __restrict__ rid_tab, const uint8_t* __restrict__ vfirst, const int V,
              const int32_t* __restrict__ iperm,
              const int32_t* __restrict__ slice_row,
              const int32_t* __restrict__ row_ptr, const uint32_t* __restrict__ half, const int32_t* __restrict__ src,
              const int32_t* __restrict__ dst, int32_t* __restrict__ wg_slot, int32_t* __restrict__ wg_vid,
              uint32_t* __restrict__ wg_meta, int32_t* __restrict__ wg_nbr, int32_t* __restrict__ wg_fetch) {
  __shared__ int s_first[64], s_o[64], s_vtx_of_lane[64], s_pub[64], s_list[64];
  const int p = blockIdx.x;
  if (p >= n_patches) return;
  const int lane = threadIdx.x;
  const int r0 = wg_info[4 * p], n_local = wg_info[4 * p + 2] & 0xffff, v0 = wg_v0[p];
  const int32_t* __restrict__ rid_of = rid_tab;
  // vertex j of the patch -> its first lane (an isolated vertex still owns one lane)
  int o = -1, deg = 0, need = 0;
  if (lane < n_local) {
    o = order_m[v0 + lane];
    deg = row_ptr[o + 1] - row_ptr[o];
    need = deg > 1 ? deg : 1;
  }
  // (the host's walk places a vertex's lanes: one after the other, or -- row-packed patches -- never across a 16-lane row)
  const int first = lane < n_local ? (int)vfirst[v0 + lane] : 0;
  s_vtx_of_lane[lane] = -1;
  s_pub[lane] = 0;
  __syncthreads();
  if (lane < n_local) {
    s_first[lane] = first, s_o[lane] = o;
    for (int k = 0; k < need; ++k) s_vtx_of_lane[first + k] = lane;
  }
  __syncthreads();
  // lane t of the patch
  const int j = s_vtx_of_lane[lane];
  int slot = -1, vid = -1, nbr = 0, cand = 0x7fffffff;
  uint32_t meta = 0u;
  if (j >= 0) {
    const int oj = s_o[j], fj = s_first[j];
    const int dj = row_ptr[oj + 1] - row_ptr[oj], needj = dj > 1 ? dj : 1;
    const int k = lane - fj;
    const int sp = iperm[oj];
    vid = sp;
    meta = (uint32_t)fj | ((uint32_t)dj << 6) | ((uint32_t)j << 13) | kWgValid;
    if (k == needj - 1) meta |= kWgTail;
    if (k == 0) meta |= kWgHead;
    if (k < dj) {
      meta |= kWgActive;
      slot = (slice_row[sp >> 6] + k) * 64 + (sp & 63);
      const uint32_t h = half[row_ptr[oj] + k];
      const int e = (int)(h & ~kRole);
      const int other = (h & kRole) ? src[e] : dst[e];
      const int r = rid_of[other];
      if (r >= r0 && r < r0 + n_local) {
        nbr = r - r0;
      } else {
        cand = r;          // a record of another patch
        s_pub[j] = 1;      // ... and that patch reads this vertex (the graph is undirected)
      }
    }
  }
  // the DISTINCT foreign records, ascending: bitonic sort of the 64 candidates, drop repeats, compact
  int v = cand;
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int d = k >> 1; d > 0; d >>= 1) {
      const int u = __shfl_xor(v, d, 64);
      const bool up = (lane & k) == 0, lower = (lane & d) == 0;
      v = (lower == up) ? (v < u ? v : u) : (v > u ? v : u);
    }
  }
  const int prev = __shfl_up(v, 1, 64);
  const bool keep = v != 0x7fffffff && (lane == 0 || v != prev);
  const unsigned long long km = __ballot(keep);
  const int rank = __popcll(km & ((1ull << lane) - 1ull));
  const int n_fetch = __popcll(km);
  if (keep) s_list[rank] = v;
  __syncthreads();
  const size_t hl = (size_t)p * 64 + lane;
  wg_fetch[hl] = lane < n_fetch ? s_list[lane] : -1;
  if (cand != 0x7fffffff) {  // index of this lane's record in the fetch list
    int lo = 0, hi = n_fetch - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_list[mid] < cand) lo = mid + 1; else hi = mid;
    }
    nbr = (int)(0x80000000u | (unsigned)lo);
  }
  if (j >= 0 && s_pub[j]) meta |= kWgPublish;
  wg_slot[hl] = slot, wg_vid[hl] = vid, wg_meta[hl] = meta, wg_nbr[hl] = nbr;
  if (lane == 0) wg_info[4 * p + 1] = n_fetch;
}

// (E2) lanes, two half-edges per lane (k_persistent_pv2): one wave per patch, as k_build_patch.  Host input per patch:
// wg_info[4p] = its first record id (= the walk position of its first vertex), wg_info[4p+2] = its vertex count; vfirst = the
// first lane of every vertex of the walk.  A vertex of d edges owns max(1, ceil(d / 2)) lanes; lane first + j holds its half-edges
// 2j (slot 0) and 2j + 1 (slot 1).  Output: wg_slot / wg_nbr [patch][slot][lane], wg_vid, wg_meta, the fetch list (the DISTINCT
// records of other patches, ascending) and wg_info[4p+1] = their number -- more than 64 cannot be fetched by 64 lanes: the
// number is left there (the kernel refuses such a patch) and *rmax tells the host.
__global__ void __launch_bounds__(64)
k_build_patch2(const int n_patches, int32_t* __restrict__ wg_info, const int32_t* __restrict__ order_m, const int32_t* __restrict__ rid_of,
               const uint8_t* __restrict__ vfirst, const int32_t* __restrict__ iperm, const int32_t* __restrict__ slice_row,
               const int32_t* __restrict__ row_ptr, const uint32_t* __restrict__ half, const int32_t* __restrict__ src,
               const int32_t* __restrict__ dst, int32_t* __restrict__ wg_slot, int32_t* __restrict__ wg_vid, uint32_t* __restrict__ wg_meta,
               int32_t* __restrict__ wg_nbr, int32_t* __restrict__ wg_fetch, int* __restrict__ rmax) {
  __shared__ int s_first[64], s_o[64], s_vtx_of_lane[64], s_pub[64], s_list[64];
  const int p = blockIdx.x;
  if (p >= n_patches) return;
  const int lane = threadIdx.x;
  const int r0 = wg_info[4 * p], n_local = wg_info[4 * p + 2] & 0xffff, v0 = r0;
  int o = -1, need = 0;
  if (lane < n_local) {
    o = order_m[v0 + lane];
    const int deg = row_ptr[o + 1] - row_ptr[o];
    need = deg > 1 ? (deg + 1) >> 1 : 1;
  }
  const int first = lane < n_local ? (int)vfirst[v0 + lane] : 0;
  s_vtx_of_lane[lane] = -1;
  s_pub[lane] = 0;
  __syncthreads();
  if (lane < n_local) {
    s_first[lane] = first, s_o[lane] = o;
    for (int k = 0; k < need; ++k) s_vtx_of_lane[first + k] = lane;
  }
  __syncthreads();
  const int j = s_vtx_of_lane[lane];
  int slot[2] = {-1, -1}, nbr[2] = {0, 0}, cand[2] = {0x7fffffff, 0x7fffffff};
  int vid = -1;
  uint32_t meta = 0u;
  if (j >= 0) {
    const int oj = s_o[j], fj = s_first[j];
    const int dj = row_ptr[oj + 1] - row_ptr[oj], needj = dj > 1 ? (dj + 1) >> 1 : 1;
    const int jj = lane - fj;
    const int sp = iperm[oj];
    vid = sp;
    meta = (uint32_t)fj | ((uint32_t)needj << 6) | ((uint32_t)j << 13) | kWgValid;
    if (jj == needj - 1) meta |= kWgTail;
    if (jj == 0) meta |= kWgHead;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int k = 2 * jj + sl;
      if (k < dj) {
        if (sl == 0) meta |= kWgActive;
        slot[sl] = (slice_row[sp >> 6] + k) * 64 + (sp & 63);
        const uint32_t h = half[row_ptr[oj] + k];
        const int e = (int)(h & ~kRole);
        const int other = (h & kRole) ? src[e] : dst[e];
        const int r = rid_of[other];
        if (r >= r0 && r < r0 + n_local) {
          nbr[sl] = r - r0;
        } else {
          cand[sl] = r;
          s_pub[j] = 1;
        }
      }
    }
  }
  // the DISTINCT foreign records, ascending: bitonic sort of the 128 candidates (element lane of slot 0, element 64 + lane of
  // slot 1), drop repeats, compact
  int a = cand[0], b = cand[1];
#pragma unroll
  for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
    for (int d = k >> 1; d > 0; d >>= 1) {
      if (d == 64) {
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        a = lo, b = hi;
      } else {
        const int ua = __shfl_xor(a, d, 64), ub = __shfl_xor(b, d, 64);
        const bool lower = (lane & d) == 0;
        const bool up_a = (lane & k) == 0, up_b = ((lane + 64) & k) == 0;
        a = (lower == up_a) ? (a < ua ? a : ua) : (a > ua ? a : ua);
        b = (lower == up_b) ? (b < ub ? b : ub) : (b > ub ? b : ub);
      }
    }
  }
  const int prev_a = __shfl_up(a, 1, 64), a_last = __shfl(a, 63, 64), prev_b0 = __shfl_up(b, 1, 64);
  const int prev_b = lane == 0 ? a_last : prev_b0;
  const bool keep_a = a != 0x7fffffff && (lane == 0 || a != prev_a);
  const bool keep_b = b != 0x7fffffff && b != prev_b;
  const unsigned long long ka = __ballot(keep_a), kb = __ballot(keep_b);
  const unsigned long long below = (1ull << lane) - 1ull;
  const int rank_a = __popcll(ka & below), rank_b = __popcll(ka) + __popcll(kb & below);
  const int n_fetch = __popcll(ka) + __popcll(kb);
  if (keep_a && rank_a < 64) s_list[rank_a] = a;
  if (keep_b && rank_b < 64) s_list[rank_b] = b;
  __syncthreads();
  const int n_list = n_fetch < 64 ? n_fetch : 64;
  const size_t hl = (size_t)p * 64 + lane, hl2 = (size_t)p * 128 + lane;
  wg_fetch[hl] = lane < n_list ? s_list[lane] : -1;
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    if (cand[sl] != 0x7fffffff) {
      int lo = 0, hi = n_list - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s_list[mid] < cand[sl]) lo = mid + 1; else hi = mid;
      }
      nbr[sl] = (int)(0x80000000u | (unsigned)lo);
    }
    wg_slot[hl2 + (size_t)sl * 64] = slot[sl];
    wg_nbr[hl2 + (size_t)sl * 64] = nbr[sl];
  }
  if (j >= 0 && s_pub[j]) meta |= kWgPublish;
  wg_vid[hl] = vid, wg_meta[hl] = meta;
  if (lane == 0) {
    wg_info[4 * p + 1] = n_fetch;
    if (n_fetch > 64) atomicMax(rmax, n_fetch);
  }
}

// ---- record placement -------------------------------------------------------------------------------------------------
// The hand-off of a 16-byte record from one XCD to another goes through the memory channel the record's address belongs
// to, and how long that takes depends on where that channel sits relative to the two XCDs: measured on MI355X
// (tools/hop_bench, HOP_BENCH_ADDR_SCAN) 0.39-0.46 us on one half of the 4 KB pages and 0.59-0.66 us on the other half for
// two XCDs of the same half of the package, 0.48-0.58 us for two XCDs of different halves -- the pattern repeats every
// four pages, with a phase that differs from allocation to allocation.  k_persistent_pv's period is the slowest hand-off
// plus a step's arithmetic, so the records other XCDs read are worth placing:
//   * k_place_calibrate measures, once per context, every page of a small pool (2 step parities x kPlacePages) for all 56
//     ordered pairs of XCDs at once: 56 single-wave blocks ping-pong a record per page; the host ranks the pages per pair
//     and direction;
//   * k_place_assign gives, per topology, every record that a patch on another XCD reads a slot on the best page of its
//     (producer XCD, consumer XCD) pair that still has room -- a patch's records of one pair stay contiguous, as they are
//     in the linear buffer -- and leaves -1 for the others (they keep their linear place).
// Only addresses change: which XCD a patch really runs on is still found out at run time (the XCC table), a wrong guess
// here costs time, not correctness.
__device__ __forceinline__ unsigned place_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

// Block b = 8 j + x (x = its XCD by dispatch order, j = 1..7) and block 8 (8 - j) + (x + j) % 8 are partners; the lower
// one starts.  Per page they exchange `iters` round trips on a 64-byte line of their own.  The two directions of a pair
// differ (the reader's side of the path is travelled more often than the writer's: 0.41-0.47 us on a page near the reader,
// 0.57-0.61 us on one near the writer, for XCDs of different halves), so each record carries the device-wide 100 MHz clock
// of its store and the receiver adds up (its clock when it sees the tag) - (the stamp):
// out[b][page] = ticks of `iters` hand-offs partner -> b, xcc_out[b] = the XCD the block really ran on.
__global__ void __launch_bounds__(64)
k_place_calibrate(char* pool, const int n_pages, const int iters, unsigned* __restrict__ out, int* __restrict__ xcc_out,
                  int* fail) {
  const int b = blockIdx.x, x = b & 7, j = b >> 3;
  if (threadIdx.x == 0) xcc_out[b] = (int)place_xcc_id();
  if (j == 0) return;
  const int pb = ((8 - j) & 7) * 8 + ((x + j) & 7);
  const bool first = b < pb;
  const int line = (first ? b : pb) * 64;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(pool, 0, 0x7fffffff, 0x00020000);
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  int tag = 0;
  bool dead = false;
  unsigned stamp = 0;
  auto wait_tag = [&](const int off, const int want) {
    for (unsigned spin = 0; spin < (1u << 17); ++spin) {
      int o = off;
      asm volatile("" : "+v"(o)::"memory");
      const v4i_t g = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 16);
      if (g.w == want) {
        stamp = (unsigned)g.x;
        return true;
      }
      if ((spin & 1023u) == 1023u && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    }
    return false;
  };
  for (int pg = 0; pg < n_pages && !dead; ++pg) {
    const int mine = pg * 4096 + line + (first ? 0 : 16), theirs = pg * 4096 + line + (first ? 16 : 0);
    unsigned sum = 0;
    for (int it = 0; it < iters + 2; ++it) {
      ++tag;
      if (!first) {
        if (!wait_tag(theirs, tag)) {
          dead = true;
          break;
        }
        if (it >= 2) sum += (unsigned)wall_clock64() - stamp;
      }
      const v4i_t rec = {(int)(unsigned)wall_clock64(), 0, 0, tag};
      __builtin_amdgcn_raw_buffer_store_b128(rec, r, mine, 0, 16);
      if (first) {
        if (!wait_tag(theirs, tag)) {
          dead = true;
          break;
        }
        if (it >= 2) sum += (unsigned)wall_clock64() - stamp;
      }
    }
    if (threadIdx.x == 0) out[(size_t)b * n_pages + pg] = dead ? 0u : sum;
  }
  if (dead && threadIdx.x == 0) __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Step 1, one thread per patch: which patch a record belongs to; every record starts unplaced; the page counters start at 0.
__global__ void __launch_bounds__(256)
k_place_patch_of_record(const int n_patches, const int32_t* __restrict__ wg_info, int32_t* __restrict__ patch_of_rec,
                        int32_t* __restrict__ rec_off, const int stride, int* __restrict__ fill, const int n_fill) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_fill) fill[p] = 0;
  if (p < 128) fill[n_fill + 16 + p] = 0;  // the cursors of k_place_assign (behind the page counters and the rotation word's line)
  if (p >= n_patches) return;
  const int r0 = wg_info[4 * p], n = wg_info[4 * p + 2] & 0xffff;
  for (int i = 0; i < n; ++i) patch_of_rec[r0 + i] = p, rec_off[r0 + i] = -1, rec_off[stride + r0 + i] = -1;
}

// Step 2, one thread per record (= position in the walk): its class -- 8 a + b for the first neighbour found on another XCD
// b, -1 if every reader is on the record's own XCD a.  (Per record, not per patch: a patch's ~70 neighbour look-ups are
// two dependent loads each; in one thread they took 170 us.)
__global__ void __launch_bounds__(256)
k_place_classify(const int V, const int per_xcd, const int32_t* __restrict__ order_m, const int32_t* __restrict__ rid_of,
                 const int32_t* __restrict__ row_ptr, const uint32_t* __restrict__ half, const int32_t* __restrict__ src,
                 const int32_t* __restrict__ dst, const int32_t* __restrict__ patch_of_rec, int8_t* __restrict__ cls) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= V) return;
  const int o = order_m[r];
  const int a = patch_of_rec[r] / per_xcd;
  int c = -1;
  for (int h = row_ptr[o]; h < row_ptr[o + 1] && c < 0; ++h) {
    const uint32_t hh = half[h];
    const int e = (int)(hh & ~kRole);
    const int other = (hh & kRole) ? src[e] : dst[e];
    const int xb = patch_of_rec[rid_of[other]] / per_xcd;
    if (xb != a) c = a * 8 + xb;
  }
  cls[r] = (int8_t)c;
}

// Step 3, one WAVE per patch, lane i = its i-th record (<= 64, ~10): per parity and class a run of slots on the first page of the
// class's ranking that has room (pages hold 256 records; the counters may overshoot, a page that refused a run simply stays a little
// emptier).  (Round 4: one THREAD per patch kept the classes in a register array indexed at run time -- 118 us per topology at
// 640x480, in front of the frame's first run; the lanes hold them now and a ballot finds the members of a class.)
__global__ void __launch_bounds__(64)
k_place_assign(const int n_patches, const int32_t* __restrict__ wg_info, const int8_t* __restrict__ cls_of_rec,
               const uint16_t* __restrict__ ranking, const int n_pages, int* __restrict__ fill, int32_t* __restrict__ rec_off,
               const int stride) {
  const int p = blockIdx.x, lane = threadIdx.x;
  if (p >= n_patches) return;
  const int r0 = wg_info[4 * p], n = min(wg_info[4 * p + 2] & 0xffff, 64);
  const int c_mine = lane < n ? (int)cls_of_rec[r0 + lane] : -1;
  unsigned long long todo = __ballot(c_mine >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int c = __shfl(c_mine, leader, 64);
    const unsigned long long members = __ballot(c_mine == c);
    todo &= ~members;
    const int m = __popcll(members);
    // (a run starts on a 128-byte line of its own: the records of one patch are written by one instruction; lines shared
    //  with another patch's records would be invalidated under the reader at that patch's pace as well)
    const int m8 = (m + 7) & ~7;
    for (int par = 0; par < 2; ++par) {
      int base = -1;
      if (lane == leader) {
        // every patch of a class wants the same best page: the walk through the ranking starts at the first page that has not
        // yet refused a run of this class (a cursor per parity and class)
        const uint16_t* const rk = ranking + ((size_t)par * 64 + c) * n_pages;
        int* const cursor = fill + 2 * n_pages + 16 + par * 64 + c;
        for (int t = __hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); t < n_pages && base < 0; ++t) {
          const int pg = rk[t];
          const int s0 = atomicAdd(&fill[par * n_pages + pg], m8);
          if (s0 + m8 <= 256) base = (par * n_pages + pg) * 4096 + s0 * 16;
          else atomicMax(cursor, t + 1);
        }
      }
      base = __shfl(base, leader, 64);
      if (c_mine == c && base >= 0)
        rec_off[(size_t)par * stride + r0 + lane] = base + 16 * __popcll(members & ((1ull << lane) - 1ull));
    }
  }
}

inline dim3 grid1d(long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace


int launch_build_sell(const CanonArgs& c, const FusedArgs& a, const int32_t* iperm, hipStream_t s) {
  const int n_packed = a.n_slices * 64;
  if (n_packed <= 0) return 0;
  hipLaunchKernelGGL(k_build_sell, grid1d(n_packed), dim3(256), 0, s, n_packed, a.perm, a.slice_row, c.row_ptr, c.half,
                     c.src, c.dst, iperm, a.rec_nbr, a.rec_edge, a.edge_src_slot);
  return (int)hipGetLastError();
}

int launch_place_calibrate(char* pool, int n_pages, int iters, unsigned* out, int* xcc_out, int* fail, hipStream_t s) {
  hipLaunchKernelGGL(k_place_calibrate, dim3(64), dim3(64), 0, s, pool, n_pages, iters, out, xcc_out, fail);
  return (int)hipGetLastError();
}

int launch_place_records(const CanonArgs& c, const FusedArgs& a, int per_xcd, const int32_t* order_m, const int32_t* rid_of,
                         int32_t* patch_of_rec, int8_t* cls, const uint16_t* ranking, int n_pages, int* fill, int32_t* rec_off,
                         int stride, hipStream_t s) {
  if (a.wg_count <= 0 || per_xcd <= 0) return 0;
  const int n0 = a.wg_count > 2 * n_pages ? a.wg_count : 2 * n_pages;
  hipLaunchKernelGGL(k_place_patch_of_record, grid1d(n0), dim3(256), 0, s, a.wg_count, a.wg_info, patch_of_rec, rec_off, stride, fill,
                     2 * n_pages);
  hipLaunchKernelGGL(k_place_classify, grid1d(c.V), dim3(256), 0, s, c.V, per_xcd, order_m, rid_of, c.row_ptr, c.half, c.src, c.dst,
                     patch_of_rec, cls);
  hipLaunchKernelGGL(k_place_assign, dim3((unsigned)a.wg_count), dim3(64), 0, s, a.wg_count, a.wg_info, cls, ranking, n_pages, fill, rec_off,
                     stride);
  return (int)hipGetLastError();
}

int launch_build_patches2(const CanonArgs& c, const FusedArgs& a, int n_patches, int32_t* wg_info, const int32_t* order_m, const int32_t* rid_tab,
                          const uint8_t* vfirst, const int32_t* iperm, int32_t* wg_slot, int32_t* wg_vid, uint32_t* wg_meta, int32_t* wg_nbr,
                          int32_t* wg_fetch, int* rmax, hipStream_t s) {
  if (n_patches <= 0) return 0;
  hipLaunchKernelGGL(k_build_patch2, dim3((unsigned)n_patches), dim3(64), 0, s, n_patches, wg_info, order_m, rid_tab, vfirst, iperm,
                     a.slice_row, c.row_ptr, c.half, c.src, c.dst, wg_slot, wg_vid, wg_meta, wg_nbr, wg_fetch, rmax);
  return (int)hipGetLastError();
}

int launch_build_patches(const CanonArgs& c, const FusedArgs& a, const int32_t* wg_v0, const int32_t* order_m,
                         const int32_t* rid_tab, const uint8_t* vfirst, const int32_t* iperm, hipStream_t s) {
  if (a.wg_count <= 0) return 0;
  hipLaunchKernelGGL(k_build_patch, dim3((unsigned)a.wg_count), dim3(64), 0, s, a.wg_count, a.wg_info, wg_v0, order_m, rid_tab, vfirst,
                     c.V, iperm, a.slice_row, c.row_ptr, c.half, c.src, c.dst, a.wg_slot, a.wg_vid, a.wg_meta, a.wg_nbr,
                     a.wg_fetch);
  return (int)hipGetLastError();
}

}  // namespace flame_hip

// ---- one blob up, one kernel to distribute it -----------------------------------------------------------------------------
// An upload is ~20 arrays of 30-200 KB; as separate host-to-device copies they cost ~7 us each on the stream (measured:
// 18 copies = 131 us of a 250 us upload) and the clears ~4.6 us each.  They travel as ONE blob instead and this kernel
// copies every piece to its buffer (and does the clears): blockIdx.y = table entry.
namespace flame_hip {
namespace {
__global__ void __launch_bounds__(256) k_scatter(const ScatterTable t, const uint8_t* __restrict__ blob) {
  const ScatterEntry e = t.e[blockIdx.y];
  const size_t n16 = e.bytes >> 4;
  uint4* __restrict__ d16 = static_cast<uint4*>(e.dst);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e.src_off == kScatterFill) {
    const uint4 v = make_uint4(e.fill, e.fill, e.fill, e.fill);
    for (size_t i = i0; i < n16; i += stride) d16[i] = v;
    uint32_t* d4 = static_cast<uint32_t*>(e.dst);
    for (size_t i = (n16 << 2) + i0; i < (e.bytes >> 2); i += stride) d4[i] = e.fill;
  } else {
    const uint4* __restrict__ s16 = reinterpret_cast<const uint4*>(blob + e.src_off);
    for (size_t i = i0; i < n16; i += stride) d16[i] = s16[i];
    uint32_t* d4 = static_cast<uint32_t*>(e.dst);
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(blob + e.src_off);
    for (size_t i = (n16 << 2) + i0; i < (e.bytes >> 2); i += stride) d4[i] = s4[i];
    // (every array here is a whole number of 32-bit words, except the byte tail handled by the first thread)
    if (i0 == 0)
      for (size_t b = e.bytes & ~size_t(3); b < e.bytes; ++b) static_cast<uint8_t*>(e.dst)[b] = blob[e.src_off + b];
  }
}
}  // namespace

namespace {
__global__ void __launch_bounds__(256) k_copy_arrays(const CopyTable t) {
  const uint4* __restrict__ src = static_cast<const uint4*>(t.e[blockIdx.y].src);
  uint4* __restrict__ dst = static_cast<uint4*>(t.e[blockIdx.y].dst);
  const size_t n16 = t.e[blockIdx.y].bytes >> 4, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace

int launch_copy_arrays(const CopyTable& t, hipStream_t s) {
  if (t.n <= 0) return 0;
  size_t mx = 0;
  for (int i = 0; i < t.n; ++i) mx = t.e[i].bytes > mx ? t.e[i].bytes : mx;
  unsigned gx = (unsigned)((mx / 16 + 4 * 256 - 1) / (4 * 256));
  gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
  hipLaunchKernelGGL(k_copy_arrays, dim3(gx, (unsigned)t.n), dim3(256), 0, s, t);
  return (int)hipGetLastError();
}

int launch_scatter(const ScatterTable& t, const void* blob, hipStream_t s) {
  if (t.n <= 0) return 0;
  size_t mx = 0;
  for (int i = 0; i < t.n; ++i) mx = t.e[i].bytes > mx ? t.e[i].bytes : mx;
  unsigned gx = (unsigned)((mx / 16 + 4 * 256 - 1) / (4 * 256));  // ~4 vectors per thread of the largest entry
  gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
  hipLaunchKernelGGL(k_scatter, dim3(gx, (unsigned)t.n), dim3(256), 0, s, t, static_cast<const uint8_t*>(blob));
  return (int)hipGetLastError();
}
}  // namespace flame_hip

// ---- per-frame graph synchronisation on the device (flame_nltgv2_sync_graph) --------------------------------------------
// The host works out WHO survives (feature ids, edge identity and orientation: index maps only); the state moves here.
namespace flame_hip {
namespace {

// Vertices, flame.cc:1996-2014, 2035-2048, 2160-2162: a survivor keeps (x, w, x_bar, w_bar, x_prev, w_prev), with the
// sticky-obstacle reset of x; a new vertex starts at x = x_bar = x_prev = init (or its data term), w = 0.  need_nbr[v] = 1
// marks a new vertex whose init value is NaN and is to be replaced by the mean of its neighbours (k_sync_init_from_neighbours).
__global__ void __launch_bounds__(256)
k_sync_vertices(const int V, const int32_t* __restrict__ old_of_new, const float* __restrict__ data, const float* __restrict__ init_x,
                const float* __restrict__ init_map, const int map_rows, const int map_cols, const float2* __restrict__ pos,
                const float graph_scale, const int check_sticky, const float sticky_threshold, const int nbr_fallback, const float* __restrict__ ox,
                const float* __restrict__ ow1, const float* __restrict__ ow2, const float* __restrict__ oxb,
                const float* __restrict__ ow1b, const float* __restrict__ ow2b, const float* __restrict__ oxp,
                const float* __restrict__ ow1p, const float* __restrict__ ow2p, float* __restrict__ x, float* __restrict__ w1,
                float* __restrict__ w2, float* __restrict__ xb, float* __restrict__ w1b, float* __restrict__ w2b,
                float* __restrict__ xp, float* __restrict__ w1p, float* __restrict__ w2p, uint8_t* __restrict__ need_nbr) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const int o = old_of_new[v];
  uint8_t need = 0;
  if (o >= 0) {
    float xv = ox[o];
    if (check_sticky && (xv - data[v] > sticky_threshold)) xv = data[v];
    x[v] = xv, w1[v] = ow1[o], w2[v] = ow2[o];
    xb[v] = oxb[o], w1b[v] = ow1b[o], w2b[v] = ow2b[o];
    xp[v] = oxp[o], w1p[v] = ow1p[o], w2p[v] = ow2p[o];
  } else {
    float xi = init_x ? init_x[v] : data[v];
    if (init_map) {  // init_with_prediction, flame.cc:2131: idepthmap(pos.y + 0.5f, pos.x + 0.5f) -- float -> int truncates --, / graph_scale
      const float2 p = pos[v];
      const int iy = (int)(p.y + 0.5f), ix = (int)(p.x + 0.5f);
      xi = (iy >= 0 && iy < map_rows && ix >= 0 && ix < map_cols) ? init_map[(size_t)iy * map_cols + ix] / graph_scale : __builtin_nanf("");
    }
    if (nbr_fallback && xi != xi) {
      xi = data[v];  // what the vertex holds (flame.cc:2046-2048) while its neighbours' means are formed
      need = 1;
    }
    x[v] = xb[v] = xp[v] = xi;
    w1[v] = w2[v] = w1b[v] = w2b[v] = w1p[v] = w2p[v] = 0.0f;
  }
  need_nbr[v] = need;
}

// Edges, flame.cc:2085-2104: a surviving edge keeps its dual, a new one starts at q = 0; every edge gets
// alpha = 1/||pos_a - pos_b|| from the NEW positions and beta = 1.
__global__ void __launch_bounds__(256)
k_sync_edges(const int E, const int32_t* __restrict__ old_of_new_edge, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
             const float2* __restrict__ pos, const float* __restrict__ oq1, const float* __restrict__ oq2, const float* __restrict__ oq3,
             float* __restrict__ q1, float* __restrict__ q2, float* __restrict__ q3, float* __restrict__ alpha, float* __restrict__ beta) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int o = old_of_new_edge[e];
  q1[e] = o >= 0 ? oq1[o] : 0.0f;
  q2[e] = o >= 0 ? oq2[o] : 0.0f;
  q3[e] = o >= 0 ? oq3[o] : 0.0f;
  const float2 a = pos[src[e]], b = pos[dst[e]];
  const float dx = a.x - b.x, dy = a.y - b.y;
  alpha[e] = 1.0f / sqrtf(dx * dx + dy * dy);  // (sqrtf: correctly rounded in this build; __fsqrt_rn is the raw v_sqrt_f32)
  beta[e] = 1.0f;
}

// flame.cc:2133-2158 (init_with_prediction, no valid prediction): mean of x * graph_scale over the neighbours with
// data_weight > 0, in ascending edge id (the reference walks a hash set: its order is unspecified); none -> data term.
// The result goes to x_bar only; k_sync_init_commit copies it to x and x_prev -- so that every mean is formed from the
// same, settled neighbour values.
__global__ void __launch_bounds__(256)
k_sync_init_from_neighbours(const int V, const uint8_t* __restrict__ need_nbr, const int32_t* __restrict__ row_ptr,
                            const uint32_t* __restrict__ half, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                            const float* __restrict__ weight, const float* __restrict__ x, const float* __restrict__ data,
                            const float graph_scale, float* __restrict__ xb) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V || !need_nbr[v]) return;
  float sum = 0.0f;
  int count = 0;
  for (int h = row_ptr[v]; h < row_ptr[v + 1]; ++h) {
    const uint32_t he = half[h];
    const int e = (int)(he & ~kRole);
    const int nb = (he & kRole) ? src[e] : dst[e];
    if (weight[nb] > 0.0f) {
      sum += x[nb] * graph_scale;
      ++count;
    }
  }
  xb[v] = count > 0 ? (sum / (float)count) / graph_scale : data[v];
}
__global__ void __launch_bounds__(256)
k_sync_init_commit(const int V, const uint8_t* __restrict__ need_nbr, float* __restrict__ x, const float* __restrict__ xb,
                   float* __restrict__ xp) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V || !need_nbr[v]) return;
  x[v] = xp[v] = xb[v];
}

}  // namespace

int launch_sync_state(const SyncArgs& a, hipStream_t s) {
  if (a.V > 0) {
    hipLaunchKernelGGL(k_sync_vertices, grid1d(a.V), dim3(256), 0, s, a.V, a.old_of_new, a.data, a.init_x, a.init_map, a.map_rows, a.map_cols,
                       a.pos, a.graph_scale, a.check_sticky,
                       a.sticky_threshold, a.graph_scale > 0.0f ? 1 : 0, a.o[0], a.o[1], a.o[2], a.o[3], a.o[4], a.o[5], a.o[6], a.o[7],
                       a.o[8], a.n[0], a.n[1], a.n[2], a.n[3], a.n[4], a.n[5], a.n[6], a.n[7], a.n[8], a.need_nbr);
  }
  if (a.E > 0) {
    hipLaunchKernelGGL(k_sync_edges, grid1d(a.E), dim3(256), 0, s, a.E, a.old_of_new_edge, a.src, a.dst, a.pos, a.oq[0], a.oq[1],
                       a.oq[2], a.nq[0], a.nq[1], a.nq[2], a.alpha, a.beta);
  }
  if (a.V > 0 && a.graph_scale > 0.0f) {
    hipLaunchKernelGGL(k_sync_init_from_neighbours, grid1d(a.V), dim3(256), 0, s, a.V, a.need_nbr, a.row_ptr, a.half, a.src, a.dst,
                       a.weight, a.n[0], a.data, a.graph_scale, a.n[3]);
    hipLaunchKernelGGL(k_sync_init_commit, grid1d(a.V), dim3(256), 0, s, a.V, a.need_nbr, a.n[0], a.n[3], a.n[6]);
  }
  return (int)hipGetLastError();
}

// Loads this translation unit's code object (the runtime does that at the first use of one of its kernels: several milliseconds that
// flame_nltgv2_create takes on itself so that the first frame does not).
void warm_module_layout() {
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, (const void*)k_scatter) != hipSuccess) (void)hipGetLastError();
}

}  // namespace flame_hip
