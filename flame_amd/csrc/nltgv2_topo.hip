// nltgv2_topo.hip -- the per-VERTEX layout tables of nltgv2_pack.hpp, built ON THE DEVICE.
//
// Every frame of the real pipeline changes the graph (Flame::syncGraph, /root/reference/src/flame/flame.cc:1985-2121).  Until round 3
// the host worked out who survives (feature id -> previous vertex, triangulator edge -> previous edge and its orientation), rebuilt
// the per-vertex tables of the new graph (CSR in ascending edge id, connected components, the (component, Morton) walk, the SELL-64
// slice table, the greedy patch walks) and sent them up: 0.7 ms of a 640x480 frame and 4-5 ms of a 1080p frame in which the solver
// stood still -- ten times the 200 iterations they feed.  Here the same tables are produced by a dozen small kernels over the
// RESIDENT previous topology and one staged copy of the frame's inputs; nltgv2_pack.hpp stays the reference they are compared with
// word for word (flame_nltgv2_layout_selftest; tests/test_sync_graph.py: test_device_expanded_layout_matches_host_builders and the self-test after every frame).
//
//   front (sync)    k_topo_init      feature id -> previous vertex through a stamped table (one probe per vertex, the table follows
//                                    the graph in the same pass), Morton codes, union-find roots
//                   k_topo_edges     per triangulator edge: the previous edge between the same two features and its orientation, found
//                                    in the previous CSR (~6 incident edges); "the first of equal pairs keeps the old edge" = an atomic
//                                    minimum per old edge; degrees
//                   one exclusive scan over [survivor flags of the old edges | new-edge flags | degrees]: survivors keep their
//                                    previous relative order, new edges follow in triangulator order (what boost::edges() yields after
//                                    the reference's erase / add_edge sequence), and the same pass yields row_ptr
//                   k_topo_new_edges the new (src, dst) list -- a survivor keeps its old orientation, flame.cc:2094-2100 --, the map
//                                    new edge -> previous edge the state gather reads, the half-edges into their rows (atomic slots)
//   front (upload)  k_topo_init, k_topo_edges_given (degrees of a given edge list), the scan
//   back            k_topo_csr_fill  half-edges into their rows (atomic slots)
//                   k_topo_rows      a row sorted by edge id (the reference's scatter order, cc:120-142); every vertex under its
//                                    smallest neighbour (the first, atomic-free round of the connected components)
//                   k_topo_hook      the remaining trees joined over the edges by lock-free hooking (ECL-CC), by pseudo-random priority
//                   k_topo_roots     a vertex's root; a component's label = its smallest vertex id, as the host's union-find gives it
//                   radix sort       (component, Morton code), stable: the walk order_m
//                   k_topo_windows   one workgroup per 512 walk positions: rid_of, the stable sort by descending degree inside the
//                                    window (perm / iperm / pdeg of SELL-64), slice widths; the last workgroup scans them (slice_row)
//                   k_topo_walk      the greedy first-fit patch walks of (E) and (E2), one LANE per segment of 256 walk positions
//                                    (a segment starts a patch: nltgv2_pack.hpp kWalkSegment), degrees staged through LDS
//                   k_topo_patches   patch tables (first vertex, count, largest degree), first lane of every vertex
// The sort and the scan are rocPRIM's (AMD's own device primitives); everything else is written here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "nltgv2_kernels.h"

namespace flame_hip {

namespace {

constexpr uint32_t kRole = 0x80000000u;
constexpr int kWalkSegment = 256;     // == nltgv2_pack.hpp kWalkSegment
constexpr int kWindow = 512;          // == nltgv2_pack.hpp kDegreeWindow
constexpr int kWalkLanes = 64;        // segments per workgroup of k_topo_walk
constexpr int kWalkStride = 65;       // dwords per LDS row of k_topo_walk: the transposing accesses then touch 64 different banks
constexpr int kCcRounds = 3;          // synchronous hooking rounds of the connected components before the asynchronous last one

inline dim3 grid1d(long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t spread16(uint32_t v) {  // == morton_spread16
  v &= 0xFFFFu;
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

// Lock-free union-find in the manner of ECL-CC (Jaiganesh & Burtscher): parent[] only ever decreases, so a stale read still names
// an ancestor, and a root is hooked under a smaller root with a compare-and-swap.  "Smaller" is by a pseudo-random PRIORITY of the
// vertex, cc_mix(v) -- a bijection of [0, 2^bits) --, not by its id: callers number their vertices along the image rows, and by id
// every vertex's smallest neighbour is the one up and to the left -- chains of ~sqrt(V) dependent loads to the top-left corner (measured:
// 444 us of hooking at 1080p).  By priority the chains are O(log V).  The label the host's union-find gives a component, its smallest
// vertex id, is formed afterwards (k_topo_roots: one atomic minimum per wave and root).
__device__ __forceinline__ uint32_t cc_mix(uint32_t x, const int bits) {
  const uint32_t mask = (1u << bits) - 1u;
  const int h = (bits + 1) >> 1;
  x ^= x >> h;
  x = (x * 0x9E3779B1u) & mask;
  x ^= x >> h;
  x = (x * 0x85EBCA6Bu) & mask;
  x ^= x >> h;
  return x;
}
__device__ int cc_root(int* parent, const int v) {
  int cur = ld_agent(&parent[v]);
  if (cur != v) {
    int prev = v, next;
    while (cur > (next = ld_agent(&parent[cur]))) {
      st_agent(&parent[prev], next);  // path halving: any ancestor is a valid parent
      prev = cur;
      cur = next;
    }
  }
  return cur;
}
__device__ void cc_hook(int* parent, const int a, const int b) {
  int ra = cc_root(parent, a), rb = cc_root(parent, b);
  while (ra != rb) {
    if (ra < rb) {
      const int got = atomicCAS(&parent[rb], rb, ra);
      if (got == rb) break;
      rb = got;
    } else {
      const int got = atomicCAS(&parent[ra], ra, rb);
      if (got == ra) break;
      ra = got;
    }
  }
}

// Feature table: feature id -> vertex, as two open-addressing hash tables with generation stamps (TopoBuild).  The graph of
// generation g lives in table g & 1; a sync looks its ids up in the previous generation's table and enters them into the other one,
// whose slots all carry older stamps -- so nothing is cleared between frames and ids may grow without bound (the reference's feature
// ids grow by one per detection for the whole session).  Linear probing, at most a quarter full.
__device__ __forceinline__ uint32_t feat_slot(int id, int bits) { return ((uint32_t)id * 0x9E3779B1u) >> (32 - bits); }

__device__ __forceinline__ void feat_insert(uint32_t* stamp, int32_t* key, int32_t* val, int bits, uint32_t gen, int id, int v) {
  const uint32_t base = (gen & 1u) << bits, mask = (1u << bits) - 1u;
  for (uint32_t h = feat_slot(id, bits);; h = (h + 1u) & mask) {
    const uint32_t seen = __hip_atomic_load(&stamp[base + h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen != gen && atomicCAS(&stamp[base + h], seen, gen) == seen) {  // (a slot of an older generation: taken)
      key[base + h] = id, val[base + h] = v;
      return;
    }
  }
}

__device__ __forceinline__ int feat_lookup(const uint32_t* stamp, const int32_t* key, const int32_t* val, int bits, uint32_t gen, int id) {
  const uint32_t base = (gen & 1u) << bits, mask = (1u << bits) - 1u;
  for (uint32_t h = feat_slot(id, bits);; h = (h + 1u) & mask) {
    if (stamp[base + h] != gen) return -1;  // (the entries of a generation are never removed: the first free slot ends the probe)
    if (key[base + h] == id) return val[base + h];
  }
}

__global__ void __launch_bounds__(256)
k_topo_feat_build(const int32_t* __restrict__ feat, const int V, uint32_t* __restrict__ stamp, int32_t* __restrict__ key,
                  int32_t* __restrict__ val, const int tab_bits, const uint32_t gen) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  feat_insert(stamp, key, val, tab_bits, gen, feat[v], v);
}

__global__ void __launch_bounds__(256) k_topo_init(const TopoBuild t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    TopoDims d;
    memset(&d, 0, sizeof d);
    *t.dims = d;
    t.counters[0] = t.counters[1] = t.counters[2] = t.counters[3] = 0;
  }
  if (i < t.Eo) t.first_k[i] = kTopoInf;
  if (i < (1 << t.cc_bits)) t.minid[i] = 0x7fffffff;
  if (i >= t.V) return;
  if (t.fid) {
    const int id = t.fid[i];
    t.old_of_new[i] = feat_lookup(t.feat_stamp, t.feat_key, t.feat_val, t.tab_bits, t.gen_prev, id);
    feat_insert(t.feat_stamp, t.feat_key, t.feat_val, t.tab_bits, t.gen_new, id, i);
  }
  t.deg[i] = 0, t.cur[i] = 0;
  const float2 p = t.pos[i];
  const uint32_t qx = (uint32_t)((p.x - t.minx) * t.sx), qy = (uint32_t)((p.y - t.miny) * t.sy);
  t.morton[i] = spread16(qx) | (spread16(qy) << 1);
}

// flame.cc:2085-2100: boost::edge(u, v) finds the previous edge between the same two features whichever way it runs (the lowest edge id
// of parallel ones: the walk is in ascending edge id); of several triangulator edges joining the same pair the first keeps it.
__global__ void __launch_bounds__(256) k_topo_edges(const TopoBuild t) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= t.E) return;
  const int a = t.tri_edges[2 * k], b = t.tri_edges[2 * k + 1];
  const int oa = t.old_of_new[a], ob = t.old_of_new[b];
  int oe = -1;
  if (oa >= 0 && ob >= 0) {
    for (int h = t.o_row_ptr[oa]; h < t.o_row_ptr[oa + 1]; ++h) {
      const uint32_t hh = t.o_half[h];
      const int e = (int)(hh & ~kRole);
      const int nbr = (hh & kRole) ? t.o_src[e] : t.o_dst[e];
      if (nbr == ob) {
        oe = e | ((hh & kRole) ? (int)0x80000000u : 0);  // bit 31: the previous edge runs ob -> oa
        atomicMin(&t.first_k[e], k);
        break;
      }
    }
  }
  t.old_edge[k] = oe;
  atomicAdd(&t.deg[a], 1);
  atomicAdd(&t.deg[b], 1);
}

__global__ void __launch_bounds__(256) k_topo_edges_given(const TopoBuild t) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= t.E) return;
  const int a = t.src[e], b = t.dst[e];
  atomicAdd(&t.deg[a], 1);
  atomicAdd(&t.deg[b], 1);
  if (e == 0) t.dims->n_edges = t.E;
}

// input of the one scan: [old edge survives | triangulator edge is new | degree | 0]
struct ScanInput {
  const int32_t *first_k, *old_edge, *deg;
  int Eo, E, V;
  __host__ __device__ int operator()(int i) const {
    if (i < Eo) return first_k[i] != kTopoInf ? 1 : 0;
    i -= Eo;
    if (i < E) return old_edge[i] == -1 ? 1 : 0;
    i -= E;
    return i < V ? deg[i] : 0;
  }
};

__global__ void __launch_bounds__(256) k_topo_new_edges(const TopoBuild t) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= t.E) return;
  const int n_edges = t.scan[t.Eo + t.E];
  if (k == 0) {
    t.dims->n_edges = n_edges, t.dims->n_keep = t.scan[t.Eo];
    if (n_edges != t.E) atomicOr(&t.dims->flags, kTopoBadEdges);
  }
  const int a = t.tri_edges[2 * k], b = t.tri_edges[2 * k + 1];
  const int oe = t.old_edge[k];
  int e_new, s = a, d = b, om = -1;
  if (oe != -1) {
    const int e = oe & 0x7fffffff;
    if (t.first_k[e] != k) return;  // an earlier triangulator edge joins the same pair: nothing is added for this one
    e_new = t.scan[e];
    if (oe < 0) s = b, d = a;       // the previous edge runs old(b) -> old(a): it keeps that orientation
    om = e;
  } else {
    e_new = t.scan[t.Eo + k];
  }
  t.src[e_new] = s, t.dst[e_new] = d, t.old_of_new_edge[e_new] = om;
  // ... and its two half-edges into their rows (sorted by edge id in k_topo_rows)
  const int off = t.Eo + t.E;
  t.half[t.scan[off + s] - n_edges + atomicAdd(&t.cur[s], 1)] = (uint32_t)e_new;
  t.half[t.scan[off + d] - n_edges + atomicAdd(&t.cur[d], 1)] = (uint32_t)e_new | kRole;
}

__global__ void __launch_bounds__(256) k_topo_csr_fill(const TopoBuild t, const int off) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= t.dims->n_edges || e >= t.E) return;
  const int s = t.src[e], d = t.dst[e];
  if ((unsigned)s >= (unsigned)t.V || (unsigned)d >= (unsigned)t.V) return;
  const int base = t.scan[off];
  t.half[t.scan[off + s] - base + atomicAdd(&t.cur[s], 1)] = (uint32_t)e;
  t.half[t.scan[off + d] - base + atomicAdd(&t.cur[d], 1)] = (uint32_t)e | kRole;
}

// Rows in ascending edge id, and the first round of the components: every vertex under its smallest neighbour (no atomics: a vertex
// writes its own parent only; ECL-CC's initialisation).  What is left for k_topo_hook are the trees of the local minima.
__global__ void __launch_bounds__(256) k_topo_rows(const TopoBuild t, const int off) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  int d = 0;
  if (v < t.V) {
    const int base = t.scan[off];
    const int r0 = t.scan[off + v] - base, r1 = t.scan[off + v + 1] - base;
    t.row_ptr[v] = r0;
    if (v == t.V - 1) t.row_ptr[t.V] = r1;
    d = r1 - r0;
    const int pv = (int)cc_mix((uint32_t)v, t.cc_bits);
    int lo = pv;
    if (d <= 64) {  // ascending edge id: the order in which the reference's edge scatter reaches this vertex
      uint32_t* const row = t.half + r0;
      for (int i = 1; i < d; ++i) {
        const uint32_t x = row[i];
        int j = i - 1;
        while (j >= 0 && (row[j] & ~kRole) > (x & ~kRole)) row[j + 1] = row[j], --j;
        row[j + 1] = x;
      }
      for (int i = 0; i < d; ++i) {
        const uint32_t h = row[i];
        const int e = (int)(h & ~kRole);
        lo = min(lo, (int)cc_mix((uint32_t)((h & kRole) ? t.src[e] : t.dst[e]), t.cc_bits));
      }
    }
    t.parent[pv] = lo;
  }
  // one atomic per wave (every vertex adding to the same word would run at ~3 ns each)
  int m = d;
  for (int s = 32; s > 0; s >>= 1) m = max(m, __shfl_xor(m, s, 64));
  if ((threadIdx.x & 63) == 0 && m > 0) {
    atomicMax(&t.dims->max_degree, m);
    if (m > 64) atomicOr(&t.dims->flags, kTopoBadDegree);
  }
}

// Components, the synchronous rounds (Shiloach-Vishkin style: plain loads, the kernel boundary keeps them coherent).  Over every edge
// whose ends stand under different roots the larger root is put under the smaller by an atomic minimum: the smallest candidate
// wins, parent < self holds (no cycle can form), and a root that was hooked a moment ago merely gets a smaller parent -- the link it
// loses is found again over its edge in the next round.  A round divides the number of trees by ~7 (a tree survives as a root only
// if no neighbouring tree has a smaller one): 52 k vertices -> 7.5 k trees after k_topo_rows -> ~1 k -> ~150 -> ...; measured, a
// 1080p Delaunay graph is one tree after three rounds (the asynchronous last one finds nothing left to do).
// Every thread finds its two roots itself and leaves its vertices directly under them (any ancestor is a valid parent).
__device__ __forceinline__ int cc_find_plain(int* parent, const int p) {
  int r = parent[p];
  if (r == p) return r;
  for (int q = parent[r]; q != r; q = parent[r]) r = q;
  parent[p] = r;
  return r;
}
__global__ void __launch_bounds__(256) k_topo_hook_min(const TopoBuild t) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  int hi = -1, lo = 0;
  if (e < t.dims->n_edges && e < t.E) {
    const int a = t.src[e], b = t.dst[e];
    if ((unsigned)a < (unsigned)t.V && (unsigned)b < (unsigned)t.V) {
      const int ra = cc_find_plain(t.parent, (int)cc_mix((uint32_t)a, t.cc_bits)), rb = cc_find_plain(t.parent, (int)cc_mix((uint32_t)b, t.cc_bits));
      if (ra != rb) hi = max(ra, rb), lo = min(ra, rb);
    }
  }
  // One atomic per wave and target root, with the smallest candidate of the lanes that share it: the edges of a wave are neighbours
  // in the triangulator's order and mostly cross between the same two trees -- every edge adding its own minimum to the same word
  // made the round in which the big trees meet 0.55 ms at 1080p (same-address atomics complete at ~3 ns each); this way 0.04 ms.
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(hi >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int target = __shfl(hi, leader, 64);
    const bool mine = hi == target;
    int m = mine ? lo : 0x7fffffff;
    for (int s = 32; s > 0; s >>= 1) m = min(m, __shfl_xor(m, s, 64));
    if (lane == leader) atomicMin(&t.parent[target], m);
    todo &= ~__ballot(mine);
  }
}
// The last round, asynchronous: the few trees left are joined over the edges that still cross between them by lock-free hooking
// (above).  Running ALL edges through it from the start made every one of them read the nodes under the final root from L2 --
// one hot cache line, and the failed compare-and-swaps on it serialise: 0.09 ms at 640x480 and 0.56 ms at 1080p; after the
// synchronous rounds nearly every edge sees equal parents and leaves at once (what this round adds is the guarantee).
__global__ void __launch_bounds__(256) k_topo_hook(const TopoBuild t) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= t.dims->n_edges || e >= t.E) return;
  const int a = t.src[e], b = t.dst[e];
  if ((unsigned)a >= (unsigned)t.V || (unsigned)b >= (unsigned)t.V) return;
  const int pa = (int)cc_mix((uint32_t)a, t.cc_bits), pb = (int)cc_mix((uint32_t)b, t.cc_bits);
  if (cc_find_plain(t.parent, pa) == cc_find_plain(t.parent, pb)) return;  // (one tree already, as of this kernel's start)
  cc_hook(t.parent, pa, pb);
}

// Root of every vertex (cur[v], free since the rows are filled) and the component's label: its smallest vertex id.  The lanes of a
// wave hold ascending vertex ids and mostly one root: the lowest lane of every distinct root in the wave does the atomic.
__global__ void __launch_bounds__(256) k_topo_roots(const TopoBuild t) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  int r = -1;
  if (v < t.V) {
    r = (int)cc_mix((uint32_t)v, t.cc_bits);
    for (int p = t.parent[r]; p != r; p = t.parent[r]) r = p;
    t.cur[v] = r;
  }
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(r >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lr = __shfl(r, leader, 64);
    if (lane == leader) atomicMin(&t.minid[lr], v);
    todo &= ~__ballot(r == lr);
  }
}

// The sort key of the walk, (component label, Morton code), formed as the radix sort reads it.
struct WalkKey {
  const int32_t *minid, *root;
  const uint32_t* morton;
  __host__ __device__ uint64_t operator()(int v) const { return ((uint64_t)(uint32_t)minid[root[v]] << 32) | morton[v]; }
};

// One workgroup per 512 walk positions (= 8 slices).  Inside a window the vertices are re-ordered by descending degree, stably,
// and never across a component border (nltgv2_pack.hpp: uniform slice widths): a bitonic sort in LDS of the keys
// (component run inside the window, 127 - degree, position) -- all distinct, so the order is the host's stable counting sort.
__global__ void __launch_bounds__(kWindow) k_topo_windows(const TopoBuild t) {
  __shared__ uint32_t s_key[kWindow];
  __shared__ int s_o[kWindow], s_cnt[kWindow / 64], s_w[8], s_last;
  const int tid = threadIdx.x, i = blockIdx.x * kWindow + tid;
  const bool valid = i < t.V;
  int o = -1, d = 0;
  bool cb = false;
  if (valid) {
    o = t.order_m[i];
    const int c = (int)(t.key_out[i] >> 32);
    d = t.row_ptr[o + 1] - t.row_ptr[o];
    t.rid_of[o] = i;
    cb = i == 0 || (int)(t.key_out[i - 1] >> 32) != c;
    t.wflag[i] = (uint8_t)((d > 127 ? 127 : d) | (cb ? 0x80 : 0));
  }
  // component run of the position inside the window = component borders at or before it
  const unsigned long long bal = __ballot(cb);
  if ((tid & 63) == 0) s_cnt[tid >> 6] = __popcll(bal);
  if (tid < 8) s_w[tid] = 0;
  s_o[tid] = o;
  __syncthreads();
  int run = __popcll(bal & ((2ull << (tid & 63)) - 1ull));
  for (int k = 0; k < (tid >> 6); ++k) run += s_cnt[k];
  s_key[tid] = valid ? ((uint32_t)run << 16) | ((uint32_t)(127 - (d > 127 ? 127 : d)) << 9) | (uint32_t)tid : 0xffffffffu;
  __syncthreads();
  for (int k = 2; k <= kWindow; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int p = tid ^ j;
      if (p > tid) {
        const uint32_t a = s_key[tid], b = s_key[p];
        const bool up = (tid & k) == 0;
        if ((a > b) == up) s_key[tid] = b, s_key[p] = a;
      }
      __syncthreads();
    }
  }
  const uint32_t key = s_key[tid];
  if (key != 0xffffffffu) {  // packed slot `i` takes the vertex that stood at window position key & 511
    const int oo = s_o[key & 511u], dd = 127 - (int)((key >> 9) & 127u);
    t.perm[i] = oo, t.iperm[oo] = i, t.pdeg[i] = dd;
    atomicMax(&s_w[tid >> 6], dd);
  } else if (i < t.n_slices * 64) {
    t.perm[i] = -1, t.pdeg[i] = 0;  // padding lanes of the last slice (the valid members fill the slots before them)
  }
  __syncthreads();
  if (tid < 8 && blockIdx.x * 8 + tid < t.n_slices) st_agent(&t.width[blockIdx.x * 8 + tid], s_w[tid]);
  // the workgroup that finishes last turns the slice widths into slice_row
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(&t.counters[0], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  int* const s_sum = reinterpret_cast<int*>(s_key);
  const int n = t.n_slices, chunk = (n + kWindow - 1) / kWindow;
  int sum = 0;
  for (int k = tid * chunk; k < min(n, (tid + 1) * chunk); ++k) sum += ld_agent(&t.width[k]);
  s_sum[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int k = 0; k < kWindow; ++k) {
      const int x = s_sum[k];
      s_sum[k] = acc, acc += x;
    }
    t.slice_row[n] = acc, t.dims->rows = acc;
  }
  __syncthreads();
  int acc = s_sum[tid];
  for (int k = tid * chunk; k < min(n, (tid + 1) * chunk); ++k) t.slice_row[k] = acc, acc += ld_agent(&t.width[k]);
}

// The greedy patch walks.  A walk is sequential inside a segment of 256 walk positions and independent between segments
// (build_patch_rows / build_patch_walk2 start a patch at every segment and component border): lane l of a workgroup walks segment
// 64 b + l, wave 0 with one lane per half-edge (E), wave 1 with two half-edges per lane (E2).  The 16 K degree bytes of the
// workgroup's segments go through LDS as dwords of four consecutive positions ([position / 4][lane]: the 64 lanes read neighbouring
// dwords), so do the results.
//
// nltgv2_pack.hpp's WaveFit with row packing, on the four row fills packed into one word (a byte each, 0..16): a vertex of `need`
// lanes goes into the first row with room; one of 9-16 lanes needs an EMPTY row and fills it -- which is "room for 16" --; one of
// more than 16 starts a patch of its own and fills whole rows.
// (the four fills live in the bytes of one word: `st + eff * 0x01010101` adds `eff` to each, a byte + 0x6f has its top bit set
//  iff it is above 16, and the lowest clear top bit names the first row with room)
__device__ __forceinline__ int walk_place(uint32_t& st, const int need, const bool force_new, bool& begins) {
  const uint32_t eff = need > 8 ? 16u : (uint32_t)need;
  const uint32_t fits = ~(st + eff * 0x01010101u + 0x6f6f6f6fu) & 0x80808080u;
  begins = force_new || need > 16 || fits == 0u;
  const int sh = (__ffs((int)fits) - 1) & 24;  // 8 * (first row with room)
  const int f = 2 * sh + (int)((st >> sh) & 0xffu);
  const uint32_t fresh = need > 16 ? 0x10101010u >> (8 * (4 - ((need + 15) >> 4))) : eff;  // a new patch: rows empty, then this vertex
  st = begins ? fresh : st + (eff << sh);
  return begins ? 0 : f;
}

__global__ void __launch_bounds__(128) k_topo_walk(const TopoBuild t, const int n_seg) {
  __shared__ uint32_t s_in[kWalkSegment / 4 * kWalkStride];
  __shared__ uint32_t s_out[2][kWalkSegment / 4 * kWalkStride];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const size_t base = (size_t)blockIdx.x * kWalkSegment * kWalkLanes;
  for (int idx = tid; idx < kWalkSegment / 4 * kWalkLanes; idx += 128)  // dword idx of the block's bytes = (segment, position / 4)
    s_in[(idx & 63) * kWalkStride + (idx >> 6)] = reinterpret_cast<const uint32_t*>(t.wflag + base)[idx];  // (the buffer is padded to whole workgroups)
  __syncthreads();
  const int seg = blockIdx.x * kWalkLanes + lane;
  const int start = seg * kWalkSegment;
  const int n = seg < n_seg ? min(kWalkSegment, t.V - start) : 0;
  uint32_t st = 0;
  int count = 0;
  for (int j4 = 0; j4 < (n + 3) >> 2; ++j4) {
    const uint32_t x = s_in[j4 * kWalkStride + lane];
    uint32_t y = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = (x >> (8 * q)) & 0xff, d = b & 0x7f;
      const int need = w == 0 ? max(d, 1) : max((d + 1) >> 1, 1);
      bool begins;
      const int f = walk_place(st, need, (j4 | q) == 0 || (b & 0x80), begins);
      const bool live = 4 * j4 + q < n;
      count += (begins && live) ? 1 : 0;
      y |= (uint32_t)(f | (begins ? 0x80 : 0)) << (8 * q);
    }
    s_out[w][j4 * kWalkStride + lane] = y;
  }
  if (seg < n_seg) t.seg_count[w][seg] = count;
  __syncthreads();
  for (int idx = tid; idx < kWalkSegment / 4 * kWalkLanes; idx += 128) {
#pragma unroll
    for (int ww = 0; ww < 2; ++ww) reinterpret_cast<uint32_t*>(t.vf[ww] + base)[idx] = s_out[ww][(idx & 63) * kWalkStride + (idx >> 6)];
  }
}

// Patch tables from the walks: workgroup (segment, which walk), one thread per walk position.  A patch's index = patches of the
// segments before + its rank inside the segment.
__global__ void __launch_bounds__(kWalkSegment) k_topo_patches(const TopoBuild t, const int n_seg) {
  __shared__ uint8_t s_b[kWalkSegment + 64];
  __shared__ int s_red[kWalkSegment / 64], s_cnt[kWalkSegment / 64], s_max;
  const int w = blockIdx.y, seg = blockIdx.x, tid = threadIdx.x, i = seg * kWalkSegment + tid;
  int32_t* const info = w == 0 ? t.wg_info : t.wg2_info;
  int32_t* const v0 = w == 0 ? t.wg_v0 : t.wg2_v0;
  uint8_t* const vfirst = w == 0 ? t.wg_vfirst : t.wg2_vfirst;
  // patches of the segments before this one
  int before = 0;
  for (int u = tid; u < seg; u += kWalkSegment) before += t.seg_count[w][u];
  for (int s = 32; s > 0; s >>= 1) before += __shfl_xor(before, s, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = before;
  const int b = i < t.V ? t.vf[w][i] : 0x80;  // (beyond the last vertex: a border, so that the last patch ends there)
  s_b[tid] = (uint8_t)b;
  if (tid < 64) s_b[kWalkSegment + tid] = 0x80;
  if (tid == 0) s_max = 0;
  const bool begins = i < t.V && (b & 0x80);
  const unsigned long long bal = __ballot(begins);
  if ((tid & 63) == 0) s_cnt[tid >> 6] = __popcll(bal);
  __syncthreads();
  int base = 0, rank = __popcll(bal & ((1ull << (tid & 63)) - 1ull)), total = 0;
  for (int k = 0; k < kWalkSegment / 64; ++k) {
    base += s_red[k];
    if (k < (tid >> 6)) rank += s_cnt[k];
    total += s_cnt[k];
  }
  if (i < t.V) vfirst[i] = (uint8_t)(b & 0x7f);
  if (begins) {
    const int p = base + rank;
    int n_local = 1;
    while (!(s_b[tid + n_local] & 0x80)) ++n_local;
    int maxd = 1;
    for (int u = 0; u < n_local; ++u) {
      const int d = t.wflag[i + u] & 0x7f;
      maxd = max(maxd, w == 0 ? d : (d + 1) >> 1);
    }
    info[4 * p] = i, info[4 * p + 1] = 0, info[4 * p + 2] = n_local, info[4 * p + 3] = maxd;
    v0[p] = i;
    atomicMax(&s_max, n_local);
  }
  __syncthreads();
  if (tid == 0) {
    atomicMax(w == 0 ? &t.dims->wg_lcap : &t.dims->wg2_lcap, s_max);
    if (seg == n_seg - 1) *(w == 0 ? &t.dims->wg_count : &t.dims->wg2_count) = base + total;
  }
}

}  // namespace

size_t topo_sort_temp_bytes(int V, int n_scan) {
  size_t a = 0, b = 0;
  const WalkKey kf{nullptr, nullptr, nullptr};
  auto keys = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), kf);
  (void)rocprim::radix_sort_pairs(nullptr, a, keys, (uint64_t*)nullptr, rocprim::counting_iterator<int32_t>(0), (int32_t*)nullptr,
                                  (size_t)std::max(V, 1), 0u, 64u, (hipStream_t) nullptr);
  ScanInput f{nullptr, nullptr, nullptr, 0, 0, 0};
  auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), f);
  (void)rocprim::exclusive_scan(nullptr, b, in, (int32_t*)nullptr, 0, (size_t)std::max(n_scan, 1), rocprim::plus<int32_t>(), (hipStream_t) nullptr);
  return std::max(a, b) + 256;
}

int launch_topo_feat_build(const int32_t* feat, int V, uint32_t* stamp, int32_t* key, int32_t* val, int tab_bits, uint32_t gen, hipStream_t s) {
  if (V <= 0) return 0;
  hipLaunchKernelGGL(k_topo_feat_build, grid1d(V), dim3(256), 0, s, feat, V, stamp, key, val, tab_bits, gen);
  return (int)hipGetLastError();
}

static int topo_scan(const TopoBuild& t, int Eo, int E, hipStream_t s) {
  ScanInput f{t.first_k, t.old_edge, t.deg, Eo, E, t.V};
  auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), f);
  size_t bytes = t.sort_tmp_bytes;
  const hipError_t e = rocprim::exclusive_scan(t.sort_tmp, bytes, in, t.scan, 0, (size_t)(Eo + E + t.V + 1), rocprim::plus<int32_t>(), s);
  return (int)e;
}

int launch_topo_sync_front(const TopoBuild& t, hipStream_t s) {
  hipLaunchKernelGGL(k_topo_init, grid1d(std::max(std::max(t.V, t.Eo), 1 << t.cc_bits)), dim3(256), 0, s, t);
  hipLaunchKernelGGL(k_topo_edges, grid1d(t.E), dim3(256), 0, s, t);
  int e = topo_scan(t, t.Eo, t.E, s);
  if (e) return e;
  hipLaunchKernelGGL(k_topo_new_edges, grid1d(t.E), dim3(256), 0, s, t);
  return (int)hipGetLastError();
}

int launch_topo_upload_front(const TopoBuild& t, hipStream_t s) {
  hipLaunchKernelGGL(k_topo_init, grid1d(std::max(t.V, 1 << t.cc_bits)), dim3(256), 0, s, t);
  if (t.E > 0) hipLaunchKernelGGL(k_topo_edges_given, grid1d(t.E), dim3(256), 0, s, t);
  return topo_scan(t, 0, 0, s);
}

// `t.Eo + t.E` = where the degrees start in the scan: the sync front leaves (Eo, E) as they are, the upload front passes Eo = 0 and
// reads its degrees from offset 0 -- the caller states which through t.tri_edges (sync) or its absence.
int launch_topo_back(const TopoBuild& t, hipStream_t s) {
  const int off = t.tri_edges ? t.Eo + t.E : 0;
  if (t.E > 0 && !t.tri_edges) hipLaunchKernelGGL(k_topo_csr_fill, grid1d(t.E), dim3(256), 0, s, t, off);  // (sync: k_topo_new_edges did it)
  hipLaunchKernelGGL(k_topo_rows, grid1d(t.V), dim3(256), 0, s, t, off);
  if (t.E > 0) {
    for (int round = 0; round < kCcRounds; ++round) hipLaunchKernelGGL(k_topo_hook_min, grid1d(t.E), dim3(256), 0, s, t);
    hipLaunchKernelGGL(k_topo_hook, grid1d(t.E), dim3(256), 0, s, t);
  }
  hipLaunchKernelGGL(k_topo_roots, grid1d(t.V), dim3(256), 0, s, t);
  size_t bytes = t.sort_tmp_bytes;
  const WalkKey kf{t.minid, t.cur, t.morton};
  auto keys = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), kf);
  const hipError_t e = rocprim::radix_sort_pairs(t.sort_tmp, bytes, keys, t.key_out, rocprim::counting_iterator<int32_t>(0), t.order_m,
                                                 (size_t)t.V, 0u, 32u + (unsigned)t.cc_bits, s);
  if (e != hipSuccess) return (int)e;
  const int n_win = (t.n_slices * 64 + kWindow - 1) / kWindow;
  hipLaunchKernelGGL(k_topo_windows, dim3((unsigned)n_win), dim3(kWindow), 0, s, t);
  const int n_seg = (t.V + kWalkSegment - 1) / kWalkSegment;
  hipLaunchKernelGGL(k_topo_walk, dim3((unsigned)((n_seg + kWalkLanes - 1) / kWalkLanes)), dim3(128), 0, s, t, n_seg);
  hipLaunchKernelGGL(k_topo_patches, dim3((unsigned)n_seg, 2), dim3(kWalkSegment), 0, s, t, n_seg);
  return (int)hipGetLastError();
}

// Loads this translation unit's code object (the runtime does that at the first use of one of its kernels: several milliseconds that
// flame_nltgv2_create takes on itself so that the first frame does not).
void warm_module_topo() {
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, (const void*)k_topo_walk) != hipSuccess) (void)hipGetLastError();
}

}  // namespace flame_hip
