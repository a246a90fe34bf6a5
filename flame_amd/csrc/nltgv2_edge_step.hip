// nltgv2_edge_step.hip -- the one-launch-per-step sweep on ONE copy of the duals (round 4).
//
// k_fused_step (nltgv2_kernels.hip) keeps a private (q1,q2,q3) per half-edge: 48 bytes per half-edge and step through HBM, twice the
// algorithmic dual traffic.  Here the duals of an edge are stored once, in rows owned by the edge's SOURCE endpoint:
//
//   hidx[slot]  = {neighbour | role bit, edge row}     8 B per half-edge slot of the SELL-64 rows (slot-major, coalesced)
//   erec[row]   = {alpha, beta, dx, dy}               16 B per edge   (static)
//   eq[2][row]  = {q1, q2, q3}                        12 B per edge, read from eq[parity], written to eq[parity ^ 1] by the source
//
// Both endpoints still evaluate the edge's dual update themselves (identical inputs, identical instructions: bit-identical results, and
// no dual -> primal dependency inside a step); the target endpoint reads the edge row its source reads -- one of the two reads comes
// from HBM, the other from the L2 of the same XCD (the slices of an XCD are contiguous, neighbours are a few slices apart).  Edge rows
// are numbered slice by slice and, within a slice, slot row by slot row in lane order: at slot row k the source lanes of a wave read and
// write one compact run of rows.  Per edge and step: 16 (hidx, both ends) + 16 (erec) + 12 + 12 (eq) = 56 bytes instead of 96.
//
// The at-rest format of the duals stays hq (the persistent kernels and the transactional snapshots use it): a run of n steps converts
// hq -> eq once, steps n times, converts back (nltgv2_run.hip).  Arithmetic: the reference's, expression for expression
// (/root/reference/src/flame/optimizers/nltgv2_l1_graph_regularizer.cc:89-174), same per-vertex accumulation order as k_fused_step.
#include "nltgv2_device.hpp"

namespace flame_hip {

namespace {

struct Q3 {
  float q1, q2, q3;
};

#ifndef STEP_WINDOW
#define STEP_WINDOW 8
#endif
constexpr int kStepWindow = STEP_WINDOW;        // slices per workgroup of k_edge_step: 512 consecutive packed vertices (a compact region: Morton order)
constexpr int kStepRowsInLds = 256 * kStepWindow;  // edge rows of a workgroup staged in LDS (a window owns ~1500; rows beyond are read from memory)

// One workgroup = kStepWindow consecutive slices.  The window's own edge rows (contiguous: rows are numbered slice by slice) and the
// bars of its 512 vertices are staged in LDS with coalesced loads; a half-edge whose edge row / neighbour lies inside the window (~88 %)
// reads them there, the others gather from memory (L2 of the same XCD, mostly).  Without the staging every half-edge costs three
// 16-byte gathers, each a 64-byte L2 -> L1 transfer: the sweep was then bound by those requests, not by HBM (profiles/r04_edge_step.txt).
template <int U, bool WRITE_PREV>
__global__ void __launch_bounds__(64 * kStepWindow)
k_edge_step(const int n_slices, const int windows_per_xcd, const int32_t* __restrict__ slice_row, const int32_t* __restrict__ slice_edges,
            const int2* __restrict__ hidx, const float4* __restrict__ erec, const Q3* __restrict__ eq_in, Q3* __restrict__ eq_out,
            float4* __restrict__ vstate, const float2* __restrict__ vaux, const float4* __restrict__ bar_in, float4* __restrict__ bar_out,
            float4* __restrict__ vprev, const SolverParams p, const int spare_row, int* __restrict__ err) {
  __shared__ float4 s_bar[64 * kStepWindow];
  __shared__ float4 s_erec[kStepRowsInLds];
  __shared__ float s_eq[3 * kStepRowsInLds];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;  // (placement as in k_fused_step: workgroup b runs on XCD b % 8, one contiguous range of windows per XCD)
  const int window = (b & 7) * windows_per_xcd + (b >> 3);
  const int slice0 = window * kStepWindow;
  if ((b >> 3) >= windows_per_xcd || slice0 >= n_slices) return;  // (uniform per workgroup)
  const int slice1 = min(slice0 + kStepWindow, n_slices);
  const int v0 = slice0 * 64;
  const int er0 = slice_edges[slice0];
  const int n_rows = min(slice_edges[slice1] - er0, kStepRowsInLds);
  {  // all loads of the staging first, then the LDS writes: one memory round trip, not one per loop iteration
    constexpr int kT = 64 * kStepWindow, kPer = kStepRowsInLds / kT;
    float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), se[kPer];
    Q3 sq[kPer];
    if (slice0 + wave < n_slices) sb = bar_in[v0 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {  // (branch-free: a load under a branch is waited for inside the branch; row er0 exists even for an empty window: the spare row)
      const int i = j * kT + (int)threadIdx.x;
      const int r = er0 + (i < n_rows ? i : 0);
      se[j] = erec[r], sq[j] = eq_in[r];
    }
    s_bar[threadIdx.x] = sb;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int i = j * kT + (int)threadIdx.x;
      if (i < n_rows) s_erec[i] = se[j], s_eq[3 * i] = sq[j].q1, s_eq[3 * i + 1] = sq[j].q2, s_eq[3 * i + 2] = sq[j].q3;
    }
  }
  __syncthreads();
  const int slice = slice0 + wave;
  if (slice >= n_slices) return;

  const int v = slice * 64 + lane;
  const int row0 = slice_row[slice];
  const int D = slice_row[slice + 1] - row0;

  const float4 st = vstate[v];
  const float2 aux = vaux[v];
  const float4 bs = s_bar[threadIdx.x];
  const int deg = __float_as_int(aux.y);

  float x = st.x, w1 = st.y, w2 = st.z;
  const float x_prev = x, w1_prev = w1, w2_prev = w2;  // step()'s prev copy, cc:37-42
  bool ok = true;

  size_t slot = (size_t)row0 * 64 + lane;
  for (int k0 = 0; k0 < D; k0 += U) {
    int2 ix[U];
    float4 er[U];
    Q3 q[U];
    float4 bn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ix[u] = hidx[slot + (size_t)u * 64];  // (spare rows keep this in bounds)
    // What the window does not hold comes from memory, all requests of the chunk back to back.  Branch-free (a load under a branch is
    // waited for inside the branch): lanes served by LDS, and unused slots, all name one harmless address (the spare edge row / the lane's own bar).
    float4 ger[U], gbn[U];
    Q3 gq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool act = (k0 + u) < deg;
      const unsigned lr = (unsigned)(ix[u].y - er0), lv = (unsigned)((ix[u].x & 0x7fffffff) - v0);
      const int gr = (act && lr >= (unsigned)n_rows) ? ix[u].y : spare_row;
      const int gv = (act && lv >= 64u * kStepWindow) ? (ix[u].x & 0x7fffffff) : v;
      ger[u] = erec[gr];
      gq[u] = eq_in[gr];
      gbn[u] = bar_in[gv];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned lr = (unsigned)(ix[u].y - er0), lv = (unsigned)((ix[u].x & 0x7fffffff) - v0);
      const bool in_r = lr < (unsigned)n_rows, in_v = lv < 64u * kStepWindow;
      const unsigned sr = in_r ? lr : 0u, sv = in_v ? lv : 0u;
      const float4 le = s_erec[sr], lb = s_bar[sv];
      const Q3 lq = Q3{s_eq[3 * sr], s_eq[3 * sr + 1], s_eq[3 * sr + 2]};
      er[u] = in_r ? le : ger[u];
      q[u] = in_r ? lq : gq[u];
      bn[u] = in_v ? lb : gbn[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool act = (k0 + u) < deg;
      const bool is_target = ix[u].x < 0;
      const float alpha = er[u].x, beta = er[u].y;
      const float dx = er[u].z, dy = er[u].w;  // pos_i - pos_j of the EDGE (source minus target)
      // (i) = source, (j) = target
      const float xbi = is_target ? bn[u].x : bs.x, xbj = is_target ? bs.x : bn[u].x;
      const float w1bi = is_target ? bn[u].y : bs.y, w1bj = is_target ? bs.y : bn[u].y;
      const float w2bi = is_target ? bn[u].z : bs.z, w2bj = is_target ? bs.z : bn[u].z;
      bool okq = true;
      const EdgeOut o = edge_dual(p, alpha, beta, dx, dy, q[u].q1, q[u].q2, q[u].q3, xbi, w1bi, w2bi, xbj, w1bj, w2bj, okq);
      // primal scatter, this endpoint's share, cc:126-141
      const float t1 = o.q1 * p.step_x * alpha;
      const float t2 = o.q2 * p.step_x * beta;
      const float t3 = o.q3 * p.step_x * beta;
      float nx, nw1, nw2;
      if (is_target) {
        nx = x + t1;
        nw1 = w1 + t2;
        nw2 = w2 + t3;
      } else {
        nx = x - t1;
        nw1 = w1 + t1 * dx;
        nw2 = w2 + t1 * dy;
        nw1 = nw1 - t2;
        nw2 = nw2 - t3;
      }
      if (act) {
        x = nx, w1 = nw1, w2 = nw2;
        ok = ok && okq;
#ifndef EDGE_VAR_NOSTORE
        if (!is_target) eq_out[ix[u].y] = Q3{o.q1, o.q2, o.q3};
#endif
      }
    }
    slot += (size_t)U * 64;
  }

  // proxL1 per vertex, cc:147-151
  x = prox_l1(p.x_min, p.x_max, p.step_x, p.data_factor * aux.x, x, st.w);
  // extraGradientStep, cc:160-171
  float xb = x + p.theta * (x - x_prev);
  xb = (xb < p.x_min) ? p.x_min : xb;
  xb = (xb > p.x_max) ? p.x_max : xb;
  const float w1b = w1 + p.theta * (w1 - w1_prev);
  const float w2b = w2 + p.theta * (w2 - w2_prev);

  vstate[v] = make_float4(x, w1, w2, st.w);
  bar_out[v] = make_float4(xb, w1b, w2b, 0.0f);
  if (WRITE_PREV) vprev[v] = make_float4(x_prev, w1_prev, w2_prev, 0.0f);
  if (!ok) atomicOr(err, 1);
}

// ---- the edge rows of a topology (built on the device at the first per-step run of that topology; nltgv2_pack.hpp build_edge_rows
// is the same numbering on the host, compared word for word by flame_nltgv2_layout_selftest) --------------------------------------
// one wave per slice: how many edges its vertices own (source role)
__global__ void __launch_bounds__(256)
k_edge_rows_count(int n_slices, const int32_t* __restrict__ slice_row, const uint32_t* __restrict__ rec_nbr,
                  const int32_t* __restrict__ rec_edge, int32_t* __restrict__ slice_edges) {
  const int slice = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= n_slices) return;
  const int r0 = slice_row[slice], r1 = slice_row[slice + 1];
  int n = 0;
  for (int r = r0; r < r1; ++r) {
    const size_t s = (size_t)r * 64 + lane;
    n += __popcll(__ballot(rec_edge[s] >= 0 && !(rec_nbr[s] & kRole)));
  }
  if (lane == 0) slice_edges[slice] = n;
}

// one block: exclusive scan of slice_edges in place (+ the total behind it)
__global__ void __launch_bounds__(1024)
k_edge_rows_scan(int n_slices, int32_t* __restrict__ slice_edges) {
  __shared__ int32_t part[1024];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n_slices; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const int32_t mine = i < n_slices ? slice_edges[i] : 0;
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int32_t add = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    const int32_t c = carry;
    if (i < n_slices) slice_edges[i] = c + part[threadIdx.x] - mine;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) slice_edges[n_slices] = carry;
}

// one wave per slice: number the owned edges slot row by slot row in lane order, and write their static words
__global__ void __launch_bounds__(256)
k_edge_rows_assign(int n_slices, const int32_t* __restrict__ slice_row, const uint32_t* __restrict__ rec_nbr,
                   const int32_t* __restrict__ rec_edge, const int32_t* __restrict__ slice_edges, const int32_t* __restrict__ src,
                   const int32_t* __restrict__ dst, const float* __restrict__ alpha, const float* __restrict__ beta,
                   const float2* __restrict__ pos, int32_t* __restrict__ edge_row, float4* __restrict__ erec) {
  const int slice = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= n_slices) return;
  const int r0 = slice_row[slice], r1 = slice_row[slice + 1];
  int next = slice_edges[slice];
  for (int r = r0; r < r1; ++r) {
    const size_t s = (size_t)r * 64 + lane;
    const int e = rec_edge[s];
    const bool own = e >= 0 && !(rec_nbr[s] & kRole);
    const unsigned long long m = __ballot(own);
    if (own) {
      const int row = next + __popcll(m & ((1ull << lane) - 1ull));
      edge_row[e] = row;
      const float2 pi = pos[src[e]], pj = pos[dst[e]];
      erec[row] = make_float4(alpha[e], beta[e], pi.x - pj.x, pi.y - pj.y);
    }
    next += __popcll(m);
  }
}

__global__ void __launch_bounds__(256)
k_edge_rows_index(int64_t n_slots, int spare_row, const uint32_t* __restrict__ rec_nbr, const int32_t* __restrict__ rec_edge,
                  const int32_t* __restrict__ edge_row, int2* __restrict__ hidx) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const int e = rec_edge[s];
  hidx[s] = make_int2((int)rec_nbr[s], e >= 0 ? edge_row[e] : spare_row);
}

// hq -> eq (the source's copy) / eq -> hq (both copies): one thread per half-edge slot
__global__ void __launch_bounds__(256)
k_q_to_edge_rows(int64_t n_slots, const int32_t* __restrict__ rec_edge, const int2* __restrict__ hidx, const float4* __restrict__ hq,
                 Q3* __restrict__ eq) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots || rec_edge[s] < 0) return;
  const int2 ix = hidx[s];
  if (ix.x < 0) return;
  const float4 q = hq[s];
  eq[ix.y] = Q3{q.x, q.y, q.z};
}

__global__ void __launch_bounds__(256)
k_q_from_edge_rows(int64_t n_slots, const int32_t* __restrict__ rec_edge, const int2* __restrict__ hidx, const Q3* __restrict__ eq,
                   float4* __restrict__ hq) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots || rec_edge[s] < 0) return;
  const Q3 q = eq[hidx[s].y];
  float* o = reinterpret_cast<float*>(hq + s);  // (.w = beta stays)
  o[0] = q.q1, o[1] = q.q2, o[2] = q.q3;
}

inline dim3 grid1d(int64_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

int launch_edge_rows_build(const CanonArgs& c, const FusedArgs& a, hipStream_t s) {
  if (a.n_slices <= 0 || a.n_slots <= 0) return 0;
  const dim3 per_slice((unsigned)((a.n_slices + 3) / 4));
  hipLaunchKernelGGL(k_edge_rows_count, per_slice, dim3(256), 0, s, a.n_slices, a.slice_row, a.rec_nbr, a.rec_edge, a.slice_edges);
  hipLaunchKernelGGL(k_edge_rows_scan, dim3(1), dim3(1024), 0, s, a.n_slices, a.slice_edges);
  hipLaunchKernelGGL(k_edge_rows_assign, per_slice, dim3(256), 0, s, a.n_slices, a.slice_row, a.rec_nbr, a.rec_edge, a.slice_edges, c.src,
                     c.dst, c.alpha, c.beta, c.pos, a.edge_row, a.erec);
  hipLaunchKernelGGL(k_edge_rows_index, grid1d(a.n_slots), dim3(256), 0, s, a.n_slots, c.E, a.rec_nbr, a.rec_edge, a.edge_row, a.hidx);
  return (int)hipGetLastError();
}

int launch_q_to_edge_rows(const FusedArgs& a, int parity, hipStream_t s) {
  if (a.n_slots <= 0) return 0;
  hipLaunchKernelGGL(k_q_to_edge_rows, grid1d(a.n_slots), dim3(256), 0, s, a.n_slots, a.rec_edge, a.hidx, a.hq, (Q3*)a.eq[parity]);
  return (int)hipGetLastError();
}

int launch_q_from_edge_rows(const FusedArgs& a, int parity, hipStream_t s) {
  if (a.n_slots <= 0) return 0;
  hipLaunchKernelGGL(k_q_from_edge_rows, grid1d(a.n_slots), dim3(256), 0, s, a.n_slots, a.rec_edge, a.hidx, (const Q3*)a.eq[parity], a.hq);
  return (int)hipGetLastError();
}

int launch_edge_step(const FusedArgs& a, const SolverParams& p, int parity, bool write_prev, int unroll, int /*waves_per_block*/,
                     hipStream_t stream) {
  if (a.n_slices <= 0) return (int)hipSuccess;
  const int n_windows = (a.n_slices + kStepWindow - 1) / kStepWindow;
  const int wpx = (n_windows + 7) / 8;  // windows per XCD
  const dim3 grid((unsigned)(wpx * 8)), block((unsigned)(64 * kStepWindow));
  const float4* bin = a.bar[parity];
  float4* bout = a.bar[parity ^ 1];
  const Q3* qin = (const Q3*)a.eq[parity];
  Q3* qout = (Q3*)a.eq[parity ^ 1];
#define FLAME_LAUNCH(UU, WP)                                                                                                     \
  hipLaunchKernelGGL((k_edge_step<UU, WP>), grid, block, 0, stream, a.n_slices, wpx, a.slice_row, a.slice_edges, a.hidx, a.erec, qin, \
                     qout, a.vstate, a.vaux, bin, bout, a.vprev, p, a.n_edge_rows, a.err)
  if (unroll >= 8) {
    if (write_prev) FLAME_LAUNCH(8, true); else FLAME_LAUNCH(8, false);
  } else {
    if (write_prev) FLAME_LAUNCH(4, true); else FLAME_LAUNCH(4, false);
  }
#undef FLAME_LAUNCH
  return (int)hipGetLastError();
}

}  // namespace flame_hip
