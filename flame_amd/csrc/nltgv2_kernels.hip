// nltgv2_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the NLTGV2-L1 primal-dual solver: the one-launch-per-step sweep, the
// reference's four loops one by one, layout conversions, costs, and the rows around the solver (photometric residual, mesh
// rasteriser, projection, rescale).  The persistent single-launch kernels are in nltgv2_persistent.hip.
//
// Reference arithmetic: /root/reference/src/flame/optimizers/nltgv2_l1_graph_regularizer.{h,cc}
// (cited per kernel).  The whole file is compiled with -ffp-contract=off: the reference build is
// plain x86-64 (no FMA, CMakeLists.txt:24), every expression below keeps the reference's
// left-to-right float evaluation order, and each vertex accumulates its incident-edge updates
// sequentially in ascending edge id -- the order of the reference's edge scatter -- so the results
// are bit-identical to the reference's sequential loops, not merely within tolerance.
//
// This is a sparse-graph stencil at ~0.3 flop/B: no MFMA.  What matters on CDNA4 here is
// (1) 16-byte-per-lane coalesced streams (one vertex per lane of a 64-wide wavefront, SELL-64
// slot-major half-edge arrays), (2) as few dependent global round trips per step as possible
// (one fused kernel per step, loads of a whole slot chunk issued back to back), (3) XCD-aware
// slice placement so a slice's neighbours were written by the same XCD's L2 one step earlier.
#include "nltgv2_device.hpp"

namespace flame_hip {

namespace {

// ------------------------------------------------------------------------------------------------
// Fused step: one launch == one reference step() (cc:33-49) on the SELL-64 layout.
//
//   wave <-> slice of 64 packed vertices, lane <-> vertex.  Slot k of the lane's vertex is its k-th
//   incident half-edge in ascending edge id.  Every half-edge keeps a PRIVATE copy of (q1,q2,q3):
//   both endpoints recompute the edge's dual update from identical inputs with identical
//   instructions, so the two copies stay bit-identical and no cross-lane exchange of q is needed
//   -- this removes the dual->primal global dependency, leaving ONE grid-wide dependency per step
//   (x_bar/w_bar of step t feed step t+1), carried by the kernel boundary through the
//   ping-ponged `bar` arrays.
//
//   U = slots processed per chunk: 2U streaming loads + U gathers are issued back to back, so a
//   slice pays ~2 dependent memory round trips per chunk, not per slot.
// ------------------------------------------------------------------------------------------------
template <int U, bool WRITE_PREV>
__global__ void __launch_bounds__(256)
k_fused_step(const int n_slices, const int slices_per_xcd, const int32_t* __restrict__ slice_row,
             const int4* __restrict__ hrec, float4* __restrict__ hq, float4* __restrict__ vstate,
             const float2* __restrict__ vaux, const float4* __restrict__ bar_in,
             float4* __restrict__ bar_out, float4* __restrict__ vprev, const SolverParams p,
             int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int waves_per_block = blockDim.x >> 6;
  // XCD-aware placement: workgroup b is dispatched to XCD b % 8 (observed, used for speed only);
  // give each XCD one contiguous range of slices so that a slice's gathers hit lines its own
  // XCD's L2 wrote in the previous step.
  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int wave_in_xcd = (b >> 3) * waves_per_block + (threadIdx.x >> 6);
  if (wave_in_xcd >= slices_per_xcd) return;
  const int slice = xcd * slices_per_xcd + wave_in_xcd;
  if (slice >= n_slices) return;

  const int v = slice * 64 + lane;
  const int row0 = slice_row[slice];
  const int D = slice_row[slice + 1] - row0;

  const float4 st = vstate[v];
  const float2 aux = vaux[v];
  const float4 bs = bar_in[v];
  const int deg = __float_as_int(aux.y);

  float x = st.x, w1 = st.y, w2 = st.z;
  const float x_prev = x, w1_prev = w1, w2_prev = w2;  // step()'s prev copy, cc:37-42
  bool ok = true;

  size_t slot = (size_t)row0 * 64 + lane;
  for (int k0 = 0; k0 < D; k0 += U) {
    int4 rec[U];
    float4 q[U];
    float4 bn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // spare rows keep these in bounds past the slice end
      rec[u] = hrec[slot + (size_t)u * 64];
      q[u] = hq[slot + (size_t)u * 64];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) bn[u] = bar_in[rec[u].x & 0x7fffffff];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool act = (k0 + u) < deg;
      const bool is_target = rec[u].x < 0;
      const float alpha = __int_as_float(rec[u].y);
      const float dx = __int_as_float(rec[u].z);  // pos_i - pos_j of the EDGE (source minus target)
      const float dy = __int_as_float(rec[u].w);
      const float beta = q[u].w;
      // (i) = source, (j) = target
      const float xbi = is_target ? bn[u].x : bs.x, xbj = is_target ? bs.x : bn[u].x;
      const float w1bi = is_target ? bn[u].y : bs.y, w1bj = is_target ? bs.y : bn[u].y;
      const float w2bi = is_target ? bn[u].z : bs.z, w2bj = is_target ? bs.z : bn[u].z;
      bool okq = true;
      const EdgeOut o = edge_dual(p, alpha, beta, dx, dy, q[u].x, q[u].y, q[u].z, xbi, w1bi, w2bi,
                                  xbj, w1bj, w2bj, okq);
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
        hq[slot + (size_t)u * 64] = make_float4(o.q1, o.q2, o.q3, beta);
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

// ------------------------------------------------------------------------------------------------
// Canonical (original order, SoA) sweeps: the individually callable pieces of a step.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_save_prev(int V, const float* __restrict__ x, const float* __restrict__ w1,
            const float* __restrict__ w2, float* __restrict__ xp, float* __restrict__ w1p,
            float* __restrict__ w2p) {  // cc:35-42
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  xp[v] = x[v];
  w1p[v] = w1[v];
  w2p[v] = w2[v];
}

// internal::dualStep, cc:89-114: one lane per edge.
__global__ void __launch_bounds__(256)
k_dual_edge_sweep(int E, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                  const float* __restrict__ alpha, const float* __restrict__ beta,
                  const float2* __restrict__ pos, const float* __restrict__ xb,
                  const float* __restrict__ w1b, const float* __restrict__ w2b,
                  float* __restrict__ q1, float* __restrict__ q2, float* __restrict__ q3,
                  const SolverParams p, int* __restrict__ err) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int i = src[e], j = dst[e];
  const float2 pi = pos[i], pj = pos[j];
  bool ok = true;
  const EdgeOut o = edge_dual(p, alpha[e], beta[e], pi.x - pj.x, pi.y - pj.y, q1[e], q2[e], q3[e],
                              xb[i], w1b[i], w2b[i], xb[j], w1b[j], w2b[j], ok);
  q1[e] = o.q1;
  q2[e] = o.q2;
  q3[e] = o.q3;
  if (!ok) atomicOr(err, 1);
}

// internal::primalStep, cc:116-154, as a per-vertex gather over the canonical CSR (ascending edge
// id == the reference's scatter order for that vertex), then proxL1.
__global__ void __launch_bounds__(256)
k_primal_vertex_gather(int V, const int32_t* __restrict__ row_ptr, const uint32_t* __restrict__ half,
                       const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                       const float* __restrict__ alpha, const float* __restrict__ beta,
                       const float2* __restrict__ pos, const float* __restrict__ q1,
                       const float* __restrict__ q2, const float* __restrict__ q3,
                       const float* __restrict__ data, const float* __restrict__ weight,
                       float* __restrict__ x, float* __restrict__ w1, float* __restrict__ w2,
                       const SolverParams p) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  float xv = x[v], w1v = w1[v], w2v = w2[v];
  const int r1 = row_ptr[v + 1];
  for (int r = row_ptr[v]; r < r1; ++r) {
    const uint32_t h = half[r];
    const int e = (int)(h & ~kRole);
    const float a = alpha[e], b = beta[e];
    const float t1 = q1[e] * p.step_x * a;
    const float t2 = q2[e] * p.step_x * b;
    const float t3 = q3[e] * p.step_x * b;
    if (h & kRole) {
      xv = xv + t1;
      w1v = w1v + t2;
      w2v = w2v + t3;
    } else {
      const float2 pi = pos[v], pj = pos[dst[e]];
      xv = xv - t1;
      w1v = w1v + t1 * (pi.x - pj.x);
      w2v = w2v + t1 * (pi.y - pj.y);
      w1v = w1v - t2;
      w2v = w2v - t3;
    }
  }
  x[v] = prox_l1(p.x_min, p.x_max, p.step_x, p.data_factor * weight[v], xv, data[v]);
  w1[v] = w1v;
  w2[v] = w2v;
}

// internal::extraGradientStep, cc:156-174.
__global__ void __launch_bounds__(256)
k_extragradient(int V, const float* __restrict__ x, const float* __restrict__ w1,
                const float* __restrict__ w2, const float* __restrict__ xp,
                const float* __restrict__ w1p, const float* __restrict__ w2p,
                float* __restrict__ xb, float* __restrict__ w1b, float* __restrict__ w2b,
                const SolverParams p) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  float nb = x[v] + p.theta * (x[v] - xp[v]);
  nb = (nb < p.x_min) ? p.x_min : nb;
  nb = (nb > p.x_max) ? p.x_max : nb;
  xb[v] = nb;
  w1b[v] = w1[v] + p.theta * (w1[v] - w1p[v]);
  w2b[v] = w2[v] + p.theta * (w2[v] - w2p[v]);
}

// ------------------------------------------------------------------------------------------------
// canonical <-> packed conversions
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pack_static(int64_t n_slots, const int32_t* __restrict__ rec_edge, const uint32_t* __restrict__ rec_nbr,
              const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
              const float* __restrict__ alpha, const float2* __restrict__ pos, int4* __restrict__ hrec) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const int e = rec_edge[s];
  int4 r;
  r.x = (int)rec_nbr[s];
  if (e >= 0) {
    const float2 pi = pos[src[e]], pj = pos[dst[e]];
    r.y = __float_as_int(alpha[e]);
    r.z = __float_as_int(pi.x - pj.x);
    r.w = __float_as_int(pi.y - pj.y);
  } else {
    r.y = r.z = r.w = 0;
  }
  hrec[s] = r;
}

// ensure_fused in ONE launch: the records (when the positions moved or the topology is new), the duals and the vertex state of the
// packed form, and the zero fill of the other ping-pong half -- thread i does slot i and packed vertex i.  (Four launches between two
// runs cost the free-running solver ~15 us of idle time at every settle point: tools/frame_loop.py --pipelined.)
__global__ void __launch_bounds__(256)
k_pack_all(int64_t n_slots, int n_packed, int with_static, const int32_t* __restrict__ rec_edge, const uint32_t* __restrict__ rec_nbr,
           const int32_t* __restrict__ src, const int32_t* __restrict__ dst, const float* __restrict__ alpha, const float* __restrict__ beta,
           const float2* __restrict__ pos, const float* __restrict__ q1, const float* __restrict__ q2, const float* __restrict__ q3,
           int4* __restrict__ hrec, float4* __restrict__ hq, const int32_t* __restrict__ perm, const int32_t* __restrict__ pdeg,
           const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ xb,
           const float* __restrict__ w1b, const float* __restrict__ w2b, const float* __restrict__ data, const float* __restrict__ weight,
           float4* __restrict__ vstate, float2* __restrict__ vaux, float4* __restrict__ bar, float4* __restrict__ bar_other) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n_slots) {
    const int e = rec_edge[s];
    if (with_static) {
      int4 r;
      r.x = (int)rec_nbr[s];
      if (e >= 0) {
        const float2 pi = pos[src[e]], pj = pos[dst[e]];
        r.y = __float_as_int(alpha[e]);
        r.z = __float_as_int(pi.x - pj.x);
        r.w = __float_as_int(pi.y - pj.y);
      } else {
        r.y = r.z = r.w = 0;
      }
      hrec[s] = r;
    }
    hq[s] = (e >= 0) ? make_float4(q1[e], q2[e], q3[e], beta[e]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (s < n_packed) {
    const int o = perm[s];
    if (o >= 0) {
      vstate[s] = make_float4(x[o], w1[o], w2[o], data[o]);
      vaux[s] = make_float2(weight[o], __int_as_float(pdeg[s]));
      bar[s] = make_float4(xb[o], w1b[o], w2b[o], 0.f);
    } else {
      vstate[s] = make_float4(0.f, 0.f, 0.f, 0.f);
      vaux[s] = make_float2(0.f, __int_as_float(0));
      bar[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bar_other[s] = make_float4(0.f, 0.f, 0.f, 0.f);  // the other ping-pong half must hold valid (zero) values in padding lanes too
  }
}

// ensure_canon in ONE launch: thread i unpacks packed vertex i and edge i.
__global__ void __launch_bounds__(256)
k_unpack_all(int n_packed, int E, const int32_t* __restrict__ perm, const float4* __restrict__ vstate, const float4* __restrict__ bar,
             const float4* __restrict__ vprev, int have_prev, float* __restrict__ x, float* __restrict__ w1, float* __restrict__ w2,
             float* __restrict__ xb, float* __restrict__ w1b, float* __restrict__ w2b, float* __restrict__ xp, float* __restrict__ w1p,
             float* __restrict__ w2p, const int32_t* __restrict__ edge_src_slot, const float4* __restrict__ hq, float* __restrict__ q1,
             float* __restrict__ q2, float* __restrict__ q3) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n_packed) {
    const int o = perm[s];
    if (o >= 0) {
      const float4 st = vstate[s], b = bar[s];
      x[o] = st.x, w1[o] = st.y, w2[o] = st.z;
      xb[o] = b.x, w1b[o] = b.y, w2b[o] = b.z;
      if (have_prev) {
        const float4 pv = vprev[s];
        xp[o] = pv.x, w1p[o] = pv.y, w2p[o] = pv.z;
      }
    }
  }
  if (s < E) {
    const float4 q = hq[edge_src_slot[s]];
    q1[s] = q.x, q2[s] = q.y, q3[s] = q.z;
  }
}

// x * graph_scale in original order (flame.cc:377), from whichever layout is current.
__global__ void __launch_bounds__(256)
k_export_packed(int n_packed, const int32_t* __restrict__ perm, const float4* __restrict__ vstate,
                float scale, float* __restrict__ out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_packed) return;
  const int o = perm[s];
  if (o >= 0) out[o] = vstate[s].x * scale;
}

__global__ void __launch_bounds__(256)
k_export_canonical(int V, const float* __restrict__ x, float scale, float* __restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < V) out[v] = x[v] * scale;
}

// The stand-alone residual sweep (photo_residual_at is defined next to the persistent kernels, which use it too).
__global__ void __launch_bounds__(256)
k_photo_residual(int V, const float2* __restrict__ pos, const float* __restrict__ x, float graph_scale,
                 PhotoGeometry geo, const uint8_t* __restrict__ ref, const uint8_t* __restrict__ cmp, int rows,
                 int cols, int step, int border, float* __restrict__ err) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  err[v] = photo_residual_at(pos[v], x[v] * graph_scale, geo, ref, cmp, rows, cols, step, border);
}

// ... and the same on the packed state (after a run on the one-launch-per-step path, which has no such epilogue)
__global__ void __launch_bounds__(256)
k_photo_residual_packed(int n_packed, const int32_t* __restrict__ perm, const float4* __restrict__ vstate, PhotoFuse photo) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_packed) return;
  const int o = perm[s];
  if (o >= 0)
    photo.err[o] = photo_residual_at(photo.pos[o], vstate[s].x * photo.graph_scale, photo.geo, photo.ref, photo.cmp,
                                     photo.rows, photo.cols, photo.step, photo.border);
}

// ------------------------------------------------------------------------------------------------
// Mesh -> dense inverse-depth map (the step right after the solver, flame.cc:409-437):
//   utils::interpolateMesh (utils/image_utils.cc:373-396) drawing every triangle with
//   utils::DrawShadedTriangleBarycentric (utils/rasterization.cc:164-246, Edge::init
//   rasterization.h:120-154).  The reference draws the triangles one after the other, so where two
//   triangles share pixels (the inclusive w >= 0 rule puts shared edges in both) the LATER triangle
//   wins.  Here every triangle is rasterised concurrently by one wavefront and the order is restored
//   with a 64-bit atomicMax on {triangle index + 1, value bits}: the surviving value is exactly the
//   one the sequential loop leaves behind.  Edge values are integers held in floats (exact below
//   2^24), evaluated per pixel; value = (v1*w1 + (v2*w2 + v3*w3)) / (w1 + (w2 + w3)) as in the SSE
//   code.  Pixel keys live in a rows*cols u64 scratch image, resolved to floats (NaN = uncovered)
//   together with the coverage count of flame.cc:428-437.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float edge_eval(int v0x, int v0y, int v1x, int v1y, int px, int py) {
  const float A = (float)(v1y - v0y);
  const float B = (float)(v0x - v1x);
  const float C = (float)(v1x * v0y - v0x * v1y);
  return (A * (float)px + B * (float)py) + C;
}

__global__ void __launch_bounds__(64)
k_raster_triangles(int T, const int32_t* __restrict__ tris, const float2* __restrict__ vtx,
                   const float* __restrict__ values, float value_scale, const uint8_t* __restrict__ vtx_valid,
                   const uint8_t* __restrict__ tri_valid, unsigned long long* __restrict__ keys, int rows,
                   int cols) {
  const int t = blockIdx.x;
  if (t >= T) return;
  const int a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
  if (tri_valid && !tri_valid[t]) return;
  if (vtx_valid && !(vtx_valid[a] && vtx_valid[b] && vtx_valid[c])) return;
  // (2,1,0): "Triangle spits out points in clockwise order, but drawing function expects CCW"
  const float2 f1 = vtx[c], f2 = vtx[b], f3 = vtx[a];
  // cv::Point2f -> cv::Point is saturate_cast<int> == cvRound == round half to even
  const int p1x = __float2int_rn(f1.x), p1y = __float2int_rn(f1.y);
  const int p2x = __float2int_rn(f2.x), p2y = __float2int_rn(f2.y);
  const int p3x = __float2int_rn(f3.x), p3y = __float2int_rn(f3.y);
  const float v1 = values[c] * value_scale, v2 = values[b] * value_scale, v3 = values[a] * value_scale;
  const int xmin = min(p1x, min(p2x, p3x)), ymin = min(p1y, min(p2y, p3y));
  const int xmax = max(p1x, max(p2x, p3x)), ymax = max(p1y, max(p2y, p3y));
  const int w = ((xmax - xmin) / 4) * 4 + 4;  // the reference walks x in blocks of 4 pixels
  const int h = ymax - ymin + 1;
  const long n = (long)w * h;
  const unsigned long long hi = (unsigned long long)(t + 1) << 32;
  for (long i = threadIdx.x; i < n; i += 64) {
    const int x = xmin + (int)(i % w), y = ymin + (int)(i / w);
    const float w1 = edge_eval(p2x, p2y, p3x, p3y, x, y);
    const float w2 = edge_eval(p3x, p3y, p1x, p1y, x, y);
    const float w3 = edge_eval(p1x, p1y, p2x, p2y, x, y);
    if (w1 >= 0.0f && w2 >= 0.0f && w3 >= 0.0f && x >= 0 && y >= 0 && x < cols && y < rows) {
      const float norm = w1 + (w2 + w3);
      const float val = (v1 * w1 + (v2 * w2 + v3 * w3)) / norm;
      atomicMax(&keys[(long)y * cols + x], hi | (unsigned long long)__float_as_uint(val));
    }
  }
}

constexpr int kResolvePerThread = 8;
__global__ void __launch_bounds__(256)
k_raster_resolve(long n, const unsigned long long* __restrict__ keys, float* __restrict__ img,
                 int* __restrict__ coverage) {
  // kResolvePerThread pixels per thread and ONE atomic per workgroup: an atomic per wave -- 32 k of them on the same word at
  // 1920x1080 -- completed at ~11 ns each and made this kernel 364 us (round 4: 2 M pixels are 24 MB of traffic, microseconds).
  __shared__ int s_count;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  int count = 0;
  const long base = (long)blockIdx.x * blockDim.x * kResolvePerThread + threadIdx.x;
#pragma unroll
  for (int j = 0; j < kResolvePerThread; ++j) {
    const long i = base + (long)j * blockDim.x;
    bool covered = false;
    if (i < n) {
      const unsigned long long k = keys[i];
      covered = (k >> 32) != 0ull;
      const float v = __uint_as_float((unsigned)k);
      img[i] = covered ? v : __builtin_nanf("");
      covered = covered && !(v != v);  // flame.cc:431: counts !isnan
    }
    count += __popcll(__ballot(covered));
  }
  if ((threadIdx.x & 63) == 0 && count) atomicAdd(&s_count, count);
  __syncthreads();
  if (threadIdx.x == 0 && s_count) atomicAdd(coverage, s_count);
}

// ------------------------------------------------------------------------------------------------
// Graph maintenance sweeps of the per-frame path, on the canonical state:
//   k_project_graph   Flame::projectGraph, flame.cc:1888-1905: re-project every vertex into the new
//                     frame with EpipolarGeometry::project(u, idepth, &u_new, &idepth_new)
//                     (stereo/epipolar_geometry.h:152-180; the rotation is Eigen's quaternion *
//                     vector: uv = q.vec x v; uv += uv; v + w*uv + q.vec x uv), write pos and
//                     x = idepth_new / scale back, and report which vertices stay
//                     (cv::Rect_<float>::contains, idepth_new >= 0).
//   k_rescale_*       the rescale_data block, flame.cc:328-351; the mean of data_term*scale by k_block_sum (the reference's
//                     order is BGL hash order, i.e. unspecified: a fixed strided / pairwise order here).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_project_graph(int V, float2* __restrict__ pos, float* __restrict__ x, float graph_scale, ProjectGeometry geo,
                uint8_t* __restrict__ keep, float2* __restrict__ pos_before) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const float2 u = pos[v];
  if (pos_before) pos_before[v] = u;  // (the positions the layout was built from, kept for its host image: nltgv2_context.hpp layout_pos)
  const float idepth = x[v] * graph_scale;
  float nx, ny, nid;
  if (idepth == 0.0f) {
    const float* M = geo.KRKinv;
    const float h0 = (M[0] * u.x + M[1] * u.y) + M[2] * 1.0f;
    const float h1 = (M[3] * u.x + M[4] * u.y) + M[5] * 1.0f;
    const float h2 = (M[6] * u.x + M[7] * u.y) + M[8] * 1.0f;
    const float inv = 1.0f / h2;
    nx = h0 * inv, ny = h1 * inv, nid = 0.0f;
  } else {
    const float depth = 1.0f / idepth;
    float p0 = geo.Kinv[0] * u.x + geo.Kinv[2], p1 = geo.Kinv[4] * u.y + geo.Kinv[5], p2 = 1.0f;
    p0 *= depth, p1 *= depth, p2 *= depth;
    const float w = geo.q[0], ux = geo.q[1], uy = geo.q[2], uz = geo.q[3];
    float uvx = uy * p2 - uz * p1;
    float uvy = uz * p0 - ux * p2;
    float uvz = ux * p1 - uy * p0;
    uvx += uvx, uvy += uvy, uvz += uvz;
    const float cx = uy * uvz - uz * uvy;
    const float cy = uz * uvx - ux * uvz;
    const float cz = ux * uvy - uy * uvx;
    const float pc0 = ((p0 + w * uvx) + cx) + geo.t[0];
    const float pc1 = ((p1 + w * uvy) + cy) + geo.t[1];
    const float pc2 = ((p2 + w * uvz) + cz) + geo.t[2];
    const float u0 = geo.K[0] * pc0 + geo.K[2] * pc2, u1 = geo.K[4] * pc1 + geo.K[5] * pc2;
    nid = 1.0f / pc2;
    nx = u0 * nid, ny = u1 * nid;
  }
  pos[v] = make_float2(nx, ny);
  x[v] = nid / graph_scale;
  const bool inside = geo.rx <= nx && nx < geo.rx + geo.rw && geo.ry <= ny && ny < geo.ry + geo.rh;
  keep[v] = (inside && !(nid < 0.0f)) ? 1 : 0;
}

// Sum of in[i] * scale over i < n in a FIXED order (so that a CPU restatement reproduces it bit for bit: oracle/photometric_oracle.c
// strided_tree_sum): thread t of one 1024-thread workgroup adds up the elements t, t + 1024, t + 2048, ... sequentially; the 1024
// partial sums are combined pairwise, p[t] += p[t + s] for s = 512, 256, ..., 1.  Used where the reference's own order is
// unspecified -- its vertex loops walk a hash set (flame.cc:328-351, nltgv2...cc:73-85) -- : one lane adding 57 k terms one after
// the other took milliseconds for an order that buys no parity.  *out = sum / divisor.
constexpr int kSumThreads = 1024;
__global__ void __launch_bounds__(kSumThreads)
k_block_sum(const int n, const float* __restrict__ in, const float scale, const float divisor, float* __restrict__ out) {
  __shared__ float p[kSumThreads];
  const int t = threadIdx.x;
  float sum = 0.0f;
  for (int i = t; i < n; i += kSumThreads) sum += in[i] * scale;
  p[t] = sum;
  __syncthreads();
  for (int s = kSumThreads / 2; s > 0; s >>= 1) {
    if (t < s) p[t] += p[t + s];
    __syncthreads();
  }
  if (t == 0) *out = p[0] / divisor;
}

__global__ void __launch_bounds__(256)
k_rescale_apply(int V, float* __restrict__ x, float* __restrict__ xb, float* __restrict__ xp, float* __restrict__ data,
                float graph_scale, const float* __restrict__ new_scale_p) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const float ns = new_scale_p[0];
  x[v] = x[v] * graph_scale / ns;
  xb[v] = xb[v] * graph_scale / ns;
  xp[v] = xp[v] * graph_scale / ns;
  data[v] = data[v] * graph_scale / ns;
}

inline dim3 grid1d(int64_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

// ------------------------------------------------------------------------------------------------
// launch wrappers (all asynchronous on `stream`; return the hipError_t of the launch)
// ------------------------------------------------------------------------------------------------
int launch_fused_step(const FusedArgs& a, const SolverParams& p, int parity, bool write_prev, int unroll,
                      int waves_per_block, hipStream_t stream) {
  if (a.n_slices <= 0) return (int)hipSuccess;
  const int spx = (a.n_slices + 7) / 8;                          // slices per XCD
  const int bpx = (spx + waves_per_block - 1) / waves_per_block;  // blocks per XCD
  const dim3 grid((unsigned)(bpx * 8)), block((unsigned)(64 * waves_per_block));
  const float4* bin = a.bar[parity];
  float4* bout = a.bar[parity ^ 1];
#define FLAME_LAUNCH(UU, WP)                                                                         \
  hipLaunchKernelGGL((k_fused_step<UU, WP>), grid, block, 0, stream, a.n_slices, spx, a.slice_row,   \
                     a.hrec, a.hq, a.vstate, a.vaux, bin, bout, a.vprev, p, a.err)
  if (unroll >= 16) {
    if (write_prev) FLAME_LAUNCH(16, true); else FLAME_LAUNCH(16, false);
  } else if (unroll >= 8) {
    if (write_prev) FLAME_LAUNCH(8, true); else FLAME_LAUNCH(8, false);
  } else {
    if (write_prev) FLAME_LAUNCH(4, true); else FLAME_LAUNCH(4, false);
  }
#undef FLAME_LAUNCH
  return (int)hipGetLastError();
}

int launch_save_prev(const CanonArgs& c, hipStream_t s) {
  if (c.V <= 0) return 0;
  hipLaunchKernelGGL(k_save_prev, grid1d(c.V), dim3(256), 0, s, c.V, c.x, c.w1, c.w2, c.xp, c.w1p, c.w2p);
  return (int)hipGetLastError();
}

int launch_dual(const CanonArgs& c, const SolverParams& p, hipStream_t s) {
  if (c.E <= 0) return 0;
  hipLaunchKernelGGL(k_dual_edge_sweep, grid1d(c.E), dim3(256), 0, s, c.E, c.src, c.dst, c.alpha, c.beta,
                     c.pos, c.xb, c.w1b, c.w2b, c.q1, c.q2, c.q3, p, c.err);
  return (int)hipGetLastError();
}

int launch_primal(const CanonArgs& c, const SolverParams& p, hipStream_t s) {
  if (c.V <= 0) return 0;
  hipLaunchKernelGGL(k_primal_vertex_gather, grid1d(c.V), dim3(256), 0, s, c.V, c.row_ptr, c.half, c.src,
                     c.dst, c.alpha, c.beta, c.pos, c.q1, c.q2, c.q3, c.data, c.weight, c.x, c.w1, c.w2, p);
  return (int)hipGetLastError();
}

int launch_extragradient(const CanonArgs& c, const SolverParams& p, hipStream_t s) {
  if (c.V <= 0) return 0;
  hipLaunchKernelGGL(k_extragradient, grid1d(c.V), dim3(256), 0, s, c.V, c.x, c.w1, c.w2, c.xp, c.w1p,
                     c.w2p, c.xb, c.w1b, c.w2b, p);
  return (int)hipGetLastError();
}

int launch_pack_static(const CanonArgs& c, const FusedArgs& a, hipStream_t s) {
  if (a.n_slots <= 0) return 0;
  hipLaunchKernelGGL(k_pack_static, grid1d(a.n_slots), dim3(256), 0, s, a.n_slots, a.rec_edge, a.rec_nbr,
                     c.src, c.dst, c.alpha, c.pos, a.hrec);
  return (int)hipGetLastError();
}

int launch_pack_state(const CanonArgs& c, const FusedArgs& a, int parity, bool with_static, hipStream_t s) {
  const int n_packed = a.n_slices * 64;
  const int64_t n = std::max<int64_t>(a.n_slots, n_packed);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_pack_all, grid1d(n), dim3(256), 0, s, a.n_slots, n_packed, with_static ? 1 : 0, a.rec_edge, a.rec_nbr, c.src, c.dst,
                     c.alpha, c.beta, c.pos, c.q1, c.q2, c.q3, a.hrec, a.hq, a.perm, a.pdeg, c.x, c.w1, c.w2, c.xb, c.w1b, c.w2b, c.data,
                     c.weight, a.vstate, a.vaux, a.bar[parity], a.bar[parity ^ 1]);
  return (int)hipGetLastError();
}

int launch_unpack_state(const CanonArgs& c, const FusedArgs& a, int parity, bool have_prev, hipStream_t s) {
  const int n_packed = a.n_slices * 64;
  const int n = std::max(n_packed, c.E);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_unpack_all, grid1d(n), dim3(256), 0, s, n_packed, c.E, a.perm, a.vstate, a.bar[parity], a.vprev, have_prev ? 1 : 0,
                     c.x, c.w1, c.w2, c.xb, c.w1b, c.w2b, c.xp, c.w1p, c.w2p, a.edge_src_slot, a.hq, c.q1, c.q2, c.q3);
  return (int)hipGetLastError();
}

int launch_export(const CanonArgs& c, const FusedArgs& a, bool packed_current, float scale, float* dst,
                  hipStream_t s) {
  if (c.V <= 0) return 0;
  if (packed_current) {
    const int n_packed = a.n_slices * 64;
    hipLaunchKernelGGL(k_export_packed, grid1d(n_packed), dim3(256), 0, s, n_packed, a.perm, a.vstate, scale, dst);
  } else {
    hipLaunchKernelGGL(k_export_canonical, grid1d(c.V), dim3(256), 0, s, c.V, c.x, scale, dst);
  }
  return (int)hipGetLastError();
}

int launch_project_graph(const CanonArgs& c, float graph_scale, const ProjectGeometry& geo, uint8_t* keep, float2* pos_before, hipStream_t s) {
  if (c.V <= 0) return 0;
  hipLaunchKernelGGL(k_project_graph, grid1d(c.V), dim3(256), 0, s, c.V, c.pos, c.x, graph_scale, geo, keep, pos_before);
  return (int)hipGetLastError();
}

int launch_rescale(const CanonArgs& c, float graph_scale, float* new_scale_dev, hipStream_t s) {
  if (c.V <= 0) return 0;
  hipLaunchKernelGGL(k_block_sum, dim3(1), dim3(kSumThreads), 0, s, c.V, c.data, graph_scale, (float)c.V, new_scale_dev);  // new_scale = mean
  hipLaunchKernelGGL(k_rescale_apply, grid1d(c.V), dim3(256), 0, s, c.V, c.x, c.xb, c.xp, c.data, graph_scale,
                     new_scale_dev);
  return (int)hipGetLastError();
}

int launch_interpolate_mesh(int T, const int32_t* tris, const float2* vtx, const float* values, float value_scale,
                            const uint8_t* vtx_valid, const uint8_t* tri_valid, unsigned long long* keys, float* img,
                            int* coverage, int rows, int cols, hipStream_t s) {
  const long n = (long)rows * cols;
  if (n <= 0) return 0;
  (void)hipMemsetAsync(keys, 0, sizeof(unsigned long long) * (size_t)n, s);
  (void)hipMemsetAsync(coverage, 0, sizeof(int), s);
  if (T > 0) {
    hipLaunchKernelGGL(k_raster_triangles, dim3((unsigned)T), dim3(64), 0, s, T, tris, vtx, values, value_scale,
                       vtx_valid, tri_valid, keys, rows, cols);
  }
  hipLaunchKernelGGL(k_raster_resolve, grid1d((n + kResolvePerThread - 1) / kResolvePerThread), dim3(256), 0, s, n, keys, img, coverage);
  return (int)hipGetLastError();
}

int launch_photo_residual(const CanonArgs& c, float graph_scale, const PhotoGeometry& geo, const uint8_t* ref,
                          const uint8_t* cmp, int rows, int cols, int step, int border, float* err, hipStream_t s) {
  if (c.V <= 0) return 0;
  hipLaunchKernelGGL(k_photo_residual, grid1d(c.V), dim3(256), 0, s, c.V, c.pos, c.x, graph_scale, geo, ref, cmp,
                     rows, cols, step, border, err);
  return (int)hipGetLastError();
}

int launch_photo_residual_packed(const FusedArgs& a, const PhotoFuse& photo, hipStream_t s) {
  const int n_packed = a.n_slices * 64;
  if (n_packed <= 0 || !photo.err) return 0;
  hipLaunchKernelGGL(k_photo_residual_packed, grid1d(n_packed), dim3(256), 0, s, n_packed, a.perm, a.vstate, photo);
  return (int)hipGetLastError();
}

// The addends of smoothnessCost (cc:51-71: two per edge, alpha*|..| and beta*|..| + beta*|..|) and of dataCost
// (cc:73-85: one per vertex), each exactly as the reference forms it.  The host adds them up in order.
__global__ void __launch_bounds__(256)
k_cost_terms(int E, int V, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
             const float* __restrict__ alpha, const float* __restrict__ beta, const float2* __restrict__ pos,
             const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ w2,
             const float* __restrict__ data, const float* __restrict__ weight, float* __restrict__ terms) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < E) {
    const int i = src[t], j = dst[t];
    const float2 pi = pos[i], pj = pos[j];
    const float dx = pi.x - pj.x, dy = pi.y - pj.y;
    float a = x[i] - x[j] - w1[i] * dx - w2[i] * dy;
    a = (a >= 0) ? a : -a;
    float b = w1[i] - w1[j];
    b = (b >= 0) ? b : -b;
    float c = w2[i] - w2[j];
    c = (c >= 0) ? c : -c;
    terms[2 * (size_t)t] = alpha[t] * a;
    terms[2 * (size_t)t + 1] = beta[t] * b + beta[t] * c;
  } else if (t < E + V) {
    const int v = t - E;
    float diff = (x[v] - data[v]) * weight[v];
    diff = (diff > 0) ? diff : -diff;
    terms[2 * (size_t)E + v] = diff;
  }
}

// both cost sums on the device (FLAME_NLTGV2_OPT_COST_SUM = 1): out2[0] = sum of the 2E smoothness addends, out2[1] = of the V data addends
int launch_cost_sums(const CanonArgs& c, const float* terms, float* out2, hipStream_t s) {
  hipLaunchKernelGGL(k_block_sum, dim3(1), dim3(kSumThreads), 0, s, 2 * c.E, terms, 1.0f, 1.0f, out2);
  hipLaunchKernelGGL(k_block_sum, dim3(1), dim3(kSumThreads), 0, s, c.V, terms + 2 * (size_t)c.E, 1.0f, 1.0f, out2 + 1);
  return (int)hipGetLastError();
}

int launch_cost_terms(const CanonArgs& c, float* terms, hipStream_t s) {
  const int n = c.E + c.V;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_cost_terms, grid1d(n), dim3(256), 0, s, c.E, c.V, c.src, c.dst, c.alpha, c.beta, c.pos, c.x, c.w1,
                     c.w2, c.data, c.weight, terms);
  return (int)hipGetLastError();
}



// Loads this translation unit's code object (the runtime does that at the first use of one of its kernels: several milliseconds that
// flame_nltgv2_create takes on itself so that the first frame does not).
void warm_module_kernels() {
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, (const void*)k_save_prev) != hipSuccess) (void)hipGetLastError();
}

}  // namespace flame_hip
