// nltgv2_persistent_tv.hip -- the vertex-per-lane persistent kernel of the NLTGV2-L1 solver (throughput form: batches of frames);
// launched by launch_persistent_run (nltgv2_persistent.hip).  Compiled with -ffp-contract=off; arithmetic and its citations:
// nltgv2_device.hpp.
#include "nltgv2_device.hpp"

namespace flame_hip {

namespace {

// ------------------------------------------------------------------------------------------------
// Persistent run, throughput form: same dataflow protocol as k_persistent_he (tagged 16-byte bar
// records, two parity buffers, bounded waits), but one VERTEX per lane with up to 8 half-edge slots
// held in registers -- ~5x fewer instructions per half-edge than the lane-per-half-edge form, at the
// price of a longer serial chain per wave.  It is the better choice when many waves share a CU
// (batches of frames, 1080p graphs), where the lane-per-half-edge kernel becomes issue bound.
//
// A vertex of degree > 8 occupies ceil(deg/8) ADJACENT lanes of one wave; pass p processes the lanes
// with chain index p, which first take over the running sums of lane-1 (DPP wave_shr:1), so the
// accumulation still follows ascending edge id exactly.  The last lane of a chain owns the vertex:
// it applies proxL1 / extragradient, publishes the record and hands the state back to its chain.
// ------------------------------------------------------------------------------------------------
typedef float v2f_t __attribute__((ext_vector_type(2)));
constexpr int kTvS = 8;
constexpr unsigned kTvOwnerBit = 1u << 16, kTvValidBit = 1u << 17;

__device__ __forceinline__ float dpp_shr1(float v) {  // lane l <- lane l-1; lane 0 keeps its value
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}

// The per-slot constants (40 dwords per lane) live in LDS, 10 KB per wave: <= 128 VGPRs, four waves per SIMD (16 per CU, which is also
// all of the CU's 160 KB of LDS): 30 frames of 640x480 resident, 4K-sized single graphs.  The duals stay in registers.
//
// Per slot: neighbour offset | role, as = +-alpha (source +, target -), bs = +-beta, dx, dy.  With the signs folded into the constants
// both roles run ONE instruction sequence (k_persistent_pv2's, same IEEE results: a - b == -(b - a), (-a) * b == -(a * b)), and the pair
// (w1, w2) / (q2, q3) is packed arithmetic (v_pk_mul_f32 / v_pk_add_f32): ~36 instead of ~45 vector instructions per slot for the duals
// and 3 instead of 12 per slot and pass for the ordered accumulation -- at 16 waves per CU this kernel is bound by VALU issue
// (profiles/r04_counters.json: SQ_ACTIVE_INST_VALU = 0.24 of every wave's cycles, four waves per SIMD).  An unused slot holds
// (as, bs, dx, dy) = (+0, +0, -0, -0) and reads a zero record: every contribution of it then is exactly -0.0, and x + (-0.0) == x
// for every x -- no predicate in the accumulation.
template <bool LDS_STATIC>
__global__ void __launch_bounds__(256, 4)
k_persistent_tv(const int wave_begin, const int n_waves, const int waves_per_xcd, const int32_t* __restrict__ tv_slot,
                const int32_t* __restrict__ tv_vid, const uint32_t* __restrict__ tv_meta,
                const uint32_t* __restrict__ tv_wave, const int4* hrec, const float4* hq, const float4* vstate,
                float4* hq_out, float4* vstate_out, const float2* vaux, const float4* bar_in, float4* bar_out,
                float4* vprev, void* xbuf, const int rec_bytes, const int dual_arg, const unsigned tag0, const int n_iters,
                const unsigned max_spins_arg, const int presleep, const SolverParams p, int* __restrict__ err,
                int* __restrict__ abort_flag, const int32_t* __restrict__ perm, const RunTail* __restrict__ tail) {
  static_assert(LDS_STATIC, "the register instance was retired in round 4");
  const unsigned max_spins = max_spins_arg & 0x7fffffffu;
  const int dual = dual_arg & 1, verify = dual_arg >> 1;  // as in k_persistent_he
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int idx = (b >> 3) * wpb + (threadIdx.x >> 6);
  if (idx >= waves_per_xcd) return;
  if (xcd * waves_per_xcd + idx >= n_waves) return;
  const int w = wave_begin + xcd * waves_per_xcd + idx;  // this launch covers waves [wave_begin, +n_waves)

  const unsigned meta = tv_meta[(size_t)w * 64 + lane];
  const int pv = tv_vid[(size_t)w * 64 + lane];
  const unsigned winfo = (unsigned)__builtin_amdgcn_readfirstlane((int)tv_wave[w]);
  const int passes = (int)(winfo & 0xffu);
  const bool has_chain = (winfo & 0x100u) != 0u;
  const int nslots = (int)(meta & 15u);
  const int cidx = (int)((meta >> 4) & 63u);
  const int owner_lane = (int)((meta >> 10) & 63u);
  const bool is_owner = (meta & kTvOwnerBit) != 0u;
  const bool valid = (meta & kTvValidBit) != 0u;

  extern __shared__ __attribute__((aligned(16))) int tv_smem[];
  int* const s_base = tv_smem + (threadIdx.x >> 6) * (5 * kTvS * 64) + lane;  // [field][slot][lane] per wave
#define TV_I(field, k) s_base[((field) * kTvS + (k)) * 64]
#define NBR(k) TV_I(0, k)
#define AS(k) (*(float*)&TV_I(1, k))
#define BS(k) (*(float*)&TV_I(2, k))
#define DX(k) (*(float*)&TV_I(3, k))
#define DY(k) (*(float*)&TV_I(4, k))
  float q1[kTvS];
  v2f_t q23[kTvS];
#pragma unroll
  for (int k = 0; k < kTvS; ++k) {
    const int sl = tv_slot[((size_t)w * kTvS + k) * 64 + lane];
    int4 r = make_int4(0, 0, 0, 0);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sl >= 0) {
      r = hrec[sl];
      q = hq[sl];
    }
    const bool tgt = r.x < 0;
    const float alpha = __int_as_float(r.y);
    NBR(k) = (int)(((unsigned)r.x & 0x80000000u) | (((unsigned)r.x & 0x07ffffffu) << 4));
    AS(k) = tgt ? -alpha : alpha, BS(k) = tgt ? -q.w : q.w;
    DX(k) = sl >= 0 ? __int_as_float(r.z) : -0.0f, DY(k) = sl >= 0 ? __int_as_float(r.w) : -0.0f;
    q1[k] = q.x, q23[k] = v2f_t{q.y, q.z};
  }
  // Neighbours that live in this very wave (a wave holds 64 vertices along the Morton curve: ~85 % of the half-edges) are not fetched
  // through L2 at all: their bars are the registers of another lane, read with ds_bpermute at the start of every step -- always
  // current, nothing to wait for.  NBR(k) of such a slot: role | lane << 4 | 1.  (Every lane of a chain holds its vertex' bars.)
  unsigned sw_mask = 0u;
  {
    int nv[kTvS];
#pragma unroll
    for (int k = 0; k < kTvS; ++k) nv[k] = (k < (int)(meta & 15u)) ? (NBR(k) & 0x7fffffff) >> 4 : -2;
    const int mine = (meta & kTvValidBit) ? pv : -1;
    for (int l = 0; l < 64; ++l) {
      const int vid_l = __builtin_amdgcn_readlane(mine, l);
#pragma unroll
      for (int k = 0; k < kTvS; ++k) {
        if (nv[k] == vid_l) {
          NBR(k) = (int)(((unsigned)NBR(k) & 0x80000000u) | ((unsigned)l << 4) | 1u);
          sw_mask |= 1u << k;
          nv[k] = -2;
        }
      }
    }
  }

  float4 st = make_float4(0.f, 0.f, 0.f, 0.f), bs = make_float4(0.f, 0.f, 0.f, 0.f);
  float2 aux = make_float2(0.f, 0.f);
  if (valid) {
    st = vstate[pv];
    aux = vaux[pv];
    bs = bar_in[pv];
  }
  const float data = st.w;
  const float lam_w = p.data_factor * aux.x;
  float x = st.x;  // invariant: every lane of a chain holds the vertex state
  v2f_t w12 = {st.y, st.z};
  float xb = bs.x;
  v2f_t wb12 = {bs.y, bs.z};
  float x_prev = x;
  v2f_t w_prev = w12;
  bool ok = true;
  bool timed_out = false, torn = false;
  const int ps = presleep;

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(xbuf);
  const int my_off = pv << 4;
  const int S = rec_bytes, par = 2 * rec_bytes;
  const unsigned all_mask = ((1u << nslots) - 1u) & ~sw_mask;  // the slots that are fetched

  if (dual) {  // one-time XCC exchange: which neighbours run on my XCD?  (bit 30 of nbr[k] := same XCD)
    const unsigned my_xcc = read_xcc_id();
    const unsigned want = (tag0 & 0x0fffffffu) << 4;
    if (is_owner) __builtin_amdgcn_raw_buffer_store_b32((int)(want | my_xcc), rx, 4 * S + (pv << 2), 0, kAuxSc1);
    unsigned pending = all_mask;
    unsigned spins = 0;
    unsigned got[kTvS];
    for (;;) {
#pragma unroll
      for (int k = 0; k < kTvS; ++k) {
        if ((pending >> k) & 1u) {
          int o = 4 * S + ((NBR(k) & 0x7fffffff) >> 2);
          asm volatile("" : "+v"(o)::"memory");
          got[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rx, o, 0, kAuxSc1);
        }
      }
#pragma unroll
      for (int k = 0; k < kTvS; ++k) {
        if (((pending >> k) & 1u) && (got[k] & ~15u) == want) {
          pending &= ~(1u << k);
          if ((got[k] & 15u) == my_xcc) NBR(k) += S;  // poll the local copy (S < 2^31: role bit untouched)
        }
      }
      if (!__any(pending != 0u)) break;
      if (++spins > max_spins) {
        timed_out = true;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }

  const bool mute = (max_spins_arg >> 31) != 0u && w == wave_begin;  // test hook, see k_persistent_he
  if (is_owner && !timed_out && !mute) {
    v4i_t o;
    o.x = __float_as_int(xb), o.y = __float_as_int(wb12.x), o.z = __float_as_int(wb12.y), o.w = (int)tag0;
    const int so = (tag0 & 1u) ? par : 0;
    __builtin_amdgcn_raw_buffer_store_b128(o, rx, my_off, so, kAuxSc1);
    if (dual) __builtin_amdgcn_raw_buffer_store_b128(o, rx, my_off + S, so, 0);
  }

  // the neighbour records of a step, then its contributions.  An unused slot is never fetched: it reads zeros in the first step and its
  // own contributions (-0.0) afterwards -- finite either way, which is all its arithmetic needs to stay at -0.0.
  v4i_t g[kTvS];
#pragma unroll
  for (int k = 0; k < kTvS; ++k) g[k] = v4i_t{0, 0, 0, 0};
  for (int it = 0; it < n_iters && !timed_out; ++it) {
    const unsigned s = tag0 + (unsigned)it;
    const int so_in = (s & 1u) ? par : 0;
    {  // same-wave neighbours: all lane indices first, then all 24 permutes back to back, one wait.  Every lane takes what it gets -- a
       // fetching slot overwrites it below, an unused slot only needs something finite.
      int from[kTvS];
#pragma unroll
      for (int k = 0; k < kTvS; ++k) from[k] = (NBR(k) >> 2) & 0xfc;  // lane * 4
#pragma unroll
      for (int k = 0; k < kTvS; ++k) {
        g[k].x = __builtin_amdgcn_ds_bpermute(from[k], __float_as_int(xb));
        g[k].y = __builtin_amdgcn_ds_bpermute(from[k], __float_as_int(wb12.x));
        g[k].z = __builtin_amdgcn_ds_bpermute(from[k], __float_as_int(wb12.y));
      }
    }
    unsigned pending = all_mask;
    unsigned spins = 0;
    for (int z = 0; z < ps; ++z) __builtin_amdgcn_s_sleep(1);
    for (;;) {
#pragma unroll
      for (int k = 0; k < kTvS; ++k) {
        if ((pending >> k) & 1u) {
          int o = NBR(k) & 0x7fffffff;
          asm volatile("" : "+v"(o)::"memory");  // opaque: re-issue on every spin
          g[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, o, so_in, kAuxSc1);
        }
      }
#pragma unroll
      for (int k = 0; k < kTvS; ++k) {
        if (((pending >> k) & 1u) && (unsigned)g[k].w == s) pending &= ~(1u << k);
      }
      if (!__any(pending != 0u)) break;
      ++spins;
      if ((spins & 63u) == 0u) {
        const int ab = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ab != 0 || spins > max_spins) {
          timed_out = true;
          break;
        }
      }
#ifndef TV_SPIN_SLEEP
#define TV_SPIN_SLEEP 1
#endif
      __builtin_amdgcn_s_sleep(TV_SPIN_SLEEP);
    }
    if (timed_out) break;
    if (verify) {  // a record is final once its tag is visible: a second read must return the same 16 bytes
      bool bad = false;
#pragma unroll
      for (int k = 0; k < kTvS; ++k) {
        if ((all_mask >> k) & 1u) {
          int o = NBR(k) & 0x7fffffff;
          asm volatile("" : "+v"(o)::"memory");
          v4i_t g2 = __builtin_amdgcn_raw_buffer_load_b128(rx, o, so_in, kAuxSc1);
          if ((verify & 2) && it == 2 && w == wave_begin && k == __ffs((int)all_mask) - 1) g2.x ^= 0x00400000;  // test hook: every lane's first fetched slot
          bad = bad || g2.x != g[k].x || g2.y != g[k].y || g2.z != g[k].z || g2.w != g[k].w;
        }
      }
      if (__any(bad)) {
        torn = true;
        break;
      }
    }

    __builtin_amdgcn_s_setprio(3);  // from the records' arrival to the publish this wave goes before the polling ones (k_persistent_pv)
    // ---- phase A, every lane at once: dual update of each slot's private q copy (cc:99-110) and its three contributions to the
    // primal scatter (cc:126-141), signs folded in: X += cx, W += A, W += B reproduces
    //   source: x - t1,  (w + t1 * d) - (t2, t3)        target: x + t1,  w + (t2, t3)     with t1 = q1 step_x alpha, (t2, t3) = q23 step_x beta
    // (A, B) overwrite the neighbour record of the slot, dead by then: the accumulation costs eight registers (cx), not forty.
    float cx[kTvS];
#pragma unroll
    for (int k = 0; k < kTvS; ++k) {
      const bool is_target = NBR(k) < 0;
      const float as = AS(k), bs_k = BS(k);
      const v2f_t d = {DX(k), DY(k)};
      const v2f_t nbw = {__int_as_float(g[k].y), __int_as_float(g[k].z)};
      const float d0 = xb - __int_as_float(g[k].x);
      const v2f_t d12 = wb12 - nbw;
      const v2f_t wbi = is_target ? nbw : wb12;
      float K1 = as * d0;
      const float alpha = __builtin_fabsf(as);
      const v2f_t P12 = {alpha * d.x, alpha * d.y};
      const v2f_t m12 = P12 * wbi;
      K1 -= m12.x;
      K1 -= m12.y;
      const v2f_t K23 = bs_k * d12;
      const float q1r = q1[k] + p.step_q * K1;
      const v2f_t q23r = q23[k] + p.step_q * K23;
      ok = ok && (__builtin_fabsf(q1r) <= 3.402823466e+38f) && (__builtin_fabsf(q23r.x) <= 3.402823466e+38f) &&
           (__builtin_fabsf(q23r.y) <= 3.402823466e+38f);  // (an unused slot: zeros, finite)
      q1[k] = __builtin_fminf(__builtin_fmaxf(q1r, -1.0f), 1.0f);
      q23[k].x = __builtin_fminf(__builtin_fmaxf(q23r.x, -1.0f), 1.0f);
      q23[k].y = __builtin_fminf(__builtin_fmaxf(q23r.y, -1.0f), 1.0f);
      const float u1 = q1[k] * p.step_x;
      const v2f_t u23 = q23[k] * p.step_x;
      cx[k] = u1 * -as;                   // source: -t1, target: +t1
      const v2f_t v = u23 * -bs_k;        // source: -(t2, t3), target: +(t2, t3)
      const v2f_t a_src = cx[k] * -d;     // source: t1 * (dx, dy)
      const v2f_t Ak = is_target ? v : a_src, Bk = is_target ? v2f_t{-0.0f, -0.0f} : v;
      g[k] = v4i_t{__float_as_int(Ak.x), __float_as_int(Ak.y), __float_as_int(Bk.x), __float_as_int(Bk.y)};
    }
    // ---- phase B, ordered accumulation in ascending edge id: pass 0 for the first lane of every vertex, pass c
    // for the c-th continuation lane of vertices with more than eight edges (it first takes over the running sums
    // of the lane before it).  A Delaunay wave contains such a vertex more often than not; every lane accumulates in
    // exactly one pass.
    float X = x;
    v2f_t Wv = w12;
    for (int pass = 0; pass < passes; ++pass) {
      if (pass > 0) {
        const float Xs = dpp_shr1(X), W1s = dpp_shr1(Wv.x), W2s = dpp_shr1(Wv.y);
        if (cidx == pass) X = Xs, Wv = v2f_t{W1s, W2s};
      }
      if (cidx == pass) {
#pragma unroll
        for (int k = 0; k < kTvS; ++k) {
          X = X + cx[k];
          Wv = Wv + v2f_t{__int_as_float(g[k].x), __int_as_float(g[k].y)};
          Wv = Wv + v2f_t{__int_as_float(g[k].z), __int_as_float(g[k].w)};
        }
      }
    }
    const float W1 = Wv.x, W2 = Wv.y;
    // ---- vertex update at the owner lane ----------------------------------------------------------
    const float xn = prox_l1(p.x_min, p.x_max, p.step_x, lam_w, X, data);
    float nb = xn + p.theta * (xn - x);
    nb = (nb < p.x_min) ? p.x_min : nb;
    nb = (nb > p.x_max) ? p.x_max : nb;
    const float w1bn = W1 + p.theta * (W1 - w12.x);
    const float w2bn = W2 + p.theta * (W2 - w12.y);
    if (is_owner) {
      v4i_t o;
      o.x = __float_as_int(nb), o.y = __float_as_int(w1bn), o.z = __float_as_int(w2bn), o.w = (int)(s + 1u);
      const int so = ((s + 1u) & 1u) ? par : 0;
      __builtin_amdgcn_raw_buffer_store_b128(o, rx, my_off, so, kAuxSc1);
      if (dual) __builtin_amdgcn_raw_buffer_store_b128(o, rx, my_off + S, so, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    x_prev = x, w_prev = w12;
    if (has_chain) {  // wave-uniform: hand the owner's new state back to every lane of its chain
      x = __shfl(xn, owner_lane, 64);
      w12 = v2f_t{__shfl(W1, owner_lane, 64), __shfl(W2, owner_lane, 64)};
      xb = __shfl(nb, owner_lane, 64);
      wb12 = v2f_t{__shfl(w1bn, owner_lane, 64), __shfl(w2bn, owner_lane, 64)};
    } else {
      x = xn, w12 = Wv, xb = nb, wb12 = v2f_t{w1bn, w2bn};
    }
  }

  if (timed_out || torn) {
    if (lane == 0) {
      __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicOr(err, torn ? 4 : 2);
    }
    return;
  }
  if (is_owner) {  // into the other copies of the state arrays, see k_persistent_he
    vstate_out[pv] = make_float4(x, w12.x, w12.y, data);
    bar_out[pv] = make_float4(xb, wb12.x, wb12.y, 0.0f);
    vprev[pv] = make_float4(x_prev, w_prev.x, w_prev.y, 0.0f);
    float* const export_out = tail->export_out;
    float* const photo_err = tail->photo.err;
    if (export_out || photo_err) {
      const int o = perm[pv];
      if (o >= 0 && export_out) export_out[o] = x * tail->export_scale;
      if (o >= 0 && photo_err) {
        const PhotoFuse& photo = tail->photo;
        photo_err[o] = photo_residual_at(photo.pos[o], x * photo.graph_scale, photo.geo, photo.ref, photo.cmp, photo.rows,
                                         photo.cols, photo.step, photo.border);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kTvS; ++k) {
    if (k < nslots) hq_out[tv_slot[((size_t)w * kTvS + k) * 64 + lane]] = make_float4(q1[k], q23[k].x, q23[k].y, __builtin_fabsf(BS(k)));
  }
  if (!ok) atomicOr(err, 1);
#undef NBR
#undef AS
#undef BS
#undef DX
#undef DY
#undef TV_I
}

}  // namespace

// The kernel instance and its dynamic LDS bytes per workgroup.  Since round 4 only the instance with the per-slot constants in LDS is
// built (the register instance, LDS_STATIC = false, ran 11-15 frame batches 9 % faster per wave at half the residency; those batches
// now run as two groups of k_persistent_pv2, 9-18 % faster still: profiles/r04_large_batches.txt).
const void* persistent_tv_kernel(bool static_in_lds, int waves_per_block, unsigned* lds_bytes) {
  (void)static_in_lds;
  *lds_bytes = (unsigned)(waves_per_block * 5 * kTvS * 64 * sizeof(int));
  return (const void*)k_persistent_tv<true>;
}

void warm_module_persistent_tv() {
  unsigned lds = 0;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, persistent_tv_kernel(true, 4, &lds)) != hipSuccess) (void)hipGetLastError();
}
}  // namespace flame_hip
