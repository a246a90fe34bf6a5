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
constexpr int kTvS = 8;
constexpr unsigned kTvOwnerBit = 1u << 16, kTvValidBit = 1u << 17;

__device__ __forceinline__ float dpp_shr1(float v) {  // lane l <- lane l-1; lane 0 keeps its value
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}

// LDS_STATIC: the per-slot constants (neighbour offset, alpha, dx, dy, beta: 40 dwords per lane) live in LDS
// (10 KB per wave) instead of registers: <= 128 VGPRs, four waves per SIMD (16 per CU, which is also all of the
// CU's 160 KB of LDS): bigger resident batches and 4K-sized single graphs.  The duals stay in registers.
template <bool LDS_STATIC>
__global__ void __launch_bounds__(256, LDS_STATIC ? 4 : 2)
k_persistent_tv(const int wave_begin, const int n_waves, const int waves_per_xcd, const int32_t* __restrict__ tv_slot,
                const int32_t* __restrict__ tv_vid, const uint32_t* __restrict__ tv_meta,
                const uint32_t* __restrict__ tv_wave, const int4* hrec, const float4* hq, const float4* vstate,
                float4* hq_out, float4* vstate_out, const float2* vaux, const float4* bar_in, float4* bar_out,
                float4* vprev, void* xbuf, const int rec_bytes, const int dual_arg, const unsigned tag0, const int n_iters,
                const unsigned max_spins_arg, const int presleep, const SolverParams p, int* __restrict__ err,
                int* __restrict__ abort_flag, const int32_t* __restrict__ perm, const RunTail* __restrict__ tail) {
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
  int r_nbr[LDS_STATIC ? 1 : kTvS];
  float r_alpha[LDS_STATIC ? 1 : kTvS], r_dx[LDS_STATIC ? 1 : kTvS], r_dy[LDS_STATIC ? 1 : kTvS], r_beta[LDS_STATIC ? 1 : kTvS];
#define TV_I(field, k) s_base[((field) * kTvS + (k)) * 64]
#define NBR(k) (*(LDS_STATIC ? &TV_I(0, k) : &r_nbr[LDS_STATIC ? 0 : (k)]))
#define ALPHA(k) (*(LDS_STATIC ? (float*)&TV_I(1, k) : &r_alpha[LDS_STATIC ? 0 : (k)]))
#define DX(k) (*(LDS_STATIC ? (float*)&TV_I(2, k) : &r_dx[LDS_STATIC ? 0 : (k)]))
#define DY(k) (*(LDS_STATIC ? (float*)&TV_I(3, k) : &r_dy[LDS_STATIC ? 0 : (k)]))
#define BETA(k) (*(LDS_STATIC ? (float*)&TV_I(4, k) : &r_beta[LDS_STATIC ? 0 : (k)]))
  float q1[kTvS], q2[kTvS], q3[kTvS];
#pragma unroll
  for (int k = 0; k < kTvS; ++k) {
    const int sl = tv_slot[((size_t)w * kTvS + k) * 64 + lane];
    int4 r = make_int4(0, 0, 0, 0);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sl >= 0) {
      r = hrec[sl];
      q = hq[sl];
    }
    NBR(k) = (int)(((unsigned)r.x & 0x80000000u) | (((unsigned)r.x & 0x07ffffffu) << 4));
    ALPHA(k) = __int_as_float(r.y), DX(k) = __int_as_float(r.z), DY(k) = __int_as_float(r.w);
    q1[k] = q.x, q2[k] = q.y, q3[k] = q.z, BETA(k) = q.w;
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
  float x = st.x, w1 = st.y, w2 = st.z;  // invariant: every lane of a chain holds the vertex state
  float xb = bs.x, w1b = bs.y, w2b = bs.z;
  float x_prev = x, w1_prev = w1, w2_prev = w2;
  bool ok = true;
  bool timed_out = false, torn = false;
  const int ps = presleep;

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(xbuf);
  const int my_off = pv << 4;
  const int S = rec_bytes, par = 2 * rec_bytes;
  const unsigned all_mask = (1u << nslots) - 1u;

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
    o.x = __float_as_int(xb), o.y = __float_as_int(w1b), o.z = __float_as_int(w2b), o.w = (int)tag0;
    const int so = (tag0 & 1u) ? par : 0;
    __builtin_amdgcn_raw_buffer_store_b128(o, rx, my_off, so, kAuxSc1);
    if (dual) __builtin_amdgcn_raw_buffer_store_b128(o, rx, my_off + S, so, 0);
  }

  for (int it = 0; it < n_iters && !timed_out; ++it) {
    const unsigned s = tag0 + (unsigned)it;
    const int so_in = (s & 1u) ? par : 0;
    // ---- wait for all neighbours' bar(s): one round of loads in flight, only pending slots re-polled
    v4i_t g[kTvS];
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
      __builtin_amdgcn_s_sleep(1);
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
          if ((verify & 2) && it == 2 && w == wave_begin && lane == 0 && k == 0) g2.x ^= 0x00400000;  // test hook
          bad = bad || g2.x != g[k].x || g2.y != g[k].y || g2.z != g[k].z || g2.w != g[k].w;
        }
      }
      if (__any(bad)) {
        torn = true;
        break;
      }
    }

    __builtin_amdgcn_s_setprio(3);  // from the records' arrival to the publish this wave goes before the polling ones (k_persistent_pv)
    // ---- phase A, every lane at once: dual update of each slot's private q copy (cc:99-110) and the three
    // step-scaled values its primal scatter needs (cc:126-141).  They overwrite the neighbour record of the slot
    // (dead from here on), so the ordered accumulation below costs no registers.
#pragma unroll
    for (int k = 0; k < kTvS; ++k) {
      const bool is_target = NBR(k) < 0;
      const float alpha_k = ALPHA(k), beta_k = BETA(k), dx_k = DX(k), dy_k = DY(k);
      const float nxb = __int_as_float(g[k].x), nw1b = __int_as_float(g[k].y), nw2b = __int_as_float(g[k].z);
      const float xbi = is_target ? nxb : xb, xbj = is_target ? xb : nxb;
      const float w1bi = is_target ? nw1b : w1b, w1bj = is_target ? w1b : nw1b;
      const float w2bi = is_target ? nw2b : w2b, w2bj = is_target ? w2b : nw2b;
      bool okq = true;
      const EdgeOut e = edge_dual(p, alpha_k, beta_k, dx_k, dy_k, q1[k], q2[k], q3[k], xbi, w1bi, w2bi,
                                  xbj, w1bj, w2bj, okq);
      g[k].x = __float_as_int(e.q1 * p.step_x * alpha_k);
      g[k].y = __float_as_int(e.q2 * p.step_x * beta_k);
      g[k].z = __float_as_int(e.q3 * p.step_x * beta_k);
      if (k < nslots) {
        q1[k] = e.q1, q2[k] = e.q2, q3[k] = e.q3;
        ok = ok && okq;
      }
    }
    // ---- phase B, ordered accumulation in ascending edge id: pass 0 for the first lane of every vertex, pass c
    // for the c-th continuation lane of vertices with more than eight edges (it first takes over the running sums
    // of the lane before it).  A Delaunay wave contains such a vertex more often than not; with the duals already
    // done a further pass is 13 instead of ~60 instructions per slot.
    float X = x, W1 = w1, W2 = w2;
    for (int pass = 0; pass < passes; ++pass) {
      if (pass > 0) {
        const float Xs = dpp_shr1(X), W1s = dpp_shr1(W1), W2s = dpp_shr1(W2);
        if (cidx == pass) X = Xs, W1 = W1s, W2 = W2s;
      }
      const bool live = (cidx == pass);
#pragma unroll
      for (int k = 0; k < kTvS; ++k) {
        const bool act = live && (k < nslots);
        const bool is_target = NBR(k) < 0;
        const float t1 = __int_as_float(g[k].x), t2 = __int_as_float(g[k].y), t3 = __int_as_float(g[k].z);
        float nx, nw1, nw2;
        if (is_target) {
          nx = X + t1;
          nw1 = W1 + t2;
          nw2 = W2 + t3;
        } else {
          nx = X - t1;
          nw1 = W1 + t1 * DX(k);
          nw2 = W2 + t1 * DY(k);
          nw1 = nw1 - t2;
          nw2 = nw2 - t3;
        }
        if (act) X = nx, W1 = nw1, W2 = nw2;
      }
    }
    // ---- vertex update at the owner lane ----------------------------------------------------------
    const float xn = prox_l1(p.x_min, p.x_max, p.step_x, lam_w, X, data);
    float nb = xn + p.theta * (xn - x);
    nb = (nb < p.x_min) ? p.x_min : nb;
    nb = (nb > p.x_max) ? p.x_max : nb;
    const float w1bn = W1 + p.theta * (W1 - w1);
    const float w2bn = W2 + p.theta * (W2 - w2);
    if (is_owner) {
      v4i_t o;
      o.x = __float_as_int(nb), o.y = __float_as_int(w1bn), o.z = __float_as_int(w2bn), o.w = (int)(s + 1u);
      const int so = ((s + 1u) & 1u) ? par : 0;
      __builtin_amdgcn_raw_buffer_store_b128(o, rx, my_off, so, kAuxSc1);
      if (dual) __builtin_amdgcn_raw_buffer_store_b128(o, rx, my_off + S, so, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    x_prev = x, w1_prev = w1, w2_prev = w2;
    if (has_chain) {  // wave-uniform: hand the owner's new state back to every lane of its chain
      x = __shfl(xn, owner_lane, 64);
      w1 = __shfl(W1, owner_lane, 64);
      w2 = __shfl(W2, owner_lane, 64);
      xb = __shfl(nb, owner_lane, 64);
      w1b = __shfl(w1bn, owner_lane, 64);
      w2b = __shfl(w2bn, owner_lane, 64);
    } else {
      x = xn, w1 = W1, w2 = W2, xb = nb, w1b = w1bn, w2b = w2bn;
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
    vstate_out[pv] = make_float4(x, w1, w2, data);
    bar_out[pv] = make_float4(xb, w1b, w2b, 0.0f);
    vprev[pv] = make_float4(x_prev, w1_prev, w2_prev, 0.0f);
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
    if (k < nslots) hq_out[tv_slot[((size_t)w * kTvS + k) * 64 + lane]] = make_float4(q1[k], q2[k], q3[k], BETA(k));
  }
  if (!ok) atomicOr(err, 1);
#undef NBR
#undef ALPHA
#undef DX
#undef DY
#undef BETA
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

}  // namespace flame_hip
