// nltgv2_persistent_pv2.hip -- k_persistent_pv2: the patch-per-wave persistent kernel with TWO half-edges per lane (layout (E2)
// of nltgv2_pack.hpp).  Same protocol, same arithmetic, same order of every vertex's accumulation as k_persistent_pv
// (nltgv2_persistent.hip, which documents both); a wave owns a patch of ~18 vertices instead of ~9, so a graph needs half as many
// waves.  The period of the lock-step network grows with the waves a CU holds (docs/DESIGN_r3.md section 4, "What the period depends
// on"): this form is for graphs that fill the chip in the one-half-edge-per-lane form (a 1920x1080 frame: 25 waves per CU there,
// 12.6 here).  Against k_persistent_pv it has no cycle probe, no placed records and takes no vertex of more than 32 edges (the planner
// keeps such graphs on the other forms); the record verification (FLAME_NLTGV2_OPT_VERIFY_RECORDS) is a second instance.
#include "nltgv2_device.hpp"

namespace flame_hip {

namespace {

constexpr unsigned kWgActiveBit = 1u << 25, kWgValidBit = 1u << 26, kWgPublishBit = 1u << 27, kWgHeadBit = 1u << 28;
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_wave_sync2() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned read_hw_id2() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}
__device__ __forceinline__ void report_expired2(int* err, int which, int wg, int it, unsigned long long pend, int frid, unsigned seen,
                                                unsigned want) {
  if (atomicCAS(&err[1], 0, which) == 0) {
    err[2] = wg, err[3] = it, err[4] = (int)(unsigned)pend, err[5] = (int)(unsigned)(pend >> 32), err[6] = frid;
    err[7] = (int)seen, err[8] = (int)want, err[9] = (int)read_xcc_id(), err[10] = (int)read_hw_id2();
  }
}

// The per-half-edge constants of one slot (see k_persistent_pv: the role selects folded into signed constants)
struct SlotConst {
  bool active, is_target;
  float as, bs, ac, nbeta, beta;
  v2f_t P12, C2;
};

// OPEN: the open run's instance (nltgv2_persistent.hip, k_persistent_pv: the same words of the error block, the same margin and interval)
template <bool VERIFY, bool OPEN = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(92)))
k_persistent_pv2(const int wg_begin, const int n_wgs, const int wgs_per_xcd, const int lcap, const int32_t* __restrict__ wg_slot,
                 const int32_t* __restrict__ wg_vid, const uint32_t* __restrict__ wg_meta, const int32_t* __restrict__ wg_nbr,
                 const int32_t* __restrict__ wg_fetch, const int32_t* __restrict__ wg_info, const int4* hrec, const float4* hq,
                 const float4* vstate, float4* hq_out, float4* vstate_out, const float2* vaux, const float4* bar_in, float4* bar_out,
                 float4* vprev, void* xbuf, const int rec_bytes, const int dual_arg, const unsigned tag0, const int n_iters,
                 const unsigned max_spins_arg, const int poll_gap_arg, const SolverParams p, int* __restrict__ err,
                 int* __restrict__ abort_flag, const int32_t* __restrict__ perm, const RunTail* __restrict__ tail) {
  extern __shared__ float4 lds[];
  constexpr int T = 64;
  const unsigned max_spins = max_spins_arg & 0x7fffffffu;
  const int dual = dual_arg & 1, verify = VERIFY ? dual_arg >> 1 : 0;  // (bits 1..: record verification and its test hook, as in k_persistent_pv)
  const int poll_gap = poll_gap_arg & 255, pv_presleep = (poll_gap_arg >> 8) & 255;
  const int lane = (int)threadIdx.x;
  const int b = blockIdx.x;
  const int xcd = b & 7, idx = b >> 3;
  if (idx >= wgs_per_xcd) return;
  if (xcd * wgs_per_xcd + idx >= n_wgs) return;
  const int wg = wg_begin + xcd * wgs_per_xcd + idx;
  const int rid_base = wg_info[4 * wg], n_fetch = wg_info[4 * wg + 1];
  if ((wg_info[4 * wg + 2] & 0xffff) == 0) return;
  const int stride = wg_info[4 * wg + 3];  // the most lanes any vertex of the patch has: how many shifts the accumulation runs
  // LDS map, float4 units: [rec area 0: lcap local + 64 fetch slots | rec area 1 | spare 64]
  const int rec_stride = lcap + T;
  const int o_ovfA = 2 * rec_stride;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(xbuf);
  constexpr int kPar = 2;
  const int S = rec_bytes, par = 2 * rec_bytes, tab_off = kPar * par;
  const unsigned xcc_want = (tag0 & 0x0fffffffu) << 4;
  const unsigned p0 = tag0 & 1u;

  const size_t hl = (size_t)wg * T + lane, hl2 = (size_t)wg * 2 * T + lane;
  const unsigned meta = wg_meta[hl];
  const int slot0 = wg_slot[hl2], slot1 = wg_slot[hl2 + T];
  const int pv = wg_vid[hl];
  const int nbr_code0 = wg_nbr[hl2], nbr_code1 = wg_nbr[hl2 + T];
  const int frid = (lane < n_fetch) ? wg_fetch[hl] : -1;
  const int loc = (int)((meta >> 13) & 2047u);
  const bool active0 = (meta & kWgActiveBit) != 0u, active1 = slot1 >= 0;
  const bool valid = (meta & kWgValidBit) != 0u, publishes = (meta & kWgPublishBit) != 0u;
  const bool state_lane = (meta & kWgHeadBit) != 0u;
  // a head takes part in shift j while j < the lanes of its vertex; the other lanes of a vertex only serve as sources; a lane without
  // a vertex is disabled altogether
  const unsigned degx = state_lane ? ((meta >> 6) & 127u) : (valid ? 255u : 0u);
  const unsigned long long rm1 = __ballot(degx > 1u), rm2 = __ballot(degx > 2u), rm3 = __ballot(degx > 3u), rm4 = __ballot(degx > 4u),
                           rm5 = __ballot(degx > 5u), rm6 = __ballot(degx > 6u), rm7 = __ballot(degx > 7u), rm8 = __ballot(degx > 8u);
  const int nbr_idx0 = active0 ? ((nbr_code0 < 0) ? lcap + (nbr_code0 & 0x7fffffff) : nbr_code0) : (valid ? loc : 0);
  const int nbr_idx1 = active1 ? ((nbr_code1 < 0) ? lcap + (nbr_code1 & 0x7fffffff) : nbr_code1) : (valid ? loc : 0);

  auto load_slot = [&](bool active, int slot, SlotConst& c, float& q1, v2f_t& q23) {
    int4 rec = make_int4(0, 0, 0, 0);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
      rec = hrec[slot];
      q = hq[slot];
    }
    c.active = active;
    c.is_target = rec.x < 0;
    const float alpha = __int_as_float(rec.y), dx = __int_as_float(rec.z), dy = __int_as_float(rec.w);
    c.beta = q.w;
    q1 = q.x;
    q23 = v2f_t{q.y, q.z};
    c.as = c.is_target ? -alpha : alpha, c.bs = c.is_target ? -c.beta : c.beta, c.ac = c.is_target ? alpha : -alpha;
    c.P12 = v2f_t{alpha * dx, alpha * dy};
    c.C2 = !active ? v2f_t{0.0f, 0.0f} : c.is_target ? v2f_t{c.beta, c.beta} : v2f_t{-dx, -dy};
    c.nbeta = -c.beta;
  };
  SlotConst c0, c1;
  float q1_0, q1_1;
  v2f_t q23_0, q23_1;
  load_slot(active0, slot0, c0, q1_0, q23_0);
  load_slot(active1, slot1, c1, q1_1, q23_1);

  float4 st = make_float4(0.f, 0.f, 0.f, 0.f), bs4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float2 aux = make_float2(0.f, 0.f);
  if (valid) {
    st = vstate[pv];
    aux = vaux[pv];
    bs4 = bar_in[pv];
  }
  const float data = st.w;
  const float thr = p.step_x * (p.data_factor * aux.x);
  float x = st.x;
  v2f_t w12 = {st.y, st.z};
  float xb = bs4.x;
  v2f_t wb12 = {bs4.y, bs4.z};
  float x_prev = x;
  v2f_t w_prev = w12;
  bool ok = true;
  bool timed_out = n_fetch > T;
  bool torn = false;

  const int my_off = (rid_base + loc) << 4;
  const int rec_w = valid ? loc : o_ovfA + lane, rec_wstride = valid ? rec_stride : 0;
  lds[lcap + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  lds[rec_stride + lcap + lane] = make_float4(0.f, 0.f, 0.f, 0.f);

  int off0 = (frid >= 0) ? (frid << 4) : 0;
  if (dual) {
    const unsigned my_xcc = read_xcc_id();
    if (state_lane && publishes)
      __builtin_amdgcn_raw_buffer_store_b32((int)(xcc_want | my_xcc), rx, tab_off + (my_off >> 2), 0, kAuxSc1);
    if (n_fetch > 0 && !timed_out) {
      bool pend = frid >= 0;
      unsigned g0 = 0, spins = 0;
      for (;;) {
        if (pend) {
          int o = tab_off + (frid << 2);
          asm volatile("" : "+v"(o)::"memory");
          g0 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rx, o, 0, kAuxSc1);
          pend = ((g0 & ~15u) != xcc_want);
        }
        if (!__any(pend)) break;
        if (++spins > max_spins) {
          timed_out = true;
          const unsigned long long pm = __ballot(pend);
          const int fl = __ffsll((long long)pm) - 1;
          const int ff = __shfl(frid, fl, 64);
          const unsigned gs = (unsigned)__shfl((int)g0, fl, 64);
          if (lane == 0) report_expired2(err, 2, wg, -1, pm, ff, gs, xcc_want);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (!timed_out && frid >= 0 && (g0 & 15u) == my_xcc) off0 += S;
    }
  }
  const unsigned long long fetch_mask = __ballot(frid >= 0);
  const bool mute = (max_spins_arg >> 31) != 0u && wg == wg_begin;  // test hook: FLAME_NLTGV2_OPT_FAULT_INJECT
  const bool pub_lane = state_lane && publishes;
  const char* const xb_base = static_cast<const char*>(xbuf);
  const char* const src0 = xb_base + off0;
  const char* const src1 = xb_base + off0 + par;
  char* const xb_w = static_cast<char*>(xbuf);
  char* const pa0 = xb_w + my_off;
  char* const pa1 = xb_w + my_off + par;
  auto publish = [&](const v4i_t o, char* pa, const int so) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(pa), "v"(o) : "memory");
    if (dual) __builtin_amdgcn_raw_buffer_store_b128(o, rx, my_off + S, so, 0);
  };
  {
    lds[(p0 ? rec_wstride : 0) + rec_w] = make_float4(xb, wb12.x, wb12.y, __uint_as_float(tag0));
    if (pub_lane && !mute) {
      v4i_t o;
      o.x = __float_as_int(xb), o.y = __float_as_int(wb12.x), o.z = __float_as_int(wb12.y), o.w = (int)tag0;
      publish(o, p0 ? pa1 : pa0, (int)(tag0 & (kPar - 1)) * par);
    }
  }
  lds_wave_sync2();
  const unsigned lds_addr0 = (unsigned)(size_t)(lds);

  auto step = [&](const unsigned s, const unsigned rd_n0, const unsigned rd_n1, const unsigned dst, const int wr_rec, const int it,
                  const char* const src, char* const pub2, const int rd_rec, const int fetch_area) {
    float4 own = lds[rd_rec];
    const int so_out = (int)((s + 1u) & (kPar - 1)) * par;
    v4f_t nbv0, nbv1;
    for (int z = 0; z < pv_presleep; ++z) __builtin_amdgcn_s_sleep(1);
    {
      unsigned cnt, keep, pend_lo, tag_a, tag_b, tagf, gapk;
      unsigned long long pnarrow, exec_saved, waiting;
      const unsigned own_slot = dst + 16u * (unsigned)lane;
      const unsigned f_sleep = (poll_gap & 1) ? 1u + (((unsigned)poll_gap >> 4) & 15u) : 0u, f_narrow = (unsigned)((poll_gap >> 1) & 1);
// What a poll round reads from LDS.  Round 4 (profiles/r04_counters.json: cfg5 / batch10 kept the chip's LDS pipes 62-72 % busy, 61 % of
// that on bank conflicts): every lane's two neighbour tags and both 16-byte neighbour records, round after round -- 11 dwords per lane,
// the records gathered from scattered slots -- is what the polling waves of a CU load its LDS pipe with.  All a round has to find out
// is whether the patch's FOREIGN records have arrived (the local ones were written by this wave before the wait, in program order):
// one tag word per lane from its own fetch slot; the neighbour records are read once, when the last of them is in.
#ifdef FLAME_PV2_FULL_POLL
#define PV2_ROUND_READS                                                                                  \
               "ds_read_b32 %[ta], %[ra] offset:12\n\t"                                                 \
               "ds_read_b32 %[tb], %[rb] offset:12\n\t"                                                 \
               "ds_read_b32 %[t2], %[fa] offset:12\n\t"                                                 \
               "ds_read_b128 %[na], %[ra]\n\t"                                                          \
               "ds_read_b128 %[nb], %[rb]\n\t"
#define PV2_ROUND_PENDING                                                                                \
               "v_cmp_ne_u32_e32 vcc, %[tag], %[ta]\n\t"                                                \
               "s_mov_b64 %[wt], vcc\n\t"                                                               \
               "v_cmp_ne_u32_e32 vcc, %[tag], %[tb]\n\t"                                                \
               "s_or_b64 %[wt], %[wt], vcc\n\t"
#define PV2_AFTER_WAIT
#else
// (only the fetch lanes that still wait read -- a lane that has seen its record keeps the matching tag in its register: -1 % more)
#define PV2_ROUND_READS                                                                                  \
               "s_mov_b64 exec, %[pn]\n\t"                                                              \
               "ds_read_b32 %[t2], %[fa] offset:12\n\t"                                                 \
               "s_mov_b64 exec, %[ex]\n\t"
#define PV2_ROUND_PENDING                                                                                \
               "s_mov_b64 %[wt], %[pn]\n\t"
#define PV2_AFTER_WAIT                                                                                   \
               "ds_read_b128 %[na], %[ra]\n\t"                                                          \
               "ds_read_b128 %[nb], %[rb]\n\t"                                                          \
               "v_mov_b32 %[ta], %[tag]\n\t"                                                            \
               "v_mov_b32 %[tb], %[tag]\n\t"                                                            \
               "s_waitcnt lgkmcnt(0)\n\t"
#endif
#define PV2_POLL                                                                                          \
  asm volatile("s_setprio 0\n\t"                                                                        \
               "s_mov_b64 %[ex], exec\n\t"                                                             \
               "s_mov_b32 %[keep], m0\n\t"                                                              \
               "s_mov_b32 m0, %[dst]\n\t"                                                               \
               "s_mov_b32 %[cnt], 0\n\t"                                                                \
               "s_mov_b64 %[pn], %[fm]\n\t"                                                             \
               "1:\n\t"                                                                                 \
               "s_mov_b64 exec, %[pn]\n\t"                                                              \
               "global_load_lds_dwordx4 %[src], off sc1\n\t"                                            \
               "s_mov_b64 exec, %[ex]\n\t"                                                              \
               "s_mov_b32 %[k], %[fs]\n\t"                                                             \
               "4:\n\t"                                                                                 \
               "s_cmp_eq_u32 %[k], 0\n\t"                                                               \
               "s_cbranch_scc1 3f\n\t"                                                                  \
               "s_sleep 1\n\t"                                                                          \
               "s_sub_u32 %[k], %[k], 1\n\t"                                                            \
               "s_branch 4b\n\t"                                                                        \
               "3:\n\t"                                                                                 \
               PV2_ROUND_READS                                                                          \
               "s_add_u32 %[cnt], %[cnt], 1\n\t"                                                        \
               "s_waitcnt lgkmcnt(0)\n\t"                                                               \
               "v_cmp_ne_u32_e32 vcc, %[tag], %[t2]\n\t"                                                \
               "s_and_b64 %[pn], vcc, %[fm]\n\t"                                                        \
               PV2_ROUND_PENDING                                                                        \
               "s_cmp_eq_u32 %[fn], 0\n\t"                                                              \
               "s_cselect_b64 %[pn], %[fm], %[pn]\n\t"                                                  \
               "s_cmp_eq_u64 %[wt], 0\n\t"                                                              \
               "s_cbranch_scc1 2f\n\t"                                                                  \
               "s_cmp_lt_u32 %[cnt], 64\n\t"                                                            \
               "s_cbranch_scc1 1b\n\t"                                                                  \
               "2:\n\t"                                                                                 \
               "s_setprio 3\n\t"                                                                        \
               PV2_AFTER_WAIT                                                                           \
               "s_mov_b32 m0, %[keep]"                                                                   \
               : [keep] "=&s"(keep), [cnt] "=&s"(cnt), [na] "=&v"(nbv0), [nb] "=&v"(nbv1), [ta] "=&v"(tag_a), [tb] "=&v"(tag_b),   \
                 [t2] "=&v"(tagf), [pn] "=&s"(pnarrow), [k] "=&s"(gapk), [ex] "=&s"(exec_saved), [wt] "=&s"(waiting)                \
               : [src] "v"(src), [dst] "s"(dst), [ra] "v"(rd_n0), [rb] "v"(rd_n1), [fa] "v"(own_slot), [tag] "s"(s), [fm] "s"(fetch_mask), \
                 [fs] "s"(f_sleep), [fn] "s"(f_narrow)                                                     \
               : "vcc", "scc", "memory")
      PV2_POLL;
      pend_lo = (unsigned)waiting | (unsigned)(waiting >> 32);
      if (__builtin_expect(pend_lo != 0u, 0)) {
        for (unsigned outer = 0;;) {
          const int ab = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ab != 0 || ++outer > (max_spins >> 4)) {
            timed_out = true;
            if (ab == 0) {
              const unsigned long long pm = __ballot(tag_a != s || tag_b != s || (frid >= 0 && tagf != s));
              const int fl = __ffsll((long long)pm) - 1;
              const unsigned gs = (unsigned)__shfl((int)tag_a, fl, 64);
              if (lane == 0) report_expired2(err, 3, wg, it, pm, -1, gs, s);
            }
            break;
          }
          PV2_POLL;
          pend_lo = (unsigned)waiting | (unsigned)(waiting >> 32);
          if (pend_lo == 0u) break;
        }
      }
#undef PV2_POLL
    }
    if (VERIFY && verify && !timed_out) {
      // every fetch lane reads its foreign record once more, with an ordinary load, and compares all four dwords with what the
      // LDS-DMA left in its slot (k_persistent_pv): a difference is a torn 16-byte access -- reported, the run is taken back
      v4i_t g2 = {0, 0, 0, 0};
      if (frid >= 0) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g2) : "v"(src) : "memory");
      }
      const float4 l4 = lds[fetch_area + lane];
      if ((verify & 2) && it == 2 && wg == wg_begin && lane == 0) g2.x ^= 0x00400000;  // test hook
      const bool bad = frid >= 0 && (g2.x != __float_as_int(l4.x) || g2.y != __float_as_int(l4.y) || g2.z != __float_as_int(l4.z) ||
                                     (unsigned)g2.w != s || __float_as_uint(l4.w) != s);
      if (__any(bad)) torn = timed_out = true;
    }
    xb = own.x, wb12 = v2f_t{own.y, own.z};
    // ---- dual update of the two half-edges' private q copies (cc:99-110) and their shares of the primal scatter (cc:126-141) ----
    float cx0, cx1, q1r0, q1r1;
    v2f_t a0, b0, a1, b1, q23r0, q23r1;
    auto edge = [&](const SlotConst& c, const v4f_t nbv, float& q1, v2f_t& q23, float& cx, v2f_t& a12, v2f_t& b12, float& q1r, v2f_t& q23r) {
      const v2f_t nbw = {nbv.y, nbv.z};
      const float d0 = xb - nbv.x;
      const v2f_t d12 = wb12 - nbw;
      const v2f_t wbi = c.is_target ? nbw : wb12;
      float K1 = c.as * d0;
      const v2f_t m12 = c.P12 * wbi;
      K1 -= m12.x;
      K1 -= m12.y;
      const v2f_t K23 = c.bs * d12;
      q1r = q1 + p.step_q * K1;
      q23r = q23 + p.step_q * K23;
      q1 = __builtin_fminf(__builtin_fmaxf(q1r, -1.0f), 1.0f);
      q23.x = __builtin_fminf(__builtin_fmaxf(q23r.x, -1.0f), 1.0f);
      q23.y = __builtin_fminf(__builtin_fmaxf(q23r.y, -1.0f), 1.0f);
      const float u1 = q1 * p.step_x;
      const v2f_t u23 = q23 * p.step_x;
      cx = u1 * c.ac;
      const v2f_t M2 = c.is_target ? u23 : v2f_t{cx, cx};
      a12 = M2 * c.C2;
      b12 = u23 * c.nbeta;
      b12 = c.is_target ? v2f_t{-0.0f, -0.0f} : b12;
    };
    edge(c0, nbv0, q1_0, q23_0, cx0, a0, b0, q1r0, q23r0);
    edge(c1, nbv1, q1_1, q23_1, cx1, a1, b1, q1r1, q23r1);
    // ---- ordered accumulation: the head starts with its own two half-edges, shift j adds the two of lane first + j ----
    float X = x + cx0, W1 = (w12.x + a0.x) + b0.x, W2 = (w12.y + a0.y) + b0.y;
    X = X + cx1;
    W1 = (W1 + a1.x) + b1.x;
    W2 = (W2 + a1.y) + b1.y;
    {
#define PV2_ADDS(J)                                                                                   \
  "v_add_f32_dpp %[X], %[cx0], %[X] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_add_f32_dpp %[W1], %[a10], %[W1] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                  \
  "v_add_f32_dpp %[W2], %[a20], %[W2] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                  \
  "v_add_f32_dpp %[X], %[cx1], %[X] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_add_f32_dpp %[W1], %[b10], %[W1] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                  \
  "v_add_f32_dpp %[W2], %[b20], %[W2] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                  \
  "v_add_f32_dpp %[W1], %[a11], %[W1] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                  \
  "v_add_f32_dpp %[W2], %[a21], %[W2] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                  \
  "v_add_f32_dpp %[W1], %[b11], %[W1] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                  \
  "v_add_f32_dpp %[W2], %[b21], %[W2] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define PV2_RM(J, M) "s_mov_b64 exec, %[" #M "]\n\t" PV2_ADDS(J)
      unsigned long long exec_saved;
      asm volatile("s_mov_b64 %[ex], exec\n\t"
                   "s_nop 1\n\t"
                   PV2_RM(1, m1)
                   "s_cmp_le_u32 %[md], 2\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   PV2_RM(2, m2)
                   "s_cmp_le_u32 %[md], 3\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   PV2_RM(3, m3)
                   "s_cmp_le_u32 %[md], 4\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   PV2_RM(4, m4)
                   "s_cmp_le_u32 %[md], 5\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   PV2_RM(5, m5)
                   "s_cmp_le_u32 %[md], 6\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   PV2_RM(6, m6) PV2_RM(7, m7)
                   "s_cmp_le_u32 %[md], 8\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   PV2_RM(8, m8) PV2_ADDS(9) PV2_ADDS(10) PV2_ADDS(11) PV2_ADDS(12) PV2_ADDS(13) PV2_ADDS(14) PV2_ADDS(15)
                   "9:\n\t"
                   "s_mov_b64 exec, %[ex]"
                   : [X] "+v"(X), [W1] "+v"(W1), [W2] "+v"(W2), [ex] "=&s"(exec_saved)
                   : [cx0] "v"(cx0), [a10] "v"(a0.x), [a20] "v"(a0.y), [b10] "v"(b0.x), [b20] "v"(b0.y), [cx1] "v"(cx1), [a11] "v"(a1.x),
                     [a21] "v"(a1.y), [b11] "v"(b1.x), [b21] "v"(b1.y), [md] "s"(stride), [m1] "s"(rm1), [m2] "s"(rm2), [m3] "s"(rm3),
                     [m4] "s"(rm4), [m5] "s"(rm5), [m6] "s"(rm6), [m7] "s"(rm7), [m8] "s"(rm8)
                   : "scc");
#undef PV2_RM
#undef PV2_ADDS
    }
    const v2f_t Wa = {W1, W2};
    // ---- vertex update: proxL1 (cc:147-151, h:179-197), extragradient (cc:160-171) ----
    const float diff = X - data, x_dn = X - thr, x_up = X + thr;
    float xn = (diff < -thr) ? x_up : data;
    xn = (diff > thr) ? x_dn : xn;
    xn = (xn < p.x_min) ? p.x_min : xn;
    xn = (xn > p.x_max) ? p.x_max : xn;
    float nb = xn + p.theta * (xn - x);
    nb = (nb < p.x_min) ? p.x_min : nb;
    nb = (nb > p.x_max) ? p.x_max : nb;
    const v2f_t wbn = Wa + p.theta * (Wa - w12);
    if (pub_lane) {
      v4i_t o;
      o.x = __float_as_int(nb), o.y = __float_as_int(wbn.x), o.z = __float_as_int(wbn.y), o.w = (int)(s + 1u);
      publish(o, pub2, so_out);
    }
    __builtin_amdgcn_s_setprio(0);
    if (state_lane || !valid) lds[wr_rec] = make_float4(nb, wbn.x, wbn.y, __uint_as_float(s + 1u));
    ok = ok && (__builtin_fabsf(q1r0) <= 3.402823466e+38f) && (__builtin_fabsf(q23r0.x) <= 3.402823466e+38f) &&
         (__builtin_fabsf(q23r0.y) <= 3.402823466e+38f) && (__builtin_fabsf(q1r1) <= 3.402823466e+38f) &&
         (__builtin_fabsf(q23r1.x) <= 3.402823466e+38f) && (__builtin_fabsf(q23r1.y) <= 3.402823466e+38f);
    x_prev = x, w_prev = w12;
    x = xn, w12 = Wa;
    xb = nb, wb12 = wbn;
  };

  const int areaA = p0 ? rec_stride : 0, areaB = p0 ? 0 : rec_stride;
  const unsigned rdA0 = lds_addr0 + 16u * (unsigned)(areaA + nbr_idx0), rdB0 = lds_addr0 + 16u * (unsigned)(areaB + nbr_idx0);
  const unsigned rdA1 = lds_addr0 + 16u * (unsigned)(areaA + nbr_idx1), rdB1 = lds_addr0 + 16u * (unsigned)(areaB + nbr_idx1);
  const unsigned dstA = __builtin_amdgcn_readfirstlane(lds_addr0 + 16u * (unsigned)(areaA + lcap));
  const unsigned dstB = __builtin_amdgcn_readfirstlane(lds_addr0 + 16u * (unsigned)(areaB + lcap));
  const int wrA_rec = (valid ? areaA : 0) + rec_w, wrB_rec = (valid ? areaB : 0) + rec_w;
  const char* const srcA = p0 ? src1 : src0;
  const char* const srcB = p0 ? src0 : src1;
  char* const pubA = p0 ? pa1 : pa0;
  char* const pubB = p0 ? pa0 : pa1;
  int it = 0;
  if (OPEN) {
    constexpr unsigned kOpenMargin = 128u, kOpenCheck = 64u;  // (as in k_persistent_pv)
    const bool decides = wg == wg_begin + n_wgs / 2;
    const unsigned* const stop_req = tail->stop_req;
    unsigned* const stop_word = reinterpret_cast<unsigned*>(err) + 12;
    unsigned stop_at = 0u;
    for (; it + 1 < n_iters && !timed_out; it += 2) {
      if (((unsigned)it & (kOpenCheck - 1u)) == 0u) {
        if (decides && stop_at == 0u && stop_req &&
            (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(stop_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == tag0) {
          stop_at = tag0 + (unsigned)it + kOpenMargin;
          if (lane == 0) {
            __hip_atomic_store(stop_word, stop_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (taken: tags start over with every topology, the next graph's first run has this tag0 again)
            __hip_atomic_store(const_cast<unsigned*>(stop_req), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (stop_at == 0u) {
          const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (w > tag0 && w - tag0 <= (unsigned)n_iters + kOpenMargin) stop_at = w;
        }
      }
      if (stop_at != 0u && stop_at - tag0 <= (unsigned)it) break;
      step(tag0 + (unsigned)it, rdA0, rdA1, dstA, wrB_rec, it, srcA, pubB, wrA_rec, areaA + lcap);
      if (timed_out) break;
      step(tag0 + (unsigned)it + 1u, rdB0, rdB1, dstB, wrA_rec, it + 1, srcB, pubA, wrB_rec, areaB + lcap);
    }
    if (decides && lane == 0 && !timed_out) reinterpret_cast<unsigned*>(err)[13] = tag0 + (unsigned)it;
  } else {
    for (; it + 1 < n_iters && !timed_out; it += 2) {
      step(tag0 + (unsigned)it, rdA0, rdA1, dstA, wrB_rec, it, srcA, pubB, wrA_rec, areaA + lcap);
      if (timed_out) break;
      step(tag0 + (unsigned)it + 1u, rdB0, rdB1, dstB, wrA_rec, it + 1, srcB, pubA, wrB_rec, areaB + lcap);
    }
    if (it < n_iters && !timed_out) step(tag0 + (unsigned)it, rdA0, rdA1, dstA, wrB_rec, it, srcA, pubB, wrA_rec, areaA + lcap);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (timed_out) {
    if (lane == 0) {
      __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicOr(err, torn ? 4 : 2);
    }
    return;
  }
  if (state_lane) {
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
  if (active0) hq_out[slot0] = make_float4(q1_0, q23_0.x, q23_0.y, c0.beta);
  if (active1) hq_out[slot1] = make_float4(q1_1, q23_1.x, q23_1.y, c1.beta);
  if (!ok && (active0 || active1)) atomicOr(err, 1);
}

}  // namespace

// Patches of k_persistent_pv2 really co-resident per CU (see pv_real_waves_per_simd in nltgv2_persistent.hip): from the kernel's
// register counts as built -- <= 96 VGPRs, <= 96 SGPRs: five waves per SIMD; the instance with the record verification
// <= 112 VGPRs: four (tests/test_abi.py re-derives both from the compiler's resource report).
int pv2_patches_per_cu(int lcap, bool verify) {
  const size_t ldsv = 16u * (size_t)(2 * (lcap + 64) + 64);
  int n = 0;
  const void* fn = verify ? (const void*)k_persistent_pv2<true> : (const void*)k_persistent_pv2<false>;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 64, ldsv) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (verify) return n < 16 ? n : 16;
  return n < 20 ? n : 20;
}

int launch_persistent_pv2(const FusedArgs& a, const Pv2Args& w, const SolverParams& p, int wg_begin, int n_wgs, int parity_in, unsigned tag0,
                          int n_iters, unsigned max_spins, int poll_gap, int dual, const RunTail* tail, bool cooperative, hipStream_t stream) {
  if (n_wgs <= 0 || n_iters <= 0) return (int)hipSuccess;
  int wgx = (n_wgs + 7) / 8;
  const dim3 gv((unsigned)(wgx * 8)), bv(64u);
  int lcap = w.lcap;
  const int32_t *w0 = w.slot, *w1 = w.vid, *w3 = w.nbr, *w4 = w.fetch, *w5 = w.info;
  const uint32_t* w2 = w.meta;
  const int4* hrec = a.hrec;
  const float4* hq = a.hq;
  const float4* vstate = a.vstate;
  float4* hq_out = a.hq_out;
  float4* vstate_out = a.vstate_out;
  const float2* vaux = a.vaux;
  const float4* bin = a.bar[parity_in];
  float4* bout = a.bar[parity_in ^ 1];
  float4* vprev = a.vprev;
  void* xbuf = a.xbuf;
  int rec_bytes = (a.n_rec > a.n_slices * 64 ? a.n_rec : a.n_slices * 64) * 16;
  SolverParams pp = p;
  int* err = a.err;
  int* abort_flag = a.abort_flag;
  const int32_t* perm = a.perm;
  const unsigned ldsv = 16u * (unsigned)(2 * (lcap + 64) + 64);
  void* vargs[] = {&wg_begin, &n_wgs, &wgx, &lcap, &w0, &w1, &w2, &w3, &w4, &w5, &hrec, &hq, &vstate, &hq_out, &vstate_out, &vaux, &bin, &bout,
                   &vprev, &xbuf, &rec_bytes, &dual, &tag0, &n_iters, &max_spins, &poll_gap, &pp, &err, &abort_flag, &perm, &tail};
  const void* fn = (dual >> 1) != 0 ? (const void*)k_persistent_pv2<true> : a.open_run ? (const void*)k_persistent_pv2<false, true> : (const void*)k_persistent_pv2<false>;
  if (cooperative) return (int)hipLaunchCooperativeKernel(fn, gv, bv, vargs, ldsv, stream);
  return (int)hipExtLaunchKernel(fn, gv, bv, vargs, ldsv, stream, nullptr, a.stop_event, 0);
}

// Loads this translation unit's code object (the runtime does that at the first use of one of its kernels: several milliseconds that
// flame_nltgv2_create takes on itself so that the first frame does not).
void warm_module_persistent_pv2() {
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, (const void*)k_persistent_pv2<false>) != hipSuccess) (void)hipGetLastError();
}

}  // namespace flame_hip
