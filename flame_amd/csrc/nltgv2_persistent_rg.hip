// nltgv2_persistent_rg.hip -- k_persistent_rg: the REGION-per-workgroup persistent run (layout (R), nltgv2_regions.hpp): n steps in
// ONE launch, a workgroup (one CU at the headline size) owns a compact region of the graph plus a ghost ring of depth k, neighbours
// meet in LDS, and the L2 hand-off that the patch-per-wave forms pay every step is paid once per BLOCK of k steps.
//
// Why (round 5; docs/LAB_NOTES.md): the period of the lock-step patch network is one wave's dependent instructions (~640-900 cycles)
// plus one L2 hand-off (~1100-1500), and three rounds of work on the hand-off's detection moved it by a few per cent.  What can move
// it is paying it less often -- a k-step block with a recomputed ring -- and that only pays if the redundant ring is cheap in issue
// slots: four SIMDs per CU, one vector instruction per SIMD and 4 cycles.  Hence the two lane kinds of this kernel:
//   E lanes  one lane per EDGE (one copy of q per edge and region): the dual update (cc:99-110) and both endpoints' ordered
//            contributions of the primal scatter (cc:126-141), written to per-(vertex, rank) LDS slots;
//   V lanes  one lane per VERTEX: the sum of its slots in ascending edge id (the order of the reference's scatter), proxL1
//            (h:179-197), the extragradient (cc:160-171), the new bar record back into LDS.
// A step is E phase, barrier, V phase, barrier; ~a quarter of the vector instructions of the lane-per-half-edge forms.
//
// Exactness.  Every lane evaluates the reference's expressions in the reference's order (-ffp-contract=off), so two regions that
// compute the same vertex from the same inputs get the same bits.  After sub-step s of a block of kb steps the vertices of depth
// <= kb - s and the edges of level (larger endpoint depth) <= kb - s + 1 are exact; lanes beyond that are skipped and what they
// hold is stale but never read by a lane that still counts.  At the block's end the owners publish 16-byte tagged records --
// {x, w1, w2, tag}, {x_bar, w1_bar, w2_bar, tag}, {q1, q2, q3, tag} -- and every region refreshes its ring from them: the data is
// its own flag, two buffers by block parity, a write-through copy for readers on other XCDs and a plain copy for readers on the
// writer's XCD (its L2), chosen from the true XCC ids -- the protocol of nltgv2_persistent.hip, per block instead of per step.
// tests/cpp/rg_layout_test.cc replays layout and schedule on the CPU against the checker.
//
// The run is transactional like the other persistent forms: reads hq / vstate / bar_in, writes hq_out / vstate_out / bar_out; every
// wait is bounded and reports through `err` (bit 1: a wait expired; the host rolls back and replays).
#include <type_traits>

#include "nltgv2_device.hpp"
#include "nltgv2_regions.hpp"

namespace flame_hip {

namespace {

typedef float v2f_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void store_rec_sc1(char* p, v4i_t o) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(o) : "memory");
}
// The workgroup's barrier between two phases of a step: LDS traffic only.  (__syncthreads() also waits for every outstanding
// global access of the wave -- vmcnt(0) -- and a write-through record store is acknowledged ~1100-1300 cycles after it was issued.)
__device__ __forceinline__ void rg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef FLAME_RG_DIAG
#define rg_barrier() do { if (!(dual & 2048)) rg_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#endif
__device__ __forceinline__ unsigned rg_hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}

// LDS map (nltgv2_regions.hpp: rg_lds_bytes), float4 units: [bar: nb_cap | poll slots: f_cap x block_threads | spare: block_threads |
// contributions c4 {a1, a2, b1, b2}: nc_cap x SD], then floats [c1 {cx}: nc_cap x SD | spare: block_threads].
// SD = slots per vertex = the largest degree rounded up to 8, + 1: a vertex's slots are consecutive (the V phase reads them with
// immediate offsets, eight per round of loads), the odd stride keeps the lanes of a wave on different banks.
// Two instances by workgroup size: up to 512 lanes (256 VGPRs per lane: eight slots per round of loads) and up to 1024 (128 VGPRs).
template <bool PROBE, int MAXT>
__global__ void __launch_bounds__(MAXT)
k_persistent_rg(const RgArgs a, const int4* __restrict__ hrec, const float4* __restrict__ hq, const float4* __restrict__ vstate,
                float4* __restrict__ hq_out, float4* __restrict__ vstate_out, const float2* __restrict__ vaux,
                const float4* __restrict__ bar_in, float4* __restrict__ bar_out, float4* __restrict__ vprev, const unsigned tag0,
                const int n_iters, const unsigned max_spins_arg, const SolverParams p, int* __restrict__ err,
                int* __restrict__ abort_flag, const int32_t* __restrict__ perm, const RunTail* __restrict__ tail, const int dual,
                unsigned* __restrict__ probe) {
  extern __shared__ float4 lds[];
  const int b = (int)blockIdx.x;
  const int xcd = b & 7, idx = b >> 3;
  const int rpx = (a.n_regions + 7) >> 3;
  if (idx >= rpx) return;
  const int region = xcd * rpx + idx;  // consecutive regions (neighbours in space: the bisection's leaf order) share an XCD
  if (region >= a.n_regions) return;
  const int t = (int)threadIdx.x;
  const int32_t* const inf = a.info + (size_t)region * kRgInfoWords;
  const int t_off = inf[0], n_threads = inf[1], n_vc = inf[2], n_vall = inf[3], n_e = inf[4], F = inf[5], f_off = inf[6];
  const int k = a.depth, NC = a.nc_cap, SD = rg_slot_stride(a.deg_cap);
  const unsigned max_spins = max_spins_arg & 0x7fffffffu;
  const bool mute = (max_spins_arg >> 31) != 0u && region == 0;  // test hook: FLAME_NLTGV2_OPT_FAULT_INJECT
  const int presleep = (dual >> 16) & 255;         // x 64 cycles between a block's end and its first poll
  const unsigned poll_gap = (unsigned)(dual >> 24) & 15u;  // x 64 cycles between two poll rounds
  const int o_poll = a.nb_cap, o_spare = o_poll + (a.f_cap > 0 ? a.f_cap : 1) * a.block_threads, o_c4 = o_spare + a.block_threads;
  float* const c1 = reinterpret_cast<float*>(lds + o_c4 + (size_t)NC * SD);

  // ---- this lane's vertex role ------------------------------------------------------------------------------------------------
  const bool v_local = t < n_vall, v_comp = t < n_vc;
  int pv = -1, v_fa = -1;
  unsigned vmeta = 0u;
  if (v_local) pv = a.v_pv[t_off + t], vmeta = a.v_meta[t_off + t], v_fa = a.v_fa[t_off + t];
  const int vdepth = v_comp ? (int)(vmeta & kRgDepthMask) : 1000;
  const bool v_owned = (vmeta & kRgOwned) != 0u;
  float x = 0.f, data = 0.f, thr = 0.f;
  v2f_t w12 = {0.f, 0.f};
  if (v_comp) {
    const float4 st = vstate[pv];
    const float2 aux = vaux[pv];
    x = st.x, w12 = v2f_t{st.y, st.z}, data = st.w;
    thr = p.step_x * (p.data_factor * aux.x);
  }
  float x_prev = x;
  v2f_t w_prev = w12;
  float xb = 0.f;
  v2f_t wb12 = {0.f, 0.f};
  if (v_local) {
    const float4 bs = bar_in[pv];
    xb = bs.x, wb12 = v2f_t{bs.y, bs.z};
    lds[t] = make_float4(bs.y, bs.z, bs.x, 0.f);  // (the order the edge lanes want: the w pair first)
  }
  // the slots the V phase of this WAVE runs over: the largest degree of its computed vertices
  int wdeg = v_comp ? (int)((vmeta >> kRgDegShift) & 255u) : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wdeg = max(wdeg, __shfl_xor(wdeg, o, 64));
  wdeg = __builtin_amdgcn_readfirstlane(wdeg);
  const int wvdepth = __builtin_amdgcn_readfirstlane(vdepth);  // (vertex lanes are sorted by depth: the wave's smallest)
  const float4* const cp = lds + o_c4 + (size_t)(v_comp ? t : 0) * SD;  // this vertex's slots
  const float* const bp = c1 + (size_t)(v_comp ? t : 0) * SD;
  const float4* const fa_p = lds + (v_fa >= 0 ? o_poll + v_fa : 0);     // where a ring vertex's {x, w1, w2} record lands
  float4* const bar_w = lds + (v_comp ? t : o_spare + t);               // where this lane's new bar record goes

  // ---- this lane's edge role ------------------------------------------------------------------------------------------------
  const bool e_on = t < n_e;
  unsigned emeta = 0u, eli = 0u;
  int slot_s = -1, slot_d = -1, eid = -1, e_fq = -1, e_fbs = -1, e_fbd = -1;
  float q1 = 0.f, alpha = 0.f, beta = 0.f;
  v2f_t q23 = {0.f, 0.f}, D12 = {0.f, 0.f};
  if (e_on) {
    emeta = a.e_meta[t_off + t], eli = a.e_li[t_off + t];
    slot_s = a.e_slot_src[t_off + t], slot_d = a.e_slot_dst[t_off + t], eid = a.e_id[t_off + t];
    e_fq = a.e_fq[t_off + t], e_fbs = a.e_fbs[t_off + t], e_fbd = a.e_fbd[t_off + t];
    const int4 rec = hrec[slot_s];
    const float4 q = hq[slot_s];
    alpha = __int_as_float(rec.y), D12 = v2f_t{__int_as_float(rec.z), __int_as_float(rec.w)};
    q1 = q.x, q23 = v2f_t{q.y, q.z}, beta = q.w;
  }
  const int elevel = e_on ? (int)(emeta >> kRgLevelShift) : 1000;
  const bool e_home = (emeta & kRgHome) != 0u;
  const v2f_t P12 = alpha * D12;  // cc:101-102: alpha * (pos_i - pos_j) first, then the product with w_bar
  const int li_s = (int)(eli & 0xffffu), li_d = (int)(eli >> 16);
  // where this edge's contributions go: slot `rank` of the endpoint's vertex; an endpoint that is not computed here (depth k) gets a
  // spare entry of the lane's own, so the E phase has no branch
  const bool s_comp = (emeta & kRgSrcComputed) != 0u, d_comp = (emeta & kRgDstComputed) != 0u;
  float4* const cs4 = lds + (s_comp ? o_c4 + li_s * SD + (int)(emeta & 255u) : o_spare + t);
  float* const cs1 = c1 + (s_comp ? li_s * SD + (int)(emeta & 255u) : NC * SD + t);
  float* const cd1 = c1 + (d_comp ? li_d * SD + (int)((emeta >> 8) & 255u) : NC * SD + t);
  const float nbeta = -beta;
  float4* const cd4 = lds + (d_comp ? o_c4 + li_d * SD + (int)((emeta >> 8) & 255u) : o_spare + t);
  // where the endpoints' bar records are read: what the region computed itself (bar[li]) -- or, in the FIRST step of a block, the
  // poll slot a ring vertex's record was fetched into (its bar entry is stale until the region has computed the vertex once)
  const float4* const bi_p = lds + li_s;
  const float4* const bj_p = lds + li_d;
  const float4* const bi_f = e_fbs >= 0 ? lds + o_poll + e_fbs : bi_p;
  const float4* const bj_f = e_fbd >= 0 ? lds + o_poll + e_fbd : bj_p;
  const float4* const fq_p = lds + (e_fq >= 0 ? o_poll + e_fq : 0);
  // the level below which this WAVE has an edge to run (edges are sorted by level: the first lane's is the wave's smallest)
  const int wlevel = __builtin_amdgcn_readfirstlane(elevel);
  bool ok = true;

  // ---- the contribution slots start as the additive identity (x + -0.0f == x for every x, both zeros included), the poll slots with
  // tag 0 (never a live tag) -----------------------------------------------------------------------------------------------------
  {
    const int n4 = NC * SD;
    const float4 z4 = make_float4(-0.0f, -0.0f, -0.0f, -0.0f);
    for (int i = t; i < n4; i += (int)blockDim.x) lds[o_c4 + i] = z4, c1[i] = -0.0f;
    for (int i = o_poll + t; i < o_spare; i += (int)blockDim.x) lds[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- exchange buffers: [parity][remote copy | same-XCD copy] of n_rec records, then one word per region ---------------------
  const size_t S = (size_t)a.n_rec * 16u, par = 2u * S;
  char* const xb_base = a.xbuf;
  unsigned* const xcc_tab = reinterpret_cast<unsigned*>(xb_base + 2u * par);
  const unsigned xcc_want = (tag0 & 0x0fffffffu) << 4;
  const unsigned my_xcc = read_xcc_id();
  if (t == 0) __hip_atomic_store(xcc_tab + region, xcc_want | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // fetch duties of this lane (at most two): record fs_j lands in poll slot j * n_threads + t
  int fs0 = -1, fs1 = -1, fp0 = 0, fp1 = 0;
  if (t < n_threads && F > 0) fs0 = a.f_src[f_off + t], fp0 = a.f_prod[f_off + t];
  if (t < n_threads && F > 1) fs1 = a.f_src[f_off + n_threads + t], fp1 = a.f_prod[f_off + n_threads + t];
  const unsigned long long fm0 = __ballot(fs0 >= 0), fm1 = __ballot(fs1 >= 0);
  const char* src0[2] = {xb_base + (size_t)(fs0 < 0 ? 0 : fs0) * 16u, xb_base + par + (size_t)(fs0 < 0 ? 0 : fs0) * 16u};  // by parity
  const char* src1[2] = {xb_base + (size_t)(fs1 < 0 ? 0 : fs1) * 16u, xb_base + par + (size_t)(fs1 < 0 ? 0 : fs1) * 16u};
  const unsigned lds_addr0 = (unsigned)(size_t)(lds);
  const int wave0 = t & ~63;
  const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds_addr0 + 16u * (unsigned)(o_poll + wave0));
  const unsigned dst1 = __builtin_amdgcn_readfirstlane(lds_addr0 + 16u * (unsigned)(o_poll + n_threads + wave0));
  const unsigned own0 = lds_addr0 + 16u * (unsigned)(o_poll + (t < n_threads ? t : 0));
  const unsigned own1 = (F > 1 && t < n_threads) ? own0 + 16u * (unsigned)n_threads : own0;
  // where this lane's own records go
  const bool pubA = v_owned && (vmeta & kRgExportA) != 0u, pubB = v_owned && (vmeta & kRgExportB) != 0u;
  const bool pubQ = e_home && (emeta & kRgExportQ) != 0u;
  const size_t offA = (size_t)(pv < 0 ? 0 : pv) * 16u, offB = (size_t)(a.n_packed + (pv < 0 ? 0 : pv)) * 16u;
  const size_t offQ = (size_t)(2 * a.n_packed + (eid < 0 ? 0 : eid)) * 16u;
  auto publish = [&](char* base, size_t off, float v0, float v1, float v2, unsigned T) {
    v4i_t o;
    o.x = __float_as_int(v0), o.y = __float_as_int(v1), o.z = __float_as_int(v2), o.w = (int)T;
    store_rec_sc1(base + off, o);
    if (dual & 1) *reinterpret_cast<v4i_t*>(base + S + off) = o;
  };
  __syncthreads();
  // the first step of EVERY block reads a ring endpoint's bar record from its poll slot: block 0 finds the initial values there
  if (e_fbs >= 0) lds[o_poll + e_fbs] = *bi_p;
  if (e_fbd >= 0) lds[o_poll + e_fbd] = *bj_p;
  __syncthreads();

  unsigned pr_t0 = 0, pr_ack = 0, pr_load = 0;
  if (PROBE) {
    // two figures the cycle account is read against: how long a write-through store stays outstanding (issue -> vmcnt 0), and what
    // one sc1 load of a line nobody is writing costs -- measured by lane 0 on this region's word of the XCC table
    if (t == 0) {
      char* const w = reinterpret_cast<char*>(xcc_tab + region);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned c0 = (unsigned)clock64();
      asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(w), "v"(xcc_want | my_xcc) : "memory");
      const unsigned c1 = (unsigned)clock64();
      unsigned v;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(w) : "memory");
      const unsigned c2 = (unsigned)clock64();
      pr_ack = c1 - c0, pr_load = c2 - c1 + (v & 0u);
    }
    __syncthreads();
    pr_t0 = (unsigned)clock64();
  }
  int left = n_iters, blk = 0;
  bool dead = false;  // (wave-uniform) a wait of this wave expired, or the run is being aborted: no more polls, the results are void
  while (left > 0) {
    const int kb = left < k ? left : k;
    unsigned pr_spins = 0, pr_t1 = 0, pr_first = 0;
    const bool fresh = blk > 0;
    if (fresh) {
      // ---- refresh the ring from the records of block blk - 1 -------------------------------------------------------------
      const unsigned T = tag0 + (unsigned)blk - 1u;
      const int pb = (int)(T & 1u);
      if (blk == 1 && (dual & 1)) {
        // once per launch (before the first wait, not at the start: block 0 needs nothing from anybody): which copy of a record
        // this lane polls -- the one in this XCD's L2 if its producer runs on this XCD
        for (int j = 0; j < 2; ++j) {
          const int fs = j ? fs1 : fs0, fp = j ? fp1 : fp0;
          unsigned v = xcc_want | my_xcc, spins = 0;
          bool pend = fs >= 0;
          while (__any(pend)) {
            if (pend) v = __hip_atomic_load(xcc_tab + fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pend = pend && (v & ~15u) != xcc_want;
            if (++spins > max_spins) {
              dead = true;
              break;
            }
            __builtin_amdgcn_s_sleep(2);
          }
          if (fs >= 0 && (v & ~15u) == xcc_want && (v & 15u) == my_xcc) {
            if (j) src1[0] += S, src1[1] += S; else src0[0] += S, src0[1] += S;
          }
        }
      }
      for (int z = 0; z < presleep; ++z) __builtin_amdgcn_s_sleep(1);  // (no record can be here sooner than one hand-off after its producer's publish)
      if (!dead) {
        // One statement: up to 64 rounds of { one LDS-DMA load per fetch duty still waiting (global_load_lds_dwordx4 sc1: lane i's 16
        // bytes land in poll slot i of the round's area, no register and no vmcnt wait -- an outstanding write-through store of this
        // wave is acknowledged ~1200 cycles after it was issued, and a wait for the loads would wait for it too), the tag words of
        // the lane's own slots back from LDS }.  The data is its own flag.  EXEC and M0 are put back as they were found.
        const char* const s0 = src0[pb];
        const char* const s1 = src1[pb];
        unsigned cnt, keep, pend_lo, tg0, tg1, gapk;
        unsigned long long p0m, p1m, exec_saved;
#define RG_POLL                                                                                              \
  asm volatile("s_mov_b64 %[ex], exec\n\t"                                                                 \
               "s_mov_b32 %[keep], m0\n\t"                                                                 \
               "s_mov_b32 %[cnt], 0\n\t"                                                                   \
               "s_mov_b64 %[p0], %[fm0]\n\t"                                                               \
               "s_mov_b64 %[p1], %[fm1]\n\t"                                                               \
               "1:\n\t"                                                                                    \
               "s_mov_b64 exec, %[p0]\n\t"                                                                 \
               "s_mov_b32 m0, %[d0]\n\t"                                                                   \
               "global_load_lds_dwordx4 %[s0], off sc1\n\t"                                                \
               "s_mov_b64 exec, %[p1]\n\t"                                                                 \
               "s_mov_b32 m0, %[d1]\n\t"                                                                   \
               "global_load_lds_dwordx4 %[s1], off sc1\n\t"                                                \
               "s_mov_b64 exec, %[ex]\n\t"                                                                 \
               "s_mov_b32 %[k], %[gap]\n\t"                                                              \
               "4:\n\t"                                                                                    \
               "s_cmp_eq_u32 %[k], 0\n\t"                                                                \
               "s_cbranch_scc1 3f\n\t"                                                                     \
               "s_sleep 1\n\t"                                                                             \
               "s_sub_u32 %[k], %[k], 1\n\t"                                                             \
               "s_branch 4b\n\t"                                                                           \
               "3:\n\t"                                                                                    \
               "ds_read_b32 %[t0], %[o0] offset:12\n\t"                                                    \
               "ds_read_b32 %[t1], %[o1] offset:12\n\t"                                                    \
               "s_add_u32 %[cnt], %[cnt], 1\n\t"                                                           \
               "s_waitcnt lgkmcnt(0)\n\t"                                                                  \
               "v_cmp_ne_u32_e32 vcc, %[tag], %[t0]\n\t"                                                   \
               "s_and_b64 %[p0], vcc, %[fm0]\n\t"                                                          \
               "v_cmp_ne_u32_e32 vcc, %[tag], %[t1]\n\t"                                                   \
               "s_and_b64 %[p1], vcc, %[fm1]\n\t"                                                          \
               "s_or_b64 vcc, %[p0], %[p1]\n\t"                                                            \
               "s_cmp_lt_u32 %[cnt], 64\n\t"                                                               \
               "s_cbranch_vccz 2f\n\t"                                                                     \
               "s_cbranch_scc1 1b\n\t"                                                                     \
               "2:\n\t"                                                                                    \
               "s_or_b32 %[pl], vcc_lo, vcc_hi\n\t"                                                        \
               "s_mov_b32 m0, %[keep]"                                                                     \
               : [keep] "=&s"(keep), [cnt] "=&s"(cnt), [pl] "=&s"(pend_lo), [t0] "=&v"(tg0), [t1] "=&v"(tg1), [p0] "=&s"(p0m),  \
                 [p1] "=&s"(p1m), [ex] "=&s"(exec_saved), [k] "=&s"(gapk)                                  \
               : [s0] "v"(s0), [s1] "v"(s1), [d0] "s"(dst0), [d1] "s"(dst1), [o0] "v"(own0), [o1] "v"(own1), [tag] "s"(T),         \
                 [fm0] "s"(fm0), [fm1] "s"(fm1), [gap] "s"(poll_gap)                                       \
               : "vcc", "scc", "memory")
        RG_POLL;
        if (PROBE) pr_spins = cnt;
        if (__builtin_expect(pend_lo != 0u, 0)) {  // 64 rounds were not enough (or the run is being aborted): keep polling, bounded
          for (unsigned outer = 0;;) {
            const int ab = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ab != 0 || ++outer > (max_spins >> 4)) {
              dead = true;
              if (ab == 0) {  // the first to give up says which wait it was (the others leave through the abort flag)
                const unsigned long long pm = p0m | p1m;
                const int fl = __ffsll((long long)pm) - 1;
                const int fr = __shfl((p0m >> (t & 63)) & 1ull ? fs0 : fs1, fl, 64);
                const unsigned gs = (unsigned)__shfl((int)((p0m >> (t & 63)) & 1ull ? tg0 : tg1), fl, 64);
                if ((t & 63) == 0) {
                  __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  atomicOr(err, 2);
                  if (atomicCAS(&err[1], 0, 3) == 0) {
                    err[2] = region, err[3] = blk * k, err[4] = (int)(unsigned)pm, err[5] = (int)(unsigned)(pm >> 32), err[6] = fr;
                    err[7] = (int)gs, err[8] = (int)T, err[9] = (int)my_xcc, err[10] = (int)rg_hw_id();
                  }
                }
              }
              break;
            }
            RG_POLL;
            if (PROBE) pr_spins += cnt;
            if (pend_lo == 0u) break;
          }
        }
#undef RG_POLL
      }
      if (PROBE) pr_first = (unsigned)clock64();
      rg_barrier();
      if (PROBE) pr_t1 = (unsigned)clock64();
      // the lanes of the ring take their state over: {x, w1, w2} of depth 1 .. k-1, q of the edges of level >= 2
      {
        const float4 sa = *fa_p, sq = *fq_p;
        x = v_fa >= 0 ? sa.x : x, w12 = v_fa >= 0 ? v2f_t{sa.y, sa.z} : w12;
        q1 = e_fq >= 0 ? sq.x : q1, q23 = e_fq >= 0 ? v2f_t{sq.y, sq.z} : q23;
      }
    } else if (PROBE) {
      pr_t1 = pr_first = pr_t0;
    }
    // ---- kb steps inside the workgroup ------------------------------------------------------------------------------------
    const bool publishes = left > kb && !mute;  // another block follows: its ring is refreshed from what this one leaves
    const unsigned Tpub = tag0 + (unsigned)blk;
    char* const pub_base = xb_base + ((Tpub & 1u) ? par : 0u);
    unsigned pr_e = 0, pr_b1 = 0, pr_v = 0, pr_b2 = 0, pr_s = 0;
    if (PROBE) pr_s = (unsigned)clock64();
    // One step of the block.  FIRST / LAST are compile-time: the first step of a block reads a ring endpoint's bar record from its poll
    // slot, the last one publishes -- a lone wave issues one instruction per ~5 cycles, scalar ones included, so what a step does not
    // need is not in its code (tools/rg_step_cost.py: the loop around two empty phases cost 416 cycles per step when it decided all
    // that at run time).
    auto substep = [&](const int s, auto first_tag, auto last_tag) {
      constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
      // E phase: the edges of level <= kb + 1 - s (both endpoints exact after sub-step s - 1)
      const int lim = kb + 1 - s;
#ifdef FLAME_RG_DIAG
      if (!(dual & 16384))
#endif
      if (wlevel <= lim) {
        {  // (every lane of the wave: a lane past its level, or without an edge, computes values nobody reads -- its slots belong to
           //  vertices that are not computed in this step either, or are a spare entry of its own -- and is refreshed before it counts again)
          const float4 bi = *(FIRST ? bi_f : bi_p), bj = *(FIRST ? bj_f : bj_p);  // a bar record: {w1_bar, w2_bar, x_bar}
          const v2f_t wbi = {bi.x, bi.y}, wbj = {bj.x, bj.y};
          // dual update, cc:99-110
          float K1 = alpha * (bi.z - bj.z);
          const v2f_t m12 = P12 * wbi;
          K1 -= m12.x;
          K1 -= m12.y;
          const v2f_t K23 = beta * (wbi - wbj);
          const float q1r = q1 + p.step_q * K1;
          const v2f_t q23r = q23 + p.step_q * K23;
          q1 = __builtin_fminf(__builtin_fmaxf(q1r, -1.0f), 1.0f);  // proxNLTGV2Conj h:171-176 == clamp for finite q
          q23.x = __builtin_fminf(__builtin_fmaxf(q23r.x, -1.0f), 1.0f);
          q23.y = __builtin_fminf(__builtin_fmaxf(q23r.y, -1.0f), 1.0f);
          // both endpoints' shares of the primal scatter, cc:126-141, as ordered contributions: a -= b == a + (-b) exactly
          const float u1 = q1 * p.step_x;
          const v2f_t u23 = q23 * p.step_x;
          const float tt = u1 * alpha;
          const v2f_t v23 = u23 * beta, nv23 = u23 * nbeta, a12 = tt * D12;  // (u * -beta == -(u * beta) exactly)
#ifdef FLAME_RG_DIAG
          if (!(dual & 512))
#endif
          {
            // a slot is {a1, a2, b1, b2} + cx: the source's {tt dx, tt dy, -v2, -v3}, -tt; the target's {v2, v3, (-0, -0)}, +tt -- the
            // halves of a slot are register pairs as the packed arithmetic leaves them, the b half of a target slot keeps its -0.0
            *cs4 = make_float4(a12.x, a12.y, nv23.x, nv23.y);
            *cs1 = -tt;
            *reinterpret_cast<v2f_t*>(cd4) = v23;
            *cd1 = tt;
          }
          if (LAST && publishes && pubQ) publish(pub_base, offQ, q1, q23.x, q23.y, Tpub);  // q is final for this block: ahead of the V phase
          // NaN/Inf: the reference's FLAME_ASSERT h:174 (home lanes report; they are exact in every step)
          ok = ok && (__builtin_fabsf(q1r) <= 3.402823466e+38f) && (__builtin_fabsf(q23r.x) <= 3.402823466e+38f) &&
               (__builtin_fabsf(q23r.y) <= 3.402823466e+38f);
        }
      }
      unsigned pe0 = 0, pe1 = 0, pe2 = 0;
      if (PROBE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pe0 = (unsigned)clock64();
      }
      rg_barrier();
      if (PROBE) pe1 = (unsigned)clock64();
      // V phase: the vertices of depth <= kb - s
#ifdef FLAME_RG_DIAG
      if (!(dual & 32768))
#endif
      if (wvdepth <= kb - s) {
        {  // (every lane of the wave, as in the E phase: a ring lane past its depth is refreshed before it counts again)
          float X = x;
          v2f_t W = w12;
          // eight slots per round of loads (all sixteen reads issued before the first add); slots past a vertex's degree hold -0.0
#define RG_SLOTS8(J)                                                                                                     \
  {                                                                                                                      \
    const float4 c0 = cp[J], c1_ = cp[J + 1], c2 = cp[J + 2], c3 = cp[J + 3], c4_ = cp[J + 4], c5 = cp[J + 5], c6 = cp[J + 6],   \
                 c7 = cp[J + 7];                                                                                        \
    const float b0 = bp[J], b1 = bp[J + 1], b2 = bp[J + 2], b3 = bp[J + 3], b4 = bp[J + 4], b5 = bp[J + 5], b6 = bp[J + 6],     \
                b7 = bp[J + 7];                                                                                         \
    X = X + b0, W = (W + v2f_t{c0.x, c0.y}) + v2f_t{c0.z, c0.w};                                                        \
    X = X + b1, W = (W + v2f_t{c1_.x, c1_.y}) + v2f_t{c1_.z, c1_.w};                                                    \
    X = X + b2, W = (W + v2f_t{c2.x, c2.y}) + v2f_t{c2.z, c2.w};                                                        \
    X = X + b3, W = (W + v2f_t{c3.x, c3.y}) + v2f_t{c3.z, c3.w};                                                        \
    X = X + b4, W = (W + v2f_t{c4_.x, c4_.y}) + v2f_t{c4_.z, c4_.w};                                                    \
    X = X + b5, W = (W + v2f_t{c5.x, c5.y}) + v2f_t{c5.z, c5.w};                                                        \
    X = X + b6, W = (W + v2f_t{c6.x, c6.y}) + v2f_t{c6.z, c6.w};                                                        \
    X = X + b7, W = (W + v2f_t{c7.x, c7.y}) + v2f_t{c7.z, c7.w};                                                        \
  }
#define RG_SLOTS4(J)                                                                                                     \
  {                                                                                                                      \
    const float4 c0 = cp[J], c1_ = cp[J + 1], c2 = cp[J + 2], c3 = cp[J + 3];                                             \
    const float b0 = bp[J], b1 = bp[J + 1], b2 = bp[J + 2], b3 = bp[J + 3];                                               \
    X = X + b0, W = (W + v2f_t{c0.x, c0.y}) + v2f_t{c0.z, c0.w};                                                        \
    X = X + b1, W = (W + v2f_t{c1_.x, c1_.y}) + v2f_t{c1_.z, c1_.w};                                                    \
    X = X + b2, W = (W + v2f_t{c2.x, c2.y}) + v2f_t{c2.z, c2.w};                                                        \
    X = X + b3, W = (W + v2f_t{c3.x, c3.y}) + v2f_t{c3.z, c3.w};                                                        \
  }
#ifdef FLAME_RG_DIAG
          if (dual & 1024) {
            X = X + data, W = W + v2f_t{thr, thr};
          } else if (dual & 4096) {  // (what an X-only lane would do: the cx chain alone)
            X = (((((((X + bp[0]) + bp[1]) + bp[2]) + bp[3]) + bp[4]) + bp[5]) + bp[6]) + bp[7];
          } else if (dual & 8192) {  // (what a W-only lane would do)
            const float4 c0 = cp[0], c1_ = cp[1], c2 = cp[2], c3 = cp[3], c4_ = cp[4], c5 = cp[5], c6 = cp[6], c7 = cp[7];
            W = (W + v2f_t{c0.x, c0.y}) + v2f_t{c0.z, c0.w}, W = (W + v2f_t{c1_.x, c1_.y}) + v2f_t{c1_.z, c1_.w};
            W = (W + v2f_t{c2.x, c2.y}) + v2f_t{c2.z, c2.w}, W = (W + v2f_t{c3.x, c3.y}) + v2f_t{c3.z, c3.w};
            W = (W + v2f_t{c4_.x, c4_.y}) + v2f_t{c4_.z, c4_.w}, W = (W + v2f_t{c5.x, c5.y}) + v2f_t{c5.z, c5.w};
            W = (W + v2f_t{c6.x, c6.y}) + v2f_t{c6.z, c6.w}, W = (W + v2f_t{c7.x, c7.y}) + v2f_t{c7.z, c7.w};
          } else
#endif
          if (MAXT <= 512) {
            RG_SLOTS8(0)
            if (wdeg > 8) {  // (a wave with a vertex of more than eight edges: four more slots per round)
              RG_SLOTS4(8)
              if (wdeg > 12) {
                RG_SLOTS4(12)
                for (int j = 16; j < wdeg; j += 4) RG_SLOTS4(j)
              }
            }
          } else {
            RG_SLOTS4(0)
            RG_SLOTS4(4)
            for (int j = 8; j < wdeg; j += 8) {
              RG_SLOTS4(j)
              RG_SLOTS4(j + 4)
            }
          }
#undef RG_SLOTS4
#undef RG_SLOTS8
          // proxL1 (cc:147-151, h:179-197), extragradient (cc:160-171)
          const float diff = X - data, x_dn = X - thr, x_up = X + thr;
          float xn = (diff < -thr) ? x_up : data;
          xn = (diff > thr) ? x_dn : xn;
          xn = (xn < p.x_min) ? p.x_min : xn;
          xn = (xn > p.x_max) ? p.x_max : xn;
          float nb = xn + p.theta * (xn - x);
          nb = (nb < p.x_min) ? p.x_min : nb;
          nb = (nb > p.x_max) ? p.x_max : nb;
          const v2f_t wbn = W + p.theta * (W - w12);
          *bar_w = make_float4(wbn.x, wbn.y, nb, 0.f);
          if (LAST && publishes) {  // ahead of the barrier: the records are on their way while the workgroup gathers
            if (pubB) publish(pub_base, offB, wbn.x, wbn.y, nb, Tpub);
            if (pubA) publish(pub_base, offA, xn, W.x, W.y, Tpub);
          }
          x_prev = x, w_prev = w12;  // step()'s prev copy, cc:37-42
          x = xn, w12 = W;
          xb = nb, wb12 = wbn;
        }
      }
      if (PROBE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pe2 = (unsigned)clock64();
      }
      rg_barrier();
      if (PROBE) {
        const unsigned pe3 = (unsigned)clock64();
        pr_e += pe0 - pr_s, pr_b1 += pe1 - pe0, pr_v += pe2 - pe1, pr_b2 += pe3 - pe2, pr_s = pe3;
      }
    };
    {
      const std::true_type yes;
      const std::false_type no;
      if (kb == 1) {
        substep(1, yes, yes);
      } else {
        substep(1, yes, no);
        for (int s = 2; s < kb; ++s) substep(s, no, no);
        substep(kb, no, yes);
      }
    }
    left -= kb;
    if (PROBE) {
      const unsigned pr_t2 = (unsigned)clock64();
      if (t == 0 && probe) {  // {poll done, xcc id, wait, compute, poll rounds, block start, 100 MHz clock, steps of the block,
                              //  E phase, barrier 1, V phase, barrier 2 (sums over the block's steps, wave 0), store ack, lone load, -, -}
        unsigned* o = probe + ((size_t)region * ((n_iters + k - 1) / k) + blk) * kRgProbeWords;
        o[0] = pr_first - pr_t0, o[1] = my_xcc, o[2] = pr_t1 - pr_t0, o[3] = pr_t2 - pr_t1;
        o[4] = pr_spins, o[5] = pr_t0, o[6] = 0u, o[7] = (unsigned)kb;
        o[8] = pr_e, o[9] = pr_b1, o[10] = pr_v, o[11] = pr_b2, o[12] = pr_ack, o[13] = pr_load;
      }
      pr_t0 = (unsigned)clock64();
    }
    ++blk;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA in flight when the wave ends
  if (dead) return;  // (the run is void: err says so, the host takes it back)

  // ---- write back: owned vertices, home edges ---------------------------------------------------------------------------------
  if (v_owned) {
    vstate_out[pv] = make_float4(x, w12.x, w12.y, data);
    bar_out[pv] = make_float4(xb, wb12.x, wb12.y, 0.0f);
    vprev[pv] = make_float4(x_prev, w_prev.x, w_prev.y, 0.0f);
    float* const export_out = tail->export_out;
    float* const photo_err = tail->photo.err;
    if (export_out || photo_err) {
      const int o = perm[pv];  // the caller's vertex index
      if (o >= 0 && export_out) export_out[o] = x * tail->export_scale;
      if (o >= 0 && photo_err) {
        const PhotoFuse& photo = tail->photo;
        photo_err[o] = photo_residual_at(photo.pos[o], x * photo.graph_scale, photo.geo, photo.ref, photo.cmp, photo.rows, photo.cols,
                                         photo.step, photo.border);
      }
    }
  }
  if (e_home) {
    const float4 o = make_float4(q1, q23.x, q23.y, beta);
    hq_out[slot_s] = o, hq_out[slot_d] = o;
    if (!ok) atomicOr(err, 1);
  }
}

}  // namespace

static const void* rg_kernel(bool probe, int block_threads) {
  if (block_threads <= 512) return probe ? (const void*)k_persistent_rg<true, 512> : (const void*)k_persistent_rg<false, 512>;
  return probe ? (const void*)k_persistent_rg<true, kRgMaxThreads> : (const void*)k_persistent_rg<false, kRgMaxThreads>;
}

// Workgroups of k_persistent_rg the runtime keeps on one CU for this layout (LDS and registers; the planner wants all regions resident).
int rg_blocks_per_cu(const RgArgs& a, bool probe) {
  int n = 0;
  const void* fn = rg_kernel(probe, a.block_threads);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, a.block_threads, a.lds_bytes) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int launch_persistent_rg(const FusedArgs& f, const RgArgs& a, const SolverParams& p, int parity_in, unsigned tag0, int n_iters,
                         unsigned max_spins, int dual, const RunTail* tail, unsigned* probe, bool cooperative, hipStream_t stream) {
  if (a.n_regions <= 0 || n_iters <= 0) return (int)hipSuccess;
  const int rpx = (a.n_regions + 7) / 8;
  const dim3 grid((unsigned)(rpx * 8)), block((unsigned)a.block_threads);
  RgArgs aa = a;
  const int4* hrec = f.hrec;
  const float4* hq = f.hq;
  const float4* vstate = f.vstate;
  float4* hq_out = f.hq_out;
  float4* vstate_out = f.vstate_out;
  const float2* vaux = f.vaux;
  const float4* bin = f.bar[parity_in];
  float4* bout = f.bar[parity_in ^ 1];
  float4* vprev = f.vprev;
  SolverParams pp = p;
  int* err = f.err;
  int* abort_flag = f.abort_flag;
  const int32_t* perm = f.perm;
  void* args[] = {&aa, &hrec, &hq, &vstate, &hq_out, &vstate_out, &vaux, &bin, &bout, &vprev, &tag0, &n_iters, &max_spins, &pp, &err,
                  &abort_flag, &perm, &tail, &dual, &probe};
  const void* fn = rg_kernel(probe != nullptr, a.block_threads);
  if (a.lds_bytes > 65536u) {
    static bool raised[4] = {false, false, false, false};
    const int inst = (probe ? 1 : 0) + (a.block_threads <= 512 ? 0 : 2);
    if (!raised[inst]) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
      raised[inst] = true;
    }
  }
  if (cooperative) return (int)hipLaunchCooperativeKernel(fn, grid, block, args, a.lds_bytes, stream);
  return (int)hipExtLaunchKernel(fn, grid, block, args, a.lds_bytes, stream, nullptr, f.stop_event, 0);
}

// Loads this translation unit's code object (the runtime does that at the first use of one of its kernels: several milliseconds that
// flame_nltgv2_create takes on itself so that the first frame does not).
void warm_module_persistent_rg() {
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, rg_kernel(false, 512)) != hipSuccess) (void)hipGetLastError();
}

}  // namespace flame_hip
