// nltgv2_persistent.hip -- the persistent single-launch kernels of the NLTGV2-L1 solver for gfx950 (MI355X, CDNA4): n steps in ONE
// launch as pure dataflow between resident waves -- k_persistent_pv (a patch per wave, a lane per half-edge: the dominant kernel
// of the bench); k_persistent_tv (a vertex per lane) is in nltgv2_persistent_tv.hip -- their launcher and the residency rule.
// Arithmetic and its citations: nltgv2_device.hpp; compiled with -ffp-contract=off like nltgv2_kernels.hip.
#include "nltgv2_device.hpp"

namespace flame_hip {

namespace {

// ------------------------------------------------------------------------------------------------
// Persistent run: ONE launch == n_iters reference step()s, as pure dataflow between resident waves.
//
// Why: a dependent kernel boundary costs ~3.5 us on this part (measured: trivial dependent kernels replay at that period) while
// one step of a 640x480 graph is < 1 us of work, so one-launch-per-step is launch bound.  Inside one launch the only thing a
// step needs from other waves is the (x_bar, w1_bar, w2_bar) of graph neighbours, and a 16-byte write-through record crosses
// the chip in ~0.45-0.55 us (tools/hop_bench.hip).  The protocol (both persistent kernels, this one and k_persistent_tv):
//
//   * the records, weights and the private (q1, q2, q3) copy of every half-edge stay in REGISTERS for the whole run (loaded
//     once from the SELL arrays);
//   * every vertex that another wave reads publishes ONE naturally aligned 16-byte record {x_bar, w1_bar, w2_bar, tag = step
//     number} with ONE write-through (sc1) dwordx4 store; a consumer re-reads the record (L1-bypassing sc1 access) until the
//     tag equals the step it needs.  The data is its own flag: no fences, no flag words, no grid barrier.  (A lane's aligned
//     16-byte access is a single request inside one cache line; tearing between value and tag has not been observed on gfx950,
//     would show up as a bit mismatch in the parity tests, and FLAME_NLTGV2_OPT_VERIFY_RECORDS checks for it at run time.)
//   * two record buffers alternate by step parity: a vertex can only overwrite its step-s record with step s+2 after ALL its
//     neighbours published s+1, i.e. after they consumed s.  Tags grow monotonically over the context's lifetime and every
//     launch starts from a fresh tag, so stale records never match.  Each buffer holds a remote copy (write-through) and a copy
//     for readers on the same XCD (their L2), chosen per record from the true XCC ids the waves publish at the start.
//   * the primal accumulation of a vertex follows the reference's edge order exactly (cc:120-142): an ORDERED sum, identity
//     contributions are -0.0f (x + -0.0f == x for every x, including both zeros).
//
// All waves must be resident (the first launch of a topology is cooperative: the runtime checks the grid; the planner knows
// what a CU really holds); every wait is bounded and reports through `err`.  The run is transactional: it reads hq / vstate /
// bar_in and writes hq_out / vstate_out / bar_out (the other copies), so the host can take an expired run back and redo it per
// step.  (Round 1's first form, one lane per half-edge with every lane polling its own neighbour from memory -- k_persistent_he
// -- was retired in round 3: the patch-per-wave form below runs everything it ran, faster; docs/DESIGN_r3.md section 4.)
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// Persistent run, patch-per-wave form ("pv"): a lane per half-edge, organised around
// what the in-kernel probes of round 2 measured (profiles/r02_persistent/, docs/DESIGN_r3.md section 4):
//   * a lone wave issues one instruction every ~4.5 cycles and an LDS round trip costs ~120: a step costs what the
//     instructions and LDS trips BETWEEN a record arriving and the next one leaving cost (round 1's k_persistent_he: ~1400 cycles,
//     of which the DPP ripple 750; here ~850);
//   * a hand-off costs a store, a load round trip, and however much later than the arrival the consumer looks; one
//     record per half-edge polled with blocking loads made that ~2200 cycles per step;
//   * the period of the whole lock-step network is set by its SLOWEST adjacent pair, not by the average wave.
//
// Layout (E) of nltgv2_pack.hpp: a wave owns a compact Morton patch of ~10 vertices (64 half-edge lanes, a vertex's
// lanes contiguous, ascending edge id).
//   exchange   Neighbours inside the patch meet in LDS.  Every DISTINCT foreign record the patch needs is fetched by
//              exactly one lane (sorted by record id; records are numbered in walk order, so a producer's records share
//              cache lines) with an LDS-DMA load (global_load_lds_dwordx4 sc1: lane i's 16 bytes land in fetch slot i of
//              the step's LDS area, no register waits for them), issued round after round from a six-instruction loop
//              that also reads every lane's neighbour record (tag word first, then the 16 bytes) from LDS and leaves when
//              all tags are the step's.  A poll that finds an old record only rewrites the slot with the bytes it already
//              holds: a producer cannot publish step s+2 into that parity before this patch has published s+1, i.e.
//              consumed s.  A vertex is published to memory only if another patch reads it.
//   step       Straight-line code: the per-role selects of the dual update are folded into signed per-lane constants
//              (exact: IEEE negation, a*(-b) == -(a*b)), the w1/w2 halves run as packed-f32 pairs.  The ordered
//              accumulation (cc:120-142: ascending edge id) runs across the lanes of a vertex towards its first lane, which
//              alone holds its state (patches are row-packed: a vertex's lanes inside one 16-lane row): shift j adds the
//              contribution of lane first + j with the DPP row shift of a plain add, EXEC masks (moved in by the scalar
//              unit, computed once per launch) say which heads a shift may still write; the other lanes take (x_bar, w_bar)
//              of their vertex from the record its head leaves in LDS.  ~26 cycles per shift, no LDS in the hand-off path.
//              A vertex of more than 16 edges starts its patch at lane 0 and fills whole rows: after row 0 the running sums
//              move (v_readlane / v_writelane) to the first lane of the next row, which adds its own row the same way, and
//              back to lane 0 at the end -- the same additions in the same order, ~500 cycles per further row.
//   hand-off   Between a record arriving and the next one leaving every instruction costs ~5 cycles, needed or not, and
//              a taken branch ~16: the NaN check, the prev copies and the LDS record come after the publish, the record
//              verification is a template flag, the wait is one statement whose common exit falls through (DESIGN.md 4,
//              "The hand-off path").
// Protocol (tags, two parity buffers, remote / XCD-local copies chosen from the true XCC ids, bounded waits,
// transactional outputs): see above.
// ------------------------------------------------------------------------------------------------
constexpr unsigned kWgActiveBit = 1u << 25, kWgValidBit = 1u << 26, kWgPublishBit = 1u << 27, kWgHeadBit = 1u << 28;
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_wave_sync() {  // LDS operations of one wave are processed in issue order
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ unsigned read_hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}

// Which wait of a persistent run expired, for the host's trace (FLAME_NLTGV2_TRACE) and flame_nltgv2_info: the first wave
// to give up leaves {which wait, patch, step, lanes still waiting, the foreign record of the first of them, tag seen / wanted,
// XCC and HW ids} in err[1..10] (err[0] stays the flag word).  which: 1 rotation word, 2 XCC table, 3 a step's records.
__device__ __forceinline__ void report_expired(int* err, int which, int wg, int it, unsigned long long pend, int frid, unsigned seen,
                                               unsigned want) {
  if (atomicCAS(&err[1], 0, which) == 0) {
    err[2] = wg, err[3] = it, err[4] = (int)(unsigned)pend, err[5] = (int)(unsigned)(pend >> 32), err[6] = frid;
    err[7] = (int)seen, err[8] = (int)want, err[9] = (int)read_xcc_id(), err[10] = (int)read_hw_id();
  }
}

// amdgpu_num_sgpr(92): 90 SGPRs as built -> 96 + the trap handler's 16 = 112 per wave, SEVEN waves per SIMD really resident (at the
// compiler's own choice, 106, it is six: "Round 3" in docs/DESIGN_r3.md section 4); the scalar spills this costs stay outside the hand-off path
// (640x480: 0.962 against 0.962 us per iteration, profiles/r03_priority.txt (7)) and a 1080p frame's 25 patches per CU fit one launch.
// OPEN (round 6): a run that goes on until the host needs the state.  n_iters is then an upper bound; ONE patch (the middle one of the
// launch) looks at a word the host sets to this run's tag0 (a 4-byte copy on a stream of its own; a request for an earlier run means nothing, so the
// word is never cleared) every kOpenCheck iterations and, when it says so, publishes the iteration every
// patch leaves at -- its own plus kOpenMargin, more than any patch can be ahead of it (a patch is ahead of another by at most their
// distance in the patch graph) -- in err[12] as tag0 + iteration (tags grow from run to run: a stale word of an earlier run is below
// this run's tag0 and means nothing).  Every patch reads that word every kOpenCheck iterations.  A patch that saw the word too late has no neighbours left to wait for: its
// wait expires and the run is taken back and redone like any other (nltgv2_run.hip finish()).  The patch that decides leaves the
// number of iterations done in err[13] (tag0 + n) on its way out.
constexpr unsigned kOpenMargin = 128u, kOpenCheck = 64u;  // (both even: an open run does an even number of iterations)
template <bool PROBE, bool VERIFY, bool OPEN = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(92)))
k_persistent_pv(const int wg_begin, const int n_wgs, const int wgs_per_xcd, const int lcap, const int slab_slots,
                const int32_t* __restrict__ wg_slot, const int32_t* __restrict__ wg_vid,
                const uint32_t* __restrict__ wg_meta, const int32_t* __restrict__ wg_nbr,
                const int32_t* __restrict__ wg_fetch, const int32_t* __restrict__ wg_info, const int4* hrec,
                const float4* hq, const float4* vstate,
                float4* hq_out, float4* vstate_out, const float2* vaux, const float4* bar_in, float4* bar_out, float4* vprev,
                void* xbuf, const int rec_bytes, const int dual_arg, const unsigned tag0, const int n_iters,
                const unsigned max_spins_arg, const int poll_gap_arg, const SolverParams p,
                int* __restrict__ err, int* __restrict__ abort_flag, const int32_t* __restrict__ perm,
                const RunTail* __restrict__ tail, unsigned* __restrict__ probe, char* place_pool,
                const int32_t* __restrict__ rec_off, const int rec_off_stride, unsigned* rot_word) {
  extern __shared__ float4 lds[];
  constexpr int T = 64;
  const unsigned max_spins = max_spins_arg & 0x7fffffffu;
  const int dual = dual_arg & 1, verify = VERIFY ? dual_arg >> 1 : 0;  // (bit 0: same-XCD exchange through L2; bits 1..: record verification, compiled in only where asked for)
  const int poll_gap = poll_gap_arg & 255, pv_presleep = (poll_gap_arg >> 8) & 255;
  const int lane = (int)threadIdx.x;
  int b = blockIdx.x;
  if (rot_word) {
    // The dispatcher deals workgroups to the XCDs round-robin, but goes on where the previous dispatch stopped: block 0 lands
    // on XCD r, block i on (i + r) % 8.  Record placement ranks pages by the XCDs a record really travels between, so the
    // blocks are renumbered to make group k of the layout the one on XCD k: block 0 says where it is, everybody rotates by
    // that (a bijection of the grid, whatever r is; if the dispatch was not a plain rotation the groups are merely less
    // well placed, as they would be without this).
    const unsigned want = (tag0 & 0x0fffffffu) << 4;
    if (b == 0 && lane == 0) __hip_atomic_store(rot_word, want | read_xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v = 0;
    for (unsigned spins = 0;; ++spins) {
      v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(rot_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if ((v & ~15u) == want) break;
      if (spins > (max_spins_arg & 0x7fffffffu)) {
        if (lane == 0) {
          __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicOr(err, 2);
          report_expired(err, 1, (int)blockIdx.x, -1, 0ull, -1, v, want);
        }
        return;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    b += (int)(v & 7u);
    if (b >= (int)gridDim.x) b -= (int)gridDim.x;
  }
  const int xcd = b & 7, idx = b >> 3;
  if (idx >= wgs_per_xcd) return;
  if (xcd * wgs_per_xcd + idx >= n_wgs) return;
  const int wg = wg_begin + xcd * wgs_per_xcd + idx;  // this launch covers patches [wg_begin, +n_wgs)
  const int rid_base = wg_info[4 * wg], n_fetch = wg_info[4 * wg + 1];
  const int count_flags = wg_info[4 * wg + 2];
  if ((count_flags & 0xffff) == 0) return;                 // (a patch without a vertex: nothing to do)
  if (unsigned* const pg = tail->progress) {               // (trace runs only) when this patch started, in us of the 100 MHz clock
    if (lane == 0) pg[n_wgs + (wg - wg_begin)] = (unsigned)(wall_clock64() / 100u) | 1u;
  }
  // the patch's largest degree: how many shifts the ordered accumulation runs (above 16: over how many rows)
  const int stride = wg_info[4 * wg + 3];
  // LDS map, float4 units: [rec area 0: lcap local + 64 fetch slots | rec area 1 | slab_slots (unused: 0) | spare 64]
  const int rec_stride = lcap + T;
  const int o_ovfA = 2 * rec_stride + slab_slots;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(xbuf);
  // Placed records (nltgv2_layout.hip, "record placement"): the remote copy of a record that another XCD reads lives in a
  // pool of 4 KB pages instead of the linear buffer, on a page whose home memory channel is close to both XCDs (the
  // hand-off through the fabric takes 0.39-0.66 us depending on the page); rec_off[parity][record] is its byte offset in
  // the pool, negative = the linear place.

  // Memory side of the exchange: TWO buffers by step parity, each [remote copy S bytes | same-XCD copy S bytes], then the XCC
  // table (a record of step s is only overwritten by s + 2 after every reader published s + 1, i.e. consumed s).
  constexpr int kPar = 2;
  const int S = rec_bytes, par = 2 * rec_bytes, tab_off = kPar * par;
  const unsigned xcc_want = (tag0 & 0x0fffffffu) << 4;
  const unsigned p0 = tag0 & 1u;  // parity of the first step: its records live in area p0

  const size_t hl = (size_t)wg * T + lane;
  const unsigned meta = wg_meta[hl];
  const int slot = wg_slot[hl];
  const int pv = wg_vid[hl];
  const int nbr_code = wg_nbr[hl];
  const int frid = (lane < n_fetch) ? wg_fetch[hl] : -1;  // the foreign record this lane fetches
  const int loc = (int)((meta >> 13) & 2047u);
  const bool active = (meta & kWgActiveBit) != 0u;
  const bool valid = (meta & kWgValidBit) != 0u, publishes = (meta & kWgPublishBit) != 0u;
  // The lane that holds a vertex's state, publishes its record and writes it back: its FIRST (the sum runs across the lanes
  // towards it).  A head takes part in shift j while j < its degree; the other lanes of a vertex only serve as sources, and a lane
  // without a half-edge is disabled altogether (a DPP read of a disabled lane leaves the destination as it is)
  const bool state_lane = (meta & kWgHeadBit) != 0u;
  const unsigned degx = (meta & kWgHeadBit) ? ((meta >> 6) & 127u) : (active ? 255u : 0u);
  // ... as a destination; the lanes written by shift j, for the shifts every patch runs
  const unsigned long long rm1 = __ballot(degx > 1u), rm2 = __ballot(degx > 2u), rm3 = __ballot(degx > 3u), rm4 = __ballot(degx > 4u),
                           rm5 = __ballot(degx > 5u), rm6 = __ballot(degx > 6u), rm7 = __ballot(degx > 7u), rm8 = __ballot(degx > 8u);
  // whose record this lane waits for: its half-edge's other end; a lane without a half-edge looks at its own vertex's
  // record, a lane without a vertex at the patch's first vertex -- both carry the step's tag from the start
  const int nbr_idx = active ? ((nbr_code < 0) ? lcap + (nbr_code & 0x7fffffff) : nbr_code) : (valid ? loc : 0);

  int4 rec = make_int4(0, 0, 0, 0);
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    rec = hrec[slot];
    q = hq[slot];
  }
  const bool is_target = rec.x < 0;
  const float alpha = __int_as_float(rec.y), dx = __int_as_float(rec.z), dy = __int_as_float(rec.w);
  const float beta = q.w;
  float q1 = q.x;
  v2f_t q23 = {q.y, q.z};
  // signed per-lane constants: see k_persistent_wg
  const float as = is_target ? -alpha : alpha, bs = is_target ? -beta : beta, ac = is_target ? alpha : -alpha;
  const v2f_t P12 = {alpha * dx, alpha * dy};
  // (a lane without a half-edge has all-zero constants and q = +0 for good: its (cx, a, b) come out as (-0, a, -0) with
  //  a = (-0) * C2 -- C2 = +0 there makes that -0 as well, so the head of an isolated vertex adds exactly nothing to its own state)
  const v2f_t C2 = !active ? v2f_t{0.0f, 0.0f} : is_target ? v2f_t{beta, beta} : v2f_t{-dx, -dy};
  const float nbeta = -beta;

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
  bool timed_out = n_fetch > T;  // (the host never launches such a layout in this form)
  bool torn = false;

  const int my_off = (rid_base + loc) << 4;
  // (a lane without a vertex writes its record to a spare entry of its own, which nobody reads)
  const int rec_w = valid ? loc : o_ovfA + lane, rec_wstride = valid ? rec_stride : 0;

  // fetch slots: tag 0 is never a live tag
  lds[lcap + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  lds[rec_stride + lcap + lane] = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- where this lane's polls read: the remote copy of its foreign record, or the copy in this XCD's L2 ------------
  int off0 = (frid >= 0) ? (frid << 4) : 0;
  bool fetch_remote = frid >= 0;
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
          if (lane == 0) report_expired(err, 2, wg, -1, pm, ff, gs, xcc_want);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (!timed_out && frid >= 0 && (g0 & 15u) == my_xcc) off0 += S, fetch_remote = false;
    }
  }
  const unsigned long long fetch_mask = __ballot(frid >= 0);  // the lanes with a fetch duty (the wave runs with all 64 lanes)
  const bool mute = (max_spins_arg >> 31) != 0u && wg == wg_begin;  // test hook: FLAME_NLTGV2_OPT_FAULT_INJECT
  const bool pub_lane = state_lane && publishes;
  const char* const xb_base = static_cast<const char*>(xbuf);
  const bool placed = place_pool != nullptr;
  int pub0 = -1, pub1 = -1;
  const char* src0 = xb_base + off0;
  const char* src1 = xb_base + off0 + par;
  if (placed) {
    if (pub_lane) pub0 = rec_off[rid_base + loc], pub1 = rec_off[rec_off_stride + rid_base + loc];
    if (fetch_remote) {
      const int o0 = rec_off[frid], o1 = rec_off[rec_off_stride + frid];
      if (o0 >= 0) src0 = place_pool + o0;
      if (o1 >= 0) src1 = place_pool + o1;
    }
  }
  // where the remote copy of this lane's record goes, by parity: its place in the pool, or the linear one (two buffers:
  // one address per parity, no decision left for the step)
  char* const xb_w = static_cast<char*>(xbuf);
  char* const pa0 = pub0 >= 0 ? place_pool + pub0 : xb_w + my_off;
  char* const pa1 = pub1 >= 0 ? place_pool + pub1 : xb_w + my_off + par;
  auto publish = [&](const v4i_t o, char* pa, const int so) {  // (pub lanes only)
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(pa), "v"(o) : "memory");
    if (dual) __builtin_amdgcn_raw_buffer_store_b128(o, rx, my_off + S, so, 0);
  };
  {
    lds[(p0 ? rec_wstride : 0) + rec_w] = make_float4(xb, wb12.x, wb12.y, __uint_as_float(tag0));  // area A
    if (pub_lane && !mute) {
      v4i_t o;
      o.x = __float_as_int(xb), o.y = __float_as_int(wb12.x), o.z = __float_as_int(wb12.y), o.w = (int)tag0;
      publish(o, p0 ? pa1 : pa0, (int)(tag0 & (kPar - 1)) * par);
    }
  }
  lds_wave_sync();
  const unsigned lds_addr0 = (unsigned)(size_t)(lds);  // LDS byte address of the dynamic array
  unsigned pr_t0 = 0;
  if (PROBE) pr_t0 = (unsigned)clock64();

  // One step.  rd_nbr: LDS byte address of the neighbour's record; dst: LDS byte address of the step's fetch slots;
  // src: this lane's poll address; wr_rec: float4 index where the own record of the next step goes; so_out: memory offset
  // of the parity published.
  auto step = [&](const unsigned s, const unsigned rd_nbr, const unsigned dst, const int wr_rec, const int fetch_area, const int it,
                  const char* const src2, char* const pub2, const int rd_rec) {
    // Only the head of a vertex computes its state; the other lanes take (x_bar, w_bar) of their vertex from the
    // record the head left in LDS at the end of the previous step (read here, ahead of the wait: off the critical path)
    float4 own = make_float4(0.f, 0.f, 0.f, 0.f);
    own = lds[rd_rec];
    const int so_out = (int)((s + 1u) & (kPar - 1)) * par;  // (wave-uniform)
    // two buffers: the step's parity is fixed at the call site (src2: where this lane polls, pub2: where it publishes)
    const char* const src = src2;
    // ---- wait for the neighbours' records of step s ----------------------------------------------------------------
    v4f_t nbv;  // the tag word is read first, the record after it: a tag that matches vouches for the payload
    unsigned rounds = 0;
    // no record can be here sooner than one hand-off after its producer's previous publish: polls before that only load
    // the L2s and the fabric (pv_presleep x 64 cycles, fixed per launch)
    for (int z = 0; z < pv_presleep; ++z) __builtin_amdgcn_s_sleep(1);
    {
      // up to 64 rounds per statement: one LDS-DMA load per fetch lane that still waits (M0 = destination base, saved and
      // restored inside the statement; `pn` = those lanes: a lane whose own slot shows the step's tag stops re-loading it,
      // unless the launch asks for the un-narrowed poll), an optional s_sleep, then every lane's neighbour record from LDS;
      // vcc = lanes still waiting.  One statement for all pacing variants (two scalar flags), and the common exit falls
      // straight through into the step: the instructions after the last record's arrival are the ones that count.
      // The statement polls at issue priority 0 and leaves at 3 (see below).
      unsigned cnt, keep, pend_lo, tagv, tagf, gapk;
      unsigned long long pnarrow, exec_saved;  // (EXEC is saved and put back by the statement itself, not assumed to be all ones)
      const unsigned own_slot = dst + 16u * (unsigned)lane;
      // (pacing: bit 0 = pause between rounds, bits 4..7 = its length - 1 in s_sleep 1 units, bit 1 = narrowed re-loads)
      const unsigned f_sleep = (poll_gap & 1) ? 1u + (((unsigned)poll_gap >> 4) & 15u) : 0u, f_narrow = (unsigned)((poll_gap >> 1) & 1);
#define PV_POLL_U                                                                                         \
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
               "ds_read_b32 %[t], %[ra] offset:12\n\t"                                                  \
               "ds_read_b32 %[t2], %[fa] offset:12\n\t"                                                 \
               "ds_read_b128 %[nb], %[ra]\n\t"                                                          \
               "s_add_u32 %[cnt], %[cnt], 1\n\t"                                                        \
               "s_waitcnt lgkmcnt(0)\n\t"                                                               \
               "v_cmp_ne_u32_e32 vcc, %[tag], %[t2]\n\t"                                                \
               "s_and_b64 %[pn], vcc, %[fm]\n\t"                                                        \
               "s_cmp_eq_u32 %[fn], 0\n\t"                                                              \
               "s_cselect_b64 %[pn], %[fm], %[pn]\n\t"                                                  \
               "v_cmp_ne_u32_e32 vcc, %[tag], %[t]\n\t"                                                 \
               "s_cmp_lt_u32 %[cnt], 64\n\t"                                                            \
               "s_cbranch_vccz 2f\n\t"                                                                  \
               "s_cbranch_scc1 1b\n\t"                                                                  \
               "2:\n\t"                                                                                 \
               "s_setprio 3\n\t"                                                                        \
               "s_or_b32 %[pl], vcc_lo, vcc_hi\n\t"                                                     \
               "s_mov_b32 m0, %[keep]"                                                                   \
               : [keep] "=&s"(keep), [cnt] "=&s"(cnt), [pl] "=&s"(pend_lo), [nb] "=&v"(nbv), [t] "=&v"(tagv), \
                 [t2] "=&v"(tagf), [pn] "=&s"(pnarrow), [k] "=&s"(gapk), [ex] "=&s"(exec_saved)            \
               : [src] "v"(src), [dst] "s"(dst), [ra] "v"(rd_nbr), [fa] "v"(own_slot), [tag] "s"(s), [fm] "s"(fetch_mask), \
                 [fs] "s"(f_sleep), [fn] "s"(f_narrow)                                                     \
               : "vcc", "scc", "memory")
      PV_POLL_U;
      rounds += cnt;
      if (__builtin_expect(pend_lo != 0u, 0)) {  // 64 rounds were not enough (or the run is being aborted): keep polling, bounded
        for (unsigned outer = 0;;) {
          const int ab = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ab != 0 || ++outer > (max_spins >> 4)) {
            timed_out = true;
            if (ab == 0) {  // (the first to give up: the others leave through the abort flag)
              const unsigned long long pm = __ballot(tagv != s);
              const int fl = __ffsll((long long)pm) - 1;
              const int ni = __shfl(nbr_idx, fl, 64);
              const int ff = ni >= lcap ? __shfl(frid, ni - lcap, 64) : -2 - ni;  // foreign record id, or -2 - (local index)
              const unsigned gs = (unsigned)__shfl((int)tagv, fl, 64);
              if (lane == 0) report_expired(err, 3, wg, it, pm, ff, gs, s);
            }
            break;
          }
          PV_POLL_U;
          rounds += cnt;
          if (pend_lo == 0u) break;  // every lane saw the step's tag
        }
      }
#undef PV_POLL_U
    }
    if (VERIFY && verify && !timed_out) {
      // Every fetch lane reads its foreign record once more, with an ordinary load, and compares all four dwords with
      // what the LDS-DMA left in its slot: a record is final once its tag is visible, so a difference means a torn
      // 16-byte access (memory side or LDS side) -- reported, the run is taken back and redone per step.
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
    unsigned pr_t1 = 0;
    if (PROBE) pr_t1 = (unsigned)clock64();
    // Issue priority (round 3): the waves of a CU are in different phases -- most of them polling, a few computing the step
    // that their neighbours are waiting for.  At equal priority the arbiter serves the OLDEST ready wave, i.e. a polling
    // loop as readily as the hand-off path; from the wait's exit (s_setprio 3 inside the statement, in the place of an
    // instruction it could do without) to the publish this wave goes first, polling waves last: -3 % at 720p, -7 % at 15-23
    // patches per CU, nothing at 4 per CU (one wave per SIMD); a level by patch length on top of it gained nothing.
    xb = own.x, wb12 = v2f_t{own.y, own.z};
    // ---- dual update of this half-edge's private q copy (cc:99-110) ------------------------------
    const v2f_t nbw = {nbv.y, nbv.z};
    const float d0 = xb - nbv.x;
    const v2f_t d12 = wb12 - nbw;
    const v2f_t wbi = is_target ? nbw : wb12;  // the SOURCE vertex's (w1_bar, w2_bar)
    float K1 = as * d0;
    const v2f_t m12 = P12 * wbi;
    K1 -= m12.x;
    K1 -= m12.y;
    const v2f_t K23 = bs * d12;
    const float q1r = q1 + p.step_q * K1;
    const v2f_t q23r = q23 + p.step_q * K23;
    q1 = __builtin_fminf(__builtin_fmaxf(q1r, -1.0f), 1.0f);
    q23.x = __builtin_fminf(__builtin_fmaxf(q23r.x, -1.0f), 1.0f);
    q23.y = __builtin_fminf(__builtin_fmaxf(q23r.y, -1.0f), 1.0f);
    // ---- this endpoint's share of the primal scatter (cc:126-141) as ordered contributions -------
    const float u1 = q1 * p.step_x;
    const v2f_t u23 = q23 * p.step_x;
    const float cx = u1 * ac;
    const v2f_t M2 = is_target ? u23 : v2f_t{cx, cx};
    const v2f_t a12 = M2 * C2;
    v2f_t b12 = u23 * nbeta;
    b12 = is_target ? v2f_t{-0.0f, -0.0f} : b12;
    float X = x;
    v2f_t Wa = w12;
    {
      // ---- ordered accumulation across the lanes of the vertex, towards its head ------------------------------------
      // The head (lane `first`) starts with (x + c_0, (w + a_0) + b_0) of its own half-edge; shift j = 1, 2, ... adds the
      // contribution of lane first + j, taken with a DPP row shift (the patch is row-packed: a vertex's lanes share a 16-lane
      // row).  Every lane stays enabled as a SOURCE; as a DESTINATION a head is masked out once j reaches its degree (it
      // would pick up the next vertex's lanes), the other lanes compute values nobody reads; a lane without a half-edge is
      // disabled (a DPP read of a disabled lane leaves the destination as it is).  The masks are computed once per launch and
      // moved to EXEC by the scalar unit (a v_cmpx per shift stalls the DPP adds behind it: +22 cycles per shift).  ~26 cycles
      // per shift (tools/ripple_bench.hip: 264 cycles for 8 contributions; an LDS slab read back by every lane took 580).
      float W1 = (w12.x + a12.x) + b12.x, W2 = (w12.y + a12.y) + b12.y;
      X = x + cx;
#define PV_ADDS(J)                                                                                    \
  "v_add_f32_dpp %[X], %[cx], %[X] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                      \
  "v_add_f32_dpp %[W1], %[a1], %[W1] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_add_f32_dpp %[W2], %[a2], %[W2] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_add_f32_dpp %[W1], %[b1], %[W1] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_add_f32_dpp %[W2], %[b2], %[W2] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define PV_RM(J, M) "s_mov_b64 exec, %[" #M "]\n\t" PV_ADDS(J)
#define PV_ROW(L)                                                                                     \
  "s_nop 1\n\t"                                                                                       \
  "v_readlane_b32 %[s0], %[X], 0\n\t"                                                                 \
  "v_readlane_b32 %[s1], %[W1], 0\n\t"                                                                \
  "v_readlane_b32 %[s2], %[W2], 0\n\t"                                                                \
  "s_nop 3\n\t"                                                                                       \
  "v_writelane_b32 %[X], %[s0], " #L "\n\t"                                                           \
  "v_writelane_b32 %[W1], %[s1], " #L "\n\t"                                                          \
  "v_writelane_b32 %[W2], %[s2], " #L "\n\t"                                                          \
  "s_sub_u32 %[s0], %[md], " #L "\n\t"                                                                \
  "s_min_u32 %[s0], %[s0], 16\n\t"                                                                    \
  "s_bfm_b64 exec, %[s0], " #L "\n\t"                                                                 \
  "s_nop 4\n\t"                                                                                       \
  "v_add_f32 %[X], %[X], %[cx]\n\t"                                                                   \
  "v_add_f32 %[W1], %[W1], %[a1]\n\t"                                                                 \
  "v_add_f32 %[W2], %[W2], %[a2]\n\t"                                                                 \
  "v_add_f32 %[W1], %[W1], %[b1]\n\t"                                                                 \
  "v_add_f32 %[W2], %[W2], %[b2]\n\t"                                                                 \
  "s_nop 1\n\t"                                                                                       \
  PV_ADDS(1)                                                                                          \
  "s_cmp_le_u32 %[s0], 2\n\t"                                                                         \
  "s_cbranch_scc1 1" #L "f\n\t"                                                                       \
  PV_ADDS(2) PV_ADDS(3)                                                                               \
  "s_cmp_le_u32 %[s0], 4\n\t"                                                                         \
  "s_cbranch_scc1 1" #L "f\n\t"                                                                       \
  PV_ADDS(4) PV_ADDS(5) PV_ADDS(6) PV_ADDS(7)                                                         \
  "s_cmp_le_u32 %[s0], 8\n\t"                                                                         \
  "s_cbranch_scc1 1" #L "f\n\t"                                                                       \
  PV_ADDS(8) PV_ADDS(9) PV_ADDS(10) PV_ADDS(11) PV_ADDS(12) PV_ADDS(13) PV_ADDS(14) PV_ADDS(15)       \
  "1" #L ":\n\t"                                                                                      \
  "s_nop 1\n\t"                                                                                       \
  "v_readlane_b32 %[s0], %[X], " #L "\n\t"                                                            \
  "v_readlane_b32 %[s1], %[W1], " #L "\n\t"                                                           \
  "v_readlane_b32 %[s2], %[W2], " #L "\n\t"                                                           \
  "s_nop 3\n\t"                                                                                       \
  "v_writelane_b32 %[X], %[s0], 0\n\t"                                                                \
  "v_writelane_b32 %[W1], %[s1], 0\n\t"                                                               \
  "v_writelane_b32 %[W2], %[s2], 0\n\t"
      // shifts 1..7: straight line, masks from registers (a shift past a head's degree finds it masked out; a patch whose
      // largest degree is below 8 runs the spare shifts on nothing).  A vertex of more than 8 edges has its row to itself
      // (nltgv2_pack.hpp, WaveFit): from shift 8 on ONE mask -- those heads and the lanes that only serve as sources -- does
      // for all shifts, the lanes past such a vertex's last edge being idle (-0.0 contributions); three exits by the patch's
      // largest degree instead of one per shift (a branch costs 16 cycles, a shift 26: tools/ripple_bench)
      unsigned long long exec_saved;  // (the masks below narrow EXEC; it is put back as it was found, not assumed to be all ones)
      unsigned row_s0, row_s1, row_s2;  // (scalar temporaries of the further rows of a vertex of more than 16 edges)
      asm volatile("s_mov_b64 %[ex], exec\n\t"
                   "s_nop 1\n\t"
                   PV_RM(1, m1) PV_RM(2, m2) PV_RM(3, m3) PV_RM(4, m4) PV_RM(5, m5) PV_RM(6, m6) PV_RM(7, m7)
                   "s_cmp_le_u32 %[md], 8\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   PV_RM(8, m8) PV_ADDS(9)
                   "s_cmp_le_u32 %[md], 10\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   PV_ADDS(10) PV_ADDS(11)
                   "s_cmp_le_u32 %[md], 12\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   PV_ADDS(12) PV_ADDS(13) PV_ADDS(14) PV_ADDS(15)
                   "s_cmp_le_u32 %[md], 16\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   // A vertex of more than 16 edges (rare: a hull vertex of a Delaunay graph): it begins at lane 0 and fills
                   // rows 0 .. ceil(md / 16) - 1 of this patch (nltgv2_pack.hpp, WaveFit).  Lane 0 now holds the sums over its
                   // first 16 edges; the first lane of each further row takes them over, adds its own edge and then its row by
                   // the same shifts (only that row's lanes enabled: min(16, md - L) of them; exits by that count), and hands the sums
                   // back to lane 0.
                   PV_ROW(16)
                   "s_cmp_le_u32 %[md], 32\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   PV_ROW(32)
                   "s_cmp_le_u32 %[md], 48\n\t"
                   "s_cbranch_scc1 9f\n\t"
                   PV_ROW(48)
                   "9:\n\t"
                   "s_mov_b64 exec, %[ex]"
                   : [X] "+v"(X), [W1] "+v"(W1), [W2] "+v"(W2), [ex] "=&s"(exec_saved), [s0] "=&s"(row_s0), [s1] "=&s"(row_s1), [s2] "=&s"(row_s2)
                   : [cx] "v"(cx), [a1] "v"(a12.x), [a2] "v"(a12.y), [b1] "v"(b12.x), [b2] "v"(b12.y), [md] "s"(stride),
                     [m1] "s"(rm1), [m2] "s"(rm2), [m3] "s"(rm3), [m4] "s"(rm4), [m5] "s"(rm5), [m6] "s"(rm6), [m7] "s"(rm7), [m8] "s"(rm8)
                   : "scc");
#undef PV_ROW
#undef PV_RM
#undef PV_ADDS
      Wa = v2f_t{W1, W2};
    }
    // ---- vertex update: proxL1 (cc:147-151, h:179-197), extragradient (cc:160-171) --------------------------
    // (both shifted values up front and two selects: as branches this was three exec-masked blocks in the hand-off path)
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
    // (between a record arriving and the next one leaving every instruction counts, needed or not: a lone wave issues
    //  one per ~5 cycles -- so what the publish does not need comes after it)
    ok = ok && (__builtin_fabsf(q1r) <= 3.402823466e+38f) && (__builtin_fabsf(q23r.x) <= 3.402823466e+38f) &&
         (__builtin_fabsf(q23r.y) <= 3.402823466e+38f);  // NaN/Inf: the reference's FLAME_ASSERT h:174
    x_prev = x, w_prev = w12;  // step()'s prev copy, cc:37-42
    x = xn, w12 = Wa;
    xb = nb, wb12 = wbn;
    if (PROBE) {
      const unsigned pr_t2 = (unsigned)clock64();
      if (lane == 0 && probe) {  // {hw id, xcc id, wait, compute, poll rounds, step start, 100 MHz clock, slab stride | fetch lanes << 8}
        unsigned* o = probe + ((size_t)wg * n_iters + it) * 8;
        o[0] = read_hw_id(), o[1] = read_xcc_id(), o[2] = pr_t1 - pr_t0, o[3] = pr_t2 - pr_t1;
        o[4] = rounds, o[5] = pr_t0, o[6] = (unsigned)wall_clock64(), o[7] = (unsigned)stride | ((unsigned)n_fetch << 8);
      }
      pr_t0 = pr_t2;
    }
  };

  // Two steps per trip with the parities fixed: area A holds the records of the first step, B those of the next.
  const int areaA = p0 ? rec_stride : 0, areaB = p0 ? 0 : rec_stride;
  const unsigned rdA_nbr = lds_addr0 + 16u * (unsigned)(areaA + nbr_idx), rdB_nbr = lds_addr0 + 16u * (unsigned)(areaB + nbr_idx);
  const unsigned dstA = __builtin_amdgcn_readfirstlane(lds_addr0 + 16u * (unsigned)(areaA + lcap));
  const unsigned dstB = __builtin_amdgcn_readfirstlane(lds_addr0 + 16u * (unsigned)(areaB + lcap));
  const int wrA_rec = (valid ? areaA : 0) + rec_w, wrB_rec = (valid ? areaB : 0) + rec_w;
  // the records of step tag0 + even are in memory buffer p0 ("A"), those of the odd steps in the other
  const char* const srcA = p0 ? src1 : src0;
  const char* const srcB = p0 ? src0 : src1;
  char* const pubA = p0 ? pa1 : pa0;
  char* const pubB = p0 ? pa0 : pa1;
  int it = 0;
  if (OPEN) {
    const bool decides = wg == wg_begin + n_wgs / 2;
    const unsigned* const stop_req = tail->stop_req;  // (device memory: the host's copy lands there)
    unsigned* const stop_word = reinterpret_cast<unsigned*>(err) + 12;
    unsigned stop_at = 0u;  // tag0 + the iteration to leave at, once known
    for (; it + 1 < n_iters && !timed_out; it += 2) {
      // Every kOpenCheck iterations -- all patches at the same ones: the network runs in lock step, so the ~0.5 us this load takes are
      // spent by everybody at once, 1-2 % of the time -- the word is looked at; the deciding patch first looks at the host's request.
      // (Asked for every trip and looked at a trip later it cost 17 %: the compiler waits for the publish stores in front of the
      //  load, and the step's polls never wait for vmcnt, so nothing hides it.)
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
          if (w > tag0 && w - tag0 <= (unsigned)n_iters + kOpenMargin) stop_at = w;  // (a word of an earlier run is below this run's tag0)
        }
      }
      if (stop_at != 0u && stop_at - tag0 <= (unsigned)it) break;
      step(tag0 + (unsigned)it, rdA_nbr, dstA, wrB_rec, areaA + lcap, it, srcA, pubB, wrA_rec);
      if (timed_out) break;
      step(tag0 + (unsigned)it + 1u, rdB_nbr, dstB, wrA_rec, areaB + lcap, it + 1, srcB, pubA, wrB_rec);
    }
    if (decides && lane == 0 && !timed_out) reinterpret_cast<unsigned*>(err)[13] = tag0 + (unsigned)it;
  } else {
    for (; it + 1 < n_iters && !timed_out; it += 2) {
      step(tag0 + (unsigned)it, rdA_nbr, dstA, wrB_rec, areaA + lcap, it, srcA, pubB, wrA_rec);
      if (timed_out) break;
      step(tag0 + (unsigned)it + 1u, rdB_nbr, dstB, wrA_rec, areaB + lcap, it + 1, srcB, pubA, wrB_rec);
    }
    if (it < n_iters && !timed_out) step(tag0 + (unsigned)it, rdA_nbr, dstA, wrB_rec, areaA + lcap, it, srcA, pubB, wrA_rec);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA in flight when the wave ends

  if (timed_out) {
    if (lane == 0) {
      __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicOr(err, torn ? 4 : 2);
      unsigned* const pg = tail->progress;  // (trace runs only) the step this patch was in when it left
      if (pg) pg[wg - wg_begin] = 0x80000000u | (unsigned)(it + 1);
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
      const int o = perm[pv];  // the caller's vertex index
      if (o >= 0 && export_out) export_out[o] = x * tail->export_scale;
      if (o >= 0 && photo_err) {
        const PhotoFuse& photo = tail->photo;
        photo_err[o] = photo_residual_at(photo.pos[o], x * photo.graph_scale, photo.geo, photo.ref, photo.cmp, photo.rows,
                                         photo.cols, photo.step, photo.border);
      }
    }
  }
  if (active) hq_out[slot] = make_float4(q1, q23.x, q23.y, beta);
  if (!ok && active) atomicOr(err, 1);
}

}  // namespace

// Patches of k_persistent_pv (one wave each) that are REALLY co-resident per CU for this layout.
//
// The runtime's occupancy query (and with it the cooperative-launch check) over-reports on this part: the hardware adds 16
// scalar registers to every wave's allocation for the trap handler, which neither hipOccupancyMaxActiveBlocksPerMultiprocessor
// nor the compiler's "Occupancy [waves/SIMD]" remark knows about.  Measured with tools/residency_probe.hip (profiles/
// r03_residency.txt: 64-thread blocks, 3.8 KB of LDS, <= 72 VGPRs): a kernel of 27 SGPRs keeps 32 waves per CU, one of 86-94
// keeps 28 where the query says 32, one of 102-106 keeps 24 where it says 28.  That is round 2's open question: the row-packed
// instance of this kernel (65 VGPRs, 106 SGPRs) passed the cooperative check at 25-28 patches per CU, 24 became resident, and
// the others only started when the first expired waits freed their slots.  So: waves per SIMD = min(512 / VGPRs rounded up to 8,
// 800 / (SGPRs rounded up to 16, + 16), 8), from the register counts of the instances as built (tests/test_abi.py re-derives them
// from the compiler's resource report and fails when an instance outgrows its row here); the runtime's answer still bounds it
// from above (it knows the LDS use, which varies with the layout).
int pv_real_waves_per_simd(bool verify_or_probe) {
  // {VGPRs, SGPRs} -> waves: plain {<= 72, <= 96} -> min(7, 7); verify / probe {<= 88, <= 96} -> min(5, 7)
  return verify_or_probe ? 5 : 7;
}

int pv_patches_per_cu(const FusedArgs& a, bool verify) {
  if (!a.wg_rowpack) return 0;  // (not row-packed: the lane-per-half-edge form's layout)
  const size_t ldsv = 16u * (size_t)(2 * (a.wg_lcap + 64) + a.wg_slab_slots + 64);
  int n = 0;
  const void* fv = verify ? (const void*)k_persistent_pv<false, true> : (const void*)k_persistent_pv<false, false>;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fv, 64, ldsv) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n < 4 * pv_real_waves_per_simd(verify) ? n : 4 * pv_real_waves_per_simd(verify);
}

// Persistent run (single launch).  form 2 = vertex per
// lane (k_persistent_tv), form 3 = patch per wave (k_persistent_pv).  Returns the hipError_t unchanged (e.g. hipErrorCooperativeLaunchTooLarge)
// so the caller can fall back to per-step launches.
int launch_persistent_run(const FusedArgs& a, const SolverParams& p, int form, int wave_begin, int n_waves,
                          int parity_in, unsigned tag0, int n_iters, int waves_per_block, unsigned max_spins,
                          int presleep, int dual, int tv_static_in_lds, int xcds, const RunTail* tail, bool cooperative,
                          hipStream_t stream) {
  if (n_waves <= 0 || n_iters <= 0) return (int)hipSuccess;
  // Workgroup b runs on XCD b & 7.  With xcds < 8 only the first `xcds` XCDs get waves (the workgroups of the
  // others find no work and exit), which keeps a small graph's whole exchange inside fewer L2s.
  if (xcds < 1 || xcds > 8) xcds = 8;
  int wpx = (n_waves + xcds - 1) / xcds;
  const int bpx = (wpx + waves_per_block - 1) / waves_per_block;
  const dim3 grid((unsigned)(bpx * 8)), block((unsigned)(64 * waves_per_block));
  if (form != 2 && form != 3) return (int)hipErrorInvalidValue;
  const int32_t* i0 = a.tv_slot;
  const int32_t* i1 = a.tv_vid;
  const uint32_t* i2 = a.tv_meta;
  const void* i3 = (const void*)a.tv_wave;
  const int4* hrec = a.hrec;
  const float4* hq = a.hq;
  const float4* vstate = a.vstate;
  float4* hq_out = a.hq_out;
  float4* vstate_out = a.vstate_out;
  const float2* vaux = a.vaux;
  const float4* bin = a.bar[parity_in];
  float4* bout = a.bar[parity_in ^ 1];  // always the other buffer: the input of a failed run stays intact
  float4* vprev = a.vprev;
  void* xbuf = a.xbuf;
  int rec_bytes = (a.n_rec > a.n_slices * 64 ? a.n_rec : a.n_slices * 64) * 16;
  SolverParams pp = p;
  int* err = a.err;
  int* abort_flag = a.abort_flag;
  const int32_t* perm = a.perm;
  void* args[] = {&wave_begin, &n_waves, &wpx, &i0, &i1, &i2, &i3, &hrec, &hq, &vstate, &hq_out, &vstate_out, &vaux,
                  &bin, &bout, &vprev, &xbuf, &rec_bytes, &dual, &tag0, &n_iters, &max_spins, &presleep, &pp, &err,
                  &abort_flag, &perm, &tail};
  if (form == 3) {  // patch-per-wave form: n_waves / wave_begin count PATCHES (one wave each)
    int wgx = (n_waves + xcds - 1) / xcds;
    const dim3 gv((unsigned)(wgx * 8)), bv(64u);
    int lcap = a.wg_lcap, slab_slots = a.wg_slab_slots, poll_gap = a.wg_poll_gap;
    const int32_t *w0 = a.wg_slot, *w1 = a.wg_vid, *w3 = a.wg_nbr, *w4 = a.wg_fetch, *w5 = a.wg_info;
    const uint32_t* w2 = a.wg_meta;
    unsigned* probe = a.probe;
    char* place_pool = (a.rec_off && wave_begin == 0 && xcds == 8) ? a.place_pool : nullptr;
    const int32_t* rec_off = a.rec_off;
    int rec_off_stride = a.rec_off_stride;
    unsigned* rot_word = place_pool ? a.rot_word : nullptr;
    const unsigned ldsv = 16u * (unsigned)(2 * (lcap + 64) + slab_slots + 64);
    void* vargs[] = {&wave_begin, &n_waves, &wgx, &lcap, &slab_slots, &w0, &w1, &w2, &w3, &w4, &w5, &hrec, &hq, &vstate,
                     &hq_out, &vstate_out, &vaux, &bin, &bout, &vprev, &xbuf, &rec_bytes, &dual, &tag0, &n_iters,
                     &max_spins, &poll_gap, &pp, &err, &abort_flag, &perm, &tail, &probe, &place_pool, &rec_off, &rec_off_stride, &rot_word};
    const bool vr = (dual >> 1) != 0;  // record verification asked for
    if (!a.wg_rowpack) return (int)hipErrorInvalidConfiguration;  // (the planner never asks: the kernel runs row-packed patches)
    const void* fv = probe ? (const void*)k_persistent_pv<true, true>
                           : vr ? (const void*)k_persistent_pv<false, true>
                                : a.open_run ? (const void*)k_persistent_pv<false, false, true> : (const void*)k_persistent_pv<false, false>;
    if (cooperative) return (int)hipLaunchCooperativeKernel(fv, gv, bv, vargs, ldsv, stream);
    return (int)hipExtLaunchKernel(fv, gv, bv, vargs, ldsv, stream, nullptr, a.stop_event, 0);
  }
  unsigned lds_bytes = 0u;
  const void* fn = persistent_tv_kernel(tv_static_in_lds != 0, waves_per_block, &lds_bytes);
  // The first launch of a topology is cooperative: the runtime verifies that the whole grid is
  // resident (hipErrorCooperativeLaunchTooLarge otherwise).  The same grid is then launched plainly
  // (identical residency, ~15 us less launch overhead per call).
  if (cooperative) return (int)hipLaunchCooperativeKernel(fn, grid, block, args, lds_bytes, stream);
  return (int)hipExtLaunchKernel(fn, grid, block, args, lds_bytes, stream, nullptr, a.stop_event, 0);
}

// Loads this translation unit's code object (the runtime does that at the first use of one of its kernels: several milliseconds that
// flame_nltgv2_create takes on itself so that the first frame does not).
namespace {
__global__ void k_warm_cooperative() {}
}  // namespace
void warm_module_persistent(hipStream_t stream, bool cooperative) {
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, (const void*)k_persistent_pv<false, false>) != hipSuccess) (void)hipGetLastError();
  // ... and the runtime's one-time set-up of cooperative launches and of the occupancy query (6 ms in the first run of a process,
  // tools/cold_start.py): one empty cooperative grid -- a kernel of its own, so that a profile's statistics of the solver kernels hold
  // solver launches only; on the context's own (high-priority) stream, see flame_nltgv2_create
  FusedArgs a;
  a.wg_rowpack = 1, a.wg_lcap = 16;
  (void)pv_patches_per_cu(a, false);
  void* no_args[] = {nullptr};
  if (cooperative && hipLaunchCooperativeKernel((const void*)k_warm_cooperative, dim3(8), dim3(64), no_args, 0, stream) != hipSuccess) (void)hipGetLastError();
}

}  // namespace flame_hip
