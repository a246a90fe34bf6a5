// tools/net_bench.hip -- the record exchange of k_persistent_pv WITHOUT the solver: a lock-step network of one-wave patches on a
// patch grid, each waiting every step for ~20 tagged 16-byte records of its ~8 neighbour patches and publishing ~8 of its own,
// with the kernel's own poll statement (LDS-DMA loads re-issued without a wait, tags read back from LDS), its two stores per
// record (write-through copy for other XCDs, plain copy for this XCD's L2), its two parity buffers and a dependent VALU chain of
// a chosen length in the place of the step's arithmetic.  period - chain = the hand-off as the NETWORK pays it; every knob below
// removes or changes one suspected contributor (not product code; profiles/r06_handoff.txt is written from its output):
//
//   net_bench [key=value ...] -- keys: px py (patch grid, default 28x28 = 784 patches, 98 per XCD in a 4x2 arrangement of tiles),
//     steps, chain (trips of 16 dependent v_add_f32 per step, ~95 cycles each), jitter (per-patch extra trips, uniform 0..jitter), coupled (1: tiles
//     exchange across their borders through the write-through copies; 0: eight disjoint nets, one per XCD), nbrs (8 = edge + corner
//     neighbours, 4 = edge only, 2 = left/right, 1 = a ring), redge / rcorner (records read from an edge / corner neighbour),
//     layout (0: records where the producer keeps them, as the product; 1: same-XCD copies in a mailbox of the CONSUMER, one slot per
//     fetch lane), order (0: write-through store first, as the product; 1: same-XCD store(s) first), far_store (0: no write-through
//     store when nobody on another XCD reads), poll (0: the product's statement; 1: near and far lanes by two load instructions;
//     2: blocking VGPR loads + ds_write; 3: product statement with s_waitcnt vmcnt(0) after every load), narrow (0/1), gap, stamp (1: log),
//     spad (bytes added to the distance between a record's write-through copy and its same-XCD copy: 0 = a multiple of 4 KB apart)
//
// Build: hipcc --offload-arch=gfx950 -O3 -o net_bench net_bench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e = (x);                                                          \
    if (e != hipSuccess) {                                                       \
      printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__);      \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

constexpr int RMAX = 4;  // mailbox layout: copies a producer lane writes at most (readers beyond that poll the producer's own copy)

struct Args {
  const int* tile_patch;  // [8][per_xcd]: the patches of XCD k
  int* xcd_count;         // [8]
  int per_xcd, n_patches;
  const int* src_off;     // [patch][64] byte offset (parity 0) this lane polls, -1 = no fetch duty
  const int* src_far;     // [patch][64] 1 = the record lives on another XCD (poll the write-through copy)
  const int* pub_far;     // [patch][64] byte offset of the write-through copy, -1 = none
  const int* pub_near;    // [patch][64][RMAX] byte offsets of the same-XCD copies (layout 0: one), -1 = none
  const int* n_fetch;     // [patch]
  const int* chain;       // [patch]
  char* base;
  int par;                // bytes between the two parity images
  int steps;
  unsigned tag0;
  unsigned* log;          // [patch][steps][8] = {t_enter, t_exit, t_pub (s_memtime of THIS CU: the CUs' counters are not aligned), rounds,
                          //                      exit, publish by the device-wide 100 MHz clock}
  long long* wall;        // [patch] 100 MHz ticks for steps 1..steps-1; < 0: a wait expired
  int* where;             // [patch] {xcc}
  int f_sleep, f_narrow;
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

template <int ORDER, int POLL, bool STAMP>
__global__ void __launch_bounds__(64) k_net(const Args a) {
  __shared__ v4f lds[2 * 64];
  const int lane = threadIdx.x;
  const unsigned xcc = xcc_id();
  int idx = 0;
  if (lane == 0) idx = atomicAdd(&a.xcd_count[xcc & 7], 1);
  idx = __builtin_amdgcn_readfirstlane(idx);
  if (idx >= a.per_xcd) return;  // (the dispatcher deals blocks round-robin: does not happen; the others would then expire)
  const int patch = a.tile_patch[(xcc & 7) * a.per_xcd + idx];
  if (patch < 0) return;
  if (lane == 0) a.where[patch] = (int)xcc;
  const size_t hl = (size_t)patch * 64 + lane;
  const int so = a.src_off[hl];
  const bool far = a.src_far[hl] != 0;
  const int pf = a.pub_far[hl];
  int pn[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) pn[r] = a.pub_near[hl * RMAX + r];
  const int nf = a.n_fetch[patch];
  const int chain_n = a.chain[patch];
  const unsigned long long fetch_mask = __ballot(so >= 0);
  const unsigned long long near_mask = __ballot(so >= 0 && !far), far_mask = __ballot(so >= 0 && far);
  const char* const src0 = a.base + (so >= 0 ? so : 0);
  const char* const src1 = src0 + a.par;
  lds[lane] = v4f{0.f, 0.f, 0.f, 0.f};
  lds[64 + lane] = v4f{0.f, 0.f, 0.f, 0.f};
  const unsigned lds0 = (unsigned)(size_t)lds;
  // every lane looks at one fetch slot (the product: at its half-edge's other end); without fetch duties the patch is alone
  const int look = nf > 0 ? lane % nf : 0;
  const unsigned tag0 = a.tag0, p0 = tag0 & 1u;
  float acc = (float)lane;
  auto publish = [&](float v, unsigned tag, int parity) {
    v4i o;
    o.x = __float_as_int(v), o.y = lane, o.z = patch, o.w = (int)tag;
    char* const pb = a.base + parity * a.par;
    if (ORDER == 0) {
      if (pf >= 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(pb + pf), "v"(o) : "memory");
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (pn[r] >= 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(pb + pn[r]), "v"(o) : "memory");
    } else {
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (pn[r] >= 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(pb + pn[r]), "v"(o) : "memory");
      if (pf >= 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(pb + pf), "v"(o) : "memory");
    }
  };
  publish(acc, tag0, (int)p0);
  if (nf == 0) lds[p0 * 64] = v4f{0.f, 0.f, 0.f, __uint_as_float(tag0)};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long w0 = 0;
  bool expired = false;
  const unsigned f_sleep = (unsigned)a.f_sleep, f_narrow = (unsigned)a.f_narrow;
  for (int it = 0; it < a.steps && !expired; ++it) {
    const unsigned s = tag0 + (unsigned)it;
    const int area = (int)(s & 1u);
    if (it == 1) w0 = wall_clock64();
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + 16u * (unsigned)(area * 64));
    const unsigned rd = lds0 + 16u * (unsigned)(area * 64 + look);
    const unsigned own_slot = dst + 16u * (unsigned)lane;
    const char* const src = area ? src1 : src0;
    unsigned t_enter = 0, t_exit = 0, t_pub = 0, w_exit = 0, w_pub = 0;
    if (STAMP) t_enter = (unsigned)clock64();
    unsigned rounds = 0;
    if (nf > 0) {
      unsigned cnt = 0, keep, pend_lo = 1, tagv, tagf, gapk;
      unsigned long long pnarrow, exec_saved;
      v4f nbv;
      for (unsigned outer = 0; pend_lo != 0u; ++outer) {
        if (outer > (1u << 12)) { expired = true; break; }
        if (POLL == 0 || POLL == 3) {
          asm volatile("s_setprio 0\n\t"
                       "s_mov_b64 %[ex], exec\n\t"
                       "s_mov_b32 %[keep], m0\n\t"
                       "s_mov_b32 m0, %[dst]\n\t"
                       "s_mov_b32 %[cnt], 0\n\t"
                       "s_mov_b64 %[pn], %[fm]\n\t"
                       "1:\n\t"
                       "s_mov_b64 exec, %[pn]\n\t"
                       "global_load_lds_dwordx4 %[src], off sc1\n\t"
                       "s_mov_b64 exec, %[ex]\n\t"
                       ".if %[wv] == 1\n\ts_waitcnt vmcnt(0)\n\t.endif\n\t"
                       "s_mov_b32 %[k], %[fs]\n\t"
                       "4:\n\t"
                       "s_cmp_eq_u32 %[k], 0\n\t"
                       "s_cbranch_scc1 3f\n\t"
                       "s_sleep 1\n\t"
                       "s_sub_u32 %[k], %[k], 1\n\t"
                       "s_branch 4b\n\t"
                       "3:\n\t"
                       "ds_read_b32 %[t], %[ra] offset:12\n\t"
                       "ds_read_b32 %[t2], %[fa] offset:12\n\t"
                       "ds_read_b128 %[nb], %[ra]\n\t"
                       "s_add_u32 %[cnt], %[cnt], 1\n\t"
                       "s_waitcnt lgkmcnt(0)\n\t"
                       "v_cmp_ne_u32_e32 vcc, %[tag], %[t2]\n\t"
                       "s_and_b64 %[pn], vcc, %[fm]\n\t"
                       "s_cmp_eq_u32 %[fn], 0\n\t"
                       "s_cselect_b64 %[pn], %[fm], %[pn]\n\t"
                       "v_cmp_ne_u32_e32 vcc, %[tag], %[t]\n\t"
                       "s_cmp_lt_u32 %[cnt], 64\n\t"
                       "s_cbranch_vccz 2f\n\t"
                       "s_cbranch_scc1 1b\n\t"
                       "2:\n\t"
                       "s_setprio 3\n\t"
                       "s_or_b32 %[pl], vcc_lo, vcc_hi\n\t"
                       "s_mov_b32 m0, %[keep]"
                       : [keep] "=&s"(keep), [cnt] "=&s"(cnt), [pl] "=&s"(pend_lo), [nb] "=&v"(nbv), [t] "=&v"(tagv), [t2] "=&v"(tagf),
                         [pn] "=&s"(pnarrow), [k] "=&s"(gapk), [ex] "=&s"(exec_saved)
                       : [src] "v"(src), [dst] "s"(dst), [ra] "v"(rd), [fa] "v"(own_slot), [tag] "s"(s), [fm] "s"(fetch_mask), [fs] "s"(f_sleep),
                         [fn] "s"(f_narrow), [wv] "n"(POLL == 3 ? 1 : 0)
                       : "vcc", "scc", "memory");
        } else if (POLL == 1) {  // near and far lanes by two instructions (far first: the near data are the ones in a hurry)
          unsigned long long pnf;
          asm volatile("s_setprio 0\n\t"
                       "s_mov_b64 %[ex], exec\n\t"
                       "s_mov_b32 %[keep], m0\n\t"
                       "s_mov_b32 m0, %[dst]\n\t"
                       "s_mov_b32 %[cnt], 0\n\t"
                       "s_mov_b64 %[pn], %[nm]\n\t"
                       "s_mov_b64 %[pf], %[fm]\n\t"
                       "1:\n\t"
                       "s_mov_b64 exec, %[pn]\n\t"
                       "global_load_lds_dwordx4 %[src], off sc1\n\t"
                       "s_mov_b64 exec, %[pf]\n\t"
                       "global_load_lds_dwordx4 %[src], off sc1\n\t"
                       "s_mov_b64 exec, %[ex]\n\t"
                       "ds_read_b32 %[t], %[ra] offset:12\n\t"
                       "ds_read_b32 %[t2], %[fa] offset:12\n\t"
                       "ds_read_b128 %[nb], %[ra]\n\t"
                       "s_add_u32 %[cnt], %[cnt], 1\n\t"
                       "s_waitcnt lgkmcnt(0)\n\t"
                       "v_cmp_ne_u32_e32 vcc, %[tag], %[t2]\n\t"
                       "s_and_b64 %[pn], vcc, %[nm]\n\t"
                       "s_and_b64 %[pf], vcc, %[fm]\n\t"
                       "v_cmp_ne_u32_e32 vcc, %[tag], %[t]\n\t"
                       "s_cmp_lt_u32 %[cnt], 64\n\t"
                       "s_cbranch_vccz 2f\n\t"
                       "s_cbranch_scc1 1b\n\t"
                       "2:\n\t"
                       "s_setprio 3\n\t"
                       "s_or_b32 %[pl], vcc_lo, vcc_hi\n\t"
                       "s_mov_b32 m0, %[keep]"
                       : [keep] "=&s"(keep), [cnt] "=&s"(cnt), [pl] "=&s"(pend_lo), [nb] "=&v"(nbv), [t] "=&v"(tagv), [t2] "=&v"(tagf),
                         [pn] "=&s"(pnarrow), [pf] "=&s"(pnf), [ex] "=&s"(exec_saved)
                       : [src] "v"(src), [dst] "s"(dst), [ra] "v"(rd), [fa] "v"(own_slot), [tag] "s"(s), [nm] "s"(near_mask), [fm] "s"(far_mask)
                       : "vcc", "scc", "memory");
          (void)gapk;
        } else {  // POLL == 2: blocking loads into registers, the record goes to LDS by ds_write, one more LDS trip for the look
          v4i g = {0, 0, 0, 0};
          cnt = 0;
          for (;;) {
            if (so >= 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(src) : "memory");
            ++cnt;
            const bool pend = so >= 0 && (unsigned)g.w != s;
            if (!__any(pend) || cnt >= 64) { pend_lo = __any(pend) ? 1u : 0u; break; }
          }
          if (so >= 0) lds[area * 64 + lane] = v4f{__int_as_float(g.x), __int_as_float(g.y), __int_as_float(g.z), __int_as_float(g.w)};
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          nbv = lds[area * 64 + look];
          (void)keep, (void)tagv, (void)tagf, (void)gapk, (void)pnarrow, (void)exec_saved;
        }
        rounds += cnt;
      }
      acc += nbv.x;
    }
    if (STAMP) t_exit = (unsigned)clock64(), w_exit = (unsigned)wall_clock64();
    // the step's arithmetic: a dependent chain, 16 additions per trip (~75 cycles + ~20 for the trip)
    for (int k = 0; k < chain_n; ++k)
      asm volatile("v_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\t"
                   "v_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\t"
                   "v_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\t"
                   "v_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0" : "+v"(acc));
    if (STAMP) t_pub = (unsigned)clock64(), w_pub = (unsigned)wall_clock64();
    publish(acc, s + 1u, (int)((s + 1u) & 1u));
    __builtin_amdgcn_s_setprio(0);
    if (nf == 0) lds[((s + 1u) & 1u) * 64] = v4f{0.f, 0.f, 0.f, __uint_as_float(s + 1u)};
    if (STAMP && lane == 0) {
      unsigned* o = a.log + ((size_t)patch * a.steps + it) * 8;
      o[0] = t_enter, o[1] = t_exit, o[2] = t_pub, o[3] = rounds, o[4] = w_exit, o[5] = w_pub;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) a.wall[patch] = expired ? -1 : wall_clock64() - w0;
  if (acc == 12345.678f && lane == 0) a.wall[patch] = 0;
}


// ---- trace mode (trace=1): the product's poll statement with a stamp per round and, per fetch lane, the round in which its slot first
// showed the step's tag (6 more instructions per round, ~12 % longer rounds: an account, not a timing).  Log per patch and step, 88 words:
// {t_enter, t_exit, t_pub (s_memtime of this CU), rounds, w_enter, w_exit, w_pub (100 MHz clock), 0}, seen[64], round stamps[16].
constexpr int kTraceW = 88;
__global__ void __launch_bounds__(64) k_trace(const Args a) {
  __shared__ v4f lds[2 * 64];
  __shared__ unsigned long long rt[64];
  const int lane = threadIdx.x;
  const unsigned xcc = xcc_id();
  int idx = 0;
  if (lane == 0) idx = atomicAdd(&a.xcd_count[xcc & 7], 1);
  idx = __builtin_amdgcn_readfirstlane(idx);
  if (idx >= a.per_xcd) return;
  const int patch = a.tile_patch[(xcc & 7) * a.per_xcd + idx];
  if (patch < 0) return;
  if (lane == 0) a.where[patch] = (int)xcc;
  const size_t hl = (size_t)patch * 64 + lane;
  const int so = a.src_off[hl];
  const int pf = a.pub_far[hl];
  const int pn0 = a.pub_near[hl * RMAX];
  const int nf = a.n_fetch[patch];
  const int chain_n = a.chain[patch];
  const unsigned long long fetch_mask = __ballot(so >= 0);
  const char* const src0 = a.base + (so >= 0 ? so : 0);
  const char* const src1 = src0 + a.par;
  lds[lane] = v4f{0.f, 0.f, 0.f, 0.f};
  lds[64 + lane] = v4f{0.f, 0.f, 0.f, 0.f};
  rt[lane] = 0;
  const unsigned lds0 = (unsigned)(size_t)lds, rt0 = (unsigned)(size_t)rt;
  const int look = nf > 0 ? lane % nf : 0;
  const unsigned tag0 = a.tag0, p0 = tag0 & 1u;
  float acc = (float)lane;
  auto publish = [&](float v, unsigned tag, int parity) {
    v4i o;
    o.x = __float_as_int(v), o.y = lane, o.z = patch, o.w = (int)tag;
    char* const pb = a.base + parity * a.par;
    if (pf >= 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(pb + pf), "v"(o) : "memory");
    if (pn0 >= 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(pb + pn0), "v"(o) : "memory");
  };
  publish(acc, tag0, (int)p0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long w0 = 0;
  bool expired = false;
  if (nf == 0) return;  // (the trace net is made of patches that wait)
  for (int it = 0; it < a.steps && !expired; ++it) {
    const unsigned s = tag0 + (unsigned)it;
    const int area = (int)(s & 1u);
    if (it == 1) w0 = wall_clock64();
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + 16u * (unsigned)(area * 64));
    const unsigned rd = lds0 + 16u * (unsigned)(area * 64 + look);
    const unsigned own_slot = dst + 16u * (unsigned)lane;
    const char* const src = area ? src1 : src0;
    const unsigned w_enter = 0, t_enter = 0;
    unsigned cnt, keep, pend_lo, tagv, tagf, tmpv, seen = 0xffffu, rta = rt0, cntv = 0;
    unsigned long long pnarrow, exec_saved, tm, tv;
    const unsigned inf = 0xffffu, rtmax = rt0 + 8u * 63u;
    v4f nbv;
    asm volatile("s_setprio 0\n\t"
                 "s_mov_b64 %[ex], exec\n\t"
                 "s_mov_b32 %[keep], m0\n\t"
                 "s_mov_b32 m0, %[dst]\n\t"
                 "s_mov_b32 %[cnt], 0\n\t"
                 "s_mov_b64 %[pn], %[fm]\n\t"
                 "1:\n\t"
                 "s_memtime %[tm]\n\t"
                 "s_mov_b64 exec, %[pn]\n\t"
                 "global_load_lds_dwordx4 %[src], off sc1\n\t"
                 "s_mov_b64 exec, %[ex]\n\t"
                 "ds_read_b32 %[t], %[ra] offset:12\n\t"
                 "ds_read_b32 %[t2], %[fa] offset:12\n\t"
                 "ds_read_b128 %[nb], %[ra]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_mov_b64 %[tv], %[tm]\n\t"
                 "ds_write_b64 %[rta], %[tv]\n\t"
                 "v_add_u32 %[rta], 8, %[rta]\n\t"
                 "v_min_u32 %[rta], %[rta], %[rtmax]\n\t"
                 "v_cmp_ne_u32_e32 vcc, %[tag], %[t2]\n\t"
                 "v_cndmask_b32_e32 %[tmp], %[cntv], %[inf], vcc\n\t"
                 "v_min_u32 %[seen], %[seen], %[tmp]\n\t"
                 "v_add_u32 %[cntv], 1, %[cntv]\n\t"
                 "s_and_b64 %[pn], vcc, %[fm]\n\t"
                 "v_cmp_ne_u32_e32 vcc, %[tag], %[t]\n\t"
                 "s_add_u32 %[cnt], %[cnt], 1\n\t"
                 "s_cmp_lt_u32 %[cnt], 60000\n\t"
                 "s_cbranch_vccz 2f\n\t"
                 "s_cbranch_scc1 1b\n\t"
                 "2:\n\t"
                 "s_setprio 3\n\t"
                 "s_or_b32 %[pl], vcc_lo, vcc_hi\n\t"
                 "s_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep), [cnt] "=&s"(cnt), [pl] "=&s"(pend_lo), [nb] "=&v"(nbv), [t] "=&v"(tagv), [t2] "=&v"(tagf), [pn] "=&s"(pnarrow),
                   [ex] "=&s"(exec_saved), [tm] "=&s"(tm), [tmp] "=&v"(tmpv), [tv] "=&v"(tv), [seen] "+v"(seen), [rta] "+v"(rta), [cntv] "+v"(cntv)
                 : [src] "v"(src), [dst] "s"(dst), [ra] "v"(rd), [fa] "v"(own_slot), [tag] "s"(s), [fm] "s"(fetch_mask), [inf] "v"(inf), [rtmax] "v"(rtmax)
                 : "vcc", "scc", "memory");
    if (pend_lo != 0u) expired = true;  // (60 000 rounds were not enough: the net is broken, leave)
    acc += nbv.x;
    const unsigned t_exit = (unsigned)clock64(), w_exit = (unsigned)wall_clock64();
    for (int k = 0; k < chain_n; ++k)
      asm volatile("v_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\t"
                   "v_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\t"
                   "v_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\t"
                   "v_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %0, 1.0, %0" : "+v"(acc));
    const unsigned t_pub = 0, w_pub = (unsigned)wall_clock64();
    publish(acc, s + 1u, (int)((s + 1u) & 1u));
    __builtin_amdgcn_s_setprio(0);
    unsigned* o = a.log + ((size_t)patch * a.steps + it) * kTraceW;
    if (lane == 0) o[0] = t_enter, o[1] = t_exit, o[2] = t_pub, o[3] = cnt, o[4] = w_enter, o[5] = w_exit, o[6] = w_pub;
    o[8 + lane] = seen;
    if (lane < 16) o[72 + lane] = (unsigned)rt[lane];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) a.wall[patch] = expired ? -1 : wall_clock64() - w0;
  if (acc == 12345.678f && lane == 0) a.wall[patch] = 0;
}

struct Cfg {
  std::map<std::string, int> kv = {{"px", 28}, {"py", 28}, {"steps", 400}, {"chain", 7}, {"jitter", 3}, {"coupled", 1}, {"nbrs", 8},
                                   {"redge", 3}, {"rcorner", 2}, {"layout", 0}, {"order", 0}, {"far_store", 1}, {"poll", 0}, {"narrow", 1},
                                   {"gap", 0}, {"stamp", 1}, {"pubs", 8}, {"seed", 1}, {"spad", 0}, {"trace", 0}};
  int operator[](const char* k) const { return kv.at(k); }
};

template <int ORDER, bool STAMP>
void launch(int poll, dim3 g, const Args& a) {
  switch (poll) {
    case 0: hipLaunchKernelGGL((k_net<ORDER, 0, STAMP>), g, dim3(64), 0, 0, a); break;
    case 1: hipLaunchKernelGGL((k_net<ORDER, 1, STAMP>), g, dim3(64), 0, 0, a); break;
    case 2: hipLaunchKernelGGL((k_net<ORDER, 2, STAMP>), g, dim3(64), 0, 0, a); break;
    default: hipLaunchKernelGGL((k_net<ORDER, 3, STAMP>), g, dim3(64), 0, 0, a); break;
  }
}

template <class T>
T* to_dev(const std::vector<T>& v) {
  T* d = nullptr;
  CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

static double pct(std::vector<double> v, double q) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  return v[std::min(v.size() - 1, (size_t)(q * (v.size() - 1) + 0.5))];
}

int run(const Cfg& c, const char* label) {
  const int px = c["px"], py = c["py"], P = px * py, steps = c["steps"];
  const int tx = 4, ty = 2, tw = px / tx, th = py / ty, per_xcd = tw * th;
  if (px % tx || py % ty) { printf("px must divide by 4, py by 2\n"); return 1; }
  auto tile_of = [&](int p) { return (p / px) / th * tx + (p % px) / tw; };
  std::vector<int> tile_patch(8 * per_xcd, -1), fill(8, 0);
  for (int p = 0; p < P; ++p) tile_patch[tile_of(p) * per_xcd + fill[tile_of(p)]++] = p;
  // records: patch p publishes `pubs` records, ids p * 16 + v; a consumer reads `redge` of an edge neighbour and `rcorner` of a corner one
  const int pubs = c["pubs"], S = P * 16 * 16 + c["spad"];       // bytes of one copy of all records (spad: the product's S is no multiple of 4 KB)
  const int mbox0 = 2 * S, mbox_bytes = P * 32 * 16;             // mailboxes behind the two copies: 32 slots per consumer
  const int par = ((2 * S + mbox_bytes + 4095) / 4096) * 4096;
  std::vector<int> src_off((size_t)P * 64, -1), src_far((size_t)P * 64, 0), pub_far((size_t)P * 64, -1), pub_near((size_t)P * 64 * RMAX, -1),
      n_fetch(P, 0), chain(P, 0);
  std::vector<std::vector<int>> readers_far((size_t)P * 16), producers(P);
  unsigned long long rs = 0x9e3779b97f4a7c15ull * (unsigned)c["seed"];
  auto rnd = [&]() { rs = rs * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rs >> 33); };
  const int nb = c["nbrs"];
  std::vector<int> near_copies((size_t)P * 16, 0);
  long lines_sum = 0, fetch_sum = 0, prod_sum = 0;
  for (int p = 0; p < P; ++p) {
    const int x = p % px, y = p / px;
    chain[p] = c["chain"] + (c["jitter"] > 0 ? (int)(rnd() % (unsigned)(c["jitter"] + 1)) : 0);
    std::vector<std::pair<int, int>> want;  // (record id, far)
    auto add = [&](int qx, int qy, int n, int first) {
      if (nb == 1) { qx = (qx + px) % px; }
      if (qx < 0 || qy < 0 || qx >= px || qy >= py) return;
      const int q = qy * px + qx;
      const bool far = tile_of(q) != tile_of(p);
      if (far && !c["coupled"]) return;
      for (int k = 0; k < n; ++k) want.push_back({q * 16 + (first + k) % pubs, far ? 1 : 0});
      producers[p].push_back(q);
    };
    if (nb == 1) add(x - 1, y, c["redge"], 0);
    if (nb >= 2) add(x - 1, y, c["redge"], 0), add(x + 1, y, c["redge"], 3);
    if (nb >= 4) add(x, y - 1, c["redge"], 5), add(x, y + 1, c["redge"], 1);
    if (nb >= 8) add(x - 1, y - 1, c["rcorner"], 6), add(x + 1, y - 1, c["rcorner"], 2), add(x - 1, y + 1, c["rcorner"], 4), add(x + 1, y + 1, c["rcorner"], 7);
    std::sort(want.begin(), want.end());
    want.erase(std::unique(want.begin(), want.end()), want.end());
    n_fetch[p] = (int)want.size();
    std::vector<int> lines;
    for (int l = 0; l < (int)want.size(); ++l) {
      const int rid = want[l].first, far = want[l].second;
      src_far[(size_t)p * 64 + l] = far;
      int off;
      if (far) {
        off = rid * 16;
        readers_far[rid].push_back(p);
      } else if (c["layout"] == 1 && near_copies[rid] < RMAX) {
        off = mbox0 + (p * 32 + l) * 16;  // the consumer's own slot: its fetch slots are contiguous in memory
        const int q = rid / 16, v = rid % 16;
        pub_near[((size_t)q * 64 + v) * RMAX + near_copies[rid]++] = off;
      } else {
        off = S + rid * 16;
        const int q = rid / 16, v = rid % 16;
        int* slot = &pub_near[((size_t)q * 64 + v) * RMAX];
        bool have = false;
        for (int r = 0; r < RMAX; ++r) have = have || slot[r] == off;
        if (!have) {
          if (near_copies[rid] < RMAX) slot[near_copies[rid]++] = off;
          else { printf("too many readers of one record\n"); return 1; }
        }
      }
      src_off[(size_t)p * 64 + l] = off;
      lines.push_back(off / 128);
    }
    std::sort(lines.begin(), lines.end());
    lines.erase(std::unique(lines.begin(), lines.end()), lines.end());
    lines_sum += (long)lines.size(), fetch_sum += n_fetch[p], prod_sum += (long)producers[p].size();
  }
  long stores_near = 0, stores_far = 0, pubv = 0;
  for (int q = 0; q < P; ++q)
    for (int v = 0; v < pubs; ++v) {
      const int rid = q * 16 + v;
      // far_store = 1: every record somebody reads is also written through, as the product does; 0: only those another XCD reads
      if (!readers_far[rid].empty() || (c["far_store"] && near_copies[rid] > 0)) pub_far[(size_t)q * 64 + v] = rid * 16;
      stores_far += pub_far[(size_t)q * 64 + v] >= 0;
      stores_near += near_copies[rid];
      pubv += near_copies[rid] > 0 || pub_far[(size_t)q * 64 + v] >= 0;
    }
  // store instructions per patch and step = 1 (far, if any lane has one) + the largest number of near copies of a lane
  Args a;
  memset(&a, 0, sizeof a);
  a.tile_patch = to_dev(tile_patch);
  std::vector<int> zero8(8, 0);
  a.xcd_count = to_dev(zero8);
  a.per_xcd = per_xcd, a.n_patches = P;
  a.src_off = to_dev(src_off), a.src_far = to_dev(src_far), a.pub_far = to_dev(pub_far), a.pub_near = to_dev(pub_near);
  a.n_fetch = to_dev(n_fetch), a.chain = to_dev(chain);
  CHECK(hipMalloc(&a.base, 2 * (size_t)par));
  CHECK(hipMemset(a.base, 0, 2 * (size_t)par));
  a.par = par, a.steps = steps;
  std::vector<unsigned> logh;
  CHECK(hipMalloc(&a.log, (size_t)P * steps * 4 * kTraceW));
  std::vector<long long> wall(P, 0);
  a.wall = to_dev(wall);
  std::vector<int> where(P, -1);
  a.where = to_dev(where);
  a.f_sleep = c["gap"], a.f_narrow = c["narrow"];
  int bad = 0, misplaced = 0;
  auto go = [&](bool stamp) -> double {  // one launch; the period in us by the device-wide clock (mean over the patches)
    CHECK(hipMemset(a.base, 0, 2 * (size_t)par));
    CHECK(hipMemset(a.xcd_count, 0, 8 * sizeof(int)));
    CHECK(hipMemset(a.log, 0, (size_t)P * steps * 4 * kTraceW));
    a.tag0 = 1000u + 2u * (unsigned)(rnd() & 0xffff);
    const dim3 g(8 * per_xcd);
    if (c["order"] == 0) { if (stamp) launch<0, true>(c["poll"], g, a); else launch<0, false>(c["poll"], g, a); }
    else { if (stamp) launch<1, true>(c["poll"], g, a); else launch<1, false>(c["poll"], g, a); }
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(wall.data(), a.wall, P * sizeof(long long), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(where.data(), a.where, P * sizeof(int), hipMemcpyDeviceToHost));
    double ticks = 0;
    for (int p = 0; p < P; ++p) {
      bad += wall[p] <= 0, ticks += (double)wall[p];
      misplaced += where[p] != tile_of(p);
    }
    return ticks / P / 100.0 / (steps - 1);
  };
  go(false);  // (warm: code object, pages)
  if (c["trace"]) {
    // ---- the account of one wait: stamps per round, per record the round that first saw it --------------------------------------
    CHECK(hipMemset(a.base, 0, 2 * (size_t)par));
    CHECK(hipMemset(a.xcd_count, 0, 8 * sizeof(int)));
    CHECK(hipMemset(a.log, 0, (size_t)P * steps * 4 * kTraceW));
    a.tag0 = 1000u + 2u * (unsigned)(rnd() & 0xffff);
    hipLaunchKernelGGL(k_trace, dim3(8 * per_xcd), dim3(64), 0, 0, a);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(wall.data(), a.wall, P * sizeof(long long), hipMemcpyDeviceToHost));
    double ticks = 0;
    int nbad = 0;
    for (int p = 0; p < P; ++p) nbad += wall[p] <= 0, ticks += (double)wall[p];
    const double us = ticks / P / 100.0 / (steps - 1);
    printf("%-40s TRACE period %.4f us%s\n", label, us, nbad ? "  ** EXPIRED **" : "");
    if (!nbad) {
      std::vector<unsigned> lg((size_t)P * steps * kTraceW);
      CHECK(hipMemcpy(lg.data(), a.log, lg.size() * 4, hipMemcpyDeviceToHost));
      auto L = [&](int p, int it, int k) { return lg[((size_t)p * steps + it) * kTraceW + k]; };
      const int i0 = steps / 4, i1 = steps - 1;
      const double cyc = (double)(unsigned)(L(0, i1, 1) - L(0, i0, 1)) / (i1 - i0), cpt = cyc / (us * 100.0);  // cycles per step, per 100 MHz tick
      std::vector<double> first_poll, round_len, rounds, near_d, far_d, first_seen, last_seen, spread, exit_after_last, hop_last, arith;
      long last_is_far = 0, steps_n = 0;
      for (int p = 0; p < P; ++p) {
        for (int it = i0; it < i1; ++it) {
          const unsigned nr = L(p, it, 3);
          if (nr == 0 || nr > 16 || it < 1) continue;
          // this CU's s_memtime -> the device-wide clock, in cycles: anchored at this step's exit
          const double off = (double)L(p, it, 5) * cpt - (double)L(p, it, 1);
          auto wallc = [&](unsigned t_local) { return (double)(int)(t_local - L(p, it, 1)) + (double)L(p, it, 5) * cpt; };
          (void)off;
          const double own_pub = (double)L(p, it - 1, 6) * cpt;  // own publish of the step before (100 MHz clock)
          first_poll.push_back(wallc(L(p, it, 72)) - own_pub);
          for (unsigned r = 1; r < nr; ++r) round_len.push_back((double)(unsigned)(L(p, it, 72 + r) - L(p, it, 72 + r - 1)));
          rounds.push_back((double)nr);
          arith.push_back((double)(int)(L(p, it, 6) - L(p, it, 5)) * cpt);
          double lo = 1e18, hi = -1e18, hop = 0;
          bool hi_far = false;
          for (int l = 0; l < n_fetch[p]; ++l) {
            const unsigned sr = L(p, it, 8 + l);
            if (sr >= nr) continue;
            const int off_b = src_off[(size_t)p * 64 + l];
            const bool far = src_far[(size_t)p * 64 + l] != 0;
            int q;  // the producer: from the record's address
            if (far) q = off_b / 256; else if (off_b >= mbox0) q = -1; else q = (off_b - S) / 256;
            if (q < 0) continue;
            const double t_seen = wallc(L(p, it, 72 + sr));
            const double d = t_seen - (double)L(q, it - 1, 6) * cpt;
            (far ? far_d : near_d).push_back(d);
            const double rel = t_seen - own_pub;
            if (rel < lo) lo = rel;
            if (rel > hi) hi = rel, hi_far = far, hop = d;
          }
          if (hi > -1e17) {
            first_seen.push_back(lo), last_seen.push_back(hi), spread.push_back(hi - lo), hop_last.push_back(hop);
            exit_after_last.push_back((double)L(p, it, 5) * cpt - (hi + own_pub));
            last_is_far += hi_far, ++steps_n;
          }
        }
      }
      printf("    %.0f cycles per step (%.3f GHz).  Times in shader cycles; a record's clock crossing is good to +-24 (the 100 MHz clock).\n", cyc, cyc / (us * 1000.0));
      printf("    own publish -> first poll load issued %.0f | a poll round %.0f (p10 %.0f p90 %.0f), %.1f rounds per wait | exit stamp -> publish stamp (the arithmetic) %.0f\n",
             pct(first_poll, 0.5), pct(round_len, 0.5), pct(round_len, 0.1), pct(round_len, 0.9), pct(rounds, 0.5), pct(arith, 0.5));
      printf("    producer's publish stamp -> the round that first sees the record: same XCD p10 %.0f p50 %.0f p90 %.0f", pct(near_d, 0.1), pct(near_d, 0.5), pct(near_d, 0.9));
      if (!far_d.empty()) printf(" | other XCD p10 %.0f p50 %.0f p90 %.0f", pct(far_d, 0.1), pct(far_d, 0.5), pct(far_d, 0.9));
      printf("\n    per wait, after the patch's own publish: first record seen %.0f, last %.0f (spread %.0f; the last one is another XCD's in %.0f %% of the waits); its hop %.0f (p10 %.0f p90 %.0f); "
             "that round's start -> exit stamp %.0f\n",
             pct(first_seen, 0.5), pct(last_seen, 0.5), pct(spread, 0.5), 100.0 * last_is_far / std::max(1l, steps_n), pct(hop_last, 0.5), pct(hop_last, 0.1), pct(hop_last, 0.9),
             pct(exit_after_last, 0.5));
    }
  }
  std::vector<double> per;
  for (int rep = 0; rep < 5 && !bad; ++rep) per.push_back(go(false));
  const double period_us = bad ? 0.0 : pct(per, 0.5);
  printf("%-40s period %.4f us (5 launches: %.4f..%.4f) | fetch %.1f records of %.1f producers in %.1f lines; stores per published record: same-XCD %.2f + write-through %.2f%s%s\n",
         label, period_us, bad ? 0.0 : pct(per, 0.0), bad ? 0.0 : pct(per, 1.0), (double)fetch_sum / P, (double)prod_sum / P, (double)lines_sum / P,
         (double)stores_near / std::max(1l, pubv), (double)stores_far / std::max(1l, pubv), bad ? "  ** EXPIRED **" : "",
         misplaced ? "  ** a patch ran on another XCD than its tile **" : "");
  if (c["stamp"] && !bad) {
    const double stamped_us = go(true);
    logh.resize((size_t)P * steps * 8);
    CHECK(hipMemcpy(logh.data(), a.log, logh.size() * 4, hipMemcpyDeviceToHost));
    auto L = [&](int p, int it, int k) { return logh[((size_t)p * steps + it) * 8 + k]; };
    const int i0 = steps / 4, i1 = steps - 1;
    const double cyc = (double)(unsigned)(L(0, i1, 0) - L(0, i0, 0)) / (i1 - i0);
    const double cyc_per_tick = cyc / (stamped_us * 100.0);
    std::vector<double> wait, comp, rounds, last_near, first_near, last_far, spread;
    // The hop as the consumer sees it: its exit (all records seen) minus the publish stamp of each of its producers, both by the
    // device-wide 100 MHz clock (10 ns = ~24 cycles; the CUs' s_memtime counters are not aligned with one another).  The publish
    // stamp is taken right before the stores issue.  LAST = the producer that published last among those the wait ended on.
    for (int p = 0; p < P; ++p) {
      for (int it = i0; it < i1; ++it) {
        wait.push_back((double)(unsigned)(L(p, it, 1) - L(p, it, 0)));
        comp.push_back((double)(unsigned)(L(p, it, 2) - L(p, it, 1)));
        rounds.push_back((double)L(p, it, 3));
        double lo = 1e18, hi = -1e18, flo = 1e18;
        for (int q : producers[p]) {
          const double d = (double)(int)(L(p, it, 4) - L(q, it - 1, 5)) * cyc_per_tick;
          if (tile_of(q) != tile_of(p)) { flo = std::min(flo, d); continue; }
          lo = std::min(lo, d), hi = std::max(hi, d);
        }
        if (hi > -1e17) last_near.push_back(lo), first_near.push_back(hi), spread.push_back(hi - lo);
        if (flo < 1e17) last_far.push_back(flo);
      }
    }
    std::vector<double> pw(P, 0), pc(P, 0);
    for (int p = 0; p < P; ++p) {
      for (int it = i0; it < i1; ++it) pw[p] += (double)(unsigned)(L(p, it, 1) - L(p, it, 0)), pc[p] += (double)(unsigned)(L(p, it, 2) - L(p, it, 1));
      pw[p] /= (i1 - i0), pc[p] /= (i1 - i0);
    }
    const int pm = (int)(std::min_element(pw.begin(), pw.end()) - pw.begin());
    printf("    with stamps %.4f us = %.0f cycles (%.3f GHz) | arithmetic: median %.0f max %.0f | wait: median %.0f, least %.0f (that patch's arithmetic %.0f) | poll rounds %.1f\n",
           stamped_us, cyc, cyc / (stamped_us * 1000.0), pct(comp, 0.5), *std::max_element(pc.begin(), pc.end()), pct(wait, 0.5), pw[pm], pc[pm], pct(rounds, 0.5));
    printf("    hop = exit - publish stamp of a producer (cycles): same XCD, the last to publish p10 %.0f p50 %.0f p90 %.0f; the first p50 %.0f; spread p50 %.0f",
           pct(last_near, 0.1), pct(last_near, 0.5), pct(last_near, 0.9), pct(first_near, 0.5), pct(spread, 0.5));
    if (!last_far.empty()) printf(" | other XCD, the last p10 %.0f p50 %.0f p90 %.0f", pct(last_far, 0.1), pct(last_far, 0.5), pct(last_far, 0.9));
    printf("\n");
  }
  hipFree((void*)a.tile_patch), hipFree(a.xcd_count), hipFree((void*)a.src_off), hipFree((void*)a.src_far), hipFree((void*)a.pub_far), hipFree((void*)a.pub_near);
  hipFree((void*)a.n_fetch), hipFree((void*)a.chain), hipFree(a.base), hipFree(a.wall), hipFree(a.where);
  hipFree(a.log);
  return bad ? 2 : 0;
}

int main(int argc, char** argv) {
  // one line = one configuration: "label key=value key=value ..."; configurations are separated by "--"
  Cfg base;
  std::vector<std::pair<std::string, Cfg>> runs;
  Cfg cur = base;
  std::string label;
  bool any = false;
  for (int i = 1; i <= argc; ++i) {
    if (i == argc || !strcmp(argv[i], "--")) {
      if (any || runs.empty()) runs.push_back({label.empty() ? "default" : label, cur});
      cur = base, label = "", any = false;
      continue;
    }
    any = true;
    label += std::string(label.empty() ? "" : " ") + argv[i];
    const char* eq = strchr(argv[i], '=');
    if (!eq) continue;  // a word: part of the label only
    const std::string k(argv[i], eq - argv[i]);
    if (!cur.kv.count(k)) { printf("unknown key %s\n", k.c_str()); return 1; }
    cur.kv[k] = atoi(eq + 1);
  }
  int rc = 0;
  for (auto& r : runs) rc |= run(r.second, r.first.c_str());
  return rc;
}
