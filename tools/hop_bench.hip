// tools/hop_bench.hip -- microbenchmarks that decide the persistent-kernel design (not product code).
//  (1) one-way latency of a 16-byte {payload,tag} record hand-off between two waves on different
//      CUs, for several store/load cache policies, same-XCD and cross-XCD, idle and with all pairs
//      ping-ponging at once;
//  (2) semantics of DPP wave_shr:1 on gfx950 (used for the ordered segmented accumulation).
// Build: hipcc --offload-arch=gfx950 -O3 -o hop_bench hop_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void* p) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000); }

// pair p = (block 2p, block 2p+stride...)  we use: block b talks to block b^partner_xor.
template <int ST_AUX, int LD_AUX, bool PLAIN_STORE>
__global__ void __launch_bounds__(64) k_pingpong(v4i* buf, int partner_xor, int iters, long long* cycles, int* xcc_of_block, int active_pairs_mask) {
  const int b = blockIdx.x;
  const int partner = b ^ partner_xor;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) xcc_of_block[b] = (int)(xcc & 0xf);
  const bool first = (b & partner_xor) == 0;
  __amdgpu_buffer_rsrc_t r = rsrc(buf);
  const int lane = threadIdx.x;
  // each lane exchanges its own 16-byte record (64 records per block) -> like a slice of vertices
  const int my_off = (b * 64 + lane) * 16, pa_off = (partner * 64 + lane) * 16;
  long long t0 = 0;
  int tag = 0;
  for (int it = 0; it <= iters; ++it) {
    if (it == 1) t0 = wall_clock64();
    if (first) {
      ++tag;
      v4i rec = {lane, it, b, tag};
      if (PLAIN_STORE) *(v4i*)((char*)buf + my_off) = rec; else __builtin_amdgcn_raw_buffer_store_b128(rec, r, my_off, 0, ST_AUX);
      // wait for the echo
      for (unsigned spin = 0;; ++spin) {
        int o = pa_off;
        asm volatile("" : "+v"(o) :: "memory");  // opaque: the compiler must re-issue the load every spin
        v4i g = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, LD_AUX);
        if (__all(g.w == tag)) break;
        if (spin > (1u << 18)) { if (threadIdx.x == 0) cycles[b] = -(long long)it - 1; return; }
      }
    } else {
      ++tag;
      for (unsigned spin = 0;; ++spin) {
        int o = pa_off;
        asm volatile("" : "+v"(o) :: "memory");  // opaque: the compiler must re-issue the load every spin
        v4i g = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, LD_AUX);
        if (__all(g.w == tag)) break;
        if (spin > (1u << 18)) { if (threadIdx.x == 0) cycles[b] = -(long long)it - 1; return; }
      }
      v4i rec = {lane, it, b, tag};
      if (PLAIN_STORE) *(v4i*)((char*)buf + my_off) = rec; else __builtin_amdgcn_raw_buffer_store_b128(rec, r, my_off, 0, ST_AUX);
    }
  }
  if (threadIdx.x == 0) cycles[b] = wall_clock64() - t0;
}


// (4) the consumer side as k_persistent_pv does it: the poll is an LDS-DMA load (no VGPR destination), re-issued every
//     round without waiting, and the tag is read back from LDS
template <bool PLAIN_STORE, int GAP>
__global__ void __launch_bounds__(64) k_pingpong_dma(v4i* buf, int partner_xor, int iters, long long* cycles, int* xcc_of_block) {
  __shared__ v4i slot[64];
  const int b = blockIdx.x, partner = b ^ partner_xor, lane = threadIdx.x;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (lane == 0) xcc_of_block[b] = (int)(xcc & 0xf);
  const bool first = (b & partner_xor) == 0;
  __amdgpu_buffer_rsrc_t r = rsrc(buf);
  const int my_off = (b * 64 + lane) * 16;
  const char* src = (const char*)buf + (partner * 64 + lane) * 16;
  const unsigned dst = (unsigned)(size_t)slot, ra = dst + 16u * lane;
  slot[lane] = v4i{0, 0, 0, 0};
  __syncthreads();
  long long t0 = 0;
  int tag = 0;
  auto wait_tag = [&](int want) -> bool {
    for (unsigned outer = 0; outer < (1u << 14); ++outer) {
      unsigned keep, cnt, pl, t;
      asm volatile("s_mov_b32 %[keep], m0\n\t"
                   "s_mov_b32 m0, %[dst]\n\t"
                   "s_mov_b32 %[cnt], 0\n\t"
                   "1:\n\t"
                   "global_load_lds_dwordx4 %[src], off sc1\n\t"
                   ".if %[gap] == 9\n\ts_waitcnt vmcnt(0)\n\t.else\n\ts_sleep %[gap]\n\t.endif\n\t"
                   "ds_read_b32 %[t], %[ra] offset:12\n\t"
                   "s_add_u32 %[cnt], %[cnt], 1\n\t"
                   "s_waitcnt lgkmcnt(0)\n\t"
                   "v_cmp_ne_u32_e32 vcc, %[tag], %[t]\n\t"
                   "s_cmp_lt_u32 %[cnt], 64\n\t"
                   "s_cbranch_vccz 2f\n\t"
                   "s_cbranch_scc1 1b\n\t"
                   "2:\n\t"
                   "s_mov_b32 %[pl], vcc_lo\n\t"
                   "s_or_b32 %[pl], %[pl], vcc_hi\n\t"
                   "s_mov_b32 m0, %[keep]"
                   : [keep] "=&s"(keep), [cnt] "=&s"(cnt), [pl] "=&s"(pl), [t] "=&v"(t)
                   : [src] "v"(src), [dst] "s"(dst), [ra] "v"(ra), [tag] "s"(want), [gap] "n"(GAP)
                   : "vcc", "scc", "memory");
      if (pl == 0u) return true;
    }
    return false;
  };
  for (int it = 0; it <= iters; ++it) {
    if (it == 1) t0 = wall_clock64();
    ++tag;
    if (!first && !wait_tag(tag)) { if (lane == 0) cycles[b] = -(long long)it - 1; return; }
    v4i rec = {lane, it, b, tag};
    if (PLAIN_STORE) *(v4i*)((char*)buf + my_off) = rec; else __builtin_amdgcn_raw_buffer_store_b128(rec, r, my_off, 0, 16);
    if (first && !wait_tag(tag)) { if (lane == 0) cycles[b] = -(long long)it - 1; return; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) cycles[b] = wall_clock64() - t0;
}


// (5) does the ADDRESS matter?  One record (all lanes the same 16 bytes) at byte offset `off` (echo in the same 64-byte
//     line, 16 bytes further), two blocks on different XCDs: the line's home channel sits somewhere between them
__global__ void __launch_bounds__(64) k_pingpong_addr(v4i* buf, int partner_xor, int iters, long long* cycles, int* xcc_of_block, unsigned off) {
  const int b = blockIdx.x;
  if (b != 0 && b != partner_xor) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) xcc_of_block[b] = (int)(xcc & 0xf);
  const bool first = b == 0;
  __amdgpu_buffer_rsrc_t r = rsrc(buf);
  const int my_off = (int)off + (first ? 0 : 16), pa_off = (int)off + (first ? 16 : 0);
  long long t0 = 0;
  int tag = 0;
  for (int it = 0; it <= iters; ++it) {
    if (it == 1) t0 = wall_clock64();
    ++tag;
    if (!first) {
      for (unsigned spin = 0;; ++spin) {
        int o = pa_off;
        asm volatile("" : "+v"(o) :: "memory");
        v4i g = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 16);
        if (__all(g.w == tag)) break;
        if (spin > (1u << 18)) { if (threadIdx.x == 0) cycles[b] = -(long long)it - 1; return; }
      }
    }
    v4i rec = {0, it, b, tag};
    __builtin_amdgcn_raw_buffer_store_b128(rec, r, my_off, 0, 16);
    if (first) {
      for (unsigned spin = 0;; ++spin) {
        int o = pa_off;
        asm volatile("" : "+v"(o) :: "memory");
        v4i g = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 16);
        if (__all(g.w == tag)) break;
        if (spin > (1u << 18)) { if (threadIdx.x == 0) cycles[b] = -(long long)it - 1; return; }
      }
    }
  }
  if (threadIdx.x == 0) cycles[b] = wall_clock64() - t0;
}

static int addr_scan(v4i* buf, long long* cyc, int* xcc) {
  const int iters = 1000;
  printf("address scan: one-way latency (us) of one 16-byte record, block 0 <-> block x, by byte offset of the record\n");
  for (int px : {1, 2, 3, 4, 5, 6, 7}) {
    for (unsigned stride : {4096u}) {
      double lo = 1e9, hi = 0, sum = 0;
      int xa = -1, xb = -1;
      std::vector<double> v;
      const int n = 64;
      for (int k = 0; k < n; ++k) {
        const unsigned off = (unsigned)k * stride;
        CHECK(hipMemset(buf, 0, 4096 * 64 * 16));
        CHECK(hipMemset(cyc, 0, 64 * 8));
        hipLaunchKernelGGL(k_pingpong_addr, dim3(8), dim3(64), 0, 0, buf, px, iters, cyc, xcc, off);
        CHECK(hipDeviceSynchronize());
        long long h[8]; int hx[8];
        CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
        const double us = h[0] / 100.0 / iters / 2;
        v.push_back(us);
        lo = us < lo ? us : lo, hi = us > hi ? us : hi, sum += us;
        xa = hx[0], xb = hx[px];
      }
      printf("xcc %d<->%d stride %7u: min %.3f mean %.3f max %.3f |", xa, xb, stride, lo, sum / n, hi);
      for (double u : v) printf(" %.2f", u);
      printf("\n");
    }
  }
  return 0;
}

// (6) the same question asked one XCD at a time: round trip of a dependent sc1 load from block b's XCD to each 4 KB page
//     (what a run-time calibration of the exchange buffer would measure)
__global__ void __launch_bounds__(64) k_load_rt(v4i* buf, int n_pages, int reps, unsigned* out, int* xcc_of_block) {
  const int b = blockIdx.x;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) xcc_of_block[b] = (int)(xcc & 0xf);
  __amdgpu_buffer_rsrc_t r = rsrc(buf);
  int chain = 0;
  for (int p = 0; p < n_pages; ++p) {
    long long best = 1ll << 60;
    for (int rep = 0; rep < 3; ++rep) {
      const long long t0 = clock64();
      for (int k = 0; k < reps; ++k) {
        int o = p * 4096 + (chain & 0);
        asm volatile("" : "+v"(o) :: "memory");
        v4i g = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 16);
        chain += g.x;
      }
      const long long t1 = clock64();
      best = (t1 - t0) < best ? (t1 - t0) : best;
    }
    if (threadIdx.x == 0) out[b * n_pages + p] = (unsigned)(best / reps);
  }
  if (chain == 12345 && threadIdx.x == 0) out[0] = 0;
}

static int load_scan(v4i* buf, int* xcc) {
  const int n_pages = 64, reps = 64;
  unsigned* out;
  CHECK(hipMalloc(&out, 8 * n_pages * 4));
  CHECK(hipMemset(buf, 0, 4096 * 64 * 16));
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: the eight blocks one after the other (each alone on the chip); pass 1: all eight at once
    std::vector<unsigned> h(8 * n_pages);
    int hx[8];
    if (pass == 0) {
      hipLaunchKernelGGL(k_load_rt, dim3(8), dim3(64), 0, 0, buf, n_pages, reps, out, xcc);
    } else {
      hipLaunchKernelGGL(k_load_rt, dim3(8), dim3(64), 0, 0, buf, n_pages, reps, out, xcc);
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
    printf("load round trip (shader cycles) by 4 KB page, pass %d\n", pass);
    for (int b = 0; b < 8; ++b) {
      printf("xcc %d:", hx[b]);
      for (int p = 0; p < n_pages; ++p) printf(" %u", h[b * n_pages + p]);
      printf("\n");
    }
  }
  return 0;
}

// (7) ... and in which DIRECTION: the producer stamps the record with the device-wide 100 MHz clock, the consumer reads the
//     clock when it sees the tag.  out[0] = mean one-way ticks block 0 -> block x, out[1] = block x -> block 0.
__global__ void __launch_bounds__(64) k_oneway_addr(v4i* buf, int partner_xor, int iters, long long* sums, int* xcc_of_block, unsigned off) {
  const int b = blockIdx.x;
  if (b != 0 && b != partner_xor) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) xcc_of_block[b] = (int)(xcc & 0xf);
  const bool first = b == 0;
  __amdgpu_buffer_rsrc_t r = rsrc(buf);
  const int my_off = (int)off + (first ? 0 : 16), pa_off = (int)off + (first ? 16 : 0);
  long long sum = 0;
  int tag = 0;
  for (int it = 0; it <= iters; ++it) {
    ++tag;
    if (!first) {
      v4i g;
      for (unsigned spin = 0;; ++spin) {
        int o = pa_off;
        asm volatile("" : "+v"(o) :: "memory");
        g = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 16);
        if (__all(g.w == tag)) break;
        if (spin > (1u << 18)) { if (threadIdx.x == 0) sums[b] = -1; return; }
      }
      if (it > 0) sum += (long long)((unsigned)wall_clock64() - (unsigned)g.x);
    }
    v4i rec = {(int)(unsigned)wall_clock64(), it, b, tag};
    __builtin_amdgcn_raw_buffer_store_b128(rec, r, my_off, 0, 16);
    if (first) {
      v4i g;
      for (unsigned spin = 0;; ++spin) {
        int o = pa_off;
        asm volatile("" : "+v"(o) :: "memory");
        g = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 16);
        if (__all(g.w == tag)) break;
        if (spin > (1u << 18)) { if (threadIdx.x == 0) sums[b] = -1; return; }
      }
      if (it > 0) sum += (long long)((unsigned)wall_clock64() - (unsigned)g.x);
    }
  }
  if (threadIdx.x == 0) sums[b] = sum;
}

static int oneway_scan(v4i* buf, long long* cyc, int* xcc) {
  const int iters = 1000;
  printf("one-way scan: latency (us) of one 16-byte record by direction, block 0 <-> block x, by 4 KB page; 'a>b' = written on a, seen on b\n");
  for (int px : {1, 2, 3, 4, 5, 6, 7}) {
    std::vector<double> f, bk;
    int xa = -1, xb = -1;
    for (int k = 0; k < 16; ++k) {
      CHECK(hipMemset(buf, 0, 4096 * 64 * 16));
      CHECK(hipMemset(cyc, 0, 64 * 8));
      hipLaunchKernelGGL(k_oneway_addr, dim3(8), dim3(64), 0, 0, buf, px, iters, cyc, xcc, (unsigned)k * 4096u);
      CHECK(hipDeviceSynchronize());
      long long h[8]; int hx[8];
      CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
      // block px measured 0 -> px, block 0 measured px -> 0
      f.push_back(h[px] / 100.0 / iters), bk.push_back(h[0] / 100.0 / iters);
      xa = hx[0], xb = hx[px];
    }
    printf("xcc %d>%d:", xa, xb); for (double u : f) printf(" %.2f", u); printf("\n");
    printf("xcc %d>%d:", xb, xa); for (double u : bk) printf(" %.2f", u); printf("\n");
  }
  return 0;
}

template <bool PLAIN, int GAP>
int run_dma(const char* name, v4i* buf, long long* cyc, int* xcc, int nblocks, int partner_xor, int iters) {
  CHECK(hipMemset(buf, 0, nblocks * 64 * 16));
  CHECK(hipMemset(cyc, 0, nblocks * 8));
  hipLaunchKernelGGL((k_pingpong_dma<PLAIN, GAP>), dim3(nblocks), dim3(64), 0, 0, buf, partner_xor, iters, cyc, xcc);
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  std::vector<long long> h(nblocks);
  std::vector<int> hx(nblocks);
  CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * nblocks, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hx.data(), xcc, sizeof(int) * nblocks, hipMemcpyDeviceToHost));
  double sum = 0; long long mx = 0;
  for (int i = 0; i < nblocks; ++i) { sum += h[i]; if (h[i] > mx) mx = h[i]; }
  double us_rt = (sum / nblocks) / 100.0 / iters;
  printf("[c0=%lld] %-40s blocks=%4d xor=%3d (xcc %d<->%d): one-way %.3f us (max %.3f)\n", h[0], name, nblocks, partner_xor, hx[0],
         hx[partner_xor], us_rt / 2, mx / 100.0 / iters / 2);
  return 0;
}

__global__ void k_dpp(int* out) {
  int lane = threadIdx.x;
  int v = lane * 10;
  // wave_shr:1 = 0x138, row_shr:1 = 0x111
  int a = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);
  int bb = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false);
  out[lane] = a;
  out[64 + lane] = bb;
}

template <int ST, int LD, bool PLAIN>
int run(const char* name, v4i* buf, long long* cyc, int* xcc, int nblocks, int partner_xor, int iters) {
  CHECK(hipMemset(buf, 0, nblocks * 64 * 16));
  CHECK(hipMemset(cyc, 0, nblocks * 8));
  hipLaunchKernelGGL((k_pingpong<ST, LD, PLAIN>), dim3(nblocks), dim3(64), 0, 0, buf, partner_xor, iters, cyc, xcc, 0);
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  std::vector<long long> h(nblocks);
  std::vector<int> hx(nblocks);
  CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * nblocks, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hx.data(), xcc, sizeof(int) * nblocks, hipMemcpyDeviceToHost));
  double sum = 0; long long mx = 0;
  for (int i = 0; i < nblocks; ++i) { sum += h[i]; if (h[i] > mx) mx = h[i]; }
  // wall_clock64 ticks at 100 MHz
  double us_rt = (sum / nblocks) / 100.0 / iters;
  printf("[c0=%lld c1=%lld] %-28s blocks=%4d xor=%3d (xcc %d<->%d): round trip %.3f us  one-way %.3f us (max %.3f)\n", h[0], h[1], name, nblocks, partner_xor,
         hx[0], hx[partner_xor], us_rt, us_rt / 2, mx / 100.0 / iters / 2);
  return 0;
}

int main() {
  v4i* buf; long long* cyc; int* xcc; int* d;
  CHECK(hipMalloc(&buf, 4096 * 64 * 16)); CHECK(hipMalloc(&cyc, 4096 * 8)); CHECK(hipMalloc(&xcc, 4096 * 4)); CHECK(hipMalloc(&d, 128 * 4));
  hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, d);
  int h[128]; CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("dpp wave_shr:1 :"); for (int i = 0; i < 64; i += 1) printf(" %d", h[i]); printf("\n");
  printf("dpp row_shr:1  :"); for (int i = 0; i < 64; i += 1) printf(" %d", h[64 + i]); printf("\n");
  const int iters = 2000;
  if (getenv("HOP_BENCH_POLICY_SCAN")) {  // does any load policy skip the reader's L2 (and with it the invalidate + refetch)?
    for (int nb : {2, 256}) {
      run<16, 16, false>("st sc1 / ld sc1", buf, cyc, xcc, nb, 1, iters);
      run<16, 18, false>("st sc1 / ld sc1 nt", buf, cyc, xcc, nb, 1, iters);
      run<16, 19, false>("st sc1 / ld sc0 sc1 nt", buf, cyc, xcc, nb, 1, iters);
      run<18, 18, false>("st sc1 nt / ld sc1 nt", buf, cyc, xcc, nb, 1, iters);
      run<19, 19, false>("st sc0sc1nt / ld sc0sc1nt", buf, cyc, xcc, nb, 1, iters);
      run<16, 17, false>("st sc1 / ld sc0 sc1", buf, cyc, xcc, nb, 1, iters);
    }
    return 0;
  }
  if (getenv("HOP_BENCH_ONEWAY_SCAN")) return oneway_scan(buf, cyc, xcc);
  if (getenv("HOP_BENCH_LOAD_SCAN")) load_scan(buf, xcc);
  if (getenv("HOP_BENCH_ADDR_SCAN")) return addr_scan(buf, cyc, xcc);
  // (3) does the KIND of memory matter for the cross-XCD hand-off?  fine-grained / uncached allocations bypass the L2s by
  //     memory type instead of by instruction policy
  {
    v4i *fg = nullptr, *uc = nullptr;
    const size_t bytes = 4096 * 64 * 16;
    if (hipExtMallocWithFlags((void**)&fg, bytes, hipDeviceMallocFinegrained) != hipSuccess) fg = nullptr;
    if (hipExtMallocWithFlags((void**)&uc, bytes, hipDeviceMallocUncached) != hipSuccess) uc = nullptr;
    (void)hipGetLastError();
    for (int nb : {2, 256, 1024}) {
      for (int px : {1, 8}) {
        if (px >= nb) continue;
        run<16, 16, false>("coarse: st sc1 / ld sc1", buf, cyc, xcc, nb, px, iters);
        if (fg) {
          run<16, 16, false>("fine-grained: st sc1 / ld sc1", fg, cyc, xcc, nb, px, iters);
          run<0, 0, true>("fine-grained: st plain / ld plain", fg, cyc, xcc, nb, px, iters);
          run<0, 16, true>("fine-grained: st plain / ld sc1", fg, cyc, xcc, nb, px, iters);
        }
        if (uc) {
          run<16, 16, false>("uncached: st sc1 / ld sc1", uc, cyc, xcc, nb, px, iters);
          run<0, 0, true>("uncached: st plain / ld plain", uc, cyc, xcc, nb, px, iters);
        }
      }
    }
    if (fg) (void)hipFree(fg);
    if (uc) (void)hipFree(uc);
    for (int nb : {2, 256, 1024}) {
      for (int px : {1, 8}) {
        if (px >= nb) continue;
        run<16, 16, false>("VGPR poll: st sc1 / ld sc1", buf, cyc, xcc, nb, px, iters);
        run_dma<false, 0>("LDS-DMA poll, no gap: st sc1 / ld sc1", buf, cyc, xcc, nb, px, iters);
        run_dma<false, 1>("LDS-DMA poll, s_sleep 1: st sc1 / ld sc1", buf, cyc, xcc, nb, px, iters);
        run_dma<false, 9>("LDS-DMA poll, one in flight: st sc1 / ld sc1", buf, cyc, xcc, nb, px, iters);
        if (px == 8) {
          run<0, 16, true>("VGPR poll: st plain / ld sc1", buf, cyc, xcc, nb, px, iters);
          run_dma<true, 0>("LDS-DMA poll, no gap: st plain / ld sc1", buf, cyc, xcc, nb, px, iters);
          run_dma<true, 1>("LDS-DMA poll, s_sleep 1: st plain / ld sc1", buf, cyc, xcc, nb, px, iters);
          run_dma<true, 9>("LDS-DMA poll, one in flight: st plain / ld sc1", buf, cyc, xcc, nb, px, iters);
        }
      }
    }
    if (getenv("HOP_BENCH_MEMTYPE_ONLY")) return 0;
  }
  for (int nb : {2, 16, 256, 1024}) {
    for (int px : {1, 8}) {  // xor 1: partner on a different XCD (b%8 differs); xor 8: same XCD
      if (px >= nb) continue;
      run<16, 16, false>("st sc1 / ld sc1", buf, cyc, xcc, nb, px, iters);
      run<17, 17, false>("st sc0sc1 / ld sc0sc1", buf, cyc, xcc, nb, px, iters);
      run<0, 16, true>("st plain / ld sc1", buf, cyc, xcc, nb, px, iters);
      run<1, 16, false>("st sc0 / ld sc1", buf, cyc, xcc, nb, px, iters);
      run<2, 16, false>("st nt / ld sc1", buf, cyc, xcc, nb, px, iters);
      run<16, 1, false>("st sc1 / ld sc0", buf, cyc, xcc, nb, px, iters);
    }
  }
  return 0;
}
