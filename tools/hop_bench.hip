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
