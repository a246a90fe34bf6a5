// tools/residency_probe.hip -- how many waves of a persistent kernel REALLY become co-resident on a CU (not product code).
//
// Round 2 left a question open: k_persistent_pv -- one 64-thread workgroup per patch -- passed the
// runtime's cooperative-launch check at 25 workgroups per CU for a 1080p frame and then sat in its first wait until it
// expired, so the form was capped at the 12 per CU it had been run at.  This tool measures the residency directly:
// a kernel of a given block size / LDS bytes / VGPR budget in which every wave reports where it runs (HW_ID, XCC_ID),
// counts itself in, and spins until everybody is in or a deadline passes.  Waves that only start after the deadline
// (because they had to wait for a resident wave to leave) are the ones a dataflow kernel would have deadlocked on.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o residency_probe residency_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__);          \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

struct WaveReport {
  unsigned hw_id, xcc_id, seen_at_exit, late;  // late: arrived after somebody's deadline had already passed
  unsigned long long t_arrive;
};

// VG: the kernel claims at least VG vector registers (an asm clobber of v[VG-1])
template <int VG, int SG>
__global__ void k_resident(unsigned* arrive, unsigned* deadline_hit, WaveReport* out, unsigned total_waves,
                           unsigned long long budget_ticks) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = (int)threadIdx.x;  // (the allocation must be real)
  if (VG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  if (VG == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
  if (VG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  // ... and SG scalar registers (VCC, FLAT_SCRATCH, XNACK_MASK come on top: the compiler reports SG + 6)
  if (SG == 80) asm volatile("s_mov_b32 s79, 0" ::: "s79");
  if (SG == 88) asm volatile("s_mov_b32 s87, 0" ::: "s87");
  if (SG == 96) asm volatile("s_mov_b32 s95, 0" ::: "s95");
  if (SG == 100) asm volatile("s_mov_b32 s99, 0" ::: "s99");
  const unsigned wave = (unsigned)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const unsigned long long t0 = wall_clock64();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned seen = 0, late = 0;
  if ((threadIdx.x & 63) == 0) {
    late = __hip_atomic_load(deadline_hit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
      seen = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen >= total_waves) break;
      if (wall_clock64() - t0 > budget_ticks) {
        __hip_atomic_store(deadline_hit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    WaveReport r;
    r.hw_id = hw, r.xcc_id = xcc & 15u, r.seen_at_exit = seen, r.late = late, r.t_arrive = t0;
    out[wave] = r;
  }
  __syncthreads();
  if (lds[threadIdx.x] < 0) out[0].hw_id = 0;  // (keeps the LDS store alive)
}

static const void* pick(int vg, int sg) {
  if (sg == 80) return vg == 64 ? (const void*)k_resident<64, 80> : (const void*)k_resident<72, 80>;
  if (sg == 88) return vg == 64 ? (const void*)k_resident<64, 88> : (const void*)k_resident<72, 88>;
  if (sg == 96) return vg == 64 ? (const void*)k_resident<64, 96> : (const void*)k_resident<72, 96>;
  if (sg == 100) return vg == 64 ? (const void*)k_resident<64, 100> : (const void*)k_resident<72, 100>;
  switch (vg) {
    case 64: return (const void*)k_resident<64, 0>;
    case 72: return (const void*)k_resident<72, 0>;
    case 96: return (const void*)k_resident<96, 0>;
    case 128: return (const void*)k_resident<128, 0>;
    default: return (const void*)k_resident<0, 0>;
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s  CUs %d  LDS/CU %zu  regs/CU %d  coop %d\n", prop.gcnArchName, cus, (size_t)prop.maxSharedMemoryPerMultiProcessor,
         prop.regsPerMultiprocessor, prop.cooperativeLaunch);
  // configurations: {threads per block, LDS bytes per block, vgprs, waves per CU asked for}
  struct Cfg { int threads, lds, vg, waves_per_cu, sg; };
  std::vector<Cfg> cfgs;
  if (argc >= 5) {
    cfgs.push_back({atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argc >= 6 ? atoi(argv[5]) : 0});
  } else {
    for (int wpc : {24, 28, 32})
      for (int threads : {64, 256}) {
        cfgs.push_back({threads, 3840 * (threads / 64), 64, wpc, 0});
        cfgs.push_back({threads, 3840 * (threads / 64), 72, wpc, 0});
      }
    // scalar registers: the trap handler's 16 come on top of a wave's allocation (800 per SIMD)
    for (int sg : {80, 88, 96, 100})
      for (int wpc : {28, 32}) {
        cfgs.push_back({64, 3840, 64, wpc, sg});
        cfgs.push_back({64, 3840, 72, wpc, sg});
      }
    cfgs.push_back({64, 10240, 64, 16, 0});
    cfgs.push_back({256, 40960, 64, 16, 0});
  }
  unsigned *d_arrive = nullptr, *d_dead = nullptr;
  WaveReport* d_out = nullptr;
  CHECK(hipMalloc(&d_arrive, 4));
  CHECK(hipMalloc(&d_dead, 4));
  const size_t max_waves = (size_t)40 * cus;
  CHECK(hipMalloc(&d_out, sizeof(WaveReport) * max_waves));
  std::vector<WaveReport> h(max_waves);
  for (const Cfg& c : cfgs) {
    const int wpb = c.threads / 64;
    const int blocks = (c.waves_per_cu * cus + wpb - 1) / wpb;
    const unsigned total = (unsigned)(blocks * wpb);
    if (total > max_waves) continue;
    const void* fn = pick(c.vg, c.sg);
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, c.threads, (size_t)c.lds);
    CHECK(hipMemset(d_arrive, 0, 4));
    CHECK(hipMemset(d_dead, 0, 4));
    CHECK(hipMemset(d_out, 0, sizeof(WaveReport) * total));
    unsigned long long budget = 2000000ull;  // 20 ms of the 100 MHz clock
    unsigned tw = total;
    void* args[] = {&d_arrive, &d_dead, &d_out, &tw, &budget};
    hipError_t le = hipLaunchKernel(fn, dim3((unsigned)blocks), dim3((unsigned)c.threads), args, (size_t)c.lds, nullptr);
    if (le != hipSuccess) {
      printf("threads %3d lds %6d vgpr %3d ask %2d/CU : launch failed: %s\n", c.threads, c.lds, c.vg, c.waves_per_cu, hipGetErrorString(le));
      (void)hipGetLastError();
      continue;
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d_out, sizeof(WaveReport) * total, hipMemcpyDeviceToHost));
    unsigned on_time = 0;
    std::map<unsigned, int> per_cu;   // (xcc, se, sh, cu) -> on-time waves
    std::map<unsigned, int> per_xcc;
    int simd_hist[4] = {0, 0, 0, 0};
    for (unsigned w = 0; w < total; ++w) {
      if (h[w].late) continue;
      ++on_time;
      const unsigned hw = h[w].hw_id;
      const unsigned cu_key = (h[w].xcc_id << 16) | ((hw >> 8) & 0xffu);  // cu_id[11:8] sh_id[12] se_id[15:13]
      per_cu[cu_key]++;
      per_xcc[h[w].xcc_id]++;
      simd_hist[(hw >> 4) & 3]++;
    }
    int cmin = 1 << 30, cmax = 0;
    for (auto& kv : per_cu) cmin = std::min(cmin, kv.second), cmax = std::max(cmax, kv.second);
    // cooperative check of the same grid
    CHECK(hipMemset(d_arrive, 0, 4));
    CHECK(hipMemset(d_dead, 0, 4));
    budget = 1000ull;
    hipError_t ce = hipLaunchCooperativeKernel(fn, dim3((unsigned)blocks), dim3((unsigned)c.threads), args, (unsigned)c.lds, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    printf("threads %3d lds %6d vgpr %3d sgpr>=%3d ask %2d/CU (%5u waves, %5d blocks): occupancy API %2d blocks/CU = %2d waves/CU; co-resident %5u "
           "(%s)  CUs seen %3zu  per-CU min %2d max %2d  per-XCD",
           c.threads, c.lds, c.vg, c.sg, c.waves_per_cu, total, blocks, occ, occ * wpb, on_time, on_time == total ? "ALL" : "NOT all",
           per_cu.size(), cmin, cmax);
    for (auto& kv : per_xcc) printf(" %d", kv.second);
    printf("  per-SIMD %d %d %d %d  coop launch: %s\n", simd_hist[0], simd_hist[1], simd_hist[2], simd_hist[3],
           ce == hipSuccess ? "accepted" : hipGetErrorString(ce));
    fflush(stdout);
  }
  return 0;
}
