// tools/sync_latency.hip -- how long the host takes to notice that a stream has drained: hipStreamSynchronize against a spin on
// hipStreamQuery / hipEventQuery / a word the kernel writes into pinned host memory.  Each case: a kernel of ~100 us is launched, the host
// sleeps 60 us (so that it starts waiting while the kernel runs, as a frame thread does behind a round in flight), waits, then launches
// an empty kernel; reported: the device-side gap between the end of the first and the start of the second kernel (wall_clock64, 100 MHz)
// and the host's own time from wake-up to the launch call's return.  hipcc -O2 --offload-arch=gfx950 -o build/sync_latency tools/sync_latency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void k_busy(unsigned long long ticks, unsigned long long* t_end, volatile unsigned* host_flag, unsigned seq) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  *t_end = wall_clock64();
  if (host_flag) {
    __threadfence_system();
    *host_flag = seq;
  }
}
__global__ void k_stamp(unsigned long long* t_start) { *t_start = wall_clock64(); }

int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned long long *d_t, h_t[2];
  hipMalloc(&d_t, 16);
  unsigned* h_flag;
  hipHostMalloc((void**)&h_flag, 64, hipHostMallocDefault);
  *h_flag = 0;
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  const char* names[] = {"hipStreamSynchronize", "spin on hipStreamQuery", "spin on hipEventQuery", "spin on a pinned word the kernel writes"};
  for (int mode = 0; mode < 4; ++mode) {
    std::vector<double> gap, host;
    for (int rep = 0; rep < 60; ++rep) {
      const unsigned seq = (unsigned)(mode * 1000 + rep + 1);
      hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s, 10000ull, d_t, mode == 3 ? h_flag : nullptr, seq);
      if (mode == 2) hipEventRecord(ev, s);
      std::this_thread::sleep_for(std::chrono::microseconds(60));
      if (mode == 0) hipStreamSynchronize(s);
      if (mode == 1) while (hipStreamQuery(s) == hipErrorNotReady) {}
      if (mode == 2) while (hipEventQuery(ev) == hipErrorNotReady) {}
      if (mode == 3) while (*(volatile unsigned*)h_flag != seq) {}
      const auto a = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, d_t + 1);
      const auto b = std::chrono::steady_clock::now();
      hipStreamSynchronize(s);
      hipMemcpy(h_t, d_t, 16, hipMemcpyDeviceToHost);
      if (rep >= 10) gap.push_back((double)(h_t[1] - h_t[0]) / 100.0), host.push_back(std::chrono::duration<double, std::micro>(b - a).count());
    }
    std::sort(gap.begin(), gap.end()), std::sort(host.begin(), host.end());
    std::printf("%-44s end of kernel -> start of the next: median %5.1f us (min %5.1f, p90 %5.1f); the launch call itself %4.1f us\n", names[mode],
                gap[gap.size() / 2], gap[0], gap[gap.size() * 9 / 10], host[host.size() / 2]);
  }
  return 0;
}
