// tools/coop_exit_repro.hip -- the exit crash of profiled processes, without this library (profiles/r06_segv.txt).
//   hipcc --offload-arch=gfx950 -o coop_exit_repro coop_exit_repro.hip
//   ./coop_exit_repro coop ; echo $?                                   -> 0
//   rocprofv3 --kernel-trace -- ./coop_exit_repro plain ; echo $?      -> 0
//   rocprofv3 --kernel-trace -- ./coop_exit_repro coop ; echo $?       -> 139 (SIGSEGV in libhsa-runtime64, under amd::Runtime::tearDown)
// One empty kernel, launched once: plainly, or cooperatively (hipLaunchCooperativeKernel makes the runtime create the device's
// cooperative queue).  Nothing else: no streams, no allocations.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

__global__ void k_empty() {}

int main(int argc, char** argv) {
  const bool coop = argc > 1 && !strcmp(argv[1], "coop");
  void* no_args[] = {nullptr};
  hipError_t e;
  if (coop) e = hipLaunchCooperativeKernel((const void*)k_empty, dim3(8), dim3(64), no_args, 0, nullptr);
  else e = hipLaunchKernel((const void*)k_empty, dim3(8), dim3(64), no_args, 0, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  std::printf("%s launch: %s\n", coop ? "cooperative" : "plain", hipGetErrorString(e));
  return e == hipSuccess ? 0 : 1;
}
