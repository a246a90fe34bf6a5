// roctx_ranges.hpp -- named ranges around the library's entry points for rocprofv3 --marker-trace (SURVEY.md section 5:
// what the reference's StatsTracker tick/tock pairs, utils/stats_tracker.h:100-121, give its maintainers: a timeline of
// "update", "syncGraph", "interpolateMesh", ... per frame).  The marker library is bound at run time
// (dlopen("librocprofiler-sdk-roctx.so"), then the older libroctx64.so), like RCCL in frames_capi.hip: the solver library
// loads and runs without it, and FLAME_NLTGV2_ROCTX=0 keeps it from being looked for at all.  A push / pop pair costs well
// under a microsecond when no tool is attached.
#ifndef FLAME_AMD_ROCTX_RANGES_HPP_
#define FLAME_AMD_ROCTX_RANGES_HPP_

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace flame_hip {

struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};

inline const RoctxApi& roctx_api() {
  static RoctxApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* off = std::getenv("FLAME_NLTGV2_ROCTX");
    if (off && std::strcmp(off, "0") == 0) return;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (!h) continue;
      auto push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      auto pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (push && pop) {
        api.push = push, api.pop = pop;
        return;
      }
      dlclose(h);
    }
  });
  return api;
}

// RAII range: flame_hip::RoctxRange r("flame_nltgv2_run");
class RoctxRange {
 public:
  explicit RoctxRange(const char* name) : on_(roctx_api().push != nullptr) {
    if (on_) roctx_api().push(name);
  }
  ~RoctxRange() {
    if (on_) roctx_api().pop();
  }
  RoctxRange(const RoctxRange&) = delete;
  RoctxRange& operator=(const RoctxRange&) = delete;

 private:
  bool on_;
};

}  // namespace flame_hip

#endif  // FLAME_AMD_ROCTX_RANGES_HPP_
