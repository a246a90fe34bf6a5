// frames_capi.hip -- include/flame_frames.h: the result gather over the GPUs of one process, on RCCL.
//
// RCCL is bound at run time (dlopen + dlsym of the six entry points used): libflame_nltgv2_hip.so then loads -- and the
// single-GPU path runs -- on a machine without librccl, and a process that never gathers does not pay for mapping it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "flame_frames.h"
#include "roctx_ranges.hpp"

namespace {

// the part of rccl.h this file needs (ncclResult_t 0 = success; ncclFloat32 = 7 in nccl.h / rccl.h)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
constexpr int kNcclFloat32 = 7;
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
  std::mutex mu;
  bool load() {
    std::lock_guard<std::mutex> lock(mu);
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) {
      const char* why = dlerror();  // (reading it clears it: once)
      error = std::string("dlopen librccl: ") + (why ? why : "?");
      return false;
    }
    bool ok = true;
    auto sym = [&](const char* n) {
      void* p = dlsym(lib, n);
      if (!p) {
        ok = false;
        error = std::string("librccl lacks ") + n;
      }
      return p;
    };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
      dlclose(lib);
      lib = nullptr;
    }
    return ok;
  }
};
Rccl g_rccl;

}  // namespace

struct flame_frames_ctx {
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;
  std::vector<float*> send, recv;
  size_t vmax = 0;
  std::string error;
};

namespace {

int hip_fail(flame_frames_ctx* ctx, hipError_t e, const char* what) {
  ctx->error = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? FLAME_NLTGV2_ERR_OOM : FLAME_NLTGV2_ERR_HIP;
}
int nccl_fail(flame_frames_ctx* ctx, ncclResult_t r, const char* what) {
  ctx->error = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return FLAME_NLTGV2_ERR_HIP;
}
// The calls below walk over the devices of the context; the caller's current device is put back on every way out.
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define FHIP(ctx, expr, what)                         \
  do {                                                \
    hipError_t _e = (expr);                           \
    if (_e != hipSuccess) return hip_fail(ctx, _e, what); \
  } while (0)

}  // namespace

extern "C" {

int flame_frames_create(flame_frames_ctx** out, int n_devices, const int* devices, int32_t vmax) {
  if (!out) return FLAME_NLTGV2_ERR_INVALID_ARG;
  *out = nullptr;
  if (n_devices <= 0 || !devices || vmax <= 0) return FLAME_NLTGV2_ERR_INVALID_ARG;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return FLAME_NLTGV2_ERR_NO_DEVICE;
  for (int k = 0; k < n_devices; ++k)
    if (devices[k] < 0 || devices[k] >= count) return FLAME_NLTGV2_ERR_NO_DEVICE;
  flame_frames_ctx* ctx = new (std::nothrow) flame_frames_ctx();
  if (!ctx) return FLAME_NLTGV2_ERR_OOM;
  *out = ctx;  // (the caller can read the error text and must destroy it, whatever happens below)
  ctx->devices.assign(devices, devices + n_devices);
  ctx->vmax = (size_t)vmax;
  if (!g_rccl.load()) {
    ctx->error = g_rccl.error;
    return FLAME_NLTGV2_ERR_HIP;
  }
  DeviceGuard guard;
  ctx->streams.assign((size_t)n_devices, nullptr);
  ctx->send.assign((size_t)n_devices, nullptr);
  ctx->recv.assign((size_t)n_devices, nullptr);
  for (int k = 0; k < n_devices; ++k) {
    FHIP(ctx, hipSetDevice(devices[k]), "hipSetDevice");
    FHIP(ctx, hipStreamCreateWithFlags(&ctx->streams[(size_t)k], hipStreamNonBlocking), "hipStreamCreate");
    FHIP(ctx, hipMalloc((void**)&ctx->send[(size_t)k], sizeof(float) * ctx->vmax), "hipMalloc");
    FHIP(ctx, hipMalloc((void**)&ctx->recv[(size_t)k], sizeof(float) * ctx->vmax * (size_t)n_devices), "hipMalloc");
    FHIP(ctx, hipMemset(ctx->send[(size_t)k], 0, sizeof(float) * ctx->vmax), "hipMemset");
  }
  ctx->comms.assign((size_t)n_devices, nullptr);
  const ncclResult_t r = g_rccl.CommInitAll(ctx->comms.data(), n_devices, devices);
  if (r != 0) {
    ctx->comms.clear();
    return nccl_fail(ctx, r, "ncclCommInitAll");
  }
  return FLAME_NLTGV2_OK;
}

int flame_frames_destroy(flame_frames_ctx* ctx) {
  if (!ctx) return FLAME_NLTGV2_OK;
  DeviceGuard guard;
  for (size_t k = 0; k < ctx->devices.size(); ++k) {
    (void)hipSetDevice(ctx->devices[k]);
    if (k < ctx->streams.size() && ctx->streams[k]) (void)hipStreamSynchronize(ctx->streams[k]);
  }
  for (ncclComm_t c : ctx->comms)
    if (c) (void)g_rccl.CommDestroy(c);
  for (size_t k = 0; k < ctx->devices.size(); ++k) {
    (void)hipSetDevice(ctx->devices[k]);
    if (k < ctx->send.size() && ctx->send[k]) (void)hipFree(ctx->send[k]);
    if (k < ctx->recv.size() && ctx->recv[k]) (void)hipFree(ctx->recv[k]);
    if (k < ctx->streams.size() && ctx->streams[k]) (void)hipStreamDestroy(ctx->streams[k]);
  }
  delete ctx;
  return FLAME_NLTGV2_OK;
}

int flame_frames_count(const flame_frames_ctx* ctx) { return ctx ? (int)ctx->comms.size() : 0; }

int flame_frames_local_row(flame_frames_ctx* ctx, int k, void** row_device) {
  if (!ctx || !row_device || k < 0 || k >= (int)ctx->comms.size()) return FLAME_NLTGV2_ERR_INVALID_ARG;
  *row_device = ctx->send[(size_t)k];
  return FLAME_NLTGV2_OK;
}

int flame_frames_stream(flame_frames_ctx* ctx, int k, void** hip_stream) {
  if (!ctx || !hip_stream || k < 0 || k >= (int)ctx->comms.size()) return FLAME_NLTGV2_ERR_INVALID_ARG;
  *hip_stream = ctx->streams[(size_t)k];
  return FLAME_NLTGV2_OK;
}

int flame_frames_gather(flame_frames_ctx* ctx) {
  flame_hip::RoctxRange roctx_range_("flame_frames_gather");
  if (!ctx || ctx->comms.empty()) return FLAME_NLTGV2_ERR_INVALID_ARG;
  ncclResult_t r = g_rccl.GroupStart();
  if (r != 0) return nccl_fail(ctx, r, "ncclGroupStart");
  ncclResult_t first = 0;
  for (size_t k = 0; k < ctx->comms.size(); ++k) {
    r = g_rccl.AllGather(ctx->send[k], ctx->recv[k], ctx->vmax, kNcclFloat32, ctx->comms[k], ctx->streams[k]);
    if (r != 0 && first == 0) first = r;
  }
  r = g_rccl.GroupEnd();
  if (first != 0) return nccl_fail(ctx, first, "ncclAllGather");
  if (r != 0) return nccl_fail(ctx, r, "ncclGroupEnd");
  return FLAME_NLTGV2_OK;
}

int flame_frames_wait(flame_frames_ctx* ctx) {
  flame_hip::RoctxRange roctx_range_("flame_frames_wait");
  if (!ctx) return FLAME_NLTGV2_ERR_INVALID_ARG;
  DeviceGuard guard;
  for (size_t k = 0; k < ctx->streams.size(); ++k) {
    FHIP(ctx, hipSetDevice(ctx->devices[k]), "hipSetDevice");
    FHIP(ctx, hipStreamSynchronize(ctx->streams[k]), "hipStreamSynchronize");
  }
  return FLAME_NLTGV2_OK;
}

int flame_frames_gathered(flame_frames_ctx* ctx, int k, void** block_device) {
  if (!ctx || !block_device || k < 0 || k >= (int)ctx->comms.size()) return FLAME_NLTGV2_ERR_INVALID_ARG;
  *block_device = ctx->recv[(size_t)k];
  return FLAME_NLTGV2_OK;
}

int flame_frames_download(flame_frames_ctx* ctx, int k, float* host_block) {
  if (!ctx || !host_block || k < 0 || k >= (int)ctx->comms.size()) return FLAME_NLTGV2_ERR_INVALID_ARG;
  DeviceGuard guard;
  FHIP(ctx, hipSetDevice(ctx->devices[(size_t)k]), "hipSetDevice");
  FHIP(ctx, hipStreamSynchronize(ctx->streams[(size_t)k]), "hipStreamSynchronize");
  FHIP(ctx, hipMemcpy(host_block, ctx->recv[(size_t)k], sizeof(float) * ctx->vmax * ctx->comms.size(), hipMemcpyDeviceToHost), "hipMemcpy");
  return FLAME_NLTGV2_OK;
}

const char* flame_frames_last_error_text(const flame_frames_ctx* ctx) { return ctx ? ctx->error.c_str() : ""; }

}  // extern "C"
