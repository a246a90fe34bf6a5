// flame_hip/frame_gather.hpp -- the multi-GPU side of the drop-in for a host that stays C++: one process, one
// DeviceGraph per GPU, frames sharded one per GPU with nothing exchanged while they iterate, and the read-back
// Flame::update does after the solve (/root/reference/src/flame/flame.cc:372-380: idepth = x * graph_scale) gathered to
// every GPU with one grouped ncclAllGather (RCCL over xGMI; include/flame_frames.h).
//
//   flame_hip::FrameGather gather(devices, vmax);                      // once
//   for k: solver[k].setStream(gather.stream(k));                      // export ordered before the gather, no host sync
//          solver[k].setExportTarget(gather.localRow(k), graph_scale); // every run() leaves x * graph_scale in the row
//   per step:  for k: solver[k].runAsync(params, n);   gather.gather();   ...   gather.wait();
//              gather.download(k, &block)  /  gather.gathered(k)  -> row j = frame j, first V_j entries
#ifndef FLAME_HIP_FRAME_GATHER_HPP_
#define FLAME_HIP_FRAME_GATHER_HPP_

#include <string>
#include <vector>

#include "flame_frames.h"
#include "flame_hip/nltgv2_l1_graph_regularizer.hpp"

namespace flame_hip {

class FrameGather {
 public:
  FrameGather(const std::vector<int>& devices, int32_t vmax) : ctx_(nullptr), vmax_(vmax) {
    const int rc = flame_frames_create(&ctx_, static_cast<int>(devices.size()), devices.data(), vmax);
    if (rc != 0) {
      const std::string why = std::string("flame_frames_create: ") + flame_frames_last_error_text(ctx_);
      flame_frames_destroy(ctx_);
      ctx_ = nullptr;
      throw Error(rc, why.c_str());
    }
  }
  ~FrameGather() { flame_frames_destroy(ctx_); }
  FrameGather(const FrameGather&) = delete;
  FrameGather& operator=(const FrameGather&) = delete;

  int size() const { return flame_frames_count(ctx_); }
  int32_t vmax() const { return vmax_; }
  void* localRow(int k) {
    void* p = nullptr;
    check(flame_frames_local_row(ctx_, k, &p), "local_row");
    return p;
  }
  void* stream(int k) {
    void* s = nullptr;
    check(flame_frames_stream(ctx_, k, &s), "stream");
    return s;
  }
  void gather() { check(flame_frames_gather(ctx_), "gather"); }
  void wait() { check(flame_frames_wait(ctx_), "wait"); }
  const float* gathered(int k) {
    void* p = nullptr;
    check(flame_frames_gathered(ctx_, k, &p), "gathered");
    return static_cast<const float*>(p);
  }
  void download(int k, std::vector<float>* block) {
    block->resize(static_cast<size_t>(size()) * static_cast<size_t>(vmax_));
    check(flame_frames_download(ctx_, k, block->data()), "download");
  }

 private:
  void check(int rc, const char* what) const {
    if (rc != 0) throw Error(rc, (std::string(what) + ": " + flame_frames_last_error_text(ctx_)).c_str());
  }
  flame_frames_ctx* ctx_;
  int32_t vmax_;
};

}  // namespace flame_hip

#endif  // FLAME_HIP_FRAME_GATHER_HPP_
