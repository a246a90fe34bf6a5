// flame_hip/frame_gather.hpp -- the multi-GPU side of the drop-in for a host that stays C++: one process, one
// DeviceGraph per GPU, frames sharded one per GPU with nothing exchanged while they iterate, and the read-back
// Flame::update does after the solve (/root/reference/src/flame/flame.cc:372-380: idepth = x * graph_scale) gathered to
// every GPU with one grouped ncclAllGather (RCCL over xGMI; include/flame_frames.h).
//
//   flame_hip::FrameGather gather(devices, vmax);                      // once
//   for k: solver[k].setStream(gather.stream(k));                      // export ordered before the gather, no host sync
//          solver[k].setExportTarget(gather.localRow(k), graph_scale); // every run() leaves x * graph_scale in the row
//   per step:  for k: solver[k].runAsync(params, n);   gather.gather();   ...   gather.settle(solvers);
//              gather.download(k, &block)  /  gather.gathered(k)  -> row j = frame j, first V_j entries
//
// ORDERING.  gather() right behind runAsync() all-gathers the export rows of runs nobody has checked yet.  A persistent run
// whose neighbour wait expires (the GPU shared with something that keeps its CUs full) leaves before its epilogue writes
// the row; DeviceGraph::sync() then takes the run back, redoes the steps and re-exports -- AFTER the collective has already
// delivered the previous frame's row to every GPU.  So the gathered block may only be consumed after settle(): it syncs
// every solver, and if any of them had to redo its run (DeviceGraph::replays() moved) it gathers again.  wait() alone is
// for callers that synced their solvers before gather().
#ifndef FLAME_HIP_FRAME_GATHER_HPP_
#define FLAME_HIP_FRAME_GATHER_HPP_

#include <memory>
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
  // gather() that remembers how often each solver has had to redo a run so far: the baseline of settle()
  template <class Solvers>
  void gather(Solvers& solvers) {
    replays_at_gather_.clear();
    for (auto& s : solvers) replays_at_gather_.push_back(deref(s).replays());
    gather();
  }
  // Completes the gather and makes sure it carried the rows of runs that really finished: syncs every solver; if one of
  // them was taken back and redone since the gather was issued, the (re-exported) rows are gathered again.  Returns the
  // number of re-gathers (0 or 1).
  template <class Solvers>
  int settle(Solvers& solvers) {
    std::vector<int> before = replays_at_gather_;
    if (before.size() != static_cast<size_t>(solvers.size())) {
      before.clear();
      for (auto& s : solvers) before.push_back(deref(s).replays());
    }
    bool again = false;
    size_t k = 0;
    for (auto& s : solvers) {
      deref(s).sync();
      again = again || deref(s).replays() != before[k++];
    }
    replays_at_gather_.clear();
    wait();
    if (!again) return 0;
    gather();
    wait();
    return 1;
  }
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
  template <class T> static T& deref(T& t) { return t; }
  template <class T> static T& deref(T* t) { return *t; }
  template <class T> static T& deref(std::unique_ptr<T>& t) { return *t; }
  flame_frames_ctx* ctx_;
  int32_t vmax_;
  std::vector<int> replays_at_gather_;
};

}  // namespace flame_hip

#endif  // FLAME_HIP_FRAME_GATHER_HPP_
