// host_workers.hpp -- a few persistent worker threads for the host-side stages that sit between GPU stages of a frame (the
// Delaunay triangulation, the index maps of the per-frame graph sync): fork-join over a handful of jobs, the calling thread
// included.  Creating a thread per call would cost what the parallel stage saves (~40 us each); these sleep on a condition
// variable between calls.  FLAME_DELAUNAY_THREADS caps the pool (default: the machine's cores, at most 32).
#ifndef FLAME_AMD_HOST_WORKERS_HPP_
#define FLAME_AMD_HOST_WORKERS_HPP_

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace flame_hip {

class Workers {
 public:
  static Workers& get() {
    static Workers w;
    return w;
  }
  int threads() const { return n_threads_; }
  // runs job(0) .. job(n - 1), the calling thread included; returns when all are done
  void run(int n, const std::function<void(int)>& job) {
    std::unique_lock<std::mutex> call(call_mtx_);  // one parallel region at a time
    if (n_threads_ <= 1 || n <= 1) {
      for (int i = 0; i < n; ++i) job(i);
      return;
    }
    // Every region has its own state {job, n, next, pending}; a worker takes the pointer under the lock when it wakes up, so a
    // worker that was preempted between two regions only ever sees the (finished) region it belongs to: its next index is
    // >= that region's n and it goes back to sleep without touching the new region's job or counters.
    auto region = std::make_shared<Region>();
    region->job = &job, region->n = n, region->pending = n;
    {
      std::lock_guard<std::mutex> lk(mtx_);
      region_ = region, ++generation_;
    }
    hint_.store(generation_, std::memory_order_release);  // (workers still spinning behind the previous region see it without a wake-up)
    cv_.notify_all();
    work(*region);
    std::unique_lock<std::mutex> lk(mtx_);
    done_cv_.wait(lk, [&] { return region->pending == 0; });
    region_.reset();
  }

 private:
  Workers() {
    int want = (int)std::thread::hardware_concurrency();
    if (const char* e = std::getenv("FLAME_DELAUNAY_THREADS")) want = std::atoi(e);
    n_threads_ = std::max(1, std::min(want, 32));
    for (int i = 1; i < n_threads_; ++i) pool_.emplace_back([this] { loop(); });
  }
  ~Workers() {
    {
      std::lock_guard<std::mutex> lk(mtx_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : pool_) t.join();
  }
  struct Region {
    const std::function<void(int)>* job = nullptr;
    int n = 0;
    std::atomic<int> next{0};
    int pending = 0;  // under mtx_
  };
  void work(Region& r) {
    for (;;) {
      const int i = r.next.fetch_add(1);
      if (i >= r.n) return;  // (r.job is only dereferenced for an index that was still outstanding: run() is still waiting)
      (*r.job)(i);
      std::lock_guard<std::mutex> lk(mtx_);
      if (--r.pending == 0) done_cv_.notify_all();
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      std::shared_ptr<Region> r;
      // A short spin before the sleep: stages that follow each other within ~0.1 ms (the triangulator's callers in a loop, the sync's
      // index maps right behind it) find the team awake -- a wake-up of 31 sleeping threads through one condition variable is 30-60 us
      // before the last of them runs (round 6: bench `delaunay.triangulate_ms` against `.every_2_ms`).  Bounded: a pool that is
      // called once per frame sleeps through the frame as before.
      {
        const auto t0 = std::chrono::steady_clock::now();
        for (int spins = 0; hint_.load(std::memory_order_acquire) == seen; ++spins) {
          if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kSpinUs)) break;
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
      }
      {
        std::unique_lock<std::mutex> lk(mtx_);
        cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
        r = region_;
      }
      if (r) work(*r);
    }
  }
  std::mutex call_mtx_, mtx_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> pool_;
  std::shared_ptr<Region> region_;  // the region in flight (under mtx_)
  int n_threads_ = 1;
  static constexpr int kSpinUs = 120;
  std::atomic<uint64_t> hint_{0};  // = generation_, readable without the lock
  uint64_t generation_ = 0;
  bool stop_ = false;
};


// job(begin, end) over [0, n) in `parts` contiguous chunks
inline void parallel_chunks(int64_t n, int parts, const std::function<void(int64_t, int64_t)>& job) {
  parts = (int)std::max<int64_t>(1, std::min<int64_t>(parts, n));
  Workers::get().run(parts, [&](int k) { job(n * k / parts, n * (k + 1) / parts); });
}

}  // namespace flame_hip

#endif  // FLAME_AMD_HOST_WORKERS_HPP_
