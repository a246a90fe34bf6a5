// flame_hip/solver_loop.hpp -- the reference's solver thread (Flame::Flame, /root/reference/src/flame/flame.cc:99-112:
// `while (true) { lock graph_mtx_; step(params, &graph_); }`, joined in ~Flame flame.cc:120-124 -- a loop that cannot
// exit) as a class that can: the device-side counterpart of that thread for a pipeline that keeps `Graph graph_` on the
// host (flame.h:536) and lets the GPU iterate on its image of it.
//
//   flame_hip::SolverLoop<Graph> loop(&graph_, &graph_mtx_, params);   // in Flame::Flame
//   loop.start();
//   ...
//   {  std::lock_guard<std::mutex> lock(graph_mtx_);                    // Flame::update, as today (flame.cc:302-381; graph_mtx_ is a
//                                                                       //  plain std::mutex, flame.h:539 -- update_mtx_ is the recursive one)
//      loop.readBack();                  // x, w1, w2, q of the device image -> graph_   (before reading / editing it)
//      syncGraph(); ...                  // edit graph_
//      loop.markDirty();                 // the loop uploads the edited graph before its next round
//   }
//   loop.stop();                         // or the destructor: sets the flag, wakes the thread, joins
//
// DEVICE MODE (round 5): a pipeline whose per-frame graph edits happen ON the device (DeviceGraph::syncPrepare / syncCommit,
// projectGraph, interpolateMeshBegin / End) has no host graph to re-upload: construct the loop with a null graph, hand it the first
// graph through withDevice() + deviceReady(), and do every later edit through withDevice() -- the solver thread keeps iterating on
// whatever the device image is between two such calls (markDirty() is never needed, nothing is ever re-uploaded):
//
//   flame_hip::SolverLoop<flame_hip::FlatGraph> loop(nullptr, &graph_mtx_, params, /*iters_per_round=*/256);
//   loop.start();
//   loop.withDevice([&](DeviceGraph& d, uint64_t) { d.upload(first_graph); });  loop.deviceReady();
//   per frame:  loop.withDevice([&](DeviceGraph& d, uint64_t iterations_so_far) { d.syncPrepare(...); });   // the solver goes on
//               ... triangulate the next frame, track features ...
//               loop.withDevice([&](DeviceGraph& d, uint64_t) { d.syncCommit(); });
//               loop.withDevice([&](DeviceGraph& d, uint64_t) { d.interpolateMeshBegin(...); });  ...  interpolateMeshEnd
//   loop.utilization(r)  -- iterations per second since deviceReady() over r, the rate of the solver alone on such a graph: the share
//                           of the time the device spent iterating (the rest: the frame thread's calls that settle the solver)
// In device mode the loop never blocks in the context: it ENQUEUES rounds (DeviceGraph::runAsync) and keeps two in flight
// (DeviceGraph::runsInFlight, a non-blocking query), so the device goes from one round to the next without waiting for the host, and
// the host part of the frame thread's calls (validation, staging, the triangles' upload) overlaps the rounds already enqueued; a
// call that needs the settled state waits for them inside the library.  Results are checked every kRoundsPerCheck rounds at the
// latest (sync(): an expired run is redone there).
// The callback's second argument is the number of iterations applied to the device image once the calls before it have settled --
// exact: rounds are counted when they are enqueued, under the lock, and every call that reads or edits the state settles them first --
// which is what lets a test replay the free-running loop on the CPU checker (tests/cpp/frame_loop_test.cc).
//
// Locking: the caller's mutex protects the HOST graph exactly as in the reference; the loop takes it only to upload.  An
// internal mutex serialises the calls into the (not thread-safe) device context: the loop holds it for one round of
// `iters_per_round` iterations (~1.3 us each at 640x480), readBack() waits for that round at most (the loop gives way to
// a waiting caller before it takes the mutex again).  Lock order is always
// caller's mutex, then the internal one.  A graph edited without markDirty() is not written to: readBack() compares vertex
// and edge counts and the edit generation with what was uploaded and returns false instead.
#ifndef FLAME_HIP_SOLVER_LOOP_HPP_
#define FLAME_HIP_SOLVER_LOOP_HPP_

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>

#include "flame_hip/nltgv2_l1_graph_regularizer.hpp"

namespace flame_hip {

template <class Graph, class GraphMutex = std::mutex>  // flame.h:539: `std::mutex graph_mtx_;`
class SolverLoop {
 public:
  typedef flame::optimizers::nltgv2_l1_graph_regularizer::hip::Params Params;
  typedef flame::optimizers::nltgv2_l1_graph_regularizer::hip::DeviceGraph DeviceGraph;

  // max_rounds_per_upload = 0: iterate until stopped (the reference's behaviour); > 0: that many rounds after every
  // upload, then idle until the next markDirty() (a fixed iteration budget per frame; also what makes tests exact).
  SolverLoop(Graph* graph, GraphMutex* graph_mtx, const Params& params, int iters_per_round = 32, int max_rounds_per_upload = 0,
             int device = 0)
      : graph_(graph), graph_mtx_(graph_mtx), params_(params), iters_per_round_(iters_per_round > 0 ? iters_per_round : 1),
        max_rounds_(max_rounds_per_upload), dev_(device) {}
  ~SolverLoop() { stop(); }
  SolverLoop(const SolverLoop&) = delete;
  SolverLoop& operator=(const SolverLoop&) = delete;

  void start() {
    std::thread dead;
    {
      std::lock_guard<std::mutex> lk(state_mtx_);
      if (thread_.joinable() && !exited_) return;  // already running
      dead.swap(thread_);                         // a thread that ended on its own (an error: see error()) is joined and replaced
    }
    if (dead.joinable()) dead.join();
    std::lock_guard<std::mutex> lk(state_mtx_);
    if (thread_.joinable()) return;
    stop_.store(false);
    exited_ = false;
    error_.clear();
    thread_ = std::thread([this] { run(); });
  }
  // Idempotent; returns after the thread has exited (at most one round later).  May be called with the graph mutex held: the
  // loop never blocks on that mutex (it polls it, see run()), so the join cannot deadlock against the caller.
  void stop() {
    std::thread t;
    {
      std::lock_guard<std::mutex> lk(state_mtx_);
      stop_.store(true);
      t.swap(thread_);
    }
    cv_.notify_all();
    if (t.joinable()) t.join();
  }
  // Device mode, before start(): instead of rounds of iters_per_round iterations, two in flight, the loop keeps ONE open run going
  // (DeviceGraph::runOpen: it ends when a withDevice() call needs the solver settled, at most max_iters iterations) -- no start-up and no
  // gap between rounds (12 us per round at 640x480), and a call waits ~0.15 ms for the solver instead of for up to two rounds.  Where
  // an open run is not applicable (DeviceGraph::runOpen says so) the loop enqueues rounds as before.  The callbacks' iteration count
  // then LAGS behind (an open run is counted once it is settled): a callback that needs the exact number asks DeviceGraph::iterations()
  // after its first call that settled the solver.
  // (max_iters: what anybody ELSE who makes the device wait -- another context's hipFree -- waits for at most: 4 096 iterations are 4-6 ms)
  void useOpenRuns(int max_iters = 1 << 12) { open_max_ = max_iters > 0 ? (max_iters & ~1) : 0; }
  bool running() const {
    std::lock_guard<std::mutex> lk(state_mtx_);
    return thread_.joinable() && !exited_ && !stop_.load();
  }

  // Caller holds graph_mtx: the graph was edited (vertices, edges, data terms); the loop re-uploads it before iterating on.
  void markDirty() {
    {
      std::lock_guard<std::mutex> lk(state_mtx_);
      ++dirty_generation_;
    }
    cv_.notify_all();
  }
  // Caller holds graph_mtx: copies x, w, x_bar, w_bar, x_prev, w_prev, q of the device image into the graph.  Returns
  // false -- and leaves the graph alone -- when the device image is not of this graph (edited since the last upload, or
  // nothing uploaded yet).
  bool readBack() {
    if (graph_ == nullptr) return false;  // (device mode: results leave through interpolateMesh / the export target)
    CallerAccess dev_lk(this);
    uint64_t want;
    {
      std::lock_guard<std::mutex> lk(state_mtx_);
      want = dirty_generation_;
    }
    if (!dev_.matches(*graph_, want)) return false;
    dev_.download(graph_, want);
    return true;
  }
  // iterations done on the device image since start(); error text of the thread, if it stopped on one
  // Exclusive access to the device image for one call (see DEVICE MODE above): f(DeviceGraph&, iterations applied so far).
  template <class F>
  void withDevice(F&& f) {
    CallerAccess dev_lk(this);
    f(dev_, iterations_.load());
    recount();
    refill();
  }
  // The same with a second part that does not need the solver stopped: f runs on the settled device image, the queue is filled
  // again, THEN g runs (still with exclusive access to the context) beside the rounds just enqueued.  Made for
  //     loop.withDevice([&](DeviceGraph& d, uint64_t it) { d.syncCommit(); },
  //                     [&](DeviceGraph& d, uint64_t it) { d.interpolateMeshBegin(tris, rows, cols, scale); });
  // with FLAME_NLTGV2_OPT_MESH_STATE = 1 on the context: the mesh is of the state f left (both callbacks get that state's iteration
  // count), and the host time of its preparation -- the triangles' upload, the rasteriser's launches: ~65 us at 640x480 -- no longer
  // stands between the commit and the solver's next round.  A g that settles the solver after all (any call that reads or edits the
  // state does) is correct, just not faster: the queue is filled once more behind it.
  template <class F, class G>
  void withDevice(F&& f, G&& g) {
    CallerAccess dev_lk(this);
    const uint64_t it = iterations_.load();
    f(dev_, it);
    recount();
    refill();
    g(dev_, open_max_ > 0 ? iterations_.load() : it);
    recount();
    refill();
  }
  // Device mode: the image uploaded through withDevice() is what the loop iterates on from now on.
  void deviceReady() {
    {
      std::lock_guard<std::mutex> lk(state_mtx_);
      device_ready_ = true;
      t_ready_ = std::chrono::steady_clock::now();
      busy_ns_.store(0);
      iterations_at_ready_.store(iterations_.load());
    }
    cv_.notify_all();
  }
  // Host mode: share of the wall time since start() that the solver thread spent inside its blocking rounds.  (Device mode enqueues
  // its rounds and never blocks: see utilization().)
  double busyFraction() const {
    std::chrono::steady_clock::time_point t0;
    {
      std::lock_guard<std::mutex> lk(state_mtx_);
      t0 = t_ready_;
    }
    const double wall = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
    return wall > 0 ? static_cast<double>(busy_ns_.load()) / wall : 0.0;
  }
  // Device mode: the share of the time since deviceReady() the device spent on solver iterations, given the solver's own rate on such a
  // graph (iterations per second of an undisturbed run, measured by the caller).
  double utilization(double free_running_iters_per_s) const {
    std::chrono::steady_clock::time_point t0;
    {
      std::lock_guard<std::mutex> lk(state_mtx_);
      t0 = t_ready_;
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double done = static_cast<double>(iterations_.load() - iterations_at_ready_.load());
    return wall > 0 && free_running_iters_per_s > 0 ? done / wall / free_running_iters_per_s : 0.0;
  }
  uint64_t iterations() const { return iterations_.load(); }
  uint64_t uploads() const { return uploads_.load(); }
  uint64_t openRuns() const { return open_runs_.load(); }  // device mode with useOpenRuns(): how many open runs the loop has started
  std::string error() const {
    std::lock_guard<std::mutex> lk(state_mtx_);
    return error_;
  }
  void setParams(const Params& p) {
    CallerAccess dev_lk(this);
    params_ = p;
  }

 private:
  // std::mutex is not fair: the loop re-takes dev_mtx_ microseconds after releasing it and a caller blocked on it can
  // starve for seconds.  Callers announce themselves; the loop lets them pass before its next round.
  struct CallerAccess {
    explicit CallerAccess(SolverLoop* s) : s_(s) {
      s_->callers_waiting_.fetch_add(1);
      s_->dev_mtx_.lock();
    }
    ~CallerAccess() {
      s_->dev_mtx_.unlock();
      s_->callers_waiting_.fetch_sub(1);
    }
    SolverLoop* s_;
  };
  void give_way() const {
    while (callers_waiting_.load() > 0 && !stop_.load()) std::this_thread::yield();
  }
  void run() {
    int rounds_left = 0;
    try {
      while (!stop_.load()) {
        uint64_t dirty;
        {
          std::lock_guard<std::mutex> lk(state_mtx_);
          dirty = dirty_generation_;
        }
        bool have = false;
        give_way();
        if (graph_ == nullptr) {  // device mode: nothing is ever uploaded from the host; idle until the first image stands
          {
            std::unique_lock<std::mutex> lk(state_mtx_);
            if (!device_ready_) {
              cv_.wait_for(lk, std::chrono::milliseconds(50), [&] { return stop_.load() || device_ready_; });
              continue;
            }
          }
          // rounds are enqueued, two in flight; the thread holds the device for microseconds at a time
          bool enqueued;
          {
            std::lock_guard<std::mutex> dev_lk(dev_mtx_);
            enqueued = top_up();
          }
          if (!enqueued) std::this_thread::sleep_for(std::chrono::microseconds(20));
          continue;
        } else {
          std::lock_guard<std::mutex> dev_lk(dev_mtx_);
          have = dev_.generation() == dirty && uploaded_once_;
        }
        if (!have) {  // (re)upload under the caller's mutex: nobody edits the graph meanwhile
          // (polled, never blocked on: a caller may hold the graph mutex while it stop()s -- or destroys -- this loop)
          std::unique_lock<GraphMutex> g_lk(*graph_mtx_, std::try_to_lock);
          while (!g_lk.owns_lock()) {
            if (stop_.load()) break;
            std::this_thread::yield();
            g_lk.try_lock();
          }
          if (!g_lk.owns_lock()) break;  // told to stop while waiting for the graph
          std::lock_guard<std::mutex> dev_lk(dev_mtx_);
          {
            std::lock_guard<std::mutex> lk(state_mtx_);
            dirty = dirty_generation_;  // the edit we are about to see
          }
          dev_.upload(*graph_, dirty);
          uploaded_once_ = true;
          uploads_.fetch_add(1);
          rounds_left = max_rounds_;
        }
        size_t V = 0, E = 0;
        bool idle = false;
        {
          std::lock_guard<std::mutex> dev_lk(dev_mtx_);
          V = dev_vertices();
          if (V == 0 || (max_rounds_ > 0 && rounds_left == 0)) {
            idle = true;
          } else {
            const auto t0 = std::chrono::steady_clock::now();
            dev_.run(params_, iters_per_round_);
            busy_ns_.fetch_add(static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count()));
            iterations_.fetch_add(static_cast<uint64_t>(iters_per_round_));
            if (max_rounds_ > 0) --rounds_left;
          }
        }
        (void)E;
        if (idle) {  // nothing to do until the graph changes or we are told to stop
          std::unique_lock<std::mutex> lk(state_mtx_);
          cv_.wait_for(lk, std::chrono::milliseconds(50), [&] { return stop_.load() || dirty_generation_ != dirty; });
        }
      }
      if (graph_ == nullptr) {  // device mode: the rounds still in flight are checked before the thread leaves
        std::lock_guard<std::mutex> dev_lk(dev_mtx_);
        if (dev_vertices() > 0) dev_.sync();
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> lk(state_mtx_);
      error_ = e.what();
      stop_.store(true);
    }
    std::lock_guard<std::mutex> lk(state_mtx_);
    exited_ = true;
  }
  // (dev_mtx_ held) device mode: a call that settled the solver leaves the queue empty -- the caller, who holds the device anyway, fills
  // it again before it lets go (the solver thread would need a wake-up for it: ~0.1 ms of a standing solver per call).  Not after the
  // solver thread has ended, on an error or a stop: nobody would ever check those rounds; and what top_up() throws -- its periodic
  // sync() -- is the LOOP's error, kept in error(): the caller's own call has succeeded and must not fail for it.
  void refill() {
    bool ready;
    {
      std::lock_guard<std::mutex> lk(state_mtx_);
      ready = device_ready_ && !stop_.load() && !exited_ && error_.empty() && thread_.joinable();
    }
    if (graph_ != nullptr || !ready) return;
    try {
      top_up();
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> lk(state_mtx_);
      error_ = e.what();
      stop_.store(true);
    }
  }
  // (dev_mtx_ held) device mode: enqueues rounds until two are in flight; true if it enqueued any
  // (dev_mtx_ held) open runs: the library counts (an open run when it is settled); rounds are counted here, when they are enqueued
  void count() {
    if (open_max_ > 0 && graph_ == nullptr) iterations_.store(count_base_ + dev_.iterations());
  }
  void recount() {  // (behind a caller's callback: it may have changed the graph)
    count();
    open_na_ = false;
  }
  bool top_up() {
    bool any = false;
    if (dev_vertices() == 0) return false;
    if (open_max_ > 0) {
      if (!open_na_) {
        bool open_now = false;
        dev_.iterations(&open_now);
        if (open_now && dev_.runsInFlight() > 0) return false;  // the open run goes on
        // (nothing open in flight -- or it reached its bound by itself: settled and counted by the call below.  Where an open run does not
        //  apply the call returns at once, WITHOUT waiting for rounds that may be in flight)
        const bool opened = dev_.runOpen(params_, open_max_);
        count();
        if (opened) {
          open_runs_.fetch_add(1);
          return true;
        }
        open_na_ = true;  // not applicable to this graph (asked again after the next withDevice(): the graph may have changed)
      }
      // rounds, counted by the library as well
      while (dev_.runsInFlight() < 2) {
        dev_.runAsync(params_, iters_per_round_);
        any = true;
      }
      count();
      return any;
    }
    while (dev_.runsInFlight() < 2) {
      dev_.runAsync(params_, iters_per_round_);
      iterations_.fetch_add(static_cast<uint64_t>(iters_per_round_));
      any = true;
      if (++rounds_unchecked_ >= kRoundsPerCheck) {
        dev_.sync();
        rounds_unchecked_ = 0;
      }
    }
    return any;
  }
  // (flame_nltgv2_graph_size, not _get_info: the latter also counts the vertex-per-lane rows, once per topology, from a host image of the
  //  layout it first has to fetch -- 0.56 ms at 640x480, which the loop paid after every syncCommit with the solver standing still:
  //  profiles/r06_cpp_frame_loop.txt)
  size_t dev_vertices() {
    int32_t V = 0, E = 0;
    return flame_nltgv2_graph_size(dev_.handle(), &V, &E) == 0 ? static_cast<size_t>(V) : 0;
  }

  Graph* graph_;
  GraphMutex* graph_mtx_;
  Params params_;
  const int iters_per_round_, max_rounds_;
  int open_max_ = 0;               // device mode: > 0 = one open run of at most this many iterations instead of rounds (useOpenRuns)
  uint64_t count_base_ = 0;        // ... iterations_ = count_base_ + the library's count
  bool open_na_ = false;           // ... an open run is not applicable to the graph as it stands (rounds until a callback has run)
  static constexpr int kRoundsPerCheck = 64;
  int rounds_unchecked_ = 0;
  DeviceGraph dev_;
  std::mutex dev_mtx_;            // serialises calls into dev_ (the context is not thread-safe)
  mutable std::mutex state_mtx_;  // dirty_generation_, error_, thread_ start/stop
  std::condition_variable cv_;
  std::thread thread_;
  std::atomic<bool> stop_{false};
  std::atomic<uint64_t> iterations_{0}, uploads_{0}, busy_ns_{0}, iterations_at_ready_{0}, open_runs_{0};
  bool device_ready_ = false;                              // device mode: the first image was handed over (state_mtx_)
  std::chrono::steady_clock::time_point t_ready_ = std::chrono::steady_clock::now();
  std::atomic<int> callers_waiting_{0};
  uint64_t dirty_generation_ = 1;  // the graph as handed to the constructor is "edit 1": uploaded by the first round
  bool uploaded_once_ = false;
  bool exited_ = false;            // the thread function has returned (state_mtx_)
  std::string error_;
};

}  // namespace flame_hip

#endif  // FLAME_HIP_SOLVER_LOOP_HPP_
