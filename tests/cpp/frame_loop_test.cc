// tests/cpp/frame_loop_test.cc -- the order of Flame::update() (/root/reference/src/flame/flame.cc:265-415) driven from C++ through
// the facade, with the solver FREE-RUNNING beside it (flame.cc:99-112: the solver thread) -- flame_hip::SolverLoop in device mode:
//
//   per frame:  FeatureTracker::addFrame + updateFeatureIDepths      (flame.cc:150, 1280-1536)
//               delaunayTriangulate                                  (utils/delaunay.cc:31-77)
//               projectGraph                                         (flame.cc:1862-1938; C-ABI call on the loop's DeviceGraph)
//               DeviceGraph::syncPrepare   -- the solver thread goes on iterating on the live graph --
//               [the host's other work of a frame]
//               DeviceGraph::syncCommit                              (flame.cc:1985-2121)
//               DeviceGraph::interpolateMeshBegin / interpolateMeshEnd  (flame.cc:409-415)
//
// The program is a DRIVER: tests/test_cpp_facade.py::test_frame_loop_end_to_end writes the frames' inputs to a file (images, poses,
// the features, and per frame the feature set that enters the graph -- produced by the chained CPU checkers), runs this program, and
// replays its log on the checkers: the log holds, for every call that touches the device image, the number of solver iterations the
// free-running loop had applied when the call took the device (read under the loop's lock: exact), so the CPU side can run exactly
// those iterations between the same edits and compare every output bit for bit: the updated features, the keep mask of projectGraph,
// the dense map and its coverage, the graph's state at the end of every frame.  It also prints what share of the time the device spent
// on solver iterations (SolverLoop::utilization: iterations per second in the loop over the rate of an undisturbed run).  With `lean` the window of the test is closed -- no read-back of the graph's state,
// the dense map is not copied while the device is held -- and the printed share is that of the loop as FLaME would drive it.
//
// usage: frame_loop_test IN OUT [iters_per_round [lean]]     exit code 0 = ran, 77 = no usable HIP device
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "flame_hip/feature_tracker.hpp"
#include "flame_hip/solver_loop.hpp"

namespace dgraph = flame::optimizers::nltgv2_l1_graph_regularizer::hip;

namespace {

// (the whole input is read before the loop starts and the log is written after it ends: the loop's time holds no file I/O)
struct Reader {
  std::vector<char> buf;
  size_t at = 0;
  bool load(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    const bool ok = buf.empty() || std::fread(buf.data(), 1, buf.size(), f) == buf.size();
    std::fclose(f);
    return ok;
  }
  template <class T>
  T one() {
    T v;
    if (at + sizeof(T) > buf.size()) std::abort();
    std::memcpy(&v, buf.data() + at, sizeof(T));
    at += sizeof(T);
    return v;
  }
  template <class T>
  std::vector<T> many(size_t n) {
    std::vector<T> v(n);
    if (at + n * sizeof(T) > buf.size()) std::abort();
    if (n) std::memcpy(v.data(), buf.data() + at, n * sizeof(T));
    at += n * sizeof(T);
    return v;
  }
};
struct Writer {
  std::vector<char> buf;
  template <class T>
  void one(const T& v) {
    const char* p = reinterpret_cast<const char*>(&v);
    buf.insert(buf.end(), p, p + sizeof(T));
  }
  template <class T>
  void many(const std::vector<T>& v) {
    const char* p = reinterpret_cast<const char*>(v.data());
    buf.insert(buf.end(), p, p + v.size() * sizeof(T));
  }
  bool save(const char* path) const {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    const bool ok = buf.empty() || std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    return std::fclose(f) == 0 && ok;
  }
};

struct Mat3 {
  float m[9];
  float operator()(int r, int c) const { return m[3 * r + c]; }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  {
    flame_nltgv2_ctx* probe = nullptr;
    const int rc = flame_nltgv2_create(&probe, 0);
    if (rc != 0) {
      std::printf("%s\n", flame_nltgv2_status_string(rc));
      return 77;
    }
    flame_nltgv2_destroy(probe);
  }
  Reader in;
  Writer out;
  if (!in.load(argv[1])) return 2;
  out.buf.assign(in.buf.size() * 2 + (64u << 20), 0);  // (touched now: the loop appends into mapped pages)
  out.buf.clear();
  const int iters_per_round = argc > 3 ? std::atoi(argv[3]) : 200;
  const bool lean = argc > 4 && std::atoi(argv[4]) != 0;
  const bool rounds = std::getenv("FRAME_LOOP_ROUNDS") != nullptr;
  const bool one_part = std::getenv("FRAME_LOOP_ONE_PART") != nullptr;  // syncCommit + interpolateMeshBegin with the solver stopped for both
  const int32_t W = in.one<int32_t>(), H = in.one<int32_t>(), pad = in.one<int32_t>();
  const int32_t n_feats = in.one<int32_t>(), n_initial = in.one<int32_t>(), n_new = in.one<int32_t>(), host_work_us = in.one<int32_t>();
  Mat3 K, Kinv;
  std::memcpy(K.m, in.many<float>(9).data(), sizeof K.m);
  std::memcpy(Kinv.m, in.many<float>(9).data(), sizeof Kinv.m);
  try {
    flame_hip::FeatureTracker tracker(K, Kinv, W, H, pad);
    for (int i = 0; i < n_initial; ++i) {
      const uint32_t id = in.one<uint32_t>();
      const std::vector<uint8_t> img = in.many<uint8_t>((size_t)W * H);
      tracker.addFrame(id, img.data(), W);
    }
    std::vector<flame_stereo_feature> feats = in.many<flame_stereo_feature>((size_t)n_feats);
    flame_stereo_params sp;
    flame_stereo_default_params(&sp);
    const dgraph::Params params;
    std::mutex graph_mtx;
    flame_hip::SolverLoop<flame_hip::FlatGraph> loop(nullptr, &graph_mtx, params, iters_per_round);
    // Round 6, third step: ONE open run instead of rounds (DeviceGraph::runOpen: it goes on until a call needs the solver settled) --
    // FRAME_LOOP_ROUNDS=1: rounds of iters_per_round, two in flight, for comparison.  Either way the iteration counts the log holds are
    // the library's (DeviceGraph::iterations() right after the call that settled the solver).
    if (!rounds) loop.useOpenRuns();
    loop.start();
    bool first = true;
    uint64_t it_base = 0;  // the library's iteration count when the first graph stood
    int meshes_begun = 0, meshes_beside_rounds = 0;  // two-part holds: the meshes begun, those with rounds still in flight behind the call
    // per-stage wall time of the frame thread (printed as medians: what a frame's time is made of)
    static const char* const kStage[] = {"addFrame", "updateFeatureIDepths", "delaunayTriangulate", "projectGraph", "syncPrepare", "other host work",
                                         "syncCommit", "interpolateMeshBegin", "other host work (2)", "interpolateMeshEnd"};
    std::vector<std::vector<double> > stage_ms(10);
    auto t_stage = std::chrono::steady_clock::now();
    auto lap = [&](int k) {
      const auto now = std::chrono::steady_clock::now();
      stage_ms[(size_t)k].push_back(std::chrono::duration<double, std::milli>(now - t_stage).count());
      t_stage = now;
    };
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<double> frame_ms;
    auto t_steady = t_begin;        // steady state: from the third frame on (the first two hold the first upload, the first launches of every
    uint64_t it_steady = 0;         // kernel, the first sync's buffer growth)
    for (int fr = 0; fr < n_new; ++fr) {
      const auto t_frame = std::chrono::steady_clock::now();
      if (fr == 2) {  // the steady-state window opens: an exact count (an open run in flight is only counted once it is settled)
        loop.withDevice([&](dgraph::DeviceGraph& d, uint64_t) { d.sync(); });
        t_steady = std::chrono::steady_clock::now(), it_steady = loop.iterations();
      }
      // ---- Frame::create + updateFeatureIDepths --------------------------------------------------------------------------
      const uint32_t id = in.one<uint32_t>(), curr_pf = in.one<uint32_t>();
      const std::vector<uint8_t> img = in.many<uint8_t>((size_t)W * H);
      const int32_t n_poses = in.one<int32_t>();
      const std::vector<flame_stereo_pose> poses = in.many<flame_stereo_pose>((size_t)n_poses);
      t_stage = std::chrono::steady_clock::now();
      tracker.addFrame(id, img.data(), W);
      lap(0);
      flame_stereo_stats st;
      tracker.updateFeatureIDepths(sp, id, curr_pf, poses, feats.data(), n_feats, &st);
      lap(1);
      if (!lean) out.many(feats), out.one(st);  // (lean: the log is not replayed; the loop's time holds the pipeline's calls only)
      // ---- the features that enter the graph (projectFeatures is scaffolding of the test, done by its Python side) ----------
      const int32_t V = in.one<int32_t>();
      const std::vector<int32_t> fid = in.many<int32_t>((size_t)V);
      const std::vector<float> pos = in.many<float>((size_t)2 * V), idepth = in.many<float>((size_t)V);
      const std::vector<float> weight((size_t)V, 1.0f);
      std::vector<int32_t> tris, edges;
      t_stage = std::chrono::steady_clock::now();
      flame_hip::delaunayTriangulate(pos, &tris, &edges);
      lap(2);
      if (!lean) {
        out.one<int32_t>((int32_t)(tris.size() / 3)), out.many(tris);
        out.one<int32_t>((int32_t)(edges.size() / 2)), out.many(edges);
      }
      uint64_t it_commit = 0, it_raster = 0, it_state = 0;
      const int32_t has_projection = in.one<int32_t>();
      flame_nltgv2_projection pr;
      if (has_projection) pr = in.one<flame_nltgv2_projection>();
      if (first) {
        flame_hip::FlatGraph g;
        g.vertices.resize((size_t)V);
        for (int32_t v = 0; v < V; ++v) {
          flame_hip::VertexData& d = g.vertices[(size_t)v];
          d.pos_x = pos[2 * v], d.pos_y = pos[2 * v + 1];
          d.data_term = d.x = d.x_bar = d.x_prev = idepth[(size_t)v];
        }
        for (size_t e = 0; e + 1 < edges.size(); e += 2) {
          flame_hip::EdgeData d;
          d.source = edges[e], d.target = edges[e + 1];
          const float dx = pos[2 * d.source] - pos[2 * d.target], dy = pos[2 * d.source + 1] - pos[2 * d.target + 1];
          d.alpha = 1.0f / std::sqrt(dx * dx + dy * dy);  // flame.cc:2087-2102
          g.edges.push_back(d);
        }
        loop.withDevice([&](dgraph::DeviceGraph& d, uint64_t) {
          d.upload(g);
          if (flame_nltgv2_set_feature_ids(d.handle(), fid.data()) != 0) throw flame_hip::Error(FLAME_NLTGV2_ERR_INVALID_ARG, "set_feature_ids");
          it_base = d.iterations();
        });
        loop.deviceReady();
        out.one<uint64_t>(0);
        first = false;
      } else {
        // ---- projectGraph, then the sync in two halves with the solver iterating in between ------------------------------
        std::vector<uint8_t> keep;
        uint64_t it_project = 0;
        t_stage = std::chrono::steady_clock::now();
        loop.withDevice([&](dgraph::DeviceGraph& d, uint64_t) {
          int32_t Vo = 0, Eo = 0;
          if (flame_nltgv2_graph_size(d.handle(), &Vo, &Eo) != 0) throw flame_hip::Error(FLAME_NLTGV2_ERR_HIP, "graph_size");
          keep.resize((size_t)Vo);
          const int rc = flame_nltgv2_project_graph(d.handle(), &pr, 1.0f, keep.data(), nullptr);
          if (rc != 0) throw flame_hip::Error(rc, "project_graph");
          it_project = d.iterations() - it_base;  // (the call has settled the solver: the state it projected had seen exactly this many)
        });
        lap(3);
        if (!lean) out.one(it_project), out.one<int32_t>((int32_t)keep.size()), out.many(keep);
        t_stage = std::chrono::steady_clock::now();
        loop.withDevice([&](dgraph::DeviceGraph& d, uint64_t) {
          d.syncPrepare(fid, pos, idepth, weight, edges, false, nullptr, 0.0f, /*edges_unique=*/true);
        });
        lap(4);
        std::this_thread::sleep_for(std::chrono::microseconds(host_work_us));  // (the host's other work of a frame: the solver iterates)
        lap(5);
        // syncCommit and the start of interpolateMesh follow each other in Flame::update() (flame.cc:307-319, 409-415): ONE hold of the
        // device for both -- the solver is settled once, the rasteriser reads the canonical arrays the commit has just written (no
        // unpack), and no freshly enqueued round has to be waited for in between (round 6: two holds cost 0.4 ms of waiting and a
        // second 0.1 ms gap per frame, profiles/r06_cpp_frame_loop.txt)
        // Round 6, second step: the mesh does not need the solver stopped either -- with FLAME_NLTGV2_OPT_MESH_STATE = 1 it is of the
        // state the commit left, read from the canonical arrays beside the next rounds, which withDevice(f, g) enqueues BEFORE the
        // mesh's host work (FRAME_LOOP_ONE_PART=1: the one-part hold, for comparison)
        if (one_part) {
          loop.withDevice([&](dgraph::DeviceGraph& d, uint64_t) {
            d.syncCommit();
            it_commit = it_raster = d.iterations() - it_base;
            lap(6);
            d.interpolateMeshBegin(tris, H, W, 1.0f);
          });
        } else {
          loop.withDevice(
              [&](dgraph::DeviceGraph& d, uint64_t) {
                d.syncCommit();
                it_commit = it_raster = d.iterations() - it_base;
                lap(6);
              },
              [&](dgraph::DeviceGraph& d, uint64_t) {
                d.interpolateMeshBegin(tris, H, W, 1.0f);
                ++meshes_begun;
                if (d.runsInFlight() > 0) ++meshes_beside_rounds;  // (it did not settle the rounds withDevice has just enqueued)
              });
        }
        lap(7);
        if (!lean) out.one(it_commit);
      }
      // ---- interpolateMesh in two halves ----------------------------------------------------------------------------------------
      if (it_raster == 0 && it_commit == 0) {  // (the first frame: no sync, the map of the uploaded graph)
        t_stage = std::chrono::steady_clock::now();
        loop.withDevice([&](dgraph::DeviceGraph& d, uint64_t) {
          d.interpolateMeshBegin(tris, H, W, 1.0f);  // (this one settles the rounds in flight: the option is set behind it)
          it_raster = d.iterations() - it_base;
          if (!one_part && flame_nltgv2_set_option(d.handle(), FLAME_NLTGV2_OPT_MESH_STATE, 1) != 0)
            throw flame_hip::Error(FLAME_NLTGV2_ERR_INVALID_ARG, "set_option");
        });
        lap(7);
      }
      std::this_thread::sleep_for(std::chrono::microseconds(host_work_us / 2));
      lap(8);
      const float* map = nullptr;
      int32_t coverage = 0;
      std::vector<float> dense(lean ? 0 : (size_t)W * H);
      loop.withDevice([&](dgraph::DeviceGraph& d, uint64_t) {
        coverage = d.interpolateMeshEnd(&map);  // (the pinned map stays valid until the next interpolateMeshBegin: FLaME reads it in place)
        if (!lean) std::memcpy(dense.data(), map, sizeof(float) * dense.size());
      });
      lap(9);
      if (lean) frame_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_frame).count());
      if (!lean) out.one(it_raster), out.one(coverage), out.many(dense);
      // ---- the graph's state at the end of the frame (a read-back the reference does not need: the test's window on the solver) ------
      flame_hip::FlatArrays a;
      if (lean) continue;
      loop.withDevice([&](dgraph::DeviceGraph& d, uint64_t) {
        int32_t Vn = 0, En = 0;
        if (flame_nltgv2_graph_size(d.handle(), &Vn, &En) != 0) throw flame_hip::Error(FLAME_NLTGV2_ERR_HIP, "graph_size");
        a.resize((size_t)Vn, (size_t)En);
        flame_nltgv2_graph v = a.view();
        const int rc = flame_nltgv2_download_state(d.handle(), &v);
        if (rc != 0) throw flame_hip::Error(rc, "download_state");
        it_state = d.iterations() - it_base;
      });
      out.one(it_state), out.one<int32_t>((int32_t)a.x.size()), out.one<int32_t>((int32_t)a.q1.size());
      out.many(a.x), out.many(a.w1), out.many(a.w2), out.many(a.x_bar), out.many(a.w1_bar), out.many(a.w2_bar);
      out.many(a.q1), out.many(a.q2), out.many(a.q3);
    }
    loop.withDevice([&](dgraph::DeviceGraph& d, uint64_t) { d.sync(); });  // (an open run still in flight is counted once it is settled)
    const auto t_end = std::chrono::steady_clock::now();
    const double wall_ms = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
    const uint64_t total = loop.iterations();
    const double steady_rate = n_new > 2 ? (double)(total - it_steady) / std::chrono::duration<double>(t_end - t_steady).count() : 0.0;
    // the solver's own rate on the last frame's graph: an undisturbed run through the same context, after the loop has stopped
    double free_rate = 0.0;
    const double util_at_stop_per_rate = loop.utilization(1.0);  // iterations per second achieved in the loop
    loop.stop();
    loop.withDevice([&](dgraph::DeviceGraph& d, uint64_t) {
      d.run(params, 400);
      const auto t0 = std::chrono::steady_clock::now();
      d.run(params, 4000);
      free_rate = 4000.0 / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    });
    const double busy = free_rate > 0 ? util_at_stop_per_rate / free_rate : 0.0;
    const std::string err = loop.error();
    if (!err.empty()) {
      std::printf("FAIL: the solver thread stopped: %s\n", err.c_str());
      return 1;
    }
    const std::string how = rounds ? std::to_string(iters_per_round) + " per round, two rounds in flight" : std::to_string((unsigned long long)loop.openRuns()) + " open runs, each until a call needs the solver settled" + (loop.openRuns() < (uint64_t)n_new ? "; rounds of " + std::to_string(iters_per_round) + " where an open run does not apply" : std::string());
    std::printf("frame loop%s: %d frames in %.2f ms (%.2f ms per frame), %llu solver iterations beside them (%s), %.0f iterations/s in the loop against %.0f undisturbed: "
                "solver busy %.1f %% of the time since the first graph, idle %.1f %%\n",
                lean ? " (lean: no state read-back)" : "", n_new, wall_ms, wall_ms / n_new, (unsigned long long)total, how.c_str(), util_at_stop_per_rate, free_rate, 100.0 * busy, 100.0 * (1.0 - busy));
    if (!one_part) std::printf("  meshes begun beside the next rounds: %d of %d\n", meshes_beside_rounds, meshes_begun);
    if (lean) {
      std::sort(frame_ms.begin() + std::min<size_t>(2, frame_ms.size()), frame_ms.end());
      const double med = frame_ms.size() > 2 ? frame_ms[2 + (frame_ms.size() - 2) / 2] : 0.0;
      std::printf("  steady state (from the third frame on): frame %.2f ms (median), %.0f iterations/s in the loop = solver busy %.1f %%\n", med, steady_rate,
                  free_rate > 0 ? 100.0 * steady_rate / free_rate : 0.0);
      std::printf("  the frame thread's stages, median ms:");
      double sum = 0.0;
      for (size_t k = 0; k < stage_ms.size(); ++k) {
        std::vector<double>& v = stage_ms[k];
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        std::printf(" %s %.3f;", kStage[k], v[v.size() / 2]);
        sum += v[v.size() / 2];
      }
      std::printf(" sum %.3f\n", sum);
      std::printf("  ... their longest, ms:");
      for (size_t k = 0; k < stage_ms.size(); ++k)
        if (!stage_ms[k].empty()) std::printf(" %s %.3f;", kStage[k], stage_ms[k].back());
      std::printf("\n");
    }
  } catch (const std::exception& e) {
    std::printf("FAIL: %s\n", e.what());
    return 1;
  }
  return out.save(argv[2]) ? 0 : 2;
}
