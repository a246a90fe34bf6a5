// tests/cpp/frame_gather_test.cc -- flame_hip::FrameGather (include/flame_hip/frame_gather.hpp): BASELINE configuration 4
// from a C++ host.  One DeviceGraph per visible GPU (one on the test box, eight on the node), a different frame on each,
// solver streams tied to the gather's streams, x * graph_scale exported on the device into the send rows, ONE grouped
// ncclAllGather; every device's gathered block is compared, row by row, with the CPU checker's x * graph_scale.
// Build+run: tests/test_cpp_facade.py.  Exit code 0 = pass, 77 = no usable HIP device.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "flame_hip/frame_gather.hpp"
#include "../../flame_amd/csrc/flame_nltgv2_test_options.h"  // (test hook: FLAME_NLTGV2_OPT_FAULT_INJECT)

namespace dgraph = flame::optimizers::nltgv2_l1_graph_regularizer::hip;

extern "C" {
struct nltgv2_params { float data_factor, step_x, step_q, theta, x_min, x_max; };
int nltgv2_oracle_run(const nltgv2_params*, flame_nltgv2_graph*, int);
int hipGetDeviceCount(int*);
}

static unsigned long long sm(unsigned long long& s) {
  unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static float u01(unsigned long long& s) { return (float)(sm(s) >> 40) * (1.0f / 16777216.0f); }

static flame_hip::FlatGraph make_graph(int nx, int ny, unsigned long long seed) {
  flame_hip::FlatGraph g;
  g.vertices.resize((size_t)nx * ny);
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      flame_hip::VertexData& v = g.vertices[(size_t)y * nx + x];
      v.pos_x = 6.0f * x + 5.0f * u01(seed), v.pos_y = 6.0f * y + 5.0f * u01(seed);
      v.data_term = (x < nx / 2 ? 0.6f + 0.01f * x : 1.4f - 0.005f * y) + 0.05f * (u01(seed) - 0.5f);
      v.x = v.x_bar = v.x_prev = v.data_term;
    }
  auto add_edge = [&](int a, int b) {
    flame_hip::EdgeData e;
    if (sm(seed) & 1) { e.source = a, e.target = b; } else { e.source = b, e.target = a; }
    const float dx = g.vertices[a].pos_x - g.vertices[b].pos_x, dy = g.vertices[a].pos_y - g.vertices[b].pos_y;
    e.alpha = 1.0f / std::sqrt(dx * dx + dy * dy);
    g.edges.push_back(e);
  };
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      const int v = y * nx + x;
      if (x + 1 < nx) add_edge(v, v + 1);
      if (y + 1 < ny) add_edge(v, v + nx);
      if (x + 1 < nx && y + 1 < ny) add_edge(v, v + nx + 1);
    }
  return g;
}

int main() {
  int n_dev = 0;
  {
    flame_nltgv2_ctx* probe = nullptr;
    const int rc = flame_nltgv2_create(&probe, 0);
    if (rc != 0) {
      std::printf("%s\n", flame_nltgv2_status_string(rc));
      return 77;
    }
    flame_nltgv2_destroy(probe);
    if (hipGetDeviceCount(&n_dev) != 0 || n_dev <= 0) return 77;
  }
  int fails = 0;
  const dgraph::Params params;
  const nltgv2_params cp = {params.data_factor, params.step_x, params.step_q, params.theta, params.x_min, params.x_max};
  const float graph_scale = 1.7f;
  const int n_iters = 60;
  std::vector<int> devices;
  for (int k = 0; k < n_dev; ++k) devices.push_back(k);
  // frames of different sizes: the rows are ragged, padded to vmax
  std::vector<flame_hip::FlatGraph> frames;
  int32_t vmax = 0;
  for (int k = 0; k < n_dev; ++k) {
    frames.push_back(make_graph(24 + 3 * (k % 3), 18 + 2 * (k % 4), 100 + k));
    vmax = std::max<int32_t>(vmax, (int32_t)frames.back().vertices.size());
  }
  try {
    flame_hip::FrameGather gather(devices, vmax);
    std::vector<std::unique_ptr<dgraph::DeviceGraph>> solver;
    for (int k = 0; k < n_dev; ++k) {
      solver.emplace_back(new dgraph::DeviceGraph(devices[k]));
      solver[k]->setStream(gather.stream(k));
      solver[k]->upload(frames[k]);
      solver[k]->setExportTarget(gather.localRow(k), graph_scale);
    }
    int regathers = 0;
    for (int step = 0; step < 3; ++step) {  // three steps of the frame loop: solve everywhere, gather once
      // step 2: solver 0's persistent run is made to time out (one patch withholds its first record): it leaves before its
      // epilogue exports the row, the gather enqueued right behind it carries step 1's row, sync() takes the run back and
      // redoes it -- settle() has to notice and gather again
      if (step == 2) flame_nltgv2_set_option(solver[0]->handle(), FLAME_NLTGV2_OPT_FAULT_INJECT, 3000);
      for (int k = 0; k < n_dev; ++k) solver[k]->runAsync(params, n_iters);
      gather.gather(solver);
      regathers += gather.settle(solver);
      if (step == 2) flame_nltgv2_set_option(solver[0]->handle(), FLAME_NLTGV2_OPT_FAULT_INJECT, 0);
    }
    {
      flame_nltgv2_info info;
      flame_nltgv2_get_info(solver[0]->handle(), &info);
      const bool ok = info.timeouts_recovered == regathers && regathers <= 1;
      std::printf("run taken back %d time(s), %d re-gather(s)                  %s\n", info.timeouts_recovered, regathers, ok ? "ok" : "FAIL");
      fails += !ok;
    }
    // the checker: 3 * n_iters iterations per frame, x * graph_scale
    std::vector<std::vector<float>> want;
    for (int k = 0; k < n_dev; ++k) {
      flame_hip::FlatArrays a;
      flame_hip::GraphAccess<flame_hip::FlatGraph>::pack(frames[k], &a);
      flame_nltgv2_graph v = a.view();
      nltgv2_oracle_run(&cp, &v, 3 * n_iters);
      std::vector<float> row(a.x.size());
      for (size_t i = 0; i < row.size(); ++i) row[i] = a.x[i] * graph_scale;
      want.push_back(row);
    }
    for (int k = 0; k < n_dev; ++k) {
      std::vector<float> block;
      gather.download(k, &block);
      int bad = 0;
      for (int j = 0; j < n_dev; ++j) {
        bad += std::memcmp(&block[(size_t)j * vmax], want[j].data(), want[j].size() * sizeof(float)) != 0;
        for (size_t i = want[j].size(); i < (size_t)vmax; ++i) bad += block[(size_t)j * vmax + i] != 0.0f;  // padding stays zero
      }
      std::printf("device %d: gathered block of %d frame(s) == checker   %s\n", devices[k], n_dev, bad ? "FAIL" : "ok");
      fails += bad != 0;
    }
    std::printf("RCCL all-gather over %d device(s), vmax %d        %s\n", gather.size(), (int)vmax, fails ? "FAIL" : "ok");
  } catch (const flame_hip::Error& e) {
    std::printf("FAIL: %s (status %d)\n", e.what(), e.status);
    return 1;
  }
  return fails ? 1 : 0;
}
