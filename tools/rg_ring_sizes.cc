// tools/rg_ring_sizes.cc -- what a ghost ring of depth k costs a region of layout (R) (flame_amd/csrc/nltgv2_regions.hpp): per graph,
// region count and k: lanes of the workgroup, vertices owned / computed / held, edge lanes, fetch duties and LDS bytes per region.
// Host code only (the builder the library runs).  Driven by tools/rg_ring_sizes.py, which writes the graphs.
//   rg_ring_sizes GRAPH.bin [regions,regions,..]      GRAPH.bin = {int32 V, E; float pos[2V]; int32 src[E]; int32 dst[E]}
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nltgv2_regions.hpp"
using namespace flame_hip;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t VE[2];
  if (std::fread(VE, 4, 2, f) != 2) return 1;
  std::vector<float> pos((size_t)2 * VE[0]);
  std::vector<int32_t> src((size_t)VE[1]), dst((size_t)VE[1]);
  if (std::fread(pos.data(), 4, pos.size(), f) != pos.size() || std::fread(src.data(), 4, src.size(), f) != src.size() ||
      std::fread(dst.data(), 4, dst.size(), f) != dst.size())
    return 1;
  std::vector<int> counts;
  for (char* s = std::strtok(argc > 2 ? argv[2] : (char*)"256", ","); s; s = std::strtok(nullptr, ",")) counts.push_back(std::atoi(s));
  flame_nltgv2_graph g{};
  g.V = VE[0], g.E = VE[1], g.pos = pos.data(), g.src = src.data(), g.dst = dst.data();
  PackedLayout L;
  if (build_layout(&g, &L, false, true) != 0) return 1;
  std::printf("%s: V %d E %d, largest degree %d\n", argv[1], g.V, g.E, L.max_degree);
  for (int nr : counts)
    for (int k = 1; k <= 5; ++k) {
      RegionLayout R;
      const auto t0 = std::chrono::steady_clock::now();
      if (build_regions(&g, L, nr, k, &R) != 0) return 1;
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      int max_e = 0, max_vc = 0, max_thr = 0, min_o = 1 << 30, max_o = 0;
      for (int r = 0; r < R.n_regions; ++r) {
        const int32_t* inf = &R.info[(size_t)r * kRgInfoWords];
        max_thr = std::max(max_thr, inf[1]), max_vc = std::max(max_vc, inf[2]), max_e = std::max(max_e, inf[4]);
      }
      std::vector<int> own((size_t)R.n_regions, 0);
      for (int v = 0; v < g.V; ++v) own[(size_t)R.region_of[v]]++;
      for (int o : own) min_o = std::min(min_o, o), max_o = std::max(max_o, o);
      const double n = R.n_regions;
      std::printf("  regions %3d k %d: %s lanes %4d (largest region %4d) | per region: owned %.1f (%d..%d) computed vertices %.1f (max %d) = %.2f x owned, held %.1f, "
                  "edge lanes %.1f (max %d) = %.2f x E/regions, fetch duties %.1f | LDS %zu B | host build %.2f ms\n",
                  R.n_regions, k, R.ok ? "fits:" : "DOES NOT FIT:", R.block_threads, max_thr, R.sum_owned / n, min_o, max_o, R.sum_computed / n, max_vc,
                  (double)R.sum_computed / R.sum_owned, R.sum_local / n, R.sum_edges / n, max_e, (double)R.sum_edges / g.E, R.sum_fetch / n, rg_lds_bytes(R), ms);
    }
  return 0;
}
