// tests/cpp/wg_layout_test.cc -- host-only check of layout (E) of flame_amd/csrc/nltgv2_pack.hpp (the
// patch-per-wave rows k_persistent_pv runs on).  The kernel's data movement is replayed on the CPU, lane
// by lane, with exactly the visibility rules of the device code:
//   * a lane sees a neighbour's record either in its patch's local record area or in the slot its
//     patch's fetch lane copied from the global exchange buffer -- nothing else;
//   * only vertices flagged "publishes" write the global exchange buffer;
//   * every lane of a vertex accumulates the contributions of the vertex's lanes first..first+deg-1 in order.
// After n steps every state array must be bit-identical to the CPU checker (oracle/liboracle_nltgv2.so, linked
// as test infrastructure).  A wrong neighbour index, a missing publish flag or fetch entry, a vertex straddling
// two waves -- all show up as a mismatch here, before any GPU time is spent.
// Build+run: tests/test_pack.py::test_wg_layout_replay.  Compile with -ffp-contract=off.  Exit code 0 = pass.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "nltgv2_pack.hpp"

extern "C" {
struct nltgv2_params { float data_factor, step_x, step_q, theta, x_min, x_max; };
int nltgv2_oracle_run(const nltgv2_params*, flame_nltgv2_graph*, int);
}

using namespace flame_hip;

static unsigned long long sm(unsigned long long& s) {
  unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static float u01(unsigned long long& s) { return (float)(sm(s) >> 40) * (1.0f / 16777216.0f); }

struct HostGraph {
  std::vector<float> pos, x, w1, w2, xb, w1b, w2b, xp, w1p, w2p, data, weight, alpha, beta, q1, q2, q3;
  std::vector<int32_t> src, dst;
  flame_nltgv2_graph view() {
    flame_nltgv2_graph g;
    std::memset(&g, 0, sizeof g);
    g.V = (int32_t)x.size(), g.E = (int32_t)src.size();
    g.pos = pos.data(), g.x = x.data(), g.w1 = w1.data(), g.w2 = w2.data();
    g.x_bar = xb.data(), g.w1_bar = w1b.data(), g.w2_bar = w2b.data();
    g.x_prev = xp.data(), g.w1_prev = w1p.data(), g.w2_prev = w2p.data();
    g.data_term = data.data(), g.data_weight = weight.data();
    g.src = src.data(), g.dst = dst.data(), g.alpha = alpha.data(), g.beta = beta.data();
    g.q1 = q1.data(), g.q2 = q2.data(), g.q3 = q3.data();
    return g;
  }
};

// n_frames disjoint jittered grids with random diagonals, random edge orientation, a few isolated vertices and one
// high-degree hub per frame; state mid-solve-like (random w, q) so that every term of the update matters.
static HostGraph make_graph(int nx, int ny, int n_frames, unsigned long long seed, bool with_hub = true) {
  HostGraph g;
  for (int f = 0; f < n_frames; ++f) {
    const int base = (int)g.x.size();
    for (int y = 0; y < ny; ++y)
      for (int x = 0; x < nx; ++x) {
        g.pos.push_back(6.0f * x + 5.0f * u01(seed));
        g.pos.push_back(6.0f * y + 5.0f * u01(seed));
        const float d = (x < nx / 2 ? 0.6f + 0.01f * x : 1.4f - 0.005f * y) + 0.2f * (u01(seed) - 0.5f);
        g.data.push_back(d);
        g.weight.push_back(0.5f + u01(seed));
        g.x.push_back(d + 0.05f * (u01(seed) - 0.5f));
        g.w1.push_back(0.02f * (u01(seed) - 0.5f));
        g.w2.push_back(0.02f * (u01(seed) - 0.5f));
        g.xb.push_back(g.x.back() + 0.01f * (u01(seed) - 0.5f));
        g.w1b.push_back(g.w1.back()), g.w2b.push_back(g.w2.back());
        g.xp.push_back(g.x.back()), g.w1p.push_back(g.w1.back()), g.w2p.push_back(g.w2.back());
      }
    auto add = [&](int a, int b) {
      a += base, b += base;
      if (sm(seed) & 1) std::swap(a, b);
      g.src.push_back(a), g.dst.push_back(b);
      const float dx = g.pos[2 * a] - g.pos[2 * b], dy = g.pos[2 * a + 1] - g.pos[2 * b + 1];
      g.alpha.push_back(1.0f / std::sqrt(dx * dx + dy * dy));
      g.beta.push_back(0.5f + u01(seed));
      g.q1.push_back(1.6f * (u01(seed) - 0.5f)), g.q2.push_back(1.6f * (u01(seed) - 0.5f)), g.q3.push_back(1.6f * (u01(seed) - 0.5f));
    };
    for (int y = 0; y < ny; ++y)
      for (int x = 0; x < nx; ++x) {
        const int v = y * nx + x;
        if ((x % 17 == 5 && y % 13 == 3)) continue;  // leaves some vertices with few edges
        if (x + 1 < nx) add(v, v + 1);
        if (y + 1 < ny) add(v, v + nx);
        if (x + 1 < nx && y + 1 < ny) {
          if (sm(seed) & 1) add(v, v + nx + 1); else add(v + 1, v + nx);
        }
      }
    // a hub: vertex (nx/2, ny/2) also connected to 40 far vertices (degree ~46 <= 64)
    const int hub = (ny / 2) * nx + nx / 2;
    for (int k = 0; k < (with_hub ? 40 : 9) && k * 7 + 3 < nx * ny; ++k)  // (without the hub: degree ~15, still row-packable)
      if (std::abs(k * 7 + 3 - hub) > nx + 1) add(hub, k * 7 + 3);
    // two isolated vertices
    for (int k = 0; k < 2; ++k) {
      g.pos.push_back(3.0f + k), g.pos.push_back(1000.0f);
      g.data.push_back(1.0f), g.weight.push_back(1.0f), g.x.push_back(1.3f), g.w1.push_back(0.01f), g.w2.push_back(-0.01f);
      g.xb.push_back(1.31f), g.w1b.push_back(0.01f), g.w2b.push_back(-0.01f);
      g.xp.push_back(1.3f), g.w1p.push_back(0.01f), g.w2p.push_back(-0.01f);
    }
  }
  // shuffle the edge order a little: the accumulation order is ascending edge id, whatever that order is
  for (size_t k = 0; k + 1 < g.src.size(); k += 3) {
    const size_t j = k + (sm(seed) % 2);
    std::swap(g.src[k], g.src[j]), std::swap(g.dst[k], g.dst[j]), std::swap(g.alpha[k], g.alpha[j]);
    std::swap(g.beta[k], g.beta[j]), std::swap(g.q1[k], g.q1[j]), std::swap(g.q2[k], g.q2[j]), std::swap(g.q3[k], g.q3[j]);
  }
  return g;
}

static float clampq(float q) { return std::fmin(std::fmax(q, -1.0f), 1.0f); }
static float prox_l1(float x_min, float x_max, float step_x, float w, float x, float data) {
  const float diff = x - data, thresh = step_x * w;
  float nx = diff > thresh ? x - thresh : (diff < -thresh ? x + thresh : data);
  nx = nx < x_min ? x_min : nx;
  return nx > x_max ? x_max : nx;
}

struct Rec { float xb, w1b, w2b; unsigned tag; };

static int replay(HostGraph& hg, int n_iters, const nltgv2_params& p, bool rowpack, bool expect_rowpack) {
  flame_nltgv2_graph g = hg.view();
  PackedLayout L;
  if (build_layout(&g, &L, true, rowpack) != 0 || !L.wg_ok) return 1;
  if (L.wg_rowpack != expect_rowpack) return 16;
  const int T = 64, V = g.V;
  // structural invariants
  std::vector<int> seen(L.n_slices * 64, 0);
  for (int wg = 0; wg < L.wg_count; ++wg) {
    for (int t = 0; t < T; ++t) {
      const size_t hl = (size_t)wg * T + t;
      const uint32_t m = L.wg_meta[hl];
      if (!(m & kWgValid)) continue;
      const int first = m & 63, deg = (m >> 6) & 127, need = deg > 1 ? deg : 1;
      if (first + need > 64) return 2;                        // a vertex inside one wave
      const bool rowp = L.wg_rowpack;  // this patch is row-packed
      if (rowp && need <= 16 && first / 16 != (first + need - 1) / 16) return 17;  // ... then a vertex lies inside one 16-lane row,
      if (rowp && need > 8 && (first % 16 != 0)) return 20;          // one of more than 8 edges at the start of a row of its own
      if (rowp && need > 16 && first != 0) return 21;                // and one of more than 16 at the start of its patch (whole rows)
      if (((m & kWgHead) != 0) != ((t & 63) == first)) return 18;
      const int k = (t & 63) - first;
      if (k < 0 || k >= need) return 3;
      if (((m & kWgTail) != 0) != (k == need - 1)) return 4;
      if (((m & kWgActive) != 0) != (k < deg)) return 5;
      if (m & kWgTail) seen[L.wg_vid[hl]]++;
      if ((int)((m >> 13) & 2047) >= (L.wg_info[4 * wg + 2] & 0xffff) || (L.wg_info[4 * wg + 2] & 0xffff) > L.wg_lcap) return 6;
    }
    if (L.wg_info[4 * wg + 1] > L.wg_rcap || L.wg_info[4 * wg + 1] > T) return 7;
    {  // slab stride: a multiple of 4, at least 8, covers the patch's largest degree, and fits the LDS sizing figure
      const int stride = L.wg_info[4 * wg + 3];
      if ((L.wg_info[4 * wg + 2] & 0xffff) == 0) continue;
      if (L.wg_rowpack) {  // row-packed: the patch's largest degree itself (the DPP shifts + 1; above 16: the rows of its big vertex)
        if (stride < 1 || stride > 64) return 13;
        bool reached = false;
        for (int t = 0; t < T; ++t)
          reached |= (L.wg_meta[(size_t)wg * T + t] & kWgValid) && (int)std::max(1u, (L.wg_meta[(size_t)wg * T + t] >> 6) & 127) == stride;
        if (!reached) return 19;
      } else if (stride < 8 || (stride & 3) || (stride + 1) * (L.wg_info[4 * wg + 2] & 0xffff) > L.wg_slab_slots) return 13;
      for (int t = 0; t < T; ++t)
        if ((L.wg_meta[(size_t)wg * T + t] & kWgValid) && (int)((L.wg_meta[(size_t)wg * T + t] >> 6) & 127) > stride) return 14;
    }
    for (int i = 1; i < L.wg_info[4 * wg + 1]; ++i)
      if (L.wg_fetch[(size_t)wg * T + i] <= L.wg_fetch[(size_t)wg * T + i - 1]) return 8;  // sorted, distinct
  }
  for (int v = 0; v < V; ++v)
    if (seen[L.iperm[v]] != 1) return 9;  // every vertex owned exactly once

  // device-like state: per lane registers, global exchange buffer (two parities), per-workgroup record areas
  const size_t NL = (size_t)L.wg_count * T;
  struct Lane { float x, w1, w2, xb, w1b, w2b, xp, w1p, w2p, q1, q2, q3; };
  std::vector<Lane> ln(NL);
  std::vector<Rec> glob[2] = {std::vector<Rec>(L.n_rec, Rec{0, 0, 0, 0}), std::vector<Rec>(L.n_rec, Rec{0, 0, 0, 0})};
  const int stride = L.wg_lcap + L.wg_rcap;
  std::vector<Rec> area[2] = {std::vector<Rec>((size_t)L.wg_count * stride), std::vector<Rec>((size_t)L.wg_count * stride)};
  auto edge_of = [&](int slot) { return L.rec_edge[slot]; };
  const unsigned tag0 = 3;
  for (size_t hl = 0; hl < NL; ++hl) {
    const uint32_t m = L.wg_meta[hl];
    if (!(m & kWgValid)) continue;
    const int o = L.perm[L.wg_vid[hl]];
    Lane& a = ln[hl];
    a.x = hg.x[o], a.w1 = hg.w1[o], a.w2 = hg.w2[o], a.xb = hg.xb[o], a.w1b = hg.w1b[o], a.w2b = hg.w2b[o];
    a.xp = a.x, a.w1p = a.w1, a.w2p = a.w2;
    if (m & kWgActive) {
      const int e = edge_of(L.wg_slot[hl]);
      a.q1 = hg.q1[e], a.q2 = hg.q2[e], a.q3 = hg.q3[e];
    }
    if (m & kWgTail) {
      const int wg = (int)(hl / T), loc = (m >> 13) & 2047;
      area[tag0 & 1][(size_t)wg * stride + loc] = Rec{a.xb, a.w1b, a.w2b, tag0};
      if (m & kWgPublish) glob[tag0 & 1][L.wg_info[4 * wg] + loc] = Rec{a.xb, a.w1b, a.w2b, tag0};
    }
  }
  std::vector<float> c4(NL * 4), c1(NL);
  for (int it = 0; it < n_iters; ++it) {
    const unsigned s = tag0 + it;
    const int par = s & 1;
    // fetch phase (all workgroups): only tagged records are accepted
    for (int wg = 0; wg < L.wg_count; ++wg)
      for (int i = 0; i < L.wg_info[4 * wg + 1]; ++i) {
        const Rec r = glob[par][L.wg_fetch[(size_t)wg * T + i]];
        if (r.tag != s) return 10;  // the producer did not publish: a real run would hang here
        area[par][(size_t)wg * stride + L.wg_lcap + i] = r;
      }
    // dual + contributions
    for (size_t hl = 0; hl < NL; ++hl) {
      const uint32_t m = L.wg_meta[hl];
      if (!(m & kWgActive)) continue;
      const int wg = (int)(hl / T), slot = L.wg_slot[hl], e = edge_of(slot);
      const bool is_target = (L.rec_nbr[slot] & kRoleBit) != 0;
      const int code = L.wg_nbr[hl];
      const int idx = code < 0 ? L.wg_lcap + (code & 0x7fffffff) : code;
      const Rec nb = area[par][(size_t)wg * stride + idx];
      if (nb.tag != s) return 11;
      Lane& a = ln[hl];
      const float alpha = hg.alpha[e], beta = hg.beta[e];
      const int si = hg.src[e], di = hg.dst[e];
      const float dx = hg.pos[2 * si] - hg.pos[2 * di], dy = hg.pos[2 * si + 1] - hg.pos[2 * di + 1];
      const float xbi = is_target ? nb.xb : a.xb, xbj = is_target ? a.xb : nb.xb;
      const float w1bi = is_target ? nb.w1b : a.w1b, w1bj = is_target ? a.w1b : nb.w1b;
      const float w2bi = is_target ? nb.w2b : a.w2b, w2bj = is_target ? a.w2b : nb.w2b;
      float K1 = alpha * (xbi - xbj);
      K1 -= alpha * dx * w1bi;
      K1 -= alpha * dy * w2bi;
      a.q1 = clampq(a.q1 + p.step_q * K1);
      a.q2 = clampq(a.q2 + p.step_q * (beta * (w1bi - w1bj)));
      a.q3 = clampq(a.q3 + p.step_q * (beta * (w2bi - w2bj)));
      const float t1 = a.q1 * p.step_x * alpha, t2 = a.q2 * p.step_x * beta, t3 = a.q3 * p.step_x * beta;
      c4[4 * hl + 0] = is_target ? t1 : -t1;
      c4[4 * hl + 1] = is_target ? t2 : t1 * dx;
      c4[4 * hl + 2] = is_target ? t3 : t1 * dy;
      c4[4 * hl + 3] = is_target ? -0.0f : -t2;
      c1[hl] = is_target ? -0.0f : -t3;
    }
    // accumulate + vertex update, by every lane of the vertex
    for (size_t hl = 0; hl < NL; ++hl) {
      const uint32_t m = L.wg_meta[hl];
      if (!(m & kWgValid)) continue;
      const int wg = (int)(hl / T), first = m & 63, deg = (m >> 6) & 127, loc = (m >> 13) & 2047;
      const size_t vb = hl - (hl & 63) + first;
      Lane& a = ln[hl];
      float X = a.x, W1 = a.w1, W2 = a.w2;
      for (int k = 0; k < deg; ++k) {
        X = X + c4[4 * (vb + k)];
        W1 = (W1 + c4[4 * (vb + k) + 1]) + c4[4 * (vb + k) + 3];
        W2 = (W2 + c4[4 * (vb + k) + 2]) + c1[vb + k];
      }
      const int o = L.perm[L.wg_vid[hl]];
      const float xn = prox_l1(p.x_min, p.x_max, p.step_x, p.data_factor * hg.weight[o], X, hg.data[o]);
      float nb = xn + p.theta * (xn - a.x);
      nb = nb < p.x_min ? p.x_min : nb;
      nb = nb > p.x_max ? p.x_max : nb;
      const float w1bn = W1 + p.theta * (W1 - a.w1), w2bn = W2 + p.theta * (W2 - a.w2);
      if (m & kWgTail) {
        area[par ^ 1][(size_t)wg * stride + loc] = Rec{nb, w1bn, w2bn, s + 1};
        if (m & kWgPublish) glob[par ^ 1][L.wg_info[4 * wg] + loc] = Rec{nb, w1bn, w2bn, s + 1};
      }
      a.xp = a.x, a.w1p = a.w1, a.w2p = a.w2;
      a.x = xn, a.w1 = W1, a.w2 = W2, a.xb = nb, a.w1b = w1bn, a.w2b = w2bn;
    }
  }
  // compare with the checker
  HostGraph ref = hg;
  flame_nltgv2_graph rg = ref.view();
  nltgv2_oracle_run(&p, &rg, n_iters);
  long bad = 0;
  for (size_t hl = 0; hl < NL; ++hl) {
    const uint32_t m = L.wg_meta[hl];
    if (!(m & kWgValid)) continue;
    const Lane& a = ln[hl];
    const int o = L.perm[L.wg_vid[hl]];
    const float got[9] = {a.x, a.w1, a.w2, a.xb, a.w1b, a.w2b, a.xp, a.w1p, a.w2p};
    const float want[9] = {ref.x[o], ref.w1[o], ref.w2[o], ref.xb[o], ref.w1b[o], ref.w2b[o], ref.xp[o], ref.w1p[o], ref.w2p[o]};
    if (std::memcmp(got, want, sizeof got) != 0) ++bad;
    if (m & kWgActive) {
      const int e = edge_of(L.wg_slot[hl]);
      const float gq[3] = {a.q1, a.q2, a.q3}, wq[3] = {ref.q1[e], ref.q2[e], ref.q3[e]};
      if (std::memcmp(gq, wq, sizeof gq) != 0) ++bad;
    }
  }
  if (bad) {
    std::printf("%ld lanes differ from the checker\n", bad);
    return 12;
  }
  long pub = 0, fetch = 0;
  for (size_t hl = 0; hl < NL; ++hl) pub += (L.wg_meta[hl] & (kWgTail | kWgPublish)) == (kWgTail | kWgPublish);
  for (int wg = 0; wg < L.wg_count; ++wg) fetch += L.wg_info[4 * wg + 1];
  std::printf("V=%d E=%d patches=%d records=%d lcap=%d rcap=%d slab slots=%d publishing=%ld fetched/step=%ld (half-edges %d) ok\n",
              V, g.E, L.wg_count, L.n_rec, L.wg_lcap, L.wg_rcap, L.wg_slab_slots, pub, fetch, 2 * g.E);
  return 0;
}

// The same replay for layout (E2): two half-edges per lane (slot 0 = half-edge 2j, slot 1 = 2j + 1 of the vertex whose j-th lane this is).
static int replay2(HostGraph& hg, int n_iters, const nltgv2_params& p) {
  flame_nltgv2_graph g = hg.view();
  PackedLayout L;
  if (build_layout(&g, &L, true, true) != 0) return 1;
  build_patch_rows2(&L);
  if (!L.wg2_ok) return 2;
  const int T = 64, V = g.V;
  std::vector<int> seen(L.n_slices * 64, 0);
  for (int wg = 0; wg < L.wg2_count; ++wg) {
    for (int t = 0; t < T; ++t) {
      const size_t hl = (size_t)wg * T + t;
      const uint32_t m = L.wg2_meta[hl];
      if (!(m & kWgValid)) {
        if (L.wg2_slot[(size_t)wg * 2 * T + t] >= 0 || L.wg2_slot[(size_t)wg * 2 * T + T + t] >= 0) return 3;
        continue;
      }
      const int first = m & 63, need = (m >> 6) & 127;
      if (need < 1 || first + need > 64 || first / 16 != (first + need - 1) / 16) return 4;
      if (need > 8 && first % 16 != 0) return 5;
      const int j = t - first;
      if (j < 0 || j >= need) return 6;
      if (((m & kWgHead) != 0) != (j == 0) || ((m & kWgTail) != 0) != (j == need - 1)) return 7;
      if (((m & kWgActive) != 0) != (L.wg2_slot[(size_t)wg * 2 * T + t] >= 0)) return 8;
      if (L.wg2_slot[(size_t)wg * 2 * T + T + t] >= 0 && !(m & kWgActive)) return 9;  // slot 1 only behind a slot 0
      if (m & kWgTail) seen[L.wg2_vid[hl]]++;
      if ((int)((m >> 13) & 2047) >= L.wg2_info[4 * wg + 2] || L.wg2_info[4 * wg + 2] > L.wg2_lcap) return 10;
      if (need > L.wg2_info[4 * wg + 3]) return 11;
    }
    if (L.wg2_info[4 * wg + 1] > L.wg2_rcap || L.wg2_info[4 * wg + 1] > T) return 12;
    for (int i = 1; i < L.wg2_info[4 * wg + 1]; ++i)
      if (L.wg2_fetch[(size_t)wg * T + i] <= L.wg2_fetch[(size_t)wg * T + i - 1]) return 13;
  }
  for (int v = 0; v < V; ++v)
    if (seen[L.iperm[v]] != 1) return 14;
  const size_t NL = (size_t)L.wg2_count * T;
  struct Lane { float x, w1, w2, xb, w1b, w2b, xp, w1p, w2p, q[2][3]; };
  std::vector<Lane> ln(NL);
  std::vector<Rec> glob[2] = {std::vector<Rec>(L.n_rec, Rec{0, 0, 0, 0}), std::vector<Rec>(L.n_rec, Rec{0, 0, 0, 0})};
  const int stride = L.wg2_lcap + L.wg2_rcap;
  std::vector<Rec> area[2] = {std::vector<Rec>((size_t)L.wg2_count * stride), std::vector<Rec>((size_t)L.wg2_count * stride)};
  auto slot_of = [&](size_t hl, int sl) { return L.wg2_slot[(hl / T) * 2 * T + (size_t)sl * T + (hl % T)]; };
  auto nbr_of = [&](size_t hl, int sl) { return L.wg2_nbr[(hl / T) * 2 * T + (size_t)sl * T + (hl % T)]; };
  const unsigned tag0 = 5;
  for (size_t hl = 0; hl < NL; ++hl) {
    const uint32_t m = L.wg2_meta[hl];
    if (!(m & kWgValid)) continue;
    const int o = L.perm[L.wg2_vid[hl]];
    Lane& a = ln[hl];
    a.x = hg.x[o], a.w1 = hg.w1[o], a.w2 = hg.w2[o], a.xb = hg.xb[o], a.w1b = hg.w1b[o], a.w2b = hg.w2b[o];
    a.xp = a.x, a.w1p = a.w1, a.w2p = a.w2;
    for (int sl = 0; sl < 2; ++sl)
      if (slot_of(hl, sl) >= 0) {
        const int e = L.rec_edge[slot_of(hl, sl)];
        a.q[sl][0] = hg.q1[e], a.q[sl][1] = hg.q2[e], a.q[sl][2] = hg.q3[e];
      }
    if (m & kWgHead) {
      const int wg = (int)(hl / T), loc = (m >> 13) & 2047;
      area[tag0 & 1][(size_t)wg * stride + loc] = Rec{a.xb, a.w1b, a.w2b, tag0};
      if (m & kWgPublish) glob[tag0 & 1][L.wg2_info[4 * wg] + loc] = Rec{a.xb, a.w1b, a.w2b, tag0};
    }
  }
  std::vector<float> c(NL * 2 * 5, -0.0f);  // per lane and slot: cx, a1, a2, b1, b2 (idle slots add -0.0)
  for (int it = 0; it < n_iters; ++it) {
    const unsigned s = tag0 + it;
    const int par = s & 1;
    for (int wg = 0; wg < L.wg2_count; ++wg)
      for (int i = 0; i < L.wg2_info[4 * wg + 1]; ++i) {
        const Rec r = glob[par][L.wg2_fetch[(size_t)wg * T + i]];
        if (r.tag != s) return 15;
        area[par][(size_t)wg * stride + L.wg2_lcap + i] = r;
      }
    for (size_t hl = 0; hl < NL; ++hl)
      for (int sl = 0; sl < 2; ++sl) {
        float* cc = &c[(hl * 2 + sl) * 5];
        cc[0] = cc[1] = cc[2] = cc[3] = cc[4] = -0.0f;
        const int slot = slot_of(hl, sl);
        if (slot < 0) continue;
        const int wg = (int)(hl / T), e = L.rec_edge[slot];
        const bool is_target = (L.rec_nbr[slot] & kRoleBit) != 0;
        const int code = nbr_of(hl, sl);
        const int idx = code < 0 ? L.wg2_lcap + (code & 0x7fffffff) : code;
        const Rec nb = area[par][(size_t)wg * stride + idx];
        if (nb.tag != s) return 16;
        // the lane's own (x_bar, w_bar): its vertex's record of this step (the head wrote it)
        const uint32_t m = L.wg2_meta[hl];
        const Rec own = area[par][(size_t)wg * stride + ((m >> 13) & 2047)];
        if (own.tag != s) return 17;
        Lane& a = ln[hl];
        const float alpha = hg.alpha[e], beta = hg.beta[e];
        const int si = hg.src[e], di = hg.dst[e];
        const float dx = hg.pos[2 * si] - hg.pos[2 * di], dy = hg.pos[2 * si + 1] - hg.pos[2 * di + 1];
        const float xbi = is_target ? nb.xb : own.xb, xbj = is_target ? own.xb : nb.xb;
        const float w1bi = is_target ? nb.w1b : own.w1b, w1bj = is_target ? own.w1b : nb.w1b;
        const float w2bi = is_target ? nb.w2b : own.w2b, w2bj = is_target ? own.w2b : nb.w2b;
        float K1 = alpha * (xbi - xbj);
        K1 -= alpha * dx * w1bi;
        K1 -= alpha * dy * w2bi;
        a.q[sl][0] = clampq(a.q[sl][0] + p.step_q * K1);
        a.q[sl][1] = clampq(a.q[sl][1] + p.step_q * (beta * (w1bi - w1bj)));
        a.q[sl][2] = clampq(a.q[sl][2] + p.step_q * (beta * (w2bi - w2bj)));
        const float t1 = a.q[sl][0] * p.step_x * alpha, t2 = a.q[sl][1] * p.step_x * beta, t3 = a.q[sl][2] * p.step_x * beta;
        cc[0] = is_target ? t1 : -t1;
        cc[1] = is_target ? t2 : t1 * dx;
        cc[2] = is_target ? t3 : t1 * dy;
        cc[3] = is_target ? -0.0f : -t2;
        cc[4] = is_target ? -0.0f : -t3;
      }
    for (size_t hl = 0; hl < NL; ++hl) {  // the head of every vertex: own slots first, then lane first + 1, ... in order
      const uint32_t m = L.wg2_meta[hl];
      if (!(m & kWgHead)) continue;
      const int wg = (int)(hl / T), need = (m >> 6) & 127, loc = (m >> 13) & 2047;
      Lane& a = ln[hl];
      float X = a.x, W1 = a.w1, W2 = a.w2;
      for (int j = 0; j < need; ++j)
        for (int sl = 0; sl < 2; ++sl) {
          const float* cc = &c[((hl + j) * 2 + sl) * 5];
          X = X + cc[0];
          W1 = (W1 + cc[1]) + cc[3];
          W2 = (W2 + cc[2]) + cc[4];
        }
      const int o = L.perm[L.wg2_vid[hl]];
      const float xn = prox_l1(p.x_min, p.x_max, p.step_x, p.data_factor * hg.weight[o], X, hg.data[o]);
      float nb = xn + p.theta * (xn - a.x);
      nb = nb < p.x_min ? p.x_min : nb;
      nb = nb > p.x_max ? p.x_max : nb;
      const float w1bn = W1 + p.theta * (W1 - a.w1), w2bn = W2 + p.theta * (W2 - a.w2);
      area[par ^ 1][(size_t)wg * stride + loc] = Rec{nb, w1bn, w2bn, s + 1};
      if (m & kWgPublish) glob[par ^ 1][L.wg2_info[4 * wg] + loc] = Rec{nb, w1bn, w2bn, s + 1};
      a.xp = a.x, a.w1p = a.w1, a.w2p = a.w2;
      a.x = xn, a.w1 = W1, a.w2 = W2, a.xb = nb, a.w1b = w1bn, a.w2b = w2bn;
    }
  }
  HostGraph ref = hg;
  flame_nltgv2_graph rg = ref.view();
  nltgv2_oracle_run(&p, &rg, n_iters);
  long bad = 0;
  for (size_t hl = 0; hl < NL; ++hl) {
    const uint32_t m = L.wg2_meta[hl];
    if (!(m & kWgValid)) continue;
    const Lane& a = ln[hl];
    if (m & kWgHead) {
      const int o = L.perm[L.wg2_vid[hl]];
      const float got[9] = {a.x, a.w1, a.w2, a.xb, a.w1b, a.w2b, a.xp, a.w1p, a.w2p};
      const float want[9] = {ref.x[o], ref.w1[o], ref.w2[o], ref.xb[o], ref.w1b[o], ref.w2b[o], ref.xp[o], ref.w1p[o], ref.w2p[o]};
      if (std::memcmp(got, want, sizeof got) != 0) ++bad;
    }
    for (int sl = 0; sl < 2; ++sl)
      if (slot_of(hl, sl) >= 0) {
        const int e = L.rec_edge[slot_of(hl, sl)];
        const float wq[3] = {ref.q1[e], ref.q2[e], ref.q3[e]};
        if (std::memcmp(a.q[sl], wq, sizeof wq) != 0) ++bad;
      }
  }
  if (bad) {
    std::printf("(E2) %ld lanes differ from the checker\n", bad);
    return 18;
  }
  long fetch = 0;
  for (int wg = 0; wg < L.wg2_count; ++wg) fetch += L.wg2_info[4 * wg + 1];
  std::printf("(E2) V=%d E=%d patches=%d (one half-edge per lane: %d) lcap=%d rcap=%d fetched/step=%ld ok\n", V, g.E, L.wg2_count, L.wg_count,
              L.wg2_lcap, L.wg2_rcap, fetch);
  return 0;
}

int main() {
  const nltgv2_params p = {0.1f, 0.001f, 125.0f, 0.25f, 0.0f, 10.0f};
  for (int frames : {1, 3})
    for (int variant = 0; variant < 3; ++variant) {  // hub of degree ~46 (a back-to-back patch among row-packed ones); degree <= 16; not row-packed by request
        const bool hub = variant == 0, rowpack = variant != 2;
        HostGraph g = make_graph(61, 47, frames, 1234 + frames, hub);
        const int rc = replay(g, 6, p, rowpack, rowpack);
        if (rc) {
          std::printf("FAILED frames=%d variant=%d rc=%d\n", frames, variant, rc);
          return 1;
        }
      }
  for (int frames : {1, 3}) {  // two half-edges per lane (no vertex of more than 32 edges)
    HostGraph g = make_graph(61, 47, frames, 4321 + frames, false);
    const int rc = replay2(g, 6, p);
    if (rc) {
      std::printf("FAILED (E2) frames=%d rc=%d\n", frames, rc);
      return 1;
    }
  }
  std::printf("all ok\n");
  return 0;
}
