// nltgv2_kernels.h -- argument bundles and launch wrappers shared by the kernel files (nltgv2_kernels.hip,
// nltgv2_persistent*.hip, nltgv2_layout.hip) and the host side (nltgv2_context.hpp and the C-ABI files behind it).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace flame_hip {

// == flame_nltgv2_params == Params, nltgv2_l1_graph_regularizer.h:121-129
struct SolverParams {
  float data_factor, step_x, step_q, theta, x_min, x_max;
};

// K*R*Kinv (row-major) and K*t of EpipolarGeometry::loadGeometry (stereo/epipolar_geometry.h:88-93)
struct PhotoGeometry {
  float KRKinv[9];
  float Kt[3];
};

// Standing photometric target of the persistent runs (flame_nltgv2_photo_fuse): err == nullptr means off.
struct PhotoFuse {
  const float2* pos = nullptr;  // canonical positions, caller's vertex order
  const uint8_t* ref = nullptr;
  const uint8_t* cmp = nullptr;
  float* err = nullptr;         // [V], caller's vertex order
  PhotoGeometry geo{};
  float graph_scale = 1.0f;
  int rows = 0, cols = 0, step = 0, border = 0;
};

// Standing outputs of a persistent run, read once in the kernels' epilogue from device memory.
struct RunTail {
  float* export_out = nullptr;  // flame_nltgv2_set_export_target (nullptr: off)
  float export_scale = 1.0f;
  int reserved_ = 0;
  PhotoFuse photo;              // flame_nltgv2_photo_fuse (photo.err == nullptr: off)
  unsigned* progress = nullptr; // FLAME_NLTGV2_TRACE: where a patch that leaves a run through an expired wait says how far it got
  const unsigned* stop_req = nullptr;  // an open run (flame_nltgv2_run_open): the device word that asks it to stop (the host copies a 1 into it)
};

// Everything EpipolarGeometry::project(u, idepth, &u_new, &idepth_new) reads (stereo/epipolar_geometry.h:152-180)
// plus the valid region of Flame::projectGraph (flame.cc:1881-1884).
struct ProjectGeometry {
  float K[9], Kinv[9], KRKinv[9];
  float q[4];  // q_ref_to_cmp as (w, x, y, z)
  float t[3];
  float rx, ry, rw, rh;
};

// Canonical SoA state in the caller's vertex / edge order (device pointers).
struct CanonArgs {
  int V = 0, E = 0;
  float2* pos = nullptr;
  float *x = nullptr, *w1 = nullptr, *w2 = nullptr;
  float *xb = nullptr, *w1b = nullptr, *w2b = nullptr;
  float *xp = nullptr, *w1p = nullptr, *w2p = nullptr;
  float *data = nullptr, *weight = nullptr;
  int32_t *src = nullptr, *dst = nullptr;
  float *alpha = nullptr, *beta = nullptr;
  float *q1 = nullptr, *q2 = nullptr, *q3 = nullptr;
  int32_t* row_ptr = nullptr;  // [V+1]
  uint32_t* half = nullptr;    // [2E] edge id | role bit
  int* err = nullptr;
};

// Packed SELL-64 layout of the fused sweep (device pointers).
struct FusedArgs {
  hipEvent_t stop_event = nullptr;  // a plain (not cooperative) persistent launch carries it as its completion signal (nltgv2_run.hip)
  int open_run = 0;                 // the patch-per-wave kernel's OPEN instance: the run ends when the host asks (RunTail::stop_req), n_iters at the latest
  int n_slices = 0;
  int64_t n_slots = 0;  // (rows + kRowPad) * 64
  int32_t* slice_row = nullptr;
  int32_t* perm = nullptr;  // [n_slices*64]
  int32_t* pdeg = nullptr;
  uint32_t* rec_nbr = nullptr;  // [n_slots]
  int32_t* rec_edge = nullptr;  // [n_slots]
  int32_t* edge_src_slot = nullptr;  // [E]
  int4* hrec = nullptr;     // [n_slots] {nbr|role, alpha, dx, dy}
  float4* hq = nullptr;     // [n_slots] {q1,q2,q3,beta}
  float4* vstate = nullptr; // [n_slices*64] {x,w1,w2,data}
  float4* hq_out = nullptr;     // the other copies: a persistent run writes its results there and the host swaps
  float4* vstate_out = nullptr; // the roles once the run is known to have succeeded
  float2* vaux = nullptr;   // [n_slices*64] {data_weight, degree bits}
  float4* bar[2] = {nullptr, nullptr};  // ping-pong {x_bar,w1_bar,w2_bar,-}
  float4* vprev = nullptr;  // {x_prev,w1_prev,w2_prev,-} written by the last step of a run
  void* xbuf = nullptr;  // persistent run: [R0|L0|R1|L1|XCC] exchange buffer, see nltgv2_persistent.hip
  int tv_waves = 0;                    // persistent run, vertex-per-lane rows (nltgv2_pack.hpp (D))
  int32_t* tv_slot = nullptr;
  int32_t* tv_vid = nullptr;
  uint32_t* tv_meta = nullptr;
  uint32_t* tv_wave = nullptr;
  // persistent run, patch-per-wave rows (nltgv2_pack.hpp (E)): one wave per patch
  int wg_count = 0, wg_lcap = 0, wg_slab_slots = 0;
  int wg_rowpack = 0;                  // patches are row-packed (nltgv2_pack.hpp): the DPP form of k_persistent_pv runs them
  int32_t* wg_slot = nullptr;
  int32_t* wg_vid = nullptr;
  uint32_t* wg_meta = nullptr;
  int32_t* wg_nbr = nullptr;
  int32_t* wg_fetch = nullptr;
  int32_t* wg_info = nullptr;
  int n_rec = 0;                       // record ids in use (= V): sizes the exchange buffers
  int wg_poll_gap = 1;                 // 1: one s_sleep between the polls of k_persistent_pv, 0: none
  char* place_pool = nullptr;          // record placement (nltgv2_layout.hip): pool of pages for the remote copies of the
  const int32_t* rec_off = nullptr;    // records other XCDs read; rec_off[parity * stride + record] = byte offset or -1
  int rec_off_stride = 0;
  unsigned* rot_word = nullptr;        // ... and the word in which block 0 of a launch says which XCD it is on
  unsigned* probe = nullptr;           // optional per-patch, per-step cycle probe of k_persistent_pv (tools/pv_probe.py)
  int* abort_flag = nullptr;
  int* err = nullptr;
};

int launch_fused_step(const FusedArgs& a, const SolverParams& p, int parity, bool write_prev, int unroll,
                      int waves_per_block, hipStream_t stream);
int launch_persistent_run(const FusedArgs& a, const SolverParams& p, int form, int wave_begin, int n_waves,
                          int parity_in, unsigned tag0, int n_iters, int waves_per_block, unsigned max_spins,
                          int presleep, int dual, int tv_static_in_lds, int xcds, const RunTail* tail, bool cooperative,
                          hipStream_t stream);
// Record placement (nltgv2_layout.hip): calibrate a pool of 2 x kPlacePages pages, then per topology give the records read
// across XCDs a slot on a page that suits their pair of XCDs
constexpr int kPlacePages = 96;
int launch_place_calibrate(char* pool, int n_pages, int iters, unsigned* out, int* xcc_out, int* fail, hipStream_t s);
int launch_place_records(const CanonArgs& c, const FusedArgs& a, int per_xcd, const int32_t* order_m, const int32_t* rid_of,
                         int32_t* patch_of_rec, int8_t* cls, const uint16_t* ranking, int n_pages, int* fill, int32_t* rec_off,
                         int stride, hipStream_t s);
// One upload blob -> its buffers (and clears), nltgv2_layout.hip
constexpr uint32_t kScatterFill = 0xffffffffu;
struct ScatterEntry {
  void* dst;
  uint32_t src_off;  // byte offset in the blob (16-byte aligned), or kScatterFill: fill with `fill`
  uint32_t fill;
  size_t bytes;
};
constexpr int kScatterMax = 48;
struct ScatterTable {
  int n = 0;
  ScatterEntry e[kScatterMax];
};
int launch_scatter(const ScatterTable& t, const void* blob, hipStream_t s);
// Up to four device-to-device copies of whole float4 arrays in ONE launch (the start-of-chain snapshot: three memcpy calls were three
// copy kernels and 19 us of the solver's queue between the first and the second round after every hold of a frame loop)
struct CopyTable {
  int n = 0;
  struct { void* dst; const void* src; size_t bytes; } e[4];  // bytes: multiples of 16
};
int launch_copy_arrays(const CopyTable& t, hipStream_t s);
// flame_nltgv2_sync_graph on the device: index maps in, state gathered from the previous arrays (o / oq) into new ones (n / nq)
struct SyncArgs {
  int V = 0, E = 0;
  const int32_t* old_of_new = nullptr;       // [V] previous index of a surviving vertex, -1 new
  const int32_t* old_of_new_edge = nullptr;  // [E] previous index of a surviving edge, -1 new
  const float* data = nullptr;               // [V] new data terms, weights
  const float* weight = nullptr;
  const float* init_x = nullptr;             // [V] or nullptr
  const float* init_map = nullptr;           // or: the dense inverse-depth map of the last interpolate_mesh (device), rows x cols --
  int map_rows = 0, map_cols = 0;            // init = map(int(pos.y + 0.5), int(pos.x + 0.5)) / graph_scale (flame.cc:2131), NaN outside
  int check_sticky = 0;
  float sticky_threshold = 0.0f;
  float graph_scale = 0.0f;                  // > 0: a NaN init value is replaced by the neighbours' mean (flame.cc:2133-2158)
  const float* o[9] = {};                    // previous x, w1, w2, x_bar, w1_bar, w2_bar, x_prev, w1_prev, w2_prev
  float* n[9] = {};
  const float* oq[3] = {};
  float* nq[3] = {};
  const int32_t *src = nullptr, *dst = nullptr, *row_ptr = nullptr;  // the NEW topology
  const uint32_t* half = nullptr;
  const float2* pos = nullptr;
  float *alpha = nullptr, *beta = nullptr;
  uint8_t* need_nbr = nullptr;               // [V] scratch
};
int launch_sync_state(const SyncArgs& a, hipStream_t s);
int pv_patches_per_cu(const FusedArgs& a, bool verify);
// layout (E2) on the device (nltgv2_persistent_pv2.hip: two half-edges per lane; experimental)
struct Pv2Args {
  const int32_t* slot = nullptr;   // [count*2*64]
  const int32_t* vid = nullptr;    // [count*64]
  const uint32_t* meta = nullptr;  // [count*64]
  const int32_t* nbr = nullptr;    // [count*2*64]
  const int32_t* fetch = nullptr;  // [count*64]
  const int32_t* info = nullptr;   // [count*4]
  int count = 0, lcap = 0;
};
int launch_build_patches2(const CanonArgs& c, const FusedArgs& a, int n_patches, int32_t* wg_info, const int32_t* order_m, const int32_t* rid_tab,
                          const uint8_t* vfirst, const int32_t* iperm, int32_t* wg_slot, int32_t* wg_vid, uint32_t* wg_meta, int32_t* wg_nbr,
                          int32_t* wg_fetch, int* rmax, hipStream_t s);
int pv2_patches_per_cu(int lcap, bool verify);
int launch_persistent_pv2(const FusedArgs& a, const Pv2Args& w, const SolverParams& p, int wg_begin, int n_wgs, int parity_in, unsigned tag0,
                          int n_iters, unsigned max_spins, int poll_gap, int dual, const RunTail* tail, bool cooperative, hipStream_t stream);
int pv_real_waves_per_simd(bool verify_or_probe);
const void* persistent_tv_kernel(bool static_in_lds, int waves_per_block, unsigned* lds_bytes);  // nltgv2_persistent_tv.hip
// device-side expansion of the layout arrays (nltgv2_layout.hip)
int launch_build_sell(const CanonArgs& c, const FusedArgs& a, const int32_t* iperm, hipStream_t s);
int launch_build_patches(const CanonArgs& c, const FusedArgs& a, const int32_t* wg_v0, const int32_t* order_m,
                         const int32_t* rid_tab, const uint8_t* vfirst, const int32_t* iperm, hipStream_t s);
// ---- the per-VERTEX layout tables built on the device (nltgv2_topo.hip) ---------------------------------------------------
// What the host reads back once the device has built a topology.
struct TopoDims {
  int32_t n_edges;            // edges of the new graph
  int32_t n_keep;             // sync: edges that survived from the previous graph
  int32_t rows;               // 64-wide rows of the SELL-64 layout (B)
  int32_t max_degree;
  int32_t wg_count, wg_lcap;  // (E): patches, most vertices of one patch
  int32_t wg2_count, wg2_lcap;  // (E2)
  int32_t flags;              // kTopoBad*: the device build cannot stand, the host builders take over
  int32_t pad_[7];
};
constexpr int32_t kTopoBadDegree = 1;  // a vertex of more than 64 edges
constexpr int32_t kTopoBadEdges = 2;   // sync: the new edge list is not the triangulator's (a vertex pair listed twice)
constexpr int kTopoInf = 0x7f7f7f7f;
// Everything the builder reads and writes (device pointers; scratch is carved out of one buffer by the host).
struct TopoBuild {
  int V = 0, E = 0;    // the new graph; sync: E = triangulator edges, every one of which becomes an edge
  int Vo = 0, Eo = 0;  // sync: the previous graph
  int n_slices = 0;
  // inputs
  const int32_t* fid = nullptr;        // sync: [V] feature ids of the new vertices
  const float2* pos = nullptr;         // [V]
  const int32_t* tri_edges = nullptr;  // sync: [2E] vertex pairs in triangulator order
  float minx = 0, miny = 0, sx = 0, sy = 0;  // Morton quantisation of pos (nltgv2_pack.hpp: build_layout)
  // sync: the previous topology and the feature table (id -> vertex of the graph of generation gen_prev)
  const int32_t* o_row_ptr = nullptr;
  const uint32_t* o_half = nullptr;
  const int32_t *o_src = nullptr, *o_dst = nullptr;
  // two open-addressing tables of 2^tab_bits slots each (>= 4 slots per vertex), table g & 1 holds the graph of generation g: a slot
  // is live iff its stamp equals that generation -- nothing is ever cleared, feature ids may be any non-negative int32
  uint32_t* feat_stamp = nullptr;  // [2 << tab_bits]
  int32_t* feat_key = nullptr;
  int32_t* feat_val = nullptr;
  int tab_bits = 0;
  uint32_t gen_prev = 0, gen_new = 0;
  // scratch
  int32_t *old_edge = nullptr, *first_k = nullptr;  // [E]: previous edge | orientation bit 31, -1 none; [Eo]: first triangulator edge keeping it
  int32_t* scan = nullptr;                           // [Eo + E + V + 1]: exclusive sums of [survivor flags | new-edge flags | degrees | 0]
  int32_t *deg = nullptr, *cur = nullptr;            // [V]
  int cc_bits = 1;                                   // 2^cc_bits >= V: the priority space of the union-find (nltgv2_topo.hip: cc_mix)
  int32_t *parent = nullptr, *minid = nullptr;       // [2^cc_bits] indexed by priority: parent, smallest vertex id of the tree rooted there
  uint32_t* morton = nullptr;                        // [V]
  uint64_t* key_out = nullptr;                       // [V] sorted (component label << 32) | Morton code
  int32_t* width = nullptr;                          // [n_slices]
  uint8_t* wflag = nullptr;                          // [V up to 16384] walk input: degree | component begins << 7
  uint8_t* vf[2] = {nullptr, nullptr};               // [V up to 16384] walk output of (E) / (E2): first lane | patch begins << 7
  int32_t* seg_count[2] = {nullptr, nullptr};        // [segments] patches per walk segment
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  int* counters = nullptr;                           // [4] "last block" tickets
  // outputs: the new topology
  int32_t* old_of_new = nullptr;       // sync: [V] previous vertex or -1
  int32_t* old_of_new_edge = nullptr;  // sync: [E] previous edge or -1
  int32_t *src = nullptr, *dst = nullptr;  // sync: written; upload: given
  int32_t* row_ptr = nullptr;
  uint32_t* half = nullptr;
  int32_t *order_m = nullptr, *rid_of = nullptr, *perm = nullptr, *iperm = nullptr, *pdeg = nullptr, *slice_row = nullptr;
  int32_t *wg_info = nullptr, *wg_v0 = nullptr, *wg2_info = nullptr, *wg2_v0 = nullptr;
  uint8_t *wg_vfirst = nullptr, *wg2_vfirst = nullptr;
  TopoDims* dims = nullptr;
};
size_t topo_sort_temp_bytes(int V, int n_scan);
int launch_topo_feat_build(const int32_t* feat, int V, uint32_t* stamp, int32_t* key, int32_t* val, int tab_bits, uint32_t gen, hipStream_t s);
int launch_topo_sync_front(const TopoBuild& t, hipStream_t s);
int launch_topo_upload_front(const TopoBuild& t, hipStream_t s);
int launch_topo_back(const TopoBuild& t, hipStream_t s);

// one per translation unit with kernels: loads its code object (flame_nltgv2_create of the first context of a process)
void warm_module_kernels();
void warm_module_persistent(hipStream_t stream, bool cooperative);
void warm_module_persistent_tv();
void warm_module_persistent_pv2();
void warm_module_layout();
void warm_module_topo();

int launch_save_prev(const CanonArgs& c, hipStream_t s);
int launch_dual(const CanonArgs& c, const SolverParams& p, hipStream_t s);
int launch_primal(const CanonArgs& c, const SolverParams& p, hipStream_t s);
int launch_extragradient(const CanonArgs& c, const SolverParams& p, hipStream_t s);
int launch_pack_static(const CanonArgs& c, const FusedArgs& a, hipStream_t s);
int launch_pack_state(const CanonArgs& c, const FusedArgs& a, int parity, bool with_static, hipStream_t s);  // (+ the records when asked)
int launch_unpack_state(const CanonArgs& c, const FusedArgs& a, int parity, bool have_prev, hipStream_t s);
int launch_export(const CanonArgs& c, const FusedArgs& a, bool packed_current, float scale, float* dst,
                  hipStream_t s);
// keep: one byte per vertex (device memory, or pinned host memory the kernel writes straight into); pos_before (optional): the positions as they stood
int launch_project_graph(const CanonArgs& c, float graph_scale, const ProjectGeometry& geo, uint8_t* keep, float2* pos_before, hipStream_t s);
int launch_rescale(const CanonArgs& c, float graph_scale, float* new_scale_dev, hipStream_t s);
int launch_interpolate_mesh(int T, const int32_t* tris, const float2* vtx, const float* values, float value_scale,
                            const uint8_t* vtx_valid, const uint8_t* tri_valid, unsigned long long* keys, float* img,
                            int* coverage, int rows, int cols, hipStream_t s);
int launch_photo_residual(const CanonArgs& c, float graph_scale, const PhotoGeometry& geo, const uint8_t* ref,
                          const uint8_t* cmp, int rows, int cols, int step, int border, float* err, hipStream_t s);
int launch_photo_residual_packed(const FusedArgs& a, const PhotoFuse& photo, hipStream_t s);
int launch_cost_terms(const CanonArgs& c, float* terms, hipStream_t s);
int launch_cost_sums(const CanonArgs& c, const float* terms, float* out2, hipStream_t s);

}  // namespace flame_hip
