/*
 * flame_nltgv2.h -- C-ABI of libflame_nltgv2_hip.so: the MI355X (gfx950) implementation of FLaME's
 * NLTGV2-L1 primal-dual graph regularizer.
 *
 * This is the drop-in boundary for ONE path of robustrobotics/flame:
 *   src/flame/optimizers/nltgv2_l1_graph_regularizer.{h,cc}   (namespace
 *   flame::optimizers::nltgv2_l1_graph_regularizer), called from src/flame/flame.cc:104 (solver
 *   thread) and flame.cc:2172-2173 (cost statistics).
 * Every entry point below names the reference interface it replaces.  Plain pointers and sizes
 * only; no C++/torch types.  All pointers are HOST memory unless the name says "device".
 *
 * Conventions
 *   - All functions return 0 (FLAME_NLTGV2_OK) on success or a negative flame_nltgv2_status.
 *     Nothing here aborts or throws (the reference's FLAME_ASSERT calls exit(1), assert.h:111).
 *   - A context is NOT thread-safe; use one per solver thread.  The caller provides the
 *     graph_mtx_-equivalent (flame.h:539; lock sites flame.cc:103, 302, 309, 329, 365).
 *   - Edge k is directed src[k] -> dst[k] == (boost::source, boost::target) of the k-th edge of
 *     boost::edges(graph) (nltgv2...cc:91-96).  Orientation is semantic (the operator uses the
 *     SOURCE vertex's w_bar only, cc:100-101,129-133); edge order fixes the floating-point
 *     accumulation order of primalStep (cc:120-142) and is honoured exactly.
 *   - There is no CPU fallback: every compute entry point fails with
 *     FLAME_NLTGV2_ERR_NO_DEVICE / _HIP when no gfx950 device is usable.
 */
#ifndef FLAME_NLTGV2_H_
#define FLAME_NLTGV2_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLAME_NLTGV2_ABI_VERSION 7 /* 7: flame_nltgv2_run_open, flame_nltgv2_iterations, FLAME_NLTGV2_OPT_MESH_STATE; 6: the region-per-workgroup form left the library (FLAME_NLTGV2_OPT_PERSISTENT = 7 is an invalid argument, last_run_path 8 never occurs, the two info words it used are reserved; the struct's layout is unchanged); 5: flame_nltgv2_info grew (last_run_waves_per_cu, ..., replays_per_step); flame_nltgv2_stream_wait_run, _runs_in_flight */

typedef struct flame_nltgv2_ctx flame_nltgv2_ctx;

typedef enum flame_nltgv2_status {
  FLAME_NLTGV2_OK = 0,
  FLAME_NLTGV2_ERR_INVALID_ARG = -1, /* NULL pointer, negative size, index out of range, self loop */
  FLAME_NLTGV2_ERR_NO_DEVICE = -2,   /* no HIP device / wrong device ordinal */
  FLAME_NLTGV2_ERR_HIP = -3,         /* a HIP runtime call failed (see flame_nltgv2_last_hip_error) */
  FLAME_NLTGV2_ERR_NO_GRAPH = -4,    /* compute call before flame_nltgv2_upload_graph */
  FLAME_NLTGV2_ERR_NAN = -5,         /* a dual variable became NaN/Inf: the condition on which the
                                        reference's FLAME_ASSERT(!std::isnan(new_q)) fires
                                        (nltgv2...h:174).  Reported once by the call that finds it; the state
                                        stays readable (q was clamped to +-1 where it happened) and usable. */
  FLAME_NLTGV2_ERR_OOM = -6,
  FLAME_NLTGV2_ERR_TIMEOUT = -7,     /* persistent run: a bounded neighbour wait expired.  run()/sync() normally hide
                                        this: the run's results go to second copies of the state, so the state is
                                        rolled back and the steps are redone with one launch per step (get_info:
                                        timeouts_recovered); runs chained with run_async() are covered too (the chain's
                                        starting state is copied aside once, the chain replayed).  Returned only if that
                                        recovery itself is impossible */
  FLAME_NLTGV2_ERR_ASSERT = -8       /* flame_stereo.h: an input on which the reference's FLAME_ASSERT would
                                        exit(1) (assert.h:111) */
} flame_nltgv2_status;

/* == struct Params, nltgv2_l1_graph_regularizer.h:121-129 (same fields, order and defaults). */
typedef struct flame_nltgv2_params {
  float data_factor; /* 0.1   lambda */
  float step_x;      /* 0.001 primal step tau */
  float step_q;      /* 125   dual step sigma */
  float theta;       /* 0.25  extra-gradient step */
  float x_min;       /* 0     feasible set */
  float x_max;       /* 10 */
} flame_nltgv2_params;

/* Flat (structure-of-arrays) image of a reference Graph (h:107-112): VertexData (h:74-90) and
 * EdgeData (h:95-102; `valid` is syncGraph bookkeeping, flame.cc:2080-2118, and has no image here).
 * Used read-only by upload and write-only by download; in a download any pointer may be NULL to
 * skip that array. */
typedef struct flame_nltgv2_graph {
  int32_t V, E;
  float* pos; /* 2*V floats, (x,y) interleaved: VertexData::pos */
  float *x, *w1, *w2;
  float *x_bar, *w1_bar, *w2_bar;
  float *x_prev, *w1_prev, *w2_prev; /* upload: may be NULL (step() overwrites them, cc:35-42) */
  float *data_term, *data_weight;
  int32_t *src, *dst;
  float *alpha, *beta;
  float *q1, *q2, *q3;
} flame_nltgv2_graph;

void flame_nltgv2_default_params(flame_nltgv2_params* p);

/* Context lifetime.  `device` is the HIP device ordinal (one process / one context per GPU).
 * Replaces: the Graph object owned by Flame (flame.h:536) as the holder of solver state. */
int flame_nltgv2_create(flame_nltgv2_ctx** out, int device);
int flame_nltgv2_destroy(flame_nltgv2_ctx* ctx);

/* Run all subsequent work of this context on the caller's hipStream_t (pass NULL to return to the
 * context's own stream).  Lets a host framework time / order the solver with its own events. */
int flame_nltgv2_set_stream(flame_nltgv2_ctx* ctx, void* hip_stream);

/* Topology + weights + state -> device.  Replaces the graph (re)construction of
 * Flame::syncGraph (flame.cc:2030-2121: add_vertex / add_edge / remove_*), after which the packed
 * device layout is rebuilt.  Buffers are reused and grown geometrically across calls. */
int flame_nltgv2_upload_graph(flame_nltgv2_ctx* ctx, const flame_nltgv2_graph* g);

/* Per-frame graph synchronisation with WARM START: the graph-edit part of Flame::syncGraph
 * (flame.cc:1985-2121) and the vertex removal of Flame::projectGraph (flame.cc:1923-1931).  The caller
 * passes the vertex set of the new frame (stable feature ids, positions, data terms) and the edges of its
 * new triangulation; vertices/edges that survive keep their primal/dual state (an edge also keeps its old
 * orientation, as boost::edge(u,v) finds it either way, flame.cc:2094-2100), new ones start as the
 * reference initialises them.  Order of the resulting edge list: surviving edges in their previous
 * relative order, then new edges in triangulator order -- what boost::edges() yields after the
 * reference's erase/add_edge sequence.  (The reference's VERTEX order is BGL hash order and therefore
 * unspecified; here it is the caller's.)  Feature ids default to the vertex index after upload_graph. */
typedef struct flame_nltgv2_sync_input {
  int32_t V;
  const int32_t* feat_id;   /* [V] unique, >= 0, stable across frames (Flame::feat_to_vtx_, flame.h:542-543) */
  const float* pos;         /* [2V] positions in the new frame (after projectGraph re-projection) */
  const float* data_term;   /* [V] feat.idepth_mu / graph_scale */
  const float* data_weight; /* [V] */
  const float* init_x;      /* [V] x of NEW vertices (flame.cc:2160-2162); NULL = data_term */
  int32_t E;
  const int32_t* edges;     /* [2E] triangulator->edges(): pairs of indices into the new vertex list */
  int32_t check_sticky_obstacles; /* params.check_sticky_obstacles, flame.cc:2011 */
  float sticky_threshold;         /* 0.25f in the reference */
  float init_graph_scale;         /* > 0: a NEW vertex whose init_x is NaN (no valid prediction) starts at the mean of
                                   * x*scale over its neighbours with data_weight > 0, / scale -- all means formed from
                                   * the neighbours' values as they stand after the vertex pass (survivors: their x; new:
                                   * init_x, or data_term where that is NaN too), neighbours in ascending edge id --, or at
                                   * data_term when it has none (init_with_prediction, flame.cc:2133-2158).  0: init_x is
                                   * taken as it is. */
  int32_t edges_unique;           /* != 0: the caller vouches that `edges` lists every vertex pair at most once (what
                                   * flame_delaunay_triangulate -- and the reference's Triangle -- return): the search for
                                   * duplicates among the new edges is skipped.  0 (default): pairs that repeat are dropped as
                                   * boost::edge() / add_edge do it in the reference (flame.cc:2094-2100), the first one stays */
  int32_t init_from_map;          /* != 0 (init_x must be NULL, init_graph_scale > 0): a NEW vertex starts at the prediction the
                                   * reference reads from its dense map, idepthmap(pos.y + 0.5f, pos.x + 0.5f) / graph_scale
                                   * (init_with_prediction, flame.cc:2131) -- looked up ON THE DEVICE in the map the context's last
                                   * flame_nltgv2_interpolate_mesh[_begin] left there (no map yet, or a position outside it: NaN, i.e.
                                   * the neighbours' mean as above).  Saves the host gather and the upload of init_x every frame */
} flame_nltgv2_sync_input;
int flame_nltgv2_sync_graph(flame_nltgv2_ctx* ctx, const flame_nltgv2_sync_input* in);
/* The same sync in two halves, so that the solver keeps iterating while the frame's graph is prepared (the reference's solver thread
 * stands still through all of Flame::syncGraph: it holds graph_mtx_, flame.cc:103, 309-318):
 *   prepare  checks the inputs (same errors as sync_graph), copies them, and enqueues the construction of the new topology on a
 *            side stream -- it only READS the live graph; returns at once.  The caller's arrays are free on return.
 *   commit   waits for that construction, settles the runs enqueued meanwhile, swaps the new topology in and moves the state
 *            (what sync_graph does after its index maps).  sync_graph == prepare; commit.  Whatever of this does not need the solver
 *            stopped is enqueued beside or behind the runs still in flight (the new graph's per-slot and per-lane tables on the side
 *            stream into spare buffers; the state's unpack and its gather into spare arrays behind the runs): the solver's stream
 *            stands empty for ~50 us at 640x480.
 * Between the two only run / run_async / sync and read-outs (download_state, export, get_info) may be called; upload_graph,
 * sync_graph, set_feature_ids or another prepare cancel the prepared sync (commit then returns FLAME_NLTGV2_ERR_INVALID_ARG). */
int flame_nltgv2_sync_prepare(flame_nltgv2_ctx* ctx, const flame_nltgv2_sync_input* in);
int flame_nltgv2_sync_commit(flame_nltgv2_ctx* ctx);
/* Declares the feature ids of the vertices of a graph brought in with flame_nltgv2_upload_graph (V ints,
 * unique): Flame::vtx_to_feat_ (flame.h:542-543). */
int flame_nltgv2_set_feature_ids(flame_nltgv2_ctx* ctx, const int32_t* feat_id);
/* Current edge list / feature ids (E, E, V ints; any pointer may be NULL). */
int flame_nltgv2_get_topology(flame_nltgv2_ctx* ctx, int32_t* src, int32_t* dst, int32_t* feat_id);
/* Vertex and edge count of the current graph (after a sync: V as passed, E = surviving + new edges). */
int flame_nltgv2_graph_size(flame_nltgv2_ctx* ctx, int32_t* V, int32_t* E);

/* Flame::projectGraph (flame.cc:1888-1905) on the device state: every vertex is re-projected into the new
 * frame with EpipolarGeometry::project(pos, x*graph_scale, &u_new, &idepth_new)
 * (stereo/epipolar_geometry.h:152-180); pos and x = idepth_new/graph_scale are written back; keep_out[v] = 0
 * where the reference would remove the vertex (outside the valid region, cv::Rect_<float>::contains, or
 * idepth_new < 0; flame.cc:1902-1906).  pos_out (2V floats, may be NULL) receives the new positions for the
 * host's re-triangulation.  Removal itself happens in the following flame_nltgv2_sync_graph, as in the
 * reference (projectGraph is always followed by syncGraph, flame.cc:301-318).
 * All matrices row-major 3x3 as EpipolarGeometry holds them; q = (w,x,y,z). */
typedef struct flame_nltgv2_projection {
  float K[9], Kinv[9], KRKinv[9];
  float q_ref_to_cmp[4];
  float t_ref_to_cmp[3];
  float region_x, region_y, region_w, region_h; /* valid_region, flame.cc:1881-1884 */
} flame_nltgv2_projection;
int flame_nltgv2_project_graph(flame_nltgv2_ctx* ctx, const flame_nltgv2_projection* pr, float graph_scale,
                               uint8_t* keep_out, float* pos_out);
/* The rescale_data block of Flame::update (flame.cc:328-351): new_scale = mean(data_term*graph_scale);
 * x, x_bar, x_prev, data_term *= graph_scale/new_scale; params->data_factor *= new_scale/graph_scale.
 * (The reference forms the mean over a hash set, i.e. in an unspecified order; here in a fixed order -- 1024 strided partial sums
 * combined pairwise -- that the checker restates: oracle/photometric_oracle.c strided_tree_sum.) */
int flame_nltgv2_rescale_data(flame_nltgv2_ctx* ctx, float graph_scale, float* new_graph_scale, flame_nltgv2_params* params);

/* Per-frame refresh of data_term/data_weight for an unchanged topology (flame.cc:1985-2018). */
int flame_nltgv2_update_data(flame_nltgv2_ctx* ctx, const float* data_term, const float* data_weight);

/* State only (x,w,x_bar,w_bar,q; NULL members are left untouched) for an unchanged topology,
 * e.g. after projectGraph re-projected x (flame.cc:1898-1900) or rescale_data (flame.cc:328-351). */
int flame_nltgv2_upload_state(flame_nltgv2_ctx* ctx, const flame_nltgv2_graph* state);

/* n_iters x step(params, graph) (nltgv2...cc:33-49), state resident on the device; returns after
 * the work has completed.  Replaces the body of the solver thread loop, flame.cc:101-107. */
int flame_nltgv2_run(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n_iters);
/* Same, but only enqueues; flame_nltgv2_sync waits and reports the NaN flag. */
int flame_nltgv2_run_async(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n_iters);
int flame_nltgv2_sync(flame_nltgv2_ctx* ctx);
/* Same as run, additionally returns the device time of the n_iters steps measured with HIP events
 * on the stream the kernels run on (what bench.py reports). */
int flame_nltgv2_run_timed(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int n_iters, float* elapsed_ms);

/* The individually callable pieces of a step, each one kernel sweep over the canonical SoA state:
 *   save_prev           step()'s x_prev/w_prev copy                 cc:35-42
 *   dual_step           internal::dualStep                          cc:89-114
 *   primal_step         internal::primalStep (+ proxL1)             cc:116-154
 *   extragradient_step  internal::extraGradientStep                 cc:156-174
 * step == save_prev; dual; primal; extragradient == flame_nltgv2_run(ctx,p,1). */
int flame_nltgv2_save_prev(flame_nltgv2_ctx* ctx);
int flame_nltgv2_dual_step(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p);
int flame_nltgv2_primal_step(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p);
int flame_nltgv2_extragradient_step(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p);
int flame_nltgv2_step(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p);

/* smoothnessCost (cc:51-71) and dataCost (cc:73-85); cost() (h:149-151) is their sum.  The device forms the
 * addends exactly as the reference does, the host adds them sequentially in float in edge / vertex order: the
 * results equal the reference's to the last bit (for the caller's vertex order; the reference's is BGL hash order). */
int flame_nltgv2_costs(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, float* smoothness, float* data);

/* Device -> host copy of the solver state in the caller's original vertex/edge order.  Replaces
 * the read-back loop flame.cc:372-380 (and is the checkpoint format). */
int flame_nltgv2_download_state(flame_nltgv2_ctx* ctx, flame_nltgv2_graph* out);

/* Writes scale * x[v] (original vertex order, V floats) into DEVICE memory `dst_device`, on the
 * context's stream: the `vtx.x * graph_scale_` of flame.cc:377, kept on the GPU so that a
 * multi-GPU host can hand the buffer straight to an RCCL gather. */
int flame_nltgv2_export_idepth_device(flame_nltgv2_ctx* ctx, void* dst_device, float scale);
/* Same, enqueue only: ordered on the context's stream (see flame_nltgv2_set_stream), no host wait -- a
 * collective enqueued on the same stream afterwards reads the finished buffer. */
int flame_nltgv2_export_idepth_device_async(flame_nltgv2_ctx* ctx, void* dst_device, float scale);
/* Standing export target: while `dst_device` is set (NULL = off), every run()/run_async() also leaves
 * scale * x[v] there (V floats, original vertex order) as part of the same launch -- the persistent kernels write
 * it in their epilogue, so a per-step result gather needs no extra kernel between two runs. */
int flame_nltgv2_set_export_target(flame_nltgv2_ctx* ctx, void* dst_device, float scale);
/* Makes another stream of the caller (`hip_stream`, a hipStream_t; not the context's own) wait for everything enqueued on the context's
 * stream so far -- the consumer side of the standing export target (the read-back of flame.cc:372-380 left on the device): a collective
 * on `hip_stream` then reads the finished row.  No
 * host wait.  Called right behind run_async() it costs the solver's stream nothing: the run's launch carries the event as its own
 * completion signal, where an event recorded by the caller is one more operation between two solver launches on an in-order queue
 * (5 us each at 640x480).  FLAME_NLTGV2_ERR_INVALID_ARG for a null stream or the context's own. */
int flame_nltgv2_stream_wait_run(flame_nltgv2_ctx* ctx, void* hip_stream);
/* How many of the last two runs enqueued with run_async() the device has not finished yet (0, 1 or 2): what a free-running solver
 * thread (flame.cc:99-112) paces itself by -- it keeps two runs in flight, so that the device never waits for the host between two of them, without ever
 * blocking in the context (include/flame_hip/solver_loop.hpp, device mode).  Does not wait, does not check the runs' results (sync()
 * does), and costs the solver's stream nothing where the launch carries the event. */
int flame_nltgv2_runs_in_flight(flame_nltgv2_ctx* ctx, int32_t* n_out);
/* A run that goes on until the caller needs the state -- the reference's solver thread is `while (true) step()` (flame.cc:99-112), and a
 * frame loop that enqueues rounds of N iterations pays each round's start-up and the gap between two launches (12 us per round at 640x480)
 * and waits for up to two rounds whenever it needs the state.  run_open enqueues ONE launch of at most max_iters (even) iterations; the next
 * call that needs the solver settled (sync, download_state, project_graph, sync_commit, a further run, ...) asks it to stop: one patch reads the
 * request and publishes the iteration every patch leaves at (~0.15 ms later at 640x480).  *opened = 0: not applicable here (a graph beyond ONE
 * launch of the patch-per-wave or the two-half-edges form at <= 20 / 14 patches per CU, record verification or the probe on) -- NOTHING was
 * enqueued and nothing waited for, use run_async.  How many iterations an open run did is known once it is settled: flame_nltgv2_iterations.  An expired open run is taken back
 * and redone like any other.  flame_nltgv2_stream_wait_run / _runs_in_flight see it like a run_async (in flight until it has been stopped). */
int flame_nltgv2_run_open(flame_nltgv2_ctx* ctx, const flame_nltgv2_params* p, int max_iters, int32_t* opened);
/* Iterations applied to the state by all runs of this context so far (run, run_async: counted when enqueued; an open run: when settled --
 * *open_in_flight = 1 says one is not yet counted).  Never waits. */
int flame_nltgv2_iterations(flame_nltgv2_ctx* ctx, int64_t* total, int32_t* open_in_flight);

/* Mesh -> dense inverse-depth map, the step right after the solver each frame:
 * utils::interpolateMesh (utils/image_utils.cc:373-396) over utils::DrawShadedTriangleBarycentric
 * (utils/rasterization.cc:164-246), as called at flame.cc:409-415, plus the coverage count of
 * flame.cc:428-437.  `triangles`: T index triples in the reference triangulator's order and winding
 * (utils::Delaunay::triangles()); later triangles win on shared pixels, exactly like the reference's
 * sequential loop.  Output: rows*cols floats, NaN where no triangle covers the pixel.
 *   interpolate_mesh         vertices = the context's graph: pos and x*graph_scale straight from the device
 *                            state (flame.cc:372-380 without the host round trip)
 *   interpolate_mesh_arrays  the reference signature: explicit vertices / values / validity arrays */
int flame_nltgv2_interpolate_mesh(flame_nltgv2_ctx* ctx, const int32_t* triangles, int32_t T, const uint8_t* tri_valid,
                                  int rows, int cols, float graph_scale, float* idepthmap_out, int32_t* coverage_out);
/* interpolate_mesh in two halves, so that the solver iterates while the map is rasterised and copied out: begin settles the runs,
 * brings the state to its canonical arrays and enqueues rasteriser + device-to-host copy on a side stream, returning at once (the
 * caller goes on with run_async); end waits for that stream.  *map_out points at the context's pinned rows*cols floats (valid
 * until the next begin); the map also stays on the device for flame_nltgv2_sync_input.init_from_map. */
int flame_nltgv2_interpolate_mesh_begin(flame_nltgv2_ctx* ctx, const int32_t* triangles, int32_t T, const uint8_t* tri_valid,
                                        int rows, int cols, float graph_scale);
int flame_nltgv2_interpolate_mesh_end(flame_nltgv2_ctx* ctx, const float** map_out, int32_t* coverage_out);
int flame_nltgv2_interpolate_mesh_arrays(flame_nltgv2_ctx* ctx, const int32_t* triangles, int32_t T,
                                         const float* vertices_xy, const float* values, int32_t V,
                                         const uint8_t* vtx_valid, const uint8_t* tri_valid, int rows, int cols,
                                         float* img_out, int32_t* coverage_out);

/* 2-D Delaunay triangulation of float32 points: the counterpart of utils::Delaunay
 * (src/flame/utils/delaunay.{h,cc}, a wrapper of the vendored Shewchuk Triangle called with "zneQB",
 * delaunay.cc:66-68) that feeds Flame::syncGraph its edge list (flame.cc:2073-2104) and interpolateMesh its
 * triangles.  HOST code, like the reference's; exact predicates, so a point set in
 * general position yields the same set of triangles as Triangle.  Triangles are counter-clockwise in
 * x-right / y-up coordinates (Triangle's convention); edges are unique undirected pairs.  Pass NULL for
 * `triangles` / `edges` to query the counts (Euler: T <= 2n - 5, E <= 3n - 6).  Needs no context, no GPU. */
int flame_delaunay_triangulate(const float* xy, int32_t n, int32_t* triangles, int32_t tri_capacity,
                               int32_t* n_triangles, int32_t* edges, int32_t edge_capacity, int32_t* n_edges);

/* Per-vertex photometric residual (BASELINE config 5).  No live reference counterpart: the only
 * occurrence is the commented-out block flame.cc:854-893; built from the live pieces
 * EpipolarGeometry::project (stereo/epipolar_geometry.h:127-143) and utils::bilinearInterp<uint8_t,float>
 * (utils/image_utils.h:230-255).  err[v] = |I_cmp(project(pos_v, x_v*graph_scale)) - I_ref(pos_v)|, NaN
 * where either pixel is outside [border, size-border).  An epilogue: it never modifies x.
 * KRKinv: 9 floats row-major, Kt: 3 floats (EpipolarGeometry::loadGeometry, h:88-93). */
int flame_nltgv2_photo_set_images(flame_nltgv2_ctx* ctx, const uint8_t* ref, const uint8_t* cmp, int rows, int cols,
                                  int step_bytes);
int flame_nltgv2_photo_residual(flame_nltgv2_ctx* ctx, const float* KRKinv, const float* Kt, float graph_scale,
                                int border, float* err_out);
/* Standing form of the same residual, computed as part of the solver's own launch (BASELINE config 5: "photometric
 * data-term residual fused into the primal step"): while enabled, every run()/run_async() leaves the residual of its
 * final x in a device buffer -- the persistent kernels evaluate it in their epilogue, right after the last primal
 * step, from the registers that hold x; the one-launch-per-step path appends one sweep.  It never feeds back into x.
 * flame_nltgv2_photo_residual_last copies that buffer out (V floats, caller's vertex order; no kernel is launched).
 * The images are those of flame_nltgv2_photo_set_images; enable = 0 switches it off. */
int flame_nltgv2_photo_fuse(flame_nltgv2_ctx* ctx, const float* KRKinv, const float* Kt, float graph_scale, int border,
                            int enable);
int flame_nltgv2_photo_residual_last(flame_nltgv2_ctx* ctx, float* err_out);

/* Options (flame_nltgv2_set_option).  These are the stable surface; a caller never needs any of them -- every default is the
 * measured best.  (Tuning knobs and test hooks of the current kernels, numbers 100 and up, live in a test-only header.) */
enum {
  FLAME_NLTGV2_OPT_SOLVER = 1,       /* 0 = fused / persistent kernels (default), 1 = the reference's four loops one by one
                                        (save_prev / dual / primal / extragradient sweeps on the canonical arrays) */
  FLAME_NLTGV2_OPT_USE_HIPGRAPH = 2, /* 1 (default) = the one-launch-per-step path replays its launches from a hipGraph */
  FLAME_NLTGV2_OPT_PERSISTENT = 5,   /* 1 (default) = run() uses ONE persistent launch for all n_iters steps when the graph
                                        fits on the chip, picking the form by size; 3 = the vertex-per-lane form by name,
                                        4 = the patch-per-wave form by name, 6 = the patch-per-wave form with two half-edges
                                        per lane by name -- each only if it fits; 0 = always one launch per step.  (2 was round 1's
                                        lane-per-half-edge form, retired in round 3; 7 was round 5's region-per-workgroup form, measured
                                        15 % slower than the patch-per-wave form at every BASELINE size -- profiles/r05_wg_region.txt --
                                        and taken out in round 6; 2, 5 and 7: invalid argument) */
  FLAME_NLTGV2_OPT_PROBE = 12,       /* 1 = the patch-per-wave kernel records a per-patch, per-step cycle probe (8 words:
                                        HW id, XCC id, wait cycles, compute cycles, poll rounds, step start, 100 MHz clock,
                                        0), read with flame_nltgv2_read_probe; 0 (default) = off */
  FLAME_NLTGV2_OPT_VERIFY_RECORDS = 14, /* persistent kernels: 1 = after a neighbour record's tag matched, read the 16 bytes
                                        once more and compare all four dwords (the exchange relies on an aligned 16-byte
                                        access never being torn between payload and tag; this checks it at run time, at the
                                        price of one more load round trip per step); a difference takes the run back like a
                                        timeout and counts in flame_nltgv2_info.torn_records_detected.  2 = the same plus a
                                        test hook that corrupts one re-read.  0 (default) = off */
  FLAME_NLTGV2_OPT_PLACEMENT = 16,   /* patch-per-wave form on all eight XCDs: 1 (default) = the records another XCD reads are
                                        placed on memory pages whose home channel suits that pair of XCDs (a hand-off across
                                        XCDs takes 0.39-0.66 us depending on the page; measured once per context, ~3 ms at
                                        the first such run), 0 = every record at its linear place.  Addresses only: results
                                        are bit-identical either way */

  FLAME_NLTGV2_OPT_SYNC_PATH = 17,   /* flame_nltgv2_sync_graph: 0 (default) = index maps and the new graph's layout tables are built on the
                                        device wherever that applies (a duplicate-free edge list -- edges_unique --, no vertex of more than 64
                                        edges; feature ids of any magnitude), on the host otherwise; 1 = always on the host; 2 = on the device or
                                        FLAME_NLTGV2_ERR_INVALID_ARG.  Same result either way */

  FLAME_NLTGV2_OPT_COST_SUM = 18,    /* flame_nltgv2_costs: 0 (default) = the addends are summed sequentially in float, in the reference's edge
                                        order and the caller's vertex order, on the host: smoothnessCost equals the reference's to the last
                                        bit; 1 = both sums on the device in a fixed strided / pairwise order (a few microseconds, no copy of
                                        2E + V floats; agrees with the sequential sums to ~1e-6 relative) */

  FLAME_NLTGV2_OPT_MESH_STATE = 19,  /* flame_nltgv2_interpolate_mesh_begin with runs enqueued since the last call that settled the solver:
                                        0 (default) = waits for them and rasterises the state they leave; 1 = rasterises the state that
                                        call left (the canonical arrays as they have stood since: a run works on its packed copies) and
                                        waits for nothing -- the runs in flight go on beside the rasteriser.  The reference's
                                        interpolateMesh reads whatever iterate its solver thread has reached (flame.cc:372-380 under
                                        graph_mtx_): the state right after syncGraph is one of them.  What a frame loop gains: it can
                                        enqueue the next round BEFORE it prepares the mesh (SolverLoop::withDevice(f, g)) */

  FLAME_NLTGV2_OPT_EXPERIMENTAL = 100 /* option numbers from here on are tuning knobs and test hooks of the current kernels: declared in
                                         flame_amd/csrc/flame_nltgv2_test_options.h (tests and tools only), not part of this surface */
};
int flame_nltgv2_set_option(flame_nltgv2_ctx* ctx, int option, int value);

typedef struct flame_nltgv2_info {
  int32_t abi_version;
  int32_t device;
  int32_t V, E;
  int32_t n_slices;        /* 64-vertex slices of the packed (SELL-64) layout */
  int32_t max_degree;
  int64_t padded_half_edges; /* slots in the packed half-edge arrays (>= 2*E) */
  int64_t device_bytes;    /* device memory held by the context */
  int64_t algorithmic_bytes_per_iter; /* 64*V + 40*E, SURVEY.md section 8(d) */
  int32_t compute_units;
  char device_name[64];
  char gcn_arch[32];
  int32_t last_run_path; /* 0 none, 1 persistent launch (lane per half-edge), 2 one launch per step
                            (hipGraph), 3 one launch per step (eager), 4 four canonical sweeps per step,
                            5 persistent launch (vertex per lane), 6 persistent launch (patch per wave),
                            7 persistent launch (patch per wave, two half-edges per lane)  (8 was the region-per-workgroup
                            form of ABI 5) */
  int32_t he_waves;      /* always 0 (the lane-per-half-edge form, retired in round 3; kept for the layout of the struct) */
  int32_t tv_waves;      /* waves of the vertex-per-lane persistent form (0: not applicable) */
  int32_t tv_wave_capacity; /* vertex-per-lane waves the device keeps resident */
  int32_t last_run_groups;  /* persistent launches the last run was split into (groups of whole components) */
  int32_t timeouts_recovered; /* persistent runs whose wait expired: state rolled back, steps redone one launch per step */
  int32_t patches;       /* waves (patches of ~10 vertices) of the patch-per-wave persistent form (0: not applicable) */
  int32_t torn_records_detected; /* persistent runs stopped by FLAME_NLTGV2_OPT_VERIFY_RECORDS (a record whose second
                                    read differed from the first): rolled back and redone the same way */
  int32_t last_sync_path; /* flame_nltgv2_sync_graph: 0 none yet, 1 index maps + layout tables on the host, 2 on the device */
  int32_t last_run_waves_per_cu; /* waves per compute unit of the last persistent run's largest launch (0: none) */
  int32_t reserved0, reserved1; /* always 0 (ABI 5: regions, region_depth of the region-per-workgroup form; kept for the layout of the struct) */
  int32_t replays_per_step; /* expired chains whose persistent replay (reduced residency) expired too: redone one launch per step */
} flame_nltgv2_info;
int flame_nltgv2_get_info(flame_nltgv2_ctx* ctx, flame_nltgv2_info* info);

/* Error reporting. */
int flame_nltgv2_last_error(flame_nltgv2_ctx* ctx);     /* last non-OK status of this context */
int flame_nltgv2_last_hip_error(flame_nltgv2_ctx* ctx); /* raw hipError_t of the last HIP failure */
const char* flame_nltgv2_status_string(int status);
int flame_nltgv2_abi_version(void);

/* Measurement aid: copies out the cycle probe the last patch-per-wave run recorded (FLAME_NLTGV2_OPT_PROBE):
 * [patch][step][8] words; *n_words = words available.  Not part of the reference's surface. */
int flame_nltgv2_read_probe(flame_nltgv2_ctx* ctx, uint32_t* out, int64_t max_words, int64_t* n_words);

/* Measurement aid: the record placement (FLAME_NLTGV2_OPT_PLACEMENT).  *state: 1 = the page ranking of the device is known (measured
 * once per device and process, when the first context is created), 0 = not yet (the measurement did not complete then: the first run
 * that can use it tries again), -1 = unavailable (it did not complete then either; records keep their linear places).
 * *placed_records = records of the current topology that were given a place in the pool (per step parity);
 * us[0..2] = one-way hand-off by choice of page as the calibration measured it, mean over the XCD pairs: the best page,
 * the mean page, the worst page.  Any output pointer may be NULL.  Not part of the reference's surface. */
int flame_nltgv2_placement_info(flame_nltgv2_ctx* ctx, int32_t* state, int32_t* placed_records, float* us);

/* Test hook: the layout arrays the device expanded for the current topology (nltgv2_layout.hip) compared word for word
 * with the host builders (nltgv2_pack.hpp); *mismatches = number of differing words -- plus, when records are placed, the
 * number of placed offsets that are misaligned, out of the pool or given out twice. */
int flame_nltgv2_layout_selftest(flame_nltgv2_ctx* ctx, int64_t* mismatches);
/* Host-only packing probe (no device needed; used by the CPU test-suite): builds the packed
 * SELL-64 layout the fused sweep runs on and copies it out.  Any output pointer may be NULL.
 *   perm[n_slices*64]       packed slot -> original vertex id (-1 = padding lane)
 *   slice_row[n_slices+1]   first 64-wide row of each slice in the half-edge arrays
 *   rec_nbr[rows*64]        packed neighbour index | role<<31 (role 1: this vertex is the TARGET)
 *   rec_edge[rows*64]       original edge id of the slot (-1 = empty)
 * Returns n_slices (>= 0) or a negative status; *rows_out receives the number of 64-wide rows. */
int flame_nltgv2_pack_probe(const flame_nltgv2_graph* g, int32_t* perm, int32_t* slice_row,
                            int32_t* rec_nbr, int32_t* rec_edge, int64_t capacity_rows, int64_t* rows_out);

#ifdef __cplusplus
}
#endif
#endif /* FLAME_NLTGV2_H_ */
