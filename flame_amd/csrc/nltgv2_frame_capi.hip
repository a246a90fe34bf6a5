// nltgv2_frame_capi.hip -- the rows around the solver that work on its device state: mesh -> dense inverse-depth map, photometric
// residual (see nltgv2_context.hpp).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "nltgv2_context.hpp"

extern "C" {

// Shared tail of the two interpolate_mesh entry points: triangles/validity -> device, rasterise, copy back.
static int interpolate_common(flame_nltgv2_ctx* ctx, const int32_t* triangles, int32_t T, int32_t V,
                              const uint8_t* vtx_valid, const uint8_t* tri_valid, const float2* d_vtx,
                              const float* d_val, float value_scale, int rows, int cols, float* out, int32_t* coverage) {
  if (T < 0 || rows <= 0 || cols <= 0 || !out || (T > 0 && !triangles)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  for (int32_t t = 0; t < 3 * T; ++t)
    if (triangles[t] < 0 || triangles[t] >= V) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  const size_t n = (size_t)rows * (size_t)cols;
  int rc = ensure(ctx, ctx->r_tris, sizeof(int32_t) * 3 * (size_t)T);
  if (!rc) rc = ensure(ctx, ctx->r_valid, (size_t)T + (size_t)V + 16);
  if (!rc) rc = ensure(ctx, ctx->r_keys, sizeof(unsigned long long) * n);
  if (!rc) rc = ensure(ctx, ctx->r_img, sizeof(float) * n);
  if (!rc) rc = ensure(ctx, ctx->r_cov, sizeof(int));
  if (rc) return rc;
  if (T > 0) HIPCHK(ctx, hipMemcpyAsync(ctx->r_tris.p, triangles, sizeof(int32_t) * 3 * (size_t)T, hipMemcpyHostToDevice, ctx->stream));
  uint8_t* d_tv = nullptr;
  uint8_t* d_vv = nullptr;
  if (tri_valid && T > 0) {
    d_tv = (uint8_t*)ctx->r_valid.p;
    HIPCHK(ctx, hipMemcpyAsync(d_tv, tri_valid, (size_t)T, hipMemcpyHostToDevice, ctx->stream));
  }
  if (vtx_valid && V > 0) {
    d_vv = (uint8_t*)ctx->r_valid.p + (size_t)T;
    HIPCHK(ctx, hipMemcpyAsync(d_vv, vtx_valid, (size_t)V, hipMemcpyHostToDevice, ctx->stream));
  }
  LAUNCHCHK(ctx, launch_interpolate_mesh(T, (const int32_t*)ctx->r_tris.p, d_vtx, d_val, value_scale, d_vv, d_tv,
                                         (unsigned long long*)ctx->r_keys.p, (float*)ctx->r_img.p, (int*)ctx->r_cov.p,
                                         rows, cols, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->r_img.p, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  int cov = 0;
  HIPCHK(ctx, hipMemcpyAsync(&cov, ctx->r_cov.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));
  if (coverage) *coverage = cov;
  return FLAME_NLTGV2_OK;
}

// interpolate_mesh on the side stream (see flame_nltgv2.h): the canonical pos / x are read there while the solver already runs again
// on its packed state; the next unpack waits for ev_raster_done (ensure_canon), so do the kernels of a prepared sync that reuse them.
int flame_nltgv2_interpolate_mesh_begin(flame_nltgv2_ctx* ctx, const int32_t* triangles, int32_t T, const uint8_t* tri_valid,
                                        int rows, int cols, float graph_scale) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_interpolate_mesh_begin");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  const int32_t V = ctx->L.V;
  if (T < 0 || rows <= 0 || cols <= 0 || (T > 0 && !triangles)) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  for (int32_t t = 0; t < 3 * T; ++t)
    if (triangles[t] < 0 || triangles[t] >= V) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  const size_t n = (size_t)rows * (size_t)cols;
  hipStream_t rs = ctx->raster_stream;
  const bool trace = std::getenv("FLAME_NLTGV2_TRACE") != nullptr;
  const auto tr0 = std::chrono::steady_clock::now();
  auto tr_us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count(); };
  double tr_a = 0, tr_b = 0, tr_c = 0, tr_d = 0;
  HIPCHK(ctx, hipStreamSynchronize(rs));  // (a begin without its end: the pinned map and the device buffers are about to be reused)
  if (ctx->h_img_cap < n + 16) {
    request_open_stop(ctx);  // (the pinned allocator waits for the device)
    if (ctx->h_img) (void)hipHostFree(ctx->h_img);
    ctx->h_img = nullptr, ctx->h_img_cap = 0;
    if (hipHostMalloc((void**)&ctx->h_img, sizeof(float) * (n + 16), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return fail(ctx, FLAME_NLTGV2_ERR_OOM);
    }
    ctx->h_img_cap = n + 16;
  }
  // the caller's triangles go up first (pageable memory: the copy holds the host), while the solver still runs ...
  rc = ensure(ctx, ctx->r_tris, sizeof(int32_t) * 3 * (size_t)T);
  if (!rc) rc = ensure(ctx, ctx->r_tvalid, (size_t)T + 16);  // (its own buffer: project_graph writes r_valid on the context's stream)
  if (!rc) rc = ensure(ctx, ctx->r_keys, sizeof(unsigned long long) * n);
  if (!rc) rc = ensure(ctx, ctx->r_img, sizeof(float) * n);
  if (!rc) rc = ensure(ctx, ctx->r_cov, sizeof(int));
  if (rc) return rc;
  if (T > 0) HIPCHK(ctx, hipMemcpyAsync(ctx->r_tris.p, triangles, sizeof(int32_t) * 3 * (size_t)T, hipMemcpyHostToDevice, rs));
  uint8_t* d_tv = nullptr;
  if (tri_valid && T > 0) {
    d_tv = (uint8_t*)ctx->r_tvalid.p;
    HIPCHK(ctx, hipMemcpyAsync(d_tv, tri_valid, (size_t)T, hipMemcpyHostToDevice, rs));
  }
  tr_a = tr_us();
  if (ctx->opt_mesh_state == 1 && !ctx->canon_valid && ctx->snap_topo == ctx->topo) {
    // FLAME_NLTGV2_OPT_MESH_STATE = 1 and runs enqueued since the last settle: the map is of the state that settle left, which the
    // canonical arrays still hold (enqueue_run recorded ev_snap behind their last writer); the runs in flight are not waited for
    tr_b = tr_us();
    HIPCHK(ctx, hipStreamWaitEvent(rs, ctx->ev_snap, 0));
  } else {
    rc = ensure_canon(ctx);  // ... it stops here ...
    if (rc) return rc;
    tr_b = tr_us();
    HIPCHK(ctx, hipEventRecord(ctx->ev_canon, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(rs, ctx->ev_canon, 0));  // ... and may go on as soon as the caller enqueues the next run
  }
  LAUNCHCHK(ctx, launch_interpolate_mesh(T, (const int32_t*)ctx->r_tris.p, ctx->c.pos, ctx->c.x, graph_scale, nullptr, d_tv,
                                         (unsigned long long*)ctx->r_keys.p, (float*)ctx->r_img.p, (int*)ctx->r_cov.p, rows, cols, rs));
  // the rasteriser's kernels are the last readers of the canonical arrays (and the writers of the resident map): whoever rewrites those
  // waits for THEM, not for the map's way out to the host (0.05 ms at 640x480, 0.3 ms at 1920x1080 -- time a commit would stand still for)
  tr_c = tr_us();
  HIPCHK(ctx, hipEventRecord(ctx->ev_raster_done, rs));
  ctx->raster_inflight = true;
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_img, ctx->r_img.p, sizeof(float) * n, hipMemcpyDeviceToHost, rs));
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_img + n, ctx->r_cov.p, sizeof(int), hipMemcpyDeviceToHost, rs));
  ctx->map_rows = rows, ctx->map_cols = cols;
  ctx->img_pending_rows = rows, ctx->img_pending_cols = cols;  // (what _end describes: a synchronous interpolate_mesh in between changes map_rows)
  tr_d = tr_us();
  if (trace)
    std::fprintf(stderr, "[flame_nltgv2] interpolate_mesh_begin: checks + triangles up %.1f us, solver settled + unpack enqueued %.1f, rasteriser enqueued %.1f, copies out enqueued %.1f\n",
                 tr_a, tr_b - tr_a, tr_c - tr_b, tr_d - tr_c);
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_interpolate_mesh_end(flame_nltgv2_ctx* ctx, const float** map_out, int32_t* coverage_out) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_interpolate_mesh_end");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->h_img || ctx->img_pending_rows == 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  HIPCHK(ctx, hipStreamSynchronize(ctx->raster_stream));
  const size_t n = (size_t)ctx->img_pending_rows * (size_t)ctx->img_pending_cols;
  if (map_out) *map_out = ctx->h_img;
  if (coverage_out) std::memcpy(coverage_out, ctx->h_img + n, sizeof(int32_t));
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_interpolate_mesh(flame_nltgv2_ctx* ctx, const int32_t* triangles, int32_t T, const uint8_t* tri_valid,
                                  int rows, int cols, float graph_scale, float* idepthmap_out, int32_t* coverage_out) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_interpolate_mesh");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  rc = ensure_canon(ctx);
  if (rc) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->raster_stream));  // (r_img may still be on its way out for an interpolate_mesh_begin)
  rc = interpolate_common(ctx, triangles, T, ctx->L.V, nullptr, tri_valid, ctx->c.pos, ctx->c.x, graph_scale, rows,
                          cols, idepthmap_out, coverage_out);
  if (!rc) ctx->map_rows = rows, ctx->map_cols = cols;  // (the map stays on the device: flame_nltgv2_sync_input.init_from_map)
  return rc;
}

int flame_nltgv2_interpolate_mesh_arrays(flame_nltgv2_ctx* ctx, const int32_t* triangles, int32_t T, const float* vertices_xy,
                                         const float* values, int32_t V, const uint8_t* vtx_valid,
                                         const uint8_t* tri_valid, int rows, int cols, float* img_out,
                                         int32_t* coverage_out) {
  flame_hip::RoctxRange roctx_range_("flame_nltgv2_interpolate_mesh_arrays");
  int rc = enter(ctx);
  if (rc) return rc;
  if (V < 0 || (V > 0 && (!vertices_xy || !values))) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  rc = ensure(ctx, ctx->r_vtx, sizeof(float) * 2 * (size_t)V);
  if (!rc) rc = ensure(ctx, ctx->r_val, sizeof(float) * (size_t)V);
  if (rc) return rc;
  if (V > 0) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->r_vtx.p, vertices_xy, sizeof(float) * 2 * (size_t)V, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->r_val.p, values, sizeof(float) * (size_t)V, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->raster_stream));
  ctx->map_rows = ctx->map_cols = 0;  // (r_img will hold an image of the caller's arrays, not the graph's map)
  return interpolate_common(ctx, triangles, T, V, vtx_valid, tri_valid, (const float2*)ctx->r_vtx.p,
                            (const float*)ctx->r_val.p, 1.0f, rows, cols, img_out, coverage_out);
}

int flame_nltgv2_photo_set_images(flame_nltgv2_ctx* ctx, const uint8_t* ref, const uint8_t* cmp, int rows, int cols,
                                  int step_bytes) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ref || !cmp || rows < 2 || cols < 2 || step_bytes < cols) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  const size_t bytes = (size_t)rows * (size_t)step_bytes;
  HIPCHK(ctx, wait_solver_stream(ctx));
  rc = ensure(ctx, ctx->img_ref, bytes + 16);
  if (!rc) rc = ensure(ctx, ctx->img_cmp, bytes + 16);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->img_ref.p, ref, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->img_cmp.p, cmp, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));
  ctx->img_rows = rows, ctx->img_cols = cols, ctx->img_step = step_bytes;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_photo_residual(flame_nltgv2_ctx* ctx, const float* KRKinv, const float* Kt, float graph_scale,
                                int border, float* err_out) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!KRKinv || !Kt || !err_out || border < 1 || ctx->img_rows == 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  rc = ensure_canon(ctx);
  if (rc) return rc;
  const size_t fV = sizeof(float) * (size_t)ctx->L.V;
  rc = ensure(ctx, ctx->photo_err, fV);
  if (rc) return rc;
  PhotoGeometry geo;
  std::memcpy(geo.KRKinv, KRKinv, sizeof(geo.KRKinv));
  std::memcpy(geo.Kt, Kt, sizeof(geo.Kt));
  LAUNCHCHK(ctx, launch_photo_residual(ctx->c, graph_scale, geo, (const uint8_t*)ctx->img_ref.p,
                                       (const uint8_t*)ctx->img_cmp.p, ctx->img_rows, ctx->img_cols, ctx->img_step,
                                       border, (float*)ctx->photo_err.p, ctx->stream));
  if (fV) HIPCHK(ctx, hipMemcpyAsync(err_out, ctx->photo_err.p, fV, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_photo_fuse(flame_nltgv2_ctx* ctx, const float* KRKinv, const float* Kt, float graph_scale, int border,
                            int enable) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!enable) {
    ctx->photo_fused = false;
    return FLAME_NLTGV2_OK;
  }
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!KRKinv || !Kt || border < 1 || ctx->img_rows == 0) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  HIPCHK(ctx, wait_solver_stream(ctx));
  rc = ensure(ctx, ctx->photo_err, sizeof(float) * (size_t)ctx->L.V);
  if (rc) return rc;
  std::memcpy(ctx->photo_geo.KRKinv, KRKinv, sizeof(ctx->photo_geo.KRKinv));
  std::memcpy(ctx->photo_geo.Kt, Kt, sizeof(ctx->photo_geo.Kt));
  ctx->photo_scale = graph_scale, ctx->photo_border = border;
  ctx->photo_fused = true;
  return FLAME_NLTGV2_OK;
}

int flame_nltgv2_photo_residual_last(flame_nltgv2_ctx* ctx, float* err_out) {
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->have_graph) return fail(ctx, FLAME_NLTGV2_ERR_NO_GRAPH);
  if (!ctx->photo_fused || !err_out) return fail(ctx, FLAME_NLTGV2_ERR_INVALID_ARG);
  if (ctx->pending.active) {
    rc = finish(ctx);
    if (rc) return rc;
  }
  const size_t fV = sizeof(float) * (size_t)ctx->L.V;
  if (fV) HIPCHK(ctx, hipMemcpyAsync(err_out, ctx->photo_err.p, fV, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, wait_solver_stream(ctx));
  return FLAME_NLTGV2_OK;
}


}  // extern "C"
