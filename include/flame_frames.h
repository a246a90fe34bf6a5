/*
 * flame_frames.h -- C-ABI of the multi-GPU result gather of robustrobotics/flame's regularizer path on an 8 x MI355X
 * node; part of libflame_nltgv2_hip.so.  SURVEY.md section 8(e), BASELINE configuration 4.
 *
 * Frames are independent: device k solves frame k with its own flame_nltgv2_ctx and nothing is exchanged while it
 * iterates.  The one exchange is the read-back Flame::update does after the solve
 * (/root/reference/src/flame/flame.cc:372-380: idepth = x * graph_scale per vertex): here every device's x * graph_scale
 * row is gathered to every device with ONE ncclAllGather per device (RCCL over xGMI), grouped over the devices of this
 * process -- the host that "stays C++" is one process that owns all GPUs of the node, as the reference's Flame object
 * owns its solver thread.  (bench.py / flame_amd/frames.py do the same gather with one process per GPU through
 * torch.distributed, as the bench driver launches it.)
 *
 * RCCL is loaded when the first flame_frames_create runs (dlopen of librccl.so), not with the library.
 * Status codes are flame_nltgv2_status.  Not thread-safe: one flame_frames_ctx per driving thread.
 */
#ifndef FLAME_FRAMES_H_
#define FLAME_FRAMES_H_

#include <stdint.h>

#include "flame_nltgv2.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct flame_frames_ctx flame_frames_ctx;

/* One communicator per listed device (ncclCommInitAll), one stream per device, and on every device a send row of `vmax`
 * floats (zero-filled) and a receive block of n_devices x vmax floats.  `vmax` = the largest vertex count of any frame. */
int flame_frames_create(flame_frames_ctx** out, int n_devices, const int* devices, int32_t vmax);
int flame_frames_destroy(flame_frames_ctx* ctx);
int flame_frames_count(const flame_frames_ctx* ctx);
/* Device k's send row: what its solver exports into (flame_nltgv2_set_export_target(solver_k, row, graph_scale)), and
 * the stream to run that solver on (flame_nltgv2_set_stream(solver_k, stream)) so that the export is ordered before the
 * gather without a host round trip. */
int flame_frames_local_row(flame_frames_ctx* ctx, int k, void** row_device);
int flame_frames_stream(flame_frames_ctx* ctx, int k, void** hip_stream);
/* Enqueues the gather on every device's stream (ncclGroupStart .. ncclAllGather x n .. ncclGroupEnd) and returns. */
int flame_frames_gather(flame_frames_ctx* ctx);
/* Waits for all streams. */
int flame_frames_wait(flame_frames_ctx* ctx);
/* Device k's receive block (n_devices x vmax floats, row j = frame j), and a copy of it to the host (waits). */
int flame_frames_gathered(flame_frames_ctx* ctx, int k, void** block_device);
int flame_frames_download(flame_frames_ctx* ctx, int k, float* host_block);
/* Text of the last RCCL / loader error ("" if none). */
const char* flame_frames_last_error_text(const flame_frames_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* FLAME_FRAMES_H_ */
