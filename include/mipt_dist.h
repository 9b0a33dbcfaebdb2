/*
 * mipt_dist.h -- single-process multi-GPU host of the path tracer (libmipt_dist.so = libmipt.so + RCCL).
 *
 * NO COUNTERPART IN THE REFERENCE: 0x5b25/CUDAPathTracer is single-GPU (device ordinal 0 hard-coded,
 * CUDATracer/OptixTracerProg/OptixTracerProg.cpp:220-229; CUDATracerProg uses the current device).  BASELINE.json's
 * north_star adds it: "samples-per-pixel ... partitioned across the 8 GPUs of one node with the scene replicated ...
 * accumulation buffers combined with an RCCL reduce over xGMI", host side in C++.
 *
 * Model: one mipt_tracer and one host thread per GPU inside one process; the scene is uploaded to every
 * GPU; rank r of G renders frame indices first + r, first + r + G, ... into its own SUM buffer
 * (mipt_render(..., frame_stride = G, sum_mode = 1)); one ncclReduce(sum, fp32, root 0) of the w*h*4-float
 * accumulation buffers closes the job and the root tonemaps sum / n_frames.  There is no data-path collective.
 * Every tracer (and with it the three streams of its frame pipeline) is created BEFORE the RCCL communicators, so the
 * pipeline keeps hardware queues of its own (GPU_MAX_HW_QUEUES is 4 per process and device).
 * Functions return 0 or a negative MIPT_E_* code and never throw; mipt_dist_last_error gives the text.
 */
#ifndef MIPT_DIST_H
#define MIPT_DIST_H

#include "mipt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mipt_dist mipt_dist;

/* n_gpus = 0: every visible GPU.  Devices 0 .. n_gpus-1; rank r runs on device r; rank 0 is the root. */
int  mipt_dist_create(int n_gpus, mipt_dist** out);
void mipt_dist_destroy(mipt_dist*);
int  mipt_dist_gpus(const mipt_dist*);
const char* mipt_dist_last_error(const mipt_dist*);
mipt_tracer* mipt_dist_tracer(mipt_dist*, int rank);            /* for mipt_set_option on a rank's tracer */
/* Uploads the scene to every GPU (one host thread per GPU).  With desc->bvhCachePath set, rank 0 builds and writes the
 * BVH first and the others load it; without, every rank builds its own (identical: the build is deterministic). */
int  mipt_dist_scene_create(mipt_dist*, const mipt_scene_desc* desc);
/* Frames first_frame .. first_frame + n_frames - 1 of `settings`' viewport, split by sample index over the GPUs, summed,
 * reduced onto the root with one ncclReduce and tonemapped there.  acc_mean_host (w*h*4 floats, may be NULL) receives
 * sum / n_frames with alpha 1, ldr_host (w*h*4 bytes, may be NULL) the tonemapped image; row 0 = bottom, as mipt_trace.
 * Blocking. */
int  mipt_dist_render(mipt_dist*, const mipt_settings* settings, uint32_t first_frame, uint32_t n_frames,
                      float* acc_mean_host, unsigned char* ldr_host);
/* what the last mipt_dist_render measured: total rays over all GPUs, wall seconds of the render phase (max over ranks)
 * and of the reduce + tonemap phase */
typedef struct { uint64_t rays; double render_s, reduce_s; uint32_t frames; int gpus; } mipt_dist_stats;
int  mipt_dist_get_stats(const mipt_dist*, mipt_dist_stats* out);
/* the partition itself (pure arithmetic, usable without a GPU): rank `rank` of `world` renders `*count` frames starting
 * at `*first` with stride `world`.  Returns MIPT_E_INVALID for world < 1 or rank outside [0, world). */
int  mipt_dist_frames_for_rank(uint32_t first_frame, uint32_t n_frames, int rank, int world, uint32_t* first, uint32_t* count);

#ifdef __cplusplus
}
#endif
#endif
