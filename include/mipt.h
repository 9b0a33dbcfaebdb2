/*
 * mipt.h -- C-ABI of the MI355X path-tracing core (libmipt.so).
 *
 * This is the drop-in boundary for ONE hot path of 0x5b25/CUDAPathTracer: the per-pixel path-tracing integrator
 * behind `CUDATracer::PathTracer::Trace` (reference CUDATracer/CUDATracer.hpp:352-358, implemented by
 * CUDATracer/CUDATracerProg/CUDATracerProg.cu:859-963) together with the scene upload it consumes
 * (`CUDAScene`, CUDATracer.hpp:82-227) and the device buffer layer it is handed (`CUDABuffer`,
 * CUDATracer/CUDABuffer.hpp:10-45).  Plain pointers and sizes only; every function returns 0 on success or a
 * negative MIPT_E_* code, and never throws.  The header-only C++ classes in cudapathtracer_amd/cpp/ re-create the
 * reference's class API (same names) on top of these entry points.
 *
 * Threading: a tracer handle owns one GPU and one HIP stream.  Handles are not thread-safe; distinct handles are
 * independent (one host thread or process per GPU).  All *_dev pointers are device pointers on the tracer's GPU.
 */
#ifndef MIPT_H
#define MIPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- PODs, byte-compatible with the reference (sizes checked in tests) ---------------------------------- */
typedef struct { float position[3]; float normal[3]; float uv[2]; } mipt_vertex;        /* 32 B  Types.hpp:29-33   */
typedef struct {
    float color[3]; float roughness; float metallic; float opacity;
    float emissiveColor[3]; float emissiveStrength; float n; uint8_t translucent; uint8_t _pad[3];
} mipt_material;                                                                         /* 48 B  Types.hpp:214-225 */
typedef struct { float color[3]; float dir[3]; float intensity; } mipt_skylight;        /* 28 B  Types.hpp:227-231 */
typedef struct { float pos[3]; float front[3]; float up[3]; float right[3]; float fov; } mipt_camera; /* 52 B Types.hpp:20-27 */
typedef struct {
    mipt_camera cam; uint32_t viewportWidth, viewportHeight, frameID, seed, maxDepth;
} mipt_settings;                                                                         /* 72 B  CUDATracer.hpp:11-17 */

typedef struct {
    const mipt_vertex* vertices; uint32_t vertexCnt;   /* already transformed (Types.hpp:560-574 bakes the matrix) */
    const uint32_t*    indices;  uint32_t indicesCnt;  /* 3 per triangle */
    mipt_material      material;
} mipt_mesh_desc;

enum { MIPT_BVH_SBVH = 0,     /* spatial-split BVH (algorithm of CUDATracer/SplitBVHBuilder.cpp:57-512) */
       MIPT_BVH_MEDIAN = 1,   /* median split, leaf <= 4, depth <= 32 (CUDATracer/Types.hpp:312-413), one tree for the scene */
       MIPT_BVH_LBVH_GPU = 2 };/* built on the tracer's GPU (no counterpart in the reference, which builds on the host): Morton codes, radix sort,
                                  Karras' radix tree, boxes bottom-up, leaves of <= maxLeafSize triangles, the same 4-wide collapse and quantisation
                                  as the host builders; checked on the host (every index, the stack bound) before it is used.  Needs a tracer and
                                  >= 64 triangles; a few times faster to build and slower to trace than the SBVH: for scenes that are rebuilt often */

typedef struct {
    const mipt_mesh_desc* meshes; uint32_t meshCnt;     /* Scene::objects       Types.hpp:427-433 */
    mipt_skylight sky;                                  /* Scene::skyLight */
    const float* envMap; uint32_t envW, envH;           /* Scene::envMap / envMapDim: float RGB rows, top row = +Y; NULL = sky */
    uint32_t bvhBuilder;                                /* MIPT_BVH_* */
    uint32_t maxLeafSize;                               /* 0 = default (4); at most 15 */
    float    splitAlpha;                                /* SBVH spatial-split threshold; 0 = reference default 1e-5 */
    float    sahTriCost;                                /* SAH cost of one triangle test relative to a node visit; 0 = default (1.5 up to 64 triangles, else 3.0) */
    uint32_t nodeWidth;                                 /* 2 = fp32 BVH2 nodes; 4 = the BVH2 collapsed to 4 children per node with
                                                           child boxes quantised to 8 bits per plane (still 64 B per node, about
                                                           half the node visits); 0 = auto: 2 if the scene fits the kernels'
                                                           3 KiB LDS window (e.g. the Cornell box), else 4 */
    const char* bvhCachePath;                           /* NULL, or a file: if it holds the BVH of exactly these triangles and
                                                           build parameters it is loaded (structure re-validated) instead of
                                                           building; otherwise the BVH is built and written there.  No
                                                           counterpart in the reference (its BVH is rebuilt per run, Types.hpp:292) */
} mipt_scene_desc;

typedef struct mipt_tracer mipt_tracer;   /* replaces CUDATracerProg            CUDATracerProg.cu:925-935 */
typedef struct mipt_scene  mipt_scene;    /* replaces CUDAScene                 CUDATracer.hpp:82-227     */
typedef struct mipt_buffer mipt_buffer;   /* replaces CUDABuffer                CUDABuffer.hpp:10-45      */

enum { MIPT_OK = 0, MIPT_E_INVALID = -1, MIPT_E_HIP = -2, MIPT_E_NOMEM = -3, MIPT_E_NODEVICE = -4, MIPT_E_LIMIT = -5 };

/* RNG stream order of the integrator */
enum { MIPT_RNG_NODE = 0,   /* wavefront integrator: stream keyed by the ray's node id in the recursion tree */
       MIPT_RNG_DFS  = 1 }; /* per-lane depth-first integrator: the reference's single stream per pixel-sample, in its order */

typedef struct {
    uint64_t rays;          /* closest-hit rays traced since the last reset */
    uint64_t frames;        /* frames (samples per pixel) rendered */
    double   kernel_ms;     /* HIP-event time of the integrator kernels */
    uint64_t launches;      /* integrator kernel launches */
    uint64_t shadow_rays;   /* of `rays`: occlusion rays of estimator 1 (0 for the reference's estimator, which has none) */
    double   kernel_span_ms; /* HIP-event time from the first integrator launch's start to the last one's end, summed over the
                                batches between synchronisations: pipelined launches overlap pairwise, so this -- not
                                kernel_ms, the sum of the launches' own windows -- is the time the kernels occupied the GPU */
} mipt_counters;

/* per recursion level (0 = primary rays): rays traced, HIP-event time of that level's kernel launches, launches.
 * Filled only while option "profile_levels" is 1. */
typedef struct {
    uint64_t rays[32]; double ms[32]; uint64_t launches[32];
    double accumulate_ms;        /* accumulate + tonemap kernel */
} mipt_level_profile;

typedef struct { float t, u, v; int32_t tri; int32_t mesh; int32_t front; } mipt_hit;   /* tri = global triangle id, -1 = miss */

/* ---- ABI version (no counterpart in the reference) -------------------------------------------------------------------
 * Public structs of this header grow at their end from one version to the next (mipt_counters gained kernel_span_ms in
 * version 2) and the library writes all of the struct it was built with: a caller compiled against another version must
 * not run.  The wrappers (cudapathtracer_amd/cpp/CUDABuffer.hpp DefaultTracer, cudapathtracer_amd/api.py) compare
 * mipt_abi_version() with the MIPT_ABI_VERSION they were compiled / written against before the first call. */
#define MIPT_ABI_VERSION 3
uint32_t mipt_abi_version(void);

/* ---- tracer: PathTracer ctor/dtor (CUDATracer.cpp:8-148, CUDATracerProg.cu:925-935) ---------------------- */
int  mipt_create(int device, mipt_tracer** out);
void mipt_destroy(mipt_tracer*);
const char* mipt_last_error(const mipt_tracer*);           /* replaces CUDA_CHECK -> __debugbreak (CUDACommon.h:34-54) */
int  mipt_set_stream(mipt_tracer*, void* hip_stream);      /* NULL = the tracer's own stream */
int  mipt_synchronize(mipt_tracer*);
int  mipt_set_option(mipt_tracer*, const char* key, int64_t value);   /* "tile_pixels", "rng_mode", "block" ... */
/* Image-tile partition (SURVEY 8e's alternative to sample striding; the reference is single-GPU and has neither): the
 * frame is cut into tiles of `tile_rows` whole rows, tile k = rows [k * tile_rows, (k+1) * tile_rows); from now on
 * mipt_render / mipt_trace on this handle render and fold only the tiles part, part + n_parts, part + 2 n_parts, ...
 * and leave every other pixel of the caller's buffers untouched.  n_parts = 1 restores the whole frame.  A pixel's
 * sample does not depend on the partition (seeding is per pixel and frame, CUDATracerProg.cu:373-377).  Needs
 * integrator 0 and rng_mode node. */
int  mipt_set_tile_partition(mipt_tracer*, uint32_t tile_rows, uint32_t n_parts, uint32_t part);
int  mipt_get_counters(mipt_tracer*, mipt_counters* out);
int  mipt_reset_counters(mipt_tracer*);
int  mipt_get_level_profile(mipt_tracer*, mipt_level_profile* out);

/* ---- scene upload: CUDAScene::CUDAScene (CUDATracer.hpp:133-203) ------------------------------------------ */
int  mipt_scene_create(mipt_tracer*, const mipt_scene_desc*, mipt_scene** out);
void mipt_scene_destroy(mipt_scene*);
/* Moved vertices or edited materials on an uploaded scene: what the reference's mutable accessors
 * (CUDAScene::GetMeshMutable, CUDATracer.hpp:217-221) allow for but nothing there does -- a change needs a new
 * CUDAScene, i.e. a full BVH build.  `desc` holds the same meshes with the same index buffers (the triangle count of
 * the scene must match); positions, normals and materials are taken from it, the tree keeps its shape and its boxes
 * are re-fitted bottom-up (4-wide nodes re-quantised), and the device buffers are overwritten in place after the
 * tracer's pending frames have finished.  The sky light is taken from the desc as well; its env map and BVH parameters are ignored.  Results are those
 * of a scene created from `desc` (closest hits do not depend on the tree); traversal cost grows as the geometry drifts
 * from the one the tree was built for, and spatial-split leaves lose their clipping. */
int  mipt_scene_refit(mipt_scene*, const mipt_scene_desc* desc);
/* flattened traversal data, for inspection/tests: nodes = 16 floats each
 *   [c0.min xyz][c0.max xyz][c1.min xyz][c1.max xyz][ref0][ref1][0][0]   (ref: bit31 leaf, bits27-30 count, low 27 first; else node index)
 * refs = 12 floats each [v0 xyz, triId][e1 xyz, meshId][e2 xyz, 0].  Pass NULL buffers to query the counts. */
int  mipt_scene_get_bvh(const mipt_scene*, float* nodes, uint32_t* n_nodes, float* refs, uint32_t* n_refs);
uint32_t mipt_scene_triangle_count(const mipt_scene*);
uint32_t mipt_scene_bvh_width(const mipt_scene*);           /* 2 or 4: node layout of mipt_scene_get_bvh's `nodes` (4-wide: 16 floats =
                                                             * [origin xyz][scale.z][qlo.x qlo.y qlo.z qhi.x qhi.y qhi.z: 4 bytes each,
                                                             * one per child][scale.x][scale.y][ref0..ref3]; plane = origin + q * scale,
                                                             * scales = powers of two as floats; unused child = 0x80000000 with an
                                                             * inverted box, qlo = 255 / qhi = 0) */
int  mipt_scene_bvh_was_cached(const mipt_scene*);
/* Option "estimator" (mipt_set_option; not in the reference, which has no light sampling):
 *   0 = the reference's estimator: every radiance term arrives through a traced child ray (CUDATracerProg.cu:426-541);
 *   1 = env-map importance sampling at opaque hits above the last level: the diffuse child (:518-522) is flagged so that
 *       its own miss adds nothing; instead one direction is drawn from the environment map's distribution
 *       p_env ~ (r+g+b) * sin(theta) per texel (two more draws of the hit's stream, after the cone draws; texel by CDF
 *       inversion, uniform inside the texel) and, if it lies inside the diffuse cap (dot(dir, N) > 0.02) and one
 *       occlusion ray from the child's origin finds nothing, adds  beta * albedo * rt * L_texel * p_cap / p_env  with
 *       p_cap = 1 / (2 pi 0.98).  Emissive triangles likewise: the diffuse child skips the emission of the surface it
 *       lands on; one point y is drawn uniformly by area on the emissive triangles (three more draws: triangle by area
 *       CDF, then b1 = 1 - sqrt(r1), b2 = r2 sqrt(r1)) and, if its direction lies inside the cap and an occlusion ray
 *       stopping at 0.9999 of the distance finds nothing, adds  beta * albedo * rt * Le * |cos_y| / dist^2 * A_total * p_cap
 *       (emission is two-sided in the reference, CUDATracerProg.cu:528, hence |cos|).  Same expectation as estimator 0,
 *       far less variance under a small bright sun or lamp.  Occlusion rays are counted in mipt_counters.shadow_rays
 *       (and in .rays).  Acts on whichever of env map / emissive triangles the scene has (error if neither); needs
 *       integrator 0 and rng_mode node.
 *   2 = the same two lobes sampled the textbook way (a different image: the lobes themselves change, not only their
 *       variance).  Specular: the microfacet normal m is drawn from the GGX distribution of visible normals (Heitz 2018)
 *       with alpha = roughness, seen from V = -d in the frame (up, right, N) of the reference's cone basis
 *       (CUDATracerProg.cu:99-104): Ve = (V.up, V.right, max(V.N, 1e-4)); Vh = normalize(alpha Ve.x, alpha Ve.y, Ve.z);
 *       T1 = (-Vh.y, Vh.x, 0) / |.| (or (1,0,0)), T2 = Vh x T1; r = sqrt(u1), phi = 2 pi u2 (double product as :112),
 *       t1 = r cos phi, t2 = (1 - s) sqrt(1 - t1^2) + s r sin phi with s = (1 + Vh.z) / 2;
 *       Nh = t1 T1 + t2 T2 + sqrt(max(0, 1 - t1^2 - t2^2)) Vh; m = normalize(alpha Nh.x, alpha Nh.y, max(1e-6, Nh.z)) back
 *       in world space; roughness 0 gives m = N exactly (two draws are consumed either way).  The mirror child's weight
 *       (:463-493, unchanged Fresnel with m) is multiplied by the Smith masking term of the reflected direction,
 *       G1 = 2 nl / (nl + sqrt(alpha^2 + (1 - alpha^2) nl^2)), nl = R.N, 0 when nl <= 0 (the child is still traced).
 *       Diffuse: cosine-weighted about N (r = sqrt(v), dir = sqrt(1 - v) N + r cos(theta) up + r sin(theta) right, theta
 *       = 2 pi u), weight albedo * rt unchanged -- a Lambertian lobe instead of the reference's uniform cap (:518-526).
 *       Refraction unchanged.  No occlusion rays; needs integrator 0 and rng_mode node.
 *   3 = the lobes of 2 with the light sampling of 1: the cosine-weighted diffuse child skips the environment / the
 *       emission of the surface it lands on, and the explicitly sampled terms of estimator 1 are accepted over the
 *       whole hemisphere (dot(dir, N) > 0) and multiplied by  (cos / pi) / p_cap = dot(dir, N) * 1.96  -- the lobe's
 *       density in place of the cap's.  Same expectation as estimator 2.
 * Sampling tables of estimator 1 (rowCdf[envH+1], then envH rows of colCdf[envW+1]); returns the float count,
 * 0 without an env map.  No counterpart in the reference (it has no light sampling). */
uint64_t mipt_scene_get_env_cdf(const mipt_scene*, float* out, uint64_t cap);
/* emitter table of estimator 1: n x 15 floats [v0 e1 e2 unit-normal Le] and the area CDF [n+1]; returns n */
uint64_t mipt_scene_get_emitters(const mipt_scene*, float* tris, float* cdf, uint64_t cap_tris);
/* "Emitters first" (no counterpart in the reference, which traces every ray to its closest hit, CUDATracerProg.cu:543-794).  A ray
 * at the last level can only add emission (or the sky): in a scene where some meshes emit and some do not, the wave-stack
 * integrator (estimators 0 and 2) intersects such a ray with a small second tree over the emitting triangles first, then walks the
 * main tree with that distance as its bound and stops at the first hit on anything that does not emit; under a black sky a ray that
 * found no emitter is finished at once.  Same image and the same per-pixel ray counts (the ray is still intersected with everything
 * that can change its contribution).  The second tree lies behind the main tree in the device arrays, in the same node layout; this
 * returns it as it lies there (node indices and leaf ranges continue the main tree's: concatenate with mipt_scene_get_bvh's arrays)
 * with its root reference -- an inner node index, or a leaf reference when the emitters fit one leaf; *root_ref = 0: the scene has
 * none (nothing or everything emits, the tree would not fit the traversal stack / the LDS window the main tree fits, or -- under a
 * lit sky, where the ray still has to be walked through the scene -- the emitters are so few (<= 15 triangles) that one more leaf
 * per ray costs more than stopping at the first hit saves).  $MIPT_EMITTER_TREE=0 builds none, =1 builds it whenever it fits. */
int mipt_scene_get_emitter_tree(const mipt_scene*, float* nodes, uint32_t* n_nodes, float* refs, uint32_t* n_refs, uint32_t* root_ref);

/* ---- the hot path: PathTracer::Trace (CUDATracer.hpp:352-358; CUDATracerProg.cu:937-963) -------------------
 * One sample per pixel for sample index settings->frameID, folded into acc_dev (w*h*4 floats, running mean with
 * weight 1/(frameID+1), alpha 1) and tonemapped into ldr_dev (w*h*4 bytes, may be NULL).  Row 0 = bottom.
 * Asynchronous on the tracer's stream, like the reference's kernel launch. */
int  mipt_trace(mipt_tracer*, mipt_scene*, const mipt_settings*, float* acc_dev, unsigned char* ldr_dev);
/* batched form of the same loop (RendererWindow::render called n_frames times, Common/RendererWindow.hpp:128-148):
 * frames first_frame .. first_frame+n_frames-1; running mean unless sum_mode != 0 (then acc += sample, for
 * multi-GPU sample striding where the caller reduces and scales). frame_stride lets rank r render r, r+G, ...
 * Asynchronous like mipt_trace: acc_dev / ldr_dev are complete once mipt_synchronize (or any other entry point on the
 * tracer) has returned.  In sum mode without an LDR target the integrator adds into a block buffer of the tracer's that is
 * folded into acc_dev every 64 frames and at that point at the latest, so that no contribution is ever added to a sum of
 * more than 64 frames (a sum of thousands of frames swallows the faint ones: measured -1e-4 of the image mean after 8192
 * frames when every ray added straight into acc_dev). */
int  mipt_render(mipt_tracer*, mipt_scene*, const mipt_settings*, uint32_t first_frame, uint32_t n_frames,
                 uint32_t frame_stride, int sum_mode, float* acc_dev, unsigned char* ldr_dev);
/* mipt_trace for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same
 * contract, same kernels and same result as mipt_trace (radiance up to the order of the fp32 adds into a pixel), but
 * every work buffer (pixel counter, sample buffer, 256 wave stacks) lives in a 32 MiB file-scope device array that each
 * GPU the library runs on holds until the process ends; the call makes no hipMalloc / hipFree / event creation and
 * queues everything on the tracer's stream.  Limits (MIPT_E_LIMIT beyond them): viewportWidth * viewportHeight <= 65536,
 * maxDepth <= 8; needs integrator 0, rng_mode node, estimator 0 or 2, no tile partition, count_pixel_rays off.  All
 * calls on one GPU share the array (what must start at zero is cleared per call): calls that overlap on one GPU --
 * two tracers, or a caller-supplied stream running beside another -- race.  Under stream capture: hand the capturing
 * stream over with mipt_set_stream first (it joins whatever mipt_render has in flight before it switches); the call then
 * records two memsets and two kernels into the capture and nothing else (tests/test_zz_no_alloc.py replays such a graph). */
int  mipt_trace_static(mipt_tracer*, mipt_scene*, const mipt_settings*, float* acc_dev, unsigned char* ldr_dev);
/* accumulate-free tonemap of an accumulation buffer (scale applied first): CUDATracerProg.cu:908-919 */
int  mipt_tonemap(mipt_tracer*, const float* acc_dev, float scale, uint32_t n_pixels, unsigned char* ldr_dev);
/* radiance of the last frame traced, before accumulation (w*h*3 floats) and its per-pixel ray counts; for parity tests.
 * A pipelined frame's fold clears its sample buffer (and a sum-mode frame without an LDR target has none): set option
 * "keep_sample" = 1 (or "count_pixel_rays" = 1, or "pipeline_frames" = 0) before rendering, else MIPT_E_INVALID. */
int  mipt_last_sample(mipt_tracer*, float* rgb_host, uint32_t* rays_host, uint32_t n_pixels);

/* closest hit for a fixed ray batch (the traversal kernel alone): TraceRay, CUDATracerProg.cu:543-794 */
int  mipt_trace_rays(mipt_tracer*, mipt_scene*, const float* origins_dev, const float* dirs_dev, uint32_t n,
                     mipt_hit* hits_dev);

/* ---- device buffer layer: CUDABuffer (CUDABuffer.hpp:10-45, CUDABuffer.cpp:111-224) ------------------------ */
enum { MIPT_HEAD_UNINITIALIZED = 0, MIPT_HEAD_AT_CPU = 1, MIPT_HEAD_AT_GPU = 2, MIPT_HEAD_SYNCED = 3 };
int   mipt_buffer_create(mipt_tracer*, size_t size, mipt_buffer** out);
void  mipt_buffer_destroy(mipt_buffer*);
const void* mipt_buffer_cpu_data(mipt_buffer*);
const void* mipt_buffer_gpu_data(mipt_buffer*);
void* mipt_buffer_mutable_cpu_data(mipt_buffer*);
void* mipt_buffer_mutable_gpu_data(mipt_buffer*);
int   mipt_buffer_head(const mipt_buffer*);
size_t mipt_buffer_size(const mipt_buffer*);
int   mipt_buffer_async_gpu_push(mipt_buffer*);           /* CUDABuffer::async_gpu_push on the tracer's stream */
/* CUDABuffer::async_gpu_push(const cudaStream_t&) (CUDABuffer.hpp:27, CUDABuffer.cpp:208-224): head must be at the CPU;
 * the host-to-device copy is queued on the caller's HIP stream (NULL = the tracer's) and the head becomes SYNCED; the
 * caller synchronises on that stream before the device side is used. */
int   mipt_buffer_async_gpu_push_on(mipt_buffer*, void* hip_stream);
/* CUDABuffer::set_cpu_data / set_gpu_data (CUDABuffer.hpp:18,20; CUDABuffer.cpp:166-191): the buffer adopts caller-owned
 * memory of at least size() bytes (pinned or pageable host memory / device memory on the tracer's GPU) without taking
 * ownership; what the buffer had allocated on that side is freed; the head moves to that side. */
int   mipt_buffer_set_cpu_data(mipt_buffer*, void* data);
int   mipt_buffer_set_gpu_data(mipt_buffer*, void* data_dev);
/* raw helpers for callers that manage device memory themselves */
int   mipt_malloc(mipt_tracer*, size_t size, void** dev_out);
int   mipt_free(mipt_tracer*, void* dev);
int   mipt_upload(mipt_tracer*, void* dev, const void* host, size_t size);
int   mipt_download(mipt_tracer*, void* host, const void* dev, size_t size);
int   mipt_memset(mipt_tracer*, void* dev, int value, size_t size);

/* ---- callers' scene construction (Types.hpp:435-574; CornellBox/tracer.cpp:16-158; DefaultScene/tracer.cpp:28-123) */
/* kind 0 quad 1 cube 2 sphere; xform = 16 floats row-major or NULL; returns vertex count, fills up to the caps */
uint32_t mipt_make_mesh(int kind, const float* xform, mipt_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap, uint32_t* icnt);
/* built-in scenes: 0 Cornell box, 1 DefaultScene */
uint32_t mipt_builtin_mesh_count(int which);
uint32_t mipt_builtin_mesh(int which, uint32_t mesh, mipt_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap,
                           uint32_t* icnt, mipt_material* mat);
void     mipt_builtin_sky(int which, mipt_skylight* sky);
/* camera basis from (pos, azimuth, zenith, fov): RendererWindow::UpdateCam, Common/RendererWindow.hpp:15-37 */
void     mipt_make_camera(const float pos[3], float azimuth, float zenith, float fov, mipt_camera* out);
/* synthetic HDRI stand-in for the missing DefaultScene/sky.hdr (sky gradient + sun disc + value-noise ground) */
void     mipt_make_synthetic_sky(uint32_t w, uint32_t h, uint32_t seed, float* rgb_out);
/* synthetic displaced-grid stress mesh (n x n quads, heights = 4-octave value noise): returns vertex count */
uint32_t mipt_make_grid_mesh(uint32_t n, float extent, float height, uint32_t seed, mipt_vertex* v, uint32_t* idx);

/* Radiance RGBE (.hdr): the env-map file format on the input side of the path (Common/rgbe.c:93-106,134-190,327-412;
 * read at DefaultScene/tracer.cpp:130-138).  decode: pass rgb_out NULL to query the size; returns 0 or MIPT_E_INVALID.
 * encode: returns the byte count (call with out NULL to size the buffer). */
int      mipt_rgbe_decode(const unsigned char* data, size_t size, float* rgb_out, uint32_t* width, uint32_t* height);
size_t   mipt_rgbe_encode(const float* rgb, uint32_t width, uint32_t height, unsigned char* out, size_t cap);

/* ---- device math self-test (used by the parity tests of the portable transcendentals) --------------------- */
/* fn 0 sin 1 cos 2 atan2(a,b) 3 acos 4 x^(1/2.2) 5 x^5 6 x^10 ; host arrays */
int  mipt_test_math(mipt_tracer*, int fn, const float* a, const float* b, uint32_t n, float* out);

/* ---- wave-stack capacity test support (no counterpart in the reference, which recurses on an 8 KiB local stack,
 * CUDATracerProg.cu:928) ------------------------------------------------------------------------------------------------
 * Option "wave_stack_entries" = k > 0 makes the integrator's waves use only k entries each of their pending-ray stacks
 * (default 256 * maxDepth + 256; the arrays keep their size and are painted before every mipt_render).  A wave that runs
 * out of room stalls its shade phase and renders the same image, slower; only a stack that cannot even hold one record's
 * children loses rays, and then mipt_synchronize returns MIPT_E_LIMIT.  mipt_test_stack_guard: out[0] = bytes of the
 * arrays beyond the waves' regions, out[1] = how many of those bytes were overwritten (always 0) -- both 0 without the
 * option --, out[2] = shade batches cut short or put off for lack of room since the last mipt_reset_counters. */
int  mipt_test_stack_guard(mipt_tracer*, uint64_t out[3]);

/* ---- VALU issue calibration (no counterpart in the reference; evidence for bench.py's VALU roofline) --------------
 * Launches, on the integrator's grid shape (256-thread blocks, CUs x blocks_per_cu blocks = blocks_per_cu waves per
 * SIMD), a kernel in which every wave issues iters x 64 independent v_fma_f32, and reports the HIP-event time of the
 * launch and the median over waves of the loop's duration in shader-clock cycles (s_memtime) and 100 MHz ticks
 * (s_memrealtime).  wave-instructions per SIMD-cycle = blocks_per_cu * iters * 64 / median_cycles. */
typedef struct {
    double ms; uint64_t wave_instructions; uint32_t waves, simds;
    double median_cycles, max_cycles, median_realtime_ticks;
    double clock_ghz;            /* median over waves of cycles / (ticks * 10 ns): the clock the chip held during the loop */
} mipt_issue_result;
/* packed 0: v_fma_f32; 1: v_pk_fma_f32 (two FMAs per lane per instruction) */
/* Exhaustive check of the integrator's short reciprocal (rcp + `steps` Newton corrections in FMA form, used for a ray's
 * inverse direction and the triangle test's 1 / det) against IEEE division over every float of magnitude 2^-100 .. 2^100:
 * out[0] = values compared, out[1] = values whose bits differ, out[2] = first differing input (bits), out[3] = differing
 * by more than one unit in the last place.  steps = 4 runs the same comparison for the sampler's short square root (the IEEE
 * expansion without the rescaling that only inputs below 2^-96 need) against sqrtf, over zero and every positive float in
 * 2^-96 .. 2^100.  (No counterpart in the reference.) */
int  mipt_test_rcp_sweep(mipt_tracer*, int steps, uint64_t out[4]);
int  mipt_test_issue(mipt_tracer*, int packed, uint32_t blocks_per_cu, uint32_t iters, mipt_issue_result* out);

#ifdef __cplusplus
}
#endif
#endif
