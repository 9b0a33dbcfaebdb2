/*
 * oracle.h -- C-ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.so, and there only as the checker / reported CPU baseline.  The
 * product library (cudapathtracer_amd/libmipt.so) never includes, links or
 * calls anything declared here.
 *
 * PARITY STATUS: pinned coarsely, by images only.  The reference
 * (0x5b25/CUDAPathTracer) ships no tests, golden vectors or fixtures for this
 * path and cannot be compiled in this image
 * (MSVC + CUDA 10.2 only).  The oracle is a scalar restatement of
 * CUDATracer/CUDATracerProg/CUDATracerProg.cu.  What the reference does hold
 * are its two README screenshots; the oracle is checked against both
 * (tests/test_oracle_render.py: Cornell box block by block in brightness,
 * colour and structure; Default scene in geometry and hues through a fitted
 * camera).  No bit-level vector of the reference exists: below the resolution
 * of those images parity is unpinned, and the oracle rests on the known-answer
 * RNG vectors in BASELINE.md section 3 and analytic / brute-force self checks.
 *
 * All structs mirror the reference PODs byte-for-byte
 * (CUDATracer/Types.hpp:20-39,214-231; CUDATracer/CUDATracer.hpp:11-17).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float position[3]; float normal[3]; float uv[2]; } orc_vertex;          /* 32 B, Types.hpp:29-33 */
typedef struct {
    float color[3]; float roughness; float metallic; float opacity;
    float emissiveColor[3]; float emissiveStrength; float n; uint8_t translucent; uint8_t _pad[3];
} orc_material;                                                                           /* 48 B, Types.hpp:214-225 */
typedef struct { float color[3]; float dir[3]; float intensity; } orc_skylight;          /* 28 B, Types.hpp:227-231 */
typedef struct { float pos[3]; float front[3]; float up[3]; float right[3]; float fov; } orc_camera; /* 52 B, Types.hpp:20-27 */
typedef struct {
    orc_camera cam; uint32_t viewportWidth, viewportHeight, frameID, seed, maxDepth;
} orc_settings;                                                                           /* 72 B, CUDATracer.hpp:11-17 */

typedef struct {
    const orc_vertex* vertices; uint32_t vertexCnt;
    const uint32_t*   indices;  uint32_t indicesCnt;
    orc_material      material;
} orc_mesh_desc;

typedef struct {
    const orc_mesh_desc* meshes; uint32_t meshCnt;
    orc_skylight sky;
    const float* envMap; uint32_t envW, envH;   /* float RGB, row-major, NULL = analytic sky */
} orc_scene_desc;

typedef struct orc_scene orc_scene;

/* which intersector answers closest-hit queries */
enum { ORC_ISECT_REF = 0,    /* reference semantics: per-mesh median BVH2, no t-culling (CUDATracerProg.cu:543-794) */
       ORC_ISECT_FLAT = 1,   /* the product's flattened BVH (set with orc_scene_set_flat_bvh), product traversal order */
       ORC_ISECT_BRUTE = 2 };/* every triangle, mesh order then triangle order */
/* RNG stream order */
enum { ORC_RNG_DFS = 0,      /* one LCG per pixel-sample threaded through the recursion (reference-faithful) */
       ORC_RNG_NODE = 1 };   /* stream keyed by the ray's node id in the binary tree (what a wavefront can reproduce) */
/* transcendental functions */
enum { ORC_MATH_LIBM = 0,    /* glibc sinf/cosf/atan2f/acosf/powf: the literal restatement */
       ORC_MATH_DET = 1 };   /* portable polynomial versions, bit-reproducible on CPU and GPU */

typedef struct {
    uint64_t rays;        /* closest-hit rays traced */
    uint64_t node_visits; /* inner-node visits (FLAT only) */
    uint64_t tri_tests;   /* triangle tests */
    uint64_t hits;        /* rays that hit something */
    uint64_t shadow_rays; /* of `rays`: occlusion rays of the env-map importance sampling estimator (estimator 1) */
    uint64_t anyhit_rays; /* of `rays`: last-level rays traced under orc_scene_set_anyhit_last_level's hit-or-miss rule */
    uint64_t emitfirst_rays; /* of `rays`: last-level rays traced under the "emitters first" rule (orc_scene_set_emitter_tree) */
} orc_stats;

typedef struct { float t, u, v; int32_t tri; int32_t mesh; int32_t front; } orc_hit; /* tri = global triangle id, -1 = miss */

/* --- RNG known answers (CUDATracerProg.cu:40-73) --- */
uint32_t orc_tea16(uint32_t v0, uint32_t v1);
void     orc_lcg_floats(uint32_t state, uint32_t n, float* out);
uint32_t orc_node_seed(uint32_t s0, uint32_t node);

/* --- deterministic math, for unit tests: fn 0 sin 1 cos 2 atan2(a,b) 3 acos 4 x^(1/2.2) 5 x^5 6 x^10 --- */
void orc_math_eval(int fn, int math_mode, const float* a, const float* b, uint32_t n, float* out);

/* --- procedural geometry & scenes (Types.hpp:435-574, CornellBox/tracer.cpp:16-143, DefaultScene/tracer.cpp:28-123) --- */
/* kind 0 quad 1 cube 2 sphere; xform = 16 floats row-major; returns vertex count, writes up to the caps given */
uint32_t orc_make_mesh(int kind, const float* xform, orc_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap, uint32_t* icnt);
/* which 0 cornell 1 default: number of meshes; per-mesh fill-in through orc_builtin_mesh */
uint32_t orc_builtin_scene_mesh_count(int which);
uint32_t orc_builtin_mesh(int which, uint32_t mesh, orc_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap,
                          uint32_t* icnt, orc_material* mat);
void     orc_builtin_sky(int which, orc_skylight* sky);
/* camera basis from (pos, azimuth, zenith, fov) -- Common/RendererWindow.hpp:15-37 */
void     orc_make_camera(const float pos[3], float azimuth, float zenith, float fov, orc_camera* out);

/* --- scene --- */
orc_scene* orc_scene_create(const orc_scene_desc* desc);
void       orc_scene_destroy(orc_scene*);
uint32_t   orc_scene_triangle_count(const orc_scene*);
/* product BVH: nodes = n_nodes*16 floats, tris = n_refs*12 floats (layout: include/mipt.h) */
int        orc_scene_set_flat_bvh(orc_scene*, const float* nodes, uint32_t n_nodes, const float* tris, uint32_t n_refs);
/* node layout of the flat BVH: 2 = fp32 BVH2 nodes (default), 4 = quantised 4-wide nodes (include/mipt.h) */
int        orc_scene_set_flat_width(orc_scene*, uint32_t width);
/* structural check of a flat BVH against the scene's triangles; returns 0 if valid, else an error code (message in buf) */
int        orc_check_flat_bvh(const orc_scene*, char* msg, uint32_t msgcap);

/* --- closest hit on fixed ray batches (CUDATracerProg.cu:185-303,543-794) --- */
int orc_trace_rays(const orc_scene*, int isect, const float* origins, const float* dirs, uint32_t n,
                   orc_hit* out, orc_stats* stats, int nthreads);

/* --- integrator (CUDATracerProg.cu:367-541,796-922) ---
 * Renders frames [first_frame, first_frame+n_frames) of the rectangle [x0,x1)x[y0,y1) into acc (running mean,
 * w*h*4 floats, row 0 = bottom) and ldr (w*h*4 bytes, may be NULL).  sample (w*h*3 floats, may be NULL) receives the
 * radiance of the LAST frame rendered; rays_per_pixel (w*h uint32, may be NULL) its per-pixel ray count. */
/* Estimator (not in the reference, which has no light sampling -- SURVEY 8f rank 4; the product's option "estimator"):
 *   0 = the reference's: every radiance term arrives through a traced child ray;
 *   1 = env-map importance sampling at opaque hits: the diffuse child no longer collects the environment when it
 *       misses; instead one direction is drawn from the environment map's luminance x sin(theta) distribution
 *       (two extra draws of the hit's stream, after the cone draws) and, if it lies inside the diffuse cap
 *       (cos > 0.02) and is unoccluded, adds  albedo * rt * L * p_cap / p_env.  Likewise for emissive triangles:
 *       the diffuse child skips the emission of the surface it lands on, and one point is drawn uniformly by area on
 *       the emissive triangles (three more draws) and connected with an occlusion ray.  Same expectation as estimator 0.
 * Acts on whichever of env map / emissive triangles the scene has; needs RNG_NODE; ignored otherwise. */
int orc_scene_set_estimator(orc_scene*, int estimator);
/* work-counting option for the FLAT intersector: last-level rays stop at their first hit when no mesh emits (what the product
 * does; same image, fewer node visits / triangle tests).  Off by default: every other mode is the reference's closest hit. */
int orc_scene_set_anyhit_last_level(orc_scene*, int on);
/* Work-counting option, with orc_scene_set_anyhit_last_level on (FLAT intersector, estimators 0 and 2): restates the product's
 * "emitters first" rule for last-level rays in a scene with emitters (include/mipt.h mipt_scene_get_emitter_tree; kernels.hip phase D).
 * The flat arrays given to orc_scene_set_flat_bvh* must hold the emitters' tree behind the main tree; root_ref = its root reference
 * (0 = off), n_refs_main = the main tree's reference count.  Same image bit for bit; fewer node visits / triangle tests. */
int orc_scene_set_emitter_tree(orc_scene*, uint32_t root_ref, uint32_t n_refs_main);
/* the sampling tables of estimator 1: rowCdf[H+1] then H rows of colCdf[W+1]; returns the float count (0 = no env map) */
uint64_t orc_scene_get_env_cdf(const orc_scene*, float* out, uint64_t cap);

/* emitter table of estimator 1: n x 15 floats [v0 e1 e2 unit-normal Le] and the area CDF [n+1]; returns n */
uint64_t orc_scene_get_emitters(const orc_scene*, float* tris, float* cdf, uint64_t cap_tris);

int orc_render(const orc_scene*, const orc_settings* s, int isect, int rng_mode, int math_mode,
               uint32_t first_frame, uint32_t n_frames, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
               float* acc, uint8_t* ldr, float* sample, uint32_t* rays_per_pixel, orc_stats* stats, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
