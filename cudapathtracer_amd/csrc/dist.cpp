// dist.cpp -- single-process multi-GPU host (include/mipt_dist.h): one mipt_tracer + one host thread per GPU, sample
// indices strided over the GPUs, one RCCL ncclReduce of the sum buffers, tonemap on the root.  Plain C++ over the
// C-ABI of libmipt plus the HIP runtime (buffers, peer-less: every rank only touches its own GPU) and RCCL.
// The reference has no multi-GPU code (device 0 only: OptixTracerProg.cpp:220-229); SURVEY 8e is the specification.
#include "../../include/mipt_dist.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

struct mipt_dist {
    int n = 0;
    std::vector<mipt_tracer*> tracer;
    std::vector<mipt_scene*> scene;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;          // one per GPU for the reduce, created after the tracers
    std::vector<float*> acc;                  // per-GPU sum buffers
    unsigned char* ldr = nullptr;             // root only
    size_t pixels = 0;
    std::string err;
    mipt_dist_stats stats{};
};

namespace {
int fail(mipt_dist* d, int code, const std::string& msg) { try { if (d) d->err = msg; } catch (...) {} return code; }
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// include/mipt_dist.h inherits mipt.h's rule that no entry point throws: every extern "C" body that can allocate on the
// host runs inside this guard (the same shape as api.hip's)
template <typename F> int guarded(mipt_dist* d, F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { return fail(d, MIPT_E_NOMEM, "out of host memory"); }
    catch (const std::exception& e) { try { if (d) d->err = std::string("exception: ") + e.what(); } catch (...) {} return MIPT_E_INVALID; }
    catch (...) { return fail(d, MIPT_E_INVALID, "unknown exception"); }
}

// runs f(rank) on one host thread per GPU and returns the first non-zero status.  A thread that cannot be started (or a
// vector that cannot grow) must not leave joinable threads behind -- their destructors would call std::terminate -- so
// the ones already running are joined before the failure is reported.
template <typename F> int per_gpu(mipt_dist* d, F f) {
    std::vector<int> rc(d->n, 0);
    std::vector<std::thread> th;
    bool spawned = true;
    try {
        th.reserve(d->n);
        for (int r = 1; r < d->n; r++) th.emplace_back([&, r]() { try { rc[r] = f(r); } catch (...) { rc[r] = MIPT_E_NOMEM; } });
    } catch (...) { spawned = false; }
    if (spawned) { try { rc[0] = f(0); } catch (...) { rc[0] = MIPT_E_NOMEM; } }
    for (auto& t : th) if (t.joinable()) t.join();
    if (!spawned) return fail(d, MIPT_E_NOMEM, "could not start one host thread per GPU");
    for (int r = 0; r < d->n; r++) if (rc[r]) {
        const char* why = mipt_last_error(d->tracer[r]);
        try { d->err = "rank " + std::to_string(r) + ": " + why; } catch (...) {}
        return rc[r];
    }
    return MIPT_OK;
}
void free_buffers(mipt_dist* d) {
    for (int r = 0; r < d->n; r++) if (r < (int)d->acc.size() && d->acc[r]) { hipSetDevice(r); hipFree(d->acc[r]); d->acc[r] = nullptr; }
    if (d->ldr) { hipSetDevice(0); hipFree(d->ldr); d->ldr = nullptr; }
    d->pixels = 0;
}
}  // namespace

extern "C" {

int mipt_dist_frames_for_rank(uint32_t first_frame, uint32_t n_frames, int rank, int world, uint32_t* first, uint32_t* count) {
    if (world < 1 || rank < 0 || rank >= world || !first || !count) return MIPT_E_INVALID;
    *first = first_frame + (uint32_t)rank;
    *count = n_frames > (uint32_t)rank ? (n_frames - (uint32_t)rank + (uint32_t)world - 1) / (uint32_t)world : 0;
    return MIPT_OK;
}

int mipt_dist_create(int n_gpus, mipt_dist** out) {
    if (!out) return MIPT_E_INVALID;
    *out = nullptr;
    return guarded(nullptr, [&]() -> int {
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return MIPT_E_NODEVICE;
    if (n_gpus == 0) n_gpus = visible;
    if (n_gpus < 1 || n_gpus > visible) return MIPT_E_INVALID;
    mipt_dist* d = nullptr;
    try {
        d = new mipt_dist();
        d->n = n_gpus;
        d->tracer.assign(n_gpus, nullptr); d->scene.assign(n_gpus, nullptr); d->comm.assign(n_gpus, nullptr);
        d->stream.assign(n_gpus, nullptr); d->acc.assign(n_gpus, nullptr);
    } catch (...) { delete d; return MIPT_E_NOMEM; }
    // tracers first: each takes the hardware queues of its frame pipeline before RCCL creates its own streams
    for (int r = 0; r < n_gpus; r++) {
        int rc = mipt_create(r, &d->tracer[r]);
        if (rc) { mipt_dist_destroy(d); return rc; }
    }
    std::vector<int> devs;
    try { devs.resize(n_gpus); } catch (...) { mipt_dist_destroy(d); return MIPT_E_NOMEM; }
    for (int r = 0; r < n_gpus; r++) devs[r] = r;
    if (ncclCommInitAll(d->comm.data(), n_gpus, devs.data()) != ncclSuccess) { mipt_dist_destroy(d); return MIPT_E_HIP; }
    for (int r = 0; r < n_gpus; r++) {
        if (hipSetDevice(r) != hipSuccess || hipStreamCreateWithFlags(&d->stream[r], hipStreamNonBlocking) != hipSuccess) { mipt_dist_destroy(d); return MIPT_E_HIP; }
    }
    *out = d;
    return MIPT_OK;
    });
}

void mipt_dist_destroy(mipt_dist* d) {
    if (!d) return;
    for (int r = 0; r < d->n; r++) if (d->tracer[r]) mipt_synchronize(d->tracer[r]);
    free_buffers(d);
    for (int r = 0; r < d->n; r++) {
        if (d->stream[r]) { hipSetDevice(r); hipStreamSynchronize(d->stream[r]); hipStreamDestroy(d->stream[r]); }
        if (d->comm[r]) ncclCommDestroy(d->comm[r]);
    }
    for (int r = 0; r < d->n; r++) { if (d->scene[r]) mipt_scene_destroy(d->scene[r]); if (d->tracer[r]) mipt_destroy(d->tracer[r]); }
    delete d;
}

int mipt_dist_gpus(const mipt_dist* d) { return d ? d->n : 0; }
const char* mipt_dist_last_error(const mipt_dist* d) { return d ? d->err.c_str() : "null handle"; }
mipt_tracer* mipt_dist_tracer(mipt_dist* d, int rank) { return d && rank >= 0 && rank < d->n ? d->tracer[rank] : nullptr; }

int mipt_dist_scene_create(mipt_dist* d, const mipt_scene_desc* desc) {
    if (!d || !desc) return MIPT_E_INVALID;
    return guarded(d, [&]() -> int {
    for (int r = 0; r < d->n; r++) if (d->scene[r]) { mipt_synchronize(d->tracer[r]); mipt_scene_destroy(d->scene[r]); d->scene[r] = nullptr; }
    const bool cache = desc->bvhCachePath && desc->bvhCachePath[0];
    if (cache) {       // rank 0 builds and writes the file; the others load it
        int rc = mipt_scene_create(d->tracer[0], desc, &d->scene[0]);
        if (rc) return fail(d, rc, std::string("rank 0: ") + mipt_last_error(d->tracer[0]));
    }
    return per_gpu(d, [&](int r) { return d->scene[r] ? MIPT_OK : mipt_scene_create(d->tracer[r], desc, &d->scene[r]); });
    });
}

int mipt_dist_render(mipt_dist* d, const mipt_settings* st, uint32_t first_frame, uint32_t n_frames, float* acc_mean_host, unsigned char* ldr_host) {
    if (!d || !st) return MIPT_E_INVALID;
    return guarded(d, [&]() -> int {
    if (!d->scene[0]) return fail(d, MIPT_E_INVALID, "no scene: call mipt_dist_scene_create first");
    const size_t pixels = (size_t)st->viewportWidth * st->viewportHeight;
    if (!pixels || !n_frames) return fail(d, MIPT_E_INVALID, "empty viewport or no frames");
    if (d->pixels != pixels) {
        free_buffers(d);
        for (int r = 0; r < d->n; r++) {
            if (hipSetDevice(r) != hipSuccess || hipMalloc((void**)&d->acc[r], pixels * 16) != hipSuccess) { free_buffers(d); return fail(d, MIPT_E_NOMEM, "accumulation buffers"); }
        }
        if (hipSetDevice(0) != hipSuccess || hipMalloc((void**)&d->ldr, pixels * 4) != hipSuccess) { free_buffers(d); return fail(d, MIPT_E_NOMEM, "LDR buffer"); }
        d->pixels = pixels;
    }
    // ---- render: every rank its own frames into its own sum buffer, no communication
    const double t0 = now();
    std::vector<mipt_counters> before(d->n), after(d->n);
    int rc = per_gpu(d, [&](int r) {
        uint32_t first = 0, count = 0;
        mipt_dist_frames_for_rank(first_frame, n_frames, r, d->n, &first, &count);
        int e = mipt_get_counters(d->tracer[r], &before[r]); if (e) return e;
        e = mipt_memset(d->tracer[r], d->acc[r], 0, pixels * 16); if (e) return e;
        if (count) { e = mipt_render(d->tracer[r], d->scene[r], st, first, count, (uint32_t)d->n, 1, d->acc[r], nullptr); if (e) return e; }
        e = mipt_synchronize(d->tracer[r]); if (e) return e;            // the reduce below runs on another stream
        return mipt_get_counters(d->tracer[r], &after[r]);
    });
    if (rc) return rc;
    const double t1 = now();
    // ---- the path's one exchange step: sum the buffers onto the root over xGMI
    if (d->n > 1) {
        if (ncclGroupStart() != ncclSuccess) return fail(d, MIPT_E_HIP, "ncclGroupStart");
        for (int r = 0; r < d->n; r++)
            if (ncclReduce(d->acc[r], d->acc[r], pixels * 4, ncclFloat, ncclSum, 0, d->comm[r], d->stream[r]) != ncclSuccess) { ncclGroupEnd(); return fail(d, MIPT_E_HIP, "ncclReduce"); }
        if (ncclGroupEnd() != ncclSuccess) return fail(d, MIPT_E_HIP, "ncclGroupEnd");
        for (int r = 0; r < d->n; r++) { hipSetDevice(r); if (hipStreamSynchronize(d->stream[r]) != hipSuccess) return fail(d, MIPT_E_HIP, "reduce stream"); }
    }
    // ---- root: scale by 1 / N inside the tonemap, read back
    const float scale = (float)(1.0 / (double)n_frames);
    rc = mipt_tonemap(d->tracer[0], d->acc[0], scale, (uint32_t)pixels, d->ldr);
    if (!rc && ldr_host) rc = mipt_download(d->tracer[0], ldr_host, d->ldr, pixels * 4);
    if (!rc && acc_mean_host) {
        rc = mipt_download(d->tracer[0], acc_mean_host, d->acc[0], pixels * 16);
        if (!rc) for (size_t i = 0; i < pixels; i++) { acc_mean_host[i * 4] *= scale; acc_mean_host[i * 4 + 1] *= scale; acc_mean_host[i * 4 + 2] *= scale; acc_mean_host[i * 4 + 3] = 1.f; }
    }
    if (!rc) rc = mipt_synchronize(d->tracer[0]);
    if (rc) return fail(d, rc, std::string("root: ") + mipt_last_error(d->tracer[0]));
    d->stats = mipt_dist_stats{};
    for (int r = 0; r < d->n; r++) d->stats.rays += after[r].rays - before[r].rays;
    d->stats.render_s = t1 - t0; d->stats.reduce_s = now() - t1; d->stats.frames = n_frames; d->stats.gpus = d->n;
    return MIPT_OK;
    });
}

int mipt_dist_get_stats(const mipt_dist* d, mipt_dist_stats* out) { if (!d || !out) return MIPT_E_INVALID; *out = d->stats; return MIPT_OK; }

}  // extern "C"
