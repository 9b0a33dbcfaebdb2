// CUDATracer.hpp -- the tracer API of the reference (CUDATracer/CUDATracer.hpp:11-17,82-227,231-361) over libmipt:
// PathTraceSettings, CUDAScene (scene upload), PathTracer::Trace and the MakeCUDATracerProg factory, same names and
// argument meaning, so a RendererWindow-style caller compiles unchanged.  Errors surface as std::runtime_error with
// libmipt's message where the reference calls __debugbreak() (CUDACommon.h:34-54).
#pragma once
#include <cstdlib>
#include <memory>
#include <vector>

#include "CUDABuffer.hpp"
#include "Types.hpp"

namespace CUDATracer {

struct PathTraceSettings {            // CUDATracer.hpp:11-17, 72 bytes == mipt_settings
    Camera cam;
    unsigned viewportWidth, viewportHeight;
    unsigned frameID;
    unsigned seed;                    // ignored by the integrator, as in the reference (CUDATracerProg.cu:374)
    unsigned maxDepth;
};
static_assert(sizeof(PathTraceSettings) == sizeof(mipt_settings), "PathTraceSettings layout");

// Scene upload: one flat BVH + triangle stream for the whole scene (CUDAScene, CUDATracer.hpp:82-227)
class CUDAScene {
public:
    // bvhCachePath: optional BVH cache file (mipt_scene_desc::bvhCachePath); $MIPT_BVH_CACHE is used when it is null
    explicit CUDAScene(const Scene& scn, unsigned bvhBuilder = MIPT_BVH_SBVH, const char* bvhCachePath = nullptr) {
        std::vector<mipt_mesh_desc> meshes; mipt_scene_desc d = Describe(scn, meshes);
        d.bvhBuilder = bvhBuilder;
        d.bvhCachePath = bvhCachePath ? bvhCachePath : std::getenv("MIPT_BVH_CACHE");
        detail::Check(mipt_scene_create(detail::DefaultTracer(), &d, &_s), "mipt_scene_create");
        _meshCnt = d.meshCnt;
    }
    CUDAScene(const CUDAScene&) = delete;
    CUDAScene& operator=(const CUDAScene&) = delete;
    ~CUDAScene() { mipt_scene_destroy(_s); }
    // opaque device-side scene (the reference returns its SceneInfoHeader blob here)
    const void* GetGPUData() { return _s; }
    mipt_scene* handle() { return _s; }
    unsigned meshCnt() const { return _meshCnt; }
    bool BvhWasCached() const { return mipt_scene_bvh_was_cached(_s) != 0; }
    // Same meshes with moved vertices and/or edited materials (an animation step): BVH boxes re-fitted, device buffers
    // overwritten in place -- mipt_scene_refit.  The reference would build a new CUDAScene.
    void Refit(const Scene& scn) {
        std::vector<mipt_mesh_desc> meshes; mipt_scene_desc d = Describe(scn, meshes);
        detail::Check(mipt_scene_refit(_s, &d), "mipt_scene_refit");
    }
private:
    static mipt_scene_desc Describe(const Scene& scn, std::vector<mipt_mesh_desc>& meshes) {
        meshes.resize(scn.objects.size());
        for (size_t i = 0; i < scn.objects.size(); i++) {
            const Mesh& m = scn.objects[i];
            meshes[i].vertices = reinterpret_cast<const mipt_vertex*>(m.GetVertices().data());
            meshes[i].vertexCnt = (uint32_t)m.GetVertices().size();
            meshes[i].indices = m.GetIndices().data();
            meshes[i].indicesCnt = (uint32_t)m.GetIndices().size();
            std::memset(&meshes[i].material, 0, sizeof(mipt_material));
            std::memcpy(&meshes[i].material, &m.GetMat(), 45);
        }
        mipt_scene_desc d{};
        d.meshes = meshes.data(); d.meshCnt = (uint32_t)meshes.size();
        std::memcpy(&d.sky, &scn.skyLight, sizeof(mipt_skylight));
        if (!scn.envMap.empty()) { d.envMap = scn.envMap.data(); d.envW = scn.envMapDim.x; d.envH = scn.envMapDim.y; }
        return d;
    }
    mipt_scene* _s = nullptr;
    unsigned _meshCnt = 0;
};

class PathTracer {
public:
    PathTracer() { detail::DefaultTracer(); }
    virtual ~PathTracer() {}
    // one sample per pixel for settings.frameID, running-mean accumulate into accBuffer, tonemap into buffer; both are
    // device pointers (w*h*4 floats / bytes).  Asynchronous; the caller's blocking readback observes completion.
    virtual void Trace(CUDAScene& scn, TypedBuffer<PathTraceSettings>& settings, float* accBuffer, char* buffer) = 0;
};

class HIPTracerProg : public PathTracer {     // stands where CUDATracerProg stood (CUDATracerProg.hpp, CUDATracerProg.cu:925-963)
public:
    // tuning / estimator options of libmipt; not part of the reference's interface
    void SetOption(const char* key, long long value) { detail::Check(mipt_set_option(detail::DefaultTracer(), key, value), "mipt_set_option"); }
    void Trace(CUDAScene& scn, TypedBuffer<PathTraceSettings>& settings, float* accBuffer, char* buffer) override {
        const PathTraceSettings& st = settings.Get<0>();
        detail::Check(mipt_trace(detail::DefaultTracer(), scn.handle(), reinterpret_cast<const mipt_settings*>(&st), accBuffer,
                                 reinterpret_cast<unsigned char*>(buffer)), "mipt_trace");
    }
};

inline PathTracer* MakeCUDATracerProg() { return new HIPTracerProg(); }     // CUDATracer.hpp:361

}  // namespace CUDATracer
