// CUDATracer.hpp -- the tracer API of the reference (CUDATracer/CUDATracer.hpp:11-17,82-227,231-361) over libmipt:
// PathTraceSettings, CUDAScene (scene upload), PathTracer::Trace and the MakeCUDATracerProg factory, same names and
// argument meaning, so a RendererWindow-style caller compiles unchanged.  Errors surface as std::runtime_error with
// libmipt's message where the reference calls __debugbreak() (CUDACommon.h:34-54).
#pragma once
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
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

struct BVHNode;     // the reference's per-mesh pointer tree (Types.hpp:234-251) has no counterpart here: one flat BVH per
                    // scene lives inside libmipt (mipt_scene_get_bvh), so MeshInfoHeader::bvh is always null

// Host mirrors of the reference's device blobs (CUDATracer.hpp:23-74), same fields, same order, same accessors.  In the
// reference these sit in CUDABuffers whose device side the kernel reads; here the kernels read libmipt's flat streams
// and the mirrors are what GetMesh / GetInfo hand out -- edits made through the mutable accessors reach the GPU with
// the next Trace (CUDAScene::Sync -> mipt_scene_refit).
struct MeshInfoHeader {                // [header | Vertex[vertexCnt] | uint32[indicesCnt]] in one allocation
    AABB aabb;
    BVHNode* bvh;
    SurfaceMaterial material;
    Math::mat4x4f transform;
    std::uint32_t vertexCnt, indicesCnt;
    unsigned TotalSize() const { return (unsigned)(sizeof(MeshInfoHeader) + sizeof(Vertex) * vertexCnt + sizeof(std::uint32_t) * indicesCnt); }
    Vertex* GetVertices() { return (Vertex*)(this + 1); }
    const Vertex* GetVertices() const { return (const Vertex*)(this + 1); }
    std::uint32_t* GetIndices() { return (std::uint32_t*)((char*)this + sizeof(Vertex) * vertexCnt + sizeof(MeshInfoHeader)); }
    const std::uint32_t* GetIndices() const { return (const std::uint32_t*)((const char*)this + sizeof(Vertex) * vertexCnt + sizeof(MeshInfoHeader)); }
};
struct SceneInfoHeader {               // [header | MeshInfoHeader*[meshCnt]]
    SkyLight skyLight;
    float* envMap;                     // host copy of the env map (the device copy is libmipt's); read-only after upload
    Math::vec2ui envMapDim;
    unsigned meshCnt;
    MeshInfoHeader** GetPayload() { return (MeshInfoHeader**)(this + 1); }
    const MeshInfoHeader** GetPayload() const { return (const MeshInfoHeader**)(this + 1); }
    MeshInfoHeader*& operator[](unsigned index) { return GetPayload()[index]; }
    const MeshInfoHeader* operator[](unsigned index) const { return GetPayload()[index]; }
};

// Scene upload: one flat BVH + triangle stream for the whole scene (CUDAScene, CUDATracer.hpp:82-227)
class CUDAScene {
public:
    // bvhCachePath: optional BVH cache file (mipt_scene_desc::bvhCachePath); $MIPT_BVH_CACHE is used when it is null
    // bvhBuilder: MIPT_BVH_*; left at its default it is $MIPT_BVH_BUILDER (0 SBVH, 1 median, 2 built on the GPU), else the SBVH
    static constexpr unsigned kBuilderFromEnv = ~0u;
    explicit CUDAScene(const Scene& scn, unsigned bvhBuilder = kBuilderFromEnv, const char* bvhCachePath = nullptr) {
        std::vector<mipt_mesh_desc> meshes; mipt_scene_desc d = Describe(scn, meshes);
        if (bvhBuilder == kBuilderFromEnv) { const char* e = std::getenv("MIPT_BVH_BUILDER"); bvhBuilder = e && e[0] ? (unsigned)std::atoi(e) : (unsigned)MIPT_BVH_SBVH; }
        d.bvhBuilder = bvhBuilder;
        d.bvhCachePath = bvhCachePath ? bvhCachePath : std::getenv("MIPT_BVH_CACHE");
        detail::Check(mipt_scene_create(detail::DefaultTracer(), &d, &_s), "mipt_scene_create");
        _meshCnt = d.meshCnt;
        Mirror(scn);
    }
    CUDAScene(const CUDAScene&) = delete;
    CUDAScene& operator=(const CUDAScene&) = delete;
    ~CUDAScene() { mipt_scene_destroy(_s); }
    // opaque device-side scene (the reference returns its SceneInfoHeader blob here)
    const void* GetGPUData() { return _s; }
    mipt_scene* handle() { return _s; }
    unsigned meshCnt() const { return _meshCnt; }
    bool BvhWasCached() const { return mipt_scene_bvh_was_cached(_s) != 0; }

    // CUDATracer.hpp:207-226.  The mutable forms mark the scene dirty, as mutable_cpu_data() moves a CUDABuffer's head
    // to the CPU in the reference; PathTracer::Trace then calls Sync().  What may change: vertex positions / normals,
    // the material, the sky light -- not the counts, the index buffers' length or the env map.
    MeshInfoHeader& GetMeshMutable(unsigned idx) { _dirty = true; return *Header(idx); }
    const MeshInfoHeader& GetMesh(unsigned idx) { return *Header(idx); }
    SceneInfoHeader& GetInfoMutable() { _dirty = true; return *reinterpret_cast<SceneInfoHeader*>(_info.data()); }
    const SceneInfoHeader& GetInfo() { return *reinterpret_cast<SceneInfoHeader*>(_info.data()); }
    bool Dirty() const { return _dirty; }
    // pushes edits made through the mutable accessors to the GPU: BVH boxes re-fitted, flat buffers overwritten in place
    void Sync() {
        if (!_dirty) return;
        std::vector<mipt_mesh_desc> meshes(_meshCnt);
        for (unsigned i = 0; i < _meshCnt; i++) {
            MeshInfoHeader* h = Header(i);
            meshes[i].vertices = reinterpret_cast<const mipt_vertex*>(h->GetVertices()); meshes[i].vertexCnt = h->vertexCnt;
            meshes[i].indices = h->GetIndices(); meshes[i].indicesCnt = h->indicesCnt;
            std::memset(&meshes[i].material, 0, sizeof(mipt_material));
            std::memcpy(&meshes[i].material, &h->material, 45);
        }
        mipt_scene_desc d{};
        d.meshes = meshes.data(); d.meshCnt = _meshCnt;
        std::memcpy(&d.sky, &GetInfo().skyLight, sizeof(mipt_skylight));
        detail::Check(mipt_scene_refit(_s, &d), "mipt_scene_refit");
        _dirty = false;
    }
    // Same meshes with moved vertices and/or edited materials (an animation step): BVH boxes re-fitted, device buffers
    // overwritten in place -- mipt_scene_refit.  The reference would build a new CUDAScene.
    void Refit(const Scene& scn) {
        std::vector<mipt_mesh_desc> meshes; mipt_scene_desc d = Describe(scn, meshes);
        detail::Check(mipt_scene_refit(_s, &d), "mipt_scene_refit");
        Mirror(scn); _dirty = false;
    }
private:
    MeshInfoHeader* Header(unsigned idx) {
        if (idx >= _meshCnt) throw std::out_of_range("CUDAScene: mesh index");       // the reference asserts (CUDATracer.hpp:66)
        return reinterpret_cast<MeshInfoHeader*>(_mesh[idx].data());
    }
    void Mirror(const Scene& scn) {
        _mesh.assign(scn.objects.size(), {});
        _env = scn.envMap;
        for (size_t i = 0; i < scn.objects.size(); i++) {
            const Mesh& m = scn.objects[i];
            MeshInfoHeader h{};
            h.aabb = m.GetAABB(); h.bvh = nullptr; h.material = m.GetMat(); h.transform = m.transform;
            h.vertexCnt = (std::uint32_t)m.GetVertices().size(); h.indicesCnt = (std::uint32_t)m.GetIndices().size();
            _mesh[i].resize((h.TotalSize() + 7) / 8);             // 8-byte units: the header holds a pointer
            std::memcpy(_mesh[i].data(), &h, sizeof h);
            MeshInfoHeader* dst = reinterpret_cast<MeshInfoHeader*>(_mesh[i].data());
            if (h.vertexCnt) std::memcpy(dst->GetVertices(), m.GetVertices().data(), sizeof(Vertex) * h.vertexCnt);
            if (h.indicesCnt) std::memcpy(dst->GetIndices(), m.GetIndices().data(), sizeof(std::uint32_t) * h.indicesCnt);
        }
        _info.assign((sizeof(SceneInfoHeader) + sizeof(void*) * _mesh.size() + 7) / 8, 0);
        SceneInfoHeader* si = reinterpret_cast<SceneInfoHeader*>(_info.data());
        si->skyLight = scn.skyLight; si->envMap = _env.empty() ? nullptr : _env.data(); si->envMapDim = scn.envMapDim; si->meshCnt = (unsigned)_mesh.size();
        for (size_t i = 0; i < _mesh.size(); i++) (*si)[(unsigned)i] = reinterpret_cast<MeshInfoHeader*>(_mesh[i].data());
    }
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
    bool _dirty = false;
    std::vector<std::vector<std::uint64_t>> _mesh;     // per mesh: MeshInfoHeader + vertices + indices
    std::vector<std::uint64_t> _info;                  // SceneInfoHeader + MeshInfoHeader*[meshCnt]
    std::vector<float> _env;
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
        scn.Sync();                                 // edits made through GetMeshMutable / GetInfoMutable
        const PathTraceSettings& st = settings.Get<0>();
        detail::Check(mipt_trace(detail::DefaultTracer(), scn.handle(), reinterpret_cast<const mipt_settings*>(&st), accBuffer,
                                 reinterpret_cast<unsigned char*>(buffer)), "mipt_trace");
    }
};

inline PathTracer* MakeCUDATracerProg() { return new HIPTracerProg(); }     // CUDATracer.hpp:361

}  // namespace CUDATracer
