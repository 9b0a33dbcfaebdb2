// Headless mesh viewer: a Wavefront OBJ (or Stanford PLY, by extension) mesh on the DefaultScene's ground quad under the synthetic sky (or a Radiance
// .hdr file), rendered through the same RendererWindow call sequence as the two reference apps.  Not in the reference
// (its scenes are procedural); it exercises the ingestion side of the path: ObjLoader.hpp, the parallel SBVH build and
// the BVH cache ($MIPT_BVH_CACHE or the 7th argument).
//   objscene mesh.obj|mesh.ply [width height frames out.ppm [out.pfm [bvh.cache [sky.hdr]]]]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "../cpp/ObjLoader.hpp"
#include "../cpp/PlyLoader.hpp"
#include "../cpp/RendererWindow.hpp"
#include "../cpp/Rgbe.hpp"

using namespace CUDATracer;

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: objscene mesh.obj [width height frames out.ppm [out.pfm [bvh.cache [sky.hdr]]]]\n"); return 2; }
    unsigned w = argc > 2 ? std::atoi(argv[2]) : 800, h = argc > 3 ? std::atoi(argv[3]) : 480, frames = argc > 4 ? std::atoi(argv[4]) : 64;
    try {
        SurfaceMaterial meshMat{}; meshMat.color = { 0.9f, 0.7f, 0.1f }; meshMat.roughness = 0.2f; meshMat.metallic = 0.8f; meshMat.opacity = 1.f; meshMat.n = 1.5f;
        SurfaceMaterial groundMat{}; groundMat.color = { 0.9f, 0.9f, 0.8f }; groundMat.roughness = 0.5f; groundMat.metallic = 0.01f; groundMat.opacity = 1.f; groundMat.n = 2.7f;
        Scene scene{ { { 0.98f, 0.97f, 0.99f }, -Math::normalize(Math::vec3f{ 1, 1, 1 }), 5.f }, {}, {}, {} };
        auto t0 = std::chrono::steady_clock::now();
        const std::string meshPath = argv[1];
        const bool ply = meshPath.size() > 4 && meshPath.compare(meshPath.size() - 4, 4, ".ply") == 0;
        scene.objects.push_back(ConstructMesh(ply ? LoadPLY(meshPath) : LoadOBJ(meshPath), meshMat, Math::mat4x4f(1.f)));
        scene.objects.push_back(ConstructMesh(MakeQuad(), groundMat, Math::MatMul(Math::MakeTranslate(Math::vec3f{ -4, 0, -4 }), Math::MakeScale(Math::vec3f{ 8, 1, 8 }))));
        if (argc > 8) {
            int mw = 0, mh = 0;
            if (!Rgbe::ReadFile(argv[8], scene.envMap, mw, mh)) { std::fprintf(stderr, "cannot read %s\n", argv[8]); return 2; }
            scene.envMapDim = Math::vec2ui{ (unsigned)mw, (unsigned)mh };
        } else {
            scene.envMap = Scenes::SyntheticSky(2048, 1024, 0x5EED);
            scene.envMapDim = Math::vec2ui{ 2048u, 1024u };
        }
        if (argc > 7 && argv[7][0]) setenv("MIPT_BVH_CACHE", argv[7], 1);
        RendererWindow wnd(scene, "ObjScene", w, h);
        double tb = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("objscene: %zu triangles loaded, BVH %s, scene ready in %.2f s\n", scene.objects[0].GetIndices().size() / 3,
                    wnd.SceneBvhWasCached() ? "from cache" : "built", tb);
        t0 = std::chrono::steady_clock::now();
        wnd.run(frames);
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        mipt_counters c; mipt_get_counters(detail::DefaultTracer(), &c);
        std::printf("objscene %ux%u: %u frames in %.3f s (%.1f spp/s, %.1f Mrays/s incl. per-frame readback)\n", w, h, frames, s, frames / s, c.rays / s / 1e6);
        if (argc > 5 && !wnd.WritePPM(argv[5])) return 2;
        if (argc > 6 && argv[6][0] && !wnd.WritePFM(argv[6])) return 2;
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
