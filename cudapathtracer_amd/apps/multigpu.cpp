// Headless multi-GPU renderer: BASELINE config C4 in one process -- the scene replicated on G GPUs, sample indices
// strided over them, one RCCL reduce of the accumulation buffers, tonemap on the root (include/mipt_dist.h).
//   multigpu [cornell|default] [width height frames gpus maxDepth out.ppm [out.pfm]]      gpus 0 = all visible
// Plain C++ over the C-ABI; no HIP in this translation unit.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/mipt_dist.h"
#include "../cpp/Scenes.hpp"

using namespace CUDATracer;

int main(int argc, char** argv) {
    const std::string which = argc > 1 ? argv[1] : "cornell";
    const unsigned w = argc > 2 ? std::atoi(argv[2]) : 800, h = argc > 3 ? std::atoi(argv[3]) : 480, frames = argc > 4 ? std::atoi(argv[4]) : 64;
    const int gpus = argc > 5 ? std::atoi(argv[5]) : 0;
    Scene scn = which == "default" ? Scenes::DefaultScene() : Scenes::CornellBox();
    unsigned depth = argc > 6 ? std::atoi(argv[6]) : (which == "default" ? 8 : 4);
    if (which == "default") { scn.envMap = Scenes::SyntheticSky(2048, 1024, 0x5EED); scn.envMapDim = { 2048u, 1024u }; }
    Scenes::CameraParams cp = which == "default" ? Scenes::CameraParams{} : Scenes::CornellCamera();
    if (which == "default") { cp.pos = { -1, 1, 0 }; cp.azimuth = 0; cp.zenith = (float)(M_PI / 2); cp.fov = (float)(M_PI / 3); }

    mipt_dist* d = nullptr;
    int rc = mipt_dist_create(gpus, &d);
    if (rc) { std::fprintf(stderr, "mipt_dist_create: %d (no MI355X visible?)\n", rc); return 1; }
    std::vector<mipt_mesh_desc> meshes(scn.objects.size());
    for (size_t i = 0; i < scn.objects.size(); i++) {
        const Mesh& m = scn.objects[i];
        meshes[i].vertices = reinterpret_cast<const mipt_vertex*>(m.GetVertices().data()); meshes[i].vertexCnt = (uint32_t)m.GetVertices().size();
        meshes[i].indices = m.GetIndices().data(); meshes[i].indicesCnt = (uint32_t)m.GetIndices().size();
        std::memset(&meshes[i].material, 0, sizeof(mipt_material)); std::memcpy(&meshes[i].material, &m.GetMat(), 45);
    }
    mipt_scene_desc desc{};
    desc.meshes = meshes.data(); desc.meshCnt = (uint32_t)meshes.size();
    std::memcpy(&desc.sky, &scn.skyLight, sizeof(mipt_skylight));
    if (!scn.envMap.empty()) { desc.envMap = scn.envMap.data(); desc.envW = scn.envMapDim.x; desc.envH = scn.envMapDim.y; }
    rc = mipt_dist_scene_create(d, &desc);
    if (rc) { std::fprintf(stderr, "scene: %s\n", mipt_dist_last_error(d)); mipt_dist_destroy(d); return 1; }

    mipt_settings st{};
    Camera cam; Scenes::UpdateCam(cp, cam);
    std::memcpy(&st.cam, &cam, sizeof(mipt_camera));
    st.viewportWidth = w; st.viewportHeight = h; st.maxDepth = depth;
    std::vector<float> acc((size_t)w * h * 4); std::vector<unsigned char> ldr((size_t)w * h * 4);
    rc = mipt_dist_render(d, &st, 0, frames, acc.data(), ldr.data());
    if (rc) { std::fprintf(stderr, "render: %s\n", mipt_dist_last_error(d)); mipt_dist_destroy(d); return 1; }
    mipt_dist_stats s; mipt_dist_get_stats(d, &s);
    std::printf("multigpu %s %ux%u depth %u: %u frames on %d GPU(s): render %.3f s, reduce+tonemap %.4f s, %.1f spp/s, %.1f Mrays/s (%llu rays)\n",
                which.c_str(), w, h, depth, frames, s.gpus, s.render_s, s.reduce_s, frames / (s.render_s + s.reduce_s), s.rays / (s.render_s + s.reduce_s) / 1e6,
                (unsigned long long)s.rays);
    if (argc > 7) {      // PPM, top row first (the buffers are bottom-up)
        FILE* f = std::fopen(argv[7], "wb"); if (!f) return 2;
        std::fprintf(f, "P6 %u %u 255\n", w, h);
        for (unsigned y = 0; y < h; y++) for (unsigned x = 0; x < w; x++) std::fwrite(&ldr[((size_t)(h - 1 - y) * w + x) * 4], 1, 3, f);
        std::fclose(f);
    }
    if (argc > 8) {      // PFM of the mean radiance, bottom-up as the format wants
        FILE* f = std::fopen(argv[8], "wb"); if (!f) return 2;
        std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
        for (size_t i = 0; i < (size_t)w * h; i++) std::fwrite(&acc[i * 4], 4, 3, f);
        std::fclose(f);
    }
    mipt_dist_destroy(d);
    return 0;
}
