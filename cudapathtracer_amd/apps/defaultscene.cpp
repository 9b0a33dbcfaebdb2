// Headless DefaultScene app: what DefaultScene/tracer.cpp:125-144 does, minus the window.  sky.hdr is absent from the
// reference checkout, so the environment map is the synthetic stand-in unless a Radiance .hdr file is given.
//   defaultscene [width height frames out.ppm [out.pfm [sky.hdr]]]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "../cpp/RendererWindow.hpp"
#include "../cpp/Rgbe.hpp"

using namespace CUDATracer;

int main(int argc, char** argv) {
    unsigned w = argc > 1 ? std::atoi(argv[1]) : 800, h = argc > 2 ? std::atoi(argv[2]) : 480, frames = argc > 3 ? std::atoi(argv[3]) : 64;
    try {
        Scene scene = Scenes::DefaultScene();
        if (argc > 6) {
            int mw = 0, mh = 0;
            if (!Rgbe::ReadFile(argv[6], scene.envMap, mw, mh)) { std::fprintf(stderr, "cannot read %s\n", argv[6]); return 2; }
            scene.envMapDim = Math::vec2ui{ (unsigned)mw, (unsigned)mh };
        } else {
            scene.envMap = Scenes::SyntheticSky(2048, 1024, 0x5EED);
            scene.envMapDim = Math::vec2ui{ 2048u, 1024u };
        }
        RendererWindow wnd(scene, "DefaultScene", w, h);
        auto t0 = std::chrono::steady_clock::now();
        wnd.run(frames);
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        mipt_counters c; mipt_get_counters(detail::DefaultTracer(), &c);
        std::printf("defaultscene %ux%u: %u frames in %.3f s (%.1f spp/s, %.1f Mrays/s incl. per-frame readback)\n", w, h, frames, s, frames / s, c.rays / s / 1e6);
        if (argc > 4 && !wnd.WritePPM(argv[4])) return 2;
        if (argc > 5 && !wnd.WritePFM(argv[5])) return 2;
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
