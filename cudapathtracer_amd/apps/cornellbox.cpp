// Headless CornellBox app: what CornellBox/tracer.cpp:145-192 does, minus the window.
//   cornellbox [width height frames out.ppm [out.pfm]]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "../cpp/RendererWindow.hpp"

using namespace CUDATracer;

class AppWindow : public RendererWindow {
public:
    AppWindow(unsigned w, unsigned h) : RendererWindow(Scenes::CornellBox(), "CornellBox", w, h) {
        auto& settings = renderSettings->GetMutable<0>();
        settings.maxDepth = 4;                                  // CornellBox/tracer.cpp:152
        camParam = Scenes::CornellCamera();
        UpdateCam(settings.cam);
    }
};

int main(int argc, char** argv) {
    unsigned w = argc > 1 ? std::atoi(argv[1]) : 800, h = argc > 2 ? std::atoi(argv[2]) : 480, frames = argc > 3 ? std::atoi(argv[3]) : 64;
    try {
        AppWindow wnd(w, h);
        auto t0 = std::chrono::steady_clock::now();
        wnd.run(frames);
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        mipt_counters c; mipt_get_counters(detail::DefaultTracer(), &c);
        std::printf("cornellbox %ux%u: %u frames in %.3f s (%.1f spp/s, %.1f Mrays/s incl. per-frame readback)\n", w, h, frames, s, frames / s, c.rays / s / 1e6);
        if (argc > 4 && !wnd.WritePPM(argv[4])) return 2;
        if (argc > 5 && !wnd.WritePFM(argv[5])) return 2;
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
