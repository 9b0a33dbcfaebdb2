// RendererWindow.hpp -- headless work-alike of the reference's window driver (Common/RendererWindow.hpp:9-185,
// Common/GLFWWindow.cpp:128-147): same members and call sequence (resize -> loop { Trace; bitmap->cpu_data();
// frameID++ }), no GLFW/OpenGL.  DrawBitmap keeps the last frame and can write it as a PPM instead of blitting it.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "CUDATracer.hpp"
#include "Scenes.hpp"

class RendererWindow {
protected:
    CUDATracer::Math::vec2i bitmapSize;
    std::shared_ptr<CUDATracer::CUDABuffer> bitmap;
    std::shared_ptr<CUDATracer::CUDABuffer> accBuffer;
    std::shared_ptr<CUDATracer::TypedBuffer<CUDATracer::PathTraceSettings>> renderSettings;
    CUDATracer::CUDAScene renderScene;
    CUDATracer::PathTracer* renderer;
    std::vector<std::uint8_t> lastFrame;
    unsigned width, height;
    std::string previewPrefix; unsigned previewEvery = 0;

    void UpdateCam(CUDATracer::Camera& cam) { CUDATracer::Scenes::UpdateCam(camParam, cam); }

public:
    using CameraParams = CUDATracer::Scenes::CameraParams;
    CameraParams camParam;

    RendererWindow(const CUDATracer::Scene& scene, const std::string& title = "CUDAPathTracer", unsigned width_ = 800, unsigned height_ = 480)
        : renderScene(scene), camParam{}, width(width_), height(height_) {
        (void)title;
        renderer = CUDATracer::MakeCUDATracerProg();
        renderSettings = std::make_shared<CUDATracer::TypedBuffer<CUDATracer::PathTraceSettings>>();
        camParam = CUDATracer::Scenes::DefaultCamera();
        auto& settings = renderSettings->GetMutable<0>();
        UpdateCam(settings.cam);
        settings.viewportHeight = height; settings.viewportWidth = width;
        settings.frameID = 0; settings.maxDepth = 8;
    }
    virtual ~RendererWindow() { renderSettings = nullptr; bitmap = nullptr; accBuffer = nullptr; delete renderer; }

    void ResetFrameID() { renderSettings->GetMutable<0>().frameID = 0; }

    virtual void resize(const CUDATracer::Math::vec2i& newSize) {
        bitmapSize = newSize;
        size_t n = (size_t)newSize.x * newSize.y * 4;
        accBuffer = std::make_shared<CUDATracer::CUDABuffer>(n * sizeof(float));
        bitmap = std::make_shared<CUDATracer::CUDABuffer>(n * sizeof(char));
        auto& settings = renderSettings->GetMutable<0>();
        settings.viewportHeight = newSize.y; settings.viewportWidth = newSize.x;
        ResetFrameID();
    }
    virtual void Update() { HandleKeyInput(); UpdateCam(renderSettings->GetMutable<0>().cam); }

    // Input without a window: the reference polls GLFW keys / mouse (RendererWindow.hpp:150-183); here the same state is
    // injected by the caller.  Same steps (0.1 per frame along front/right, 1/400 rad per pixel) and the same rule:
    // any camera change restarts the accumulation (ResetFrameID).
    struct KeyState { bool w = false, s = false, a = false, d = false; } keys;
    bool mouseLeftDown = false;
    CUDATracer::Math::vec2f lastMousePos{ 0.f, 0.f };
    void HandleKeyInput() {
        int fwd = (keys.w ? 1 : 0) - (keys.s ? 1 : 0), right = (keys.d ? 1 : 0) - (keys.a ? 1 : 0);
        if (fwd != 0 || right != 0) {
            auto& cam = renderSettings->GetMutable<0>().cam;
            camParam.pos += cam.front * (float)(fwd * 0.1);
            camParam.pos += cam.right * (float)(right * 0.1);
            ResetFrameID();
        }
    }
    virtual void mouseMotion(double px, double py) {
        double dx = px - lastMousePos.x, dy = py - lastMousePos.y;
        lastMousePos = { (float)px, (float)py };
        if (std::abs(dx) < 1e-3 && std::abs(dy) < 1e-3) return;
        if (mouseLeftDown) { camParam.azimuth += (float)(dx / 400); camParam.zenith += (float)(dy / 400); ResetFrameID(); }
    }

    // progressive preview: write "<prefix>_<frames>.ppm" after every `every` accumulated frames (0 = off)
    void SetPreview(const std::string& prefix, unsigned every) { previewPrefix = prefix; previewEvery = every; }
    bool SceneBvhWasCached() { return renderScene.BvhWasCached(); }
    // an animation step: same meshes, moved vertices / edited materials (CUDAScene::Refit); accumulation restarts
    void UpdateScene(const CUDATracer::Scene& scn) { renderScene.Refit(scn); ResetFrameID(); }

    virtual void render() {
        renderer->Trace(renderScene, *renderSettings, (float*)accBuffer->mutable_gpu_data(), (char*)bitmap->mutable_gpu_data());
        auto tbuffer = (const std::uint8_t*)bitmap->cpu_data();       // blocking readback, observes the frame
        renderSettings->GetMutable<0>().frameID++;                    // the CALLER advances the sample index
        DrawBitmap(bitmapSize.x, bitmapSize.y, tbuffer);
        const unsigned done = renderSettings->Get<0>().frameID;
        if (previewEvery && done % previewEvery == 0) WritePPM(previewPrefix + "_" + std::to_string(done) + ".ppm");
    }
    virtual void DrawBitmap(int w, int h, const std::uint8_t* rgba) { lastFrame.assign(rgba, rgba + (size_t)w * h * 4); }

    // GLFWWindow::run with a frame budget instead of a close button
    void run(unsigned frames) {
        resize({ (int)width, (int)height });
        for (unsigned f = 0; f < frames; f++) { Update(); render(); }
    }
    unsigned frameID() { return renderSettings->Get<0>().frameID; }
    const float* accumulation() { return (const float*)accBuffer->cpu_data(); }
    const std::vector<std::uint8_t>& frame() const { return lastFrame; }

    bool WritePPM(const std::string& path) const {      // row 0 of the buffers is the BOTTOM of the image
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) return false;
        std::fprintf(f, "P6 %d %d 255\n", bitmapSize.x, bitmapSize.y);
        for (int y = bitmapSize.y - 1; y >= 0; y--) for (int x = 0; x < bitmapSize.x; x++) std::fwrite(&lastFrame[((size_t)y * bitmapSize.x + x) * 4], 1, 3, f);
        std::fclose(f);
        return true;
    }
    // Checkpoint / resume.  The accumulation buffer, the sample index and the camera are the whole state of a progressive
    // render (CUDATracerProg.cu:878,895-897: frame k folds with weight 1/(k+1), frame 0 overwrites); the reference keeps it
    // only in the window's lifetime.  A checkpoint is a 64-byte header + the RGBA32F accumulation, bottom-up as in memory.
    struct CheckpointHeader {
        char magic[8]; std::uint32_t version, width, height, frameID, maxDepth; CameraParams cam; std::uint32_t pad[3];
    };
    static_assert(sizeof(CheckpointHeader) == 64, "checkpoint header is 64 bytes");
    bool SaveCheckpoint(const std::string& path) {
        if (!accBuffer) return false;
        CheckpointHeader h{};
        std::memcpy(h.magic, "MIPTCKP", 8); h.version = 1;
        h.width = (std::uint32_t)bitmapSize.x; h.height = (std::uint32_t)bitmapSize.y;
        h.frameID = renderSettings->Get<0>().frameID; h.maxDepth = renderSettings->Get<0>().maxDepth; h.cam = camParam;
        const float* a = accumulation();                   // blocking read-back: every frame traced so far is in it
        const std::string tmp = path + ".tmp";             // written beside the target and renamed: never a half-written checkpoint
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) return false;
        const size_t n = (size_t)h.width * h.height * 4;
        bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && std::fwrite(a, sizeof(float), n, f) == n;
        ok = (std::fclose(f) == 0) && ok;
        if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) { std::remove(tmp.c_str()); return false; }
        return true;
    }
    // Resumes from a checkpoint: buffers of the stored size, the stored camera, depth and sample index; the next render()
    // traces sample `frameID` and folds it with weight 1 / (frameID + 1), exactly as if the run had never stopped.
    // False (and nothing changed) when the file is missing, damaged, truncated, or of another version.
    bool LoadCheckpoint(const std::string& path) {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        CheckpointHeader h{};
        bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, "MIPTCKP", 8) == 0 && h.version == 1
                  && h.width >= 1 && h.height >= 1 && h.width <= 16384 && h.height <= 16384 && h.maxDepth >= 1 && h.maxDepth <= 64;
        std::vector<float> acc;
        if (ok) {
            const size_t n = (size_t)h.width * h.height * 4;
            acc.resize(n);
            ok = std::fread(acc.data(), sizeof(float), n, f) == n && std::fgetc(f) == EOF;
        }
        std::fclose(f);
        if (!ok) return false;
        resize({ (int)h.width, (int)h.height });
        std::memcpy(accBuffer->mutable_cpu_data(), acc.data(), acc.size() * sizeof(float));     // uploaded with the next Trace
        camParam = h.cam;
        auto& settings = renderSettings->GetMutable<0>();
        settings.frameID = h.frameID; settings.maxDepth = h.maxDepth;
        UpdateCam(settings.cam);
        return true;
    }

    bool WritePFM(const std::string& path) {             // PFM is bottom-up like the buffers
        const float* a = accumulation();
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) return false;
        std::fprintf(f, "PF\n%d %d\n-1.0\n", bitmapSize.x, bitmapSize.y);
        for (size_t p = 0; p < (size_t)bitmapSize.x * bitmapSize.y; p++) std::fwrite(a + p * 4, sizeof(float), 3, f);
        std::fclose(f);
        return true;
    }
};
