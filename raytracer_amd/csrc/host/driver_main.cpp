// rtx_run — headless driver for the HIP backend with the reference's flags
// (src/Configurator.cpp:78-142): -L <scn | scene.blob> -w <px> -h <px> -B -F <frames> -O,
// plus -o <file.pfm|file.bmp> to write the frame (the reference's -F mode never
// writes one, src/Main.cpp:46-54), -g <gamma> for the BMP, -N <triangles> for the synthetic scene, and
// -P to run the windowed loop's hand-over headlessly (src/WindowManager.cpp:160-187: Render, then the frame goes to the
// display as 8-bit RGBA — rtx_present_rgba8 with context->gamma — every frame; -o x.bmp then writes that 8-bit image),
// -G <n> to render on a group of n GPUs of this process (RtxConfig.numDevices; ordinals wrap around the
// devices present, so -G 2 also runs on a one-GPU box, as two members on the same device).
// It replays src/Main.cpp:13-54 (bounded frames) and :56-82 (orbit camera, run
// headlessly for -F frames) with HIPShader in the place of ThreadedShader.
// Per-frame timing goes to Performance_log.csv in the reference's format
// ("fps;frametime(ms)", src/PerformanceMonitor.cpp:10,27).
#include "hip_shader.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace rtxhost;

static void write_pfm(const char* path, const Color* px, uint32_t w, uint32_t h) {
    FILE* f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path); return; }
    fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) fwrite(&px[(size_t)y * w + x], sizeof(float), 3, f);   // bottom-up like the reference's y axis
    fclose(f);
}

// 24-bit BMP, bottom-up rows like the reference's y axis; gamma > 0 applies v^(1/gamma) first (the reference
// carries gamma = 2.2 in its context, src/RenderingContext.h:48, but only the OpenCL display path uses it)
static void write_bmp(const char* path, const Color* px, uint32_t w, uint32_t h, float gamma) {
    FILE* f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path); return; }
    const uint32_t row = (w * 3 + 3) & ~3u, size = 54 + row * h;
    unsigned char hd[54] = {'B', 'M'};
    auto put32 = [&](int off, uint32_t v) { memcpy(hd + off, &v, 4); };
    put32(2, size); put32(10, 54); put32(14, 40); put32(18, w); put32(22, h);
    hd[26] = 1; hd[28] = 24; put32(34, row * h);
    fwrite(hd, 1, 54, f);
    std::vector<unsigned char> line(row, 0);
    for (uint32_t y = 0; y < h; ++y) {
        for (uint32_t x = 0; x < w; ++x) {
            const Color& c = px[(size_t)y * w + x];
            auto q = [gamma](float v) { v = v < 0 ? 0 : (v > 1 ? 1 : v); if (gamma > 0) v = powf(v, 1.0f / gamma); return (unsigned char)(v * 255.0f + 0.5f); };
            line[3 * x] = q(c.z); line[3 * x + 1] = q(c.y); line[3 * x + 2] = q(c.x);
        }
        fwrite(line.data(), 1, row, f);
    }
    fclose(f);
}

int main(int argc, char** argv) {
    const char* scenePath = nullptr; const char* outPath = nullptr;
    int w = 1000, h = 1000, frames = 1, gpus = 1; bool bvh = false, orbit = false, present = false; long long synth = 0; float gamma = 0.0f;
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        if (a[0] != '-' || !a[1] || a[2]) { fprintf(stderr, "Unknown argument: %s\n", a); return -1; }
        auto val = [&](const char* what) -> const char* {
            if (i + 1 < argc && argv[i + 1][0] != '-') return argv[++i];
            fprintf(stderr, "Error: -%c flag requires %s\n", a[1], what); exit(-1);
        };
        switch (a[1]) {
            case 'L': scenePath = val("a scene file path"); break;
            case 'w': w = std::max(atoi(val("window width")), 100); break;
            case 'h': h = std::max(atoi(val("window height")), 100); break;
            case 'F': frames = std::max(atoi(val("number of frames")), 1); break;
            case 'T': val("number of threads"); break;                      // CPU threads: not used by this backend
            case 'B': bvh = true; break;
            case 'O': orbit = true; break;
            case 'o': outPath = val("an output file"); break;
            case 'N': synth = atoll(val("a triangle count")); break;
            case 'g': gamma = (float)atof(val("a gamma value")); break;
            case 'G': gpus = std::min(8, std::max(1, atoi(val("a number of GPUs")))); break;
            case 'P': present = true; break;
            case 'V': case 'S': break;
            case 'H': printf("Usage: rtx_run [-L scene.scn | -L scene.blob | -N triangles] [-w W] [-h H] [-B] [-F frames] [-O] [-o out.pfm|out.bmp] [-g gamma (BMP only, e.g. 2.2)] [-G gpus] [-P]\n"); return 0;
            default: fprintf(stderr, "Unknown argument: %s\n", a); return -1;
        }
    }
    // -L x.blob: a binary scene blob (the five arrays + camera + resolution as rtxh_scene_save_blob wrote them; -w / -h do
    // not apply, the BVH comes with the blob) instead of a .scn text scene
    const size_t pathLen = scenePath ? strlen(scenePath) : 0;
    const bool blob = pathLen > 5 && strcmp(scenePath + pathLen - 5, ".blob") == 0;
    RtxHostScene* scene = blob ? rtxh_scene_load_blob(scenePath) : rtxh_scene_new((uint32_t)w, (uint32_t)h);
    if (!scene) { fprintf(stderr, "%s\n", rtxh_last_error()); return -1; }
    if (scenePath && !blob && rtxh_scene_load_scn(scene, scenePath) != 0) { fprintf(stderr, "%s\n", rtxh_last_error()); return -1; }
    if (synth > 0) {
        RtxSceneDesc d; rtxh_scene_desc(scene, &d);
        const float aspect = (float)d.width / (float)d.height, scale = 100.0f * aspect;
        const float halfH = 0.41421356f * 1140.0f, halfW = halfH * aspect;
        const float pos[3] = {500, 100, 500}, n[3] = {0, 1, 0}, sc[3] = {5000, 5000, 5000};
        const float lp[3] = {500, 4000, 500}, ln[3] = {0, -1, 0};
        RtxMaterial light; memset(&light, 0, sizeof(light));
        light.albedo = RtxVec4{1, 1, 1, 1}; light.emmissionIntensity = 0.5f; light.indexOfRefraction = 1.45f; light.roughness = 0.5f; light.tintRoughness = 0.5f;
        int lid = rtxh_scene_add_material(scene, &light, "Light");
        rtxh_scene_add_disk(scene, lp, ln, 1000, lid);
        rtxh_scene_add_plane(scene, pos, n, sc, 0);
        const int32_t mats[1] = {0};
        rtxh_scene_add_heightfield(scene, (uint64_t)synth, 12345, 2.1f * halfW / scale, 2.1f * halfH / scale, 60.0f / scale, 25.0f,
                                   (500 - d.width / 2.0f) / scale, (560 - d.height / 2.0f) / scale, 0.0f, mats, 1);
    }
    if (bvh && !blob) rtxh_scene_build_bvh(scene, 8);

    RenderingContext context;
    FillContext(context, scene);
    context.numBoundedFrames = (uint32_t)frames;
    printf("Resolution : %d x %d\n", context.width, context.height);
    // src/WindowManager.cpp:26-27 (`pixels = new Color[w * h]`, zero-filled) — in page-locked memory, so that every frame lands in
    // it by one direct copy (include/rtx.h rtx_alloc_host); RTX_SHADER_PAGEABLE=1 keeps the reference's ordinary array
    const size_t numPixels = (size_t)context.width * context.height;
    std::vector<Color> pageable;
    Color* locked = getenv("RTX_SHADER_PAGEABLE") ? nullptr : (Color*)rtx_alloc_host(numPixels * sizeof(Color));
    if (!locked) pageable.assign(numPixels, Color{0, 0, 0, 0});
    else memset(locked, 0, numPixels * sizeof(Color));
    Color* const pixels = locked ? locked : pageable.data();

    RtxConfig cfg; memset(&cfg, 0, sizeof(cfg));
    if (gpus > 1) {
        int present = 0;
        if (hipGetDeviceCount(&present) != hipSuccess || present < 1) { fprintf(stderr, "no GPU\n"); return -1; }
        cfg.numDevices = gpus;
        for (int i = 0; i < gpus; ++i) cfg.devices[i] = i % present;
        printf("Device group : %d members on %d GPU(s)\n", gpus, std::min(gpus, present));
    }
    std::vector<uint32_t> rgba8(present ? (size_t)context.width * context.height : 0);
    IFrameRender* service = new HIPShader(&context, &cfg);
    FILE* csv = fopen("Performance_log.csv", "w");
    if (csv) fprintf(csv, "fps;frametime(ms)\n");
    float yaw = 0.0f, pitch = 0.0f;
    auto t0 = std::chrono::steady_clock::now();
    double total = 0;
    for (uint32_t frame = 0; frame < context.numBoundedFrames; ++frame) {
        if (orbit) {                                        // src/Main.cpp:68-82
            context.frameCounter = 0;
            rtxh_camera_orbit(&context.camera, yaw, pitch, 5000.0f);
            yaw = (yaw >= 360.0f) * (yaw - 360.0f) + (yaw < 360.0f) * yaw; yaw += 0.1f;
            pitch = (pitch >= 360.0f) * (pitch - 360.0f) + (pitch < 360.0f) * pitch; pitch += 0.1f;
        }
        if (present) {
            // UpdateWindow (src/WindowManager.cpp:160-166) without the window: the frame stays on the device and
            // what the texture upload needs — 8-bit RGBA — comes back
            RtxRenderer* h = static_cast<HIPShader*>(service)->handle();
            if (rtx_render(h, &context.camera, context.frameCounter, nullptr) != 0 ||
                rtx_present_rgba8(h, gamma > 0 ? gamma : context.gamma, (uint8_t*)rgba8.data()) != 0) {
                fprintf(stderr, "present: %s\n", rtx_last_error()); return -1;
            }
        } else {
            service->Render(pixels);                 // src/WindowManager.cpp:156
        }
        context.frameCounter++;                             // :157
        auto t1 = std::chrono::steady_clock::now();
        double dt = std::chrono::duration<double>(t1 - t0).count(); t0 = t1; total += dt;
        if (csv) fprintf(csv, "%u;%f\n", (unsigned)(1.0 / dt), dt * 1e3);
    }
    if (csv) fclose(csv);
    RtxCounters c;
    rtx_get_counters(static_cast<HIPShader*>(service)->handle(), &c);
    printf("%u frames, %.3f ms/frame wall (incl. host copy), %.3f ms/frame device, %.1f Mrays/s (device)\n",
           context.numBoundedFrames, total / frames * 1e3, c.totalMs / frames, c.rays / (c.totalMs * 1e-3) / 1e6);
    if (outPath && present) {
        // the displayed image, as a 32-bit BMP (what WindowManager::DumpContent writes, with a correct header)
        FILE* f = fopen(outPath, "wb");
        if (f) {
            const uint32_t w = context.width, h = context.height, size = 54 + w * h * 4;
            unsigned char hd[54] = {'B', 'M'};
            auto put32 = [&](int off, uint32_t v) { memcpy(hd + off, &v, 4); };
            put32(2, size); put32(10, 54); put32(14, 40); put32(18, w); put32(22, h);
            hd[26] = 1; hd[28] = 32; put32(34, w * h * 4);
            fwrite(hd, 1, 54, f);
            for (size_t i = 0; i < rgba8.size(); ++i) {                 // RGBA words -> BGRA bytes
                const uint32_t p = rgba8[i];
                const unsigned char px4[4] = {(unsigned char)(p >> 16), (unsigned char)(p >> 8), (unsigned char)p, (unsigned char)(p >> 24)};
                fwrite(px4, 1, 4, f);
            }
            fclose(f);
        }
    } else if (outPath) {
        const size_t n = strlen(outPath);
        if (n > 4 && strcmp(outPath + n - 4, ".bmp") == 0) write_bmp(outPath, pixels, context.width, context.height, gamma);
        else write_pfm(outPath, pixels, context.width, context.height);
    }
    delete service;
    if (locked) rtx_free_host(locked);
    rtxh_scene_free(scene);
    return 0;
}
