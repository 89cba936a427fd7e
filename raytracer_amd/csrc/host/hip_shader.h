// hip_shader.h — the C++ face of the backend: a render service with the
// reference's interface shape, sitting where ThreadedShader / CLShader sit.
//
//   reference                                   here
//   class IFrameRender  (src/IFrameRender.h:6-9)      rtxhost::IFrameRender
//   class ComputeShader (src/ComputeShader.h:7-18)    rtxhost::ComputeShader
//   struct RenderingContext (src/RenderingContext.h)  rtxhost::RenderingContext (the fields a backend reads)
//   ThreadedShader / CLShader                         rtxhost::HIPShader
//
// The reference's own RenderingContext.h cannot be included in this image (it
// pulls in GL/glew.h and GLFW/glfw3.h), so the context is mirrored with the same
// member names and element layouts; inside the reference tree HIPShader is the
// 40-line class shown in INTEGRATION.md and uses the real headers.
#pragma once
#include "../../../include/rtx.h"
#include "../../../include/rtx_host.h"

#include <cstdint>
#include <vector>

namespace rtxhost {

typedef RtxVec4 Color;   // src/Color.h:9-145 (R,G,B,A)

struct RenderingContext {
    uint32_t width = 640, height = 480, depth = 480;          // src/RenderingContext.h:17-19
    bool bvhAcceleration = false, boundedFrames = false, followCenter = false, useCPU = false;
    std::vector<RtxObject> objects;                           // :32
    std::vector<RtxMaterial> materials;                       // :33
    std::vector<RtxBoundingBox> boxes;                        // :36
    std::vector<RtxTexture> textureInfo;                      // :39
    std::vector<unsigned int> textureData;                    // :40
    RtxCamera camera;                                         // :43
    uint32_t frameCounter = 0;                                // :44
    uint32_t numBoundedFrames = 1, numThreads = 1;            // :46-47
    float gamma = 2.2f;                                       // :48
};

class IFrameRender {
public:
    virtual void Render(Color* _pixels) = 0;
    virtual ~IFrameRender() {}
};

class ComputeShader : public IFrameRender {
protected:
    RenderingContext* context;
public:
    explicit ComputeShader(RenderingContext* _context) : context(_context) {}
    virtual void Render(Color* _pixels) = 0;
};

class HIPShader : public ComputeShader {
    RtxRenderer* renderer = nullptr;
public:
    // Binds to the filled context (scene + BVH complete, as at src/Main.cpp:33-39).
    // Setup errors follow the reference's convention: message, then exit(-1)
    // (src/ComputeEnvironment.cpp:129-132).
    HIPShader(RenderingContext* _context, const RtxConfig* cfg = nullptr);
    void Render(Color* _pixels) override;       // synchronous; frame complete on return
    RtxRenderer* handle() const { return renderer; }
    ~HIPShader() override;
};

// Fill a context from the host scene container (what Configurator::ParseArgs leaves behind).
void FillContext(RenderingContext& ctx, RtxHostScene* scene);

} // namespace rtxhost
