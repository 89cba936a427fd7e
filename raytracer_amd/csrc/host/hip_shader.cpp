#include "hip_shader.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rtxhost {

HIPShader::HIPShader(RenderingContext* _context, const RtxConfig* cfg) : ComputeShader(_context) {
    RtxSceneDesc d;
    memset(&d, 0, sizeof(d));
    d.objects = context->objects.data();         d.numObjects = context->objects.size();
    d.materials = context->materials.data();     d.numMaterials = context->materials.size();
    d.boxes = context->boxes.data();             d.numBoxes = context->boxes.size();
    d.textureInfo = context->textureInfo.data(); d.numTextures = context->textureInfo.size();
    d.textureData = context->textureData.data(); d.numTexels = context->textureData.size();
    d.width = context->width; d.height = context->height;
    d.useBVH = context->bvhAcceleration ? 1 : 0;
    if (rtx_create(&d, cfg, &renderer) != 0) {
        fprintf(stderr, "HIPShader: %s\n", rtx_last_error());
        exit(-1);
    }
}

void HIPShader::Render(Color* _pixels) {
    if (rtx_render(renderer, &context->camera, context->frameCounter, (float*)_pixels) != 0) {
        fprintf(stderr, "HIPShader::Render: %s\n", rtx_last_error());
        exit(-1);
    }
}

HIPShader::~HIPShader() { rtx_destroy(renderer); }

void FillContext(RenderingContext& ctx, RtxHostScene* scene) {
    RtxSceneDesc d;
    rtxh_scene_desc(scene, &d);
    ctx.width = d.width; ctx.height = d.height;
    ctx.objects.assign(d.objects, d.objects + d.numObjects);
    ctx.materials.assign(d.materials, d.materials + d.numMaterials);
    if (d.numBoxes) ctx.boxes.assign(d.boxes, d.boxes + d.numBoxes); else ctx.boxes.clear();
    ctx.textureInfo.assign(d.textureInfo, d.textureInfo + d.numTextures);
    ctx.textureData.assign(d.textureData, d.textureData + d.numTexels);
    ctx.bvhAcceleration = d.useBVH != 0;
    rtxh_scene_camera(scene, &ctx.camera);
}

} // namespace rtxhost
