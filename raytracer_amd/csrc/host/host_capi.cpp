// host_capi.cpp — extern "C" surface of the host scene container (include/rtx_host.h).
#include "host_scene.h"

#include <algorithm>
#include <cstdio>

using namespace rtxh;

struct RtxHostScene { HostScene s; };

static thread_local std::string g_lastError;
static int fail(const std::string& m) { g_lastError = m; return 1; }

extern "C" {

const char* rtxh_last_error(void) { return g_lastError.c_str(); }

RtxHostScene* rtxh_scene_new(uint32_t reqW, uint32_t reqH) {
    RtxHostScene* h = new RtxHostScene();
    HostScene& s = h->s;
    // src/Configurator.cpp:6-11 — camera is placed from the 1000x1000 defaults, before any -w/-h
    memset(&s.camera, 0, sizeof(s.camera));
    s.camera.front = v4(0, 0, 1); s.camera.up = v4(0, 1, 0); s.camera.right = v4(1, 0, 0);
    s.camera.position = v4(1000 / 2.0f, 1000 / 2.0f, -900.0f);
    s.camera.movementSpeed = 500.0f; s.camera.rotationSpeed = 100.0f;
    s.camera.nearView = 0.1f; s.camera.farView = 100.0f; s.camera.fov = 45.0f;
    s.camera.pitch = 0.0f; s.camera.yaw = 90.0f;                       // src/Camera.h:11-24
    s.depth = 480;
    // src/Configurator.cpp:144-149
    s.width = std::max((uint32_t)32, (reqW + 16) / 32 * 32);
    s.height = std::max((uint32_t)32, (reqH + 16) / 32 * 32);
    s.camera.aspectRatio = s.width / (float)s.height;
    // src/Configurator.cpp:23-48
    RtxMaterial m; memset(&m, 0, sizeof(m));
    m.albedo = v4(0.5f, 0.5f, 0.5f, 1.0f);
    m.indexOfRefraction = 1.45f; m.roughness = 0.5f; m.tintRoughness = 0.5f; m.textureID = 0;
    s.materials.push_back(m);
    RtxTexture t; t.width = 1; t.height = 1; t.offset = 0; t.checksum = UINT32_MAX;
    s.textureInfo.push_back(t);
    s.textureData.push_back(UINT32_MAX);
    return h;
}

void rtxh_scene_free(RtxHostScene* h) { delete h; }

int rtxh_scene_set_resolution(RtxHostScene* h, uint32_t w, uint32_t hgt) {
    if (!h->s.objects.empty()) return fail("set the resolution before loading geometry (mesh placement depends on it)");
    h->s.width = w; h->s.height = hgt;
    h->s.camera.aspectRatio = w / (float)hgt;
    return 0;
}

int rtxh_scene_load_scn(RtxHostScene* h, const char* path) { return load_scn(h->s, path) ? 0 : fail(get_error()); }
int rtxh_scene_load_mtl(RtxHostScene* h, const char* path) { return load_mtl(h->s, path) ? 0 : fail(get_error()); }
int rtxh_scene_load_obj(RtxHostScene* h, const char* path) { return load_obj(h->s, path) ? 0 : fail(get_error()); }

int rtxh_scene_material_id(RtxHostScene* h, const char* name) {
    auto it = h->s.materialNames.find(name);
    return it == h->s.materialNames.end() ? 0 : (int)it->second;
}
int rtxh_scene_add_texture(RtxHostScene* h, const uint32_t* texels, int32_t w, int32_t hgt) {
    return add_texture(h->s, texels, w, hgt, 0x7e57u + (uint32_t)h->s.textureInfo.size());
}
int rtxh_scene_add_material(RtxHostScene* h, const RtxMaterial* m, const char* name) {
    h->s.materials.push_back(*m);
    int id = (int)h->s.materials.size() - 1;
    if (name && *name) h->s.materialNames[name] = (uint32_t)id;
    return id;
}
int rtxh_scene_add_plane(RtxHostScene* h, const float p[3], const float n[3], const float sc[3], int mat) {
    RtxVec4 normals[3] = {v4(n[0], n[1], n[2]), v4(0, 1, 0), v4(0, 1, 0)};   // "normal" sets normals[0] only (src/SceneSerializer.cpp:131-139)
    add_plane(h->s, v4(p[0], p[1], p[2]), normals, v4(sc[0], sc[1], sc[2]), mat);
    return 0;
}
int rtxh_scene_add_disk(RtxHostScene* h, const float p[3], const float n[3], float radius, int mat) {
    add_disk(h->s, v4(p[0], p[1], p[2]), v4(n[0], n[1], n[2]), radius, mat);
    return 0;
}
int rtxh_scene_add_cube(RtxHostScene* h, const float p[3], const float sc[3], int mat) {
    add_cube(h->s, v4(p[0], p[1], p[2]), v4(sc[0], sc[1], sc[2]), mat);
    return 0;
}
int rtxh_scene_add_sphere(RtxHostScene* h, const float p[3], float radius, int mat) {
    RtxObject o = blank_object();
    o.type = RTX_SPHERE; o.radius = radius; o.position = v4(p[0], p[1], p[2]);
    o.normals[0] = o.normals[1] = o.normals[2] = v4(0, 1, 0);
    o.materialID = mat;
    h->s.objects.push_back(o);
    return 0;
}
int rtxh_scene_add_mesh(RtxHostScene* h, const float* vertices, const float* normals, uint64_t nv,
                        const uint32_t* indices, const int32_t* materialIDs, uint64_t ntri, int materialID0) {
    for (uint64_t i = 0; i < 3 * ntri; ++i) if (indices[i] >= nv) return fail("mesh index out of range");
    add_mesh(h->s, vertices, normals, nv, indices, materialIDs, ntri, materialID0);
    return 0;
}
int rtxh_scene_add_heightfield(RtxHostScene* h, uint64_t ntri, uint32_t seed, float extentU, float extentV, float amplitude,
                               float tiltDeg, float cx, float cy, float cz, const int32_t* mats, int32_t nmats) {
    add_heightfield(h->s, ntri, seed, extentU, extentV, amplitude, tiltDeg, cx, cy, cz, mats, nmats);
    return 0;
}
int rtxh_write_heightfield_obj(RtxHostScene* h, const char* path, uint64_t ntri, uint32_t seed, float extentU, float extentV,
                               float amplitude, float tiltDeg, float cx, float cy, float cz, const int32_t* mats, int32_t nmats) {
    return write_heightfield_obj(h->s, path, ntri, seed, extentU, extentV, amplitude, tiltDeg, cx, cy, cz, mats, nmats) ? 0 : fail(get_error());
}

int rtxh_scene_build_bvh(RtxHostScene* h, int nthreads) { build_bvh(h->s, nthreads < 1 ? 1 : nthreads); return 0; }

int rtxh_scene_desc(RtxHostScene* h, RtxSceneDesc* out) {
    HostScene& s = h->s;
    memset(out, 0, sizeof(*out));
    out->objects = s.objects.data(); out->numObjects = s.objects.size();
    out->materials = s.materials.data(); out->numMaterials = s.materials.size();
    out->boxes = s.boxes.data(); out->numBoxes = s.boxes.size();
    out->textureInfo = s.textureInfo.data(); out->numTextures = s.textureInfo.size();
    out->textureData = s.textureData.data(); out->numTexels = s.textureData.size();
    out->width = s.width; out->height = s.height;
    out->useBVH = s.bvhBuilt ? 1 : 0;
    return 0;
}
RtxHostScene* rtxh_scene_from_arrays(const RtxSceneDesc* d, const RtxCamera* cam, uint32_t depth) {
    RtxHostScene* h = new RtxHostScene();
    HostScene& s = h->s;
    s.width = d->width; s.height = d->height; s.depth = depth; s.bvhBuilt = d->useBVH != 0;
    s.camera = *cam;
    s.objects.assign(d->objects, d->objects + d->numObjects);
    s.materials.assign(d->materials, d->materials + d->numMaterials);
    if (d->numBoxes) s.boxes.assign(d->boxes, d->boxes + d->numBoxes);
    s.textureInfo.assign(d->textureInfo, d->textureInfo + d->numTextures);
    s.textureData.assign(d->textureData, d->textureData + d->numTexels);
    return h;
}

int rtxh_scene_camera(RtxHostScene* h, RtxCamera* out) { *out = h->s.camera; return 0; }
int rtxh_scene_set_camera(RtxHostScene* h, const RtxCamera* in) { h->s.camera = *in; return 0; }

// src/Camera.h:51-55 (worldUp = (0,1,0), src/Camera.h:7)
void rtxh_camera_look_at(RtxCamera* cam, const float target[3]) {
    const RtxVec4 worldUp = v4(0, 1, 0);
    cam->front = unit(v4(target[0], target[1], target[2]) - cam->position);
    cam->right = unit(cross3(worldUp, cam->front));
    cam->up = unit(cross3(cam->front, cam->right));
}
// src/Camera.h:26-43 — the compound assignments and sin / cos / fmax / fmin run in double (float operands promoted)
void rtxh_camera_rotate(RtxCamera* cam, double offsetX, double offsetY) {
    const float deg2rad = 0.0174532925f;
    const RtxVec4 worldUp = v4(0, 1, 0);
    cam->yaw = (float)((double)cam->yaw + offsetX * (double)cam->rotationSpeed);
    cam->pitch = (float)((double)cam->pitch + offsetY * (double)cam->rotationSpeed);
    cam->pitch = (float)fmax((double)-89.0f, fmin((double)cam->pitch, (double)89.0f));
    const float pitch2rad = cam->pitch * deg2rad;
    const float yaw2rad = cam->yaw * deg2rad;
    cam->front.x = (float)(cos((double)yaw2rad) * cos((double)pitch2rad));
    cam->front.y = (float)sin((double)pitch2rad);
    cam->front.z = (float)(sin((double)yaw2rad) * cos((double)pitch2rad));
    cam->right = unit(cross3(worldUp, cam->front));
    cam->up = unit(cross3(cam->front, cam->right));
}
// src/Main.cpp:68-77 — sin/cos are the double overloads of float arguments
void rtxh_camera_orbit(RtxCamera* cam, float yawDeg, float pitchDeg, float radius) {
    const float deg2rad = 0.0174532925f;
    float yawInRad = yawDeg * deg2rad;
    float pitchInRad = pitchDeg * deg2rad;
    float xNew = (float)(radius * sin((double)yawInRad));
    float yNew = (float)(500.0f * cos((double)pitchInRad) + 500.0f);
    float zNew = (float)(radius * cos((double)yawInRad));
    cam->position = v4(xNew, yNew, zNew);
    const float center[3] = {0, 0, 0};
    rtxh_camera_look_at(cam, center);
}

// ---- blob ---------------------------------------------------------------------
// "RTXSCN01" | u32 width,height,depth,useBVH | u64 nObj,nMat,nBox,nTex,nTexel |
// camera(96) | objects | materials | boxes | textureInfo | textureData
struct BlobHeader {
    char magic[8];
    uint32_t width, height, depth, useBVH;
    uint64_t nObj, nMat, nBox, nTex, nTexel;
};

int rtxh_scene_save_blob(RtxHostScene* h, const char* path) {
    HostScene& s = h->s;
    FILE* f = fopen(path, "wb");
    if (!f) return fail(std::string("cannot write ") + path);
    BlobHeader hd; memset(&hd, 0, sizeof(hd));
    memcpy(hd.magic, "RTXSCN01", 8);
    hd.width = s.width; hd.height = s.height; hd.depth = s.depth; hd.useBVH = s.bvhBuilt;
    hd.nObj = s.objects.size(); hd.nMat = s.materials.size(); hd.nBox = s.boxes.size();
    hd.nTex = s.textureInfo.size(); hd.nTexel = s.textureData.size();
    bool ok = fwrite(&hd, sizeof(hd), 1, f) == 1 && fwrite(&s.camera, sizeof(RtxCamera), 1, f) == 1;
    auto put = [&](const void* p, size_t bytes) { if (bytes && fwrite(p, 1, bytes, f) != bytes) ok = false; };
    put(s.objects.data(), s.objects.size() * sizeof(RtxObject));
    put(s.materials.data(), s.materials.size() * sizeof(RtxMaterial));
    put(s.boxes.data(), s.boxes.size() * sizeof(RtxBoundingBox));
    put(s.textureInfo.data(), s.textureInfo.size() * sizeof(RtxTexture));
    put(s.textureData.data(), s.textureData.size() * sizeof(uint32_t));
    fclose(f);
    return ok ? 0 : fail("short write");
}

RtxHostScene* rtxh_scene_load_blob(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fail(std::string("cannot open ") + path); return nullptr; }
    BlobHeader hd;
    RtxHostScene* h = new RtxHostScene();
    HostScene& s = h->s;
    bool ok = fread(&hd, sizeof(hd), 1, f) == 1 && memcmp(hd.magic, "RTXSCN01", 8) == 0 && fread(&s.camera, sizeof(RtxCamera), 1, f) == 1;
    if (ok) {
        s.width = hd.width; s.height = hd.height; s.depth = hd.depth; s.bvhBuilt = hd.useBVH != 0;
        s.objects.resize(hd.nObj); s.materials.resize(hd.nMat); s.boxes.resize(hd.nBox);
        s.textureInfo.resize(hd.nTex); s.textureData.resize(hd.nTexel);
        auto get = [&](void* p, size_t bytes) { if (bytes && fread(p, 1, bytes, f) != bytes) ok = false; };
        get(s.objects.data(), hd.nObj * sizeof(RtxObject));
        get(s.materials.data(), hd.nMat * sizeof(RtxMaterial));
        get(s.boxes.data(), hd.nBox * sizeof(RtxBoundingBox));
        get(s.textureInfo.data(), hd.nTex * sizeof(RtxTexture));
        get(s.textureData.data(), hd.nTexel * sizeof(uint32_t));
    }
    fclose(f);
    if (!ok) { delete h; fail(std::string("bad scene blob ") + path); return nullptr; }
    return h;
}

} // extern "C"
