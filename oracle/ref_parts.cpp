/*
 * ref_parts.cpp — thin extern "C" shims around the part of the reference's
 * hot path that compiles standalone in this image.
 *
 * TEST INFRASTRUCTURE (checker only). Built by oracle/Makefile into
 * oracle/_ref/libref_parts.so, which is git-ignored. The reference sources are
 * compiled WHERE THEY LIE under $(REF)/src — nothing is copied into the repo:
 *     Shading.h, Camera.h, BoundingBox.h, Vector3.h, Color.h, Material.h, Object.h, Texture.h,
 *     SpatialType.h, Random.h, BitmapReader.h  (included below by path)
 *     Random.cpp, BitmapReader.cpp             (compiled as their own translation units)
 *
 * What is NOT here, and why: ThreadedShader.cpp, BVHTree.cpp and the
 * serializers all include RenderingContext.h, which includes <GL/glew.h> and
 * <GLFW/glfw3.h> (src/RenderingContext.h:10-11). Neither header exists in this
 * image and stand-ins are not allowed, so those files are unbuildable here;
 * rt_oracle.cpp restates them and is pinned by recorded reference outputs
 * instead (tests/test_oracle.py).
 */
#include "Shading.h"   // -I$(REF)/src
#include "Camera.h"
#include "Random.h"
#include "BitmapReader.h"
#include "BoundingBox.h"

#include <cstddef>

#include <cstdint>
#include <cstring>

static Vector3 V(const float* p) { return Vector3(p[0], p[1], p[2]); }

extern "C" {

float ref_rand(uint32_t* seed) { return Random::Rand(*seed); }

void ref_pixel_position(const void* camera96, int x, int y, int w, int h, float* out4) {
    Camera cam; static_assert(sizeof(Camera) == 96, "Camera layout");
    memcpy((void*)&cam, camera96, 96);
    Vector3 p = cam.CalculatePixelPosition(x, y, w, h);
    out4[0] = p.x; out4[1] = p.y; out4[2] = p.z; out4[3] = p.w;
}

void ref_shading_terms(const void* material128, const float* n3, const float* l3, const float* v3, const float* h3, const float* c3, float* out9) {
    Material m; static_assert(sizeof(Material) == 128, "Material layout");
    memcpy((void*)&m, material128, 128);
    Vector3 n = V(n3), l = V(l3), v = V(v3), h = V(h3);
    out9[0] = Shading::SchlickFresnel(c3[0]);
    out9[1] = Shading::DiffuseBRDF(c3[0], c3[1], m);
    out9[2] = Shading::SpecularBSDF(n, l, v, h, m);
    out9[3] = Shading::SpecularTransmissionBSDF(l, v, h, m).R;
    out9[4] = Shading::ClearcoatBRDF(v, l, h, m).R;
    Color sh = Shading::Sheen(c3[2], m);
    out9[5] = sh.R; out9[6] = sh.G; out9[7] = sh.B; out9[8] = sh.A;
}

void ref_get_texture_pixel(const uint32_t* texels, const void* object256, const void* texture16, const float* p3, const float* n3, float* out4) {
    Object o; static_assert(sizeof(Object) == 256, "Object layout");
    memcpy((void*)&o, object256, 256);
    Texture t; static_assert(sizeof(Texture) == 16, "Texture layout");
    memcpy((void*)&t, texture16, 16);
    Color c = Shading::GetTexturePixel(texels, o, t, V(p3), V(n3));
    out4[0] = c.R; out4[1] = c.G; out4[2] = c.B; out4[3] = c.A;
}

void ref_color_sample(const uint32_t* texels, float u, float v, const void* texture16, float* out4) {
    Texture t; memcpy((void*)&t, texture16, 16);
    Color c = Shading::ColorSample(texels, u, v, t.width, t.height, t.offset);
    out4[0] = c.R; out4[1] = c.G; out4[2] = c.B; out4[3] = c.A;
}

// Vector3 / Color primitives (SSE paths)
void ref_vec_ops(const float* a4, const float* b4, float s, float* out /* 8 x 4 */) {
    Vector3 a, b; memcpy((void*)&a, a4, 16); memcpy((void*)&b, b4, 16);
    Vector3 r[8];
    r[0] = a + b; r[1] = a - b; r[2] = a * b; r[3] = a * s; r[4] = a / s;
    r[5] = Vector3::Minimal(a, b); r[6] = Vector3::Maximal(a, b); r[7] = Vector3::Clamp(a);
    memcpy(out, r, sizeof(r));
}
void ref_vec_geom(const float* a4, const float* b4, float t, float* out /* dot, mag, cross xyz w, normalize xyzw, lerp xyzw */) {
    Vector3 a, b; memcpy((void*)&a, a4, 16); memcpy((void*)&b, b4, 16);
    out[0] = Vector3::DotProduct(a, b);
    out[1] = a.Magnitude();
    Vector3 c = Vector3::CrossProduct(a, b); memcpy(out + 2, &c, 16);
    Vector3 n = a.Normalize(); memcpy(out + 6, &n, 16);
    float tt = t; Vector3 l = Vector3::Lerp(a, b, tt); memcpy(out + 10, &l, 16);
    out[14] = tt;
}
void ref_color_ops(const float* a4, const float* b4, float t, float* out /* 3 x 4 */) {
    Color a, b; memcpy((void*)&a, a4, 16); memcpy((void*)&b, b4, 16);
    Color r[3];
    r[0] = Color::Lerp(a, b, t); r[1] = Color::Clamp(a); r[2] = a * b;
    memcpy(out, r, sizeof(r));
}

// Camera::LookAt / Camera::Rotate (src/Camera.h:26-55) on a 96-byte camera, in place
void ref_camera_look_at(void* camera96, const float* target3) {
    Camera cam; memcpy((void*)&cam, camera96, 96);
    cam.LookAt(V(target3));
    memcpy(camera96, (void*)&cam, 96);
}
void ref_camera_rotate(void* camera96, double offsetX, double offsetY) {
    Camera cam; memcpy((void*)&cam, camera96, 96);
    cam.Rotate(offsetX, offsetY);
    memcpy(camera96, (void*)&cam, 96);
}

// BoundingBox (src/BoundingBox.h:7-48): layout as include/rtx.h RtxBoundingBox states it, and the default box
static_assert(sizeof(BoundingBox) == 64 && alignof(BoundingBox) == 64, "BoundingBox layout");
static_assert(offsetof(BoundingBox, objectID) == 0 && offsetof(BoundingBox, parentID) == 4 && offsetof(BoundingBox, leftID) == 8 &&
              offsetof(BoundingBox, rightID) == 12 && offsetof(BoundingBox, minimalPosition) == 16 && offsetof(BoundingBox, maximalPosition) == 32,
              "BoundingBox offsets");
static_assert(offsetof(Camera, front) == 0 && offsetof(Camera, up) == 16 && offsetof(Camera, right) == 32 && offsetof(Camera, position) == 48 &&
              offsetof(Camera, aspectRatio) == 72 && offsetof(Camera, nearView) == 76 && offsetof(Camera, fov) == 84 &&
              offsetof(Camera, pitch) == 88 && offsetof(Camera, yaw) == 92, "Camera offsets");
static_assert(offsetof(Object, type) == 0 && offsetof(Object, radius) == 4 && offsetof(Object, position) == 16 && offsetof(Object, normals) == 32 &&
              offsetof(Object, vertices) == 80 && offsetof(Object, uvs) == 128 && offsetof(Object, materialID) == 176, "Object offsets");
void ref_box_layout(int32_t* out8) {      // sizes / offsets for a run-time comparison with the ctypes / numpy mirrors
    out8[0] = (int32_t)sizeof(BoundingBox); out8[1] = (int32_t)offsetof(BoundingBox, objectID); out8[2] = (int32_t)offsetof(BoundingBox, parentID);
    out8[3] = (int32_t)offsetof(BoundingBox, leftID); out8[4] = (int32_t)offsetof(BoundingBox, rightID);
    out8[5] = (int32_t)offsetof(BoundingBox, minimalPosition); out8[6] = (int32_t)offsetof(BoundingBox, maximalPosition); out8[7] = (int32_t)alignof(BoundingBox);
}
// default-constructed box, then Expand()ed by `n` boxes given as 64-byte records; SurfaceArea() of the result
float ref_box_expand(const void* boxes64, int n, void* out64) {
    BoundingBox acc;
    for (int i = 0; i < n; ++i) { BoundingBox b; memcpy((void*)&b, (const char*)boxes64 + 64 * (size_t)i, 64); acc.Expand(b); }
    memcpy(out64, (void*)&acc, 64);
    return acc.SurfaceArea();
}

// BitmapReader::ReadFile (src/BitmapReader.cpp:70-137): texels as the reference unpacks them
// (R | G<<8 | B<<16 | 255<<24, rows in file order). Returns the texel count, 0 if the reader
// rejects the file; copies at most `cap` texels. The checksum is not exported: the reference
// accumulates it into an uninitialised field (src/BitmapReader.cpp:129-134,168).
uint64_t ref_read_bmp(const char* path, uint32_t* width, uint32_t* height, uint32_t* texels, uint64_t cap) {
    Image im = BitmapReader::ReadFile(path);
    if (!im.data) return 0;
    *width = im.width; *height = im.height;
    const uint64_t n = (uint64_t)im.width * im.height;
    memcpy(texels, im.data, (n < cap ? n : cap) * sizeof(uint32_t));
    delete[] im.data;
    return n;
}

} // extern "C"
