// host_scene.h — host-side scene container (the five arrays of
// src/RenderingContext.h:31-40 plus camera / resolution) and the small float4
// helpers the ingest code shares. Host only; no HIP.
#pragma once
#include "../../../include/rtx.h"
#include "../../../include/rtx_host.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace rtxh {

// 4-lane arithmetic with the lane semantics of the reference's SSE-backed
// Vector3 (src/Vector3.h:93-174): every op acts on x,y,z and w.
inline RtxVec4 v4(float x, float y, float z, float w = 0.0f) { RtxVec4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
inline RtxVec4 operator+(RtxVec4 a, RtxVec4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline RtxVec4 operator-(RtxVec4 a, RtxVec4 b) { return v4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline RtxVec4 operator*(RtxVec4 a, RtxVec4 b) { return v4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline RtxVec4 operator*(RtxVec4 a, float s) { return v4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline RtxVec4 div_scalar(RtxVec4 a, float s) { return a * (1.0f / s); }           // src/Vector3.h:152-162
inline float dot3(RtxVec4 a, RtxVec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // src/Vector3.h:34-36
inline RtxVec4 cross3(RtxVec4 a, RtxVec4 b) {                                       // src/Vector3.h:42-47
    return v4(a.y * b.z - b.y * a.z, -(a.x * b.z - b.x * a.z), a.x * b.y - b.x * a.y);
}
inline RtxVec4 unit(RtxVec4 a) { return div_scalar(a, sqrtf(dot3(a, a))); }          // src/Vector3.h:184-199
inline float lane_min(float a, float b) { return a < b ? a : b; }                    // _mm_min_ps rule
inline float lane_max(float a, float b) { return a > b ? a : b; }                    // _mm_max_ps rule
inline RtxVec4 vmin(RtxVec4 a, RtxVec4 b) { return v4(lane_min(a.x, b.x), lane_min(a.y, b.y), lane_min(a.z, b.z), lane_min(a.w, b.w)); }
inline RtxVec4 vmax(RtxVec4 a, RtxVec4 b) { return v4(lane_max(a.x, b.x), lane_max(a.y, b.y), lane_max(a.z, b.z), lane_max(a.w, b.w)); }

struct HostScene {
    uint32_t width = 0, height = 0, depth = 480;
    bool bvhBuilt = false;
    RtxCamera camera;
    std::vector<RtxObject> objects;
    std::vector<RtxMaterial> materials;
    std::vector<RtxBoundingBox> boxes;
    std::vector<RtxTexture> textureInfo;
    std::vector<uint32_t> textureData;
    std::map<std::string, uint32_t> materialNames;

    // MaterialBuilder keeps clearcoatThickness / clearcoatRoughness from the
    // previous material (src/MaterialBuilder.cpp:10-33 never resets them).
    float stickyClearcoatThickness = 0.0f, stickyClearcoatRoughness = 0.0f;
};

void set_error(const std::string& msg);

RtxObject blank_object();
void add_plane(HostScene& s, RtxVec4 position, const RtxVec4 normals[3], RtxVec4 scales, int materialID);
void add_disk(HostScene& s, RtxVec4 position, RtxVec4 normal, float radius, int materialID);
void add_cube(HostScene& s, RtxVec4 position, RtxVec4 scales, int materialID);
bool load_scn(HostScene& s, const std::string& path);
bool load_mtl(HostScene& s, const std::string& path);
bool load_obj(HostScene& s, const std::string& path);
int  add_texture(HostScene& s, const uint32_t* texels, int32_t w, int32_t h, uint32_t checksum);
void build_bvh(HostScene& s, int nthreads);
void add_mesh(HostScene& s, const float* vertices, const float* normals, uint64_t nv,
              const uint32_t* indices, const int32_t* materialIDs, uint64_t ntri, int materialID0);
void add_heightfield(HostScene& s, uint64_t ntri, uint32_t seed, float extentU, float extentV, float amplitude,
                     float tiltDeg, float cx, float cy, float cz, const int32_t* mats, int32_t nmats);
bool write_heightfield_obj(const HostScene& s, const std::string& path, uint64_t ntri, uint32_t seed, float extentU, float extentV,
                           float amplitude, float tiltDeg, float cx, float cy, float cz, const int32_t* mats, int32_t nmats);
const std::string& get_error();

} // namespace rtxh
