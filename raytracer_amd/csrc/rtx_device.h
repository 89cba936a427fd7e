// rtx_device.h — device-side data layout and arithmetic of the HIP backend.
//
// HBM layout (all arrays 64-byte aligned, read-only during a frame):
//   inner[]   64 B / inner node : the reference's topology (pre-order), both children's AABBs + both child refs:
//             q0 = (lo0.x lo0.y lo0.z hi0.x) q1 = (hi0.y hi0.z lo1.x lo1.y)
//             q2 = (lo1.z hi1.x hi1.y hi1.z) q3 = (ref0 ref1 - -)      ref>=0 inner, ref<0 leaf ~ref
//   fast[]   128 B / 4-wide node: the renderer's own SAH tree over the same leaves (rtx_build.h Node4):
//             lo.x[4] hi.x[4] lo.y[4] hi.y[4] lo.z[4] hi.z[4] ref[4] pad[4]; an empty slot holds a NaN box
//   qnodes[]  64 B / 4-wide node: the same tree with byte boxes (rtx_build.h NodeQ, 48 bytes used): RTX_TRAVERSE_WIDE
//   leaf[]    64 B / primitive  : q0 = (A.x A.y A.z e1.x) q1 = (e1.y e1.z e2.x e2.y) q2 = (e2.z rank<<3|type lo.x lo.y)
//             q3 = (lo.z hi.x hi.y hi.z): the primitive, its tie-break rank and its own exact box.
//             sphere: A = centre, e1.x = radius. The ordered walks read the first 40 bytes (three lane loads).
//   shade[]  128 B / primitive  : q0 q1 as in leaf[], q2 = (e2.z objectID boxID type) q3 = (B.x B.y C.x C.y),
//                                 q4 = (n0.x n0.y n0.z n1.x) q5 = (n1.y n1.z n2.x n2.y)
//                                 q6 = (n2.z materialID - -) q7 unused: one line per shaded hit
//   materials 128 B each, reference layout (src/Material.h:7-25)
// One inner visit = one 64-byte line (the reference touches three: the node and
// both children, src/ThreadedShader.cpp:355,381,388); one leaf test = one line
// (the reference copies a 256-byte Object, :362). e1 = B-A and e2 = C-A are
// stored pre-subtracted: the reference forms exactly these two differences first
// (:298-299), so the rounding is identical.
//
// Arithmetic contract: every float op below is a separate IEEE operation
// (compiled with -ffp-contract=off, no fast-math), in the reference's order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTX_PI_F 3.1415926535f
#define RTX_TEX_PAD 8194   // zero words behind the last texel (src/Shading.h:170-173 reads past the end)

struct F4 { float x, y, z, w; };

__device__ __forceinline__ F4 mk4(float x, float y, float z, float w = 0.0f) { F4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
__device__ __forceinline__ F4 from4(float4 v) { return mk4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ F4 add4(F4 a, F4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ F4 sub4(F4 a, F4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ F4 mul4(F4 a, F4 b) { return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ F4 mul4(F4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }
// _mm_min_ps / _mm_max_ps NaN rule: second operand when either is NaN
__device__ __forceinline__ float sse_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float sse_max(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ F4 clamp01_sse(F4 a) {
    return mk4(sse_max(sse_min(a.x, 1.0f), 0.0f), sse_max(sse_min(a.y, 1.0f), 0.0f),
               sse_max(sse_min(a.z, 1.0f), 0.0f), sse_max(sse_min(a.w, 1.0f), 0.0f));
}
__device__ __forceinline__ float dot3(F4 a, F4 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F4 cross3(F4 a, F4 b) {
    return mk4(a.y * b.z - b.y * a.z, -(a.x * b.z - b.x * a.z), a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ F4 divs4(F4 a, float s) { return mul4(a, 1.0f / s); }
__device__ __forceinline__ F4 normalize4(F4 a) { return divs4(a, sqrtf(dot3(a, a))); }
// libm fmax/fmin: the non-NaN operand wins
__device__ __forceinline__ float clamp01_libm(float t) { return fmaxf(0.0f, fminf(t, 1.0f)); }
__device__ __forceinline__ F4 lerp4(F4 a, F4 b, float tClamped) { return add4(a, mul4(sub4(b, a), tClamped)); }

// src/Random.cpp:3-7
__device__ __forceinline__ float rtx_rand(uint32_t& seed) {
    seed = seed * 747796405u + 2891336453u;
    uint32_t word = ((seed >> 17) ^ seed) * 277803737u;
    return (float)((word >> 22u) ^ word) / 4294967296.0f;
}

struct DevScene {
    const float4* fast;   // own 4-wide SAH tree over the same leaves (may be NULL): 8 float4 per node (rtx_build.h Node4)
    int fastRoot;
    const uint4* qnodes;  // quantised 4-wide nodes of RTX_TRAVERSE_WIDE (rtx_build.h NodeQ, 4 uint4 per node; root = node 0), or NULL
    const float4* inner;
    const float4* leaf;
    const float4* shade;
    const float4* materials;
    const int4* texInfo;
    const uint32_t* texels;
    long long numTexelsPadded;
    int rootRef;        // >=0 inner index, <0 ~leaf
    int numLeaves;
    int numInner;
    int numMaterials;
    int numTextures;
    int skyTex;
    int useBVH;
    int stackDepth;     // stack entries a lane can need (tree depth + 2)
    int2* spill;        // [stackDepth][threads of the trace grid]: stack entries below the LDS window
    float cullSlack;    // relative slack of distance culling (rtx_traverse.h); +inf = never cull
};

struct DevFrame {
    // camera (src/Camera.h) with the per-frame constant tanHalfFOV evaluated on the
    // host with the host libm (tan is the double overload, src/Camera.h:58)
    float front[3], up[3], right[3], pos[4];
    float aspect, nearView, tanHalfFOV;
    uint32_t frameCounter;
    float lerpScale;    // 1.0f / (frameCounter + 1), src/ThreadedShader.cpp:179
    int width, height;
    int tileW, tileH, tilesX;
    int numLocalTiles;
    int numSlots;       // numLocalTiles * tileW * tileH
};

// Frames that share one set of launches (rtx_api.hip "frame batches"): a rank that owns 1/N of the tiles
// renders up to RTX_MAX_BATCH queued frames together, so its ray queues are as long as a whole frame's.
// Batch slot s = (tile * count + frame) * tilePixels + pixel-in-tile: the frames of one tile are neighbours,
// so a queue segment is still an image region (and the same few nodes serve all of a tile's frames).
#ifndef RTX_MAX_BATCH
#define RTX_MAX_BATCH 32
#endif
#define RTX_AHEAD_MAX 8       // most frames of a set started ahead (rtx_api.hip look-ahead): k_shade counts their hits per frame
struct DevBatch {
    DevFrame fr[RTX_MAX_BATCH];
    int count;
};
static_assert(RTX_AHEAD_MAX <= RTX_MAX_BATCH, "a set started ahead is a batch");
