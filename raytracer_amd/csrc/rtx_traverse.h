// rtx_traverse.h — closest-hit query over the flattened BVH (or the linear list).
//
// Semantics reproduced (src/ThreadedShader.cpp:337-395, :209-237):
//   * a leaf is tested iff every box on its root path passes AABBIntersection
//     (the reference never culls by distance; the root itself is not tested);
//   * a hit is accepted iff t < best && t > 0.01f; the reference visits leaves in
//     descending boxID, so among equal t the LARGEST leaf boxID wins (BVH) /
//     the LOWEST object id wins (linear list);
//   * t is computed with the reference's operation sequence.
// RTX_TRAVERSE_EXACT visits exactly that leaf set. RTX_TRAVERSE_CULLED orders
// children near-first and drops a child whose slab entry distance lies beyond
// the current best hit by more than a relative slack; every box on the way down
// is still tested with the reference's own test, so the visited set is a subset
// of the reference's and the winner is the same unless a triangle's computed t
// undercuts its own box's computed entry distance by more than the slack
// (DESIGN.md "Culling"). Lanes whose direction has a zero / non-finite component
// (0*inf = NaN in the slab test) never cull.
#pragma once
#include "rtx_device.h"

#define RTX_NO_CHILD ((int)0x80000000)
// Distance culling drops a subtree whose entry distance exceeds best.t * (1 + slack). The slack absorbs the
// rounding by which a primitive's computed t can undercut its own box's computed entry distance; for a
// triangle that error grows like 2^-20 * shape / |cos(incidence)| (shape = longest edge^2 / (2 area)), so no
// finite slack is safe for every grazing ray. rtx_create chooses DevScene::cullSlack per scene: 2^-6 when the
// triangles are well shaped (measured: 0 of 46 M bounce rays then differ from the unculled result), +inf —
// no culling at all — when more than 0.1 % of them are slivers.
#define RTX_CULL_SLACK_DEFAULT 0.015625f   // 2^-6

struct Hit { float t; int leaf; int box; };

// src/ThreadedShader.cpp:275-289; returns pass flag and tNear
__device__ __forceinline__ bool slab_test(float ox, float oy, float oz, float ix, float iy, float iz,
                                          float lx, float ly, float lz, float hx, float hy, float hz, float& tNearOut) {
    float tminx = (lx - ox) * ix, tminy = (ly - oy) * iy, tminz = (lz - oz) * iz;
    float tmaxx = (hx - ox) * ix, tmaxy = (hy - oy) * iy, tmaxz = (hz - oz) * iz;
    float t1x = sse_min(tminx, tmaxx), t1y = sse_min(tminy, tmaxy), t1z = sse_min(tminz, tmaxz);
    float t2x = sse_max(tminx, tmaxx), t2y = sse_max(tminy, tmaxy), t2z = sse_max(tminz, tmaxz);
    float tNear = fmaxf(t1x, fmaxf(t1y, t1z));
    float tFar = fminf(t2x, fminf(t2y, t2z));
    tNearOut = tNear;
    return tNear <= tFar && tFar > 0.0f;
}

// src/ThreadedShader.cpp:291-321 with e1, e2 pre-subtracted
__device__ __forceinline__ float intersect_triangle(float ox, float oy, float oz, float dx, float dy, float dz,
                                                    float ax, float ay, float az, float e1x, float e1y, float e1z,
                                                    float e2x, float e2y, float e2z) {
    // (the early returns are kept as branches: most leaf tests fail on u, and a straight-line
    // select form measured 7 % slower)
    // normal = cross(dir, e2)
    float nx = dy * e2z - e2y * dz;
    float ny = -(dx * e2z - e2x * dz);
    float nz = dx * e2y - e2x * dy;
    float det = e1x * nx + e1y * ny + e1z * nz;
    if (fabsf(det) < 1e-6f) return -1.0f;
    float f = 1.0f / det;
    float sx = ox - ax, sy = oy - ay, sz = oz - az;
    float u = f * (sx * nx + sy * ny + sz * nz);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    // q = cross(s, e1)
    float qx = sy * e1z - e1y * sz;
    float qy = -(sx * e1z - e1x * sz);
    float qz = sx * e1y - e1x * sy;
    float v = f * (dx * qx + dy * qy + dz * qz);
    if (v < 0.0f || (u + v) > 1.0f) return -1.0f;
    return f * (e2x * qx + e2y * qy + e2z * qz);
}

// src/ThreadedShader.cpp:323-335 (NaN when the ray misses; rejected by the caller's compares)
__device__ __forceinline__ float intersect_sphere(float ox, float oy, float oz, float dx, float dy, float dz,
                                                  float cx, float cy, float cz, float radius) {
    float ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
    float b = ocx * dx + ocy * dy + ocz * dz;
    float c = (ocx * ocx + ocy * ocy + ocz * ocz) - radius * radius;
    float delta = b * b - c;
    float sq = sqrtf(delta);
    float t1 = (-b - sq);
    float t2 = (-b + sq);
    return fminf(t1, t2);
}

// Leaf record (rtx_device.h): q0 = (A.xyz e1.x) q1 = (e1.yz e2.xy) q2 = (e2.z boxID<<3|type - -) q3 unused.
// What a primitive test needs sits in the first 40 bytes: three lane loads (16 + 16 + 8 bytes) — the
// traversal kernels are bound by the number of lane loads the vector L1 accepts.
__device__ __forceinline__ float intersect_leaf(const float4* __restrict__ leafBase, int leafIdx,
                                                float ox, float oy, float oz, float dx, float dy, float dz,
                                                int& boxOut) {
    const float4* L = leafBase + (size_t)leafIdx * 4;
    float4 q0 = L[0], q1 = L[1];
    const float2 q2 = *(const float2*)(L + 2);
    const float e2z = q2.x;
    const int tag = __float_as_int(q2.y);
    boxOut = tag >> 3;
    if ((tag & 7) == 4 /* RTX_TRIANGLE */)
        return intersect_triangle(ox, oy, oz, dx, dy, dz, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, e2z);
    return intersect_sphere(ox, oy, oz, dx, dy, dz, q0.x, q0.y, q0.z, q0.w);
}

// BVH closest hit. `stack` is this lane's LDS stack: entry k at stack[k * stride].
template <bool EXACT, bool COUNT>
__device__ __forceinline__ Hit traverse_bvh(const DevScene& sc, float ox, float oy, float oz, float dx, float dy, float dz,
                                            int* stack, int stride, uint32_t& innerCount, uint32_t& leafCount) {
    Hit best; best.t = INFINITY; best.leaf = -1; best.box = -1;
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    // Reference topology, reference arithmetic (SSE min/max NaN rule). EXACT = no distance culling.
    const bool cull = !EXACT;
    const float4* __restrict__ nodes = sc.inner;
    int sp = 0;
    int cur = sc.rootRef;
    const int maxSp = sc.stackDepth;
    for (;;) {
        if (cur >= 0) {
            const float4* N = nodes + (size_t)cur * 4;
            float4 q0 = N[0], q1 = N[1], q2 = N[2], q3 = N[3];
            if (COUNT) innerCount++;
            float tn0, tn1;
            bool p0 = slab_test(ox, oy, oz, ix, iy, iz, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tn0);
            bool p1 = slab_test(ox, oy, oz, ix, iy, iz, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tn1);
            int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
            p0 = p0 && (c0 != RTX_NO_CHILD);   // absent child (-1 in the reference's boxes, src/ThreadedShader.cpp:380,387)
            p1 = p1 && (c1 != RTX_NO_CHILD);
            if (cull) {
                float limit = best.t + best.t * sc.cullSlack;   // inf while nothing is hit
                p0 = p0 && !(tn0 > limit);
                p1 = p1 && !(tn1 > limit);
            }
            if (p0 && p1) {
                // near child first; the far one waits on the stack
                bool swap = tn1 < tn0;
                int nearC = swap ? c1 : c0, farC = swap ? c0 : c1;
                if (sp < maxSp) { stack[sp * stride] = farC; sp++; }
                cur = nearC;
                continue;
            } else if (p0) { cur = c0; continue; }
            else if (p1) { cur = c1; continue; }
        } else {
            int leafIdx = ~cur, box;
            float t = intersect_leaf(sc.leaf, leafIdx, ox, oy, oz, dx, dy, dz, box);
            if (COUNT) leafCount++;
            if (t > 0.01f && (t < best.t || (t == best.t && best.leaf >= 0 && box > best.box))) { best.t = t; best.leaf = leafIdx; best.box = box; }
        }
        if (sp == 0) break;
        sp--;
        cur = stack[sp * stride];
    }
    return best;
}

// A lane is "clean" when its slab arithmetic cannot produce NaN: finite origin, finite non-zero
// direction components with finite reciprocals (boxes are NaN-free whenever the own tree exists).
__device__ __forceinline__ bool lane_is_clean(float ox, float oy, float oz, float dx, float dy, float dz) {
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    return (dx != 0.0f) && (dy != 0.0f) && (dz != 0.0f) && isfinite(ix) && isfinite(iy) && isfinite(iz) &&
           isfinite(ox) && isfinite(oy) && isfinite(oz);
}

// Slab test for clean lanes: without NaNs, _mm_min_ps/_mm_max_ps and libm fmax/fmin are plain
// min/max, so the hardware min/max give the same values (zero signs may differ; they only feed
// comparisons). Same subtractions and multiplications as the reference.
__device__ __forceinline__ bool slab_test_clean(float ox, float oy, float oz, float ix, float iy, float iz,
                                                float lx, float ly, float lz, float hx, float hy, float hz, float& tNearOut) {
    float tminx = (lx - ox) * ix, tminy = (ly - oy) * iy, tminz = (lz - oz) * iz;
    float tmaxx = (hx - ox) * ix, tmaxy = (hy - oy) * iy, tmaxz = (hz - oz) * iz;
    float tNear = fmaxf(fminf(tminx, tmaxx), fmaxf(fminf(tminy, tmaxy), fminf(tminz, tmaxz)));
    float tFar = fminf(fmaxf(tminx, tmaxx), fminf(fmaxf(tminy, tmaxy), fmaxf(tminz, tmaxz)));
    tNearOut = tNear;
    return tNear <= tFar && tFar > 0.0f;
}

#define RTX_DONE ((int)0x80000000)
#ifndef RTX_LDS_WINDOW
#define RTX_LDS_WINDOW 8
#endif
//    // stack entries per lane kept in LDS; older ones live in the HBM spill buffer

// Per-lane traversal stack of (node ref, entry distance) pairs. The newest RTX_LDS_WINDOW entries
// sit in LDS (slot = index % window; 4 KB per wave, so LDS never limits occupancy); when the window
// is full the oldest entry moves to a per-thread spill column in HBM, [entry][thread], and is read
// from there when the stack unwinds below the window. Near-first traversal keeps almost all
// traffic inside the window.
extern __shared__ int rtx_lds[];   // the kernels' dynamic LDS; addressed by index so the accesses stay ds_* instructions

struct LaneStack {
    int lds;               // this lane's column: ref k at rtx_lds[lds + (k % W) * stride], tNear at rtx_lds[lds + (W + k % W) * stride]
    int stride;
    int2* spill;           // this thread's column: entry k at spill[k * spillStride]
    long long spillStride;
    int sp, base;          // entries [base, sp) are in LDS, [0, base) in the spill buffer

    __device__ __forceinline__ void push(int ref, float tn) {
        const int slot = sp & (RTX_LDS_WINDOW - 1);
        if (sp - base == RTX_LDS_WINDOW) {
            spill[(long long)base * spillStride] = make_int2(rtx_lds[lds + slot * stride], rtx_lds[lds + (RTX_LDS_WINDOW + slot) * stride]);
            base++;
        }
        rtx_lds[lds + slot * stride] = ref;
        rtx_lds[lds + (RTX_LDS_WINDOW + slot) * stride] = __float_as_int(tn);
        sp++;
    }
    __device__ __forceinline__ void pop(int& ref, float& tn) {
        sp--;
        if (sp >= base) {
            const int slot = sp & (RTX_LDS_WINDOW - 1);
            ref = rtx_lds[lds + slot * stride]; tn = __int_as_float(rtx_lds[lds + (RTX_LDS_WINDOW + slot) * stride]);
        } else {
            const int2 e = spill[(long long)sp * spillStride];
            ref = e.x; tn = __int_as_float(e.y); base = sp;
        }
    }
};

#define RTX_NEED_POP ((int)0x80000001)

#define RTX_CSWAP(ta, ra, tb, rb) { const bool s_ = tb < ta; const float tt_ = s_ ? ta : tb; const int rr_ = s_ ? ra : rb; \
                                    ta = s_ ? tb : ta; ra = s_ ? rb : ra; tb = tt_; rb = rr_; }

typedef float rtx_f4 __attribute__((ext_vector_type(4)));
typedef float rtx_f2 __attribute__((ext_vector_type(2)));

// The ray as the 4-wide step wants it: origin and reciprocal direction packed in three register pairs
// (packed FP32 instructions broadcast either half of a pair), and for each axis the byte offset (0 or
// 16) of the near plane inside a node's lo/hi block of that axis; the far plane sits at offset ^ 16
// (rtx_build.h Node4: lo.x[4] hi.x[4] lo.y[4] hi.y[4] lo.z[4] hi.z[4] ref[4]). With min <= max on
// every axis (checked by rtx_create) and no NaN, min((lo-o)*i, (hi-o)*i) IS the product taken with the
// plane the sign of i selects — rounding is monotonic — so choosing the plane first gives the
// reference's value with a third fewer instructions.
struct WideRay {
    rtx_f2 a, b, c;          // (ox, oy) (oz, ix) (iy, iz)
    uint32_t nx, ny, nz;
};
__device__ __forceinline__ WideRay wide_ray(float ox, float oy, float oz, float ix, float iy, float iz) {
    WideRay w;
    w.a = rtx_f2{ox, oy}; w.b = rtx_f2{oz, ix}; w.c = rtx_f2{iy, iz};
    w.nx = (__float_as_uint(ix) >> 27) & 16u;
    w.ny = (__float_as_uint(iy) >> 27) & 16u;
    w.nz = (__float_as_uint(iz) >> 27) & 16u;
    return w;
}

// (v - o) * i on two children at once; o and i are the low (0) or high (1) half of a register pair.
// v_pk_add_f32 with a negated operand and v_pk_mul_f32 round exactly like v_sub_f32 / v_mul_f32.
template <int OH, int IH>
__device__ __forceinline__ rtx_f2 pk_plane_distance(rtx_f2 v, rtx_f2 oPair, rtx_f2 iPair) {
    rtx_f2 d, t;
    if (OH == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(v), "v"(oPair));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(v), "v"(oPair));
    if (IH == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(d), "v"(iPair));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t) : "v"(d), "v"(iPair));
    return t;
}

// Tail of a node step: the children that passed, ordered by entry distance; all but the nearest are
// pushed (farthest first), the nearest is returned — or RTX_NEED_POP when none passed.
__device__ __forceinline__ int order_and_push(float t0, float t1, float t2, float t3, bool p0, bool p1, bool p2, bool p3,
                                              const int4& rf, LaneStack& st) {
    int r0 = rf.x, r1 = rf.y, r2 = rf.z, r3 = rf.w;
    t0 = p0 ? t0 : INFINITY; t1 = p1 ? t1 : INFINITY; t2 = p2 ? t2 : INFINITY; t3 = p3 ? t3 : INFINITY;
    r0 = p0 ? r0 : RTX_NEED_POP; r1 = p1 ? r1 : RTX_NEED_POP; r2 = p2 ? r2 : RTX_NEED_POP; r3 = p3 ? r3 : RTX_NEED_POP;
    const int n = (int)p0 + (int)p1 + (int)p2 + (int)p3;
    // a passing entry distance is finite or -inf, so misses (+inf) sort last
    RTX_CSWAP(t0, r0, t1, r1) RTX_CSWAP(t2, r2, t3, r3) RTX_CSWAP(t0, r0, t2, r2) RTX_CSWAP(t1, r1, t3, r3) RTX_CSWAP(t1, r1, t2, r2)
    if (n > 1) {
        if (n > 3) st.push(r3, t3);
        if (n > 2) st.push(r2, t2);
        st.push(r1, t1);
    }
    return r0;
}

// One step through a 4-wide node: tests the four child boxes, drops those that miss or start beyond
// `limit`, orders the rest by entry distance, pushes all but the nearest (farthest first) and returns
// the nearest — or RTX_NEED_POP when none is left. Empty slots hold NaN boxes: their distances compare
// false and never pass. Nodes are addressed as base + 32-bit byte offset (rtx_create keeps to the
// reference topology for trees beyond 4 GiB of nodes).
//
// PACKED selects how the 24 plane distances are computed: as 12 + 12 packed FP32 instructions written
// out by hand, or as plain scalar code. Same values either way; measured at 4K / 10 M triangles the
// packed form is 4% faster in k_primary and 5% slower in the lane-refill kernel (where it costs
// occupancy), so each caller picks.
//
// BIG: trees beyond 2^25 nodes (4 GiB; ~100 M triangles) cannot use SGPR-base + 32-bit-offset addressing;
// the same loads then take 64-bit lane addresses (a few more instructions and registers per step).
template <bool PACKED, bool BIG>
__device__ __forceinline__ int wide_step(const float4* __restrict__ nodes, int cur, const WideRay& w, float limit, LaneStack& st) {
    rtx_f4 nx, fx, ny, fy, nz, fz; int4 rf;
    if (BIG) {
        const uintptr_t node = (uintptr_t)nodes + ((uintptr_t)(uint32_t)cur << 7);      // 128-byte aligned
        const uintptr_t ax = node | w.nx, ay = node | w.ny, az = node | w.nz;
        nx = *(const rtx_f4*)(ax); fx = *(const rtx_f4*)(ax ^ 16u);
        ny = *(const rtx_f4*)(ay + 32); fy = *(const rtx_f4*)((ay ^ 16u) + 32);
        nz = *(const rtx_f4*)(az + 64); fz = *(const rtx_f4*)((az ^ 16u) + 64);
        rf = *(const int4*)(node + 96);
    } else {
        const char* base = (const char*)nodes;
        const uint32_t off = (uint32_t)cur << 7;
        const uint32_t ax = off | w.nx, ay = off | w.ny, az = off | w.nz;
        nx = *(const rtx_f4*)(base + ax); fx = *(const rtx_f4*)(base + (ax ^ 16u));
        ny = *(const rtx_f4*)(base + ay + 32); fy = *(const rtx_f4*)(base + (ay ^ 16u) + 32);
        nz = *(const rtx_f4*)(base + az + 64); fz = *(const rtx_f4*)(base + (az ^ 16u) + 64);
        rf = *(const int4*)(base + off + 96);
    }
    float t0, t1, t2, t3, f0, f1, f2, f3;
    if (PACKED) {
    const rtx_f2 tnx01 = pk_plane_distance<0, 1>(nx.xy, w.a, w.b), tnx23 = pk_plane_distance<0, 1>(nx.zw, w.a, w.b);
    const rtx_f2 tny01 = pk_plane_distance<1, 0>(ny.xy, w.a, w.c), tny23 = pk_plane_distance<1, 0>(ny.zw, w.a, w.c);
    const rtx_f2 tnz01 = pk_plane_distance<0, 1>(nz.xy, w.b, w.c), tnz23 = pk_plane_distance<0, 1>(nz.zw, w.b, w.c);
    const rtx_f2 tfx01 = pk_plane_distance<0, 1>(fx.xy, w.a, w.b), tfx23 = pk_plane_distance<0, 1>(fx.zw, w.a, w.b);
    const rtx_f2 tfy01 = pk_plane_distance<1, 0>(fy.xy, w.a, w.c), tfy23 = pk_plane_distance<1, 0>(fy.zw, w.a, w.c);
    const rtx_f2 tfz01 = pk_plane_distance<0, 1>(fz.xy, w.b, w.c), tfz23 = pk_plane_distance<0, 1>(fz.zw, w.b, w.c);
        t0 = fmaxf(tnx01.x, fmaxf(tny01.x, tnz01.x)); t1 = fmaxf(tnx01.y, fmaxf(tny01.y, tnz01.y));
        t2 = fmaxf(tnx23.x, fmaxf(tny23.x, tnz23.x)); t3 = fmaxf(tnx23.y, fmaxf(tny23.y, tnz23.y));
        f0 = fminf(tfx01.x, fminf(tfy01.x, tfz01.x)); f1 = fminf(tfx01.y, fminf(tfy01.y, tfz01.y));
        f2 = fminf(tfx23.x, fminf(tfy23.x, tfz23.x)); f3 = fminf(tfx23.y, fminf(tfy23.y, tfz23.y));
    } else {
        const float ox = w.a.x, oy = w.a.y, oz = w.b.x, ix = w.b.y, iy = w.c.x, iz = w.c.y;
        t0 = fmaxf((nx.x - ox) * ix, fmaxf((ny.x - oy) * iy, (nz.x - oz) * iz));
        t1 = fmaxf((nx.y - ox) * ix, fmaxf((ny.y - oy) * iy, (nz.y - oz) * iz));
        t2 = fmaxf((nx.z - ox) * ix, fmaxf((ny.z - oy) * iy, (nz.z - oz) * iz));
        t3 = fmaxf((nx.w - ox) * ix, fmaxf((ny.w - oy) * iy, (nz.w - oz) * iz));
        f0 = fminf((fx.x - ox) * ix, fminf((fy.x - oy) * iy, (fz.x - oz) * iz));
        f1 = fminf((fx.y - ox) * ix, fminf((fy.y - oy) * iy, (fz.y - oz) * iz));
        f2 = fminf((fx.z - ox) * ix, fminf((fy.z - oy) * iy, (fz.z - oz) * iz));
        f3 = fminf((fx.w - ox) * ix, fminf((fy.w - oy) * iy, (fz.w - oz) * iz));
    }
    const bool p0 = t0 <= f0 && f0 > 0.0f && !(t0 > limit);
    const bool p1 = t1 <= f1 && f1 > 0.0f && !(t1 > limit);
    const bool p2 = t2 <= f2 && f2 > 0.0f && !(t2 > limit);
    const bool p3 = t3 <= f3 && f3 > 0.0f && !(t3 > limit);
    return order_and_push(t0, t1, t2, t3, p0, p1, p2, p3, rf, st);
}

// Fast path for clean lanes (RTX_TRAVERSE_CULLED): own 4-wide SAH tree over the reference's leaves,
// near-first order, distance culling at push and at pop. One flat loop: every pass does at most one
// pop, then one node step or one leaf test per lane — no nested per-lane loops for the wave to
// serialise (nested while-while forms measured 1.8x slower here).
template <bool COUNT, bool BIG>
__device__ __forceinline__ Hit traverse_fast(const DevScene& sc, float ox, float oy, float oz, float dx, float dy, float dz,
                                             LaneStack st, uint32_t& innerCount, uint32_t& leafCount) {
    Hit best; best.t = INFINITY; best.leaf = -1; best.box = -1;
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    const WideRay w = wide_ray(ox, oy, oz, ix, iy, iz);
    const float4* __restrict__ nodes = sc.fast;
    st.sp = 0; st.base = 0;
    int cur = sc.fastRoot;
    float limit = INFINITY;
    for (;;) {
        if (cur == RTX_NEED_POP) {
            if (st.sp == 0) break;
            st.sp--;
            int ref; float tn;
            if (st.sp >= st.base) {
                const int slot = st.sp & (RTX_LDS_WINDOW - 1);
                ref = rtx_lds[st.lds + slot * st.stride]; tn = __int_as_float(rtx_lds[st.lds + (RTX_LDS_WINDOW + slot) * st.stride]);
            } else {
                const int2 e = st.spill[(long long)st.sp * st.spillStride];
                ref = e.x; tn = __int_as_float(e.y); st.base = st.sp;
            }
            if (tn > limit) continue;
            cur = ref;
        }
        if (cur >= 0) {
            if (COUNT) innerCount++;
            cur = wide_step<true, BIG>(nodes, cur, w, limit, st);
        }
        // not "else": a lane whose step ends at a leaf tests it in this same pass (the leaf section
        // runs in practically every pass anyway: some lane of the 64 is at a leaf)
        if (cur < 0 && cur != RTX_NEED_POP) {
            int leafIdx = ~cur, box;
            float t = intersect_leaf(sc.leaf, leafIdx, ox, oy, oz, dx, dy, dz, box);
            if (COUNT) leafCount++;
            if (t > 0.01f && (t < best.t || (t == best.t && best.leaf >= 0 && box > best.box))) {
                best.t = t; best.leaf = leafIdx; best.box = box;
                limit = t + t * sc.cullSlack;
            }
            cur = RTX_NEED_POP;
        }
    }
    return best;
}

// ---- RTX_TRAVERSE_WIDE: quantised nodes, no order, no culling ---------------------------------------------------------
// Without distance culling the set of leaves a ray tests does not depend on the order of the walk (a
// leaf is tested iff its own box passes), so this path drops everything that only served the order: no sorting
// network, no entry distances on the stack. And since the leaf's own exact box has the last word, the inner tests only
// have to be conservative: 48-byte nodes with byte boxes (rtx_build.h NodeQ), three lane loads per step instead of seven.
//
// A lane keeps two work lists, inner nodes and leaves, each as a current entry plus a small LDS stack; an entry is
// (base << 4 | 4-bit slot mask): "children base + j of one node, for every set bit j" — a node's passing inner children
// are ONE entry, its passing leaf children another. Every pass a lane takes one node step and one leaf test, whichever
// of the two it has, so the two sections of a pass run on well-filled waves and a ray needs about max(node steps,
// leaf tests) passes instead of their interleaved sum.
#ifndef RTX_QFAST
#define RTX_QFAST 1        // 1: inner tests in the folded form with a proven error margin (q_step); 0: the reference's own form on the decoded planes
#endif
// The folded inner test (RTX_QFAST). An inner test only has to be CONSERVATIVE (pass whenever the exact test of a leaf below
// passes; the leaf's own exact box, tested with the reference's arithmetic in q_leaf, has the last word — DESIGN.md 3.1).
// The reference form at a decoded plane p_f = fl(org + b*s) is G = fl(fl(p_f - o) * inv): cvt, fma, sub, mul per plane.
// Folded: per node and axis S = s * inv (exact: s is a power of two), A = fl(fl(org - o) * inv), and per plane
// H = fl(b * S + A): cvt + fma. With T = (org + b*s - o) * inv in real arithmetic and u = 2^-24 (no overflow, see below):
//     |G - T| <= 1.001 u |inv| (|org| + 255 s) + 2.001 u |T|          (rounding of p_f, of the difference, of the product)
//     |H - T| <= 2.002 u |A| + u |T|                                   (two roundings of A, one of the fma)
//     |T| <= |A| + 255 |S|,   |org * inv| <= |A| (1 + 2.01 u) + |o * inv|
// hence |H - G| <= u (6.01 |A| + 1022 |S| + 1.01 |o * inv|) =: d. The test uses A - 2d for the entry planes and A + 2d for
// the exit planes (the factor 2 covers the roundings of d itself and of A -+ 2d, plus 2^-120 for results that round in the
// subnormal range), so Hn <= G(near plane) and Hf >= G(far plane) per axis: max Hn <= max Gn <= min Gf <= min Hf and
// min Hf >= min Gf > 0 whenever the reference-form test of the decoded box passes — which it does whenever a leaf below
// passes (monotonic rounding). False positives cost a node step, never a hit.
// No overflow: lanes take this walk only when |inv| <= 2^40 and |o| <= 2^40 (q_lane_ok; others go to the reference-topology
// side kernel), and rtx_create quantises only scenes whose boxes lie within +-2^40 (|A| <= 2^82, |S| <= 2^76).
#define RTX_Q_RANGE 1.099511627776e12f      // 2^40
struct QRay { float ox, oy, oz, ix, iy, iz; bool nx, ny, nz;      // n*: the reciprocal is negative, the near plane is hi
#if RTX_QFAST
              float ex, ey, ez;                                   // 2^-23 * 1.01 |o * inv| + 2^-120: the ray's share of the margin 2d
#endif
};
__device__ __forceinline__ QRay q_ray(float ox, float oy, float oz, float ix, float iy, float iz) {
    QRay r; r.ox = ox; r.oy = oy; r.oz = oz; r.ix = ix; r.iy = iy; r.iz = iz;
    r.nx = ix < 0.0f; r.ny = iy < 0.0f; r.nz = iz < 0.0f;
#if RTX_QFAST
    const float k = 1.2040138e-7f /* 1.01 * 2^-23 */, tiny = 7.5231638e-37f /* 2^-120 */;
    r.ex = fmaf(k, fabsf(ox * ix), tiny); r.ey = fmaf(k, fabsf(oy * iy), tiny); r.ez = fmaf(k, fabsf(oz * iz), tiny);
#endif
    return r;
}
// A lane walks the quantised tree only when its slab arithmetic cannot produce NaN (lane_is_clean) and, with the folded
// inner test, cannot overflow either.
__device__ __forceinline__ bool q_lane_ok(float ox, float oy, float oz, float dx, float dy, float dz) {
    bool ok = lane_is_clean(ox, oy, oz, dx, dy, dz);
#if RTX_QFAST
    ok = ok && fabsf(1.0f / dx) <= RTX_Q_RANGE && fabsf(1.0f / dy) <= RTX_Q_RANGE && fabsf(1.0f / dz) <= RTX_Q_RANGE &&
         fabsf(ox) <= RTX_Q_RANGE && fabsf(oy) <= RTX_Q_RANGE && fabsf(oz) <= RTX_Q_RANGE;
#endif
    return ok;
}

#ifndef RTX_QN             // (a power of two; tests build a variant with 2 so that small trees spill, tests/test_gpu_spill.py)
#define RTX_QN 16          // node entries per lane kept in LDS (older ones spill to HBM; the stack is at most tree depth + 1 deep)
#endif
#define RTX_QL 4           // leaf entries per lane; a lane whose leaf stack is full takes no node step until it has room
#define RTX_QWORDS (RTX_QN + RTX_QL)

struct QStack {
    int lds, stride;           // this lane's LDS column: node entry k at rtx_lds[lds + (k % RTX_QN) * stride], leaf entry k at rtx_lds[lds + (RTX_QN + k) * stride]
    int* spill;                // this thread's HBM column: node entry k at spill[k * spillStride]
    long long spillStride;
    int nsp, nbase, lsp;       // node entries [nbase, nsp) are in LDS, [0, nbase) in HBM; lsp leaf entries
    __device__ __forceinline__ void push_node(int e) {
        const int slot = nsp & (RTX_QN - 1);
        if (nsp - nbase == RTX_QN) { spill[(long long)nbase * spillStride] = rtx_lds[lds + slot * stride]; nbase++; }
        rtx_lds[lds + slot * stride] = e;
        nsp++;
    }
    __device__ __forceinline__ int pop_node() {
        nsp--;
        // the LDS slot is read whether or not the entry still lives there: written as "LDS or else HBM" the two loads become
        // one flat_load through a selected pointer, and every pop waits on the flat path
        int e = rtx_lds[lds + (nsp & (RTX_QN - 1)) * stride];
        asm volatile("" : "+v"(e));      // (keeps the two loads apart)
        if (nsp < nbase) { nbase = nsp; e = __builtin_nontemporal_load(spill + (long long)nsp * spillStride); }
        return e;
    }
    __device__ __forceinline__ void push_leaf(int e) { rtx_lds[lds + (RTX_QN + lsp) * stride] = e; lsp++; }
    __device__ __forceinline__ int pop_leaf() { lsp--; return rtx_lds[lds + (RTX_QN + lsp) * stride]; }
};

#define RTX_QENTRY(base, mask) ((int)(((uint32_t)(base) << 4) | (mask)))

#ifndef RTX_QNODE_WORDS
#define RTX_QNODE_WORDS 16      // rtx_build.h NodeQ
#endif
// The three lane loads of a node step. BIG: node arrays beyond 4 GiB (2^26 nodes) take 64-bit lane addresses.
struct QNodeData { uint4 a, b, c; };
template <bool BIG>
__device__ __forceinline__ QNodeData q_load_node(const uint4* __restrict__ nodes, int idx) {
    QNodeData n;
    if (BIG) {
        const uint4* N = (const uint4*)((const char*)nodes + (size_t)(uint32_t)idx * (RTX_QNODE_WORDS * 4));
        n.a = N[0]; n.b = N[1]; n.c = N[2];
    } else {
        const char* base = (const char*)nodes;
        const uint32_t off = (uint32_t)idx * (uint32_t)(RTX_QNODE_WORDS * 4);
        n.a = *(const uint4*)(base + off); n.b = *(const uint4*)(base + off + 16); n.c = *(const uint4*)(base + off + 32);
    }
    return n;
}

// One step through a quantised node: returns the slots whose decoded box passes the reference's slab test.
__device__ __forceinline__ uint32_t q_step(const QNodeData& n, const QRay& r, int& baseInner, int& baseLeaf, uint32_t& leafBits) {
    const uint4 a = n.a, b = n.b, c = n.c;
    const float orx = __uint_as_float(a.x), ory = __uint_as_float(a.y), orz = __uint_as_float(a.z);
    const uint32_t meta = a.w;
    const float sx = __uint_as_float((meta & 0xffu) << 23), sy = __uint_as_float((meta & 0xff00u) << 15), sz = __uint_as_float((meta & 0xff0000u) << 7);
    const uint32_t nxw = r.nx ? b.y : b.x, fxw = r.nx ? b.x : b.y;
    const uint32_t nyw = r.ny ? b.w : b.z, fyw = r.ny ? b.z : b.w;
    const uint32_t nzw = r.nz ? c.y : c.x, fzw = r.nz ? c.x : c.y;
    baseInner = (int)c.z; baseLeaf = (int)c.w; leafBits = meta >> 28;
    uint32_t mask = 0;
#if RTX_QFAST
    // folded form with margin (see QRay): per axis S, A -+ 2d; per plane one conversion and one fma
    const float Sx = sx * r.ix, Sy = sy * r.iy, Sz = sz * r.iz;
    const float Ax = (orx - r.ox) * r.ix, Ay = (ory - r.oy) * r.iy, Az = (orz - r.oz) * r.iz;
    const float kA = 7.1644784e-7f /* 6.01 * 2^-23 */, kS = 1.2183190e-4f /* 1022 * 2^-23 */;
    const float mx = fmaf(kA, fabsf(Ax), fmaf(kS, fabsf(Sx), r.ex)), my = fmaf(kA, fabsf(Ay), fmaf(kS, fabsf(Sy), r.ey)), mz = fmaf(kA, fabsf(Az), fmaf(kS, fabsf(Sz), r.ez));
    const float Anx = Ax - mx, Any = Ay - my, Anz = Az - mz, Afx = Ax + mx, Afy = Ay + my, Afz = Az + mz;
#define RTX_QBYTE(word, k) ((float)(((word) >> (8 * (k))) & 0xffu))
#define RTX_QCHILD(k) {                                                                                                      \
        const float tn = fmaxf(fmaf(RTX_QBYTE(nxw, k), Sx, Anx), fmaxf(fmaf(RTX_QBYTE(nyw, k), Sy, Any), fmaf(RTX_QBYTE(nzw, k), Sz, Anz))); \
        const float tf = fminf(fmaf(RTX_QBYTE(fxw, k), Sx, Afx), fminf(fmaf(RTX_QBYTE(fyw, k), Sy, Afy), fmaf(RTX_QBYTE(fzw, k), Sz, Afz))); \
        mask |= (tn <= tf && tf > 0.0f) ? (1u << (k)) : 0u; }
    RTX_QCHILD(0) RTX_QCHILD(1) RTX_QCHILD(2) RTX_QCHILD(3)
#undef RTX_QCHILD
#undef RTX_QBYTE
    (void)sx; (void)sy; (void)sz;
    return mask & ((meta >> 24) & 15u);
#endif
    // decoded plane = fma(byte, step, origin) (the value rtx_build.h verified against the child's box), then the
    // reference's own (p - o) * inv: separate subtraction and multiplication. (Written for pairs of children as packed
    // FP32 — v_pk_fma / v_pk_add / v_pk_mul — the step needs 108 registers instead of 77, occupancy falls from 6 to 4
    // waves per SIMD and the bounce-1 launch goes from 3.15 to 3.36 ms; capped at 80 registers it spills: 5.2 ms.)
#define RTX_QPLANE(word, k, st, org, o, i) ((fmaf((float)(((word) >> (8 * (k))) & 0xffu), st, org) - (o)) * (i))
#define RTX_QCHILD(k) {                                                                                                      \
        const float tn = fmaxf(RTX_QPLANE(nxw, k, sx, orx, r.ox, r.ix), fmaxf(RTX_QPLANE(nyw, k, sy, ory, r.oy, r.iy), RTX_QPLANE(nzw, k, sz, orz, r.oz, r.iz))); \
        const float tf = fminf(RTX_QPLANE(fxw, k, sx, orx, r.ox, r.ix), fminf(RTX_QPLANE(fyw, k, sy, ory, r.oy, r.iy), RTX_QPLANE(fzw, k, sz, orz, r.oz, r.iz))); \
        mask |= (tn <= tf && tf > 0.0f) ? (1u << (k)) : 0u; }
    RTX_QCHILD(0) RTX_QCHILD(1) RTX_QCHILD(2) RTX_QCHILD(3)
#undef RTX_QCHILD
#undef RTX_QPLANE
    return mask & ((meta >> 24) & 15u);
}

// Leaf record (64 bytes, four lane loads): q0 = (A.xyz e1.x) q1 = (e1.yz e2.xy) q2 = (e2.z tag lo.x lo.y) q3 = (lo.z hi.xyz)
// — the primitive and its own EXACT box. The box is tested with the reference's arithmetic (planes picked by the sign of
// the reciprocal, equal to the reference's min/max for a clean lane and min <= max); only if it passes is the primitive
// intersected, which is exactly when the reference intersects it.
struct QLeafData { float4 q0, q1, q2, q3; };
__device__ __forceinline__ QLeafData q_load_leaf(const float4* __restrict__ leafBase, int leafIdx) {
    const float4* L = leafBase + (size_t)leafIdx * 4;
    QLeafData d; d.q0 = L[0]; d.q1 = L[1]; d.q2 = L[2]; d.q3 = L[3];
    return d;
}
template <bool COUNT>
__device__ __forceinline__ void q_leaf(const QLeafData& d, int leafIdx, const QRay& r, float dx, float dy, float dz,
                                       Hit& best, uint32_t& leafCount) {
    const float4 q0 = d.q0, q1 = d.q1, q2 = d.q2, q3 = d.q3;
    const float lx = q2.z, ly = q2.w, lz = q3.x, hx = q3.y, hy = q3.z, hz = q3.w;
    const float tn = fmaxf(((r.nx ? hx : lx) - r.ox) * r.ix, fmaxf(((r.ny ? hy : ly) - r.oy) * r.iy, ((r.nz ? hz : lz) - r.oz) * r.iz));
    const float tf = fminf(((r.nx ? lx : hx) - r.ox) * r.ix, fminf(((r.ny ? ly : hy) - r.oy) * r.iy, ((r.nz ? lz : hz) - r.oz) * r.iz));
    if (tn <= tf && tf > 0.0f) {
        const int tag = __float_as_int(q2.y);
        const int box = tag >> 3;
        float t;
        if ((tag & 7) == 4) t = intersect_triangle(r.ox, r.oy, r.oz, dx, dy, dz, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x);
        else t = intersect_sphere(r.ox, r.oy, r.oz, dx, dy, dz, q0.x, q0.y, q0.z, q0.w);
        if (COUNT) leafCount++;
        if (t > 0.01f && (t < best.t || (t == best.t && best.leaf >= 0 && box > best.box))) { best.t = t; best.leaf = leafIdx; best.box = box; }
    }
}

// One pass of a lane: a node step if it has a node (and room for the leaves the step may find) and a leaf test if it has
// a leaf — the loads of both are issued together at the top of the pass, so a pass costs ONE memory round trip — then both
// lists are topped up from their stacks. Returns false when the ray has nothing left to do.
// (Running the leaf section only when enough lanes hold a leaf — leaves can wait, nothing depends on their order — was
// tried with thresholds of 20 / 28 / 36 lanes and a starvation guard: every variant slower, 3.24 -> 3.5-3.9 ms on the
// bounce-1 launch, because a lane that has run out of nodes idles until the section runs.)
template <bool COUNT, bool BIG>
__device__ __forceinline__ bool q_pass(const uint4* __restrict__ nodes, const float4* __restrict__ leaves, const QRay& r, float dx, float dy, float dz,
                                       int& nodeCur, int& leafCur, QStack& st, Hit& best, uint32_t& innerCount, uint32_t& leafCount) {
    const bool doNode = (nodeCur & 15) && st.lsp < RTX_QL;
    const bool doLeaf = (leafCur & 15) != 0;
    // Both records are loaded by every lane in the pass, a lane without a node (or leaf) reading record 0: the loads then
    // sit in straight-line code ahead of everything that waits for them (inside divergent blocks the compiler parks
    // copies behind each block, and the pass waits for memory twice); the idle lanes' reads of one address merge into
    // one L1 access.
    const int nodeIdx = doNode ? (nodeCur >> 4) + (__ffs(nodeCur & 15) - 1) : 0;
    const QNodeData nd = q_load_node<BIG>(nodes, nodeIdx);
    if (doNode) nodeCur &= nodeCur - 1;                 // clears the lowest set bit, which lies inside the slot mask
    const int leafIdx = doLeaf ? (leafCur >> 4) + (__ffs(leafCur & 15) - 1) : 0;
    const QLeafData ld = q_load_leaf(leaves, leafIdx);
    if (doLeaf) leafCur &= leafCur - 1;
    if (doNode) {
        int bi, bl; uint32_t leafBits;
        const uint32_t m = q_step(nd, r, bi, bl, leafBits);
        if (COUNT) innerCount++;
        const uint32_t mi = m & ~leafBits, ml = m & leafBits;
        if (mi) { if (nodeCur & 15) st.push_node(nodeCur); nodeCur = RTX_QENTRY(bi, mi); }
        if (ml) { if (leafCur & 15) st.push_leaf(leafCur); leafCur = RTX_QENTRY(bl, ml); }
    }
    if (doLeaf) q_leaf<COUNT>(ld, leafIdx, r, dx, dy, dz, best, leafCount);
    if (!(nodeCur & 15) && st.nsp > 0) nodeCur = st.pop_node();
    if (!(leafCur & 15) && st.lsp > 0) leafCur = st.pop_leaf();
    return ((nodeCur | leafCur) & 15) != 0;
}

// closest hit of one ray per lane (primary rays: a wave's 64 rays are neighbours and run about equally long)
template <bool COUNT, bool BIG>
__device__ __forceinline__ Hit traverse_q(const DevScene& sc, float ox, float oy, float oz, float dx, float dy, float dz,
                                          QStack st, uint32_t& innerCount, uint32_t& leafCount) {
    Hit best; best.t = INFINITY; best.leaf = -1; best.box = -1;
    const QRay r = q_ray(ox, oy, oz, 1.0f / dx, 1.0f / dy, 1.0f / dz);
    st.nsp = 0; st.nbase = 0; st.lsp = 0;
    int nodeCur = RTX_QENTRY(0, 1), leafCur = 0;         // the root is node 0
    while (q_pass<COUNT, BIG>(sc.qnodes, sc.leaf, r, dx, dy, dz, nodeCur, leafCur, st, best, innerCount, leafCount)) {}
    return best;
}

// ---- leaf tests on compacted lanes (k_trace_qw, k_primary) ----------------------------------------------------------------
// In q_pass the leaf section runs in practically every pass of a wave (some lane holds a leaf) with about a fifth of its
// lanes active. Here the lanes of a wave hand the leaves their node steps find to a queue of the WAVE — one entry
// (leaf base << 4 | slot mask, owner lane) per node step that found any — and whenever 64 entries have gathered, the wave
// expands them into single leaves and tests those 64 at a time, each lane testing a leaf of whichever ray it belongs to: the
// ray comes from its owner's registers (ds_bpermute), the result goes into the owner's (t, ~box) key in LDS with a 64-bit
// minimum — the reference's rule "smaller t, among equal t the larger box" (see q_leaf) as one unsigned comparison, so the
// order in which a ray's leaves are tested does not matter, as before. A leaf is still tested iff its own exact box passes
// the reference's arithmetic, with the same operations on the same operands: which lane evaluates them changes nothing.
// Per-wave LDS (ints, from `key`): 64 keys (2 words each) | 64 best leaves | RTX_WQ_ENTRIES queue entries | 256 expanded
// leaves | owner lanes of the entries and of the expanded leaves (bytes).
#define RTX_WQ_ENTRIES 128u      // a pass adds at most 64 and the queue is served as soon as it holds 64
#define RTX_WQ_WORDS (128 + 64 + (int)RTX_WQ_ENTRIES + 256 + (int)RTX_WQ_ENTRIES / 4 + 256 / 4)
#define RTX_WQ_NOHIT 0x7f800000ffffffffull      // (+inf, no box)
struct QWave {
    int key, leaf, q, s, qOwner, sOwner;      // int indices into rtx_lds (qOwner, sOwner: byte indices)
    uint32_t head, tail;                      // wave-uniform: entries [head, tail) wait in the queue
    __device__ __forceinline__ void init(int firstWord) {
        key = firstWord; leaf = key + 128; q = leaf + 64; s = q + (int)RTX_WQ_ENTRIES; qOwner = (s + 256) * 4; sOwner = qOwner + (int)RTX_WQ_ENTRIES;
        head = 0; tail = 0;
    }
};
__device__ __forceinline__ uint32_t q_mbcnt(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ float q_from_lane(float v, int byteAddr) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byteAddr, __float_as_int(v))); }

// a lane's node step found leaves (mask != 0): one entry each; wave-uniform control flow
__device__ __forceinline__ void q_wave_push(QWave& w, uint32_t lane, int baseLeaf, uint32_t mask) {
    const unsigned long long has = __ballot(mask != 0u);
    if (has) {
        const uint32_t slot = (w.tail + q_mbcnt(has)) & (RTX_WQ_ENTRIES - 1u);
        if (mask) {
            rtx_lds[w.q + (int)slot] = RTX_QENTRY(baseLeaf, mask);
            ((unsigned char*)rtx_lds)[w.qOwner + (int)slot] = (unsigned char)lane;
        }
        w.tail += (uint32_t)__popcll(has);
    }
}

// The first n (<= 64, wave-uniform) entries of the queue: expanded, tested, merged. ox .. dz: THIS lane's ray (read by the
// lanes that test its leaves).
template <bool COUNT>
__device__ __forceinline__ void q_wave_serve(const float4* __restrict__ leaves, QWave& w, uint32_t n, uint32_t lane,
                                             float ox, float oy, float oz, float ix, float iy, float iz, float dx, float dy, float dz,
                                             uint32_t& leafCount) {
    unsigned char* const lds8 = (unsigned char*)rtx_lds;
    __builtin_amdgcn_wave_barrier();
    const bool valid = lane < n;
    const uint32_t slot = (w.head + lane) & (RTX_WQ_ENTRIES - 1u);
    const int e = valid ? rtx_lds[w.q + (int)slot] : 0;
    const unsigned char own = lds8[w.qOwner + (int)slot];
    const uint32_t m = (uint32_t)e & 15u;
    const int base = e >> 4;
    const uint32_t c = (uint32_t)__popc(m);
    const unsigned long long b0 = __ballot((c & 1u) != 0u), b1 = __ballot((c & 2u) != 0u), b2 = __ballot((c & 4u) != 0u);
    uint32_t p = q_mbcnt(b0) + 2u * q_mbcnt(b1) + 4u * q_mbcnt(b2);
    const uint32_t total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (m & (1u << k)) { rtx_lds[w.s + (int)p] = base + k; lds8[w.sOwner + (int)p] = own; ++p; }
    w.head += n;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t c0 = 0; c0 < total; c0 += 64u) {
        const bool act = c0 + lane < total;
        const uint32_t j = act ? c0 + lane : 0u;
        const int leafIdx = rtx_lds[w.s + (int)j];
        const QLeafData d = q_load_leaf(leaves, leafIdx);
        const int owner = (int)lds8[w.sOwner + (int)j];
        const int oa = owner << 2;
        const float rox = q_from_lane(ox, oa), roy = q_from_lane(oy, oa), roz = q_from_lane(oz, oa);
        const float rix = q_from_lane(ix, oa), riy = q_from_lane(iy, oa), riz = q_from_lane(iz, oa);
        const float rdx = q_from_lane(dx, oa), rdy = q_from_lane(dy, oa), rdz = q_from_lane(dz, oa);
        const bool nx = rix < 0.0f, ny = riy < 0.0f, nz = riz < 0.0f;
        float4 q0 = d.q0, q1 = d.q1, q2 = d.q2, q3 = d.q3;
        // the whole record is taken here, in one round trip: left to itself the compiler moves the loads of the primitive's part
        // into the branch below, behind the wait for the box
        asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w), "+v"(q2.x), "+v"(q2.y));
        const float lx = q2.z, ly = q2.w, lz = q3.x, hx = q3.y, hy = q3.z, hz = q3.w;
        const float tn = fmaxf(((nx ? hx : lx) - rox) * rix, fmaxf(((ny ? hy : ly) - roy) * riy, ((nz ? hz : lz) - roz) * riz));
        const float tf = fminf(((nx ? lx : hx) - rox) * rix, fminf(((ny ? ly : hy) - roy) * riy, ((nz ? lz : hz) - roz) * riz));
        if (act && tn <= tf && tf > 0.0f) {
            const int tag = __float_as_int(q2.y);
            const int box = tag >> 3;
            float t;
            if ((tag & 7) == 4) t = intersect_triangle(rox, roy, roz, rdx, rdy, rdz, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x);
            else t = intersect_sphere(rox, roy, roz, rdx, rdy, rdz, q0.x, q0.y, q0.z, q0.w);
            if (COUNT) leafCount++;
            if (t > 0.01f && t < INFINITY) {
                // (+inf never wins in q_leaf either: not below an initial +inf, and no finite best equals it)
                const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (uint32_t)~box;
                unsigned long long* const slotKey = (unsigned long long*)&rtx_lds[w.key + 2 * owner];
                __hip_atomic_fetch_min(slotKey, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                // several lanes may have lowered one ray's key in that instruction: the one whose key stands names the leaf
                if (__hip_atomic_load(slotKey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == key) rtx_lds[w.leaf + owner] = leafIdx;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Closest hits of a WAVE's rays, one per lane (k_primary), with the leaf tests on compacted lanes. Every lane of the wave
// calls this together; `active`: the lane has a ray to trace. Returns the lane's hit (box is not filled in).
template <bool COUNT, bool BIG>
__device__ __forceinline__ Hit traverse_q_wave(const DevScene& sc, bool active, float ox, float oy, float oz, float dx, float dy, float dz,
                                               QStack st, QWave& wq, uint32_t lane, uint32_t& innerCount, uint32_t& leafCount) {
    const QRay r = q_ray(ox, oy, oz, 1.0f / dx, 1.0f / dy, 1.0f / dz);
    unsigned long long* const myKey = (unsigned long long*)&rtx_lds[wq.key + 2 * (int)lane];
    __hip_atomic_store(myKey, RTX_WQ_NOHIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    rtx_lds[wq.leaf + (int)lane] = -1;
    st.nsp = 0; st.nbase = 0; st.lsp = 0;
    int nodeCur = active ? RTX_QENTRY(0, 1) : 0;         // the root is node 0
    bool busy = active;
    while (__any(busy)) {
        const int nodeIdx = busy ? (nodeCur >> 4) + (__ffs(nodeCur & 15) - 1) : 0;
        const QNodeData nd = q_load_node<BIG>(sc.qnodes, nodeIdx);
        uint32_t ml = 0; int bl = 0;
        if (busy) {
            nodeCur &= nodeCur - 1;
            int bi; uint32_t leafBits;
            const uint32_t m = q_step(nd, r, bi, bl, leafBits);
            if (COUNT) innerCount++;
            const uint32_t mi = m & ~leafBits;
            ml = m & leafBits;
            if (mi) { if (nodeCur & 15) st.push_node(nodeCur); nodeCur = RTX_QENTRY(bi, mi); }
            if (!(nodeCur & 15) && st.nsp > 0) nodeCur = st.pop_node();
            busy = (nodeCur & 15) != 0;
        }
        q_wave_push(wq, lane, bl, ml);
        if (wq.tail - wq.head >= 64u)
            q_wave_serve<COUNT>(sc.leaf, wq, 64u, lane, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, dx, dy, dz, leafCount);
    }
    if (wq.tail != wq.head)
        q_wave_serve<COUNT>(sc.leaf, wq, wq.tail - wq.head, lane, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, dx, dy, dz, leafCount);
    Hit best;
    best.t = __uint_as_float((uint32_t)(__hip_atomic_load(myKey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >> 32));
    best.leaf = rtx_lds[wq.leaf + (int)lane];
    best.box = -1;
    return best;
}

// Linear list (src/ThreadedShader.cpp:219-237): leaves are stored in object order.
template <bool COUNT>
__device__ __forceinline__ Hit traverse_linear(const DevScene& sc, float ox, float oy, float oz, float dx, float dy, float dz,
                                               uint32_t& leafCount) {
    Hit best; best.t = INFINITY; best.leaf = -1; best.box = -1;
    for (int i = 0; i < sc.numLeaves; ++i) {
        int box;
        float t = intersect_leaf(sc.leaf, i, ox, oy, oz, dx, dy, dz, box);
        if (COUNT) leafCount++;
        if ((t < best.t) && (t > 0.01f)) { best.t = t; best.leaf = i; best.box = box; }
    }
    return best;
}
