// bvh_build.cpp — builds the same BVH as the reference's BVHTree::BuildBVH
// (src/BVHTree.cpp:42-60,148-195) without its per-level vector copies:
//   * split axis = depth % 3, split point = n >> 1, chosen by std::nth_element
//     over Object.position[axis] (same comparator, same element sequence, same
//     libstdc++ introselect => same permutation as the reference's copies);
//   * one object per leaf; 2N-1 nodes in pre-order with the left subtree first.
// Because a subtree over n objects always has 2n-1 nodes, every node's index is
// known before its subtree is built: the array is allocated once, the id array
// is partitioned in place, and sibling subtrees are built on separate threads.
#include "host_scene.h"

#include <algorithm>
#include <thread>

namespace rtxh {

namespace {

// src/Vector3.h:180-182 — operator[] is x*(id==0) + y*(id==1) + z*(id==2)
inline float axis_key(const RtxVec4& p, int axis) {
    return p.x * (float)(axis == 0) + p.y * (float)(axis == 1) + p.z * (float)(axis == 2);
}

// src/BVHTree.cpp:8-40
inline void leaf_box(const RtxObject& o, int32_t objectID, int32_t parent, RtxBoundingBox& box) {
    memset(&box, 0, sizeof(box));
    box.objectID = objectID;
    box.parentID = parent;           // left uninitialised by the reference; never read by any backend
    box.leftID = -1; box.rightID = -1;
    if (o.type == RTX_SPHERE) {
        RtxVec4 r = v4(o.radius, o.radius, o.radius);
        box.minimalPosition = o.position - r;
        box.maximalPosition = o.position + r;
    } else {
        box.minimalPosition = vmin(vmin(o.vertices[0], o.vertices[1]), o.vertices[2]);
        box.maximalPosition = vmax(vmax(o.vertices[0], o.vertices[1]), o.vertices[2]);
    }
}

struct Builder {
    const std::vector<RtxObject>& objects;
    std::vector<RtxBoundingBox>& boxes;
    int32_t* ids;
    int parallelDepth;
    const RtxVec4* keys;   // objects[i].position, packed (same values, 16-byte stride instead of 256)

    void build(int64_t begin, int64_t end, int32_t node, int32_t parent, uint32_t depth) {
        const int64_t n = end - begin;
        if (n == 1) { leaf_box(objects[ids[begin]], ids[begin], parent, boxes[node]); return; }
        const int axis = (int)(depth % 3);
        const int64_t half = n >> 1;
        const RtxVec4* k = keys;
        std::nth_element(ids + begin, ids + begin + half, ids + end, [axis, k](const int32_t& a, const int32_t& b) {
            return axis_key(k[a], axis) < axis_key(k[b], axis);
        });
        const int32_t left = node + 1;
        const int32_t right = (int32_t)(node + 2 * half);
        if ((int)depth < parallelDepth && n > 4096) {
            std::thread other([&]() { build(begin, begin + half, left, node, depth + 1); });
            build(begin + half, end, right, node, depth + 1);
            other.join();
        } else {
            build(begin, begin + half, left, node, depth + 1);
            build(begin + half, end, right, node, depth + 1);
        }
        RtxBoundingBox& b = boxes[node];
        memset(&b, 0, sizeof(b));
        b.objectID = -1;
        b.parentID = parent;
        b.leftID = left; b.rightID = right;
        // src/BVHTree.cpp:185-189 (child ids are always > 0 here)
        b.minimalPosition = vmin(boxes[left].minimalPosition, boxes[right].minimalPosition);
        b.maximalPosition = vmax(boxes[left].maximalPosition, boxes[right].maximalPosition);
    }
};

} // namespace

void build_bvh(HostScene& s, int nthreads) {
    s.boxes.clear();
    s.bvhBuilt = true;
    const int64_t n = (int64_t)s.objects.size();
    if (n == 0) return;
    s.boxes.resize((size_t)(2 * n - 1));
    std::vector<int32_t> ids((size_t)n);
    for (int64_t i = 0; i < n; ++i) ids[(size_t)i] = (int32_t)i;
    std::vector<RtxVec4> keys((size_t)n);
    for (int64_t i = 0; i < n; ++i) keys[(size_t)i] = s.objects[(size_t)i].position;
    int par = 0;
    while ((1 << par) < nthreads) ++par;
    Builder b{s.objects, s.boxes, ids.data(), par, keys.data()};
    b.build(0, n, 0, -1, 0);
}

} // namespace rtxh
