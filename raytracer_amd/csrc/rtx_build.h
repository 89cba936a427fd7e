// rtx_build.h — host-side construction of the traversal tree the fast path uses.
//
// The reference's closest-hit result does not depend on the tree's topology
// (DESIGN.md "Topology freedom"): with boxes that contain their descendants, a
// leaf is tested iff its OWN box passes the reference slab test, the winner is
// the smallest t, ties go to the largest reference boxID. So rtx_create may
// index the same leaves (same leaf boxes, same primitives) with a better tree
// than the host's object-median one — whose few huge primitives (floor, light)
// sit at the bottom of the left spine and bloat every box above them.
//
// Binned SAH, top-down, one primitive per leaf, boxes = exact min/max unions of
// the given leaf boxes. A subtree over n leaves has n-1 inner nodes, so node and
// leaf positions are known before a subtree is built: arrays are allocated
// once and sibling subtrees are built on separate threads.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace rtxbuild {

struct Box { float lo[3], hi[3]; };

struct Node {           // one 64-byte inner record (layout of rtx_device.h)
    float q[16];
};

inline void grow(Box& b, const Box& o) {
    for (int a = 0; a < 3; ++a) { b.lo[a] = o.lo[a] < b.lo[a] ? o.lo[a] : b.lo[a]; b.hi[a] = o.hi[a] > b.hi[a] ? o.hi[a] : b.hi[a]; }
}
inline Box empty_box() { Box b; for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; } return b; }
inline double half_area(const Box& b) {
    double d0 = (double)b.hi[0] - b.lo[0], d1 = (double)b.hi[1] - b.lo[1], d2 = (double)b.hi[2] - b.lo[2];
    if (!(d0 >= 0) || !(d1 >= 0) || !(d2 >= 0)) return 0.0;
    return d0 * d1 + d1 * d2 + d2 * d0;
}
inline int32_t f2i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
inline float i2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

struct SahBuilder {
    const Box* prim;        // leaf boxes, indexed by primitive (= input leaf index)
    uint32_t* order;        // permutation being partitioned in place; final = leaf order of the new tree
    Node* nodes;            // n-1 inner records
    int parallelDepth;
    int maxDepth = 0;

    static const int kBins = 16;

    // builds the subtree over order[begin,end); returns its ref (>=0 inner index, <0 ~leafPosition)
    int32_t build(uint32_t begin, uint32_t end, int32_t nodeIdx, int depth, Box& outBox, int& outDepth) {
        const uint32_t n = end - begin;
        if (n == 1) { outBox = prim[order[begin]]; outDepth = depth; return ~(int32_t)begin; }
        // centroid bounds
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = begin; i < end; ++i) {
            const Box& b = prim[order[i]];
            for (int a = 0; a < 3; ++a) { float c = 0.5f * b.lo[a] + 0.5f * b.hi[a]; clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c); }
        }
        int bestAxis = -1, bestBin = -1; double bestCost = INFINITY;
        for (int a = 0; a < 3; ++a) {
            const float ext = chi[a] - clo[a];
            if (!(ext > 0.0f) || !std::isfinite(ext)) continue;
            Box bb[kBins]; uint32_t cnt[kBins];
            for (int k = 0; k < kBins; ++k) { bb[k] = empty_box(); cnt[k] = 0; }
            const float scale = (float)kBins / ext;
            for (uint32_t i = begin; i < end; ++i) {
                const Box& b = prim[order[i]];
                int k = (int)((0.5f * b.lo[a] + 0.5f * b.hi[a] - clo[a]) * scale);
                k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                grow(bb[k], b); cnt[k]++;
            }
            double rightArea[kBins]; uint32_t rightCnt[kBins];
            Box acc = empty_box(); uint32_t c = 0;
            for (int k = kBins - 1; k > 0; --k) { grow(acc, bb[k]); c += cnt[k]; rightArea[k] = half_area(acc); rightCnt[k] = c; }
            acc = empty_box(); c = 0;
            for (int k = 0; k < kBins - 1; ++k) {
                grow(acc, bb[k]); c += cnt[k];
                if (c == 0 || rightCnt[k + 1] == 0) continue;
                double cost = half_area(acc) * c + rightArea[k + 1] * rightCnt[k + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; }
            }
        }
        uint32_t mid;
        if (bestAxis >= 0) {
            const int a = bestAxis;
            const float scale = (float)kBins / (chi[a] - clo[a]);
            const float lo = clo[a];
            uint32_t* m = std::partition(order + begin, order + end, [&](uint32_t p) {
                const Box& b = prim[p];
                int k = (int)((0.5f * b.lo[a] + 0.5f * b.hi[a] - lo) * scale);
                k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                return k <= bestBin;
            });
            mid = (uint32_t)(m - order);
            if (mid == begin || mid == end) mid = begin + n / 2;
        } else {
            mid = begin + n / 2;      // all centroids coincide (or are not finite): split by count
        }
        const uint32_t nl = mid - begin;
        const int32_t leftIdx = nodeIdx + 1, rightIdx = nodeIdx + (int32_t)nl;   // left subtree has nl-1 inner nodes
        Box bl, br; int dl = 0, dr = 0; int32_t rl, rr;
        if (depth < parallelDepth && n > 8192) {
            std::thread t([&]() { rl = build(begin, mid, leftIdx, depth + 1, bl, dl); });
            rr = build(mid, end, rightIdx, depth + 1, br, dr);
            t.join();
        } else {
            rl = build(begin, mid, leftIdx, depth + 1, bl, dl);
            rr = build(mid, end, rightIdx, depth + 1, br, dr);
        }
        float* q = nodes[nodeIdx].q;
        q[0] = bl.lo[0]; q[1] = bl.lo[1]; q[2] = bl.lo[2]; q[3] = bl.hi[0]; q[4] = bl.hi[1]; q[5] = bl.hi[2];
        q[6] = br.lo[0]; q[7] = br.lo[1]; q[8] = br.lo[2]; q[9] = br.hi[0]; q[10] = br.hi[1]; q[11] = br.hi[2];
        q[12] = i2f(rl); q[13] = i2f(rr); q[14] = 0; q[15] = 0;
        outBox = bl; grow(outBox, br);
        outDepth = std::max(dl, dr);
        return nodeIdx;
    }
};

// order[] in: any permutation of 0..n-1; out: leaf order of the new tree.
// Returns the root ref; nodes must hold n-1 records.
inline int32_t build_sah(const Box* prim, uint32_t n, uint32_t* order, Node* nodes, int nthreads, int& maxDepth) {
    int par = 0;
    while ((1 << par) < nthreads) ++par;
    SahBuilder b{prim, order, nodes, par};
    Box root; int depth = 0;
    int32_t ref = b.build(0, n, 0, 0, root, depth);
    maxDepth = depth;
    return ref;
}

// ---- 4-wide collapse --------------------------------------------------------------
// A traversal step costs the wave one dependent memory round trip plus the fixed overhead of the
// pop / push / leaf sections, so the binary tree is folded into nodes of up to four children.
// Child boxes are copied unchanged (a leaf child keeps its exact leaf box), so the traversal
// semantics do not change; only the number of steps does.
//
// Which binary nodes survive as wide nodes is chosen by dynamic programming over the binary tree
// (the SAH-optimal collapse of Ylitie et al. 2017 restricted to 4 slots): the expected number of
// steps of a random ray is proportional to the summed surface area of the surviving nodes (the leaf
// term is the same for every collapse), so T[n][j] = the least such sum with which the subtree of n
// can be presented to its parent as at most j children:
//     T[n][1] = area(n) + split(n, 4)                     n survives, its own 4 slots filled optimally
//     T[n][j] = min(T[n][1], split(n, j))                 or n dissolves into its parent's slots
//     split(n, j) = min over k of T[left][k] + T[right][j - k];   T[leaf][.] = 0
// Greedy "expand the largest child" leaves 3.0 children per node on the synthetic meshes; this gives
// fuller nodes and fewer, smaller ones.
struct Node4 {          // 128-byte record: lo.x[4] hi.x[4] lo.y[4] hi.y[4] lo.z[4] hi.z[4] ref[4] pad[4]
    float q[32];
};

struct Collapser {
    const Node* bvh2;
    std::vector<Node4>& out;
    std::vector<float> T;      // 4 per binary node: T[4 * n + j - 1]
    int maxDepth = 0;

    struct Child { Box box; int32_t ref; };

    static void children_of(const Node& n, Child c[2]) {
        for (int k = 0; k < 2; ++k) {
            for (int a = 0; a < 3; ++a) { c[k].box.lo[a] = n.q[6 * k + a]; c[k].box.hi[a] = n.q[6 * k + 3 + a]; }
            c[k].ref = f2i(n.q[12 + k]);
        }
    }
    float t(int32_t ref, int j) const { return ref < 0 ? 0.0f : T[4 * (size_t)ref + (size_t)(j - 1)]; }
    // best way to hand `j` slots to the two children: returns the cost, k = slots of the left child
    float split(const Child c[2], int j, int& k) const {
        float best = INFINITY; k = 1;
        for (int i = 1; i < j; ++i) { const float v = t(c[0].ref, i) + t(c[1].ref, j - i); if (v < best) { best = v; k = i; } }
        return best;
    }

    // children point at higher indices than their parent (build_sah numbers nodes in pre-order)
    void solve(size_t numNodes) {
        T.assign(4 * numNodes, 0.0f);
        for (size_t n = numNodes; n-- > 0;) {
            Child c[2]; children_of(bvh2[n], c);
            Box u = c[0].box; grow(u, c[1].box);
            int k;
            const float s2 = split(c, 2, k), s3 = split(c, 3, k), s4 = split(c, 4, k);
            const float keep = (float)half_area(u) + s4;
            T[4 * n + 0] = keep; T[4 * n + 1] = std::min(keep, s2); T[4 * n + 2] = std::min(keep, s3); T[4 * n + 3] = std::min(keep, s4);
        }
    }

    // the children that subtree `ref` contributes when it may use up to j slots
    void collect(const Child& me, int j, Child* slots, int& n) const {
        if (me.ref < 0 || j == 1 || !(t(me.ref, j) < t(me.ref, 1))) { slots[n++] = me; return; }
        Child c[2]; children_of(bvh2[me.ref], c);
        int k; split(c, j, k);
        collect(c[0], k, slots, n);
        collect(c[1], j - k, slots, n);
    }

    // emits the wide node for binary node `ref2` (>= 0) and returns its index
    int32_t emit(int32_t ref2, int depth) {
        maxDepth = std::max(maxDepth, depth);
        Child c2[2]; children_of(bvh2[ref2], c2);
        int k; split(c2, 4, k);
        Child c[4]; int n = 0;
        collect(c2[0], k, c, n);
        collect(c2[1], 4 - k, c, n);
        const int32_t idx = (int32_t)out.size();
        out.emplace_back();
        int32_t refs[4];
        for (int i = 0; i < 4; ++i) {
            if (i < n) refs[i] = c[i].ref >= 0 ? emit(c[i].ref, depth + 1) : c[i].ref;
            else refs[i] = (int32_t)0x80000000;                       // RTX_NO_CHILD
        }
        float* q = out[idx].q;
        for (int i = 0; i < 4; ++i) {
            const bool used = i < n;
            for (int a = 0; a < 3; ++a) {
                q[8 * a + i] = used ? c[i].box.lo[a] : NAN;           // an empty slot never passes the slab test:
                q[8 * a + 4 + i] = used ? c[i].box.hi[a] : NAN;       // every comparison with its NaN distances is false
            }
            q[24 + i] = i2f(refs[i]);
            q[28 + i] = 0.0f;
        }
        return idx;
    }
};

// Folds the binary tree of `numNodes` inner nodes rooted at node `root2` into 4-wide nodes (root = index 0).
inline void collapse4(const Node* bvh2, size_t numNodes, int32_t root2, std::vector<Node4>& out, int& maxDepth) {
    Collapser c{bvh2, out};
    c.solve(numNodes);
    c.emit(root2, 0);
    maxDepth = c.maxDepth;
}

// A wide node holding a single leaf (scenes with one primitive) or nothing at all.
inline void single_leaf_node4(const Box* leafBox, std::vector<Node4>& out) {
    out.emplace_back();
    float* q = out.back().q;
    for (int k = 0; k < 4; ++k) {
        const bool used = leafBox && k == 0;
        for (int a = 0; a < 3; ++a) { q[8 * a + k] = used ? leafBox->lo[a] : NAN; q[8 * a + 4 + k] = used ? leafBox->hi[a] : NAN; }
        q[24 + k] = i2f(used ? ~0 : (int32_t)0x80000000);
        q[28 + k] = 0.0f;
    }
}

} // namespace rtxbuild
