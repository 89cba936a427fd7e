// Host-side check of raytracer_amd/csrc/rtx_build.h (binned-SAH build + 4-wide fold + byte-box quantisation), compiled and run by
// tests/test_build_tree.py. Properties: every leaf is referenced exactly once, every child box stored in a
// wide node is the exact union of the leaf boxes below it, empty slots are NaN + RTX_NO_CHILD, the reported
// depth is the real depth, and the fold never needs more nodes than the binary tree had.
#include "../../raytracer_amd/csrc/rtx_build.h"
#include <cstdio>
#include <cstdlib>
#include <random>

using namespace rtxbuild;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { if (fails++ < 10) printf("FAIL line %d: %s\n", __LINE__, #c); } } while (0)

struct Walker {
    const Raw<Node4>& nodes; const std::vector<Box>& prim; const std::vector<uint32_t>& order;
    std::vector<int> seen; int maxDepth = 0; size_t visited = 0;
    Box walk(int32_t ref, int depth) {
        if (ref < 0) {
            const uint32_t pos = (uint32_t)~ref;
            CHECK(pos < order.size());
            seen[pos]++;
            return prim[order[pos]];
        }
        visited++;
        maxDepth = std::max(maxDepth, depth);
        const float* q = nodes[(size_t)ref].q;
        Box u = empty_box(); int used = 0;
        for (int k = 0; k < 4; ++k) {
            const int32_t r = f2i(q[24 + k]);
            if (r == (int32_t)0x80000000) { for (int a = 0; a < 3; ++a) { CHECK(std::isnan(q[8 * a + k])); CHECK(std::isnan(q[8 * a + 4 + k])); } continue; }
            used++;
            const Box below = walk(r, depth + 1);
            for (int a = 0; a < 3; ++a) { CHECK(q[8 * a + k] == below.lo[a]); CHECK(q[8 * a + 4 + k] == below.hi[a]); }
            grow(u, below);
        }
        CHECK(used >= 2 || nodes.size() == 1);
        return u;
    }
};

// quantize4 (RTX_TRAVERSE_WIDE's 64-byte nodes): same tree, children regrouped (inner first, then leaves), a node's
// inner children neighbours in the node array and its leaf children neighbours in the new leaf order; every DECODED
// box — fma(byte, 2^e, origin), the operation the kernel performs — contains the float box of the child it stands for
// (the property the traversal's exactness argument rests on), and is at most one grid step larger on each side.
static void check_quantised(const Raw<Node4>& wide, uint32_t n, int threads, float huge = 0.0f) {
    Raw<NodeQ> q; std::vector<uint32_t> newToOld;
    const bool ok = quantize4(wide, q, newToOld, threads);
    CHECK(ok);
    if (!ok) return;
    CHECK(q.size() == wide.size());
    CHECK(newToOld.size() == n);
    std::vector<int> seenLeaf(n, 0);
    for (uint32_t p : newToOld) { CHECK(p < n); if (p < n) seenLeaf[p]++; }
    for (uint32_t i = 0; i < n; ++i) CHECK(seenLeaf[i] == 1);
    // walk both trees together: (old Node4 index, new NodeQ index)
    std::vector<std::pair<int32_t, int32_t>> st{{0, 0}};
    std::vector<int> seenNode(q.size(), 0);
    size_t visited = 0;
    while (!st.empty()) {
        auto [o, nw] = st.back(); st.pop_back();
        visited++;
        CHECK(nw >= 0 && (size_t)nw < q.size());
        seenNode[(size_t)nw]++;
        const float* w = wide[(size_t)o].q;
        const NodeQ& node = q[(size_t)nw];
        const uint32_t meta = node.w[3], valid = (meta >> 24) & 15u, leaf = meta >> 28;
        // old children in the order quantize4 regroups them: inner first, then leaves
        int slotOf[4], nc = 0, ni = 0;
        for (int k = 0; k < 4; ++k) if (f2i(w[24 + k]) >= 0) { slotOf[nc++] = k; ni++; }
        for (int k = 0; k < 4; ++k) { const int32_t r = f2i(w[24 + k]); if (r < 0 && r != (int32_t)0x80000000) slotOf[nc++] = k; }
        CHECK(valid == (1u << nc) - 1u);
        CHECK(leaf == (valid & ~((1u << ni) - 1u)));
        const int32_t baseInner = (int32_t)node.w[10], baseLeaf = (int32_t)node.w[11];
        for (int j = 0; j < nc; ++j) {
            const int k = slotOf[j];
            for (int a = 0; a < 3; ++a) {
                const float org = i2f((int32_t)node.w[a]);
                const float step = i2f((int32_t)(((meta >> (8 * a)) & 0xffu) << 23));
                const float lo = std::fmaf((float)((node.w[4 + 2 * a] >> (8 * j)) & 0xffu), step, org);
                const float hi = std::fmaf((float)((node.w[5 + 2 * a] >> (8 * j)) & 0xffu), step, org);
                CHECK(lo <= w[8 * a + k]);
                CHECK(hi >= w[8 * a + 4 + k]);
                // tight: within one step (plus the rounding of the decode) of the child's box
                CHECK((double)w[8 * a + k] - (double)lo <= 1.01 * (double)step + 1e-6 * std::fabs((double)w[8 * a + k]) + huge);
                CHECK((double)hi - (double)w[8 * a + 4 + k] <= 1.01 * (double)step + 1e-6 * std::fabs((double)w[8 * a + 4 + k]) + huge);
            }
            const int32_t r = f2i(w[24 + k]);
            if (r >= 0) st.push_back({r, baseInner + j});
            else CHECK(newToOld[(size_t)(baseLeaf + j)] == (uint32_t)~r);
        }
    }
    CHECK(visited == q.size());
    for (int c : seenNode) CHECK(c == 1);
}

static void run(uint32_t n, unsigned seed, int threads, bool degenerate) {
    std::mt19937 rng(seed);
    auto uni = [&](float lo, float hi) { return lo + (hi - lo) * (float)(rng() >> 8) * (1.0f / 16777216.0f); };
    std::vector<Box> prim(n);
    for (auto& b : prim) {
        const float cx = degenerate ? 5.0f : uni(-100, 100), cy = degenerate ? 5.0f : uni(-100, 100), cz = degenerate ? 5.0f : uni(0, 50);
        const float s = (rng() % 50 == 0) ? uni(20, 80) : uni(0.01f, 1.0f);      // a few huge primitives
        b.lo[0] = cx - s; b.lo[1] = cy - s; b.lo[2] = cz - s; b.hi[0] = cx + s; b.hi[1] = cy + s; b.hi[2] = cz + s;
    }
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::vector<Node> bin(n - 1);
    int binDepth = 0;
    const int32_t root = build_sah(prim.data(), n, order.data(), bin.data(), threads, binDepth);
    CHECK(root == 0);
    std::vector<uint32_t> sorted = order; std::sort(sorted.begin(), sorted.end());
    for (uint32_t i = 0; i < n; ++i) CHECK(sorted[i] == i);                       // still a permutation
    Raw<Node4> wide; int depth4 = 0;
    collapse4(bin.data(), bin.size(), root, wide, depth4, threads);
    CHECK(wide.size() <= bin.size());
    Walker w{wide, prim, order, std::vector<int>(n, 0)};
    w.walk(0, 0);
    for (uint32_t i = 0; i < n; ++i) CHECK(w.seen[i] == 1);
    CHECK(w.visited == wide.size());
    CHECK(w.maxDepth == depth4);
    CHECK(depth4 <= binDepth);
    check_quantised(wide, n, threads);
    printf("n=%u seed=%u threads=%d%s: %zu binary -> %zu wide nodes (%.2f children/node), depth %d -> %d\n", n, seed, threads,
           degenerate ? " (coincident centroids)" : "", bin.size(), wide.size(), (double)(n + wide.size() - 1) / wide.size(), binDepth, depth4);
}

int main() {
    run(2, 1, 1, false); run(3, 2, 1, false); run(5, 3, 2, false); run(1000, 4, 4, false); run(200000, 5, 8, false);
    run(600000, 7, 5, false);      // uneven thread shares: subtrees change hands between threads (SahBuilder offers)
    run(4097, 6, 4, true);
    Raw<Node4> one; Box b; for (int a = 0; a < 3; ++a) { b.lo[a] = 1; b.hi[a] = 2; }
    single_leaf_node4(&b, one);
    CHECK(one.size() == 1 && f2i(one[0].q[24]) == ~0 && f2i(one[0].q[25]) == (int32_t)0x80000000 && std::isnan(one[0].q[1]));
    check_quantised(one, 1, 1);
    Raw<Node4> none; single_leaf_node4(nullptr, none);
    check_quantised(none, 0, 1);
    {   // extreme coordinates: a huge box next to a tiny one far from the origin, and an infinite extent (must be refused)
        std::vector<Box> prim(3);
        // (+-2^40 is the range rtx_create quantises: the walk's folded inner test is proven overflow-free inside it)
        const float v[3][2] = {{-1.0e12f, 1.0e12f}, {1.0e6f, 1.0e6f + 0.125f}, {-1.0e-30f, 1.0e-30f}};
        for (int i = 0; i < 3; ++i) for (int a = 0; a < 3; ++a) { prim[i].lo[a] = v[i][0]; prim[i].hi[a] = v[i][1]; }
        std::vector<uint32_t> order{0, 1, 2}; std::vector<Node> bin(2); int d = 0;
        const int32_t root = build_sah(prim.data(), 3, order.data(), bin.data(), 1, d);
        Raw<Node4> wide; int d4 = 0; collapse4(bin.data(), bin.size(), root, wide, d4, 1);
        check_quantised(wide, 3, 1, 1e9f);
        {   // beyond the range: refused (the caller keeps the float nodes)
            std::vector<Box> far(prim); far[0].lo[2] = -3.0e37f; far[0].hi[2] = 3.0e37f;
            std::vector<uint32_t> o2{0, 1, 2}; std::vector<Node> b2(2); int dd = 0, dd4 = 0;
            const int32_t r2 = build_sah(far.data(), 3, o2.data(), b2.data(), 1, dd);
            Raw<Node4> w2; collapse4(b2.data(), b2.size(), r2, w2, dd4, 1);
            Raw<NodeQ> q2; std::vector<uint32_t> n2;
            CHECK(!quantize4(w2, q2, n2, 1));
        }
        CHECK(q_entry_base(q_entry(kQuantMaxLeaves + 3, 15)) == kQuantMaxLeaves + 3 && q_entry_base(q_entry(-4, 15)) == -4 && (q_entry(-4, 15) & 15) == 15);
        CHECK(q_entry_base(q_entry(1 << 27, 1)) != (1 << 27));      // where the encoding wraps: kQuantMaxLeaves stays below it
        prim[0].hi[1] = INFINITY;
        order = {0, 1, 2};
        const int32_t root2 = build_sah(prim.data(), 3, order.data(), bin.data(), 1, d);
        wide.clear(); collapse4(bin.data(), bin.size(), root2, wide, d4, 1);
        Raw<NodeQ> q; std::vector<uint32_t> n2o;
        CHECK(!quantize4(wide, q, n2o, 1));
    }
    if (fails) { printf("%d check(s) failed\n", fails); return 1; }
    printf("OK\n");
    return 0;
}
