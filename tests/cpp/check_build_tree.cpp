// Host-side check of raytracer_amd/csrc/rtx_build.h (binned-SAH build + 4-wide fold), compiled and run by
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
    const std::vector<Node4>& nodes; const std::vector<Box>& prim; const std::vector<uint32_t>& order;
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
    std::vector<Node4> wide; int depth4 = 0;
    collapse4(bin.data(), bin.size(), root, wide, depth4, threads);
    CHECK(wide.size() <= bin.size());
    Walker w{wide, prim, order, std::vector<int>(n, 0)};
    w.walk(0, 0);
    for (uint32_t i = 0; i < n; ++i) CHECK(w.seen[i] == 1);
    CHECK(w.visited == wide.size());
    CHECK(w.maxDepth == depth4);
    CHECK(depth4 <= binDepth);
    printf("n=%u seed=%u threads=%d%s: %zu binary -> %zu wide nodes (%.2f children/node), depth %d -> %d\n", n, seed, threads,
           degenerate ? " (coincident centroids)" : "", bin.size(), wide.size(), (double)(n + wide.size() - 1) / wide.size(), binDepth, depth4);
}

int main() {
    run(2, 1, 1, false); run(3, 2, 1, false); run(5, 3, 2, false); run(1000, 4, 4, false); run(200000, 5, 8, false);
    run(4097, 6, 4, true);
    std::vector<Node4> one; Box b; for (int a = 0; a < 3; ++a) { b.lo[a] = 1; b.hi[a] = 2; }
    single_leaf_node4(&b, one);
    CHECK(one.size() == 1 && f2i(one[0].q[24]) == ~0 && f2i(one[0].q[25]) == (int32_t)0x80000000 && std::isnan(one[0].q[1]));
    if (fails) { printf("%d check(s) failed\n", fails); return 1; }
    printf("OK\n");
    return 0;
}
