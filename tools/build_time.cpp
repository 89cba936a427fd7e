// build_time — host-side phases of rtx_create's own-tree build (rtx_build.h) on a synthetic height-field-like set of
// boxes: binned-SAH build, 4-wide fold, quantisation. Usage: ./build_time [primitives=10000000] [threads=16]
#include "../raytracer_amd/csrc/rtx_build.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <algorithm>
using namespace rtxbuild;
// ---- evaluation only: the topology a device-side LBVH build (Morton sort + Karras) would produce, on the host ----
static inline uint64_t spread21(uint64_t v) { v &= 0x1fffffull; v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full; v = (v | v << 4) & 0x10c30c30c30c30c3ull; v = (v | v << 2) & 0x1249249249249249ull; return v; }
struct Lbvh {
    const Box* prim; std::vector<std::pair<uint64_t, uint32_t>> keys; Node* nodes;
    int32_t rec(uint32_t begin, uint32_t end, int32_t nodeIdx, int bit, int depth, Box& out, int& outDepth) {
        if (end - begin == 1) { out = prim[keys[begin].second]; outDepth = depth; return ~(int32_t)begin; }
        uint32_t mid = begin;
        while (bit >= 0) {
            const uint64_t m = 1ull << bit;
            if ((keys[begin].first & m) == (keys[end - 1].first & m)) { --bit; continue; }
            uint32_t lo = begin, hi = end - 1;      // first index with the bit set
            while (hi - lo > 1) { const uint32_t c = lo + (hi - lo) / 2; if (keys[c].first & m) hi = c; else lo = c; }
            mid = hi; --bit; break;
        }
        if (mid == begin) mid = begin + (end - begin) / 2;      // identical codes
        const uint32_t nl = mid - begin;
        Box bl, br; int dl = 0, dr = 0;
        const int32_t rl = rec(begin, mid, nodeIdx + 1, bit, depth + 1, bl, dl), rr = rec(mid, end, nodeIdx + (int32_t)nl, bit, depth + 1, br, dr);
        float* q = nodes[nodeIdx].q;
        for (int a = 0; a < 3; ++a) { q[a] = bl.lo[a]; q[3 + a] = bl.hi[a]; q[6 + a] = br.lo[a]; q[9 + a] = br.hi[a]; }
        q[12] = i2f(rl); q[13] = i2f(rr); q[14] = 0; q[15] = 0;
        out = bl; grow(out, br); outDepth = std::max(dl, dr);
        return nodeIdx;
    }
};
static int32_t build_lbvh(const Box* prim, uint32_t n, uint32_t* order, Node* nodes, int& maxDepth) {
    Box cb = empty_box();
    for (uint32_t i = 0; i < n; ++i) { Box c; for (int a = 0; a < 3; ++a) c.lo[a] = c.hi[a] = 0.5f * prim[i].lo[a] + 0.5f * prim[i].hi[a]; grow(cb, c); }
    Lbvh L; L.prim = prim; L.nodes = nodes; L.keys.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t code = 0;
        for (int a = 0; a < 3; ++a) {
            const double ext = (double)cb.hi[a] - cb.lo[a], c = 0.5 * prim[i].lo[a] + 0.5 * prim[i].hi[a];
            double u = ext > 0 ? (c - cb.lo[a]) / ext : 0.0; u = u < 0 ? 0 : (u > 1 ? 1 : u);
            uint64_t k = (uint64_t)(u * 2097151.0);
            code |= spread21(k) << (2 - a);
        }
        if (getenv("LBVH_HOIST")) {       // primitives wider than 1/8 of the scene on some axis sort behind all others: the root split sets them apart
            bool large = false; for (int a = 0; a < 3; ++a) large = large || (prim[i].hi[a] - prim[i].lo[a]) > 0.125f * (cb.hi[a] - cb.lo[a]);
            code = (code >> 1) | (large ? 1ull << 62 : 0);
        }
        L.keys[i] = {code, i};
    }
    std::sort(L.keys.begin(), L.keys.end());
    Box root; int d = 0;
    const int32_t r = L.rec(0, n, 0, 62, 0, root, d);
    for (uint32_t i = 0; i < n; ++i) order[i] = L.keys[i].second;
    maxDepth = d; return r;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 10000000u; const int threads = argc > 2 ? atoi(argv[2]) : 16;
    std::mt19937 rng(1);
    auto uni = [&](float lo, float hi) { return lo + (hi - lo) * (float)(rng() >> 8) * (1.0f / 16777216.0f); };
    Raw<Box> prim(n);
    if (argc > 3) {            // boxes from a file (n x 6 floats: lo.xyz hi.xyz), e.g. a scene's leaf boxes in the caller's order
        FILE* f = fopen(argv[3], "rb");
        if (!f || fread(prim.data(), sizeof(Box), n, f) != n) { fprintf(stderr, "cannot read %u boxes from %s\n", n, argv[3]); return 1; }
        fclose(f);
    } else {
    const uint32_t side = (uint32_t)std::sqrt((double)n);
    for (uint32_t i = 0; i < n; ++i) {
        const float cx = (i % side) * 0.5f + uni(0, .3f), cy = (i / side) * 0.5f + uni(0, .3f), cz = std::sin(cx * 0.05f) * 20 + uni(0, .2f), s = 0.3f;
        prim[i].lo[0] = cx - s; prim[i].lo[1] = cy - s; prim[i].lo[2] = cz - s; prim[i].hi[0] = cx + s; prim[i].hi[1] = cy + s; prim[i].hi[2] = cz + s;
    }
    }
    for (int rep = 0; rep < 2; ++rep) {
        std::vector<uint32_t> order(n); for (uint32_t i = 0; i < n; ++i) order[i] = i;
        double t0 = now();
        Raw<Node> bin(n - 1); int d = 0;
        const bool lbvh = argc > 4 && std::string(argv[4]) == "lbvh";
        const int32_t root = lbvh ? build_lbvh(prim.data(), n, order.data(), bin.data(), d) : build_sah(prim.data(), n, order.data(), bin.data(), threads, d);
        double t1 = now();
        Raw<Node4> wide; int d4 = 0;
        collapse4(bin.data(), bin.size(), root, wide, d4, threads);
        double t2 = now();
        {   // expected node steps of a random line: summed surface area of the wide nodes over the root's (what the fold minimises)
            auto node_box = [&](const Node4& nd, int& kids) { Box u = empty_box(); kids = 0;
                for (int k = 0; k < 4; ++k) { if (f2i(nd.q[24 + k]) == (int32_t)0x80000000) continue; Box c; for (int a = 0; a < 3; ++a) { c.lo[a] = nd.q[8 * a + k]; c.hi[a] = nd.q[8 * a + 4 + k]; } grow(u, c); ++kids; }
                return u; };
            int kids = 0; const double rootArea = half_area(node_box(wide[0], kids));
            double sum = 0, leafSum = 0; size_t children = 0;
            for (const Node4& nd : wide) { sum += half_area(node_box(nd, kids)); children += (size_t)kids;
                for (int k = 0; k < 4; ++k) { const int32_t r = f2i(nd.q[24 + k]); if (r < 0 && r != (int32_t)0x80000000) { Box c; for (int a = 0; a < 3; ++a) { c.lo[a] = nd.q[8 * a + k]; c.hi[a] = nd.q[8 * a + 4 + k]; } leafSum += half_area(c); } } }
            printf("   SAH: expected wide-node steps of a random line %.2f, leaf boxes pierced %.2f (%.2f children per node)\n", sum / rootArea, leafSum / rootArea, (double)children / wide.size());
        }
        Raw<NodeQ> q; std::vector<uint32_t> n2o;
        double ph[3] = {0, 0, 0};
        const bool ok = quantize4(wide, q, n2o, threads, ph);
        double t3 = now();
        printf("%u primitives, %d threads: SAH %.3f s (depth %d), fold %.3f s (%zu nodes, depth %d), quantise %.3f s = sizes %.3f + numbering %.3f + boxes %.3f (%s)\n", n, threads, t1 - t0, d, t2 - t1,
               wide.size(), d4, t3 - t2, ph[0], ph[1], ph[2], ok ? "ok" : "refused");
    }
    return 0;
}
