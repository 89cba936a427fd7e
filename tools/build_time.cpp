// build_time — host-side phases of rtx_create's own-tree build (rtx_build.h) on a synthetic height-field-like set of
// boxes: binned-SAH build, 4-wide fold, quantisation. Usage: ./build_time [primitives=10000000] [threads=16]
#include "../raytracer_amd/csrc/rtx_build.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace rtxbuild;
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
        const int32_t root = build_sah(prim.data(), n, order.data(), bin.data(), threads, d);
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
