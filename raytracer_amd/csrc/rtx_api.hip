// rtx_api.hip — host side of the C-ABI in include/rtx.h: validation, AoS -> SoA
// flattening of the reference's boxes/objects, device residency, and the
// per-frame launch sequence. No torch types, no CPU rendering fallback: every
// entry point that computes anything launches HIP kernels or fails.
#include "../../include/rtx.h"
#include "rtx_kernels.h"
#include "rtx_build.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <deque>
#include <mutex>
#include <thread>
#include <chrono>
#include <vector>

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) { (void)hipGetLastError(); return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)

// ---- scratch of rtx_trace_rays_noalloc --------------------------------------------------------------------------
// One file-scope device array per GPU (every device that loads this code object holds its own copy at its own
// address, until the process ends): work counters, hit distances, hit leaves, the list of rays for the
// reference-topology side kernel, and the staged directions of up to RTX_NOALLOC_MAX_RAYS rays. In host code the
// array's NAME is only a handle for hipGetSymbolAddress; kernels and memsets get the per-device address g_scratch[dev].
#define RTX_NOALLOC_MAX_RAYS (1u << 19)
#define RTX_NOALLOC_CTRL_BYTES ((RTX_CTRL_WORDS * 4u + 255u) & ~255u)
#define RTX_NOALLOC_BYTES (RTX_NOALLOC_CTRL_BYTES + (size_t)RTX_NOALLOC_MAX_RAYS * (4u + 4u + 4u + 16u))     // 14.0 MiB + 8.25 KiB
static __device__ __align__(256) unsigned char scratch[RTX_NOALLOC_BYTES];
static const int kMaxDevices = 64;
static unsigned char* g_scratch[kMaxDevices] = {};      // per device: address of that device's copy of scratch[]
static std::mutex g_scratchMutex;

// work-list entries of the quantised walk at the limit flatten() enforces (rtx_traverse.h q_pass decodes with >> 4)
static_assert((RTX_QENTRY(rtxbuild::kQuantMaxLeaves + 3, 15) >> 4) == rtxbuild::kQuantMaxLeaves + 3, "largest leaf base must survive the entry encoding");
static_assert((RTX_QENTRY(-4, 15) >> 4) == -4 && (RTX_QENTRY(-4, 15) & 15) == 15, "pre-biased (negative) bases must survive the entry encoding");
static_assert((RTX_QENTRY(1 << 27, 1) >> 4) != (1 << 27), "2^27 is where the encoding wraps: kQuantMaxLeaves must stay below it");
static_assert(RTX_QENTRY(5, 3) == rtxbuild::q_entry(5, 3) && rtxbuild::q_entry_base(rtxbuild::q_entry(-2, 9)) == -2, "host mirror of the entry encoding");

// The runtime's first steps on a device — its initialisation, and loading this library's code object there, which the
// first symbol lookup or kernel launch on a device triggers — are taken by the thread that calls rtx_create, before
// any helper thread makes a HIP call. (Leaving them to a thread that uploads the raw arrays while the caller's thread
// went on to its own first HIP calls ended, about once in twenty fresh processes, in a segmentation fault inside that
// first rtx_create: tests/spill_check.py run in a loop.) Also records where the device keeps its copy of scratch[].
static int ensure_device_ready(int device) {
    if (device < 0 || device >= kMaxDevices) return fail("device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> g(g_scratchMutex);
    if (!g_scratch[device]) {
        unsigned char* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        HIP_TRY(hipDeviceSynchronize());
        g_scratch[device] = ptr;
    }
    return 0;
}

struct RtxRenderer;
static const int kRing = 32;      // frames whose timing events are kept; a frame's timing is read kLag frames later
static const int kLag = 16;
static const int kEvents = 10;    // start + one per kernel (8) + end

struct RtxRenderer {
    RtxConfig cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    int numCUs = 256;
    uint32_t width = 0, height = 0;
    int mode = RTX_MODE_CULLED;
    bool linear = false;

    DevScene sc{};
    // device allocations
    float4 *dFast = nullptr;
    uint4* dQNodes = nullptr;
    int2* dSpill = nullptr;
    uint32_t* dSlow = nullptr;
    size_t ldsSlow = 0;
    float4 *dInner = nullptr, *dLeaf = nullptr, *dShade = nullptr, *dMaterials = nullptr;
    int4* dTexInfo = nullptr;
    uint32_t* dTexels = nullptr;
    int2* dTiles = nullptr;
    float4 *dRqO[2] = {nullptr, nullptr}, *dRqD[2] = {nullptr, nullptr};
    int2* dHit = nullptr;                 // per queue entry: (bits of the hit distance, leaf)
    float4 *dLight = nullptr, *dAcc = nullptr, *dFb = nullptr;
    uint32_t *dSeeds = nullptr, *dCtrl = nullptr;
    unsigned long long* dStats = nullptr;
    unsigned long long* curStats = nullptr;    // where the launches of the current set count: dStats, or the context's look-ahead counters

    // tiles
    int tileW = 64, tileH = 32, tilesX = 0, tilesY = 0;
    std::vector<int2> localTiles;
    std::vector<std::vector<int2>> tilesOfRank;   // for unpack
    std::vector<int2*> dTilesOfRank;
    uint64_t localPixels = 0;
    uint32_t numSlots = 0;

    // launch geometry
    int gridTrace = 0, gridShade = 0;       // blocks that are resident at once: the largest grids ever launched
    int launchTrace = 0, launchShade = 0;   // grids of the next launch (set_grids): smaller for short queues
    uint32_t queueHint[4] = {0, 0, 0, 0};   // expected queue length per bounce, from the last frame whose counts arrived
    size_t ldsTrace = 0;
    bool leafQueue = false;          // bounce launches test leaves on compacted lanes (k_trace_qw)
    int lateBlocksPerCU = 4;         // blocks per CU of the launches of bounces 2 and 3 (k_trace_qw)

    // Frames in flight: each has its own stream and its own per-path state / queues / work counters, so that
    // the thin tails of one frame's launches overlap with the next frame's work. The fields above
    // (stream, dRqO ... dCtrl, dSpill, dSlow, sc.spill) always alias the context selected by activate().
    struct FrameCtx {
        hipStream_t stream = nullptr;
        float4 *dRqO[2] = {nullptr, nullptr}, *dRqD[2] = {nullptr, nullptr};
        int2* dHit = nullptr;                 // per queue entry: (bits of the hit distance, leaf)
        float4 *dLight = nullptr, *dAcc = nullptr;
        uint32_t *dSeeds = nullptr, *dCtrl = nullptr, *dSlow = nullptr;
        int2* dSpill = nullptr;
        hipEvent_t resolved = nullptr;     // recorded after the context's latest k_resolve
        uint32_t* hCounts = nullptr;       // pinned: the frame's queue lengths, copied back without a sync
        hipEvent_t countsReady = nullptr;
        bool countsPending = false;
        int countsFrames = 1;              // how many frames those counts belong to
        unsigned long long* dStatsAhead = nullptr;   // work counters of a look-ahead frame, added to dStats when the frame is asked for
        int capFrames = 0;                 // frames of a batch this context's queues and path state hold (alloc_ctx)
        uint64_t queueBytes = 0;           // what alloc_ctx asked the device for
        hipEvent_t abortDone = nullptr;    // recorded on the control stream behind an abort of this context's set (abort_ahead)
        bool abortPending = false;         // the context's next set of launches has to wait for abortDone before it clears ctrl[]
    };
    std::vector<FrameCtx> ctx;
    int active = 0;
    int lastResolved = -1;               // context whose k_resolve was queued last
    uint64_t frameSeq = 0;
    uint64_t primaryRays = 0;            // rays of bounce 0 since the last reset (known on the host)
    uint64_t deviceBytes = 0;            // device memory allocated by rtx_create
    int capFrames = 0;                   // frames of a batch EVERY context's queues and path state hold right now (alloc_queues / grow_queues / grow_ctx)
    bool needSlow = false;               // the contexts carry a side list for k_trace_slow

    // Frame batches: a rank that owns 1/N of the tiles has queues 1/N as long, and short launches are mostly
    // tail. Frames queued with pixels == NULL are therefore collected and up to `batch` of them rendered by one
    // set of launches (DevBatch); every entry point that needs results flushes the collection first.
    int batch = 1;
    std::vector<std::pair<RtxCamera, uint32_t>> queued;

    // timing
    hipEvent_t ev[kRing][kEvents];
    bool pending[kRing] = {};
    bool aheadSlot[kRing] = {};      // the slot times a frame started ahead: its timing counts only once the frame is asked for
    int lastFinished = -1;           // ring slot of the frame whose timing was read last
    int ringPos = 0;
    RtxCounters totals{};
    RtxInfo info{};
    // device group (rtx_create with numDevices > 1): the leader (member 0) lists the other members; each of those holds
    // a packed-tile buffer on its own device and one on the leader's
    std::vector<RtxRenderer*> peers;
    RtxRenderer* leader = nullptr;
    float* dPackedOwn = nullptr;
    float* dPackedAtLeader = nullptr;
    hipEvent_t arrived = nullptr;   // member: its packed tiles have landed on the leader's device
    int leaderDevice = -1;
    uint32_t* dPresent = nullptr;    // rtx_present_rgba8: the 8-bit image on the device (allocated on first use)
    std::vector<float> hostStage;   // D2H staging for worldSize > 1
    char* hStage = nullptr;         // page-locked staging for delivering the frame to a (pageable) host array
    size_t hStageBytes = 0;
    std::vector<hipEvent_t> stageEv;

    // Look-ahead for the synchronous pattern (the reference's loop: Render(pixels); frameCounter++, src/WindowManager.cpp:154-158;
    // or rtx_render(NULL) + rtx_present_rgba8 / rtx_download_pixels per frame).
    // A call that has to hand the frame back cannot overlap with the next one — unless the next one is started before it
    // is asked for. When two frames in a row were (camera, n) and (camera, n + 1), each delivered before the next was asked
    // for, frames n + 2 (and n + 3) are queued on the next contexts before frame n + 1 is copied out: everything but their
    // resolve, with work counters of their own. The next rtx_render either asks for exactly the oldest of them (its resolve
    // is queued, its counters are added, and the look-ahead is topped up at the next delivery) or for something else (they
    // are dropped: their launches run out unused, nothing they wrote is ever read). Every other entry point drops them too.
    struct Ahead { RtxCamera cam{}; uint32_t fc = 0; int ctx = -1; DevBatch bt{}; int ringSlot = -1; int index = 0; };   // frame `index` of the batch `bt` started on context `ctx`
    std::deque<Ahead> ahead;            // frames started ahead, oldest first: consecutive frame counters, one camera
    bool haveLast = false; RtxCamera lastCam{}; uint32_t lastFc = 0;      // the last frame asked for
    int streak = 0;                     // calls in a row that asked for (same camera, frame counter + 1)
    int rendersSinceDelivery = 0;       // the loop is: one frame, one delivery
    int lookAhead = 0;                  // how many sets of launches are started ahead (0: never)
    int aheadBatch = 1;                 // most frames per set started ahead (1 with countStats: visit counters are per set)
    int aheadNext = 1;                  // frames of the next set: 1 when a loop is recognised, doubling with every set started while it lasts
    hipStream_t ctl = nullptr;          // control stream (highest priority, a hardware queue of its own): aborts of sets started for nothing
    uint64_t abortedSets = 0;
};

// ---------------------------------------------------------------------------------
// validation + flatten
// ---------------------------------------------------------------------------------
namespace {

struct Flat {
    std::vector<float4> inner, leaf, shade;
    rtxbuild::Raw<rtxbuild::Node4> fast;  // own 4-wide SAH tree over the same leaves (empty: not built)
    rtxbuild::Raw<rtxbuild::NodeQ> qnodes; // the same tree with byte boxes (RTX_TRAVERSE_WIDE); `fast` is dropped when this is built
    int rootRef = 0, fastRoot = 0;
    int maxDepth = 0, fastDepth = 0;
    bool nested = true;                 // every box contains its children (precondition of the own tree)
    double sliverFraction = 0.0;        // triangles with longest edge^2 / (2 area) > 32 (or no area at all)
    double seconds[3] = {0, 0, 0};      // validate + walk, own-tree build, record emission
    // BVH scenes: the records are emitted on the device (k_emit_leaf_records / k_emit_inner_records) from the caller's
    // arrays; the host only plans them: for every leaf position its box and tie-break rank, and the index maps the
    // reference-topology records need. (leaf / shade above are filled on the host only for the linear list.)
    std::vector<int2> leafPlan;
    std::vector<int32_t, rtxbuild::NoInitAlloc<int32_t>> slotOf, posOf;
    size_t numLeaves = 0, numInner = 0;
    bool deviceEmit = false;
};

inline float i2f(int32_t v) { float f; memcpy(&f, &v, 4); return f; }

// tieRank: the leaf's position in a left-child-first walk of the caller's tree. The reference pops the right child
// first and accepts a hit only when strictly closer (src/ThreadedShader.cpp:370,380-392), so among equal t the leaf it
// visits first wins = the one with the LARGEST left-first rank. For trees numbered like BVHTree::Insert numbers them
// (pre-order, left subtree first, src/BVHTree.cpp:160,180-183) the rank orders leaves exactly like their boxID.
void emit_leaf(const RtxObject& o, int32_t objectID, int32_t boxID, int32_t tieRank, const RtxBoundingBox* box, float4* leaf, float4* shade) {
    const RtxVec4 A = o.vertices[0], B = o.vertices[1], C = o.vertices[2];
    float4 old[4];      // the shading record keeps (e2.z objectID boxID type) / (B.x B.y C.x C.y) in its third and fourth quad
    if (o.type == RTX_TRIANGLE) {
        // e1 = B - A, e2 = C - A: the same two subtractions the reference performs
        // first in IntersectTriangle (src/ThreadedShader.cpp:298-299)
        const float e1x = B.x - A.x, e1y = B.y - A.y, e1z = B.z - A.z;
        const float e2x = C.x - A.x, e2y = C.y - A.y, e2z = C.z - A.z;
        old[0] = make_float4(A.x, A.y, A.z, e1x);
        old[1] = make_float4(e1y, e1z, e2x, e2y);
        old[2] = make_float4(e2z, i2f(objectID), i2f(boxID), i2f(o.type));
        old[3] = make_float4(B.x, B.y, C.x, C.y);
    } else {
        old[0] = make_float4(o.position.x, o.position.y, o.position.z, o.radius);
        old[1] = make_float4(0, 0, 0, 0);
        old[2] = make_float4(0, i2f(objectID), i2f(boxID), i2f(o.type));
        old[3] = make_float4(0, 0, 0, 0);
    }
    // traversal record: the primitive, then tie-break rank and type in one word, so that a leaf test reads 40 contiguous
    // bytes. Every type but TRIANGLE is intersected as a sphere (src/ThreadedShader.cpp:364-368).
    const int32_t tag = (int32_t)(((uint32_t)tieRank << 3) | (o.type == RTX_TRIANGLE ? 4u : 0u));
    leaf[0] = old[0];
    leaf[1] = old[1];
    // ... followed by the leaf's own box as the caller gave it (the unordered walk tests it here, rtx_traverse.h q_leaf)
    leaf[2] = make_float4(old[2].x, i2f(tag), box ? box->minimalPosition.x : 0.0f, box ? box->minimalPosition.y : 0.0f);
    leaf[3] = box ? make_float4(box->minimalPosition.z, box->maximalPosition.x, box->maximalPosition.y, box->maximalPosition.z)
                  : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const RtxVec4 n0 = o.normals[0], n1 = o.normals[1], n2 = o.normals[2];
    // the shading record is one 128-byte line of its own: the primitive (hit_normal needs the
    // vertices) followed by the normals and the material, so that k_shade touches ONE scattered line per hit
    // instead of a leaf line plus a 48-byte record that straddles lines a third of the time
    for (int k = 0; k < 4; ++k) shade[k] = old[k];
    shade[4] = make_float4(n0.x, n0.y, n0.z, n1.x);
    shade[5] = make_float4(n1.y, n1.z, n2.x, n2.y);
    shade[6] = make_float4(n2.z, i2f(o.materialID), 0, 0);
    shade[7] = make_float4(0, 0, 0, 0);
}


struct PhaseClock {
    const bool on = getenv("RTX_VERBOSE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    double lap(const char* what) {
        const auto now = std::chrono::steady_clock::now();
        const double sec = std::chrono::duration<double>(now - t).count();
        if (on) fprintf(stderr, "[rtx]   %-28s %.3f s\n", what, sec);
        t = now;
        return sec;
    }
};

// f(begin, end) over [0, n) on up to `threads` host threads
template <class F>
void parallel_for(uint64_t n, int threads, F&& f) {
    if (threads <= 1 || n < 65536) { f((uint64_t)0, n); return; }
    const uint64_t per = (n + (uint64_t)threads - 1) / (uint64_t)threads;
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) {
        const uint64_t b = std::min(n, (uint64_t)t * per), e = std::min(n, b + per);
        if (b < e) pool.emplace_back([&f, b, e]() { f(b, e); });
    }
    f((uint64_t)0, std::min(n, per));
    for (auto& th : pool) th.join();
}

// first error reported by any worker
struct ErrorSlot {
    std::mutex m; std::string msg;
    void set(const std::string& s) { std::lock_guard<std::mutex> g(m); if (msg.empty()) msg = s; }
    bool any() { std::lock_guard<std::mutex> g(m); return !msg.empty(); }
};

int flatten(const RtxSceneDesc& s, bool useBVH, bool wantFast, bool wantQuant, int nthreads, Flat& f) {
    PhaseClock clock;
    const uint64_t nObj = s.numObjects, nBox = s.numBoxes;
    ErrorSlot err;
    {   // one pass over the objects: material range + shape statistics for the culling slack (rtx_traverse.h)
        std::atomic<uint64_t> tris{0}, slivers{0};
        parallel_for(nObj, nthreads, [&](uint64_t b, uint64_t e) {
            uint64_t t = 0, sl = 0;
            for (uint64_t i = b; i < e; ++i) {
                const RtxObject& o = s.objects[i];
                const int32_t m = o.materialID;
                if (m < 0 || (uint64_t)m >= s.numMaterials) { err.set("object " + std::to_string(i) + ": materialID out of range"); return; }
                if (o.type != RTX_TRIANGLE) continue;
                const RtxVec4 &A = o.vertices[0], &B = o.vertices[1], &C = o.vertices[2];
                const double e1[3] = {(double)B.x - A.x, (double)B.y - A.y, (double)B.z - A.z};
                const double e2[3] = {(double)C.x - A.x, (double)C.y - A.y, (double)C.z - A.z};
                const double e3[3] = {(double)C.x - B.x, (double)C.y - B.y, (double)C.z - B.z};
                const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                // sliver: longest edge^2 > 32 x (2 x area), compared squared (no square root per triangle); also NaN / zero area
                const double area2sq = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                const double l2 = std::max({e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2],
                                            e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]});
                ++t;
                if (!(l2 * l2 <= 1024.0 * area2sq)) ++sl;
            }
            tris += t; slivers += sl;
        });
        if (err.any()) return fail(err.msg);
        f.sliverFraction = tris ? (double)slivers / (double)tris : 0.0;
    }
    if (!useBVH) {
        f.leaf.resize(nObj * 4); f.shade.resize(nObj * 8);
        parallel_for(nObj, nthreads, [&](uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) emit_leaf(s.objects[i], (int32_t)i, (int32_t)i, 0, nullptr, &f.leaf[i * 4], &f.shade[i * 8]);
        });
        f.rootRef = 0; f.maxDepth = 0;
        f.numLeaves = (size_t)nObj; f.numInner = 0;
        return 0;
    }
    if (nBox > 0x7fffffffull) return fail("too many boxes");
    // index of every box inside its own array (inner / leaf), in box order
    std::vector<int32_t, rtxbuild::NoInitAlloc<int32_t>> slotOf(nBox);
    std::vector<int32_t, rtxbuild::NoInitAlloc<int32_t>> leafBox;        // reference leaf index -> box index
    int32_t nInner = 0, nLeaf = 0;
    {   // leaves and inner boxes are numbered in box order: count per chunk, prefix, then number (all threads; the
        // boxes are 64 bytes each and this is the first of two streaming reads of them)
        const int T = (nthreads > 1 && nBox >= 65536) ? nthreads : 1;
        const uint64_t per = (nBox + (uint64_t)T - 1) / (uint64_t)T;
        std::vector<int32_t> leavesIn((size_t)T + 1, 0);
        auto chunk = [&](int t, uint64_t& b0, uint64_t& b1) { b0 = std::min(nBox, (uint64_t)t * per); b1 = std::min(nBox, b0 + per); };
        auto run = [&](auto&& body) {
            std::vector<std::thread> pool;
            for (int t = 1; t < T; ++t) pool.emplace_back([&body, t]() { body(t); });
            body(0);
            for (auto& th : pool) th.join();
        };
        run([&](int t) {
            uint64_t b0, b1; chunk(t, b0, b1);
            int32_t c = 0;
            for (uint64_t b = b0; b < b1; ++b) {
                const int32_t id = s.boxes[b].objectID;
                if (id != -1) { if (id < 0 || (uint64_t)id >= nObj) { err.set("box " + std::to_string(b) + ": objectID out of range"); return; } ++c; }
            }
            leavesIn[(size_t)t + 1] = c;
        });
        if (err.any()) return fail(err.msg);
        for (int t = 0; t < T; ++t) leavesIn[(size_t)t + 1] += leavesIn[(size_t)t];
        nLeaf = leavesIn[(size_t)T]; nInner = (int32_t)(nBox - (uint64_t)nLeaf);
        leafBox.resize((size_t)nLeaf);
        run([&](int t) {
            uint64_t b0, b1; chunk(t, b0, b1);
            int32_t l = leavesIn[(size_t)t], in = (int32_t)b0 - l;
            for (uint64_t b = b0; b < b1; ++b) {
                if (s.boxes[b].objectID != -1) { leafBox[(size_t)l] = (int32_t)b; slotOf[b] = l++; }
                else slotOf[b] = in++;
            }
        });
    }
    // depth + tree check (every box reached at most once) + nesting check (parent contains child) + well-formed
    // boxes + the leaves' tie-break ranks (emit_leaf). The walk starts breadth-first on one thread; once the frontier
    // is wide enough its subtrees are walked depth-first on all threads (a box reached twice — by any two walkers —
    // is an error). The frontier keeps the left-to-right order of the tree, a subtree's walker numbers its leaves
    // locally, and a prefix sum over the frontier turns that into the global left-first rank.
    if (nLeaf >= (1 << 28)) return fail("too many leaves (2^28 or more)");
    // the quantised walk packs (base << 4 | slot mask) into an int and decodes the base with an arithmetic shift (bases are
    // pre-biased and may be negative): bases from 2^27 on would come back negative. Larger scenes walk the float nodes
    // (same tree, same leaf set, no distance culling: RtxInfo.traversal still says WIDE).
    if (nLeaf > rtxbuild::kQuantMaxLeaves) wantQuant = false;
    std::vector<int32_t, rtxbuild::NoInitAlloc<int32_t>> rankLocal((size_t)nLeaf), rankPart((size_t)nLeaf);   // per leaf: right-first index inside its part, part
    parallel_for((uint64_t)nLeaf, nthreads, [&](uint64_t k0, uint64_t k1) { for (uint64_t k = k0; k < k1; ++k) { rankLocal[k] = 0; rankPart[k] = 0; } });
    std::vector<int64_t> partBase;                                                   // per frontier entry: leaves left of it
    std::vector<int32_t> partCount;
    {
        std::vector<uint8_t> seen(nBox, 0);
        std::atomic<int> maxDepthAll{0};
        std::atomic<bool> nestedAll{true};
        auto visit = [&](int32_t b, int depth, std::vector<std::pair<int32_t, int32_t>>& out) -> bool {
            if (__atomic_exchange_n(&seen[(size_t)b], (uint8_t)1, __ATOMIC_RELAXED)) {
                err.set("boxes do not form a tree (box " + std::to_string(b) + " reached twice)");
                return false;
            }
            const RtxBoundingBox& box = s.boxes[b];
            // the fast path picks a box's near and far planes by the sign of the ray direction, which
            // equals the reference's min/max of the two plane distances only when min <= max (NaN fails too)
            if (b != 0 && !(box.minimalPosition.x <= box.maximalPosition.x && box.minimalPosition.y <= box.maximalPosition.y &&
                            box.minimalPosition.z <= box.maximalPosition.z)) nestedAll = false;
            if (box.objectID != -1) return true;
            const int32_t child[2] = {box.leftID, box.rightID};
            for (int k = 0; k < 2; ++k) {
                if (child[k] == -1) continue;   // the reference guards children with != -1 (src/ThreadedShader.cpp:380,387)
                if (child[k] < 0 || (uint64_t)child[k] >= nBox) { err.set("box " + std::to_string(b) + ": child out of range"); return false; }
                const RtxBoundingBox& c = s.boxes[child[k]];
                const bool in = box.minimalPosition.x <= c.minimalPosition.x && box.minimalPosition.y <= c.minimalPosition.y &&
                                box.minimalPosition.z <= c.minimalPosition.z && box.maximalPosition.x >= c.maximalPosition.x &&
                                box.maximalPosition.y >= c.maximalPosition.y && box.maximalPosition.z >= c.maximalPosition.z;
                if (!in && b != 0) nestedAll = false;     // the root's own box is never tested by the reference
                out.emplace_back(child[k], depth + 1);
            }
            return true;
        };
        // depth-first over one part (children are appended left, right and popped from the back: right subtree first)
        auto walk = [&](std::vector<std::pair<int32_t, int32_t>>& st, int32_t part) -> int32_t {
            int deepest = 0;
            int32_t leaves = 0;
            while (!st.empty()) {
                auto [b, depth] = st.back(); st.pop_back();
                deepest = std::max(deepest, depth);
                if (!visit(b, depth, st)) return leaves;
                if (s.boxes[b].objectID != -1) { rankLocal[(size_t)slotOf[b]] = leaves++; rankPart[(size_t)slotOf[b]] = part; }
            }
            int cur = maxDepthAll.load();
            while (deepest > cur && !maxDepthAll.compare_exchange_weak(cur, deepest)) {}
            return leaves;
        };
        // breadth-first phase: inner boxes are replaced by their children in place (left, right), leaves stay put
        std::vector<std::pair<int32_t, int32_t>> frontier{{0, 0}}, next, kids;
        std::vector<uint8_t> expanded{0}, nextExpanded;      // 1: the entry was visited here and is a leaf
        int topDepth = 0;
        for (;;) {
            size_t open = 0;
            for (size_t i = 0; i < frontier.size(); ++i) open += expanded[i] ? 0 : 1;
            if (open == 0 || open >= (size_t)(8 * std::max(1, nthreads)) || err.any()) break;
            next.clear(); nextExpanded.clear();
            for (size_t i = 0; i < frontier.size() && !err.any(); ++i) {
                if (expanded[i]) { next.push_back(frontier[i]); nextExpanded.push_back(1); continue; }
                auto [b, depth] = frontier[i];
                topDepth = std::max(topDepth, depth);
                kids.clear();
                if (!visit(b, depth, kids)) break;
                if (s.boxes[b].objectID != -1) { next.push_back(frontier[i]); nextExpanded.push_back(1); }
                else for (auto& k : kids) { next.push_back(k); nextExpanded.push_back(0); }
            }
            frontier.swap(next); expanded.swap(nextExpanded);
        }
        maxDepthAll = topDepth;
        partCount.assign(frontier.size(), 0);
        for (size_t i = 0; i < frontier.size(); ++i)
            if (expanded[i]) { partCount[i] = 1; rankLocal[(size_t)slotOf[frontier[i].first]] = 0; rankPart[(size_t)slotOf[frontier[i].first]] = (int32_t)i; }
        if (!err.any() && !frontier.empty()) {
            std::atomic<size_t> nextRoot{0};
            std::vector<std::thread> pool;
            auto worker = [&]() {
                std::vector<std::pair<int32_t, int32_t>> st;
                for (size_t i = nextRoot++; i < frontier.size() && !err.any(); i = nextRoot++) {
                    if (expanded[i]) continue;
                    st.assign(1, frontier[i]);
                    partCount[i] = walk(st, (int32_t)i);
                }
            };
            for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
            worker();
            for (auto& th : pool) th.join();
        }
        if (err.any()) return fail(err.msg);
        partBase.assign(frontier.size() + 1, 0);
        for (size_t i = 0; i < frontier.size(); ++i) partBase[i + 1] = partBase[i] + partCount[i];
        f.maxDepth = maxDepthAll;
        f.nested = nestedAll;
        std::atomic<bool> unreachable{false};
        parallel_for(nBox, nthreads, [&](uint64_t b0, uint64_t b1) {
            for (uint64_t b = b0; b < b1; ++b) if (!seen[b] && s.boxes[b].objectID != -1) { unreachable = true; return; }
        });
        if (unreachable) f.nested = false;   // unreachable leaf: keep the reference topology (it is never tested; rank 0)
    }
    f.seconds[0] = clock.lap("validate + walk boxes");
    // own tree over the same leaf boxes (DESIGN.md "Topology freedom")
    rtxbuild::Raw<uint32_t> order((size_t)nLeaf);
    parallel_for((uint64_t)nLeaf, nthreads, [&](uint64_t k0, uint64_t k1) { for (uint64_t k = k0; k < k1; ++k) order[k] = (uint32_t)k; });
    if (wantFast && f.nested && nLeaf >= 2) {
        rtxbuild::Raw<rtxbuild::Box> prim((size_t)nLeaf);
        parallel_for((uint64_t)nLeaf, nthreads, [&](uint64_t k0, uint64_t k1) {
            for (uint64_t k = k0; k < k1; ++k) {
                const RtxBoundingBox& b = s.boxes[leafBox[k]];
                prim[k].lo[0] = b.minimalPosition.x; prim[k].lo[1] = b.minimalPosition.y; prim[k].lo[2] = b.minimalPosition.z;
                prim[k].hi[0] = b.maximalPosition.x; prim[k].hi[1] = b.maximalPosition.y; prim[k].hi[2] = b.maximalPosition.z;
            }
        });
        rtxbuild::Raw<rtxbuild::Node> bin((size_t)nLeaf - 1);
        int binDepth = 0;
        const int32_t binRoot = rtxbuild::build_sah(prim.data(), (uint32_t)nLeaf, order.data(), bin.data(), nthreads, binDepth);
        f.seconds[1] += clock.lap("binned-SAH build");
        if (binDepth > 60) {
            for (int32_t k = 0; k < nLeaf; ++k) order[k] = (uint32_t)k;      // degenerate input: keep the reference topology
        } else {
            f.fast.reserve((size_t)nLeaf / 2);
            rtxbuild::collapse4(bin.data(), bin.size(), binRoot, f.fast, f.fastDepth, nthreads);
            f.fastRoot = 0;
            f.seconds[1] += clock.lap("fold into 4-wide nodes");
        }
    } else if (wantFast && f.nested && nLeaf <= 1 && !(nBox > 0 && s.boxes[0].objectID != -1)) {
        // (a root box that is itself the leaf is never tested by the reference, src/ThreadedShader.cpp:349-368: such a
        // one-object tree stays on the reference topology, where the primitive is intersected unconditionally)
        rtxbuild::Box b;
        if (nLeaf == 1) {
            const RtxBoundingBox& lb = s.boxes[leafBox[0]];
            b.lo[0] = lb.minimalPosition.x; b.lo[1] = lb.minimalPosition.y; b.lo[2] = lb.minimalPosition.z;
            b.hi[0] = lb.maximalPosition.x; b.hi[1] = lb.maximalPosition.y; b.hi[2] = lb.maximalPosition.z;
        }
        rtxbuild::single_leaf_node4(nLeaf == 1 ? &b : nullptr, f.fast);
        f.fastRoot = 0; f.fastDepth = 0;
    }
    if (wantQuant && !f.fast.empty()) {
        // RTX_TRAVERSE_WIDE: byte boxes, children regrouped (inner first, leaves as neighbours): a new leaf order
        std::vector<uint32_t> newToOld;
        if (rtxbuild::quantize4(f.fast, f.qnodes, newToOld, nthreads) && newToOld.size() == (size_t)nLeaf) {
            rtxbuild::Raw<uint32_t> order2((size_t)nLeaf);
            parallel_for((uint64_t)nLeaf, nthreads, [&](uint64_t p0, uint64_t p1) { for (uint64_t p = p0; p < p1; ++p) order2[p] = order[newToOld[p]]; });
            order.swap(order2);
            rtxbuild::Raw<rtxbuild::Node4>().swap(f.fast);
        } else {
            f.qnodes.clear();          // a box extent that is not finite: keep the reference topology
            rtxbuild::Raw<rtxbuild::Node4>().swap(f.fast);
            for (int32_t k = 0; k < nLeaf; ++k) order[k] = (uint32_t)k;
        }
        f.seconds[1] += clock.lap("quantise nodes");
    }
    // leaves in the new order; reference leaf index -> position. The records themselves are written on the device.
    std::vector<int32_t, rtxbuild::NoInitAlloc<int32_t>> posOf((size_t)nLeaf);     // a permutation: every entry is written below
    f.leafPlan.resize((size_t)nLeaf);
    parallel_for((uint64_t)nLeaf, nthreads, [&](uint64_t p0, uint64_t p1) {
        for (uint64_t p = p0; p < p1; ++p) {
            posOf[order[p]] = (int32_t)p;
            const int32_t b = leafBox[order[p]];
            // left-first rank = leaves left of the part + (the part walked right subtree first: count - 1 - index)
            const size_t k = order[p];
            const int32_t part = rankPart[k];
            const int32_t rank = (int32_t)std::max<int64_t>(0, partBase[(size_t)part] + (partCount[(size_t)part] - 1 - rankLocal[k]));
            f.leafPlan[p] = make_int2(b, rank);
        }
    });
    f.rootRef = s.boxes[0].objectID != -1 ? ~posOf[slotOf[0]] : slotOf[0];
    f.slotOf.swap(slotOf); f.posOf.swap(posOf);
    f.numLeaves = (size_t)nLeaf; f.numInner = (size_t)nInner;
    f.deviceEmit = true;
    f.seconds[2] = clock.lap("leaf order and ranks");
    return 0;
}

template <class T>
int upload(T** dst, const void* src, size_t bytes, size_t allocBytes = 0) {
    if (allocBytes < bytes) allocBytes = bytes;
    if (allocBytes == 0) allocBytes = 64;
    HIP_TRY(hipMalloc((void**)dst, allocBytes));
    if (allocBytes > bytes) HIP_TRY(hipMemset((char*)*dst + bytes, 0, allocBytes - bytes));
    if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}

// The caller's objects and boxes on one device, as they are. rtx_create starts the copy on a thread of its own before it
// flattens, so the 3.8 GB (10 M triangles) cross PCIe while the host builds the tree.
struct RawUpload {
    int device = -1;
    float4 *dObjects = nullptr, *dBoxes = nullptr;
    std::thread worker;
    hipError_t err = hipSuccess;
    bool started = false;
    size_t freeAtStart = 0;              // free device memory before the raw arrays were allocated (RtxInfo.deviceBytes)
    static hipError_t copy(int dev, const RtxSceneDesc* sc, float4** dObj, float4** dBox) {
        hipError_t e = hipSetDevice(dev);
        const size_t ob = (size_t)sc->numObjects * sizeof(RtxObject), bb = (size_t)sc->numBoxes * sizeof(RtxBoundingBox);
        if (e == hipSuccess) e = hipMalloc((void**)dObj, std::max<size_t>(ob, 64));
        if (e == hipSuccess) e = hipMalloc((void**)dBox, std::max<size_t>(bb, 64));
        if (e == hipSuccess && ob) e = hipMemcpy(*dObj, sc->objects, ob, hipMemcpyHostToDevice);
        if (e == hipSuccess && bb) e = hipMemcpy(*dBox, sc->boxes, bb, hipMemcpyHostToDevice);
        return e;
    }
    void start(int dev, const RtxSceneDesc* sc) {
        device = dev; started = true;
        worker = std::thread([this, dev, sc]() {
            size_t total = 0;
            if (hipSetDevice(dev) != hipSuccess || hipMemGetInfo(&freeAtStart, &total) != hipSuccess) freeAtStart = 0;
            err = copy(dev, sc, &dObjects, &dBoxes);
        });
    }
    hipError_t wait() { if (worker.joinable()) worker.join(); return err; }
    void release() {
        wait();
        if (device >= 0 && (dObjects || dBoxes)) { hipSetDevice(device); if (dObjects) hipFree(dObjects); if (dBoxes) hipFree(dBoxes); }
        dObjects = dBoxes = nullptr; started = false;
    }
    ~RawUpload() { release(); }
};

// leaf / shade / reference-topology records of `r`, written by its device from the raw arrays (rtx_kernels.h "device-side flatten")
int emit_records_on_device(RtxRenderer* r, const RtxSceneDesc* scene, const Flat& flat, RawUpload* raw) {
    HIP_TRY(hipSetDevice(r->device));
    float4 *dObj = nullptr, *dBox = nullptr;
    bool own = false;
    if (raw && raw->started && raw->device == r->device) {
        HIP_TRY(raw->wait());
        dObj = raw->dObjects; dBox = raw->dBoxes;
    } else {
        const hipError_t e = RawUpload::copy(r->device, scene, &dObj, &dBox);
        own = true;
        if (e != hipSuccess) { if (dObj) hipFree(dObj); if (dBox) hipFree(dBox); return fail(std::string("upload of objects / boxes: ") + hipGetErrorString(e)); }
    }
    int2* dPlan = nullptr; int *dSlotOf = nullptr, *dPosOf = nullptr;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; return e == hipSuccess; };
    const size_t nl = flat.numLeaves, ni = flat.numInner, nb = (size_t)scene->numBoxes;
    ok(hipMalloc((void**)&r->dLeaf, std::max<size_t>(nl * 4 * sizeof(float4), 64)));
    ok(hipMalloc((void**)&r->dShade, std::max<size_t>(nl * 8 * sizeof(float4), 64)));
    ok(hipMalloc((void**)&r->dInner, std::max<size_t>(ni * 4 * sizeof(float4), 64)));
    ok(hipMalloc((void**)&dPlan, std::max<size_t>(nl * sizeof(int2), 64)));
    ok(hipMalloc((void**)&dSlotOf, std::max<size_t>(nb * sizeof(int), 64)));
    ok(hipMalloc((void**)&dPosOf, std::max<size_t>(nl * sizeof(int), 64)));
    if (nl) ok(hipMemcpy(dPlan, flat.leafPlan.data(), nl * sizeof(int2), hipMemcpyHostToDevice));
    if (nb) ok(hipMemcpy(dSlotOf, flat.slotOf.data(), nb * sizeof(int), hipMemcpyHostToDevice));
    if (nl) ok(hipMemcpy(dPosOf, flat.posOf.data(), nl * sizeof(int), hipMemcpyHostToDevice));
    if (e == hipSuccess) {
        if (nl) hipLaunchKernelGGL(k_emit_leaf_records, dim3((uint32_t)((nl + 255) / 256)), dim3(256), 0, 0, dObj, dBox, dPlan, (uint32_t)nl, r->dLeaf, r->dShade);
        if (nb) hipLaunchKernelGGL(k_emit_inner_records, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, 0, dBox, dSlotOf, dPosOf, (uint32_t)nb, r->dInner);
        ok(hipGetLastError());
        ok(hipDeviceSynchronize());
    }
    void* tmp[] = {dPlan, dSlotOf, dPosOf};
    for (void* p : tmp) if (p) hipFree(p);
    if (own) { if (dObj) hipFree(dObj); if (dBox) hipFree(dBox); }
    else raw->release();                  // consumed
    if (e != hipSuccess) return fail(std::string("device-side flatten: ") + hipGetErrorString(e));
    return 0;
}

DevFrame make_frame(const RtxRenderer* r, const RtxCamera* cam, uint32_t fc) {
    DevFrame fr{};
    fr.front[0] = cam->front.x; fr.front[1] = cam->front.y; fr.front[2] = cam->front.z;
    fr.up[0] = cam->up.x; fr.up[1] = cam->up.y; fr.up[2] = cam->up.z;
    fr.right[0] = cam->right.x; fr.right[1] = cam->right.y; fr.right[2] = cam->right.z;
    fr.pos[0] = cam->position.x; fr.pos[1] = cam->position.y; fr.pos[2] = cam->position.z; fr.pos[3] = cam->position.w;
    fr.aspect = cam->aspectRatio; fr.nearView = cam->nearView;
    // src/Camera.h:58 — tan() is the double overload of a float product; a per-frame constant
    fr.tanHalfFOV = (float)tan((double)(0.0174532925f * cam->fov * 0.5f));
    fr.frameCounter = fc;
    fr.lerpScale = 1.0f / (float)(fc + 1u);   // uint32 arithmetic, then float (src/ThreadedShader.cpp:179)
    fr.width = (int)r->width; fr.height = (int)r->height;
    fr.tileW = r->tileW; fr.tileH = r->tileH; fr.tilesX = r->tilesX;
    fr.numLocalTiles = (int)r->localTiles.size();
    fr.numSlots = (int)r->numSlots;
    return fr;
}

void activate(RtxRenderer* r, int i) {
    const RtxRenderer::FrameCtx& c = r->ctx[i];
    r->active = i;
    r->stream = c.stream;
    for (int k = 0; k < 2; ++k) { r->dRqO[k] = c.dRqO[k]; r->dRqD[k] = c.dRqD[k]; }
    r->dHit = c.dHit; r->dLight = c.dLight; r->dAcc = c.dAcc;
    r->dSeeds = c.dSeeds; r->dCtrl = c.dCtrl; r->dSlow = c.dSlow; r->dSpill = c.dSpill;
    r->sc.spill = c.dSpill;
    r->curStats = r->dStats;
}

// A persistent grid costs ~70 ns per block to launch and drain whatever it finds to do (2048 blocks: 0.15 ms),
// which is most of a launch over a short queue (a late bounce, a small image, a 1/8 tile share). The grid is
// therefore sized by the expected queue length: one block per RTX_BLOCK * RTX_RAYS_PER_LANE rays, at least one
// per CU. The expectation comes from the last frame whose queue lengths have arrived (copied back
// asynchronously, never waited for); it only sizes the launch — the kernels distribute work dynamically
// and are correct for any grid.
void set_grids(RtxRenderer* r, uint32_t expected) {
    const uint64_t perTraceBlock = (uint64_t)RTX_BLOCK * RTX_RAYS_PER_LANE, perShadeBlock = (uint64_t)RTX_BLOCK * 16u;
    const uint64_t e = (uint64_t)expected + expected / 4u;
    r->launchTrace = (int)std::max<uint64_t>((uint64_t)std::min(r->numCUs, r->gridTrace), std::min<uint64_t>((uint64_t)r->gridTrace, (e + perTraceBlock - 1) / perTraceBlock));
    r->launchShade = (int)std::max<uint64_t>((uint64_t)std::min(r->numCUs / 2, r->gridShade), std::min<uint64_t>((uint64_t)r->gridShade, (e + perShadeBlock - 1) / perShadeBlock));
}


void finish_timing(RtxRenderer* r, int slot) {
    if (!r->pending[slot]) return;
    if (r->aheadSlot[slot]) {          // started ahead and never asked for: speculative device time is not the caller's
        r->aheadSlot[slot] = false; r->pending[slot] = false;
        return;
    }
    hipEventSynchronize(r->ev[slot][kEvents - 1]);
    float total = 0, trace = 0, shade = 0, ms = 0;
    hipEventElapsedTime(&total, r->ev[slot][0], r->ev[slot][kEvents - 1]);
    // kernel k (0..7) runs between events k and k+1: even = trace (primary, trace), odd = shade
    for (int k = 0; k < 8; ++k) {
        hipEventElapsedTime(&ms, r->ev[slot][k], r->ev[slot][k + 1]);
        r->totals.kernelMs[k] += ms;
        if (k & 1) shade += ms; else trace += ms;
    }
    r->totals.lastFrameMs = total; r->totals.lastTraceMs = trace; r->totals.lastShadeMs = shade;
    // totalMs adds up device time without counting overlap twice: a frame contributes what it ran past the
    // end of the frame before it (frames in flight overlap; with framesInFlight = 1 this is the frame's own time)
    float past = total;
    if (r->lastFinished >= 0 && hipEventElapsedTime(&ms, r->ev[r->lastFinished][kEvents - 1], r->ev[slot][kEvents - 1]) == hipSuccess)
        past = std::max(0.0f, std::min(total, ms));
    r->totals.totalMs += past;
    r->lastFinished = slot;
    r->pending[slot] = false;
}

void dispatch_trace(RtxRenderer* r, const float4* o, const float4* d, int countIdx, int workIdx, uint32_t fixedCount);

template <int MODE>
void launch_primary(RtxRenderer* r, const DevBatch& bt, int workIdx) {
    const size_t lds = (RTX_PRIMARY_LEAFQ && (MODE == RTX_MODE_WIDEQ || MODE == RTX_MODE_WIDEQ_BIG)) ? std::max<size_t>(r->ldsTrace, RTX_TRACEQW_LDS_BYTES) : r->ldsTrace;
    if (r->cfg.countStats)
        hipLaunchKernelGGL((k_primary<MODE, true>), dim3(r->launchTrace), dim3(RTX_BLOCK), lds, r->stream, r->sc, bt, r->dTiles,
                           r->dRqO[0], r->dRqD[0], r->dHit, r->dSeeds, r->dCtrl, r->curStats, workIdx);
    else
        hipLaunchKernelGGL((k_primary<MODE, false>), dim3(r->launchTrace), dim3(RTX_BLOCK), lds, r->stream, r->sc, bt, r->dTiles,
                           r->dRqO[0], r->dRqD[0], r->dHit, r->dSeeds, r->dCtrl, r->curStats, workIdx);
}
template <int MODE>
void launch_trace(RtxRenderer* r, const float4* o, const float4* d, int countIdx, int workIdx, uint32_t fixedCount) {
    if (r->cfg.countStats)
        hipLaunchKernelGGL((k_trace<MODE, true>), dim3(r->launchTrace), dim3(RTX_BLOCK), r->ldsTrace, r->stream, r->sc, o, d,
                           r->dHit, r->dCtrl, r->curStats, fixedCount, countIdx, workIdx);
    else
        hipLaunchKernelGGL((k_trace<MODE, false>), dim3(r->launchTrace), dim3(RTX_BLOCK), r->ldsTrace, r->stream, r->sc, o, d,
                           r->dHit, r->dCtrl, r->curStats, fixedCount, countIdx, workIdx);
}
void dispatch_primary(RtxRenderer* r, const DevBatch& bt, int workIdx) {
    // Bounce 0 is coherent (one 8x8 pixel block per wave): it keeps one ray per lane per batch, fused with
    // ray generation. Measured at 4K / 10 M triangles: fused 2.48 ms, ray generation + k_trace 2.47 ms,
    // ray generation + the lane-refill kernel 3.7 ms (refill mixes rays of different blocks in a wave).
    if (r->mode == RTX_MODE_LINEAR) launch_primary<RTX_MODE_LINEAR>(r, bt, workIdx);
    else if (r->mode == RTX_MODE_EXACT) launch_primary<RTX_MODE_EXACT>(r, bt, workIdx);
    else if (r->mode == RTX_MODE_CULLED_BIG) launch_primary<RTX_MODE_CULLED_BIG>(r, bt, workIdx);
    else if (r->mode == RTX_MODE_WIDEQ) launch_primary<RTX_MODE_WIDEQ>(r, bt, workIdx);
    else if (r->mode == RTX_MODE_WIDEQ_BIG) launch_primary<RTX_MODE_WIDEQ_BIG>(r, bt, workIdx);
    else launch_primary<RTX_MODE_CULLED>(r, bt, workIdx);
}
// fixedCount: queue length when it is known on the host (bounce 0 / probes), else read from ctrl[countIdx]
void dispatch_trace(RtxRenderer* r, const float4* o, const float4* d, int countIdx, int workIdx, uint32_t fixedCount) {
    if (r->mode == RTX_MODE_CULLED || r->mode == RTX_MODE_CULLED_BIG || r->mode == RTX_MODE_WIDEQ || r->mode == RTX_MODE_WIDEQ_BIG) {
        const bool big = r->mode == RTX_MODE_CULLED_BIG || r->mode == RTX_MODE_WIDEQ_BIG;
        const bool quant = r->mode == RTX_MODE_WIDEQ || r->mode == RTX_MODE_WIDEQ_BIG;
        switch (countIdx * 4 + (r->cfg.countStats ? 2 : 0) + (big ? 1 : 0)) {
#define RTX_REFILL_CASE(B, C, G)                                                                                              \
        case (B) * 4 + ((C) ? 2 : 0) + ((G) ? 1 : 0):                                                                          \
            if (quant && r->leafQueue)                                                                                         \
            hipLaunchKernelGGL((k_trace_qw<C, B, G>), dim3((B) >= 2 ? std::min(r->launchTrace, r->numCUs * r->lateBlocksPerCU) : r->launchTrace), dim3(RTX_BLOCK), RTX_TRACEQW_LDS_BYTES, r->stream, r->sc, \
                               o, d, r->dHit, r->dCtrl, r->curStats, r->dSlow, fixedCount, workIdx);               \
            else if (quant)                                                                                                    \
            hipLaunchKernelGGL((k_trace_q<C, B, G>), dim3(r->launchTrace), dim3(RTX_BLOCK), r->ldsTrace, r->stream, r->sc,     \
                               o, d, r->dHit, r->dCtrl, r->curStats, r->dSlow, fixedCount, workIdx);               \
            else                                                                                                               \
            hipLaunchKernelGGL((k_trace_refill<C, B, G>), dim3(r->launchTrace), dim3(RTX_BLOCK), r->ldsTrace, r->stream, r->sc, \
                               o, d, r->dHit, r->dCtrl, r->curStats, r->dSlow, fixedCount, workIdx);               \
            break;
#define RTX_REFILL_BOUNCE(B) RTX_REFILL_CASE(B, false, false) RTX_REFILL_CASE(B, false, true) RTX_REFILL_CASE(B, true, false) RTX_REFILL_CASE(B, true, true)
            RTX_REFILL_BOUNCE(0) RTX_REFILL_BOUNCE(1) RTX_REFILL_BOUNCE(2) RTX_REFILL_BOUNCE(3)
#undef RTX_REFILL_BOUNCE
#undef RTX_REFILL_CASE
        }
        if (r->cfg.countStats)
            hipLaunchKernelGGL((k_trace_slow<true>), dim3(r->numCUs), dim3(RTX_BLOCK), r->ldsSlow, r->stream, r->sc, o, d,
                               r->dHit, r->dCtrl, r->curStats, r->dSlow, countIdx);
        else
            hipLaunchKernelGGL((k_trace_slow<false>), dim3(r->numCUs), dim3(RTX_BLOCK), r->ldsSlow, r->stream, r->sc, o, d,
                               r->dHit, r->dCtrl, r->curStats, r->dSlow, countIdx);
        return;
    }
    if (r->mode == RTX_MODE_LINEAR) launch_trace<RTX_MODE_LINEAR>(r, o, d, countIdx, workIdx, fixedCount);
    else if (r->mode == RTX_MODE_EXACT) launch_trace<RTX_MODE_EXACT>(r, o, d, countIdx, workIdx, fixedCount);
    else launch_trace<RTX_MODE_CULLED>(r, o, d, countIdx, workIdx, fixedCount);
}

int check_camera(const RtxCamera* cam) {
    if (!cam) return fail("camera is NULL");
    return 0;
}

} // namespace

static hipError_t alloc_queues(RtxRenderer* r, int frames);
static int build_renderer(const RtxSceneDesc* scene, const RtxConfig& cfg, const Flat& flat, bool useBVH, PhaseClock& createClock, RawUpload* raw, RtxRenderer** out);

// ---------------------------------------------------------------------------------
extern "C" {

const char* rtx_last_error(void) { return g_err.c_str(); }
const char* rtx_version(void) { return "raytracer_amd HIP backend 0.1 (gfx950)"; }

void rtx_destroy(RtxRenderer* r) {
    if (!r) return;
    for (RtxRenderer* m : r->peers) { m->leader = nullptr; rtx_destroy(m); }
    r->peers.clear();
    hipSetDevice(r->device);
    if (r->arrived) hipEventDestroy(r->arrived);
    if (r->dPackedOwn) hipFree(r->dPackedOwn);
    if (r->dPresent) hipFree(r->dPresent);
    if (r->dPackedAtLeader && r->leaderDevice >= 0) { hipSetDevice(r->leaderDevice); hipFree(r->dPackedAtLeader); hipSetDevice(r->device); }
    if (r->ctl) { hipStreamSynchronize(r->ctl); hipStreamDestroy(r->ctl); r->ctl = nullptr; }      // (an abort writes into the contexts' ctrl[])
    for (auto& c : r->ctx) {
        if (c.stream) hipStreamSynchronize(c.stream);
        if (c.abortDone) hipEventDestroy(c.abortDone);
        void* own[] = {c.dRqO[0], c.dRqO[1], c.dRqD[0], c.dRqD[1], c.dHit, c.dLight, c.dAcc, c.dSeeds, c.dCtrl, c.dSlow, c.dSpill};
        for (void* p : own) if (p) hipFree(p);
        if (c.resolved) hipEventDestroy(c.resolved);
        if (c.countsReady) hipEventDestroy(c.countsReady);
        if (c.hCounts) hipHostFree(c.hCounts);
        if (c.dStatsAhead) hipFree(c.dStatsAhead);
        if (c.stream) hipStreamDestroy(c.stream);
    }
    if (r->hStage) hipHostFree(r->hStage);
    for (hipEvent_t e : r->stageEv) hipEventDestroy(e);
    void* ptrs[] = {r->dFast, r->dQNodes, r->dInner, r->dLeaf, r->dShade, r->dMaterials, r->dTexInfo, r->dTexels, r->dTiles, r->dFb, r->dStats};
    for (void* p : ptrs) if (p) hipFree(p);
    for (int2* p : r->dTilesOfRank) if (p) hipFree(p);
    for (int i = 0; i < kRing; ++i) for (int k = 0; k < kEvents; ++k) if (r->ev[i][k]) hipEventDestroy(r->ev[i][k]);
    delete r;
}

int rtx_create(const RtxSceneDesc* scene, const RtxConfig* cfgIn, RtxRenderer** out) {
    if (!scene || !out) return fail("rtx_create: NULL argument");
    *out = nullptr;
    RtxConfig cfg{};
    if (cfgIn) cfg = *cfgIn;
    if (cfg.worldSize <= 0) { cfg.worldSize = 1; cfg.rank = 0; }
    if (cfg.rank < 0 || cfg.rank >= cfg.worldSize) return fail("rank outside [0, worldSize)");
    if (cfg.tileW == 0) cfg.tileW = 64;
    if (cfg.tileH == 0) cfg.tileH = 32;
    if (cfg.tileW < 8 || cfg.tileH < 8 || (cfg.tileW & 7) || (cfg.tileH & 7) || cfg.tileW > 1024 || cfg.tileH > 1024)
        return fail("tileW/tileH must be multiples of 8 in [8,1024]");
    if (scene->width == 0 || scene->height == 0 || scene->width > 32768 || scene->height > 32768) return fail("bad resolution");
    if (scene->numObjects > 0 && !scene->objects) return fail("objects is NULL");
    if (scene->numObjects > 0x7fffffffull) return fail("too many objects");
    if (scene->numMaterials == 0 || !scene->materials) return fail("at least one material is required");
    if (scene->numTextures == 0 || !scene->textureInfo || !scene->textureData) return fail("at least one texture is required");
    for (uint64_t i = 0; i < scene->numMaterials; ++i) {
        const int32_t t = scene->materials[i].textureID;
        if (t < 0 || (uint64_t)t >= scene->numTextures) return fail("material " + std::to_string(i) + ": textureID out of range");
    }
    for (uint64_t i = 0; i < scene->numTextures; ++i) {
        const RtxTexture& t = scene->textureInfo[i];
        if (t.width <= 0 || t.height <= 0 || t.width > 4096 || t.height > 65536 || t.offset < 0 ||
            (uint64_t)t.offset + (uint64_t)t.width * (uint64_t)t.height > scene->numTexels)
            return fail("texture " + std::to_string(i) + ": extent outside textureData (or wider than 4096)");
    }
    if (cfg.numDevices < 0) return fail("RtxConfig.numDevices must not be negative");
    if (cfg.numDevices > 1) {
        if (cfg.numDevices > 8) return fail("RtxConfig.numDevices: at most 8");
        if (cfg.worldSize != 1) return fail("a device group (numDevices > 1) cannot also be one rank of a process group (worldSize > 1)");
        int count = 0;
        HIP_TRY(hipGetDeviceCount(&count));
        for (int i = 0; i < cfg.numDevices; ++i)
            if (cfg.devices[i] < 0 || cfg.devices[i] >= count) return fail("RtxConfig.devices[" + std::to_string(i) + "]: no such device");
    }
    // src/ThreadedShader.cpp:15-19
    const bool useBVH = scene->useBVH && scene->numBoxes > 0;
    if (useBVH && !scene->boxes) return fail("boxes is NULL");

    Flat flat;
    if (cfg.traversal != RTX_TRAVERSE_CULLED && cfg.traversal != RTX_TRAVERSE_EXACT && cfg.traversal != RTX_TRAVERSE_WIDE)
        return fail("RtxConfig.traversal: unknown mode");
    const bool wantFast = cfg.traversal != RTX_TRAVERSE_EXACT;
    const bool wantQuant = cfg.traversal == RTX_TRAVERSE_WIDE && !getenv("RTX_WIDE_ORDERED");
    PhaseClock createClock;
    // every device of this renderer is first touched here, by the caller's thread
    if (cfg.numDevices > 1) { for (int i = 0; i < cfg.numDevices; ++i) if (ensure_device_ready(cfg.devices[i])) return 1; }
    else if (ensure_device_ready(cfg.device)) return 1;
    // the raw arrays start for the (first) device now; the records are written there once the leaf order is known
    RawUpload raw;
    if (useBVH && !getenv("RTX_UPLOAD_LATE")) raw.start(cfg.numDevices > 1 ? cfg.devices[0] : cfg.device, scene);
    int buildThreads = 16;
    if (const char* e = getenv("RTX_BUILD_THREADS")) buildThreads = std::max(1, std::min(64, atoi(e)));      // experiments
    if (flatten(*scene, useBVH, wantFast, wantQuant, buildThreads, flat)) return 1;
    createClock.lap("flatten");
    if (useBVH && flat.maxDepth > 62) return fail("BVH deeper than 62 levels is not supported (the reference's own stack holds 32)");

    if (cfg.numDevices <= 1) return build_renderer(scene, cfg, flat, useBVH, createClock, &raw, out);

    // Device group (single process, several GPUs): member i renders the tiles of rank i of numDevices on devices[i],
    // all from the one flattened scene; member 0 is the handle the caller gets, and delivering a frame gathers the
    // other members' tiles into its framebuffer (gather_group).
    // The members are built side by side, one host thread per member (uploads, record emission and allocations of
    // different devices do not wait for each other); only member 0 consumes the raw arrays that started crossing PCIe
    // before the flatten. Links first: the gather is one peer copy per member into devices[0].
    for (int i = 1; i < cfg.numDevices; ++i) {
        if (cfg.devices[i] == cfg.devices[0]) continue;                // rehearsal: several members on one GPU
        int can = 0;
        HIP_TRY(hipDeviceCanAccessPeer(&can, cfg.devices[0], cfg.devices[i]));
        if (!can && !getenv("RTX_GROUP_STAGED"))
            return fail("device group: device " + std::to_string(cfg.devices[0]) + " cannot access device " + std::to_string(cfg.devices[i]) +
                        " as a peer (no xGMI / PCIe P2P path: check `rocm-smi --showtopo`, IOMMU / ACS settings and HSA_ENABLE_IPC_MODE_LEGACY=0); "
                        "set RTX_GROUP_STAGED=1 to gather through host memory anyway, or run one process per GPU (RtxConfig.rank / worldSize)");
        if (can) {
            HIP_TRY(hipSetDevice(cfg.devices[0]));
            const hipError_t pe = hipDeviceEnablePeerAccess(cfg.devices[i], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return fail(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe));
            (void)hipGetLastError();
        }
    }
    std::vector<RtxRenderer*> members((size_t)cfg.numDevices, nullptr);
    std::vector<std::string> errors((size_t)cfg.numDevices);
    {
        std::vector<std::thread> pool;
        auto buildOne = [&](int i) {
            RtxConfig c = cfg;
            c.device = cfg.devices[i]; c.rank = i; c.worldSize = cfg.numDevices; c.numDevices = -1;      // -1: a member of a device group
            PhaseClock clock = createClock;
            if (build_renderer(scene, c, flat, useBVH, clock, i == 0 ? &raw : nullptr, &members[(size_t)i])) { errors[(size_t)i] = g_err; members[(size_t)i] = nullptr; }
        };
        const bool serial = getenv("RTX_GROUP_SERIAL_CREATE") != nullptr;
        for (int i = 1; i < cfg.numDevices; ++i) { if (serial) buildOne(i); else pool.emplace_back(buildOne, i); }
        buildOne(0);
        for (auto& th : pool) th.join();
    }
    for (int i = 0; i < cfg.numDevices; ++i)
        if (!members[(size_t)i]) {
            const std::string msg = "device group member " + std::to_string(i) + " (device " + std::to_string(cfg.devices[i]) + "): " + errors[(size_t)i];
            for (RtxRenderer* x : members) if (x) rtx_destroy(x);
            return fail(msg);
        }
    RtxRenderer* lead = members[0];
    lead->peers.assign(members.begin() + 1, members.end());
    lead->info.devices = cfg.numDevices;
    for (RtxRenderer* m : lead->peers) {
        m->leader = lead; m->leaderDevice = lead->device;
        const size_t bytes = (size_t)rtx_packed_floats(m, m->cfg.rank) * sizeof(float);
        hipError_t e = hipSetDevice(m->device);
        if (e == hipSuccess) e = hipMalloc((void**)&m->dPackedOwn, std::max<size_t>(bytes, 64));
        if (e == hipSuccess) e = hipSetDevice(lead->device);
        if (e == hipSuccess) e = hipMalloc((void**)&m->dPackedAtLeader, std::max<size_t>(bytes, 64));
        if (e != hipSuccess) { fail(std::string("device group buffers: ") + hipGetErrorString(e)); rtx_destroy(lead); return 1; }
    }
    *out = lead;
    return 0;
}

static int build_renderer(const RtxSceneDesc* scene, const RtxConfig& cfg, const Flat& flat, bool useBVH, PhaseClock& createClock, RawUpload* raw, RtxRenderer** out) {
    RtxRenderer* r = new RtxRenderer();
    memset(r->ev, 0, sizeof(r->ev));
    r->cfg = cfg; r->device = cfg.device;
    r->width = scene->width; r->height = scene->height;
    r->linear = !useBVH;
    // distance culling needs boxes that contain their descendants; otherwise keep the reference's visit set
    r->mode = r->linear ? RTX_MODE_LINEAR : ((cfg.traversal == RTX_TRAVERSE_EXACT || !flat.nested || (flat.fast.empty() && flat.qnodes.empty())) ? RTX_MODE_EXACT
                                             : (flat.qnodes.empty() ? RTX_MODE_CULLED : RTX_MODE_WIDEQ));
    const size_t bigNodes = (size_t)1 << 25;      // 128-byte nodes / 64-byte nodes: 4 GiB of either
    if (r->mode == RTX_MODE_CULLED && (flat.fast.size() > bigNodes || getenv("RTX_FORCE_BIG"))) r->mode = RTX_MODE_CULLED_BIG;
    if (r->mode == RTX_MODE_WIDEQ && (flat.qnodes.size() * sizeof(rtxbuild::NodeQ) > ((size_t)4 << 30) || getenv("RTX_FORCE_BIG"))) r->mode = RTX_MODE_WIDEQ_BIG;
    const bool quant = r->mode == RTX_MODE_WIDEQ || r->mode == RTX_MODE_WIDEQ_BIG;
    const bool culled = r->mode == RTX_MODE_CULLED || r->mode == RTX_MODE_CULLED_BIG || quant;      // any walk of the own tree
    r->tileW = cfg.tileW; r->tileH = cfg.tileH;

#define CREATE_TRY(expr) do { if (expr) { rtx_destroy(r); return 1; } } while (0)
#define CREATE_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); fail(std::string(#expr) + ": " + hipGetErrorString(e_)); rtx_destroy(r); return 1; } } while (0)

    CREATE_HIP(hipSetDevice(r->device));
    hipDeviceProp_t prop;
    CREATE_HIP(hipGetDeviceProperties(&prop, r->device));
    r->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    CREATE_TRY(ensure_device_ready(r->device));      // (rtx_create has been here already, on its own thread; a member of a group is built on a helper thread)
    size_t freeBefore = 0, freeAfter = 0, totalMem = 0;
    CREATE_HIP(hipMemGetInfo(&freeBefore, &totalMem));
    if (raw && raw->started && raw->device == r->device) {      // the raw arrays are on this device already (and go away again below)
        raw->wait();
        if (raw->freeAtStart) freeBefore = raw->freeAtStart;
    }
    for (int i = 0; i < kRing; ++i) for (int k = 0; k < kEvents; ++k) CREATE_HIP(hipEventCreate(&r->ev[i][k]));

    // tiles: tile t (row-major over the tile grid) belongs to rank t % worldSize
    r->tilesX = (int)((r->width + r->tileW - 1) / r->tileW);
    r->tilesY = (int)((r->height + r->tileH - 1) / r->tileH);
    r->tilesOfRank.assign(cfg.worldSize, {});
    for (int t = 0; t < r->tilesX * r->tilesY; ++t) {
        int2 org = make_int2((t % r->tilesX) * r->tileW, (t / r->tilesX) * r->tileH);
        r->tilesOfRank[t % cfg.worldSize].push_back(org);
    }
    r->localTiles = r->tilesOfRank[cfg.rank];
    r->localPixels = 0;
    for (const int2& t : r->localTiles) {
        int w = std::min(r->tileW, (int)r->width - t.x), h = std::min(r->tileH, (int)r->height - t.y);
        r->localPixels += (uint64_t)w * h;
    }
    const uint64_t slots = (uint64_t)r->localTiles.size() * r->tileW * r->tileH;
    if (slots > 0x7fffff00ull) { fail("too many pixels for one rank"); rtx_destroy(r); return 1; }
    r->numSlots = (uint32_t)slots;
    r->dTilesOfRank.assign(cfg.worldSize, nullptr);
    for (int k = 0; k < cfg.worldSize; ++k)
        CREATE_TRY(upload(&r->dTilesOfRank[k], r->tilesOfRank[k].data(), r->tilesOfRank[k].size() * sizeof(int2)));
    CREATE_TRY(upload(&r->dTiles, r->localTiles.data(), r->localTiles.size() * sizeof(int2)));

    // scene
    if (flat.deviceEmit) CREATE_TRY(emit_records_on_device(r, scene, flat, raw));
    else CREATE_TRY(upload(&r->dInner, flat.inner.data(), flat.inner.size() * sizeof(float4)));
    if (!flat.fast.empty()) CREATE_TRY(upload(&r->dFast, flat.fast.data(), flat.fast.size() * sizeof(rtxbuild::Node4)));
    if (!flat.qnodes.empty()) CREATE_TRY(upload(&r->dQNodes, flat.qnodes.data(), flat.qnodes.size() * sizeof(rtxbuild::NodeQ)));
    if (!flat.deviceEmit) {
        CREATE_TRY(upload(&r->dLeaf, flat.leaf.data(), flat.leaf.size() * sizeof(float4)));
        CREATE_TRY(upload(&r->dShade, flat.shade.data(), flat.shade.size() * sizeof(float4)));
    }
    CREATE_TRY(upload(&r->dMaterials, scene->materials, scene->numMaterials * sizeof(RtxMaterial)));
    CREATE_TRY(upload(&r->dTexInfo, scene->textureInfo, scene->numTextures * sizeof(RtxTexture)));
    CREATE_TRY(upload(&r->dTexels, scene->textureData, scene->numTexels * sizeof(uint32_t),
                      (scene->numTexels + RTX_TEX_PAD) * sizeof(uint32_t)));
    r->sc.inner = r->dInner; r->sc.leaf = r->dLeaf; r->sc.shade = r->dShade; r->sc.materials = r->dMaterials;
    r->sc.texInfo = r->dTexInfo; r->sc.texels = r->dTexels;
    r->sc.numTexelsPadded = (long long)scene->numTexels + RTX_TEX_PAD;
    r->sc.rootRef = flat.rootRef;
    r->sc.fast = r->dFast;            // 4-wide nodes, 8 float4 each (NULL unless the mode is CULLED / CULLED_BIG)
    r->sc.fastRoot = flat.fastRoot;
    r->sc.qnodes = r->dQNodes;
    r->sc.cullSlack = (flat.sliverFraction > 1e-3 || cfg.traversal == RTX_TRAVERSE_WIDE) ? INFINITY : RTX_CULL_SLACK_DEFAULT;
    if (const char* e = getenv("RTX_CULL_SLACK")) if (cfg.traversal == RTX_TRAVERSE_CULLED) r->sc.cullSlack = (float)atof(e);      // experiments
    r->sc.numLeaves = (int)flat.numLeaves;
    r->sc.numInner = (int)flat.numInner;
    r->sc.numMaterials = (int)scene->numMaterials;
    r->sc.numTextures = (int)scene->numTextures;
    // textureInfo[1] is hard-wired as the sky (src/ThreadedShader.cpp:162); with a
    // single texture the reference reads past the vector, here texture 0 is used
    r->sc.skyTex = scene->numTextures > 1 ? 1 : 0;
    r->sc.useBVH = useBVH ? 1 : 0;
    // reference topology: one pending entry per level; 4-wide tree: up to three per level
    r->sc.stackDepth = std::max(2, std::max(flat.maxDepth, (flat.fast.empty() && flat.qnodes.empty()) ? 0 : 3 * (flat.fastDepth + 1)) + 2);

    // persistent shade grid: as many blocks as are resident at once
    int shadeBlocksPerCU = 0;
    CREATE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&shadeBlocksPerCU, k_shade, RTX_BLOCK, RTX_SHADE_LDS_BYTES));
    if (const char* e = getenv("RTX_SHADE_BLOCKS_PER_CU")) shadeBlocksPerCU = std::max(1, std::min(atoi(e), shadeBlocksPerCU));      // experiments
    r->gridShade = r->numCUs * std::max(1, std::min(shadeBlocksPerCU, 8));

    // per-path state (indexed by slot) and queues (indexed by queue position; the bounce queues hold
    // up to one unfinished run of holes per shade wave on top of the live rays)
    // Frames per batch: the thin ends of the launches are shared by the frames of a batch, and the longer the queues the less they
    // weigh (4K / 10 M triangles, queued frames: 6.58 ms per frame one at a time with 14 in flight, 6.27 / 6.06 / 5.95 ms in batches
    // of 2 / 4 / 8 with round 2's kernels; 5.36 / 5.17 / 5.12 / 5.05 ms in batches of 8 / 16 / 24 / 32 with round 3's). The default is
    // as many as fit a budget — 35 % of the device memory that is free now, for the batches in flight (three by default) — up to RTX_MAX_BATCH:
    // 32 frames at 4K (87 GB of the 288), whatever the rank's share of the tiles.
    {
        const uint64_t bytesPerSlot = 4 * sizeof(float4) + sizeof(int2) + 2 * sizeof(float4) + sizeof(uint32_t) + (culled ? sizeof(uint32_t) : 0);
        size_t freeNow = 0, totalNow = 0;
        CREATE_HIP(hipMemGetInfo(&freeNow, &totalNow));
        const uint64_t fit = (uint64_t)((double)freeNow * 0.35) / std::max<uint64_t>(1, (uint64_t)(cfg.framesInFlight > 0 ? cfg.framesInFlight : 3) * std::max<uint64_t>(r->numSlots, 64) * bytesPerSlot);
        uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>(RTX_MAX_BATCH, fit));
        if (cfg.framesPerBatch > 0) want = (uint64_t)cfg.framesPerBatch;
        if (const char* e = getenv("RTX_BATCH")) want = (uint64_t)std::max(1, atoi(e));
        while (want > 1 && (uint64_t)r->numSlots * want > 0x7fffff00ull) --want;
        r->batch = (int)std::max<uint64_t>(1, std::min<uint64_t>(RTX_MAX_BATCH, want));
    }
    CREATE_HIP(hipMalloc((void**)&r->dStats, RTX_STAT_WORDS * sizeof(unsigned long long)));
    CREATE_HIP(hipMemset(r->dStats, 0, RTX_STAT_WORDS * sizeof(unsigned long long)));
    const size_t fbBytes = (size_t)r->width * r->height * sizeof(float4);
    CREATE_HIP(hipMalloc((void**)&r->dFb, fbBytes));
    CREATE_HIP(hipMemset(r->dFb, 0, fbBytes));      // zero-filled like src/WindowManager.cpp:26-27

    // launch geometry: persistent grids sized by the LDS the traversal stack needs
    // culled path: a fixed window of the stack in LDS; exact path: the whole stack (refs only)
    r->ldsTrace = r->linear ? 0 : (quant ? (size_t)RTX_QWORDS : culled ? (size_t)2 * RTX_LDS_WINDOW : (size_t)r->sc.stackDepth) * RTX_BLOCK * sizeof(int);

    r->leafQueue = quant && !(getenv("RTX_LEAFQ") && atoi(getenv("RTX_LEAFQ")) == 0);      // RTX_LEAFQ=0: the per-lane leaf section of k_trace_q (A/B)
    int blocksPerCU = 8;
    if (r->ldsTrace) blocksPerCU = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(r->ldsTrace, r->leafQueue ? (size_t)RTX_TRACEQW_LDS_BYTES : 0)));
    if (const char* e = getenv("RTX_TRACE_BLOCKS_PER_CU")) blocksPerCU = std::max(1, std::min(atoi(e), blocksPerCU));      // experiments: room for other streams' kernels
    r->gridTrace = r->numCUs * blocksPerCU;
    // bounces 2 and 3 start from scattered hit points: their launches are bound by cache misses (L2 hit rate 36 % against 49 % for
    // bounce 1), and fewer rays in flight per CU miss less — 4 blocks per CU instead of 6: 1.30 -> 1.19 and 0.70 -> 0.64 ms (3: the same,
    // 2: 1.52 / 0.72); the bounce-1 launch and k_primary lose with fewer (2.40 -> 2.56 ms at 4)
    r->lateBlocksPerCU = std::min(4, blocksPerCU);
    if (const char* e = getenv("RTX_TRACE_BPC_LATE")) r->lateBlocksPerCU = std::max(1, std::min(atoi(e), blocksPerCU));
    if (culled) r->ldsSlow = (size_t)std::max(2, flat.maxDepth + 2) * RTX_BLOCK * sizeof(int);
    // Streams beyond the hardware queues of the process share a queue and serialise (measured: 4 frames on the
    // default 4 queues are slower than 3), so the default depth follows GPU_MAX_HW_QUEUES, which HIP reads when
    // the process initialises it: 3 frames on 4 queues, 6 on 8, 14 on 16 (deeper pipelines matter most when the
    // queues are short — a rank's share of the tiles on a multi-GPU run: 1/8 of a 4K frame takes 1.33 ms per
    // frame with 6 in flight and 1.17 ms with 14).
    int hwQueues = 4;
    if (const char* e = getenv("GPU_MAX_HW_QUEUES")) hwQueues = std::max(1, atoi(e));
    int depth = cfg.framesInFlight > 0 ? cfg.framesInFlight : (hwQueues <= 4 ? std::max(1, hwQueues - 1) : std::min(hwQueues - 2, 14));
    // three batches in flight, more only when the batches are small (a caller's framesPerBatch): about 24 whole frames' worth of
    // queues at most, whatever the share of the tiles
    if (cfg.framesInFlight <= 0 && r->batch > 1) depth = std::min(depth, std::max(3, 24 * cfg.worldSize / r->batch));
    if (const char* e = getenv("RTX_FRAMES_IN_FLIGHT")) depth = atoi(e);
    depth = std::max(1, std::min(depth, kLag));
    r->ctx.resize((size_t)depth);
    for (auto& c : r->ctx) {
        CREATE_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        CREATE_HIP(hipEventCreateWithFlags(&c.resolved, hipEventDisableTiming));
        CREATE_HIP(hipEventCreateWithFlags(&c.countsReady, hipEventDisableTiming));
        CREATE_HIP(hipEventCreateWithFlags(&c.abortDone, hipEventDisableTiming));
        CREATE_HIP(hipHostMalloc((void**)&c.hCounts, 8 * sizeof(uint32_t), hipHostMallocDefault));
        CREATE_HIP(hipMalloc((void**)&c.dCtrl, RTX_CTRL_WORDS * sizeof(uint32_t)));
        CREATE_HIP(hipMalloc((void**)&c.dStatsAhead, (size_t)RTX_AHEAD_MAX * RTX_STAT_WORDS * sizeof(unsigned long long)));
        if (culled) CREATE_HIP(hipMalloc((void**)&c.dSpill, (size_t)r->sc.stackDepth * r->gridTrace * RTX_BLOCK * sizeof(int2)));
    }
    r->needSlow = culled;
    // queues for the sets of the synchronous loop; the first queued frame grows them to whole batches (grow_queues)
    CREATE_HIP(alloc_queues(r, getenv("RTX_QUEUES_AT_CREATE") ? r->batch : std::min(r->batch, cfg.countStats ? 1 : 4)));
    r->lookAhead = cfg.lookAhead < 0 ? 0 : std::min(cfg.lookAhead > 0 ? cfg.lookAhead : 3, depth - 1);       // 2 with HIP's default 4 hardware queues (3 contexts)
    if (const char* e = getenv("RTX_LOOKAHEAD")) r->lookAhead = std::max(0, std::min(atoi(e), depth - 1));
    // sets of up to 8 frames (128 delivered 4K frames: 5.84 ms per frame with sets of up to 4, 5.44 - 5.49 ms with 8; round 2's kernels: 7.44 / 6.81 /
    // 6.72 / 6.65 ms with 1 / 2 / 4 / 8, which is why the cap used to be 4)
    r->aheadBatch = cfg.countStats ? 1 : std::min(r->batch, RTX_AHEAD_MAX);
    if (const char* e = getenv("RTX_AHEAD_BATCH")) r->aheadBatch = std::max(1, std::min(atoi(e), std::min(r->batch, RTX_AHEAD_MAX)));      // experiments
    if (cfg.worldSize != 1 && cfg.numDevices != -1) r->lookAhead = 0;       // a rank of a process group delivers through its own gather, once per batch (a device group looks ahead on all members)
    if (r->lookAhead > 0 && !getenv("RTX_NO_ABORT")) {
        // Streams of the highest priority draw their hardware queue from a pool of their own, so the control stream does not
        // share a queue with a context that is busy with the very launches it is there to cut short.
        int least = 0, greatest = 0;
        CREATE_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        CREATE_HIP(hipStreamCreateWithPriority(&r->ctl, hipStreamNonBlocking, greatest));
    }
    activate(r, 0);
    for (int b = 0; b < 4; ++b) r->queueHint[b] = r->numSlots * (uint32_t)r->batch;
    set_grids(r, r->numSlots * (uint32_t)r->batch);
    CREATE_HIP(hipDeviceSynchronize());
    {
        RtxInfo& in = r->info;
        in.traversal = r->linear ? RTX_TRAVERSE_EXACT : (culled ? (std::isinf(r->sc.cullSlack) ? RTX_TRAVERSE_WIDE : RTX_TRAVERSE_CULLED) : RTX_TRAVERSE_EXACT);
        in.linear = r->linear ? 1 : 0;
        in.devices = 1;
        in.lookAhead = r->lookAhead;
        in.bigNodes = (r->mode == RTX_MODE_CULLED_BIG || r->mode == RTX_MODE_WIDEQ_BIG) ? 1 : 0;
        in.cullSlack = culled ? r->sc.cullSlack : INFINITY;
        in.framesInFlight = depth; in.framesPerBatch = r->batch; in.hwQueues = hwQueues;
        in.refTreeDepth = flat.maxDepth; in.wideTreeDepth = (flat.fast.empty() && flat.qnodes.empty()) ? 0 : flat.fastDepth;
        in.stackEntries = r->sc.stackDepth;
        in.leaves = (uint64_t)r->sc.numLeaves; in.innerNodes = (uint64_t)r->sc.numInner; in.wideNodes = flat.fast.size() + flat.qnodes.size();
        in.sliverFraction = flat.sliverFraction;
        if (hipMemGetInfo(&freeAfter, &totalMem) == hipSuccess && freeBefore > freeAfter) r->deviceBytes = freeBefore - freeAfter;
        in.deviceBytes = r->deviceBytes;
        for (int k = 0; k < 3; ++k) in.createSeconds[k] = flat.seconds[k];
        in.createSeconds[3] = createClock.lap("upload + allocation");
    }
    if (getenv("RTX_VERBOSE"))
        fprintf(stderr, "[rtx] culling slack %g (%.3f %% slivers), %d leaves, reference tree depth %d, own tree %s depth %d, stack %d entries, LDS %zu B/block, %d blocks/CU x %d CUs, mode %d\n",
                (double)r->sc.cullSlack, 100.0 * flat.sliverFraction, r->sc.numLeaves, flat.maxDepth, flat.fast.empty() ? "absent" : "built", flat.fastDepth, r->sc.stackDepth, r->ldsTrace,
                blocksPerCU, r->numCUs, r->mode),
        fprintf(stderr, "[rtx] %d frames per batch, %d batches in flight (%d hardware queues), shade grid %d blocks (%d per CU), queues allocated for %d frames of %u slots, %zu wide nodes (%.2f per leaf)\n",
                r->batch, depth, hwQueues, r->gridShade, shadeBlocksPerCU, r->capFrames, r->numSlots, flat.fast.size(), r->sc.numLeaves ? (double)flat.fast.size() / r->sc.numLeaves : 0.0);
    *out = r;
    return 0;
}

// Ray queues and per-path state of the contexts, for batches of up to `frames` frames. rtx_create allocates them for the
// first sets the synchronous loop starts ahead (min(batch, 4) frames); a context grows on its own when a larger set lands
// on it (grow_ctx: sets of 8), and the first frame that is QUEUED (rtx_render without pixels) grows all of them to the full
// batch — a caller that never queues frames (the reference's own loop) never holds the 29 GB per context that a batch
// of 32 frames needs at 4K.
static void free_ctx(RtxRenderer::FrameCtx& c) {
    void* own[] = {c.dRqO[0], c.dRqO[1], c.dRqD[0], c.dRqD[1], c.dHit, c.dLight, c.dAcc, c.dSeeds, c.dSlow};
    for (void* p : own) if (p) hipFree(p);
    c.dRqO[0] = c.dRqO[1] = c.dRqD[0] = c.dRqD[1] = nullptr; c.dHit = nullptr; c.dLight = c.dAcc = nullptr; c.dSeeds = c.dSlow = nullptr;
    c.capFrames = 0; c.queueBytes = 0;
}
static hipError_t alloc_ctx(RtxRenderer* r, RtxRenderer::FrameCtx& c, int frames) {
    const size_t n = std::max<size_t>((size_t)r->numSlots * (size_t)frames, 64);
    const size_t nq = n + (size_t)r->gridShade * (RTX_BLOCK / 64) * RTX_MAX_RUN;
    hipError_t e = hipSuccess;
    uint64_t asked = 0;
    auto get = [&](void** p, size_t bytes) { if (e == hipSuccess) { e = hipMalloc(p, bytes); asked += bytes; } };
    for (int k = 0; k < 2; ++k) { get((void**)&c.dRqO[k], nq * sizeof(float4)); get((void**)&c.dRqD[k], nq * sizeof(float4)); }
    get((void**)&c.dHit, nq * sizeof(int2));
    get((void**)&c.dLight, n * sizeof(float4));
    get((void**)&c.dAcc, n * sizeof(float4));
    get((void**)&c.dSeeds, n * sizeof(uint32_t));
    if (r->needSlow) get((void**)&c.dSlow, nq * sizeof(uint32_t));
    if (e != hipSuccess) { (void)hipGetLastError(); free_ctx(c); return e; }
    c.capFrames = frames; c.queueBytes = asked;
    return hipSuccess;
}
static void free_queues(RtxRenderer* r) {
    for (auto& c : r->ctx) free_ctx(c);
    r->capFrames = 0;
}
static hipError_t alloc_queues(RtxRenderer* r, int frames) {
    for (auto& c : r->ctx) {
        const hipError_t e = alloc_ctx(r, c, frames);
        if (e != hipSuccess) { free_queues(r); return e; }
    }
    r->capFrames = frames;
    return hipSuccess;
}
// RtxInfo.deviceBytes follows the queues (by what was asked for: differences of the free memory the runtime reports say little
// once its allocator has blocks of its own to hand out again)
static uint64_t queue_bytes(const RtxRenderer* r) { uint64_t b = 0; for (const auto& c : r->ctx) b += c.queueBytes; return b; }
static void account_growth(RtxRenderer* r, uint64_t queueBytesBefore) {
    r->deviceBytes = r->deviceBytes - std::min(r->deviceBytes, queueBytesBefore) + queue_bytes(r);
    r->info.deviceBytes = r->deviceBytes;
}
// room for full batches in every context (r->batch frames; fewer if the device cannot hold them any more). Everything in
// flight finishes first.
static int grow_queues(RtxRenderer* r) {
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    const uint64_t bytesBefore = queue_bytes(r);
    const int had = r->capFrames;
    free_queues(r);
    hipError_t e = hipSuccess;
    for (int b = r->batch; ; b = std::max(had, b / 2)) {
        e = alloc_queues(r, b);
        if (e == hipSuccess) { r->batch = b; break; }
        if (b <= std::max(1, had)) break;
    }
    if (e != hipSuccess) return fail(std::string("device memory for the frame queues: ") + hipGetErrorString(e));
    account_growth(r, bytesBefore);
    r->info.framesPerBatch = r->batch;
    activate(r, r->active);
    for (int b = 0; b < 4; ++b) r->queueHint[b] = r->numSlots * (uint32_t)r->batch;
    return 0;
}
// room for `frames` frames in context i alone, which holds nothing that is still needed (a set started ahead lands on the
// context whose previous set has been asked for to the last frame); the other contexts go on
static int grow_ctx(RtxRenderer* r, int i, int frames) {
    HIP_TRY(hipSetDevice(r->device));
    RtxRenderer::FrameCtx& c = r->ctx[(size_t)i];
    for (const RtxRenderer::Ahead& a : r->ahead) if (a.ctx == i) return fail("internal: a context with frames started ahead cannot grow");
    HIP_TRY(hipStreamSynchronize(c.stream));
    const uint64_t bytesBefore = queue_bytes(r);
    const int had = c.capFrames;
    free_ctx(c);
    hipError_t e = alloc_ctx(r, c, frames);
    if (e != hipSuccess) e = alloc_ctx(r, c, had);          // what it had, at least
    if (e != hipSuccess) return fail(std::string("device memory for the frame queues: ") + hipGetErrorString(e));
    account_growth(r, bytesBefore);
    return 0;
}

// the running mean takes the frames in order: a batch's resolve waits for the resolve queued before it
// [f0, f1): the frames of the batch that enter the running mean now
static int queue_resolve(RtxRenderer* r, const DevBatch& bt, int f0, int f1) {
    if (r->lastResolved >= 0 && r->lastResolved != r->active)
        HIP_TRY(hipStreamWaitEvent(r->stream, r->ctx[r->lastResolved].resolved, 0));
    if (r->numSlots)
        hipLaunchKernelGGL(k_resolve, dim3((r->numSlots + RTX_BLOCK - 1) / RTX_BLOCK), dim3(RTX_BLOCK), 0, r->stream, bt, r->dTiles,
                           r->dAcc, r->dFb, f0, f1);
    HIP_TRY(hipEventRecord(r->ctx[r->active].resolved, r->stream));
    r->lastResolved = r->active;
    return 0;
}

// One set of launches for `count` frames (1 <= count <= r->batch), in call order.
// ahead: frames started before they are asked for (RtxRenderer::Ahead) — everything but the resolve, hits counted per frame.
static int render_batch(RtxRenderer* r, const std::pair<RtxCamera, uint32_t>* frames, int count, bool primaryOnly, bool ahead = false) {
    HIP_TRY(hipSetDevice(r->device));
    if (ahead && count > RTX_AHEAD_MAX) return fail("look-ahead set larger than RTX_AHEAD_MAX");
    // batches take the contexts in turn; a context's stream orders a batch after the one that used it last
    {
        const int next = (int)(r->frameSeq % r->ctx.size());
        if (count > r->ctx[(size_t)next].capFrames) {
            if (ahead) { if (grow_ctx(r, next, std::max(count, r->aheadBatch))) return 1; }
            else if (grow_queues(r)) return 1;
            if (count > r->ctx[(size_t)next].capFrames) return fail("frame batch larger than the queues");
        }
    }
    activate(r, (int)(r->frameSeq++ % r->ctx.size()));
    if (r->ctx[r->active].abortPending) {      // the abort of the context's last set must have landed before ctrl[] is cleared for this one
        HIP_TRY(hipStreamWaitEvent(r->stream, r->ctx[r->active].abortDone, 0));
        r->ctx[r->active].abortPending = false;
    }
    if (ahead) {
        r->curStats = r->ctx[r->active].dStatsAhead;
        HIP_TRY(hipMemsetAsync(r->curStats, 0, (size_t)RTX_AHEAD_MAX * RTX_STAT_WORDS * sizeof(unsigned long long), r->stream));
    }
    const int slot = r->ringPos;
    r->ringPos = (r->ringPos + 1) % kRing;
    finish_timing(r, (slot + kRing - kLag) % kRing);     // the batch queued kLag batches ago has long finished
    DevBatch bt{};
    bt.count = count;
    for (int i = 0; i < RTX_MAX_BATCH; ++i) bt.fr[i] = make_frame(r, &frames[std::min(i, count - 1)].first, frames[std::min(i, count - 1)].second);
    const uint32_t slots = r->numSlots * (uint32_t)count;
    {   // queue lengths of this context's previous batch, if they have arrived
        RtxRenderer::FrameCtx& c = r->ctx[r->active];
        if (c.countsPending && hipEventQuery(c.countsReady) == hipSuccess) {
            // the hints describe a full batch
            for (int b = 1; b < 4; ++b) r->queueHint[b] = (uint32_t)std::min<uint64_t>((uint64_t)r->numSlots * r->batch, (uint64_t)c.hCounts[b] * r->batch / std::max(1, c.countsFrames));
            c.countsPending = false;
        }
    }
    HIP_TRY(hipMemsetAsync(r->dCtrl, 0, RTX_CTRL_WORDS * sizeof(uint32_t), r->stream));
    hipEvent_t* ev = r->ev[slot];
    HIP_TRY(hipEventRecord(ev[0], r->stream));
    set_grids(r, slots);
    dispatch_primary(r, bt, 0);
    HIP_TRY(hipEventRecord(ev[1], r->stream));
    if (!primaryOnly) {
        int in = 0;
        for (int b = 0; b < 4; ++b) {
            if (b > 0) {
                set_grids(r, std::min(slots, (uint32_t)((uint64_t)std::min(r->queueHint[b], r->queueHint[b - 1]) * count / r->batch)));
                dispatch_trace(r, r->dRqO[in], r->dRqD[in], b, 2 * b, 0);
                HIP_TRY(hipEventRecord(ev[2 * b + 1], r->stream));
            }
            hipLaunchKernelGGL(k_shade, dim3(r->launchShade), dim3(RTX_BLOCK), RTX_SHADE_LDS_BYTES, r->stream, r->sc, bt, r->dTiles, b, r->linear ? 1 : 0,
                               r->dRqO[in], r->dRqD[in], r->dHit, r->dRqO[in ^ 1], r->dRqD[in ^ 1],
                               r->dLight, r->dAcc, r->dSeeds, r->dCtrl, r->curStats, 2 * b + 1, ahead ? 1 : 0);
            HIP_TRY(hipEventRecord(ev[2 * b + 2], r->stream));
            in ^= 1;
        }
        if (!ahead && queue_resolve(r, bt, 0, count)) return 1;
        {
            RtxRenderer::FrameCtx& c = r->ctx[r->active];
            if (!c.countsPending) {
                c.countsFrames = count;
                HIP_TRY(hipMemcpyAsync(c.hCounts, r->dCtrl, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
                HIP_TRY(hipEventRecord(c.countsReady, r->stream));
                c.countsPending = true;
            }
        }
        HIP_TRY(hipEventRecord(ev[kEvents - 1], r->stream));
        r->pending[slot] = true;
        if (ahead) {
            r->aheadSlot[slot] = true;
            for (int i = 0; i < count; ++i) r->ahead.push_back(RtxRenderer::Ahead{frames[i].first, frames[i].second, r->active, bt, slot, i});
        } else {
            r->totals.frames += (uint64_t)count;
            r->primaryRays += r->localPixels * (uint64_t)count;
        }
    }
    r->curStats = r->dStats;
    HIP_TRY(hipGetLastError());
    return 0;
}

// the look-ahead frame is the one that is asked for: its resolve, its counters
static int take_ahead(RtxRenderer* r) {
    HIP_TRY(hipSetDevice(r->device));
    const RtxRenderer::Ahead& a = r->ahead.front();
    activate(r, a.ctx);
    if (queue_resolve(r, a.bt, a.index, a.index + 1)) return 1;
    hipLaunchKernelGGL(k_add_stats, dim3(1), dim3(64), 0, r->stream, r->dStats, r->ctx[r->active].dStatsAhead + (size_t)a.index * RTX_STAT_WORDS, RTX_STAT_WORDS);
    HIP_TRY(hipGetLastError());
    r->totals.frames += 1;
    r->primaryRays += r->localPixels;
    if (r->ahead.front().ringSlot >= 0) r->aheadSlot[r->ahead.front().ringSlot] = false;      // the set's device time counts from its first frame asked for
    r->ahead.pop_front();
    return 0;
}

// renders whatever rtx_render(pixels = NULL) has collected
static int flush_queued(RtxRenderer* r) {
    if (r->queued.empty()) return 0;
    std::vector<std::pair<RtxCamera, uint32_t>> q;
    q.swap(r->queued);
    return render_batch(r, q.data(), (int)q.size(), false);
}

// Sets started ahead that are not going to be asked for are made to run out at once instead of in full (up to three sets of
// eight frames: 120 ms at 4K, behind which the frame the caller really asked for would wait). Every persistent kernel draws
// its work through the work counters in its context's ctrl[] (WorkIter::next, the refills of the trace kernels) and a wave
// stops drawing when a counter has passed the end of its queue segment — so the control stream overwrites ALL work counters
// of the set's context (every launch of the set, started or not) with a value beyond any queue: waves finish the run they
// hold, every later draw comes back empty, and the launches still queued find nothing to do (a few microseconds each).
// Nothing the set wrote is ever read: only sets of which NO frame has been asked for are cut short (a frame that was taken
// has its resolve queued behind the set's launches); their counters live in the context's own dStatsAhead, their queue
// lengths are not used as hints. The context's next set waits for abortDone before it clears ctrl[] (render_batch).
// kAbortValue + a segment's start + the draws that follow must stay below 2^32 (k_trace_qw adds the three): queues of up
// to 2^30 entries qualify (a batch of 32 frames at 4K has 2^28), longer ones run out as before.
static const uint32_t kAbortValue = 0x7fffff00u;
static void abort_ahead(RtxRenderer* r) {
    if (!r->ctl || r->ahead.empty()) return;
    bool device = false;
    for (const RtxRenderer::Ahead& a : r->ahead) {
        if (a.index != 0 || a.ctx < 0) continue;          // frame 0 of a set is still waiting: no frame of that set was taken
        RtxRenderer::FrameCtx& c = r->ctx[(size_t)a.ctx];
        if ((uint64_t)r->numSlots * (uint64_t)std::max(1, c.capFrames) > 0x3fffffffull || c.abortPending) continue;
        if (!device) { if (hipSetDevice(r->device) != hipSuccess) { (void)hipGetLastError(); return; } device = true; }
        if (hipMemsetD32Async((hipDeviceptr_t)(c.dCtrl + RTX_CTRL_XWORK(0, 0)), (int)kAbortValue, RTX_CTRL_WORDS - RTX_CTRL_XWORK(0, 0), r->ctl) != hipSuccess ||
            hipEventRecord(c.abortDone, r->ctl) != hipSuccess) { (void)hipGetLastError(); return; }      // (not cut short, then: the set runs out in full)
        c.abortPending = true;
        c.countsPending = false;       // queue lengths of a set that was cut short say nothing about the next one
        r->abortedSets++;
    }
}
static void clear_ahead(RtxRenderer* r) { abort_ahead(r); r->ahead.clear(); r->aheadNext = 1; }

// every frame queued or in flight finished (the host-side entry points that touch the framebuffer start here)
// frames started ahead are not going to be asked for (and the loop, if there was one, is over)
static void drop_ahead(RtxRenderer* r) { clear_ahead(r); r->haveLast = false; r->streak = 0; r->rendersSinceDelivery = 0; }

// May another set be started ahead? Every set occupies a context until its last frame has been asked for (or all are
// dropped); sets take the contexts in turn, so the next launch lands on the context of the oldest set. With a set
// partly used up (the one the caller is being served from) every context may hold one; otherwise one context stays free
// for the frame the caller asks for directly.
static bool may_start_ahead(const RtxRenderer* r) {
    if (r->lookAhead <= 0) return false;
    int sets = 0, last = -1;
    for (const RtxRenderer::Ahead& a : r->ahead) if (a.ringSlot != last) { ++sets; last = a.ringSlot; }
    const bool partial = !r->ahead.empty() && r->ahead.front().index > 0;
    const int limit = std::min(r->lookAhead + (partial ? 1 : 0), (int)r->ctx.size() - (partial ? 0 : 1));
    return sets < limit;
}
// the next `aheadBatch` frames of the loop (same camera, consecutive frame counters) on every renderer of `all`
static int start_ahead(RtxRenderer* const* all, int n, const RtxCamera& cam, uint32_t firstFc) {
    RtxRenderer* r = all[0];
    std::vector<std::pair<RtxCamera, uint32_t>> frames;
    // the sets grow while the loop lasts (1, 2, 4 ... aheadBatch frames): a short loop, or a wrong guess early on, wastes
    // little, a long one gets the efficiency of whole batches
    const int count = std::max(1, std::min(r->aheadNext, r->aheadBatch));
    for (int i = 0; i < count; ++i) frames.emplace_back(cam, firstFc + (uint32_t)i);
    for (int k = 0; k < n; ++k) if (render_batch(all[k], frames.data(), (int)frames.size(), false, true)) return 1;
    r->aheadNext = std::min(r->aheadBatch, count * 2);
    return 0;
}

static int sync_all(RtxRenderer* r) {
    drop_ahead(r);
    if (flush_queued(r)) return 1;
    if (r->ctl) HIP_TRY(hipStreamSynchronize(r->ctl));      // (before the contexts: an abort is what lets their streams finish early)
    for (auto& c : r->ctx) { if (c.stream) HIP_TRY(hipStreamSynchronize(c.stream)); c.abortPending = false; }
    return 0;
}

static int render_frame(RtxRenderer* r, const RtxCamera* cam, uint32_t fc, bool primaryOnly) {
    if (flush_queued(r)) return 1;
    const std::pair<RtxCamera, uint32_t> one(*cam, fc);
    return render_batch(r, &one, 1, primaryOnly);
}

// device framebuffer -> host array
static int dev_to_host(RtxRenderer* r, const void* devSrc, void* hostDst, size_t bytes);
static int fb_to_host(RtxRenderer* r, float* pixels, size_t bytes) { return dev_to_host(r, r->dFb, pixels, bytes); }

// The caller's array is ordinary pageable memory (page-locking it per call costs more than the copy, and caching a
// registration across calls would outlive the array): frames are copied through a page-locked staging buffer of the
// renderer in 8 MiB pieces, and four host threads move each piece on as soon as it has landed.
// 133.7 MB at 4K: 8.5 ms through the runtime's pageable path, 3.0 ms this way.
static const size_t kStagePiece = (size_t)8 << 20;

// queues the copy of src[0, bytes) into the staging buffer on `stream`, one event per piece
static int stage_queue(RtxRenderer* r, hipStream_t stream, const char* src, size_t bytes) {
    if (r->hStage && r->hStageBytes < bytes) { hipHostFree(r->hStage); r->hStage = nullptr; }
    if (!r->hStage) { HIP_TRY(hipHostMalloc((void**)&r->hStage, bytes, hipHostMallocDefault)); r->hStageBytes = bytes; }
    const size_t n = (bytes + kStagePiece - 1) / kStagePiece;
    while (r->stageEv.size() < n) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); r->stageEv.push_back(e); }
    for (size_t k = 0; k < n; ++k) {
        const size_t off = k * kStagePiece, len = std::min(kStagePiece, bytes - off);
        HIP_TRY(hipMemcpyAsync(r->hStage + off, src + off, len, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(r->stageEv[k], stream));
    }
    return 0;
}
// staging buffer -> dst, piece by piece as they land
static int stage_deliver(RtxRenderer* r, char* dst, size_t bytes) {
    const size_t n = (bytes + kStagePiece - 1) / kStagePiece;
    const int T = 4;
    std::atomic<int> failed{0};
    auto mover = [&](int t) {
        if (hipSetDevice(r->device) != hipSuccess) { failed = 1; return; }
        for (size_t k = (size_t)t; k < n; k += T) {
            if (hipEventSynchronize(r->stageEv[k]) != hipSuccess) { failed = 1; return; }
            const size_t off = k * kStagePiece, len = std::min(kStagePiece, bytes - off);
            memcpy(dst + off, r->hStage + off, len);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(mover, t);
    mover(0);
    for (auto& th : pool) th.join();
    if (failed) return fail("device-to-host copy of the frame failed");
    return 0;
}

// A caller's array that is page-locked (hipHostMalloc, hipHostRegister, a pinned tensor) takes the copy directly: the DMA
// engine writes it at the link's rate and no host thread touches the bytes.
static bool host_is_page_locked(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }      // ordinary memory: "invalid value", not an error of ours
    return a.type == hipMemoryTypeHost;
}
// src[0, bytes) on the device -> dst on the host, behind whatever `stream` holds; returns when the bytes are there
static int deliver_bytes(RtxRenderer* r, hipStream_t stream, const char* src, char* dst, size_t bytes) {
    if (host_is_page_locked(dst)) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return 0;
    }
    if (stage_queue(r, stream, src, bytes)) return 1;
    return stage_deliver(r, dst, bytes);
}

// device array of this renderer -> caller's host array
static int dev_to_host(RtxRenderer* r, const void* devSrc, void* hostDst, size_t bytes) {
    if (bytes >= kStagePiece && !getenv("RTX_PLAIN_COPY")) return deliver_bytes(r, r->stream, (const char*)devSrc, (char*)hostDst, bytes);
    HIP_TRY(hipMemcpyAsync(hostDst, devSrc, bytes, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return 0;
}

static int copy_out(RtxRenderer* r, float* pixels) {
    const size_t fbBytes = (size_t)r->width * r->height * sizeof(float4);
    if (r->cfg.worldSize == 1) {
        return fb_to_host(r, pixels, fbBytes);
    }
    r->hostStage.resize((size_t)r->width * r->height * 4);
    HIP_TRY(hipMemcpyAsync(r->hostStage.data(), r->dFb, fbBytes, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    for (const int2& t : r->localTiles) {
        const int w = std::min(r->tileW, (int)r->width - t.x), h = std::min(r->tileH, (int)r->height - t.y);
        for (int y = 0; y < h; ++y) {
            const size_t off = ((size_t)(t.y + y) * r->width + t.x) * 4;
            memcpy(pixels + off, r->hostStage.data() + off, (size_t)w * 4 * sizeof(float));
        }
    }
    return 0;
}

// Device group: every member's tiles into the leader's resident framebuffer, once per delivered frame. Nothing in it waits
// on the host until the end: a member's pack rides on the stream of the context that resolved last on that member (behind
// its last resolve; the resolves are chained), the peer copy (xGMI between GPUs) behind the pack, and an event marks the
// arrival at the leader; the leader's unpacks wait for those events on the stream behind its own last resolve. The members
// work side by side — packs on their own devices, copies on their own links.
static int gather_group(RtxRenderer* r, bool keepAhead = false, hipStream_t* leaderStream = nullptr) {
    if (r->peers.empty()) return 0;
    if (!keepAhead) drop_ahead(r);
    if (flush_queued(r)) return 1;
    for (RtxRenderer* m : r->peers) {
        HIP_TRY(hipSetDevice(m->device));
        if (!keepAhead) drop_ahead(m);
        if (flush_queued(m)) return 1;
        const uint64_t total = (uint64_t)m->localTiles.size() * m->tileW * m->tileH;
        const hipStream_t ms = m->lastResolved >= 0 ? m->ctx[m->lastResolved].stream : m->stream;
        if (total) {
            RtxCamera dummy{}; const DevFrame fr = make_frame(m, &dummy, 0);
            hipLaunchKernelGGL(k_pack_tiles, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ms, fr, m->dTiles,
                               (int)m->localTiles.size(), m->dFb, (float4*)m->dPackedOwn);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyPeerAsync(m->dPackedAtLeader, r->device, m->dPackedOwn, m->device, total * sizeof(float4), ms));
        }
        if (!m->arrived) HIP_TRY(hipEventCreateWithFlags(&m->arrived, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(m->arrived, ms));
    }
    HIP_TRY(hipSetDevice(r->device));
    const hipStream_t ls = r->lastResolved >= 0 ? r->ctx[r->lastResolved].stream : r->stream;
    for (RtxRenderer* m : r->peers) {
        HIP_TRY(hipStreamWaitEvent(ls, m->arrived, 0));
        const int src = m->cfg.rank;
        const uint64_t total = (uint64_t)r->tilesOfRank[src].size() * r->tileW * r->tileH;
        if (!total) continue;
        RtxCamera dummy{}; const DevFrame fr = make_frame(r, &dummy, 0);
        hipLaunchKernelGGL(k_unpack_tiles, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ls, fr, r->dTilesOfRank[src],
                           (int)r->tilesOfRank[src].size(), (const float4*)m->dPackedAtLeader, r->dFb);
        HIP_TRY(hipGetLastError());
    }
    if (leaderStream) { *leaderStream = ls; return 0; }      // the caller goes on behind the unpacks on this stream
    // the caller copies the frame out next (on whatever stream is current), and the members may be given new frames
    HIP_TRY(hipStreamSynchronize(ls));
    return 0;
}

static bool delivers_with_look_ahead(const RtxRenderer* r) {
    return r->lookAhead > 0 && r->peers.empty() && !r->leader && !getenv("RTX_PLAIN_COPY");
}

// Hands the image back: the float frame, or (present) its 8-bit conversion. Waits for the resolve that was queued last —
// the resolves are chained, so every frame asked for so far is in — and not for frames started ahead; tops the look-ahead
// up before the copy when the caller is in the loop "one frame, one delivery".
static int deliver(RtxRenderer* r, bool present, float gamma, void* dst) {
    HIP_TRY(hipSetDevice(r->device));
    // The delivery rides on the stream of the context whose resolve was queued last: behind that resolve, and on a stream
    // that has nothing else to do until its context comes round again. (A stream of its own would be the fifth of the
    // process with HIP's default 4 hardware queues — null stream, three contexts — and share a queue with one of them:
    // when that one is a context busy with a frame started ahead, the copy waits for the frame: 9.8 instead of 7.9 ms.)
    const hipStream_t ds = r->lastResolved >= 0 ? r->ctx[r->lastResolved].stream : r->stream;
    const uint32_t n = r->width * r->height;
    const char* src = (const char*)r->dFb;
    size_t bytes = (size_t)n * sizeof(float4);
    if (present) {
        if (!r->dPresent) HIP_TRY(hipMalloc((void**)&r->dPresent, (size_t)n * sizeof(uint32_t)));
        hipLaunchKernelGGL(k_present_rgba8, dim3((n + 255u) / 256u), dim3(256), 0, ds, r->dFb, r->dPresent, n, gamma > 0.0f ? 1.0f / gamma : 0.0f);
        HIP_TRY(hipGetLastError());
        src = (const char*)r->dPresent; bytes = (size_t)n * sizeof(uint32_t);
    }
    const bool loop = r->rendersSinceDelivery == 1 && r->streak >= 1 && r->haveLast;
    r->rendersSinceDelivery = 0;
    while (loop && may_start_ahead(r))
        if (start_ahead(&r, 1, r->lastCam, r->lastFc + 1u + (uint32_t)r->ahead.size())) return 1;
    return deliver_bytes(r, ds, src, (char*)dst, bytes);
}

static int render_one(RtxRenderer* r, const RtxCamera* camera, uint32_t frameCounter, float* pixels_rgba);

int rtx_render(RtxRenderer* r, const RtxCamera* camera, uint32_t frameCounter, float* pixels_rgba) {
    if (!r) return fail("renderer is NULL");
    if (check_camera(camera)) return 1;
    if (r->leader) return fail("this renderer is a member of a device group: call its leader");
    if (r->peers.empty()) return render_one(r, camera, frameCounter, pixels_rgba);
    if (r->lookAhead > 0 && !getenv("RTX_PLAIN_COPY")) {
        // Device group in the loop "one frame, one delivery": the same look-ahead as a single renderer's (render_one / deliver),
        // kept in step on all members — every member holds the same frames ahead, the leader keeps the books.
        std::vector<RtxRenderer*> all{r};
        all.insert(all.end(), r->peers.begin(), r->peers.end());
        const bool predicted = r->haveLast && frameCounter == r->lastFc + 1u && memcmp(camera, &r->lastCam, sizeof(RtxCamera)) == 0;
        r->streak = predicted ? r->streak + 1 : 0;
        r->haveLast = true; r->lastCam = *camera; r->lastFc = frameCounter; r->rendersSinceDelivery++;
        const bool match = !r->ahead.empty() && r->ahead.front().fc == frameCounter && memcmp(camera, &r->ahead.front().cam, sizeof(RtxCamera)) == 0;
        for (RtxRenderer* m : all) {
            if (match && !m->ahead.empty()) { if (take_ahead(m)) return 1; continue; }
            clear_ahead(m);
            if (!pixels_rgba && m->batch > 1) {
                if (m->capFrames < m->batch && grow_queues(m)) return 1;      // the first queued frame: room for whole batches
                m->queued.emplace_back(*camera, frameCounter);
                if ((int)m->queued.size() >= m->batch && flush_queued(m)) return 1;
            } else if (render_frame(m, camera, frameCounter, false)) return 1;
        }
        if (!pixels_rgba) return 0;
        hipStream_t ls = nullptr;
        if (gather_group(r, true, &ls)) return 1;
        const bool loop = r->rendersSinceDelivery == 1 && r->streak >= 1;
        r->rendersSinceDelivery = 0;
        while (loop && may_start_ahead(r))
            if (start_ahead(all.data(), (int)all.size(), *camera, frameCounter + 1u + (uint32_t)r->ahead.size())) return 1;
        HIP_TRY(hipSetDevice(r->device));
        const size_t bytes = (size_t)r->width * r->height * sizeof(float4);
        return deliver_bytes(r, ls, (const char*)r->dFb, (char*)pixels_rgba, bytes);
    }
    // device group: the frame is queued on every member (their launches run side by side on their devices) ...
    if (render_one(r, camera, frameCounter, nullptr)) return 1;
    for (RtxRenderer* m : r->peers) if (render_one(m, camera, frameCounter, nullptr)) return 1;
    if (!pixels_rgba) return 0;
    // ... and delivered from the leader after one gather
    if (gather_group(r)) return 1;
    return fb_to_host(r, pixels_rgba, (size_t)r->width * r->height * sizeof(float4));
}

static int render_one(RtxRenderer* r, const RtxCamera* camera, uint32_t frameCounter, float* pixels_rgba) {
    const bool ahead = delivers_with_look_ahead(r);
    if (ahead) {
        const bool predicted = r->haveLast && frameCounter == r->lastFc + 1u && memcmp(camera, &r->lastCam, sizeof(RtxCamera)) == 0;
        r->streak = predicted ? r->streak + 1 : 0;
        r->haveLast = true; r->lastCam = *camera; r->lastFc = frameCounter; r->rendersSinceDelivery++;
        if (!r->ahead.empty() && r->ahead.front().fc == frameCounter && memcmp(camera, &r->ahead.front().cam, sizeof(RtxCamera)) == 0) {
            if (take_ahead(r)) return 1;           // (nothing is collected at this point: frames are started ahead by a delivery, which renders the collection first)
            return pixels_rgba ? deliver(r, false, 0.0f, pixels_rgba) : 0;
        }
        clear_ahead(r);                            // started for nothing: cut short
    }
    if (!pixels_rgba && r->batch > 1) {
        // nothing has to be delivered yet: collect the frame; a full batch is rendered by one set of launches
        if (r->capFrames < r->batch && grow_queues(r)) return 1;      // the first queued frame: room for whole batches
        r->queued.emplace_back(*camera, frameCounter);
        if ((int)r->queued.size() >= r->batch) return flush_queued(r);
        return 0;
    }
    if (ahead) {
        if (render_frame(r, camera, frameCounter, false)) return 1;
        return pixels_rgba ? deliver(r, false, 0.0f, pixels_rgba) : 0;
    }
    if (render_frame(r, camera, frameCounter, false)) return 1;
    if (pixels_rgba) return copy_out(r, pixels_rgba);
    return 0;
}

static int primary_hits_one(RtxRenderer* r, const RtxCamera* camera, uint32_t frameCounter, int32_t* objectIDs);

int rtx_primary_hits(RtxRenderer* r, const RtxCamera* camera, uint32_t frameCounter, int32_t* objectIDs) {
    if (!r || !objectIDs) return fail("NULL argument");
    if (check_camera(camera)) return 1;
    if (r->peers.empty()) return primary_hits_one(r, camera, frameCounter, objectIDs);
    // device group: every member answers for its own tiles
    if (primary_hits_one(r, camera, frameCounter, objectIDs)) return 1;
    std::vector<int32_t> part((size_t)r->width * r->height);
    for (RtxRenderer* m : r->peers) {
        if (primary_hits_one(m, camera, frameCounter, part.data())) return 1;
        for (const int2& t : m->localTiles) {
            const int w = std::min(m->tileW, (int)m->width - t.x), h = std::min(m->tileH, (int)m->height - t.y);
            for (int y = 0; y < h; ++y) {
                const size_t off = (size_t)(t.y + y) * m->width + t.x;
                memcpy(objectIDs + off, part.data() + off, (size_t)w * sizeof(int32_t));
            }
        }
    }
    return 0;
}

static int primary_hits_one(RtxRenderer* r, const RtxCamera* camera, uint32_t frameCounter, int32_t* objectIDs) {
    drop_ahead(r);
    if (render_frame(r, camera, frameCounter, true)) return 1;
    const size_t npx = (size_t)r->width * r->height;
    int* dIds = nullptr;
    HIP_TRY(hipMalloc((void**)&dIds, npx * sizeof(int)));
    hipError_t e = hipMemsetAsync(dIds, 0xff, npx * sizeof(int), r->stream);
    const DevFrame fr = make_frame(r, camera, frameCounter);
    if (e == hipSuccess && r->numSlots) {
        hipLaunchKernelGGL(k_hits_to_ids, dim3((r->numSlots + 255) / 256), dim3(256), 0, r->stream, r->sc, fr, r->dTiles, r->dHit, dIds);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(objectIDs, dIds, npx * sizeof(int), hipMemcpyDeviceToHost, r->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
    hipFree(dIds);
    if (e != hipSuccess) return fail(std::string("rtx_primary_hits: ") + hipGetErrorString(e));
    return 0;
}

int rtx_trace_rays(RtxRenderer* r, const float* origins, const float* directions, uint64_t n,
                   int32_t* objectIDs, float* points, float* normals) {
    if (!r || !origins || !directions) return fail("NULL argument");
    if (n == 0) return 0;
    if (n > 0x7fffff00ull) return fail("too many rays");
    HIP_TRY(hipSetDevice(r->device));
    if (sync_all(r)) return 1;
    float4 *dO = nullptr, *dD = nullptr, *dP = nullptr, *dN = nullptr; int2* dH = nullptr; int* dI = nullptr;
    uint32_t* dSlowTmp = nullptr;
    // the 4th direction lane is a flag inside the queues (1 = no ray); callers' 4th lane is "ignored"
    std::vector<float> dirs(directions, directions + 4 * n);
    for (uint64_t i = 0; i < n; ++i) dirs[4 * i + 3] = 0.0f;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; return e == hipSuccess; };
    ok(hipMalloc((void**)&dO, n * sizeof(float4))); ok(hipMalloc((void**)&dD, n * sizeof(float4)));
    ok(hipMalloc((void**)&dP, n * sizeof(float4))); ok(hipMalloc((void**)&dN, n * sizeof(float4)));
    ok(hipMalloc((void**)&dH, n * sizeof(int2))); ok(hipMalloc((void**)&dI, n * sizeof(int)));
    ok(hipMalloc((void**)&dSlowTmp, n * sizeof(uint32_t)));
    if (e == hipSuccess) {
        ok(hipMemcpyAsync(dO, origins, n * sizeof(float4), hipMemcpyHostToDevice, r->stream));
        ok(hipMemcpyAsync(dD, dirs.data(), n * sizeof(float4), hipMemcpyHostToDevice, r->stream));
        ok(hipMemsetAsync(r->dCtrl, 0, RTX_CTRL_WORDS * sizeof(uint32_t), r->stream));
        uint32_t cnt = (uint32_t)n;
        ok(hipMemcpyAsync(r->dCtrl + RTX_CTRL_COUNT(1), &cnt, sizeof(cnt), hipMemcpyHostToDevice, r->stream));
        ok(hipStreamSynchronize(r->stream));   // cnt is a stack variable
    }
    if (e == hipSuccess) {
        int2* saveH = r->dHit; uint32_t* saveS = r->dSlow;
        r->dHit = dH; r->dSlow = dSlowTmp;
        hipEvent_t t0 = nullptr, t1 = nullptr;
        ok(hipEventCreate(&t0)); ok(hipEventCreate(&t1));
        if (t0) ok(hipEventRecord(t0, r->stream));
        set_grids(r, (uint32_t)n);
        dispatch_trace(r, dO, dD, 1, 0, 0);
        if (t1) ok(hipEventRecord(t1, r->stream));
        r->dHit = saveH; r->dSlow = saveS;
        hipLaunchKernelGGL(k_probe_finish, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, r->stream, r->sc, r->linear ? 1 : 0,
                           dO, dD, dH, (uint32_t)n, dI, dP, dN);
        ok(hipGetLastError());
        if (objectIDs) ok(hipMemcpyAsync(objectIDs, dI, n * sizeof(int), hipMemcpyDeviceToHost, r->stream));
        if (points) ok(hipMemcpyAsync(points, dP, n * sizeof(float4), hipMemcpyDeviceToHost, r->stream));
        if (normals) ok(hipMemcpyAsync(normals, dN, n * sizeof(float4), hipMemcpyDeviceToHost, r->stream));
        ok(hipStreamSynchronize(r->stream));
        float ms = 0.0f;
        if (t0 && t1 && hipEventElapsedTime(&ms, t0, t1) == hipSuccess) r->totals.lastTraceMs = ms;   // device time of the trace launches
        if (t0) hipEventDestroy(t0);
        if (t1) hipEventDestroy(t1);
    }
    void* ptrs[] = {dO, dD, dP, dN, dH, dI, dSlowTmp};
    for (void* p : ptrs) if (p) hipFree(p);
    if (e != hipSuccess) return fail(std::string("rtx_trace_rays: ") + hipGetErrorString(e));
    return 0;
}

// See include/rtx.h. Same kernels and launch sequence as rtx_trace_rays; what that call allocates per call (hit
// distances, hit leaves, side list, staged directions) and the work counters live in this device's scratch[] instead.
int rtx_trace_rays_noalloc(RtxRenderer* r, const float* d_origins, const float* d_directions, uint64_t n,
                           int32_t* d_objectIDs, float* d_points, float* d_normals) {
    if (!r || !d_origins || !d_directions) return fail("NULL argument");
    if (n == 0) return 0;
    if (n > RTX_NOALLOC_MAX_RAYS)
        return fail("rtx_trace_rays_noalloc: " + std::to_string(n) + " rays exceed the fixed scratch of " + std::to_string(RTX_NOALLOC_MAX_RAYS) +
                    " rays per call (split the batch, or use rtx_trace_rays)");
    HIP_TRY(hipSetDevice(r->device));
    unsigned char* base = g_scratch[r->device];          // set in rtx_create on this device
    if (!base) return fail("rtx_trace_rays_noalloc: scratch address of this device is unknown");
    if (sync_all(r)) return 1;
    uint32_t* sCtrl = (uint32_t*)base;
    int2* sH = (int2*)(base + RTX_NOALLOC_CTRL_BYTES);
    uint32_t* sSlow = (uint32_t*)(sH + RTX_NOALLOC_MAX_RAYS);
    float4* sD = (float4*)(sSlow + RTX_NOALLOC_MAX_RAYS);
    const uint32_t cnt = (uint32_t)n;
    // the array keeps the last call's contents: the counters must start at zero, the queue length is this call's
    HIP_TRY(hipMemsetAsync(sCtrl, 0, RTX_CTRL_WORDS * sizeof(uint32_t), r->stream));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(sCtrl + RTX_CTRL_COUNT(1)), (int)cnt, 1, r->stream));
    hipLaunchKernelGGL(k_stage_directions, dim3((cnt + 255u) / 256u), dim3(256), 0, r->stream, (const float4*)d_directions, sD, cnt);
    int2* saveH = r->dHit; uint32_t* saveS = r->dSlow; uint32_t* saveC = r->dCtrl;
    r->dHit = sH; r->dSlow = sSlow; r->dCtrl = sCtrl;
    set_grids(r, cnt);
    dispatch_trace(r, (const float4*)d_origins, sD, 1, 0, 0);          // counted like rtx_trace_rays: as a bounce-1 queue
    r->dHit = saveH; r->dSlow = saveS; r->dCtrl = saveC;
    hipLaunchKernelGGL(k_probe_finish, dim3((cnt + 255u) / 256u), dim3(256), 0, r->stream, r->sc, r->linear ? 1 : 0,
                       (const float4*)d_origins, sD, sH, cnt, d_objectIDs, (float4*)d_points, (float4*)d_normals);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(r->stream));           // the next call on this GPU reuses the array
    return 0;
}

int rtx_upload_pixels(RtxRenderer* r, const float* pixels) {
    if (!r || !pixels) return fail("NULL argument");
    for (RtxRenderer* m : r->peers) if (rtx_upload_pixels(m, pixels)) return 1;      // each member reads its own tiles of it
    HIP_TRY(hipSetDevice(r->device));
    if (sync_all(r)) return 1;
    HIP_TRY(hipMemcpyAsync(r->dFb, pixels, (size_t)r->width * r->height * sizeof(float4), hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return 0;
}
int rtx_download_pixels(RtxRenderer* r, float* pixels) {
    if (!r || !pixels) return fail("NULL argument");
    if (delivers_with_look_ahead(r)) {
        if (flush_queued(r)) return 1;
        return deliver(r, false, 0.0f, pixels);
    }
    if (gather_group(r)) return 1;
    HIP_TRY(hipSetDevice(r->device));
    if (sync_all(r)) return 1;
    const size_t bytes = (size_t)r->width * r->height * sizeof(float4);
    return fb_to_host(r, pixels, bytes);
}
int rtx_present_rgba8(RtxRenderer* r, float gamma, uint8_t* rgba8) {
    if (!r || !rgba8) return fail("NULL argument");
    if (r->leader) return fail("this renderer is a member of a device group: call its leader");
    if (delivers_with_look_ahead(r)) {
        if (flush_queued(r)) return 1;
        return deliver(r, true, gamma, rgba8);
    }
    if (gather_group(r)) return 1;
    HIP_TRY(hipSetDevice(r->device));
    if (sync_all(r)) return 1;
    const uint32_t n = r->width * r->height;
    if (!r->dPresent) HIP_TRY(hipMalloc((void**)&r->dPresent, (size_t)n * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_present_rgba8, dim3((n + 255u) / 256u), dim3(256), 0, r->stream, r->dFb, r->dPresent, n, gamma > 0.0f ? 1.0f / gamma : 0.0f);
    HIP_TRY(hipGetLastError());
    return dev_to_host(r, r->dPresent, rgba8, (size_t)n * sizeof(uint32_t));       // the staged copy of the float frame, a quarter of the bytes
}

int rtx_clear_pixels(RtxRenderer* r) {
    if (!r) return fail("NULL argument");
    for (RtxRenderer* m : r->peers) if (rtx_clear_pixels(m)) return 1;
    HIP_TRY(hipSetDevice(r->device));
    if (sync_all(r)) return 1;
    HIP_TRY(hipMemsetAsync(r->dFb, 0, (size_t)r->width * r->height * sizeof(float4), r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));      // frames on the other streams must not overtake the clear
    return 0;
}

void* rtx_alloc_host(uint64_t bytes) {
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, (size_t)std::max<uint64_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); fail(std::string("rtx_alloc_host: ") + hipGetErrorString(e)); return nullptr; }
    return p;
}
void rtx_free_host(void* p) { if (p && hipHostFree(p) != hipSuccess) (void)hipGetLastError(); }

int rtx_get_info(RtxRenderer* r, RtxInfo* out) {
    if (!r || !out) return fail("NULL argument");
    *out = r->info;
    out->setsCutShort = r->abortedSets;
    for (const RtxRenderer* m : r->peers) out->setsCutShort += m->abortedSets;
    return 0;
}

int rtx_get_counters(RtxRenderer* r, RtxCounters* out) {
    if (!r || !out) return fail("NULL argument");
    HIP_TRY(hipSetDevice(r->device));
    if (sync_all(r)) return 1;
    // oldest pending frame first so that lastFrameMs ends up describing the newest one
    for (int i = 0; i < kRing; ++i) finish_timing(r, (r->ringPos + i) % kRing);
    unsigned long long st[RTX_STAT_WORDS];
    HIP_TRY(hipMemcpy(st, r->dStats, sizeof(st), hipMemcpyDeviceToHost));
    RtxCounters c = r->totals;
    c.rays = 0;
    // rays of bounce 0 are known on the host; every hit before the last bounce continues (src/ThreadedShader.cpp:155-177),
    // so the rays of bounce b+1 are the hits of bounce b (queue lengths would include the holes of k_shade's runs)
    for (int b = 0; b < 4; ++b) {
        c.hitsPerBounce[b] = st[RTX_STAT_HITS(b)];
        c.raysPerBounce[b] = b == 0 ? r->primaryRays : st[RTX_STAT_HITS(b - 1)];
        c.rays += c.raysPerBounce[b];
    }
    for (int b = 0; b < 4; ++b) { c.innerVisits[b] = st[RTX_STAT_INNER(b)]; c.leafTests[b] = st[RTX_STAT_LEAF(b)]; c.wavePasses[b] = st[RTX_STAT_PASSES(b)]; }
    // device group: work adds up over the members, device times are those of the slowest member
    for (RtxRenderer* m : r->peers) {
        RtxCounters p;
        if (rtx_get_counters(m, &p)) return 1;
        c.rays += p.rays;
        for (int b = 0; b < 4; ++b) {
            c.raysPerBounce[b] += p.raysPerBounce[b]; c.hitsPerBounce[b] += p.hitsPerBounce[b];
            c.innerVisits[b] += p.innerVisits[b]; c.leafTests[b] += p.leafTests[b]; c.wavePasses[b] += p.wavePasses[b];
        }
        c.lastFrameMs = std::max(c.lastFrameMs, p.lastFrameMs); c.lastTraceMs = std::max(c.lastTraceMs, p.lastTraceMs);
        c.lastShadeMs = std::max(c.lastShadeMs, p.lastShadeMs); c.totalMs = std::max(c.totalMs, p.totalMs);
        for (int k = 0; k < 8; ++k) c.kernelMs[k] = std::max(c.kernelMs[k], p.kernelMs[k]);
    }
    *out = c;
    return 0;
}
int rtx_reset_counters(RtxRenderer* r) {
    if (!r) return fail("NULL argument");
    for (RtxRenderer* m : r->peers) if (rtx_reset_counters(m)) return 1;
    HIP_TRY(hipSetDevice(r->device));
    if (sync_all(r)) return 1;
    for (int i = 0; i < kRing; ++i) finish_timing(r, i);
    HIP_TRY(hipMemset(r->dStats, 0, RTX_STAT_WORDS * sizeof(unsigned long long)));
    r->totals = RtxCounters{};
    r->primaryRays = 0;
    return 0;
}

void* rtx_framebuffer_device_ptr(RtxRenderer* r) { return r ? (void*)r->dFb : nullptr; }
void* rtx_stream(RtxRenderer* r) { return r ? (void*)r->stream : nullptr; }
uint64_t rtx_local_pixel_count(RtxRenderer* r) {
    if (!r) return 0;
    uint64_t n = r->localPixels;
    for (RtxRenderer* m : r->peers) n += m->localPixels;       // a device group owns all its members' tiles
    return n;
}
uint32_t rtx_local_tile_count(RtxRenderer* r) { return r ? (uint32_t)r->localTiles.size() : 0; }
uint64_t rtx_packed_floats(RtxRenderer* r, int32_t rank) {
    if (!r || rank < 0 || rank >= r->cfg.worldSize) return 0;
    return (uint64_t)r->tilesOfRank[rank].size() * r->tileW * r->tileH * 4;
}
int rtx_pack_tiles(RtxRenderer* r, void* d_packed) {
    if (!r || !d_packed) return fail("NULL argument");
    if (!r->peers.empty()) return fail("rtx_pack_tiles: a device group delivers whole frames (it is not a rank of a process group)");
    HIP_TRY(hipSetDevice(r->device));
    if (sync_all(r)) return 1;
    const uint64_t total = (uint64_t)r->localTiles.size() * r->tileW * r->tileH;
    if (total == 0) return 0;
    RtxCamera dummy{}; const DevFrame fr = make_frame(r, &dummy, 0);
    hipLaunchKernelGGL(k_pack_tiles, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, r->stream, fr, r->dTiles,
                       (int)r->localTiles.size(), r->dFb, (float4*)d_packed);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(r->stream));
    return 0;
}
int rtx_unpack_tiles(RtxRenderer* r, const void* d_packed, int32_t srcRank) {
    if (!r || !d_packed) return fail("NULL argument");
    if (srcRank < 0 || srcRank >= r->cfg.worldSize) return fail("srcRank out of range");
    HIP_TRY(hipSetDevice(r->device));
    if (sync_all(r)) return 1;
    const uint64_t total = (uint64_t)r->tilesOfRank[srcRank].size() * r->tileW * r->tileH;
    if (total == 0) return 0;
    RtxCamera dummy{}; const DevFrame fr = make_frame(r, &dummy, 0);
    hipLaunchKernelGGL(k_unpack_tiles, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, r->stream, fr, r->dTilesOfRank[srcRank],
                       (int)r->tilesOfRank[srcRank].size(), (const float4*)d_packed, r->dFb);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(r->stream));
    return 0;
}

} // extern "C"
