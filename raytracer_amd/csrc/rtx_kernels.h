// rtx_kernels.h — the per-frame kernels of the HIP backend.
//
// Frame pipeline (one stream, no host round trip inside a frame):
//   k_primary                 slot -> pixel, seed, jitter, primary ray, closest hit     (bounce 0)
//   k_shade        (bounce b) sky or material shading, running-mean write-back of finished
//                             paths, surviving paths appended to the next ray queue in
//                             wave-private runs
//   k_trace_refill (bounce b) closest hit for the queued rays on the 4-wide tree, idle lanes
//                             refilled from the queue inside the traversal loop   (b = 1..3)
//   k_trace_slow   (bounce b) the few rays whose slab arithmetic can produce NaN: reference
//                             topology, reference semantics
//   k_trace        (bounce b) EXACT / LINEAR modes: reference topology or the linear list
// Every kernel is a persistent grid whose waves reserve work from device counters, so queue
// lengths never have to travel to the host. Per-path state (light, accumulator, seed) lives
// in SoA arrays indexed by the path's slot; slots enumerate this rank's screen tiles in
// 8x8-pixel blocks, one block per wave, so a wave's primary rays are spatially coherent.
#pragma once
#include "rtx_device.h"
#include "rtx_traverse.h"
#include "rtx_shade.h"

#define RTX_BLOCK 256
#define RTX_MODE_CULLED 0
#define RTX_MODE_EXACT 1
#define RTX_MODE_LINEAR 2
#define RTX_MODE_CULLED_BIG 3     // CULLED over a tree of more than 2^25 wide nodes (64-bit node addresses)
#define RTX_MODE_WIDEQ 4          // RTX_TRAVERSE_WIDE: quantised nodes, unordered walk, no distance culling (rtx_traverse.h)
#define RTX_MODE_WIDEQ_BIG 5      // ... over more than 2^26 nodes (64-bit node addresses)
#define RTX_INVALID_SLOT (-2)

// ctrl[] layout (uint32): [0..7] queue counts per bounce, [24..27] slow-list lengths, then the work counters:
// one per (launch of the frame, queue segment), each on a 128-byte line of its own. Every queue is cut
// into 8 contiguous segments and the blocks that share an XCD (blockIdx % 8, MI355X_MICROARCH.md
// "Workgroup dispatch") draw from the same segment first, helping with the others once it is handed
// out: an XCD's L2 then sees the rays of one image region at a time instead of a slice of every
// region, and the counter atomics of a launch spread over 8 lines.
#define RTX_CTRL_COUNT(b) (b)
#define RTX_CTRL_SLOW(b) (24 + (b))     // rays of bounce b handed to the reference-topology kernel
#define RTX_CTRL_XWORK(launch, seg) (64 + ((launch) * 8 + (seg)) * 32)     // launch: 2*bounce trace, 2*bounce+1 shade
#define RTX_CTRL_WORDS (64 + 8 * 8 * 32)
// stats[] layout (uint64)
#define RTX_STAT_INNER(b) (b)
#define RTX_STAT_LEAF(b) (4 + (b))
#define RTX_STAT_RAYS(b) (8 + (b))
#define RTX_STAT_HITS(b) (12 + (b))
#define RTX_STAT_PASSES(b) (16 + (b))   // traversal-loop passes of whole waves (lane-refill kernel, countStats only)
#define RTX_STAT_WORDS 20

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Work distribution: a wave reserves a run of items with one device atomic and walks it 64 at a
// time. One word takes ~88 atomics/us chip-wide (MI355X_MICROARCH.md "dequeue"), so reserving 64 at
// a time on one counter caps a launch at ~5.6 G items/s (measured: the shade kernels ran 1.8x faster
// with longer runs), while long runs unbalance the tail of the expensive trace launches (64 at a time
// measured best there). Guided self-scheduling for the cheap items: run length = what is left of the
// segment / (GUIDE x waves drawing from it), between 64 and 4096; GUIDE = 0 means always 64.
// Segments: see RTX_CTRL_XWORK.
__device__ __forceinline__ uint32_t segment_length(uint32_t count) { return ((count + 7u) / 8u + 63u) & ~63u; }

// SEGMENTED = false walks the queue front to back with the whole grid (segment 0 = the queue): k_shade
// appends survivors in the order it consumes paths, and the next trace launch's segments are image
// regions only if that order is the queue order.
template <uint32_t GUIDE, bool SEGMENTED>
struct WorkIter {
    uint32_t cur, end, seen;          // wave-uniform: the reserved run, and how far the segment was seen handed out
    uint32_t seg, tried, segLen, limit;
    __device__ __forceinline__ uint32_t run_length(uint32_t remaining) const {
        if (GUIDE == 0) return 64u;
        const uint32_t waves = SEGMENTED ? (gridDim.x * (RTX_BLOCK / 64u) + 7u) / 8u : gridDim.x * (RTX_BLOCK / 64u);
        const uint32_t g = (remaining / ((GUIDE ? GUIDE : 1u) * waves)) & ~63u;
        return g < 64u ? 64u : (g > 4096u ? 4096u : g);
    }
    __device__ __forceinline__ explicit WorkIter(uint32_t limit_)
        : cur(0), end(0), seen(0), seg(SEGMENTED ? (blockIdx.x & 7u) : 0u), tried(SEGMENTED ? 0u : 7u),
          segLen(SEGMENTED ? segment_length(limit_) : ((limit_ + 63u) & ~63u)), limit(limit_) {}
    // base of this wave's next 64 items, or 0xffffffff when the queue is exhausted
    __device__ __forceinline__ uint32_t next(uint32_t* ctrl, int launch) {
        while (cur >= end) {
            if (tried == 8u) return 0xffffffffu;
            const uint32_t segStart = seg * segLen, segEnd = segStart + segLen < limit ? segStart + segLen : limit;
            if (segStart < limit) {
                const uint32_t grab = run_length(segEnd - segStart > seen ? segEnd - segStart - seen : 0u);
                uint32_t base = 0;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(&ctrl[RTX_CTRL_XWORK(launch, seg)], grab);
                base = __builtin_amdgcn_readfirstlane(base);
                seen = base + grab;
                if (base < segEnd - segStart) {
                    cur = segStart + base;
                    end = cur + grab < segEnd ? cur + grab : segEnd;
                    break;
                }
            }
            seg = (seg + 1u) & 7u; tried++; seen = 0;      // handed out: help with the next segment
        }
        const uint32_t b = cur;
        cur += 64u;
        return b;
    }
};

// Length of the wave-private runs k_shade reserves in the next queue: long enough that the shared
// counter sees few atomics, short enough that the holes left at the end (half a run per wave on
// average) stay a few percent of the queue. RTX_MAX_RUN bounds the queue slack rtx_create allocates.
#define RTX_MAX_RUN 512u
__device__ __forceinline__ uint32_t append_run_length(uint32_t count) {
    const uint32_t waves = gridDim.x * (RTX_BLOCK / 64u);
    const uint32_t g = (count / (8u * waves)) & ~63u;
    return g < 64u ? 64u : (g > RTX_MAX_RUN ? RTX_MAX_RUN : g);
}

__device__ __forceinline__ bool slot_to_pixel(const DevFrame& fr, const int2* __restrict__ tiles, uint32_t slot, int& x, int& y) {
    const uint32_t perTile = (uint32_t)(fr.tileW * fr.tileH);
    const uint32_t tile = slot / perTile, within = slot % perTile;
    const int2 org = tiles[tile];
    const uint32_t blk = within >> 6, l = within & 63u;
    const uint32_t blocksX = (uint32_t)fr.tileW >> 3;
    x = org.x + (int)((blk % blocksX) << 3) + (int)(l & 7u);
    y = org.y + (int)((blk / blocksX) << 3) + (int)(l >> 3);
    return x < fr.width && y < fr.height;
}

// LDS per block: CULLED 2*RTX_LDS_WINDOW words per lane (the stack window); EXACT stackDepth words per lane.
// The culled path's rare reference-topology lanes (axis-parallel rays) keep their stack in the spill buffer.
template <int MODE, bool COUNT>
__device__ __forceinline__ Hit closest_hit(const DevScene& sc, float ox, float oy, float oz, float dx, float dy, float dz,
                                           int* stack, uint32_t& ic, uint32_t& lc) {
    if (MODE == RTX_MODE_LINEAR) return traverse_linear<COUNT>(sc, ox, oy, oz, dx, dy, dz, lc);
    if (MODE == RTX_MODE_EXACT) return traverse_bvh<true, COUNT>(sc, ox, oy, oz, dx, dy, dz, stack, RTX_BLOCK, ic, lc);
    // own tree: clean lanes take the fast path; a lane whose slab test could produce NaN (axis-parallel
    // direction, non-finite input) walks the reference topology with the reference's exact semantics
    Hit h; h.t = INFINITY; h.leaf = -1; h.box = -1;
    const bool clean = (MODE == RTX_MODE_WIDEQ || MODE == RTX_MODE_WIDEQ_BIG) ? q_lane_ok(ox, oy, oz, dx, dy, dz) : lane_is_clean(ox, oy, oz, dx, dy, dz);
    const long long nthreads = (long long)gridDim.x * RTX_BLOCK;
    const long long gtid = (long long)blockIdx.x * RTX_BLOCK + threadIdx.x;
    if (MODE == RTX_MODE_WIDEQ || MODE == RTX_MODE_WIDEQ_BIG) {
        if (clean) {
            // the spill column is the one the reference-topology walk below uses for a lane that is not clean: int2
            // column gtid, ints 2 * nthreads apart. A lane is either clean or not, so the two uses never share a word
            // (an int column per thread, ints nthreads apart, would: entry 2j of thread 2B is entry j of thread B's).
            QStack st;
            st.lds = (int)threadIdx.x; st.stride = RTX_BLOCK; st.spill = (int*)(sc.spill + gtid); st.spillStride = 2 * nthreads; st.nsp = 0; st.nbase = 0; st.lsp = 0;
            h = traverse_q<COUNT, MODE == RTX_MODE_WIDEQ_BIG>(sc, ox, oy, oz, dx, dy, dz, st, ic, lc);
        }
    } else if (clean) {
        LaneStack st;
        st.lds = (int)threadIdx.x; st.stride = RTX_BLOCK; st.spill = sc.spill + gtid; st.spillStride = nthreads; st.sp = 0; st.base = 0;
        h = traverse_fast<COUNT, MODE == RTX_MODE_CULLED_BIG>(sc, ox, oy, oz, dx, dy, dz, st, ic, lc);
    }
    if (__any(!clean)) {
        if (!clean) h = traverse_bvh<true, COUNT>(sc, ox, oy, oz, dx, dy, dz, (int*)(sc.spill + gtid), (int)(2 * nthreads), ic, lc);
    }
    return h;
}

// src/ThreadedShader.cpp:141-149 + src/Camera.h:57-64
__device__ __forceinline__ void primary_ray(const DevFrame& fr, int x, int y, uint32_t& seed, F4& o, F4& d) {
    uint32_t index = (uint32_t)(y * fr.width + x);
    seed = (fr.frameCounter << 16) ^ ((fr.frameCounter >> 13) + index);
    F4 offset = random_direction(seed);
    int px = (int)((float)x + offset.x);
    int py = (int)((float)y + offset.y);
    float cameraX = (float)((2.0 * (double)px / (double)(float)fr.width - 1.0) * (double)fr.aspect * (double)fr.tanHalfFOV * (double)fr.nearView);
    float cameraY = (float)((2.0 * (double)py / (double)(float)fr.height - 1.0) * (double)fr.tanHalfFOV * (double)fr.nearView);
    F4 pos = mk4(fr.pos[0], fr.pos[1], fr.pos[2]);
    F4 p = add4(pos, mul4(mk4(fr.front[0], fr.front[1], fr.front[2]), fr.nearView));
    p = add4(p, mul4(mk4(fr.right[0], fr.right[1], fr.right[2]), cameraX));
    p = add4(p, mul4(mk4(fr.up[0], fr.up[1], fr.up[2]), cameraY));
    o = pos;
    d = normalize4(sub4(p, o));
}

#ifndef RTX_PRIMARY_GUIDE
#define RTX_PRIMARY_GUIDE 0
#endif
// 5 waves per SIMD: ray generation (double-precision jitter and NDC arithmetic) would otherwise claim 115
// VGPRs for the whole kernel and leave 4 waves; capped at 96 it spills 13 dwords outside the traversal loop
// and the launch runs 9 % faster.
#ifndef RTX_PRIMARY_WAVES
#define RTX_PRIMARY_WAVES 5
#endif
#if RTX_PRIMARY_WAVES > 0
#define RTX_PRIMARY_ATTR __attribute__((amdgpu_waves_per_eu(RTX_PRIMARY_WAVES, RTX_PRIMARY_WAVES)))
#else
#define RTX_PRIMARY_ATTR
#endif
#ifndef RTX_QUEUE_NT
#define RTX_QUEUE_NT 1
#endif
// Queue entries are written once and read once, gigabytes apart: streaming loads and stores keep them from displacing
// tree nodes and leaf records in L2 (+0.5 to 1 % of a batch frame; the same on the per-path light and colour arrays
// measured slightly slower, and on the leaf records themselves, which neighbouring rays share, half again slower).
typedef float rtx_v4f __attribute__((ext_vector_type(4)));
typedef int rtx_v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 q_ld4(const float4* p) {
#if RTX_QUEUE_NT
    const rtx_v4f v = __builtin_nontemporal_load((const rtx_v4f*)p); return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void q_st4(float4* p, float4 a) {
#if RTX_QUEUE_NT
    rtx_v4f v; v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w; __builtin_nontemporal_store(v, (rtx_v4f*)p);
#else
    *p = a;
#endif
}
__device__ __forceinline__ int2 q_ld2(const int2* p) {
#if RTX_QUEUE_NT
    const rtx_v2i v = __builtin_nontemporal_load((const rtx_v2i*)p); return make_int2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ void q_st2(int2* p, int2 a) {
#if RTX_QUEUE_NT
    rtx_v2i v; v.x = a.x; v.y = a.y; __builtin_nontemporal_store(v, (rtx_v2i*)p);
#else
    *p = a;
#endif
}
// LDS of the kernels that test leaves on compacted lanes: RTX_QN node-stack words per lane, then RTX_WQ_WORDS per wave
#define RTX_TRACEQW_LDS_BYTES ((RTX_QN * RTX_BLOCK + (RTX_BLOCK / 64) * RTX_WQ_WORDS) * 4)
static_assert(RTX_TRACEQW_LDS_BYTES <= 65536, "a block's LDS is at most 64 KB");
#ifndef RTX_PRIMARY_LEAFQ
#define RTX_PRIMARY_LEAFQ 1        // WIDE: primary rays with the leaf tests on compacted lanes too (needs RTX_TRACEQW_LDS_BYTES of LDS)
#endif
template <int MODE, bool COUNT>
__global__ __launch_bounds__(RTX_BLOCK) RTX_PRIMARY_ATTR void k_primary(DevScene sc, DevBatch bt, const int2* __restrict__ tiles,
                                                       float4* __restrict__ rqO, float4* __restrict__ rqD,
                                                       int2* __restrict__ hit,
                                                       uint32_t* __restrict__ seeds, uint32_t* __restrict__ ctrl,
                                                       unsigned long long* __restrict__ stats, int workIdx) {
    int* stack = rtx_lds + threadIdx.x;
    uint32_t ic = 0, lc = 0;
    const int countIdx = 0;
    const uint32_t nf = (uint32_t)bt.count, tilePix = (uint32_t)(bt.fr[0].tileW * bt.fr[0].tileH);
    WorkIter<RTX_PRIMARY_GUIDE, true> work((uint32_t)bt.fr[0].numSlots * nf);
    for (;;) {
        const uint32_t base = work.next(ctrl, workIdx);
        if (base == 0xffffffffu) break;
        const uint32_t slot = base + (threadIdx.x & 63u);
        // batch slot -> (frame, slot inside the frame); a wave's 64 slots lie inside one tile of one frame
        uint32_t f = 0, p = slot;
        if (nf > 1) {
            const uint32_t tb = base / tilePix;
            f = tb % nf;
            p = (tb / nf) * tilePix + (slot - tb * tilePix);
        }
        const DevFrame& fr = bt.fr[f];
        int x, y;
#if RTX_PRIMARY_LEAFQ
        if (MODE == RTX_MODE_WIDEQ || MODE == RTX_MODE_WIDEQ_BIG) {
            // the wave walks together: node steps per lane, leaf tests on compacted lanes (rtx_traverse.h: QWave)
            const bool inside = slot_to_pixel(fr, tiles, p, x, y);
            uint32_t seed = 0; F4 o = mk4(0, 0, 0), d = mk4(1, 1, 1);
            if (inside) primary_ray(fr, x, y, seed, o, d);
            const bool clean = inside && q_lane_ok(o.x, o.y, o.z, d.x, d.y, d.z);
            QStack st;
            st.lds = (int)threadIdx.x; st.stride = RTX_BLOCK;
            st.spill = (int*)(sc.spill + ((long long)blockIdx.x * RTX_BLOCK + threadIdx.x)); st.spillStride = 2 * (long long)gridDim.x * RTX_BLOCK;      // see closest_hit
            QWave wq;
            wq.init(RTX_QN * RTX_BLOCK + (int)(threadIdx.x >> 6) * RTX_WQ_WORDS);
            Hit h = traverse_q_wave<COUNT, MODE == RTX_MODE_WIDEQ_BIG>(sc, clean, o.x, o.y, o.z, d.x, d.y, d.z, st, wq, threadIdx.x & 63u, ic, lc);
            if (__any(inside && !clean)) {
                if (inside && !clean) h = traverse_bvh<true, COUNT>(sc, o.x, o.y, o.z, d.x, d.y, d.z, (int*)(sc.spill + ((long long)blockIdx.x * RTX_BLOCK + threadIdx.x)), (int)(2 * (long long)gridDim.x * RTX_BLOCK), ic, lc);
            }
            if (inside) {
                q_st4(rqO + slot, make_float4(o.x, o.y, o.z, __int_as_float((int)slot)));
                q_st4(rqD + slot, make_float4(d.x, d.y, d.z, 0.0f));
                q_st2(hit + slot, make_int2(__float_as_int(h.t), h.leaf));
                seeds[slot] = seed;
            } else {
                hit[slot] = make_int2(0, RTX_INVALID_SLOT);
            }
            continue;
        }
#endif
        if (slot_to_pixel(fr, tiles, p, x, y)) {
            uint32_t seed; F4 o, d;
            primary_ray(fr, x, y, seed, o, d);
            Hit h = closest_hit<MODE, COUNT>(sc, o.x, o.y, o.z, d.x, d.y, d.z, stack, ic, lc);
            rqO[slot] = make_float4(o.x, o.y, o.z, __int_as_float((int)slot));
            rqD[slot] = make_float4(d.x, d.y, d.z, 0.0f);
            hit[slot] = make_int2(__float_as_int(h.t), h.leaf);       // distance and leaf in one 8-byte store
            seeds[slot] = seed;
        } else {
            hit[slot] = make_int2(0, RTX_INVALID_SLOT);
        }
    }
    if (COUNT) {
        ic = wave_sum(ic); lc = wave_sum(lc);
        if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[RTX_STAT_INNER(countIdx)], (unsigned long long)ic); atomicAdd(&stats[RTX_STAT_LEAF(countIdx)], (unsigned long long)lc); }
    }
}

template <int MODE, bool COUNT>
__global__ __launch_bounds__(RTX_BLOCK) void k_trace(DevScene sc, const float4* __restrict__ rqO, const float4* __restrict__ rqD,
                                                     int2* __restrict__ hit,
                                                     uint32_t* __restrict__ ctrl, unsigned long long* __restrict__ stats,
                                                     uint32_t fixedCount, int countIdx, int workIdx) {
    int* stack = rtx_lds + threadIdx.x;
    uint32_t ic = 0, lc = 0;
    const uint32_t count = countIdx == 0 ? fixedCount : ctrl[RTX_CTRL_COUNT(countIdx)];
    WorkIter<0, true> work(count);
    for (;;) {
        const uint32_t base = work.next(ctrl, workIdx);
        if (base == 0xffffffffu) break;
        const uint32_t q = base + (threadIdx.x & 63u);
        if (q < count) {
            float4 o = rqO[q], d = rqD[q];
            if (d.w != 0.0f) {
                hit[q] = make_int2(0, RTX_INVALID_SLOT);      // hole left by k_shade's run reservation
            } else {
                Hit h = closest_hit<MODE, COUNT>(sc, o.x, o.y, o.z, d.x, d.y, d.z, stack, ic, lc);
                hit[q] = make_int2(__float_as_int(h.t), h.leaf);
            }
        }
    }
    if (COUNT) {
        ic = wave_sum(ic); lc = wave_sum(lc);
        if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[RTX_STAT_INNER(countIdx)], (unsigned long long)ic); atomicAdd(&stats[RTX_STAT_LEAF(countIdx)], (unsigned long long)lc); }
    }
}

// Culled-mode trace launch with lane refill. Bounce rays are incoherent: in a wave of 64 the
// traversal lengths differ by several x, and with one ray per lane per batch the wave runs as long
// as its slowest ray. Here a lane that finishes its ray goes idle, and as soon as RTX_REFILL_MIN
// lanes are idle (or the whole wave is) they pull the next rays of the queue — ballot + prefix
// popcount over the idle mask, one device atomic per RTX_RESERVE rays — so the traversal loop keeps
// running on well-filled waves. Per-ray results do not depend on which lane traced them.
// RTX_REFILL_MIN = 32 measured best (8: +5 %, 16: +3 %, 24: +1.5 %, 40: +5 %, 48: +9 % on the bounce-1
// launch): a refill stalls the wave for a memory round trip; running one pass between the issue of
// the new rays' loads and their first use was tried and cost more (registers) than it hid.
#ifndef RTX_REFILL_MIN
#define RTX_REFILL_MIN 32
#endif
#ifndef RTX_RESERVE
#define RTX_RESERVE 64u
#endif
#define RTX_IDLE ((int)0x80000002)
#ifndef RTX_RAYS_PER_LANE
#define RTX_RAYS_PER_LANE 2u
#endif

// BOUNCE (= the queue the launch reads) is a template parameter only so that profilers list the launches of
// the three bounces — whose durations differ by 4x — under three kernel names.
template <bool COUNT, int BOUNCE, bool BIG>
__global__ __launch_bounds__(RTX_BLOCK) void k_trace_refill(DevScene sc, const float4* __restrict__ rqO, const float4* __restrict__ rqD,
                                                            int2* __restrict__ hit,
                                                            uint32_t* __restrict__ ctrl, unsigned long long* __restrict__ stats,
                                                            uint32_t* __restrict__ slowList, uint32_t fixedCount, int workIdx) {
    const int countIdx = BOUNCE;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long laneLt = (1ull << lane) - 1ull;
    const long long nthreads = (long long)gridDim.x * RTX_BLOCK;
    const long long gtid = (long long)blockIdx.x * RTX_BLOCK + threadIdx.x;
    LaneStack st;
    st.lds = (int)threadIdx.x; st.stride = RTX_BLOCK; st.spill = sc.spill + gtid; st.spillStride = nthreads; st.sp = 0; st.base = 0;
    const float4* __restrict__ nodes = sc.fast;
    const int root = sc.fastRoot;
    const uint32_t count = countIdx == 0 ? fixedCount : ctrl[RTX_CTRL_COUNT(countIdx)];
    // A short queue is traced by fewer blocks: once the queue is drained every lane finishes the ray
    // it holds without refill, so with one or two rays per lane most of the launch would run as that
    // thinning tail. Blocks beyond count / (RTX_RAYS_PER_LANE * block size) leave at once.
    if ((unsigned long long)blockIdx.x * (RTX_BLOCK * RTX_RAYS_PER_LANE) >= (unsigned long long)count && blockIdx.x > 0) return;
    uint32_t ic = 0, lc = 0, wp = 0;
    uint32_t rcur = 0, rend = 0;          // wave-uniform: the wave's reserved run of the queue
    const uint32_t segLen = segment_length(count);
    uint32_t seg = blockIdx.x & 7u, segsTried = 0;
    bool drained = false;                 // wave-uniform
    // per-lane ray state
    int cur = RTX_IDLE;
    uint32_t q = 0;
    float dx = 0, dy = 0, dz = 0;
    WideRay w = wide_ray(0, 0, 0, 1, 1, 1);      // carries the origin too
    Hit best; best.t = INFINITY; best.leaf = -1; best.box = -1;
    float limit = INFINITY;
    // Two nested loops, both wave-uniform: the outer one refills idle lanes, the inner one runs traversal
    // passes until enough lanes are idle again. Inside the inner loop the ray (origin, direction,
    // reciprocals, queue index) is loop-invariant, which keeps it in place in registers.
    for (;;) {
        const unsigned long long idle = __ballot(cur == RTX_IDLE);
        int nIdle = __popcll(idle);
        if (!drained && nIdle >= RTX_REFILL_MIN) {
            while (rcur >= rend && !drained) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&ctrl[RTX_CTRL_XWORK(workIdx, seg)], RTX_RESERVE);
                const uint32_t segStart = seg * segLen, segEnd = segStart + segLen < count ? segStart + segLen : count;
                rcur = segStart + __builtin_amdgcn_readfirstlane(base);
                rend = rcur + RTX_RESERVE < segEnd ? rcur + RTX_RESERVE : segEnd;
                if (rcur >= segEnd || segStart >= count) {
                    // this segment is handed out: help with the next one
                    rend = rcur;
                    seg = (seg + 1u) & 7u;
                    if (++segsTried == 8u) drained = true;
                }
            }
            if (!drained) {
                const uint32_t avail = rend - rcur;
                const uint32_t take = (uint32_t)nIdle < avail ? (uint32_t)nIdle : avail;
                const uint32_t rank = (uint32_t)__popcll(idle & laneLt);
                if (cur == RTX_IDLE && rank < take) {
                    q = rcur + rank;
                    const float4 o = rqO[q], d = rqD[q];
                    dx = d.x; dy = d.y; dz = d.z;
                    if (d.w != 0.0f) {
                        hit[q] = make_int2(0, RTX_INVALID_SLOT);     // slot outside the image, or a hole in a bounce queue
                    } else if (lane_is_clean(o.x, o.y, o.z, dx, dy, dz)) {
                        w = wide_ray(o.x, o.y, o.z, 1.0f / dx, 1.0f / dy, 1.0f / dz);
                        best.t = INFINITY; best.leaf = -1; best.box = -1; limit = INFINITY;
                        st.sp = 0; st.base = 0;
                        cur = root;
                    } else {
                        // axis-parallel / non-finite ray: k_trace_slow walks the reference topology for it
                        slowList[atomicAdd(&ctrl[RTX_CTRL_SLOW(countIdx)], 1u)] = q;
                    }
                }
                rcur += take;
            }
            continue;
        }
        if (nIdle == 64) break;          // drained and nothing in flight
        do {
            if (COUNT) wp++;
            if (cur != RTX_IDLE) {
                if (cur >= 0) {
                    if (COUNT) ic++;
                    cur = wide_step<false, BIG>(nodes, cur, w, limit, st);
                }
                // not "else": a lane whose step ends at a leaf tests it in this same pass — the leaf
                // section runs in practically every pass anyway (some lane of the 64 is at a leaf)
                if (cur < 0 && cur != RTX_NEED_POP) {
                    int leafIdx = ~cur, box;
                    float t = intersect_leaf(sc.leaf, leafIdx, w.a.x, w.a.y, w.b.x, dx, dy, dz, box);
                    if (COUNT) lc++;
                    if (t > 0.01f && (t < best.t || (t == best.t && best.leaf >= 0 && box > best.box))) {
                        best.t = t; best.leaf = leafIdx; best.box = box;
                        limit = t + t * sc.cullSlack;
                    }
                    cur = RTX_NEED_POP;
                }
                // the pop closes the pass (entries beyond the current hit are dropped in a loop, they
                // cost no pass of their own), so a lane whose stack runs empty goes idle — and counts
                // towards the next refill — one pass earlier than if it popped at the start of the next
                if (cur == RTX_NEED_POP) {
                    bool finished = true;
                    while (st.sp > 0) {
                        int ref; float tn;
                        st.pop(ref, tn);
                        if (!(tn > limit)) { cur = ref; finished = false; break; }
                    }
                    if (finished) {
                        hit[q] = make_int2(__float_as_int(best.t), best.leaf);
                        cur = RTX_IDLE;
                    }
                }
            }
            nIdle = __popcll(__ballot(cur == RTX_IDLE));
        } while (drained ? nIdle < 64 : nIdle < RTX_REFILL_MIN);
    }
    if (COUNT) {
        ic = wave_sum(ic); lc = wave_sum(lc);
        if (lane == 0) {
            atomicAdd(&stats[RTX_STAT_INNER(countIdx)], (unsigned long long)ic); atomicAdd(&stats[RTX_STAT_LEAF(countIdx)], (unsigned long long)lc);
            atomicAdd(&stats[RTX_STAT_PASSES(countIdx)], (unsigned long long)wp);
        }
    }
}

// RTX_TRAVERSE_WIDE's bounce launch: the lane-refill loop of k_trace_refill around the unordered two-list pass of
// rtx_traverse.h (q_pass). Same queue protocol, same outputs.
#if defined(RTX_TRACEQ_WAVES)
#define RTX_TRACEQ_ATTR __attribute__((amdgpu_waves_per_eu(RTX_TRACEQ_WAVES, RTX_TRACEQ_WAVES)))
#else
#define RTX_TRACEQ_ATTR
#endif
template <bool COUNT, int BOUNCE, bool BIG>
__global__ __launch_bounds__(RTX_BLOCK) RTX_TRACEQ_ATTR void k_trace_q(DevScene sc, const float4* __restrict__ rqO, const float4* __restrict__ rqD,
                                                       int2* __restrict__ hit,
                                                       uint32_t* __restrict__ ctrl, unsigned long long* __restrict__ stats,
                                                       uint32_t* __restrict__ slowList, uint32_t fixedCount, int workIdx) {
    const int countIdx = BOUNCE;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long laneLt = (1ull << lane) - 1ull;
    const long long nthreads = (long long)gridDim.x * RTX_BLOCK;
    const long long gtid = (long long)blockIdx.x * RTX_BLOCK + threadIdx.x;
    QStack st;
    st.lds = (int)threadIdx.x; st.stride = RTX_BLOCK; st.spill = (int*)sc.spill + gtid; st.spillStride = nthreads; st.nsp = 0; st.nbase = 0; st.lsp = 0;
    const uint4* __restrict__ nodes = sc.qnodes;
    const float4* __restrict__ leaves = sc.leaf;
    const uint32_t count = countIdx == 0 ? fixedCount : ctrl[RTX_CTRL_COUNT(countIdx)];
    if ((unsigned long long)blockIdx.x * (RTX_BLOCK * RTX_RAYS_PER_LANE) >= (unsigned long long)count && blockIdx.x > 0) return;
    uint32_t ic = 0, lc = 0, wp = 0;
    uint32_t rcur = 0, rend = 0;          // wave-uniform: the wave's reserved run of the queue
    const uint32_t segLen = segment_length(count);
    uint32_t seg = blockIdx.x & 7u, segsTried = 0;
    bool drained = false;                 // wave-uniform
    // per-lane ray state
    bool busy = false;
    uint32_t q = 0;
    float dx = 0, dy = 0, dz = 0;
    QRay r = q_ray(0, 0, 0, 1, 1, 1);
    int nodeCur = 0, leafCur = 0;
    Hit best; best.t = INFINITY; best.leaf = -1; best.box = -1;
    for (;;) {
        const unsigned long long idle = __ballot(!busy);
        int nIdle = __popcll(idle);
        if (!drained && nIdle >= RTX_REFILL_MIN) {
            while (rcur >= rend && !drained) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&ctrl[RTX_CTRL_XWORK(workIdx, seg)], RTX_RESERVE);
                const uint32_t segStart = seg * segLen, segEnd = segStart + segLen < count ? segStart + segLen : count;
                rcur = segStart + __builtin_amdgcn_readfirstlane(base);
                rend = rcur + RTX_RESERVE < segEnd ? rcur + RTX_RESERVE : segEnd;
                if (rcur >= segEnd || segStart >= count) {
                    rend = rcur;
                    seg = (seg + 1u) & 7u;
                    if (++segsTried == 8u) drained = true;
                }
            }
            if (!drained) {
                // Straight-line refill: every lane loads a queue entry (a lane that takes none reads the run's first
                // one), both loads leave together, and the lane state is replaced by selects. Written as "if (this lane
                // takes a ray) { load origin; if (valid) { load direction; ... } }" the block cost two memory round
                // trips one after the other and about seventy register copies around its branches — a whole traversal
                // pass per refill, which is why refilling early (few idle lanes) used to lose.
                const uint32_t avail = rend - rcur;
                const uint32_t take = (uint32_t)nIdle < avail ? (uint32_t)nIdle : avail;
                const uint32_t rank = (uint32_t)__popcll(idle & laneLt);
                const bool mine = !busy && rank < take;
                const uint32_t qn = mine ? rcur + rank : rcur;
                const float4 o = rqO[qn], d = rqD[qn];
                const bool hole = d.w != 0.0f;             // slot outside the image, or a hole in a bounce queue
                const bool clean = q_lane_ok(o.x, o.y, o.z, d.x, d.y, d.z);
                const QRay nr = q_ray(o.x, o.y, o.z, 1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                const bool start = mine && !hole && clean;
                q = mine ? qn : q;
                dx = start ? d.x : dx; dy = start ? d.y : dy; dz = start ? d.z : dz;
                r.ox = start ? nr.ox : r.ox; r.oy = start ? nr.oy : r.oy; r.oz = start ? nr.oz : r.oz;
                r.ix = start ? nr.ix : r.ix; r.iy = start ? nr.iy : r.iy; r.iz = start ? nr.iz : r.iz;
                r.nx = start ? nr.nx : r.nx; r.ny = start ? nr.ny : r.ny; r.nz = start ? nr.nz : r.nz;
#if RTX_QFAST
                r.ex = start ? nr.ex : r.ex; r.ey = start ? nr.ey : r.ey; r.ez = start ? nr.ez : r.ez;
#endif
                best.t = start ? INFINITY : best.t; best.leaf = start ? -1 : best.leaf; best.box = start ? -1 : best.box;
                st.nsp = start ? 0 : st.nsp; st.nbase = start ? 0 : st.nbase; st.lsp = start ? 0 : st.lsp;
                nodeCur = start ? RTX_QENTRY(0, 1) : nodeCur; leafCur = start ? 0 : leafCur;
                busy = busy || start;
                if (mine && hole) hit[qn] = make_int2(0, RTX_INVALID_SLOT);
                // axis-parallel / non-finite ray: k_trace_slow walks the reference topology for it
                if (mine && !hole && !clean) slowList[atomicAdd(&ctrl[RTX_CTRL_SLOW(countIdx)], 1u)] = qn;
                rcur += take;
            }
            continue;
        }
        if (nIdle == 64) break;          // drained and nothing in flight
        do {
            if (COUNT) wp++;
            if (busy) {
                if (!q_pass<COUNT, BIG>(nodes, leaves, r, dx, dy, dz, nodeCur, leafCur, st, best, ic, lc)) {
                    hit[q] = make_int2(__float_as_int(best.t), best.leaf);
                    busy = false;
                }
            }
            nIdle = __popcll(__ballot(!busy));
        } while (drained ? nIdle < 64 : nIdle < RTX_REFILL_MIN);
    }
    if (COUNT) {
        ic = wave_sum(ic); lc = wave_sum(lc);
        if (lane == 0) {
            atomicAdd(&stats[RTX_STAT_INNER(countIdx)], (unsigned long long)ic); atomicAdd(&stats[RTX_STAT_LEAF(countIdx)], (unsigned long long)lc);
            atomicAdd(&stats[RTX_STAT_PASSES(countIdx)], (unsigned long long)wp);
        }
    }
}

// k_trace_q with the leaf tests on compacted lanes (rtx_traverse.h: QWave). A lane's pass is a node step only; the leaves
// a step finds go to the wave's queue and are tested 64 at a time by whichever lanes. A ray whose nodes have run out keeps
// its lane (origin, direction and queue index stay in its registers) until the queue has been emptied at the next refill;
// its result is then read from the wave's LDS. Same queue protocol and outputs as k_trace_q.
// LDS: RTX_TRACEQW_LDS_BYTES.
template <bool COUNT, int BOUNCE, bool BIG>
__global__ __launch_bounds__(RTX_BLOCK) RTX_TRACEQ_ATTR void k_trace_qw(DevScene sc, const float4* __restrict__ rqO, const float4* __restrict__ rqD,
                                                        int2* __restrict__ hit,
                                                        uint32_t* __restrict__ ctrl, unsigned long long* __restrict__ stats,
                                                        uint32_t* __restrict__ slowList, uint32_t fixedCount, int workIdx) {
    const int countIdx = BOUNCE;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long laneLt = (1ull << lane) - 1ull;
    const long long nthreads = (long long)gridDim.x * RTX_BLOCK;
    const long long gtid = (long long)blockIdx.x * RTX_BLOCK + threadIdx.x;
    QStack st;
    st.lds = (int)threadIdx.x; st.stride = RTX_BLOCK; st.spill = (int*)sc.spill + gtid; st.spillStride = nthreads; st.nsp = 0; st.nbase = 0; st.lsp = 0;
    QWave wq;
    wq.init(RTX_QN * RTX_BLOCK + (int)(threadIdx.x >> 6) * RTX_WQ_WORDS);
    unsigned long long* const myKey = (unsigned long long*)&rtx_lds[wq.key + 2 * (int)lane];
    const uint4* __restrict__ nodes = sc.qnodes;
    const float4* __restrict__ leaves = sc.leaf;
    const uint32_t count = countIdx == 0 ? fixedCount : ctrl[RTX_CTRL_COUNT(countIdx)];
    if ((unsigned long long)blockIdx.x * (RTX_BLOCK * RTX_RAYS_PER_LANE) >= (unsigned long long)count && blockIdx.x > 0) return;
    uint32_t ic = 0, lc = 0, wp = 0;
    uint32_t rcur = 0, rend = 0;          // wave-uniform: the wave's reserved run of the queue
    const uint32_t segLen = segment_length(count);
    uint32_t seg = blockIdx.x & 7u, segsTried = 0;
    bool drained = false;                 // wave-uniform
    // per-lane ray state
    bool busy = false;                    // has nodes to visit
    bool owing = false;                   // holds a ray whose result is not written yet
    uint32_t q = 0;
    float dx = 0, dy = 0, dz = 0;
    QRay r = q_ray(0, 0, 0, 1, 1, 1);
    int nodeCur = 0;
    for (;;) {
        // the leaves still queued, then the results of the rays that only waited for them
        while (wq.tail != wq.head) {
            const uint32_t left = wq.tail - wq.head;
            q_wave_serve<COUNT>(leaves, wq, left < 64u ? left : 64u, lane, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, dx, dy, dz, lc);
        }
        if (owing && !busy) {
            const unsigned long long key = __hip_atomic_load(myKey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            q_st2(hit + q, make_int2((int)(uint32_t)(key >> 32), rtx_lds[wq.leaf + (int)lane]));
            owing = false;
        }
        const unsigned long long idle = __ballot(!busy);
        int nIdle = __popcll(idle);
        if (!drained && nIdle >= RTX_REFILL_MIN) {
            while (rcur >= rend && !drained) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&ctrl[RTX_CTRL_XWORK(workIdx, seg)], RTX_RESERVE);
                const uint32_t segStart = seg * segLen, segEnd = segStart + segLen < count ? segStart + segLen : count;
                rcur = segStart + __builtin_amdgcn_readfirstlane(base);
                rend = rcur + RTX_RESERVE < segEnd ? rcur + RTX_RESERVE : segEnd;
                if (rcur >= segEnd || segStart >= count) {
                    rend = rcur;
                    seg = (seg + 1u) & 7u;
                    if (++segsTried == 8u) drained = true;
                }
            }
            if (!drained) {
                // straight-line refill, as in k_trace_q
                const uint32_t avail = rend - rcur;
                const uint32_t take = (uint32_t)nIdle < avail ? (uint32_t)nIdle : avail;
                const uint32_t rank = (uint32_t)__popcll(idle & laneLt);
                const bool mine = !busy && rank < take;
                const uint32_t qn = mine ? rcur + rank : rcur;
                const float4 o = q_ld4(rqO + qn), d = q_ld4(rqD + qn);
                const bool hole = d.w != 0.0f;
                const bool clean = q_lane_ok(o.x, o.y, o.z, d.x, d.y, d.z);
                const QRay nr = q_ray(o.x, o.y, o.z, 1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                const bool start = mine && !hole && clean;
                q = mine ? qn : q;
                dx = start ? d.x : dx; dy = start ? d.y : dy; dz = start ? d.z : dz;
                r.ox = start ? nr.ox : r.ox; r.oy = start ? nr.oy : r.oy; r.oz = start ? nr.oz : r.oz;
                r.ix = start ? nr.ix : r.ix; r.iy = start ? nr.iy : r.iy; r.iz = start ? nr.iz : r.iz;
                r.nx = start ? nr.nx : r.nx; r.ny = start ? nr.ny : r.ny; r.nz = start ? nr.nz : r.nz;
#if RTX_QFAST
                r.ex = start ? nr.ex : r.ex; r.ey = start ? nr.ey : r.ey; r.ez = start ? nr.ez : r.ez;
#endif
                st.nsp = start ? 0 : st.nsp; st.nbase = start ? 0 : st.nbase;
                nodeCur = start ? RTX_QENTRY(0, 1) : nodeCur;
                if (start) {
                    __hip_atomic_store(myKey, RTX_WQ_NOHIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    rtx_lds[wq.leaf + (int)lane] = -1;
                }
                busy = busy || start;
                owing = owing || start;
                if (mine && hole) hit[qn] = make_int2(0, RTX_INVALID_SLOT);
                if (mine && !hole && !clean) slowList[atomicAdd(&ctrl[RTX_CTRL_SLOW(countIdx)], 1u)] = qn;
                rcur += take;
            }
            continue;
        }
        if (nIdle == 64) break;          // drained and nothing in flight
        do {
            if (COUNT) wp++;
            // every lane loads a node (an idle one node 0: those reads merge), so that the loads sit in straight-line code
            const int nodeIdx = busy ? (nodeCur >> 4) + (__ffs(nodeCur & 15) - 1) : 0;
            const QNodeData nd = q_load_node<BIG>(nodes, nodeIdx);
            uint32_t ml = 0; int bl = 0;
            if (busy) {
                nodeCur &= nodeCur - 1;
                int bi; uint32_t leafBits;
                const uint32_t m = q_step(nd, r, bi, bl, leafBits);
                if (COUNT) ic++;
                const uint32_t mi = m & ~leafBits;
                ml = m & leafBits;
                if (mi) { if (nodeCur & 15) st.push_node(nodeCur); nodeCur = RTX_QENTRY(bi, mi); }
                if (!(nodeCur & 15) && st.nsp > 0) nodeCur = st.pop_node();
                busy = (nodeCur & 15) != 0;
            }
            q_wave_push(wq, lane, bl, ml);
            if (wq.tail - wq.head >= 64u)
                q_wave_serve<COUNT>(leaves, wq, 64u, lane, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, dx, dy, dz, lc);
            nIdle = __popcll(__ballot(!busy));
        } while (drained ? nIdle < 64 : nIdle < RTX_REFILL_MIN);
    }
    if (COUNT) {
        ic = wave_sum(ic); lc = wave_sum(lc);
        if (lane == 0) {
            atomicAdd(&stats[RTX_STAT_INNER(countIdx)], (unsigned long long)ic); atomicAdd(&stats[RTX_STAT_LEAF(countIdx)], (unsigned long long)lc);
            atomicAdd(&stats[RTX_STAT_PASSES(countIdx)], (unsigned long long)wp);
        }
    }
}

// The few rays whose slab arithmetic can produce NaN (a zero direction component: 0*inf) keep the
// reference's topology and the reference's exact semantics. Full-depth stack in LDS (small grid).
template <bool COUNT>
__global__ __launch_bounds__(RTX_BLOCK) void k_trace_slow(DevScene sc, const float4* __restrict__ rqO, const float4* __restrict__ rqD,
                                                          int2* __restrict__ hit,
                                                          const uint32_t* __restrict__ ctrl, unsigned long long* __restrict__ stats,
                                                          const uint32_t* __restrict__ slowList, int countIdx) {
    int* stack = rtx_lds + threadIdx.x;
    const uint32_t n = ctrl[RTX_CTRL_SLOW(countIdx)];
    uint32_t ic = 0, lc = 0;
    for (uint32_t i = blockIdx.x * RTX_BLOCK + threadIdx.x; i < n; i += gridDim.x * RTX_BLOCK) {
        const uint32_t q = slowList[i];
        const float4 o = rqO[q], d = rqD[q];
        Hit h = traverse_bvh<true, COUNT>(sc, o.x, o.y, o.z, d.x, d.y, d.z, stack, RTX_BLOCK, ic, lc);
        hit[q] = make_int2(__float_as_int(h.t), h.leaf);
    }
    if (COUNT) {
        ic = wave_sum(ic); lc = wave_sum(lc);
        if ((threadIdx.x & 63) == 0 && (ic | lc)) { atomicAdd(&stats[RTX_STAT_INNER(countIdx)], (unsigned long long)ic); atomicAdd(&stats[RTX_STAT_LEAF(countIdx)], (unsigned long long)lc); }
    }
}

#if defined(RTX_SHADE_WAVES)
#define RTX_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(RTX_SHADE_WAVES, RTX_SHADE_WAVES)))
#else
#define RTX_SHADE_ATTR
#endif
#ifndef RTX_SHADE_WINDOW
#define RTX_SHADE_WINDOW 512u     // queue entries a wave sorts into hits and misses at a time (LDS: 8 bytes each)
#endif
#define RTX_SHADE_LDS_BYTES ((RTX_BLOCK / 64) * 2 * RTX_SHADE_WINDOW * 4)
static_assert(RTX_SHADE_LDS_BYTES <= 65536 && RTX_SHADE_WINDOW % 64u == 0u, "a block's LDS is at most 64 KB; a window is whole runs of 64");
// src/ThreadedShader.cpp:155-180 for bounce `bounce` of every queued path.
__global__ __launch_bounds__(RTX_BLOCK) RTX_SHADE_ATTR void k_shade(DevScene sc, DevBatch bt, const int2* __restrict__ tiles, int bounce, int linearMode,
                                                     const float4* __restrict__ rqOin, const float4* __restrict__ rqDin,
                                                     const int2* __restrict__ hit,
                                                     float4* __restrict__ rqOout, float4* __restrict__ rqDout,
                                                     float4* __restrict__ lightArr, float4* __restrict__ accArr,
                                                     uint32_t* __restrict__ seeds,
                                                     uint32_t* __restrict__ ctrl, unsigned long long* __restrict__ stats, int workIdx, int perFrameStats) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nf = (uint32_t)bt.count, tilePix = (uint32_t)(bt.fr[0].tileW * bt.fr[0].tileH);
    const uint32_t count = bounce == 0 ? (uint32_t)bt.fr[0].numSlots * nf : ctrl[RTX_CTRL_COUNT(bounce)];
    uint32_t hits = 0;
    // perFrameStats (a set of at most RTX_AHEAD_MAX frames started ahead, which are asked for — and counted — one by one, rtx_api.hip take_ahead):
    // hits are counted per frame of the batch, frame f in stats[f * RTX_STAT_WORDS + ...]; wave-uniform counters
    uint32_t hitsOf[RTX_AHEAD_MAX];
    for (int f = 0; f < RTX_AHEAD_MAX; ++f) hitsOf[f] = 0;
    // Survivors go to the next queue in wave-private runs: one device atomic reserves `run` entries,
    // the wave fills them densely across iterations, and what is left of its last run when the kernel
    // ends is marked as holes (rqD.w = 1), which the trace kernels skip. One atomic per 64 paths on a
    // single counter was the bottleneck of this kernel (same-address atomics retire at ~8 ns each).
    const uint32_t run = append_run_length(count);
    uint32_t outCur = 0, outEnd = 0;      // wave-uniform
    WorkIter<2, false> work(count);
    // Hits and misses are shaded apart. A wave first sorts a window of RTX_SHADE_WINDOW queue entries into two lists (LDS:
    // queue positions, in queue order), then shades the hits 64 at a time and the misses 64 at a time. Shading a run of 64
    // as it comes costs the hit branch AND the sky branch whenever the run holds both — every run of bounces 1 to 3, where
    // a third of the paths hit something — with each branch's lanes a third or two thirds full.
    int* const hitList = rtx_lds + (threadIdx.x >> 6) * (2 * RTX_SHADE_WINDOW);
    int* const missList = hitList + RTX_SHADE_WINDOW;
    for (;;) {
        uint32_t nHit = 0, nMiss = 0;     // wave-uniform
        bool any = false;
        for (uint32_t it = 0; it < RTX_SHADE_WINDOW / 64u; ++it) {
            const uint32_t base = work.next(ctrl, workIdx);
            if (base == 0xffffffffu) break;
            any = true;
            const uint32_t q = base + lane;
            int leaf = RTX_INVALID_SLOT;
            if (q < count) leaf = hit[q].y;
            const bool isHit = leaf >= 0, isMiss = leaf < 0 && leaf != RTX_INVALID_SLOT;
            const unsigned long long bh = __ballot(isHit), bm = __ballot(isMiss);
            if (isHit) hitList[nHit + q_mbcnt(bh)] = (int)q;
            if (isMiss) missList[nMiss + q_mbcnt(bm)] = (int)q;
            nHit += (uint32_t)__popcll(bh); nMiss += (uint32_t)__popcll(bm);
        }
        if (!any) break;
        __builtin_amdgcn_wave_barrier();
        // ---- the hits: ComputeColor, survivors to the next queue ----
        for (uint32_t i0 = 0; i0 < nHit; i0 += 64u) {
            const bool active = i0 + lane < nHit;
            bool alive = false;
            F4 o = mk4(0, 0, 0), d = mk4(0, 0, 0), light = mk4(1, 1, 1, 1), acc = mk4(0, 0, 0, 0);
            uint32_t seed = 0; int slot = 0;
            uint32_t hitFrame = 0xffffffffu;
            if (active) {
                const uint32_t q = (uint32_t)hitList[i0 + lane];
                const int2 h = q_ld2(hit + q);
                const int leaf = h.y; const float hitDistance = __int_as_float(h.x);
                float4 o4 = q_ld4(rqOin + q), d4 = q_ld4(rqDin + q);
                slot = __float_as_int(o4.w);
                o = mk4(o4.x, o4.y, o4.z); d = mk4(d4.x, d4.y, d4.z);
                if (bounce > 0) { light = from4(lightArr[slot]); acc = from4(accArr[slot]); }
                hits++;
                seed = seeds[slot];
                HitPrim prim = load_prim(sc, leaf);
                float t = hitDistance;
                F4 dirForPoint = linearMode ? mul4(d, 1.0000001f) : d;   // src/ThreadedShader.cpp:217,232
                F4 point = add4(o, mul4(dirForPoint, t));
                F4 normal = hit_normal(prim, point);
                const uint32_t f = nf > 1 ? ((uint32_t)slot / tilePix) % nf : 0u;      // the path's frame: its camera
                hitFrame = f;
                const F4 camPos = mk4(bt.fr[f].pos[0], bt.fr[f].pos[1], bt.fr[f].pos[2]);
                F4 color = compute_color(sc, camPos, o, d, point, normal, prim, light, seed);
                light = clamp01_sse(light);
                acc = clamp01_sse(add4(acc, color));
                if (bounce == 3) {
                    // the path's colour waits in its slot for k_resolve: frames overlap on the device, and the
                    // running mean must be updated in frame order
                    accArr[slot] = make_float4(acc.x, acc.y, acc.z, acc.w);
                } else {
                    alive = true;
                }
            }
            if (perFrameStats) {
                for (uint32_t f = 0; f < (uint32_t)RTX_AHEAD_MAX; ++f) hitsOf[f] += (uint32_t)__popcll(__ballot(hitFrame == f));
            }
            // regroup survivors: dense positions in the next queue
            const unsigned long long m = __ballot(alive);
            if (m) {
                const uint32_t n = (uint32_t)__popcll(m), room = outEnd - outCur;
                uint32_t fresh = 0;
                if (n > room) {
                    if (lane == 0) fresh = atomicAdd(&ctrl[RTX_CTRL_COUNT(bounce + 1)], run);
                    fresh = __builtin_amdgcn_readfirstlane(fresh);
                }
                if (alive) {
                    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    const uint32_t pos = rank < room ? outCur + rank : fresh + (rank - room);
                    q_st4(rqOout + pos, make_float4(o.x, o.y, o.z, __int_as_float(slot)));
                    q_st4(rqDout + pos, make_float4(d.x, d.y, d.z, 0.0f));
                    lightArr[slot] = make_float4(light.x, light.y, light.z, light.w);
                    accArr[slot] = make_float4(acc.x, acc.y, acc.z, acc.w);
                    seeds[slot] = seed;
                }
                if (n > room) { outCur = fresh + (n - room); outEnd = fresh + run; }
                else outCur += n;
            }
        }
        // ---- the misses: the sky's colour ends the path (src/ThreadedShader.cpp:160-171) ----
        for (uint32_t i0 = 0; i0 < nMiss; i0 += 64u) {
            if (i0 + lane < nMiss) {
                const uint32_t q = (uint32_t)missList[i0 + lane];
                const float4 o4 = q_ld4(rqOin + q), d4 = q_ld4(rqDin + q);
                const int slot = __float_as_int(o4.w);
                const F4 d = mk4(d4.x, d4.y, d4.z);
                F4 light = mk4(1, 1, 1, 1), acc = mk4(0, 0, 0, 0);
                if (bounce > 0) { light = from4(lightArr[slot]); acc = from4(accArr[slot]); }
                const F4 texel = sky_texel(sc, d);
                acc = add4(acc, mul4(texel, light));          // no clamp on the sky term (src/ThreadedShader.cpp:169)
                accArr[slot] = make_float4(acc.x, acc.y, acc.z, acc.w);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t i = outCur + lane; i < outEnd; i += 64u) rqDout[i] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    if (perFrameStats) {
        if (lane == 0)
            for (int f = 0; f < RTX_AHEAD_MAX; ++f)
                if (hitsOf[f]) atomicAdd(&stats[f * RTX_STAT_WORDS + RTX_STAT_HITS(bounce)], (unsigned long long)hitsOf[f]);
        return;
    }
    hits = wave_sum(hits);
    if (lane == 0 && hits) atomicAdd(&stats[RTX_STAT_HITS(bounce)], (unsigned long long)hits);
}

// After the last shade of a frame: pixels = Lerp(pixels, colour, 1 / (frameCounter + 1)) for every path of the
// frame (src/ThreadedShader.cpp:179-180). A launch of its own because frames are pipelined (several in flight
// on separate streams) while the running mean has to take them in order: the resolve launches
// are chained by events, everything before them overlaps freely.
// [f0, f1): the frames of the batch to apply, in call order (a batch started ahead is applied one frame per call).
__global__ __launch_bounds__(RTX_BLOCK) void k_resolve(DevBatch bt, const int2* __restrict__ tiles, const float4* __restrict__ accArr,
                                                       float4* __restrict__ fb, int f0, int f1) {
    const DevFrame& fr0 = bt.fr[0];
    const uint32_t p = blockIdx.x * RTX_BLOCK + threadIdx.x;
    if (p >= (uint32_t)fr0.numSlots) return;
    int x, y;
    if (!slot_to_pixel(fr0, tiles, p, x, y)) return;
    const size_t pix = (size_t)y * (size_t)fr0.width + (size_t)x;
    const uint32_t nf = (uint32_t)bt.count, tilePix = (uint32_t)(fr0.tileW * fr0.tileH);
    const uint32_t tile = p / tilePix, within = p - tile * tilePix;
    F4 res = from4(fb[pix]);
    for (uint32_t f = (uint32_t)f0; f < (uint32_t)f1; ++f)       // the frames of a batch in call order
        res = lerp_color(res, from4(accArr[(size_t)(tile * nf + f) * tilePix + within]), bt.fr[f].lerpScale);
    fb[pix] = make_float4(res.x, res.y, res.z, res.w);
}

// counters of a look-ahead frame into the renderer's (rtx_api.hip take_ahead); other streams add to dst at the same time
__global__ void k_add_stats(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, int n) {
    const int i = (int)threadIdx.x;
    if (i < n && src[i]) atomicAdd(&dst[i], src[i]);
}

// ---- probes -------------------------------------------------------------------
__global__ void k_hits_to_ids(DevScene sc, DevFrame fr, const int2* __restrict__ tiles, const int2* __restrict__ hit, int* __restrict__ ids) {
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (uint32_t)fr.numSlots) return;
    int x, y;
    if (!slot_to_pixel(fr, tiles, slot, x, y)) return;
    int leaf = hit[slot].y;
    int id = -1;
    if (leaf >= 0) id = __float_as_int(sc.shade[(size_t)leaf * 8 + 2].y);
    ids[(size_t)y * fr.width + x] = id;
}

__global__ void k_probe_finish(DevScene sc, int linearMode, const float4* __restrict__ rqO, const float4* __restrict__ rqD,
                               const int2* __restrict__ hit, uint32_t n,
                               int* __restrict__ ids, float4* __restrict__ points, float4* __restrict__ normals) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int2 h = hit[q];
    int leaf = h.y;
    int id = -1; F4 point = mk4(0, 0, 0), normal = mk4(0, 0, 0);
    if (leaf >= 0) {
        HitPrim prim = load_prim(sc, leaf);
        id = __float_as_int(sc.shade[(size_t)leaf * 8 + 2].y);
        float4 o4 = rqO[q], d4 = rqD[q];
        F4 o = mk4(o4.x, o4.y, o4.z), d = mk4(d4.x, d4.y, d4.z);
        F4 dirForPoint = linearMode ? mul4(d, 1.0000001f) : d;
        point = add4(o, mul4(dirForPoint, __int_as_float(h.x)));
        normal = hit_normal(prim, point);
    }
    if (ids) ids[q] = id;
    if (points) points[q] = make_float4(point.x, point.y, point.z, 0.0f);
    if (normals) normals[q] = make_float4(normal.x, normal.y, normal.z, 0.0f);
}

// rtx_trace_rays_noalloc: the caller's directions (4th lane "ignored") -> queue entries (4th lane 0 = a ray)
__global__ void k_stage_directions(const float4* __restrict__ in, float4* __restrict__ out, uint32_t n) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const float4 d = in[q];
    out[q] = make_float4(d.x, d.y, d.z, 0.0f);
}

// ---- device-side flatten -------------------------------------------------------------
// rtx_create uploads the caller's objects (256 B) and boxes (64 B) as they are and builds the traversal records here:
// the host only decides WHERE every leaf goes (the own tree's leaf order) and its tie-break rank. Same arithmetic as
// the host would do — e1 = B - A, e2 = C - A are the reference's own first two subtractions (src/ThreadedShader.cpp:298-299).
// Object words: [0] = (type radius - -) [1] = position [2..4] = normals [5..7] = vertices [11].x = materialID.
// Box words: [0] = (objectID parentID leftID rightID) [1] = min [2] = max.
__global__ void k_emit_leaf_records(const float4* __restrict__ objects, const float4* __restrict__ boxes, const int2* __restrict__ plan,
                                    uint32_t n, float4* __restrict__ leaf, float4* __restrict__ shade) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int2 pl = plan[p];                                     // (box index, tie-break rank)
    const float4* B = boxes + (size_t)pl.x * 4;
    const float4 b0 = B[0], bmin = B[1], bmax = B[2];
    const int objectID = __float_as_int(b0.x);
    const float4* O = objects + (size_t)objectID * 16;
    const float4 o0 = O[0], pos = O[1], n0 = O[2], n1 = O[3], n2 = O[4], vA = O[5], vB = O[6], vC = O[7];
    const int type = __float_as_int(o0.x);
    const int materialID = __float_as_int(O[11].x);
    float4 q0, q1, q3; float e2z;
    if (type == 4 /* RTX_TRIANGLE */) {
        const float e1x = vB.x - vA.x, e1y = vB.y - vA.y, e1z = vB.z - vA.z;
        const float e2x = vC.x - vA.x, e2y = vC.y - vA.y; e2z = vC.z - vA.z;
        q0 = make_float4(vA.x, vA.y, vA.z, e1x);
        q1 = make_float4(e1y, e1z, e2x, e2y);
        q3 = make_float4(vB.x, vB.y, vC.x, vC.y);
    } else {
        q0 = make_float4(pos.x, pos.y, pos.z, o0.y);
        q1 = make_float4(0, 0, 0, 0); q3 = make_float4(0, 0, 0, 0); e2z = 0.0f;
    }
    // every type but TRIANGLE is intersected as a sphere (src/ThreadedShader.cpp:364-368)
    const int tag = (int)(((uint32_t)pl.y << 3) | (type == 4 ? 4u : 0u));
    float4* L = leaf + (size_t)p * 4;
    L[0] = q0; L[1] = q1;
    L[2] = make_float4(e2z, __int_as_float(tag), bmin.x, bmin.y);
    L[3] = make_float4(bmin.z, bmax.x, bmax.y, bmax.z);
    float4* S = shade + (size_t)p * 8;
    S[0] = q0; S[1] = q1;
    S[2] = make_float4(e2z, __int_as_float(objectID), __int_as_float(pl.x), __int_as_float(type));
    S[3] = q3;
    S[4] = make_float4(n0.x, n0.y, n0.z, n1.x);
    S[5] = make_float4(n1.y, n1.z, n2.x, n2.y);
    S[6] = make_float4(n2.z, __int_as_float(materialID), 0, 0);
    S[7] = make_float4(0, 0, 0, 0);
}

// The reference's own topology as 64-byte records: both children's boxes + both child refs (>= 0 inner record,
// < 0 ~leaf position, RTX_NO_CHILD for an absent child).
__global__ void k_emit_inner_records(const float4* __restrict__ boxes, const int* __restrict__ slotOf, const int* __restrict__ posOf,
                                     uint32_t nBox, float4* __restrict__ inner) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBox) return;
    const float4 b0 = boxes[(size_t)b * 4];
    if (__float_as_int(b0.x) != -1) return;                      // a leaf box
    const int child[2] = {__float_as_int(b0.z), __float_as_int(b0.w)};
    float4 lo[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)}, hi[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    int ref[2] = {RTX_NO_CHILD, RTX_NO_CHILD};
    for (int k = 0; k < 2; ++k) {
        if (child[k] < 0 || (uint32_t)child[k] >= nBox) continue;
        const float4* C = boxes + (size_t)child[k] * 4;
        const float4 c0 = C[0];
        lo[k] = C[1]; hi[k] = C[2];
        ref[k] = __float_as_int(c0.x) != -1 ? ~posOf[slotOf[child[k]]] : slotOf[child[k]];
    }
    float4* N = inner + (size_t)slotOf[b] * 4;
    N[0] = make_float4(lo[0].x, lo[0].y, lo[0].z, hi[0].x);
    N[1] = make_float4(hi[0].y, hi[0].z, lo[1].x, lo[1].y);
    N[2] = make_float4(lo[1].z, hi[1].x, hi[1].y, hi[1].z);
    N[3] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), 0, 0);
}

// ---- present ---------------------------------------------------------------------
// The running mean as the 8-bit RGBA image a display path consumes (the reference uploads its float pixels into a
// GL_RGBA texture, src/WindowManager.cpp:18,165, i.e. GL clamps to [0,1] and rounds to 8 bits; here that conversion
// — with an optional display gamma, src/RenderingContext.h:48 — runs on the device and 4 bytes per pixel travel).
__global__ void k_present_rgba8(const float4* __restrict__ fb, uint32_t* __restrict__ out, uint32_t n, float invGamma) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = fb[i];
    float v[4] = {c.x, c.y, c.z, c.w};
    uint32_t word = 0;
    for (int k = 0; k < 4; ++k) {
        float x = v[k];
        x = x > 0.0f ? (x < 1.0f ? x : 1.0f) : 0.0f;                  // NaN -> 0
        if (invGamma > 0.0f && k < 3) x = powf(x, invGamma);
        word |= (uint32_t)(x * 255.0f + 0.5f) << (8 * k);             // GL's float -> unorm8: round(x * 255)
    }
    out[i] = word;
}

// ---- multi-GPU tile plumbing ----------------------------------------------------
// packed layout: tile k of the rank at k*tileW*tileH float4, row-major inside the tile
__global__ void k_pack_tiles(DevFrame fr, const int2* __restrict__ tiles, int numTiles, const float4* __restrict__ fb, float4* __restrict__ packed) {
    const uint32_t perTile = (uint32_t)(fr.tileW * fr.tileH);
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)numTiles * perTile) return;
    uint32_t tile = (uint32_t)(i / perTile), within = (uint32_t)(i % perTile);
    int x = tiles[tile].x + (int)(within % (uint32_t)fr.tileW), y = tiles[tile].y + (int)(within / (uint32_t)fr.tileW);
    float4 v = make_float4(0, 0, 0, 0);
    if (x < fr.width && y < fr.height) v = fb[(size_t)y * fr.width + x];
    packed[i] = v;
}
__global__ void k_unpack_tiles(DevFrame fr, const int2* __restrict__ tiles, int numTiles, const float4* __restrict__ packed, float4* __restrict__ fb) {
    const uint32_t perTile = (uint32_t)(fr.tileW * fr.tileH);
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)numTiles * perTile) return;
    uint32_t tile = (uint32_t)(i / perTile), within = (uint32_t)(i % perTile);
    int x = tiles[tile].x + (int)(within % (uint32_t)fr.tileW), y = tiles[tile].y + (int)(within / (uint32_t)fr.tileW);
    if (x < fr.width && y < fr.height) fb[(size_t)y * fr.width + x] = packed[i];
}
