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
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>
#if defined(__SSE__)
#include <xmmintrin.h>
#endif

namespace rtxbuild {

// std::vector whose resize / sized constructor leaves trivially constructible elements untouched: the big arrays of
// the build are written in full by the threads that fill them, and a value-initialising vector would first walk
// (and page in) every one of them on the calling thread — a quarter of the build time at 10 M primitives.
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    template <class U> void construct(U*) {}                                             // value-init request: do nothing
    template <class U, class A0, class... A> void construct(U* p, A0&& a0, A&&... a) { ::new ((void*)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
};
template <class T> using Raw = std::vector<T, NoInitAlloc<T>>;

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

// One binning pass per node covers the three axes at once, and the partition sweep collects both children's
// centroid bounds on the way: two sweeps over a node's primitives per level (bin, partition) instead of five. Nodes above
// 128 K primitives run both sweeps on all the threads they own (thread-local bins; partition through a
// scratch array with per-chunk offsets), smaller subtrees are built one thread each.
struct Item { float lo[3]; uint32_t id; float hi[3]; uint32_t pad;
              Box box() const { Box b; for (int a = 0; a < 3; ++a) { b.lo[a] = lo[a]; b.hi[a] = hi[a]; } return b; } };     // 32 bytes: a primitive's box travels with its index, so that
                                                          // every sweep over a node's primitives is a linear read
struct SahBuilder {
    Item* items;            // partitioned in place; final order = leaf order of the new tree
    Node* nodes;            // n-1 inner records
    Item* scratch;          // n entries, for the parallel partition

#ifndef RTX_SAH_BINS
#define RTX_SAH_BINS 16
#endif
#ifndef RTX_SAH_SWEEP
#define RTX_SAH_SWEEP 16
#endif
    static const int kBins = RTX_SAH_BINS;
    static const uint32_t kParallelNode = 1u << 17;
    static const uint32_t kSweepNode = RTX_SAH_SWEEP;

    // a bin's box as two 4-float rows (lo.xyz _, hi.xyz _), the layout of an Item, so that growing it by an
    // item is two packed min/max; the 4th lanes carry the item's id / padding bits and are never read
    struct alignas(16) Bins {
        float bb[3][kBins][8]; uint32_t cnt[3][kBins];
        void clear() {
            for (int a = 0; a < 3; ++a) for (int k = 0; k < kBins; ++k) {
                for (int i = 0; i < 4; ++i) { bb[a][k][i] = INFINITY; bb[a][k][4 + i] = -INFINITY; }
                cnt[a][k] = 0;
            }
        }
        void merge(const Bins& o) {
            for (int a = 0; a < 3; ++a) for (int k = 0; k < kBins; ++k) {
                for (int i = 0; i < 3; ++i) { bb[a][k][i] = std::min(bb[a][k][i], o.bb[a][k][i]); bb[a][k][4 + i] = std::max(bb[a][k][4 + i], o.bb[a][k][4 + i]); }
                cnt[a][k] += o.cnt[a][k];
            }
        }
        Box box(int a, int k) const { Box b; for (int i = 0; i < 3; ++i) { b.lo[i] = bb[a][k][i]; b.hi[i] = bb[a][k][4 + i]; } return b; }
        inline void add(int a, int k, const float* item8) {
#if defined(__SSE__)
            _mm_store_ps(bb[a][k], _mm_min_ps(_mm_load_ps(bb[a][k]), _mm_loadu_ps(item8)));
            _mm_store_ps(bb[a][k] + 4, _mm_max_ps(_mm_load_ps(bb[a][k] + 4), _mm_loadu_ps(item8 + 4)));
#else
            for (int i = 0; i < 3; ++i) { bb[a][k][i] = std::min(bb[a][k][i], item8[i]); bb[a][k][4 + i] = std::max(bb[a][k][4 + i], item8[4 + i]); }
#endif
            cnt[a][k]++;
        }
    };
    struct Split { float lo[3], scale[3]; bool valid[3]; };

    static inline void centroid(const Box& b, float c[3]) { for (int a = 0; a < 3; ++a) c[a] = 0.5f * b.lo[a] + 0.5f * b.hi[a]; }
    static inline void grow_centroid(Box& cb, const Item& b) {
        for (int a = 0; a < 3; ++a) { const float c = 0.5f * b.lo[a] + 0.5f * b.hi[a]; cb.lo[a] = c < cb.lo[a] ? c : cb.lo[a]; cb.hi[a] = c > cb.hi[a] ? c : cb.hi[a]; }
    }
    static inline int bin_of(float c, float lo, float scale) {
        int k = (int)((c - lo) * scale);
        return k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
    }
    void bin_range(uint32_t begin, uint32_t end, const Split& sp, Bins& bins) const {
        for (uint32_t i = begin; i < end; ++i) {
            const Item& it = items[i];
            for (int a = 0; a < 3; ++a) {
                if (!sp.valid[a]) continue;
                bins.add(a, bin_of(0.5f * it.lo[a] + 0.5f * it.hi[a], sp.lo[a], sp.scale[a]), it.lo);
            }
        }
    }
    template <class F>
    static void parallel_chunks(uint32_t begin, uint32_t end, int threads, F&& f) {
        const uint32_t n = end - begin, per = (n + (uint32_t)threads - 1) / (uint32_t)threads;
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) {
            const uint32_t b = std::min(end, begin + (uint32_t)t * per), e = std::min(end, b + per);
            pool.emplace_back([&f, t, b, e]() { f(t, b, e); });
        }
        f(0, begin, std::min(end, begin + per));
        for (auto& th : pool) th.join();
    }

    void write_node(int32_t nodeIdx, const Box& bl, const Box& br, int32_t rl, int32_t rr) {
        float* q = nodes[nodeIdx].q;
        q[0] = bl.lo[0]; q[1] = bl.lo[1]; q[2] = bl.lo[2]; q[3] = bl.hi[0]; q[4] = bl.hi[1]; q[5] = bl.hi[2];
        q[6] = br.lo[0]; q[7] = br.lo[1]; q[8] = br.lo[2]; q[9] = br.hi[0]; q[10] = br.hi[1]; q[11] = br.hi[2];
        q[12] = i2f(rl); q[13] = i2f(rr); q[14] = 0; q[15] = 0;
    }

    // Subtrees of at most kSweepNode primitives — 15 of every 16 nodes of the tree — are split by an exact
    // sweep: per axis, sort by centroid, evaluate all n-1 split positions. Cheaper than clearing and
    // evaluating 3 x 16 bins for a handful of primitives, and the best split there is.
    int32_t build_small(uint32_t begin, uint32_t end, int32_t nodeIdx, int depth, Box& outBox, int& outDepth) {
        const uint32_t n = end - begin;
        if (n == 1) { outBox = items[begin].box(); outDepth = depth; return ~(int32_t)begin; }
        uint8_t idx[3][kSweepNode]; int bestAxis = -1; uint32_t bestPos = 0; double bestCost = INFINITY;
        for (int a = 0; a < 3; ++a) {
            float key[kSweepNode];
            for (uint32_t i = 0; i < n; ++i) {                         // insertion sort by centroid
                const float c = 0.5f * items[begin + i].lo[a] + 0.5f * items[begin + i].hi[a];
                uint32_t j = i;
                while (j > 0 && key[j - 1] > c) { key[j] = key[j - 1]; idx[a][j] = idx[a][j - 1]; --j; }
                key[j] = c; idx[a][j] = (uint8_t)i;
            }
            double rightArea[kSweepNode];
            Box acc = empty_box();
            for (uint32_t i = n - 1; i > 0; --i) { grow(acc, items[begin + idx[a][i]].box()); rightArea[i] = half_area(acc); }
            acc = empty_box();
            for (uint32_t i = 0; i + 1 < n; ++i) {
                grow(acc, items[begin + idx[a][i]].box());
                const double cost = half_area(acc) * (i + 1) + rightArea[i + 1] * (n - i - 1);
                // equal costs (identical boxes): the split nearest the middle, not a chain
                const uint32_t off = i + 1 > n / 2 ? i + 1 - n / 2 : n / 2 - (i + 1), bestOff = bestPos > n / 2 ? bestPos - n / 2 : n / 2 - bestPos;
                if (cost < bestCost || (cost == bestCost && off < bestOff)) { bestCost = cost; bestAxis = a; bestPos = i + 1; }
            }
        }
        if (bestAxis < 0) { bestAxis = 0; bestPos = n / 2; }           // non-finite areas: split by count
        Item tmp[kSweepNode];
        for (uint32_t i = 0; i < n; ++i) tmp[i] = items[begin + idx[bestAxis][i]];
        for (uint32_t i = 0; i < n; ++i) items[begin + i] = tmp[i];
        const uint32_t mid = begin + bestPos;
        Box bl, br; int dl = 0, dr = 0;
        const int32_t rl = build_small(begin, mid, nodeIdx + 1, depth + 1, bl, dl);
        const int32_t rr = build_small(mid, end, nodeIdx + (int32_t)bestPos, depth + 1, br, dr);
        write_node(nodeIdx, bl, br, rl, rr);
        outBox = bl; grow(outBox, br);
        outDepth = std::max(dl, dr);
        return nodeIdx;
    }

    // one binned split of items[begin,end) (more than kSweepNode of them, centroids inside `cbounds`): partitions the
    // range, returns the first item of the right side and both sides' centroid bounds
    uint32_t split_node(uint32_t begin, uint32_t end, const Box& cbounds, int threads, Box& cl, Box& cr) {
        const uint32_t n = end - begin;
        Split sp; bool any = false;
        for (int a = 0; a < 3; ++a) {
            const float ext = cbounds.hi[a] - cbounds.lo[a];
            sp.valid[a] = ext > 0.0f && std::isfinite(ext);
            sp.lo[a] = cbounds.lo[a]; sp.scale[a] = sp.valid[a] ? (float)kBins / ext : 0.0f;
            any = any || sp.valid[a];
        }
        const bool wide = threads > 1 && n >= kParallelNode;
        Bins bins; bins.clear();
        if (any) {
            if (wide) {
                std::vector<Bins> local((size_t)threads);
                parallel_chunks(begin, end, threads, [&](int t, uint32_t b, uint32_t e) { local[(size_t)t].clear(); bin_range(b, e, sp, local[(size_t)t]); });
                for (const Bins& l : local) bins.merge(l);
            } else {
                bin_range(begin, end, sp, bins);
            }
        }
        int bestAxis = -1, bestBin = -1; double bestCost = INFINITY;
        for (int a = 0; a < 3; ++a) {
            if (!sp.valid[a]) continue;
            double rightArea[kBins]; uint32_t rightCnt[kBins];
            Box acc = empty_box(); uint32_t c = 0;
            for (int k = kBins - 1; k > 0; --k) { grow(acc, bins.box(a, k)); c += bins.cnt[a][k]; rightArea[k] = half_area(acc); rightCnt[k] = c; }
            acc = empty_box(); c = 0;
            for (int k = 0; k < kBins - 1; ++k) {
                grow(acc, bins.box(a, k)); c += bins.cnt[a][k];
                if (c == 0 || rightCnt[k + 1] == 0) continue;
                const double cost = half_area(acc) * c + rightArea[k + 1] * rightCnt[k + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; }
            }
        }
        uint32_t mid;
        cl = empty_box(); cr = empty_box();
        if (bestAxis >= 0) {
            const int a = bestAxis;
            const float lo = sp.lo[a], scale = sp.scale[a];
            auto left = [&](const Item& it) { return bin_of(0.5f * it.lo[a] + 0.5f * it.hi[a], lo, scale) <= bestBin; };
            if (wide) {
                // stable two-sided scatter through the scratch array: per-chunk counts give every chunk its
                // own output windows on both sides, so the chunks write without meeting
                std::vector<uint32_t> nl((size_t)threads, 0), chunkBegin((size_t)threads, end), chunkEnd((size_t)threads, end);
                parallel_chunks(begin, end, threads, [&](int t, uint32_t b, uint32_t e) {
                    uint32_t c = 0; for (uint32_t i = b; i < e; ++i) c += left(items[i]) ? 1u : 0u;
                    nl[(size_t)t] = c; chunkBegin[(size_t)t] = b; chunkEnd[(size_t)t] = e;
                });
                uint32_t totalLeft = 0; for (uint32_t c : nl) totalLeft += c;
                std::vector<uint32_t> offL((size_t)threads), offR((size_t)threads);
                uint32_t accL = begin, accR = begin + totalLeft;
                for (int t = 0; t < threads; ++t) {
                    offL[(size_t)t] = accL; offR[(size_t)t] = accR;
                    accL += nl[(size_t)t]; accR += (chunkEnd[(size_t)t] - chunkBegin[(size_t)t]) - nl[(size_t)t];
                }
                std::vector<Box> lcl((size_t)threads, empty_box()), lcr((size_t)threads, empty_box());
                parallel_chunks(begin, end, threads, [&](int t, uint32_t b, uint32_t e) {
                    uint32_t l = offL[(size_t)t], r = offR[(size_t)t];
                    Box ml = empty_box(), mr = empty_box();
                    for (uint32_t i = b; i < e; ++i) {
                        if (left(items[i])) { grow_centroid(ml, items[i]); scratch[l++] = items[i]; }
                        else { grow_centroid(mr, items[i]); scratch[r++] = items[i]; }
                    }
                    lcl[(size_t)t] = ml; lcr[(size_t)t] = mr;
                });
                for (int t = 0; t < threads; ++t) { grow(cl, lcl[(size_t)t]); grow(cr, lcr[(size_t)t]); }
                parallel_chunks(begin, end, threads, [&](int, uint32_t b, uint32_t e) { memcpy(items + b, scratch + b, (size_t)(e - b) * sizeof(Item)); });
                mid = begin + totalLeft;
            } else {
                // two-pointer partition that also collects both sides' centroid bounds (the children need them)
                uint32_t i = begin, j = end;
                for (;;) {
                    while (i < j && left(items[i])) { grow_centroid(cl, items[i]); ++i; }
                    while (i < j && !left(items[j - 1])) { grow_centroid(cr, items[j - 1]); --j; }
                    if (i >= j) break;
                    std::swap(items[i], items[j - 1]);
                    grow_centroid(cl, items[i]); grow_centroid(cr, items[j - 1]);
                    ++i; --j;
                }
                mid = i;
            }
        } else {
            mid = begin;          // all centroids coincide (or are not finite): split by count
        }
        if (mid == begin || mid == end) {
            mid = begin + n / 2;
            cl = empty_box(); cr = empty_box();
            for (uint32_t i = begin; i < end; ++i) {
                grow_centroid(i < mid ? cl : cr, items[i]);
            }
        }
        return mid;
    }

    // builds the subtree over items[begin,end) whose centroids lie in `cbounds`; returns its ref
    // (>= 0 inner index, < 0 ~leafPosition)
    int32_t build(uint32_t begin, uint32_t end, int32_t nodeIdx, int depth, const Box& cbounds, int threads, Box& outBox, int& outDepth) {
        const uint32_t n = end - begin;
        if (n == 1) { outBox = items[begin].box(); outDepth = depth; return ~(int32_t)begin; }
        if (n <= kSweepNode) return build_small(begin, end, nodeIdx, depth, outBox, outDepth);
        Box cl, cr;
        const uint32_t mid = split_node(begin, end, cbounds, threads, cl, cr);
        const uint32_t nl = mid - begin;
        const int32_t leftIdx = nodeIdx + 1, rightIdx = nodeIdx + (int32_t)nl;   // left subtree has nl-1 inner nodes
        Box bl, br; int dl = 0, dr = 0; int32_t rl, rr;
        if (threads > 1 && n > 8192) {
            // threads are shared out by size, so that both sides finish together — as far as whole threads can be
            // shared out; a side that finishes early takes subtrees off the other's hands meanwhile (below)
            int tl = (int)((double)threads * nl / n + 0.5);
            tl = tl < 1 ? 1 : (tl > threads - 1 ? threads - 1 : tl);
            std::atomic<bool> leftDone{false}, rightDone{false};
            std::thread t([&]() {
                rl = build(begin, mid, leftIdx, depth + 1, cl, tl, bl, dl);
                leftDone.store(true, std::memory_order_release);
                while (!rightDone.load(std::memory_order_acquire)) if (!help_one()) std::this_thread::yield();
            });
            rr = build(mid, end, rightIdx, depth + 1, cr, threads - tl, br, dr);
            rightDone.store(true, std::memory_order_release);
            while (!leftDone.load(std::memory_order_acquire)) if (!help_one()) std::this_thread::yield();
            t.join();
        } else if (stealing && n > kStealGrain) {
            // one thread, a large subtree: the right side is offered to threads that have run out of work
            Offer o{mid, end, rightIdx, depth + 1, &cr, &br, &dr, &rr, {0}};
            { std::lock_guard<std::mutex> g(offerMutex); offers.push_back(&o); offerCount.store((int)offers.size(), std::memory_order_relaxed); }
            rl = build(begin, mid, leftIdx, depth + 1, cl, 1, bl, dl);
            bool mine = false;
            {
                std::lock_guard<std::mutex> g(offerMutex);
                if (o.state.load(std::memory_order_relaxed) == 0) {
                    for (size_t k = offers.size(); k-- > 0;) if (offers[k] == &o) { offers.erase(offers.begin() + (long)k); break; }
                    offerCount.store((int)offers.size(), std::memory_order_relaxed);
                    o.state.store(1, std::memory_order_relaxed);
                    mine = true;
                }
            }
            if (mine) rr = build(mid, end, rightIdx, depth + 1, cr, 1, br, dr);
            else while (o.state.load(std::memory_order_acquire) != 2) if (!help_one()) std::this_thread::yield();
        } else {
            rl = build(begin, mid, leftIdx, depth + 1, cl, 1, bl, dl);
            rr = build(mid, end, rightIdx, depth + 1, cr, 1, br, dr);
        }
        write_node(nodeIdx, bl, br, rl, rr);
        outBox = bl; grow(outBox, br);
        outDepth = std::max(dl, dr);
        return nodeIdx;
    }

    // ---- subtrees on offer ------------------------------------------------------------------------------------------
    // Whole threads cannot be shared out evenly (3 threads over a 50 / 50 split leave one side twice the time of the
    // other, and a scene's few huge primitives make the top splits lopsided), so a thread that builds a subtree of
    // more than kStealGrain primitives alone puts the right side of each split on offer before it descends into the
    // left, and takes it back afterwards unless a thread with nothing left to do has taken it: those are the threads
    // that would otherwise sit in join(). 0.56 -> see tools/build_time on a scene's own leaf boxes.
    static const uint32_t kStealGrain = 1u << 15;
    struct Offer { uint32_t begin, end; int32_t nodeIdx; int depth; const Box* cbounds; Box* outBox; int* outDepth; int32_t* ref; std::atomic<int> state; };   // 0 offered, 1 taken, 2 done
    bool stealing = false;
    std::mutex offerMutex;
    std::vector<Offer*> offers;           // oldest (largest) first
    std::atomic<int> offerCount{0};
    bool help_one() {
        static thread_local int nesting = 0;
        if (!stealing || nesting >= 4 || offerCount.load(std::memory_order_relaxed) == 0) return false;          // every level of help is a recursion of its own on this thread's stack
        Offer* o = nullptr;
        {
            std::lock_guard<std::mutex> g(offerMutex);
            if (offers.empty()) return false;
            o = offers.front(); offers.erase(offers.begin());
            offerCount.store((int)offers.size(), std::memory_order_relaxed);
            o->state.store(1, std::memory_order_relaxed);
        }
        ++nesting;
        *o->ref = build(o->begin, o->end, o->nodeIdx, o->depth, *o->cbounds, 1, *o->outBox, *o->outDepth);
        --nesting;
        o->state.store(2, std::memory_order_release);
        return true;
    }

    // (Tried instead of sharing the threads out between the two sides of every split: all nodes of 128 K primitives or
    // more split one after the other on all threads, the subtrees below them from a queue, largest first. Balanced,
    // but the scratch-array partition of the parallel split then runs on six levels instead of four: 0.45 s against
    // 0.31 s at 10 M primitives on 16 threads.)
};

// order[] in: any permutation of 0..n-1; out: leaf order of the new tree.
// Returns the root ref; nodes must hold n-1 records.
inline int32_t build_sah(const Box* prim, uint32_t n, uint32_t* order, Node* nodes, int nthreads, int& maxDepth) {
    Raw<Item> items(n), scratch(n);
    // the root's centroid bounds: the only pass of this kind (on all threads: it is also the first touch of `items`)
    Box cb = empty_box();
    {
        const int T = nthreads < 1 ? 1 : (n < 65536 ? 1 : nthreads);
        std::vector<Box> part((size_t)T, empty_box());
        SahBuilder::parallel_chunks(0, n, T, [&](int t, uint32_t b, uint32_t e) {
            Box local = empty_box();
            for (uint32_t i = b; i < e; ++i) {
                const Box& pb = prim[order[i]];
                for (int a = 0; a < 3; ++a) { items[i].lo[a] = pb.lo[a]; items[i].hi[a] = pb.hi[a]; }
                items[i].id = order[i]; items[i].pad = 0;
                SahBuilder::grow_centroid(local, items[i]);
            }
            part[(size_t)t] = local;
        });
        for (const Box& pbx : part) grow(cb, pbx);
    }
    SahBuilder b;
    b.items = items.data(); b.nodes = nodes; b.scratch = scratch.data(); b.stealing = nthreads > 1;
    Box root; int depth = 0;
    const int32_t ref = b.build(0, n, 0, 0, cb, nthreads < 1 ? 1 : nthreads, root, depth);
    SahBuilder::parallel_chunks(0, n, nthreads < 1 ? 1 : (n < 65536 ? 1 : nthreads), [&](int, uint32_t b, uint32_t e) { for (uint32_t i = b; i < e; ++i) order[i] = items[i].id; });
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
struct Node4 {          // 128-byte record: lo.x[4] hi.x[4] lo.y[4] hi.y[4] lo.z[4] hi.z[4] ref[4] (nodes, leaves of the subtree, as ints) pad[2]
    float q[32];
};

struct Collapser {
    const Node* bvh2;
    Raw<Node4>& out;
    Raw<float> T;              // 4 per binary node: T[4 * n + j - 1]; every entry is written by solve()
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

    void solve_node(size_t n) {
        Child c[2]; children_of(bvh2[n], c);
        Box u = c[0].box; grow(u, c[1].box);
        int k;
        const float s2 = split(c, 2, k), s3 = split(c, 3, k), s4 = split(c, 4, k);
        const float keep = (float)half_area(u) + s4;
        T[4 * n + 0] = keep; T[4 * n + 1] = std::min(keep, s2); T[4 * n + 2] = std::min(keep, s3); T[4 * n + 3] = std::min(keep, s4);
    }
    // build_sah numbers nodes in pre-order: the subtree of node n is the index range [n, end), its left
    // subtree starts at n + 1 and its right subtree where the right child's index says. Bottom-up = by
    // descending index; disjoint subtrees are solved on separate threads.
    void solve(size_t n, size_t end, int threads) {
        Child c[2]; children_of(bvh2[n], c);
        if (threads > 1 && end - n > 65536 && c[0].ref >= 0 && c[1].ref >= 0) {
            const size_t l = (size_t)c[0].ref, r = (size_t)c[1].ref;
            int tl = (int)((double)threads * (double)(r - l) / (double)(end - n) + 0.5);
            tl = tl < 1 ? 1 : (tl > threads - 1 ? threads - 1 : tl);
            std::thread t([&]() { solve(l, r, tl); });
            solve(r, end, threads - tl);
            t.join();
            solve_node(n);
            return;
        }
        for (size_t i = end; i-- > n;) solve_node(i);
    }

    // the children that subtree `ref` contributes when it may use up to j slots
    void collect(const Child& me, int j, Child* slots, int& n) const {
        if (me.ref < 0 || j == 1 || !(t(me.ref, j) < t(me.ref, 1))) { slots[n++] = me; return; }
        Child c[2]; children_of(bvh2[me.ref], c);
        int k; split(c, j, k);
        collect(c[0], k, slots, n);
        collect(c[1], j - k, slots, n);
    }

    static void write_node4(Node4& node, const Child* c, int n, const int32_t* refs, int32_t subNodes, int32_t subLeaves) {
        float* q = node.q;
        for (int i = 0; i < 4; ++i) {
            const bool used = i < n;
            for (int a = 0; a < 3; ++a) {
                q[8 * a + i] = used ? c[i].box.lo[a] : NAN;           // an empty slot never passes the slab test:
                q[8 * a + 4 + i] = used ? c[i].box.hi[a] : NAN;       // every comparison with its NaN distances is false
            }
            q[24 + i] = i2f(refs[i]);
        }
        q[28] = i2f(subNodes); q[29] = i2f(subLeaves); q[30] = 0.0f; q[31] = 0.0f;     // sizes of the subtree (quantize4 numbers with them)
    }

    // emits the wide node for binary node `ref2` (>= 0) into dst (pre-order: a node before its subtrees) and returns its index
    int32_t emit(int32_t ref2, int depth, Raw<Node4>& dst, int& deepest) const {
        deepest = std::max(deepest, depth);
        Child c2[2]; children_of(bvh2[ref2], c2);
        int k; split(c2, 4, k);
        Child c[4]; int n = 0;
        collect(c2[0], k, c, n);
        collect(c2[1], 4 - k, c, n);
        const int32_t idx = (int32_t)dst.size();
        dst.emplace_back();
        int32_t refs[4], nodes = 1, leaves = 0;
        for (int i = 0; i < 4; ++i) {
            if (i >= n) { refs[i] = (int32_t)0x80000000; continue; }  // RTX_NO_CHILD
            if (c[i].ref < 0) { refs[i] = c[i].ref; leaves++; continue; }
            refs[i] = emit(c[i].ref, depth + 1, dst, deepest);
            nodes += f2i(dst[(size_t)refs[i]].q[28]); leaves += f2i(dst[(size_t)refs[i]].q[29]);
        }
        write_node4(dst[(size_t)idx], c, n, refs, nodes, leaves);
        return idx;
    }

    // The same tree, emitted on several threads: the top of the tree is emitted here, and every subtree of at most
    // `grain` binary nodes it meets becomes a task; the tasks are emitted into vectors of their own (indices relative
    // to the subtree) and appended behind the top in the order they were met, so the layout does not depend on timing.
    // Binary nodes are in pre-order (build_sah): the subtree of node x is the index range [x, end).
    struct Task { int32_t ref2, end; int depth; int32_t parent; int slot; Raw<Node4> nodes; int deepest = 0; };
    int32_t emit_top(int32_t ref2, int32_t end, int depth, size_t grain, std::vector<Task>& tasks) {
        maxDepth = std::max(maxDepth, depth);
        Child c2[2]; children_of(bvh2[ref2], c2);
        int k; split(c2, 4, k);
        Child c[4]; int32_t ends[4]; int n = 0;
        // index ranges of the two binary children, then of whatever `collect` dissolves them into
        collect_ranges(c2[0], c2[1].ref >= 0 ? c2[1].ref : end, k, c, ends, n);
        collect_ranges(c2[1], end, 4 - k, c, ends, n);
        const int32_t idx = (int32_t)out.size();
        out.emplace_back();
        int32_t refs[4];
        for (int i = 0; i < 4; ++i) {
            if (i >= n) { refs[i] = (int32_t)0x80000000; continue; }
            if (c[i].ref < 0) { refs[i] = c[i].ref; continue; }
            if ((size_t)(ends[i] - c[i].ref) <= grain) {
                refs[i] = 0;                                           // patched once the task's position is known
                tasks.emplace_back();
                Task& t = tasks.back();
                t.ref2 = c[i].ref; t.end = ends[i]; t.depth = depth + 1; t.parent = idx; t.slot = i;
            } else {
                refs[i] = emit_top(c[i].ref, ends[i], depth + 1, grain, tasks);
            }
        }
        write_node4(out[(size_t)idx], c, n, refs, 0, 0);              // sizes: collapse4 fills them in when the tasks are done
        return idx;
    }
    void collect_ranges(const Child& me, int32_t meEnd, int j, Child* slots, int32_t* ends, int& n) const {
        if (me.ref < 0 || j == 1 || !(t(me.ref, j) < t(me.ref, 1))) { ends[n] = meEnd; slots[n++] = me; return; }
        Child c[2]; children_of(bvh2[me.ref], c);
        int k; split(c, j, k);
        collect_ranges(c[0], c[1].ref >= 0 ? c[1].ref : meEnd, k, slots, ends, n);
        collect_ranges(c[1], meEnd, j - k, slots, ends, n);
    }
};

// Folds the binary tree of `numNodes` inner nodes rooted at node `root2` into 4-wide nodes (root = index 0).
inline void collapse4(const Node* bvh2, size_t numNodes, int32_t root2, Raw<Node4>& out, int& maxDepth, int nthreads = 1) {
    Collapser c{bvh2, out};
    c.T.resize(4 * numNodes);
    if (nthreads < 1) nthreads = 1;
    c.solve((size_t)root2, numNodes, nthreads);
    if (nthreads == 1 || numNodes < 65536) {
        c.emit(root2, 0, out, c.maxDepth);
        maxDepth = c.maxDepth;
        return;
    }
    std::vector<Collapser::Task> tasks;
    const size_t grain = std::max<size_t>(4096, numNodes / ((size_t)nthreads * 8));
    c.emit_top(root2, (int32_t)numNodes, 0, grain, tasks);
    const size_t top = out.size();
    {
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        auto worker = [&]() {
            for (size_t i = next++; i < tasks.size(); i = next++) {
                Collapser::Task& t = tasks[i];
                t.nodes.reserve((size_t)(t.end - t.ref2) / 2 + 4);
                c.emit(t.ref2, t.depth, t.nodes, t.deepest);
            }
        };
        for (int k = 1; k < nthreads; ++k) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
    }
    std::vector<size_t> base(tasks.size() + 1, top);
    for (size_t i = 0; i < tasks.size(); ++i) base[i + 1] = base[i] + tasks[i].nodes.size();
    out.resize(base.back());
    {
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        auto worker = [&]() {
            for (size_t i = next++; i < tasks.size(); i = next++) {
                const Collapser::Task& t = tasks[i];
                const int32_t off = (int32_t)base[i];
                Node4* dst = out.data() + base[i];
                for (size_t k = 0; k < t.nodes.size(); ++k) {
                    dst[k] = t.nodes[k];
                    for (int j = 0; j < 4; ++j) { const int32_t r = f2i(dst[k].q[24 + j]); if (r >= 0) dst[k].q[24 + j] = i2f(r + off); }
                }
            }
        };
        for (int k = 1; k < nthreads; ++k) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
    }
    for (size_t i = 0; i < tasks.size(); ++i) {
        out[(size_t)tasks[i].parent].q[24 + tasks[i].slot] = i2f((int32_t)base[i]);
        c.maxDepth = std::max(c.maxDepth, tasks[i].deepest);
    }
    for (size_t i = top; i-- > 0;) {                                  // subtree sizes of the top nodes (children have higher indices)
        int32_t nodes = 1, leaves = 0;
        for (int j = 0; j < 4; ++j) {
            const int32_t r = f2i(out[i].q[24 + j]);
            if (r >= 0) { nodes += f2i(out[(size_t)r].q[28]); leaves += f2i(out[(size_t)r].q[29]); }
            else if (r != (int32_t)0x80000000) leaves++;
        }
        out[i].q[28] = i2f(nodes); out[i].q[29] = i2f(leaves);
    }
    maxDepth = c.maxDepth;
}

// A wide node holding a single leaf (scenes with one primitive) or nothing at all.
inline void single_leaf_node4(const Box* leafBox, Raw<Node4>& out) {
    out.emplace_back();
    float* q = out.back().q;
    for (int k = 0; k < 4; ++k) {
        const bool used = leafBox && k == 0;
        for (int a = 0; a < 3; ++a) { q[8 * a + k] = used ? leafBox->lo[a] : NAN; q[8 * a + 4 + k] = used ? leafBox->hi[a] : NAN; }
        q[24 + k] = i2f(used ? ~0 : (int32_t)0x80000000);
        q[28 + k] = 0.0f;
    }
    q[28] = i2f(1); q[29] = i2f(leafBox ? 1 : 0);
}

// ---- quantised 4-wide nodes ----------------------------------------------------------
// RTX_TRAVERSE_WIDE never culls by distance and never needs an order, so an inner test only has to be
// CONSERVATIVE: it must pass whenever the exact test of a leaf below it passes (the leaf's own exact box,
// stored with the leaf, has the last word). A node therefore stores its children's boxes as bytes on a grid
// of 256 steps per axis spanned from the node's corner: plane = fma(byte, 2^e, origin), one rounding, and the
// bytes are chosen so that the DECODED float plane lies outside the child's float box (lo rounded down, hi up,
// verified with the same fma). The kernel then applies the reference's own (p - o) * inv to the decoded
// plane; that function is monotonic in p under IEEE rounding, hence decoded box contains child box =>
// entry <= child's entry, exit >= child's exit: whenever a descendant leaf passes, every node above it passes.
// 48 bytes instead of 112: three 16-byte lane loads per step instead of seven (the traversal kernels are
// bound by the number of lane loads the vector L1 accepts).
//
// Children are sorted inner first, then leaves; a node's inner children are neighbours in the node array and
// its leaf children are neighbours in the leaf array, so the node stores two bases instead of four refs:
// child in slot j = node baseInner + j, or leaf baseLeaf + j (bases are pre-biased; slots are what the valid /
// leaf masks and the byte lanes are indexed by).
#ifndef RTX_QNODE_WORDS
#define RTX_QNODE_WORDS 16      // words from one node to the next: 16 = one node per 64-byte slot, 12 = packed
#endif
// Work-list entries of the quantised walk (rtx_traverse.h RTX_QENTRY / q_pass): children base + j of one node for every set
// bit j of the mask, as base << 4 | mask in an int; the base comes back through an arithmetic shift, because the bases are
// pre-biased (baseLeaf = first leaf child - number of inner children) and may be as low as -4. So bases must stay below
// 2^27: rtx_create only quantises trees of at most kQuantMaxLeaves leaves (a node index is always below the leaf count).
constexpr int32_t kQuantMaxLeaves = (1 << 27) - 4;
constexpr int32_t q_entry(int32_t base, uint32_t mask) { return (int32_t)(((uint32_t)base << 4) | mask); }
constexpr int32_t q_entry_base(int32_t e) { return e >> 4; }

struct NodeQ {          // 48 bytes used
    // w[0..2] origin.xyz  w[3] = ex | ey << 8 | ez << 16 | valid << 24 | leaf << 28   (e* = biased float exponent of the step)
    // w[4] lo.x bytes (slot j in byte j)  w[5] hi.x  w[6] lo.y  w[7] hi.y  w[8] lo.z  w[9] hi.z  w[10] baseInner  w[11] baseLeaf
    uint32_t w[RTX_QNODE_WORDS];
};

// Converts the Node4 tree (root = node 0) into NodeQ records (root = node 0 again) and returns, for every new leaf
// position, the position the leaf had in the order the Node4 tree refers to. false: a box extent is not finite
// (the caller keeps the reference topology then).
inline bool quantize4(const Raw<Node4>& in, Raw<NodeQ>& out, std::vector<uint32_t>& leafNewToOld, int nthreads, double* phaseSeconds = nullptr) {
    auto tick = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tStart = tick();
    const int32_t kNone = (int32_t)0x80000000;
    const size_t n = in.size();
    struct Plan { int32_t newIdx, baseInner, baseLeaf; uint8_t slot[4], ni, nl; };
    Raw<Plan> plan(n);
    leafNewToOld.clear();
    // Numbering: a node's inner children take consecutive indices ("block"), handed out when the node is visited, and
    // the walk then descends into the first child — so the descendants of child k follow the block and the
    // descendants of the children before it, and likewise for leaves. With the subtree sizes the fold left in every
    // node that is arithmetic: the top of the tree is numbered here, subtrees of at most `grain` nodes on all threads.
    if (n && (f2i(in[0].q[28]) != (int32_t)n || f2i(in[0].q[29]) < 0)) return false;       // sizes missing
    leafNewToOld.assign(n ? (size_t)f2i(in[0].q[29]) : 0, 0u);
    const double tSizes = tick();
    struct Job { int32_t node, newIdx, descBase, leafBase; };
    auto number_node = [&](const Job& j, Job* kids) -> int {          // fills plan[j.node], returns the inner children as jobs
        Plan& p = plan[(size_t)j.node];
        const float* q = in[(size_t)j.node].q;
        p.newIdx = j.newIdx;
        p.ni = p.nl = 0;
        for (int k = 0; k < 4; ++k) { const int32_t r = f2i(q[24 + k]); if (r >= 0) p.slot[p.ni++] = (uint8_t)k; }
        for (int k = 0; k < 4; ++k) { const int32_t r = f2i(q[24 + k]); if (r < 0 && r != kNone) p.slot[p.ni + p.nl++] = (uint8_t)k; }
        p.baseInner = j.descBase;
        p.baseLeaf = j.leafBase - (int32_t)p.ni;
        for (int m = 0; m < p.nl; ++m) leafNewToOld[(size_t)(j.leafBase + m)] = (uint32_t)~f2i(q[24 + p.slot[p.ni + m]]);
        int32_t nextDesc = j.descBase + (int32_t)p.ni, nextLeaf = j.leafBase + (int32_t)p.nl;
        for (int m = 0; m < p.ni; ++m) {
            const int32_t c = f2i(q[24 + p.slot[m]]);
            kids[m] = Job{c, j.descBase + m, nextDesc, nextLeaf};
            nextDesc += f2i(in[(size_t)c].q[28]) - 1; nextLeaf += f2i(in[(size_t)c].q[29]);
        }
        return p.ni;
    };
    auto number_subtree = [&](const Job& root) {
        std::vector<Job> st{root};
        Job kids[4];
        while (!st.empty()) {
            const Job j = st.back(); st.pop_back();
            const int k = number_node(j, kids);
            for (int m = k; m-- > 0;) st.push_back(kids[m]);
        }
    };
    if (n) {
        const int32_t grain = (int32_t)std::max<size_t>(1024, n / ((size_t)std::max(1, nthreads) * 8));
        std::vector<Job> top{Job{0, 0, 1, 0}}, jobs;
        Job kids[4];
        while (!top.empty()) {
            const Job j = top.back(); top.pop_back();
            if (nthreads > 1 && f2i(in[(size_t)j.node].q[28]) <= grain) { jobs.push_back(j); continue; }
            const int k = number_node(j, kids);
            for (int m = k; m-- > 0;) top.push_back(kids[m]);
        }
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        auto worker = [&]() { for (size_t i = next++; i < jobs.size(); i = next++) number_subtree(jobs[i]); };
        for (int t = 1; t < nthreads && t < (int)jobs.size(); ++t) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
    }
    const double tNumbered = tick();
    out.resize(n);                                                  // every word is written below
    std::vector<uint8_t> bad((size_t)std::max(1, nthreads), 0);
    auto work = [&](int t, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const Plan& p = plan[i];
            const float* q = in[i].q;
            NodeQ& o = out[(size_t)p.newIdx];
            for (int k = 12; k < RTX_QNODE_WORDS; ++k) o.w[k] = 0;
            const int nc = p.ni + p.nl;
            uint32_t meta = 0;
            for (int a = 0; a < 3; ++a) {
                float lo = INFINITY, hi = -INFINITY;
                for (int j = 0; j < nc; ++j) { lo = std::min(lo, q[8 * a + p.slot[j]]); hi = std::max(hi, q[8 * a + 4 + p.slot[j]]); }
                if (nc == 0) { lo = 0.0f; hi = 0.0f; }
                const double ext = (double)hi - (double)lo;
                // (coordinates beyond +-2^40: the folded inner test of the walk is proven overflow-free only inside that range, rtx_traverse.h)
                if (!std::isfinite(ext) || !std::isfinite(lo) || ext < 0.0 || std::fabs(lo) > 1.099511627776e12f || std::fabs(hi) > 1.099511627776e12f) { bad[(size_t)t] = 1; lo = 0.0f; hi = 0.0f; }
                int eb = 1;                                           // biased exponent of the grid step, 1..254
                if (ext > 0.0 && std::isfinite(ext)) { int ex; std::frexp(ext / 255.0, &ex); eb = std::min(254, std::max(1, ex + 127)); }   // 2^ex >= ext / 255
                uint32_t wl = 0, wh = 0;
                // The grid step is a power of two, so position / step is a multiplication by its (exact) reciprocal, and
                // the positions are not negative, so floor is the cast. The decoded plane the kernel computes is
                // fl32(q * step + lo); q * step + lo rounded to double instead (one rounding of the exact sum, q * step
                // being exact) lies on the same side of the child's plane or rounds onto it exactly when the float does,
                // so "double <= clo" implies "float <= clo" (and likewise >= chi): no libm call per plane.
                for (;;) {
                    const double step = (double)i2f(eb << 23), inv = std::ldexp(1.0, 127 - eb), dlo = (double)lo;
                    bool fits = true; wl = 0; wh = 0;
                    for (int j = 0; j < nc && fits; ++j) {
                        const float clo = q[8 * a + p.slot[j]], chi = q[8 * a + 4 + p.slot[j]];
                        const double xl = ((double)clo - dlo) * inv, xh = ((double)chi - dlo) * inv;
                        if (!(xh <= 255.0)) { fits = false; break; }
                        int ql = xl >= 255.0 ? 255 : (int)xl;
                        while (ql > 0 && (double)ql * step + dlo > (double)clo) --ql;           // decoded plane must not lie inside
                        int qh = (int)xh;
                        if ((double)qh < xh) ++qh;
                        while (qh <= 255 && (double)qh * step + dlo < (double)chi) ++qh;
                        if (qh > 255) { fits = false; break; }
                        wl |= (uint32_t)ql << (8 * j); wh |= (uint32_t)qh << (8 * j);
                    }
                    if (fits) break;
                    if (eb >= 254) { bad[(size_t)t] = 1; break; }
                    ++eb;
                }
                o.w[a] = (uint32_t)f2i(lo);
                meta |= (uint32_t)eb << (8 * a);
                o.w[4 + 2 * a] = wl; o.w[5 + 2 * a] = wh;
            }
            meta |= ((1u << nc) - 1u) << 24;
            meta |= (((1u << nc) - 1u) & ~((1u << p.ni) - 1u)) << 28;
            o.w[3] = meta;
            o.w[10] = (uint32_t)p.baseInner; o.w[11] = (uint32_t)p.baseLeaf;
        }
    };
    if (nthreads > 1 && n > 4096) {
        std::vector<std::thread> pool;
        const size_t per = (n + (size_t)nthreads - 1) / (size_t)nthreads;
        for (int t = 0; t < nthreads; ++t) { const size_t b = std::min(n, (size_t)t * per), e = std::min(n, b + per); if (b < e) pool.emplace_back(work, t, b, e); }
        for (auto& th : pool) th.join();
    } else {
        work(0, 0, n);
    }
    if (phaseSeconds) { phaseSeconds[0] = tSizes - tStart; phaseSeconds[1] = tNumbered - tSizes; phaseSeconds[2] = tick() - tNumbered; }
    for (uint8_t b : bad) if (b) return false;
    return true;
}

} // namespace rtxbuild
