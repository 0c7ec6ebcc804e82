#include "bvh_builder.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <thread>
#include <stdexcept>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "../kernels/bvh_layout.h"
#include "knobs.h"
#include "parallel.h"

namespace ptr {
namespace {

constexpr int kBins = 16;
constexpr uint32_t kDepthLimit = ptrk::kMaxTreeDepth - 2;   // leaves sit at depth <= kDepthLimit

// Boxes are four floats wide (the fourth lane is never read) so that growing one is two SSE instructions: the binning and partition
// passes of a 29 M-primitive build spend their time in these min / max, six scalar ones per primitive and axis before.
#if defined(__SSE2__)
#define PTR_BOX_SSE 1
#else
#define PTR_BOX_SSE 0
#endif
struct alignas(16) Aabb {
    float lo[4], hi[4];
    void reset() {
        for (int a = 0; a < 4; ++a) {
            lo[a] = std::numeric_limits<float>::infinity();
            hi[a] = -std::numeric_limits<float>::infinity();
        }
    }
    // l, h: FOUR readable floats each (a Rec's bounds are followed by its id / flag word, a box by its padding lane)
    void grow4(const float* l, const float* h) {
#if PTR_BOX_SSE
        _mm_store_ps(lo, _mm_min_ps(_mm_load_ps(lo), _mm_loadu_ps(l)));
        _mm_store_ps(hi, _mm_max_ps(_mm_load_ps(hi), _mm_loadu_ps(h)));
#else
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], l[a]);
            hi[a] = std::max(hi[a], h[a]);
        }
#endif
    }
    void grow(const Aabb& o) { grow4(o.lo, o.hi); }
    void grow(const float l[3], const float h[3]) {   // three floats readable
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], l[a]);
            hi[a] = std::max(hi[a], h[a]);
        }
    }
    void growPoint(const float p[3]) { grow(p, p); }
    float halfArea() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

// Arrays of tens of millions of plain records: allocated WITHOUT being filled (a std::vector would write every element once on
// the calling thread - seconds for a 29 M-triangle scene - and place all its pages next to that one core); whoever writes an
// element first touches its page.
template <typename T>
struct RawArray {
    std::unique_ptr<T[]> items;
    size_t capacity = 0;
    void allocate(size_t n) {
        items.reset(new T[n]);
        capacity = n;
    }
    void release() {
        items.reset();
        capacity = 0;
    }
    T& operator[](size_t i) { return items[i]; }
    const T& operator[](size_t i) const { return items[i]; }
    T* data() { return items.get(); }
};

struct TempNode {   // no initialisers: see RawArray.  A leaf sets count (> 0) and the totals, an internal node everything but `first`.
    Aabb box;
    uint32_t left, right;   // children (internal)
    uint32_t first, count;  // range in `order` (leaf when count > 0)
    uint32_t sphereLeaf;
    // subtree totals, filled on the way back up: what the flattener needs to give every subtree its own slice of the node /
    // triangle / sphere arrays, so that subtrees can be laid out in parallel
    uint32_t internalNodes, triPrims, spherePrims;
};

// The builder's working copy of a primitive: bounds + input index in one 32 B record.  The records of a node are one contiguous
// run that is partitioned in place, so every pass over a node streams through memory; with an index array into the caller's
// primitives every access below the top of a large tree was a cache miss (5.7 of the 7.5 s a 29 M-triangle build took).
struct alignas(16) Rec {
    float lo[3];
    uint32_t id;
    float hi[3];
    uint32_t isSphere;
    float centre(int axis) const { return 0.5f * (lo[axis] + hi[axis]); }
};

struct Builder {
    RawArray<Rec> recs;
    std::vector<uint32_t> order;      // recs[i].id once the tree stands (what the flattener reads)
    RawArray<TempNode> nodes;
    std::atomic<uint32_t> nextNode{1};
    uint32_t kLeafMax = 4;   // SAH may stop at <= kLeafMax primitives

    // Two nodes for the children of a split.  The subtrees of the task list take them from a block of their own (`cursor`): with one
    // shared counter, neighbouring pairs belong to different threads and every write to a node drags its cache line across the
    // machine (64 threads: 1 us per primitive and level instead of 70 ns).
    uint32_t allocPair(uint32_t* cursor) {
        if (cursor) {
            const uint32_t at = *cursor;
            *cursor += 2;
            return at;
        }
        return nextNode.fetch_add(2);
    }

    static uint32_t log2Ceil(uint32_t n) {
        uint32_t l = 0;
        while ((1u << l) < n) ++l;
        return l;
    }

    void makeLeaf(uint32_t node, uint32_t begin, uint32_t end) {
        nodes[node].first = begin;
        nodes[node].count = end - begin;
        nodes[node].sphereLeaf = recs[begin].isSphere;
        nodes[node].internalNodes = 0;
        nodes[node].triPrims = nodes[node].sphereLeaf ? 0u : end - begin;
        nodes[node].spherePrims = nodes[node].sphereLeaf ? end - begin : 0u;
    }

    // passes over large nodes are cut into one chunk per 64 Ki primitives, at most wideThreads of them
    uint32_t wideThreads = 1;
    RawArray<Rec> scratch;   // parallel partition of wide nodes

    template <typename Fn>
    void forChunks(uint32_t begin, uint32_t end, Fn&& fn) const {   // fn(chunkIndex, chunkBegin, chunkEnd)
        const uint32_t t = std::max(1u, std::min(wideThreads, (end - begin) >> 16));
        const uint32_t chunk = (end - begin + t - 1) / t;
        WorkerPool::instance().run(t, [&](uint32_t k) {   // (threads that are kept: a pass is a few milliseconds)
            const uint32_t b = std::min(end, begin + chunk * k), e = std::min(end, begin + chunk * (k + 1));
            if (b < e || k == 0u) fn(k, b, e);
        });
    }

    // What a node knows about its range before it looks at a single primitive: the parent computed it while partitioning.
    struct NodeInfo {
        Aabb box, cbox;
        uint32_t spheres = 0;
        void reset() {
            box.reset();
            cbox.reset();
            spheres = 0;
        }
        void add(const Rec& p) {
            box.grow4(p.lo, p.hi);
            const float c[4] = {p.centre(0), p.centre(1), p.centre(2), 0.0f};
            cbox.grow4(c, c);
            spheres += p.isSphere;
        }
        void merge(const NodeInfo& o) {
            if (o.box.lo[0] <= o.box.hi[0]) {
                box.grow(o.box);
                cbox.grow(o.cbox);
            }
            spheres += o.spheres;
        }
    };

    // bounds of a range by one pass over it (the root, and the rare splits that do not come out of the SAH partition)
    NodeInfo measure(uint32_t begin, uint32_t end) {
        NodeInfo info;
        info.reset();
        if ((end - begin) >= kTaskNode && wideThreads > 1) {
            std::vector<NodeInfo> parts(wideThreads);
            for (NodeInfo& p : parts) p.reset();
            forChunks(begin, end, [&](uint32_t k, uint32_t b, uint32_t e) {
                NodeInfo part;   // on the thread's own stack: neighbours in `parts` share cache lines
                part.reset();
                for (uint32_t i = b; i < e; ++i) part.add(recs[i]);
                parts[k] = part;
            });
            for (const NodeInfo& p : parts) info.merge(p);
        } else {
            for (uint32_t i = begin; i < end; ++i) info.add(recs[i]);
        }
        return info;
    }

    struct Bins {
        Aabb box[3][kBins];
        uint32_t count[3][kBins];
        void reset() {
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < kBins; ++b) {
                    box[a][b].reset();
                    count[a][b] = 0;
                }
            }
        }
    };

    // A node costs two passes over its primitives: one fills the bins of all three axes, one partitions the range and measures both
    // halves on the way (the first version took five: bounds, one binning pass per axis, partition - 7.7 of the 11.6 s a
    // 29 M-triangle scene took to prepare).
    // Decides one node: a leaf (returns false), or the place `mid` where its range is cut, with the range partitioned and both
    // halves measured.  `wide`: the passes are split over the host threads.
    bool split(uint32_t node, uint32_t begin, uint32_t end, uint32_t depth, const NodeInfo& info, bool wide, uint32_t* cursor, uint32_t& mid,
               NodeInfo& leftInfo, NodeInfo& rightInfo) {
        const Aabb& box = info.box;
        const Aabb& cbox = info.cbox;
        const uint32_t sphereCount = info.spheres;
        nodes[node].box = box;
        const uint32_t count = end - begin;
        const bool mixed = sphereCount != 0 && sphereCount != count;

        mid = begin;
        bool haveSplit = false;
        const bool forceBalanced = depth + log2Ceil(count) + 1 >= kDepthLimit;

        if (count == 1 || (count <= kLeafMax && !mixed && forceBalanced)) {
            makeLeaf(node, begin, end);
            return false;
        }

        if (!forceBalanced) {
            // binned SAH over the three axes, all bins in one pass
            float lo[3], scale[3];
            bool usable[3];
            for (int axis = 0; axis < 3; ++axis) {
                lo[axis] = cbox.lo[axis];
                usable[axis] = cbox.hi[axis] > cbox.lo[axis];
                scale[axis] = usable[axis] ? static_cast<float>(kBins) / (cbox.hi[axis] - cbox.lo[axis]) : 0.0f;
            }
            auto fill = [&](Bins& bins, uint32_t b, uint32_t e) {
                for (uint32_t i = b; i < e; ++i) {
                    const Rec& p = recs[i];
                    for (int axis = 0; axis < 3; ++axis) {
                        if (!usable[axis]) continue;
                        const int bin = std::min(kBins - 1, static_cast<int>((p.centre(axis) - lo[axis]) * scale[axis]));
                        bins.box[axis][bin].grow4(p.lo, p.hi);
                        ++bins.count[axis][bin];
                    }
                }
            };
            Bins bins;
            bins.reset();
            std::vector<Bins> parts;   // wide nodes: the bins of every chunk (kept: the partition reads its counts from them)
            if (wide) {
                parts.resize(wideThreads);
                for (Bins& p : parts) p.reset();   // chunks that do not exist stay empty
                forChunks(begin, end, [&](uint32_t k, uint32_t cb, uint32_t ce) {
                    Bins mine;
                    mine.reset();
                    fill(mine, cb, ce);
                    parts[k] = mine;
                });
                for (const Bins& p : parts) {
                    for (int axis = 0; axis < 3; ++axis) {
                        for (int b = 0; b < kBins; ++b) {
                            if (p.count[axis][b]) {
                                bins.box[axis][b].grow(p.box[axis][b]);
                                bins.count[axis][b] += p.count[axis][b];
                            }
                        }
                    }
                }
            } else {
                fill(bins, begin, end);
            }
            int bestAxis = -1, bestBin = -1;
            float bestCost = std::numeric_limits<float>::infinity();
            for (int axis = 0; axis < 3; ++axis) {
                if (!usable[axis]) continue;
                const Aabb* binBox = bins.box[axis];
                const uint32_t* binCount = bins.count[axis];
                float rightArea[kBins];
                uint32_t rightCount[kBins];
                Aabb acc;
                acc.reset();
                uint32_t n = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(binBox[b]);
                    n += binCount[b];
                    rightArea[b] = n ? acc.halfArea() : 0.0f;
                    rightCount[b] = n;
                }
                acc.reset();
                n = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(binBox[b]);
                    n += binCount[b];
                    if (n == 0 || rightCount[b + 1] == 0) continue;
                    const float cost = acc.halfArea() * static_cast<float>(n) + rightArea[b + 1] * static_cast<float>(rightCount[b + 1]);
                    if (cost < bestCost) {
                        bestCost = cost;
                        bestAxis = axis;
                        bestBin = b;
                    }
                }
            }
            if (bestAxis >= 0) {
                const float parentArea = std::max(box.halfArea(), 1e-30f);
                const float splitCost = 1.0f + bestCost / parentArea;   // c_trav = c_int = 1 (tinybvh defaults)
                if (count <= kLeafMax && !mixed && splitCost >= static_cast<float>(count)) {
                    makeLeaf(node, begin, end);
                    return false;
                }
                const float splitLo = cbox.lo[bestAxis];
                const float splitScale = static_cast<float>(kBins) / (cbox.hi[bestAxis] - splitLo);
                auto goesLeft = [&](const Rec& p) {
                    return std::min(kBins - 1, static_cast<int>((p.centre(bestAxis) - splitLo) * splitScale)) <= bestBin;
                };
                leftInfo.reset();
                rightInfo.reset();
                if (wide) {
                    // chunk-wise counts, prefix sums, stable scatter through the scratch array (which side a primitive lands on is all
                    // that matters: the subtrees re-partition their ranges anyway); the scatter measures both halves
                    // how many of a chunk go left is already in that chunk's bins (same chunks as the binning pass)
                    std::vector<uint32_t> lefts(wideThreads + 1u, 0u), sizes(wideThreads + 1u, 0u);
                    for (uint32_t k = 0; k < wideThreads; ++k) {
                        for (int b = 0; b < kBins; ++b) {
                            sizes[k + 1u] += parts[k].count[bestAxis][b];
                            if (b <= bestBin) lefts[k + 1u] += parts[k].count[bestAxis][b];
                        }
                    }
                    uint32_t totalLeft = 0;
                    for (uint32_t k = 1; k <= wideThreads; ++k) totalLeft += lefts[k];
                    std::vector<uint32_t> leftAt(wideThreads + 1u, 0u), rightAt(wideThreads + 1u, 0u);
                    for (uint32_t k = 0; k < wideThreads; ++k) {
                        leftAt[k + 1u] = leftAt[k] + lefts[k + 1u];
                        rightAt[k + 1u] = rightAt[k] + (sizes[k + 1u] - lefts[k + 1u]);
                    }
                    std::vector<NodeInfo> leftParts(wideThreads), rightParts(wideThreads);
                    for (uint32_t k = 0; k < wideThreads; ++k) {
                        leftParts[k].reset();
                        rightParts[k].reset();
                    }
                    forChunks(begin, end, [&](uint32_t k, uint32_t cb, uint32_t ce) {
                        uint32_t l = begin + leftAt[k], r = begin + totalLeft + rightAt[k];
                        NodeInfo li, ri;
                        li.reset();
                        ri.reset();
                        for (uint32_t i = cb; i < ce; ++i) {
                            const Rec& p = recs[i];
                            if (goesLeft(p)) {
                                scratch[l++] = p;
                                li.add(p);
                            } else {
                                scratch[r++] = p;
                                ri.add(p);
                            }
                        }
                        leftParts[k] = li;
                        rightParts[k] = ri;
                    });
                    forChunks(begin, end, [&](uint32_t, uint32_t cb, uint32_t ce) { std::memcpy(&recs[cb], &scratch[cb], static_cast<size_t>(ce - cb) * sizeof(Rec)); });
                    for (uint32_t k = 0; k < wideThreads; ++k) {
                        leftInfo.merge(leftParts[k]);
                        rightInfo.merge(rightParts[k]);
                    }
                    mid = begin + totalLeft;
                } else {
                    // std::partition's two-ended walk (so the order inside the halves, and with it the order of the primitives in the
                    // leaves, is what it always was), measuring both halves as the elements settle
                    auto settleLeft = [&](const Rec& p) { leftInfo.add(p); };
                    auto settleRight = [&](const Rec& p) { rightInfo.add(p); };
                    uint32_t first = begin, last = end;
                    while (true) {
                        bool finished = false;
                        while (true) {
                            if (first == last) {
                                finished = true;
                                break;
                            }
                            if (goesLeft(recs[first])) {
                                settleLeft(recs[first]);
                                ++first;
                            } else {
                                break;
                            }
                        }
                        if (finished) break;
                        --last;   // order[first] goes right and has not been counted yet
                        while (true) {
                            if (first == last) {
                                settleRight(recs[first]);
                                finished = true;
                                break;
                            }
                            if (!goesLeft(recs[last])) {
                                settleRight(recs[last]);
                                --last;
                            } else {
                                break;
                            }
                        }
                        if (finished) break;
                        std::swap(recs[first], recs[last]);
                        settleLeft(recs[first]);
                        settleRight(recs[last]);
                        ++first;
                    }
                    mid = first;
                }
                haveSplit = mid != begin && mid != end;
            }
        }
        if (!haveSplit) {
            if (mixed) {
                // never mix spheres and triangles in one leaf: separate the kinds first
                Rec* const it = std::partition(recs.data() + begin, recs.data() + end, [&](const Rec& p) { return p.isSphere == 0; });
                mid = static_cast<uint32_t>(it - recs.data());
            } else if (count <= kLeafMax && !forceBalanced) {
                makeLeaf(node, begin, end);
                return false;
            } else {
                // object median along the widest centroid axis (also the depth-bounding fallback)
                int axis = 0;
                float widest = -1.0f;
                for (int a = 0; a < 3; ++a) {
                    const float w = cbox.hi[a] - cbox.lo[a];
                    if (w > widest) {
                        widest = w;
                        axis = a;
                    }
                }
                mid = begin + count / 2;
                std::nth_element(recs.data() + begin, recs.data() + mid, recs.data() + end, [&](const Rec& a, const Rec& b) {
                    const float ca = a.centre(axis), cb = b.centre(axis);
                    return ca < cb || (ca == cb && a.id < b.id);
                });
            }
            leftInfo = measure(begin, mid);
            rightInfo = measure(mid, end);
        }
        const uint32_t left = allocPair(cursor);
        nodes[node].left = left;
        nodes[node].right = left + 1;
        nodes[node].count = 0;
        return true;
    }

    void sumChildren(uint32_t node) {
        const uint32_t left = nodes[node].left;
        nodes[node].internalNodes = 1u + nodes[left].internalNodes + nodes[left + 1].internalNodes;
        nodes[node].triPrims = nodes[left].triPrims + nodes[left + 1].triPrims;
        nodes[node].spherePrims = nodes[left].spherePrims + nodes[left + 1].spherePrims;
    }

    // one subtree, on the calling thread
    void build(uint32_t node, uint32_t begin, uint32_t end, uint32_t depth, const NodeInfo& info, uint32_t* cursor) {
        uint32_t mid;
        NodeInfo leftInfo, rightInfo;
        if (!split(node, begin, end, depth, info, false, cursor, mid, leftInfo, rightInfo)) return;
        build(nodes[node].left, begin, mid, depth + 1, leftInfo, cursor);
        build(nodes[node].right, mid, end, depth + 1, rightInfo, cursor);
        sumChildren(node);
    }

    // The whole tree.  Nodes of kTaskNode primitives and more are split one after the other with every pass spread over the threads
    // (the top eight levels of a 29 M-triangle tree touch every primitive once per level); the subtrees below them - a few hundred -
    // are a task list that a pool of threads works through, largest first.  (The first scheme gave each large split a thread while
    // any were free: a third of 32 threads busy on average.)
    static constexpr uint32_t kTaskNode = 1u << 17;
    struct Task {
        uint32_t node, begin, end, depth;
        NodeInfo info;
        uint32_t firstNode = 0;
    };
    void buildTop(uint32_t node, uint32_t begin, uint32_t end, uint32_t depth, const NodeInfo& info, std::vector<Task>& tasks,
                  std::vector<uint32_t>& topNodes) {
        if (end - begin < kTaskNode || wideThreads <= 1) {
            tasks.push_back({node, begin, end, depth, info, 0u});
            return;
        }
        uint32_t mid;
        NodeInfo leftInfo, rightInfo;
        if (!split(node, begin, end, depth, info, true, nullptr, mid, leftInfo, rightInfo)) return;
        buildTop(nodes[node].left, begin, mid, depth + 1, leftInfo, tasks, topNodes);
        buildTop(nodes[node].right, mid, end, depth + 1, rightInfo, tasks, topNodes);
        topNodes.push_back(node);   // children first
    }
    void buildAll(uint32_t count) {
        std::vector<Task> tasks;
        std::vector<uint32_t> topNodes;
        const auto t0 = std::chrono::steady_clock::now();
        buildTop(0, 0, count, 0, measure(0, count), tasks, topNodes);
        if (verbose) {
            std::fprintf(stderr, "[bvh] top of the tree: %zu nodes, %zu subtrees left, %.2f s\n", topNodes.size(), tasks.size(),
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        std::sort(tasks.begin(), tasks.end(), [](const Task& a, const Task& b) { return (a.end - a.begin) > (b.end - b.begin); });
        // A subtree over c primitives has at most 2 c - 1 nodes and its root already exists (its parent allocated it): it takes
        // exactly 2 c - 2 from a block of its own.  With T serial top splits (1 + 2 T nodes, T + 1 subtrees) the whole tree is
        // 1 + 2 T + 2 n - 2 (T + 1) = 2 n - 1 nodes whatever the shape of the top.
        uint64_t firstFree = nextNode.load();
        for (Task& t : tasks) {
            t.firstNode = static_cast<uint32_t>(firstFree);
            firstFree += 2ull * (t.end - t.begin) - 2ull;
        }
        if (firstFree > nodes.capacity) throw std::runtime_error("BVH builder: node pool too small for the task blocks");
        std::atomic<size_t> next{0};
        const uint32_t workers = static_cast<uint32_t>(std::min<size_t>(std::max(poolThreads, 1u), tasks.size()));
        runOnThreads(workers, [&](uint32_t) {
            while (true) {
                const size_t t = next.fetch_add(1);
                if (t >= tasks.size()) break;
                uint32_t cursor = tasks[t].firstNode;
                build(tasks[t].node, tasks[t].begin, tasks[t].end, tasks[t].depth, tasks[t].info, &cursor);
            }
        });
        for (uint32_t node : topNodes) sumChildren(node);
    }
    uint32_t poolThreads = 1;
    bool verbose = false;
};


struct Flattener {
    const Builder& b;
    FlatBvh& out;
    std::vector<uint32_t> primToTri, primToSphere;  // input index -> n-th triangle / sphere

    struct Stats {
        double sahCost = 0.0;
        uint32_t leafCount = 0, maxDepth = 0, maxLeafSize = 0;
    };
    struct Item {
        uint32_t temp, device, depth, triBase, sphereBase;
    };

    // the leaf's primitives land at their precomputed place in the leaf-order arrays
    uint32_t leafRef(const TempNode& n, uint32_t triBase, uint32_t sphereBase, Stats& st) {
        const bool sphere = n.sphereLeaf != 0;
        std::vector<uint32_t>& dst = sphere ? out.sphereOrder : out.triOrder;
        const uint32_t first = sphere ? sphereBase : triBase;
        for (uint32_t i = 0; i < n.count; ++i) {
            const uint32_t id = b.order[n.first + i];
            dst[first + i] = sphere ? primToSphere[id] : primToTri[id];
        }
        ++st.leafCount;
        st.maxLeafSize = std::max(st.maxLeafSize, n.count);
        return ptrk::kRefLeafBit | (sphere ? ptrk::kRefSphereBit : 0u) | ((n.count - 1u) << ptrk::kRefCountShift) | first;
    }

    void setChild(uint32_t deviceNode, int slot, const Aabb& box, uint32_t ref) {
        float* n = out.nodes.data() + static_cast<size_t>(deviceNode) * 16;
        float* lo = n + (slot == 0 ? 0 : 8);
        float* hi = n + (slot == 0 ? 4 : 12);
        std::memcpy(lo, box.lo, 12);
        std::memcpy(hi, box.hi, 12);
        if (slot == 0) {
            std::memcpy(n + 3, &ref, 4);
        } else {
            std::memcpy(n + 7, &ref, 4);
        }
    }

    // One internal node: its two child slots.  Device indices are preorder - the left subtree follows its parent, the right one
    // follows the left subtree - and every subtree owns a contiguous slice of the node and leaf-order arrays (sizes known from the
    // build), so disjoint subtrees can be written by different threads.
    void emit(const Item& it, double rootArea, Stats& st, std::vector<Item>& pending) {
        const TempNode& n = b.nodes[it.temp];
        st.sahCost += n.box.halfArea() / rootArea;
        const TempNode& l = b.nodes[n.left];
        const uint32_t kids[2] = {n.left, n.right};
        const uint32_t triBase[2] = {it.triBase, it.triBase + l.triPrims}, sphereBase[2] = {it.sphereBase, it.sphereBase + l.spherePrims};
        const uint32_t device[2] = {it.device + 1u, it.device + 1u + l.internalNodes};
        Item next[2];
        bool internal[2] = {false, false};
        for (int s = 0; s < 2; ++s) {
            const TempNode& c = b.nodes[kids[s]];
            if (c.count > 0) {
                setChild(it.device, s, c.box, leafRef(c, triBase[s], sphereBase[s], st));
                st.sahCost += (c.box.halfArea() / rootArea) * c.count;
                st.maxDepth = std::max(st.maxDepth, it.depth + 2);
            } else {
                setChild(it.device, s, c.box, device[s]);
                next[s] = {kids[s], device[s], it.depth + 1, triBase[s], sphereBase[s]};
                internal[s] = true;
            }
        }
        if (internal[1]) pending.push_back(next[1]);
        if (internal[0]) pending.push_back(next[0]);
    }

    void run(double rootArea, uint32_t threads) {
        const TempNode& root = b.nodes[0];
        out.triOrder.assign(root.triPrims, 0u);
        out.sphereOrder.assign(root.spherePrims, 0u);
        Stats total;
        if (root.count > 0) {  // whole scene fits one leaf: wrap it in a single-child root
            out.nodes.assign(16, 0.0f);
            out.nodeCount = 1;
            const uint32_t empty = ptrk::kRefEmpty;
            std::memcpy(out.nodes.data() + 3, &empty, 4);
            std::memcpy(out.nodes.data() + 7, &empty, 4);
            setChild(0, 0, root.box, leafRef(root, 0u, 0u, total));
            out.leafCount = total.leafCount;
            out.maxLeafSize = total.maxLeafSize;
            out.maxDepth = 1;
            out.sahCost = root.count;
            return;
        }
        out.nodeCount = root.internalNodes;
        out.nodes.assign(static_cast<size_t>(out.nodeCount) * 16, 0.0f);
        // the top of the tree serially (breadth first, until there are a few subtrees per thread), the subtrees in parallel
        std::vector<Item> tasks{{0u, 0u, 0u, 0u, 0u}};
        const size_t wanted = out.nodeCount >= (1u << 16) ? static_cast<size_t>(std::max(threads, 1u)) * 4u : 1u;
        while (tasks.size() < wanted) {
            // expand the largest subtree
            size_t pick = 0;
            for (size_t i = 1; i < tasks.size(); ++i) {
                if (b.nodes[tasks[i].temp].internalNodes > b.nodes[tasks[pick].temp].internalNodes) pick = i;
            }
            if (b.nodes[tasks[pick].temp].internalNodes < 1024u) break;
            const Item it = tasks[pick];
            tasks.erase(tasks.begin() + static_cast<std::ptrdiff_t>(pick));
            emit(it, rootArea, total, tasks);
        }
        std::vector<Stats> partial(tasks.size());
        std::atomic<size_t> nextTask{0};
        const uint32_t workers = tasks.size() > 1 ? std::min<uint32_t>(std::max(threads, 1u), static_cast<uint32_t>(tasks.size())) : 1u;
        runOnThreads(workers, [&](uint32_t) {
            std::vector<Item> stack;
            while (true) {
                const size_t t = nextTask.fetch_add(1);
                if (t >= tasks.size()) break;
                stack.assign(1, tasks[t]);
                while (!stack.empty()) {
                    const Item it = stack.back();
                    stack.pop_back();
                    emit(it, rootArea, partial[t], stack);
                }
            }
        });
        for (const Stats& p : partial) {   // in task order: the floating-point sum does not depend on the thread schedule
            total.sahCost += p.sahCost;
            total.leafCount += p.leafCount;
            total.maxDepth = std::max(total.maxDepth, p.maxDepth);
            total.maxLeafSize = std::max(total.maxLeafSize, p.maxLeafSize);
        }
        out.sahCost = total.sahCost;
        out.leafCount = total.leafCount;
        out.maxDepth = total.maxDepth;
        out.maxLeafSize = total.maxLeafSize;
    }
};

}  // namespace

void BuildFlatBvh(const std::vector<BuildPrim>& prims, FlatBvh& out, uint32_t threads, uint32_t leafMax) {
    out = FlatBvh{};
    const uint32_t n = static_cast<uint32_t>(prims.size());
    if (n == 0) return;

    // ---- oversize primitives ------------------------------------------------------------------------------------------
    // The 32 B nodes store boxes on ONE 16-bit grid over the bounds of the tree.  A handful of triangles far larger than the
    // rest (the floor and the light of a room around a finely tessellated statue) stretch that grid until a cell is coarse
    // next to the small triangles, and the scene falls back to 64 B float nodes - on exactly the scenes whose node array
    // does not fit any cache.  If taking at most kMaxOversize of the largest triangles out of the tree makes the grid fine
    // enough (same gate as the device side: cell <= 1/8 of the mean primitive extent), they are kept out: appended after the
    // leaf-order triangles and tested first by every ray.
    constexpr uint32_t kMaxOversize = 16u;   // one leaf reference holds up to 16 primitives
    std::vector<uint8_t> oversize(n, 0);
    uint32_t oversizeCount = 0;
    const Knobs knobs = readKnobs();
    if (n > 64u && !knobs.noOversize) {
        auto extentOf = [&](uint32_t i) { return std::max(std::max(prims[i].hi[0] - prims[i].lo[0], prims[i].hi[1] - prims[i].lo[1]), prims[i].hi[2] - prims[i].lo[2]); };
        // candidates: the kMaxOversize largest triangles, largest first
        std::vector<uint32_t> tris;
        tris.reserve(n);
        for (uint32_t i = 0; i < n; ++i) {
            if (!prims[i].isSphere) tris.push_back(i);
        }
        const size_t top = std::min<size_t>(kMaxOversize, tris.size());
        std::partial_sort(tris.begin(), tris.begin() + static_cast<std::ptrdiff_t>(top), tris.end(), [&](uint32_t a, uint32_t b_) {
            const float ea = extentOf(a), eb = extentOf(b_);
            return ea > eb || (ea == eb && a < b_);
        });
        std::vector<uint8_t> candidate(n, 0);
        for (size_t k = 0; k < top; ++k) candidate[tris[k]] = 1;
        // bounds and extent sum of everything that is not a candidate: one pass over the primitives
        Aabb restBox;
        restBox.reset();
        double restSum = 0.0;
        uint64_t restCount = 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (candidate[i]) continue;
            restBox.grow(prims[i].lo, prims[i].hi);
            restSum += extentOf(i);
            ++restCount;
        }
        // grid gate with the first k candidates out of the tree (k = 0: the tree as it would be built anyway)
        auto gridIsFine = [&](size_t k) {
            Aabb box = restBox;
            double sum = restSum;
            uint64_t kept = restCount;
            for (size_t c = k; c < top; ++c) {
                box.grow(prims[tris[c]].lo, prims[tris[c]].hi);
                sum += extentOf(tris[c]);
                ++kept;
            }
            if (kept == 0) return false;
            double maxCell = 0.0;
            for (int a = 0; a < 3; ++a) maxCell = std::max(maxCell, (static_cast<double>(box.hi[a]) - box.lo[a]) / 65531.0);
            return maxCell * 8.0 <= sum / static_cast<double>(kept);
        };
        if (!gridIsFine(0)) {
            for (size_t k = 1; k <= top; ++k) {
                // only whole groups of equally large triangles (the two halves of a rectangle go together)
                if (k < top && extentOf(tris[k]) == extentOf(tris[k - 1])) continue;
                if (gridIsFine(k)) {
                    for (size_t c = 0; c < k; ++c) oversize[tris[c]] = 1;
                    oversizeCount = static_cast<uint32_t>(k);
                    break;
                }
            }
        }
    }

    Builder b;
    b.kLeafMax = std::min(std::max(leafMax, 1u), ptrk::kMaxLeafPrims);
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    if (knobs.buildThreads != 0u) threads = knobs.buildThreads;   // test knob
    b.wideThreads = std::min(threads, 64u);
    b.poolThreads = std::min(threads, 64u);
    const uint32_t inTree = n - oversizeCount;
    b.recs.allocate(inTree);
    {
        // in input order, the oversize ones left out: chunk k starts at its first index minus the oversize primitives before it
        std::vector<uint32_t> skipped(b.wideThreads + 1u, 0u);
        b.forChunks(0, n, [&](uint32_t k, uint32_t cb, uint32_t ce) {
            uint32_t s = 0;
            for (uint32_t i = cb; i < ce; ++i) s += oversize[i];
            skipped[k + 1u] = s;
        });
        for (uint32_t k = 0; k < b.wideThreads; ++k) skipped[k + 1u] += skipped[k];
        b.forChunks(0, n, [&](uint32_t k, uint32_t cb, uint32_t ce) {
            uint32_t at = cb - skipped[k];
            for (uint32_t i = cb; i < ce; ++i) {
                if (oversize[i]) continue;
                Rec& r = b.recs[at++];
                for (int a = 0; a < 3; ++a) {
                    r.lo[a] = prims[i].lo[a];
                    r.hi[a] = prims[i].hi[a];
                }
                r.id = i;
                r.isSphere = prims[i].isSphere;
            }
        });
    }
    b.nodes.allocate(static_cast<size_t>(2) * inTree + 2u);   // 2 n - 1 nodes exactly (see buildAll)
    if (inTree >= Builder::kTaskNode && b.wideThreads > 1) b.scratch.allocate(inTree);
    const bool verbose = knobs.verboseBuild;
    b.verbose = verbose;
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (verbose) std::fprintf(stderr, "[bvh] %-10s %.2f s\n", what, std::chrono::duration<double>(now - tick).count());
        tick = now;
    };
    lap("setup");
    b.buildAll(inTree);
    b.scratch.release();
    b.order.resize(inTree);
    b.forChunks(0, inTree, [&](uint32_t, uint32_t cb, uint32_t ce) {
        for (uint32_t i = cb; i < ce; ++i) b.order[i] = b.recs[i].id;
    });
    b.recs.release();
    lap("sah build");

    Flattener f{b, out, {}, {}};
    f.primToTri.resize(n);
    f.primToSphere.resize(n);
    uint32_t tri = 0, sph = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (prims[i].isSphere) {
            f.primToSphere[i] = sph++;
        } else {
            f.primToTri[i] = tri++;
        }
    }
    f.run(std::max(static_cast<double>(b.nodes[0].box.halfArea()), 1e-30), std::min(threads, 32u));
    out.rootRef = 0u;
    if (oversizeCount > 0) {
        const uint32_t first = static_cast<uint32_t>(out.triOrder.size());
        for (uint32_t i = 0; i < n; ++i) {
            if (oversize[i]) out.triOrder.push_back(f.primToTri[i]);
        }
        out.oversizeRef = ptrk::kRefLeafBit | ((oversizeCount - 1u) << ptrk::kRefCountShift) | first;
    }
    lap("flatten");

    // 16-bit grid version of the nodes.  lo is rounded down and hi up, plus one cell of padding on each
    // side, so a quantised box always contains the float box (the traversal stays conservative).
    const Aabb& scene = b.nodes[0].box;
    double cell[3];
    for (int a = 0; a < 3; ++a) {
        const double extent = static_cast<double>(scene.hi[a]) - scene.lo[a];
        cell[a] = extent > 0.0 ? extent / 65531.0 : 1.0;          // cells 2..65533 span the scene, rest is padding
        out.gridCell[a] = static_cast<float>(cell[a]);
        out.gridOrigin[a] = static_cast<float>(scene.lo[a] - 2.0 * cell[a]);
    }
    out.qnodes.assign(static_cast<size_t>(out.nodeCount) * 8, 0u);
    auto quant = [&](float v, int axis, bool up) -> uint32_t {
        const double g = (static_cast<double>(v) - out.gridOrigin[axis]) / static_cast<double>(out.gridCell[axis]);
        const double q = up ? std::ceil(g) + 1.0 : std::floor(g) - 1.0;
        return static_cast<uint32_t>(std::min(std::max(q, 0.0), 65535.0));
    };
    auto quantiseRange = [&](uint32_t begin, uint32_t end) {
        for (uint32_t i = begin; i < end; ++i) {
            const float* n = out.nodes.data() + static_cast<size_t>(i) * 16;
            uint32_t* q = out.qnodes.data() + static_cast<size_t>(i) * 8;
            for (int c = 0; c < 2; ++c) {
                const float* lo = n + c * 8;
                const float* hi = n + c * 8 + 4;
                uint32_t ref;
                std::memcpy(&ref, n + (c == 0 ? 3 : 7), 4);
                uint32_t* w = q + c * 4;
                if (ref == ptrk::kRefEmpty) {
                    w[0] = w[1] = w[2] = 0u;
                } else {
                    w[0] = quant(lo[0], 0, false) | (quant(lo[1], 1, false) << 16);
                    w[1] = quant(lo[2], 2, false) | (quant(hi[0], 0, true) << 16);
                    w[2] = quant(hi[1], 1, true) | (quant(hi[2], 2, true) << 16);
                }
                w[3] = ref;
            }
        }
    };
    {
        const uint32_t workers = out.nodeCount >= (1u << 16) ? std::min(threads, 32u) : 1u;
        const uint32_t chunk = (out.nodeCount + workers - 1) / workers;
        runOnThreads(workers, [&](uint32_t t) {
            const uint32_t lo = std::min(out.nodeCount, chunk * t), hi = std::min(out.nodeCount, chunk * (t + 1));
            if (lo < hi) quantiseRange(lo, hi);
        });
    }
    double extentSum = 0.0;
    for (uint32_t i = 0; i < n; ++i) {
        const BuildPrim& p = prims[i];
        if (!oversize[i]) extentSum += std::max(std::max(p.hi[0] - p.lo[0], p.hi[1] - p.lo[1]), p.hi[2] - p.lo[2]);
    }
    out.meanPrimExtent = static_cast<float>(extentSum / std::max(inTree, 1u));
    lap("quantise");
}

namespace {

// the unused place of a wide node: an inverted box (lo = 65535, hi = 0 on every axis) - the wide step's ordered slab test (near plane
// by the sign of the direction) can never pass it, so the step needs no test of the reference
void emptyWidePlace(uint32_t* rec) {
    rec[0] = 0xFFFFFFFFu;
    rec[1] = 0x0000FFFFu;
    rec[2] = 0x00000000u;
    rec[3] = ptrk::kRefEmpty;
}

// The (at most four) child records of the wide node rooted at binary node n, as pointers into the quantised binary array.
// byArea: the internal child with the largest box is opened until four children stand (or only leaves do); otherwise both children of
// n are opened (every second level of the binary tree is collapsed).
struct WideChoice {
    const uint32_t* rec[4];
    uint32_t count = 0;
};
inline bool internalRef(uint32_t ref, uint32_t nodeCount) { return ref != ptrk::kRefEmpty && !(ref & ptrk::kRefLeafBit) && ref < nodeCount; }
inline double quantArea(const uint32_t* rec, const float* cell) {
    const double x = (static_cast<double>(rec[1] >> 16) - (rec[0] & 0xFFFFu)) * cell[0];
    const double y = (static_cast<double>(rec[2] & 0xFFFFu) - (rec[0] >> 16)) * cell[1];
    const double z = (static_cast<double>(rec[2] >> 16) - (rec[1] & 0xFFFFu)) * cell[2];
    return x * y + y * z + z * x;
}
WideChoice chooseWideChildren(const uint32_t* q, uint32_t nodeCount, const float* cell, uint32_t n, bool byArea) {
    WideChoice c;
    auto childrenOf = [&](uint32_t node, const uint32_t** dst) {
        uint32_t got = 0;
        for (uint32_t side = 0; side < 2u; ++side) {
            const uint32_t* rec = q + static_cast<size_t>(node) * 8u + side * 4u;
            if (rec[3] != ptrk::kRefEmpty) dst[got++] = rec;
        }
        return got;
    };
    c.count = childrenOf(n, c.rec);
    if (!byArea) {
        WideChoice g;
        for (uint32_t k = 0; k < c.count; ++k) {
            if (internalRef(c.rec[k][3], nodeCount)) g.count += childrenOf(c.rec[k][3], g.rec + g.count);
            else g.rec[g.count++] = c.rec[k];
        }
        return g;
    }
    while (c.count < 4u) {
        int best = -1;
        double bestArea = -1.0;
        for (uint32_t k = 0; k < c.count; ++k) {
            if (!internalRef(c.rec[k][3], nodeCount)) continue;
            const double a = quantArea(c.rec[k], cell);
            if (a > bestArea) {
                bestArea = a;
                best = static_cast<int>(k);
            }
        }
        if (best < 0) break;
        const uint32_t* two[2];
        const uint32_t got = childrenOf(c.rec[best][3], two);
        if (got == 0u) break;   // (a node without children: not produced by the builder)
        c.rec[best] = two[0];
        if (got > 1u) c.rec[c.count++] = two[1];
    }
    return c;
}

}  // namespace

uint32_t BuildWideNodes(const FlatBvh& bvh, WideCollapse how, std::unique_ptr<uint32_t[]>& wide, uint32_t* depthOut) {
    const uint32_t* q = bvh.qnodes.data();
    const uint32_t nodeCount = bvh.nodeCount;
    constexpr uint32_t kNotWide = 0xFFFFFFFFu;
    if (depthOut) *depthOut = 0u;
    if (nodeCount == 0 || bvh.qnodes.size() < static_cast<size_t>(nodeCount) * 8u) return 0;
    const bool byArea = how == WideCollapse::ByArea;
    // Which binary nodes root a wide node, and how deep the wide tree gets: the root does, and every internal child a wide node keeps.
    // A parent precedes its children in the (preorder) binary array, so one forward pass settles it.
    std::unique_ptr<uint32_t[]> wideIndex(new uint32_t[nodeCount]);
    std::vector<uint8_t> depth(nodeCount, 0);   // depth of the wide node rooted here (1 = the root), 0: not a wide root
    depth[0] = 1u;
    uint32_t maxDepth = 1u;
    for (uint32_t n = 0; n < nodeCount; ++n) {
        if (depth[n] == 0u) continue;
        const WideChoice c = chooseWideChildren(q, nodeCount, bvh.gridCell, n, byArea);
        for (uint32_t k = 0; k < c.count; ++k) {
            const uint32_t ref = c.rec[k][3];
            if (internalRef(ref, nodeCount)) {
                depth[ref] = static_cast<uint8_t>(std::min<uint32_t>(depth[n] + 1u, 255u));
                maxDepth = std::max<uint32_t>(maxDepth, depth[ref]);
            }
        }
    }
    if (depthOut) *depthOut = maxDepth;
    uint32_t wideCount = 0;
    for (uint32_t n = 0; n < nodeCount; ++n) wideIndex[n] = depth[n] ? wideCount++ : kNotWide;
    wide.reset(new uint32_t[static_cast<size_t>(wideCount) * 16u]);
    const uint32_t workers = nodeCount >= (1u << 16) ? std::min(32u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    auto collapse = [&](uint32_t begin, uint32_t end) {
        for (uint32_t n = begin; n < end; ++n) {
            if (wideIndex[n] == kNotWide) continue;
            uint32_t* w = wide.get() + static_cast<size_t>(wideIndex[n]) * 16u;
            const WideChoice c = chooseWideChildren(q, nodeCount, bvh.gridCell, n, byArea);
            for (uint32_t k = 0; k < c.count; ++k) {
                uint32_t* dst = w + 4u * k;
                std::memcpy(dst, c.rec[k], 16);
                if (internalRef(dst[3], nodeCount)) dst[3] = wideIndex[dst[3]];   // its wide node
            }
            for (uint32_t k = c.count; k < 4u; ++k) emptyWidePlace(w + 4u * k);
        }
    };
    const uint32_t chunk = (nodeCount + workers - 1u) / workers;
    runOnThreads(workers, [&](uint32_t k) {
        const uint32_t b = std::min(nodeCount, chunk * k), e = std::min(nodeCount, chunk * (k + 1u));
        if (b < e) collapse(b, e);
    });
    return wideCount;
}

}  // namespace ptr
