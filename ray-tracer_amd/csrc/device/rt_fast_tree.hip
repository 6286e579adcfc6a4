// rt_fast_tree.hip — host code only: builds the hierarchy of RT_TRAVERSE_FAST.
#include "rt_fast_tree.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

namespace rtd {
namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;

void store_box(float *lo, float *hi, const float *blo, const float *bhi) {
    for (int k = 0; k < 3; k++) {
        lo[k] = blo[k];
        hi[k] = bhi[k];
    }
}

// f64 bounds -> f32 bounds that contain them after growing by pad
void round_out(const BoxD &b, double pad, float *lo, float *hi) {
    for (int k = 0; k < 3; k++) {
        lo[k] = std::nextafterf((float)(b.lo[k] - pad), -INFINITY);
        hi[k] = std::nextafterf((float)(b.hi[k] + pad), INFINITY);
    }
}

uint32_t leaf_ref(uint32_t begin, uint32_t count) { return REF_LEAF | ((count - 1) << REF_CNT_SHIFT) | begin; }

// ---------------------------------------------------------------------------
// binned SAH
// ---------------------------------------------------------------------------
struct BuildPrim { // 32 bytes, partitioned in place
    float lo[3], hi[3];
    uint32_t id, pad;
};
struct TNode {
    uint32_t left, right; // pool indices; left == NONE marks a leaf
    uint32_t begin, count;
};

constexpr int BINS = 16;
// Cost model in units of one node step. Measured on MI355X (hippo scene, 16.7 M rays): small
// leaves win; (0, 1, 1, leaf <= 2) is 20 % faster than (0.7, 0.3, 1, leaf <= 8).
struct SahConfig {
    float cost_leaf_base = 0.0f, cost_per_prim = 1.0f, cost_step = 1.0f;
    uint32_t leaf_max = 2;
    uint32_t max_depth = FAST_TREE_MAX_DEPTH;
    // Ranges up to this many primitives also try every split position (a full sweep). Off: measured on the hippo
    // scene, sweeping ranges of up to 4,096 primitives is worth 1 % (6.94 instead of 7.14 node steps per ray before the
    // other candidates existed), sweeping everything doubles the build time of a 12 M primitive scene.
    uint32_t sweep_max = 0;
    uint32_t rotation_passes = 2; // passes of tree rotations over the finished tree
    float rotation_min_gain = 0.2f; // a rotation that makes its subtree one level higher must gain this share of the node's area
    bool split_off_large = true; // also try the few largest primitives of a range against all the others ...
    float large_fraction = 16.f; // ... when the largest one's box has at least 1/16 of the range's area
};

struct FBox {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; k++) {
            lo[k] = INFINITY;
            hi[k] = -INFINITY;
        }
    }
    void grow(const float *l, const float *h) {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], l[k]);
            hi[k] = std::max(hi[k], h[k]);
        }
    }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.f)) return 0.f; // empty
        return dx * dy + dy * dz + dz * dx;
    }
};

// levels a median-split subtree over c primitives still needs below its root
uint32_t levels_needed(uint32_t c, uint32_t leaf_max) {
    uint32_t leaves = (c + leaf_max - 1) / leaf_max, l = 0;
    while ((1u << l) < leaves) l++;
    return l;
}

struct SahBuilder {
    SahConfig cfg;
    BuildPrim *bp;
    std::unique_ptr<TNode[]> pool;
    std::atomic<uint32_t> next{1};

    struct Task {
        uint32_t node, begin, count, depth;
    };

    // Splits [begin, begin+count); returns the size of the first part, or 0 for "make a leaf".
    uint32_t split(uint32_t begin, uint32_t count, uint32_t depth) {
        BuildPrim *p = bp + begin;
        FBox bounds, cb;
        bounds.reset();
        cb.reset();
        // the few largest primitives of the range (by box area, then id), largest first
        constexpr uint32_t BIG = 8;
        struct Big {
            float area;
            uint32_t id;
        } big[BIG];
        uint32_t n_big = 0;
        const uint32_t want_big = cfg.split_off_large && count > 2 ? std::min(BIG, count - 1) : 0;
        auto box_area = [](const BuildPrim &q) {
            float dx = q.hi[0] - q.lo[0], dy = q.hi[1] - q.lo[1], dz = q.hi[2] - q.lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        for (uint32_t i = 0; i < count; i++) {
            bounds.grow(p[i].lo, p[i].hi);
            float c[3] = {0.5f * (p[i].lo[0] + p[i].hi[0]), 0.5f * (p[i].lo[1] + p[i].hi[1]),
                          0.5f * (p[i].lo[2] + p[i].hi[2])};
            cb.grow(c, c);
            if (want_big) {
                const Big b{box_area(p[i]), p[i].id};
                auto before = [](const Big &x, const Big &y) { return x.area > y.area || (x.area == y.area && x.id < y.id); };
                if (n_big < want_big || before(b, big[n_big - 1])) {
                    uint32_t k = n_big < want_big ? n_big++ : n_big - 1;
                    for (; k > 0 && before(b, big[k - 1]); k--) big[k] = big[k - 1];
                    big[k] = b;
                }
            }
        }
        float scale[3];
        bool usable[3];
        for (int a = 0; a < 3; a++) {
            float w = cb.hi[a] - cb.lo[a];
            usable[a] = w > 0.f && std::isfinite(w);
            scale[a] = usable[a] ? (float)BINS * (1.0f - 1e-6f) / w : 0.f;
        }
        FBox bins[3][BINS];
        uint32_t cnt[3][BINS];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < BINS; b++) {
                bins[a][b].reset();
                cnt[a][b] = 0;
            }
        auto bin_of = [&](const BuildPrim &q, int a) {
            float c = 0.5f * (q.lo[a] + q.hi[a]);
            int b = (int)((c - cb.lo[a]) * scale[a]);
            return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
        };
        for (uint32_t i = 0; i < count; i++)
            for (int a = 0; a < 3; a++) {
                if (!usable[a]) continue;
                int b = bin_of(p[i], a);
                bins[a][b].grow(p[i].lo, p[i].hi);
                cnt[a][b]++;
            }
        // large ranges: the classic area x count; ranges that fit one leaf: two leaves' cost
        const bool small = count <= cfg.leaf_max;
        if (small && depth + 1 >= cfg.max_depth) return 0;
        const float base = small ? cfg.cost_leaf_base : 0.f;
        float best = INFINITY;
        int best_axis = -1, best_plane = 0;
        uint32_t best_left = 0;
        for (int a = 0; a < 3; a++) {
            if (!usable[a]) continue;
            float right_area[BINS];
            uint32_t right_cnt[BINS];
            FBox acc;
            acc.reset();
            uint32_t c = 0;
            for (int b = BINS - 1; b > 0; b--) {
                acc.grow(bins[a][b].lo, bins[a][b].hi);
                c += cnt[a][b];
                right_area[b] = acc.area();
                right_cnt[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = 1; b < BINS; b++) { // plane b: bins [0, b) left, [b, BINS) right
                acc.grow(bins[a][b - 1].lo, bins[a][b - 1].hi);
                c += cnt[a][b - 1];
                if (c == 0 || right_cnt[b] == 0) continue;
                float cost = acc.area() * (base + cfg.cost_per_prim * (float)c) +
                             right_area[b] * (base + cfg.cost_per_prim * (float)right_cnt[b]);
                if (cost < best) {
                    best = cost;
                    best_axis = a;
                    best_plane = b;
                    best_left = c;
                }
            }
        }
        // Ranges of at most `sweep_max` primitives: every split position along every axis (primitives sorted by
        // centre), not only the 15 bin planes.
        int sweep_axis = -1;
        uint32_t sweep_left = 0;
        if (count <= cfg.sweep_max && count >= 2) {
            std::vector<BuildPrim> sorted(p, p + count);
            std::vector<float> right_area(count);
            for (int a = 0; a < 3; a++) {
                if (!usable[a]) continue;
                std::sort(sorted.begin(), sorted.end(), [a](const BuildPrim &x, const BuildPrim &y) {
                    const float cx = x.lo[a] + x.hi[a], cy = y.lo[a] + y.hi[a];
                    return cx < cy || (cx == cy && x.id < y.id);
                });
                FBox acc;
                acc.reset();
                for (uint32_t i = count - 1; i > 0; i--) {
                    acc.grow(sorted[i].lo, sorted[i].hi);
                    right_area[i] = acc.area();
                }
                acc.reset();
                for (uint32_t i = 1; i < count; i++) { // the first i primitives left
                    acc.grow(sorted[i - 1].lo, sorted[i - 1].hi);
                    float cost = acc.area() * (base + cfg.cost_per_prim * (float)i) +
                                 right_area[i] * (base + cfg.cost_per_prim * (float)(count - i));
                    if (cost < best) {
                        best = cost;
                        best_axis = a;
                        sweep_axis = a;
                        sweep_left = i;
                        best_left = i;
                    }
                }
            }
        }
        // A few primitives far larger than the rest (walls around a mesh) make every box they fall into as large
        // as themselves, whichever plane cuts the range: also try "the k largest apart from the others".
        uint32_t split_off = 0;
        if (n_big && !small && big[0].area * cfg.large_fraction >= bounds.area()) {
            FBox rest;
            rest.reset();
            const BuildPrim *big_prim[BIG] = {};
            for (uint32_t i = 0; i < count; i++) {
                uint32_t k = 0;
                while (k < n_big && p[i].id != big[k].id) k++;
                if (k < n_big) big_prim[k] = &p[i];
                else rest.grow(p[i].lo, p[i].hi);
            }
            // from "all n_big apart" down to "the largest alone": the others' box takes one primitive back each time
            float big_area[BIG];
            FBox acc;
            acc.reset();
            for (uint32_t k = 0; k < n_big; k++) {
                acc.grow(big_prim[k]->lo, big_prim[k]->hi);
                big_area[k] = acc.area();
            }
            for (uint32_t k = n_big; k >= 1; k--) { // the k largest apart
                if (count - k >= 1) {
                    float cost = big_area[k - 1] * (cfg.cost_per_prim * (float)k) + rest.area() * (cfg.cost_per_prim * (float)(count - k));
                    if (cost < best) {
                        best = cost;
                        best_axis = 3;
                        split_off = k;
                        best_left = count - k;
                        sweep_axis = -1;
                    }
                }
                rest.grow(big_prim[k - 1]->lo, big_prim[k - 1]->hi);
            }
        }
        if (small) {
            // split only if a node step plus two leaves is cheaper than one leaf
            float area = bounds.area();
            float as_leaf = area * (cfg.cost_leaf_base + cfg.cost_per_prim * (float)count);
            float as_split = best_axis >= 0 ? area * cfg.cost_step + best : INFINITY;
            if (!(as_split < as_leaf)) return 0;
        }
        uint32_t nl = 0;
        bool median = best_axis < 0;
        if (!median) {
            const uint32_t larger = std::max(best_left, count - best_left);
            if (depth + 1 + levels_needed(larger, cfg.leaf_max) >= cfg.max_depth) median = true; // keep the depth bounded
        }
        if (!median && split_off) {
            BuildPrim *mid = std::partition(p, p + count, [&](const BuildPrim &q) {
                for (uint32_t k = 0; k < split_off; k++)
                    if (q.id == big[k].id) return false;
                return true;
            });
            nl = (uint32_t)(mid - p);
        } else if (!median && sweep_axis >= 0) {
            const int a = sweep_axis;
            std::sort(p, p + count, [a](const BuildPrim &x, const BuildPrim &y) {
                const float cx = x.lo[a] + x.hi[a], cy = y.lo[a] + y.hi[a];
                return cx < cy || (cx == cy && x.id < y.id);
            });
            nl = sweep_left;
        } else if (!median) {
            int a = best_axis, plane = best_plane;
            BuildPrim *mid = std::partition(p, p + count, [&](const BuildPrim &q) { return bin_of(q, a) < plane; });
            nl = (uint32_t)(mid - p);
            if (nl == 0 || nl == count) median = true;
        }
        if (median) {
            int a = 0;
            float w = -1.f;
            for (int k = 0; k < 3; k++)
                if (usable[k] && cb.hi[k] - cb.lo[k] > w) {
                    w = cb.hi[k] - cb.lo[k];
                    a = k;
                }
            nl = count / 2;
            std::nth_element(p, p + nl, p + count, [a](const BuildPrim &x, const BuildPrim &y) {
                return x.lo[a] + x.hi[a] < y.lo[a] + y.hi[a];
            });
        }
        return nl;
    }

    // Builds the subtree of pool node t.node; tasks larger than `defer` are split once
    // and their halves appended to `out` instead of being recursed into.
    void build(const Task &t, uint32_t defer, std::vector<Task> *out) {
        TNode &n = pool[t.node];
        n.begin = t.begin;
        n.count = t.count;
        n.left = n.right = NONE;
        if (t.count <= 1) return;
        uint32_t nl = split(t.begin, t.count, t.depth);
        if (nl == 0) return;
        uint32_t kids = next.fetch_add(2);
        n.left = kids;
        n.right = kids + 1;
        Task a{kids, t.begin, nl, t.depth + 1}, b{kids + 1, t.begin + nl, t.count - nl, t.depth + 1};
        if (out && t.count > defer) {
            out->push_back(a);
            out->push_back(b);
        } else {
            build(a, defer, out);
            build(b, defer, out);
        }
    }
};

} // namespace

FastTree build_fast_tree_sah(const std::vector<BoxD> &slot_box, double pad, uint32_t top_levels, unsigned threads) {
    FastTree ft;
    const uint32_t n = (uint32_t)slot_box.size();
    if (n == 0) {
        ft.fnodes.emplace_back();
        std::memset(&ft.fnodes[0], 0, sizeof(FNode));
        return ft;
    }
    SahConfig cfg;
    if (const char *tune = std::getenv("RT_TUNE_SAH")) { // "leaf_base,per_prim,step,leaf_max" for experiments
        float b = 0, p = 0, st = 0;
        unsigned lm = 0;
        if (std::sscanf(tune, "%f,%f,%f,%u", &b, &p, &st, &lm) == 4 && p > 0 && st > 0 && lm >= 1 && lm <= FAST_LEAF_MAX) {
            cfg.cost_leaf_base = b;
            cfg.cost_per_prim = p;
            cfg.cost_step = st;
            cfg.leaf_max = lm;
        }
    }
    // 30 binary levels collapse into at most 15 wide ones, which the walk's 16-entry stack takes at no cost in
    // occupancy. Config 5 (11.9 M primitives, natural SAH depth 34), batch / render in M queries/s with the
    // cheapest collapse: cap 26: 927 / 1867, 28: 952 / 1997, 30: 974 / 2029; with 32 and more the stack grows
    // and a wavefront per SIMD is lost (1005 / 1864 at 32).
    cfg.max_depth = levels_needed(n, cfg.leaf_max) + 3 <= 30 ? 30 : FAST_TREE_MAX_DEPTH;
    if (const char *tune = std::getenv("RT_TUNE_SPLIT_OFF")) { // experiments: 0 = off, else the area fraction's inverse
        cfg.split_off_large = std::atof(tune) > 0.0;
        if (cfg.split_off_large) cfg.large_fraction = (float)std::atof(tune);
    }
    if (const char *tune = std::getenv("RT_TUNE_ROTATIONS")) { // "passes,growth" for experiments
        unsigned ps = 0;
        float mg = 0.f;
        if (std::sscanf(tune, "%u,%f", &ps, &mg) >= 1) cfg.rotation_passes = ps, cfg.rotation_min_gain = mg;
    }
    if (const char *tune = std::getenv("RT_TUNE_SWEEP")) cfg.sweep_max = (uint32_t)std::max(0, std::atoi(tune)); // experiments
    if (const char *tune = std::getenv("RT_TUNE_MAX_DEPTH")) { // experiments
        unsigned md = (unsigned)std::atoi(tune);
        if (md >= 8 && md <= FAST_TREE_MAX_DEPTH) cfg.max_depth = md;
    }
    std::vector<BuildPrim> prims(n);
    for (uint32_t s = 0; s < n; s++) {
        round_out(slot_box[s], pad, prims[s].lo, prims[s].hi);
        prims[s].id = s;
        prims[s].pad = 0;
    }
    SahBuilder sb;
    sb.cfg = cfg;
    sb.bp = prims.data();
    sb.pool.reset(new TNode[2 * (size_t)n + 2]);

    // top of the tree on this thread until there are enough independent subtrees, then a pool
    std::vector<SahBuilder::Task> queue, ready;
    queue.push_back({0, 0, n, 0});
    threads = std::max(1u, std::min(threads, 64u));
    const uint32_t defer = threads > 1 ? std::max<uint32_t>(4096, n / (8 * threads)) : NONE;
    while (!queue.empty()) {
        std::vector<SahBuilder::Task> nextq;
        for (const auto &t : queue) {
            if (t.count > defer) sb.build(t, defer, &nextq);
            else ready.push_back(t);
        }
        queue.swap(nextq);
    }
    if (threads > 1 && ready.size() > 1) {
        std::sort(ready.begin(), ready.end(), [](const auto &a, const auto &b) { return a.count > b.count; });
        std::atomic<size_t> cursor{0};
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < threads; w++)
            pool.emplace_back([&]() {
                for (;;) {
                    size_t k = cursor.fetch_add(1);
                    if (k >= ready.size()) return;
                    sb.build(ready[k], NONE, nullptr);
                }
            });
        for (auto &th : pool) th.join();
    } else {
        for (const auto &t : ready) sb.build(t, NONE, nullptr);
    }
    const uint32_t used = sb.next.load();

    ft.ref_of_fast.resize(n);
    for (uint32_t i = 0; i < n; i++) ft.ref_of_fast[i] = prims[i].id;

    // boxes bottom-up: a child is always allocated after its parent
    std::vector<float> box(6 * (size_t)used);
    for (uint32_t k = used; k-- > 0;) {
        const TNode &t = sb.pool[k];
        float *lo = &box[6 * (size_t)k], *hi = lo + 3;
        FBox b;
        b.reset();
        if (t.left == NONE) {
            for (uint32_t i = t.begin; i < t.begin + t.count; i++) b.grow(prims[i].lo, prims[i].hi);
        } else {
            b.grow(&box[6 * (size_t)t.left], &box[6 * (size_t)t.left + 3]);
            b.grow(&box[6 * (size_t)t.right], &box[6 * (size_t)t.right + 3]);
        }
        store_box(lo, hi, b.lo, b.hi);
    }
    std::vector<BuildPrim>().swap(prims);

    // Rotations (Kensler 2008): where swapping a node's child with a grandchild on the other side makes the box
    // of the one inner node that changes smaller, the tree costs less (every other box stays as it is). Passes
    // from the leaves up. Nearly every rotation that gains pushes the larger subtree one level down, and a deeper
    // tree costs the collapse into 15 wide levels more than small gains return (measured, hippo scene: one
    // unrestricted pass 5.28 node steps per ray and 14 wide levels, two passes 6.16 and 15), so a rotation that
    // makes its subtree higher must gain a fifth of its node's surface: what is left are the few at the top that
    // the surface area heuristic misjudges (a range of a few walls and a mesh: it prices the mesh as one leaf).
    // Hippo: root {mesh | 4 walls} becomes {3 walls | {wall, mesh}}, 5.62 -> 5.25 node steps per ray.
    if (cfg.rotation_passes && sb.pool[0].left != NONE) {
        TNode *pool = sb.pool.get();
        std::vector<uint8_t> height(used, 0); // inner levels below and including the node; 0 for a leaf
        for (uint32_t k = used; k-- > 0;)
            if (pool[k].left != NONE) height[k] = (uint8_t)(1 + std::max(height[pool[k].left], height[pool[k].right]));
        auto bx = [&](uint32_t k) { return &box[6 * (size_t)k]; };
        auto union_area = [&](uint32_t a, uint32_t b) {
            FBox u;
            u.reset();
            u.grow(bx(a), bx(a) + 3);
            u.grow(bx(b), bx(b) + 3);
            return u.area();
        };
        auto refit = [&](uint32_t k) {
            FBox u;
            u.reset();
            u.grow(bx(pool[k].left), bx(pool[k].left) + 3);
            u.grow(bx(pool[k].right), bx(pool[k].right) + 3);
            store_box(bx(k), bx(k) + 3, u.lo, u.hi);
            height[k] = (uint8_t)(1 + std::max(height[pool[k].left], height[pool[k].right]));
        };
        struct Frame {
            uint32_t node, depth;
            bool expanded;
        };
        for (uint32_t pass = 0; pass < cfg.rotation_passes; pass++) {
            uint64_t applied = 0;
            std::vector<Frame> todo{{0u, 1u, false}};
            while (!todo.empty()) {
                Frame fr = todo.back();
                todo.pop_back();
                TNode &t = pool[fr.node];
                if (t.left == NONE) continue;
                if (!fr.expanded) { // children first
                    todo.push_back({fr.node, fr.depth, true});
                    todo.push_back({t.left, fr.depth + 1, false});
                    todo.push_back({t.right, fr.depth + 1, false});
                    continue;
                }
                refit(fr.node); // the children may have changed
                // candidates: (inner child c, its grandchild g that moves up, the other child o that moves down)
                float best_gain = 0.f;
                int best = -1;
                const float node_area = FBox{{bx(fr.node)[0], bx(fr.node)[1], bx(fr.node)[2]}, {bx(fr.node)[3], bx(fr.node)[4], bx(fr.node)[5]}}.area();
                uint32_t *side[2] = {&t.left, &t.right};
                for (int s2 = 0; s2 < 2; s2++) {
                    const uint32_t c = *side[s2], o = *side[1 - s2];
                    if (pool[c].left == NONE) continue;
                    const float before = FBox{{bx(c)[0], bx(c)[1], bx(c)[2]}, {bx(c)[3], bx(c)[4], bx(c)[5]}}.area();
                    for (int g2 = 0; g2 < 2; g2++) {
                        const uint32_t g = g2 ? pool[c].right : pool[c].left, stay = g2 ? pool[c].left : pool[c].right;
                        // o takes g's place under c, g takes o's place under the node
                        const float gain = before - union_area(o, stay);
                        const uint32_t new_height = std::max<uint32_t>(1u + height[g], 2u + std::max(height[o], height[stay]));
                        const bool grows = new_height > (uint32_t)height[fr.node];
                        if (gain > best_gain && fr.depth - 1 + new_height <= cfg.max_depth &&
                            (!grows || gain >= cfg.rotation_min_gain * node_area)) {
                            best_gain = gain;
                            best = 2 * s2 + g2;
                        }
                    }
                }
                if (best >= 0) {
                    const int s2 = best >> 1, g2 = best & 1;
                    const uint32_t c = *side[s2], o = *side[1 - s2];
                    uint32_t &gslot = g2 ? pool[c].right : pool[c].left;
                    const uint32_t g = gslot;
                    gslot = o;
                    *side[1 - s2] = g;
                    refit(c);
                    refit(fr.node);
                    applied++;
                }
            }
            if (applied == 0) break;
        }
    }

    if (sb.pool[0].left == NONE) { // a single leaf
        ft.root_ref = leaf_ref(0, n);
        ft.fnodes.emplace_back();
        std::memset(&ft.fnodes[0], 0, sizeof(FNode));
        return ft;
    }

    // layout: the first top_levels levels breadth first, every subtree below depth first
    uint32_t inner = 0;
    for (uint32_t k = 0; k < used; k++)
        if (sb.pool[k].left != NONE) inner++;
    ft.fnodes.resize(inner);
    std::vector<uint32_t> index_of(used, NONE); // pool node -> fnode index
    uint32_t emitted = 0;
    std::vector<std::pair<uint32_t, uint32_t>> frontier{{0u, 1u}}, below; // (pool node, depth)
    for (uint32_t level = 0; level < top_levels && !frontier.empty(); level++) {
        std::vector<std::pair<uint32_t, uint32_t>> nxt;
        for (auto [k, depth] : frontier) {
            index_of[k] = emitted++;
            ft.max_depth = std::max(ft.max_depth, depth);
            for (uint32_t c : {sb.pool[k].left, sb.pool[k].right})
                if (sb.pool[c].left != NONE) nxt.push_back({c, depth + 1});
        }
        frontier.swap(nxt);
    }
    ft.top_nodes = emitted;
    std::vector<std::pair<uint32_t, uint32_t>> stack;
    for (auto it = frontier.rbegin(); it != frontier.rend(); ++it) stack.push_back(*it);
    while (!stack.empty()) {
        auto [k, depth] = stack.back();
        stack.pop_back();
        index_of[k] = emitted++;
        ft.max_depth = std::max(ft.max_depth, depth);
        const TNode &t = sb.pool[k];
        if (sb.pool[t.right].left != NONE) stack.push_back({t.right, depth + 1});
        if (sb.pool[t.left].left != NONE) stack.push_back({t.left, depth + 1});
    }
    for (uint32_t k = 0; k < used; k++) {
        const TNode &t = sb.pool[k];
        if (t.left == NONE) continue;
        FNode &f = ft.fnodes[index_of[k]];
        std::memset(&f, 0, sizeof f);
        auto ref = [&](uint32_t c) {
            const TNode &ch = sb.pool[c];
            return ch.left == NONE ? leaf_ref(ch.begin, ch.count) : index_of[c];
        };
        store_box(f.lo0, f.hi0, &box[6 * (size_t)t.left], &box[6 * (size_t)t.left + 3]);
        store_box(f.lo1, f.hi1, &box[6 * (size_t)t.right], &box[6 * (size_t)t.right + 3]);
        f.ref0 = ref(t.left);
        f.ref1 = ref(t.right);
    }
    ft.root_ref = 0;
    return ft;
}

// ---------------------------------------------------------------------------
// reference topology
// ---------------------------------------------------------------------------
namespace {

struct RefTopology {
    const rt_scene_desc &d;
    const std::vector<BoxD> &slot_box;
    double pad;
    FastTree &ft;

    static BoxD empty() {
        BoxD b;
        for (int k = 0; k < 3; k++) {
            b.lo[k] = INFINITY;
            b.hi[k] = -INFINITY;
        }
        return b;
    }
    static void grow(BoxD &b, const BoxD &o) {
        for (int k = 0; k < 3; k++) {
            b.lo[k] = std::min(b.lo[k], o.lo[k]);
            b.hi[k] = std::max(b.hi[k], o.hi[k]);
        }
    }
    void fill(uint32_t idx, uint32_t r0, const BoxD &b0, uint32_t r1, const BoxD &b1) {
        FNode &f = ft.fnodes[idx];
        std::memset(&f, 0, sizeof f);
        round_out(b0, pad, f.lo0, f.hi0);
        round_out(b1, pad, f.lo1, f.hi1);
        f.ref0 = r0;
        f.ref1 = r1;
    }
    // a run of slots -> leaf reference, or a small balanced subtree when it is too long
    uint32_t build_range(uint32_t begin, uint32_t count, BoxD &box, uint32_t depth) {
        if (count <= FAST_LEAF_MAX) {
            box = empty();
            for (uint32_t s = begin; s < begin + count; s++) grow(box, slot_box[s]);
            return leaf_ref(begin, count);
        }
        uint32_t idx = (uint32_t)ft.fnodes.size();
        ft.fnodes.emplace_back();
        ft.max_depth = std::max(ft.max_depth, depth + 1);
        uint32_t half = count / 2;
        BoxD b0, b1;
        uint32_t r0 = build_range(begin, half, b0, depth + 1);
        uint32_t r1 = build_range(begin + half, count - half, b1, depth + 1);
        fill(idx, r0, b0, r1, b1);
        box = b0;
        grow(box, b1);
        return idx;
    }
    uint32_t build(uint32_t node, BoxD &box, uint32_t depth) {
        const rt_node &n = d.nodes[node];
        if (n.left == 0) return build_range(n.slot_begin, n.slot_count, box, depth);
        uint32_t idx = (uint32_t)ft.fnodes.size();
        ft.fnodes.emplace_back();
        ft.max_depth = std::max(ft.max_depth, depth + 1);
        BoxD b0, b1;
        uint32_t r0 = build(node + 1, b0, depth + 1); // the reference visits the right child first
        uint32_t r1 = build(n.left, b1, depth + 1);
        fill(idx, r0, b0, r1, b1);
        box = b0;
        grow(box, b1);
        return idx;
    }
};

} // namespace

FastTree build_fast_tree_reference(const rt_scene_desc &d, const std::vector<BoxD> &slot_box, double pad) {
    FastTree ft;
    ft.ref_of_fast.resize(slot_box.size());
    for (uint32_t s = 0; s < slot_box.size(); s++) ft.ref_of_fast[s] = s;
    ft.fnodes.reserve(d.n_nodes / 2 + 1);
    if (!slot_box.empty()) {
        RefTopology rt{d, slot_box, pad, ft};
        BoxD root;
        ft.root_ref = rt.build(0, root, 0);
    }
    if (ft.fnodes.empty()) {
        ft.fnodes.emplace_back();
        std::memset(&ft.fnodes[0], 0, sizeof(FNode));
    }
    return ft;
}

} // namespace rtd
