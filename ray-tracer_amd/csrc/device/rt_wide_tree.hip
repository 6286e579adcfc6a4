// rt_wide_tree.hip — host code only: collapses a binary build tree into the 4-wide hierarchy
// RT_TRAVERSE_FAST walks.
//
// A wide node takes the place of a binary node and some of its descendants: its two to four children
// are a cut through the binary subtree below it. Two rules choose the cuts:
//  * the cheapest collapse: the cuts that minimise the sum of the surfaces of all wide nodes, i.e. the
//    expected number of node steps of a random ray (dynamic programme over "this binary subtree in i places
//    of a wide node", i = 1..3);
//  * level skipping: a node's children are the binary node's grandchildren (a binary child that is a leaf
//    stays a leaf child), so every root-to-leaf path loses every other level and the wide depth is at most
//    ceil(binary depth / 2) whatever the shape of the tree.
// The first is used wherever its result fits into the depth the walk's stack takes at no cost, the second
// below the nodes where it does not.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "rt_fast_tree.hpp"

namespace rtd {
namespace {

struct Child {
    uint32_t ref; // binary reference: inner index or leaf reference
    float lo[3], hi[3];
};

float half_area(const Child &c) {
    const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

struct Collapser {
    const FastTree &bin;
    WideTree &out;
    // Cheapest way (sum of the surfaces of all wide nodes below, i.e. expected node steps) to put the
    // subtree of a binary inner node into i = 1, 2, 3 places of a wide node: cost[3*b + i-1]. In one place
    // the node becomes a wide node itself and hands its four places to its two children; in more places it
    // may also dissolve, its children sharing the places. split[4*b + j-1] = places the FIRST child gets
    // when the node's children share j places (j = 2..4); dissolve bit j-2 of flags[b]: with j places,
    // sharing them is cheaper than using j-1. Empty when the level-skipping rule is used instead.
    // levels[3*b + i-1]: wide levels below (and including) the wide nodes of that cheapest arrangement;
    // height[b]: binary levels of the subtree (what level skipping would halve).
    std::vector<float> cost;
    std::vector<uint8_t> split, flags, levels, height;

    void children_of(uint32_t binary_index, Child c[2]) const {
        const FNode &f = bin.fnodes[binary_index];
        c[0].ref = f.ref0;
        c[1].ref = f.ref1;
        for (int k = 0; k < 3; k++) {
            c[0].lo[k] = f.lo0[k]; c[0].hi[k] = f.hi0[k];
            c[1].lo[k] = f.lo1[k]; c[1].hi[k] = f.hi1[k];
        }
    }

    static bool is_inner(uint32_t ref) { return ref != REF_NONE && !(ref & REF_LEAF); }
    float cost_of(uint32_t ref, int places) const { return is_inner(ref) ? cost[3 * (size_t)ref + places - 1] : 0.0f; }
    uint32_t levels_of(uint32_t ref, int places) const { return is_inner(ref) ? levels[3 * (size_t)ref + places - 1] : 0u; }
    uint32_t height_of(uint32_t ref) const { return is_inner(ref) ? height[ref] : 0u; }

    // fills cost/split/flags for the subtree of inner node b whose box has half-surface `area`
    void plan(uint32_t b, float area) {
        Child c[2];
        children_of(b, c);
        for (int s = 0; s < 2; s++)
            if (is_inner(c[s].ref)) plan(c[s].ref, half_area(c[s]));
        height[b] = (uint8_t)(1u + std::max(height_of(c[0].ref), height_of(c[1].ref)));
        float share[5] = {0, 0, 0, 0, 0}; // share[j]: the two children in j places
        uint32_t share_levels[5] = {0, 0, 0, 0, 0};
        for (int j = 2; j <= 4; j++) {
            float best = INFINITY;
            int best_k = 1;
            for (int k = 1; k < j; k++) {
                if (k > 3 || j - k > 3) continue;
                const float v = cost_of(c[0].ref, k) + cost_of(c[1].ref, j - k);
                if (v < best) {
                    best = v;
                    best_k = k;
                }
            }
            share[j] = best;
            share_levels[j] = std::max(levels_of(c[0].ref, best_k), levels_of(c[1].ref, j - best_k));
            split[4 * (size_t)b + j - 1] = (uint8_t)best_k;
        }
        float *cb = &cost[3 * (size_t)b];
        uint8_t *lb = &levels[3 * (size_t)b];
        cb[0] = area + share[4];
        lb[0] = (uint8_t)(1u + share_levels[4]);
        uint8_t fl = 0;
        for (int i = 2; i <= 3; i++) {
            if (share[i] < cb[i - 2]) {
                cb[i - 1] = share[i];
                lb[i - 1] = (uint8_t)share_levels[i];
                fl |= (uint8_t)(1u << (i - 2));
            } else {
                cb[i - 1] = cb[i - 2];
                lb[i - 1] = lb[i - 2];
            }
        }
        flags[b] = fl;
    }

    // the children that `child` contributes to a wide node when it is given `places` of it
    void place(const Child &child, int places, Child kids[4], uint32_t &n) const {
        if (child.ref == REF_NONE) return;
        if (!is_inner(child.ref)) {
            kids[n++] = child;
            return;
        }
        while (places > 1 && !(flags[child.ref] & (1u << (places - 2)))) places--;
        if (places == 1) {
            kids[n++] = child; // a wide node of its own
            return;
        }
        Child c[2];
        children_of(child.ref, c);
        const int k = split[4 * (size_t)child.ref + places - 1];
        place(c[0], k, kids, n);
        place(c[1], places - k, kids, n);
    }

    // Emits the wide node replacing binary node `binary_index`; returns its index, and through
    // `need` the number of wide nodes on the longest path from it down (itself included).
    // `budget`: wide levels this subtree may use. The cheapest arrangement is taken where it fits; where it
    // does not, this node takes its grandchildren (level skipping), which always fits as long as the budget
    // covers half the subtree's binary height, and the question is asked again one level down.
    uint32_t emit(uint32_t binary_index, uint32_t &need, uint32_t budget) {
        const uint32_t idx = (uint32_t)out.nodes.size();
        if (idx > WREF_INDEX_MASK) throw std::length_error("fast tree has more than 2^25 wide nodes");
        out.nodes.emplace_back();
        Child two[2], kids[4];
        uint32_t n = 0;
        children_of(binary_index, two);
        if (!cost.empty() && levels[3 * (size_t)binary_index] <= budget) {
            const int k = split[4 * (size_t)binary_index + 3];
            place(two[0], k, kids, n);
            place(two[1], 4 - k, kids, n);
        } else {
            for (int s = 0; s < 2; s++) {
                if (two[s].ref == REF_NONE) continue;
                if (two[s].ref & REF_LEAF) {
                    kids[n++] = two[s];
                } else {
                    Child g[2];
                    children_of(two[s].ref, g);
                    for (int t = 0; t < 2; t++)
                        if (g[t].ref != REF_NONE) kids[n++] = g[t];
                }
            }
        }
        // the walk tests places 0 and 1 without looking at their references
        if (n < 2) throw std::logic_error("binary inner node with fewer than two children");
        uint32_t refs[4] = {REF_NONE, REF_NONE, REF_NONE, REF_NONE};
        uint32_t deepest = 0;
        for (uint32_t k = 0; k < n; k++) {
            if (kids[k].ref & REF_LEAF) {
                refs[k] = kids[k].ref;
            } else {
                uint32_t child_need = 0;
                uint32_t child = emit(kids[k].ref, child_need, budget > 1 ? budget - 1 : 1);
                deepest = std::max(deepest, child_need);
                refs[k] = child | (std::min(child_need, WREF_NEED_MAX) << WREF_NEED_SHIFT);
            }
        }
        WNode &w = out.nodes[idx]; // after the recursion: the vector may have moved
        std::memset(&w, 0, sizeof w);
        for (uint32_t k = 0; k < 4; k++) {
            w.ref[k] = refs[k];
            const bool used = k < n;
            // an unused child gets an inverted box; the walk also checks the reference
            w.lox[k] = used ? kids[k].lo[0] : INFINITY; w.hix[k] = used ? kids[k].hi[0] : -INFINITY;
            w.loy[k] = used ? kids[k].lo[1] : INFINITY; w.hiy[k] = used ? kids[k].hi[1] : -INFINITY;
            w.loz[k] = used ? kids[k].lo[2] : INFINITY; w.hiz[k] = used ? kids[k].hi[2] : -INFINITY;
        }
        need = deepest + 1;
        return idx;
    }
};

} // namespace

WideTree collapse_to_wide(const FastTree &bin, uint32_t depth_wanted) {
    WideTree wt;
    wt.root_ref = bin.root_ref;
    if (bin.root_ref == REF_NONE || (bin.root_ref & REF_LEAF)) { // empty scene or a single leaf
        wt.nodes.emplace_back();
        std::memset(&wt.nodes[0], 0, sizeof(WNode));
        for (int k = 0; k < 4; k++) wt.nodes[0].ref[k] = REF_NONE;
        wt.depth = 0;
        return wt;
    }
    // The cheapest collapse wherever it fits into `depth_wanted` levels, level skipping where it does not
    // (everywhere, if even level skipping needs more: then the tree is as shallow as it can be made).
    const bool cheapest = !std::getenv("RT_TUNE_LEVEL_SKIPPING"); // experiment hook
    wt.nodes.reserve(bin.fnodes.size() / 2 + 16);
    Collapser c{bin, wt, {}, {}, {}, {}, {}};
    if (cheapest) {
        c.cost.assign(3 * bin.fnodes.size(), 0.0f);
        c.split.assign(4 * bin.fnodes.size(), 1);
        c.flags.assign(bin.fnodes.size(), 0);
        c.levels.assign(3 * bin.fnodes.size(), 0);
        c.height.assign(bin.fnodes.size(), 0);
        Child two[2];
        c.children_of(bin.root_ref, two);
        Child root = two[0];
        for (int k = 0; k < 3; k++) {
            root.lo[k] = std::min(two[0].lo[k], two[1].lo[k]);
            root.hi[k] = std::max(two[0].hi[k], two[1].hi[k]);
        }
        c.plan(bin.root_ref, half_area(root));
    }
    uint32_t need = 0;
    const uint32_t budget = std::max(depth_wanted, (bin.max_depth + 1) / 2);
    uint32_t root = c.emit(bin.root_ref, need, budget);
    if (need > WREF_NEED_MAX) throw std::length_error("fast tree deeper than 31 wide levels");
    wt.depth = need;
    wt.root_ref = root | (need << WREF_NEED_SHIFT);
    // Order of the records in memory. emit() leaves them depth first (a node, then the whole subtree of its first
    // child, ...); they are laid out breadth first instead: the children of a node side by side (two 64-byte
    // records share a memory line) and the levels every ray visits together. Measured against depth first: hippo
    // render +0.4 %, 11.9 M primitive scene batch +1.0 %, render +0.8 %; the first 6 or 10 levels breadth first and
    // the subtrees below depth first (RT_TUNE_WIDE_BFS=levels, 0 = all depth first) lie in between. Either way a
    // child's index is larger than its parent's.
    static const int bfs_levels = [] {
        const char *e = std::getenv("RT_TUNE_WIDE_BFS");
        return e ? std::atoi(e) : 99;
    }();
    if (bfs_levels > 0 && wt.nodes.size() > 1) {
        const uint32_t n_nodes = (uint32_t)wt.nodes.size();
        std::vector<uint32_t> order; // new index -> old index
        order.reserve(n_nodes);
        std::vector<uint32_t> frontier{root}, below;
        for (int level = 0; level < bfs_levels && !frontier.empty(); level++) {
            std::vector<uint32_t> nxt;
            for (uint32_t k : frontier) {
                order.push_back(k);
                for (int c = 0; c < 4; c++) {
                    const uint32_t r = wt.nodes[k].ref[c];
                    if (r != REF_NONE && !(r & REF_LEAF)) nxt.push_back(r & WREF_INDEX_MASK);
                }
            }
            frontier.swap(nxt);
        }
        std::vector<uint32_t> stack(frontier.rbegin(), frontier.rend());
        while (!stack.empty()) {
            const uint32_t k = stack.back();
            stack.pop_back();
            order.push_back(k);
            for (int c = 3; c >= 0; c--) {
                const uint32_t r = wt.nodes[k].ref[c];
                if (r != REF_NONE && !(r & REF_LEAF)) stack.push_back(r & WREF_INDEX_MASK);
            }
        }
        if (order.size() != n_nodes) throw std::logic_error("wide tree layout lost a node");
        std::vector<uint32_t> new_of(n_nodes);
        for (uint32_t i = 0; i < n_nodes; i++) new_of[order[i]] = i;
        std::vector<WNode> moved(n_nodes);
        for (uint32_t i = 0; i < n_nodes; i++) {
            moved[i] = wt.nodes[order[i]];
            for (int c = 0; c < 4; c++) {
                uint32_t &r = moved[i].ref[c];
                if (r != REF_NONE && !(r & REF_LEAF)) r = (r & ~WREF_INDEX_MASK) | new_of[r & WREF_INDEX_MASK];
            }
        }
        wt.nodes.swap(moved);
        wt.root_ref = new_of[root] | (need << WREF_NEED_SHIFT);
    }
    return wt;
}

void decode_child_box(const QNode &q, int k, double lo[3], double hi[3]) {
    for (int a = 0; a < 3; a++) {
        const double scale = (double)q.scale[a];
        lo[a] = (double)q.org[a] + (double)((q.qlo[a] >> (8 * k)) & 0xFFu) * scale;
        hi[a] = (double)q.org[a] + (double)((q.qhi[a] >> (8 * k)) & 0xFFu) * scale;
    }
}

std::vector<QNode> quantise_wide(const WideTree &wide) {
    std::vector<QNode> out(wide.nodes.size());
    for (size_t i = 0; i < wide.nodes.size(); i++) {
        const WNode &w = wide.nodes[i];
        QNode &q = out[i];
        std::memset(&q, 0, sizeof q);
        const float *lo[3] = {w.lox, w.loy, w.loz}, *hi[3] = {w.hix, w.hiy, w.hiz};
        for (int k = 0; k < 4; k++) q.ref[k] = w.ref[k];
        for (int a = 0; a < 3; a++) {
            double nlo = INFINITY, nhi = -INFINITY; // the node's own box along this axis
            for (int k = 0; k < 4; k++) {
                if (w.ref[k] == REF_NONE) continue;
                nlo = std::min(nlo, (double)lo[a][k]);
                nhi = std::max(nhi, (double)hi[a][k]);
            }
            if (!(nlo <= nhi)) { // no children (the unused record of a tiny scene)
                q.org[a] = 0.0f;
                q.scale[a] = 1.0f;
                continue;
            }
            q.org[a] = (float)nlo; // exact: nlo is one of the f32 inputs
            // smallest power of two with 255 steps covering the extent
            int e = -126;
            const double extent = nhi - nlo;
            if (extent > 0.0) {
                (void)std::frexp(extent / 255.0, &e); // extent/255 = m * 2^e, m in [0.5, 1)
                e = std::max(e - 1, -126);
                while (std::ldexp(255.0, e) < extent) e++;
            }
            if (e > 127) throw std::length_error("fast tree box too large to quantise");
            const double scale = std::ldexp(1.0, e);
            q.scale[a] = (float)scale; // exact: a power of two in the normal range
            for (int k = 0; k < 4; k++) {
                if (w.ref[k] == REF_NONE) continue; // stays 0 / 0; the walk looks at the reference
                double l = std::floor(((double)lo[a][k] - nlo) / scale);
                double h = std::ceil(((double)hi[a][k] - nlo) / scale);
                l = std::min(std::max(l, 0.0), 255.0);
                h = std::min(std::max(h, 0.0), 255.0);
                if (nlo + h * scale < (double)hi[a][k] || nlo + l * scale > (double)lo[a][k])
                    throw std::logic_error("quantised box does not contain its child");
                q.qlo[a] |= (uint32_t)l << (8 * k);
                q.qhi[a] |= (uint32_t)h << (8 * k);
            }
        }
    }
    return out;
}

} // namespace rtd
