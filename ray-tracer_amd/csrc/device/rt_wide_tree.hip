// rt_wide_tree.hip — host code only: collapses a binary build tree into the 4-wide hierarchy
// RT_TRAVERSE_FAST walks.
//
// Every wide node takes the place of a binary node and its two children: its children are the
// binary node's grandchildren (a binary child that is a leaf stays a leaf child, so a wide node has
// two to four children). Every root-to-leaf path therefore loses every other level, which bounds
// the wide depth by ceil(binary depth / 2) whatever the shape of the tree: the walk sizes its
// per-lane stack from that bound.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

#include "rt_fast_tree.hpp"

namespace rtd {
namespace {

struct Child {
    uint32_t ref; // binary reference: inner index or leaf reference
    float lo[3], hi[3];
};

struct Collapser {
    const FastTree &bin;
    WideTree &out;

    void children_of(uint32_t binary_index, Child c[2]) const {
        const FNode &f = bin.fnodes[binary_index];
        c[0].ref = f.ref0;
        c[1].ref = f.ref1;
        for (int k = 0; k < 3; k++) {
            c[0].lo[k] = f.lo0[k]; c[0].hi[k] = f.hi0[k];
            c[1].lo[k] = f.lo1[k]; c[1].hi[k] = f.hi1[k];
        }
    }

    // Emits the wide node replacing binary node `binary_index`; returns its index, and through
    // `need` the number of wide nodes on the longest path from it down (itself included).
    uint32_t emit(uint32_t binary_index, uint32_t &need) {
        const uint32_t idx = (uint32_t)out.nodes.size();
        if (idx > WREF_INDEX_MASK) throw std::length_error("fast tree has more than 2^25 wide nodes");
        out.nodes.emplace_back();
        Child two[2], kids[4];
        uint32_t n = 0;
        children_of(binary_index, two);
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
        uint32_t refs[4] = {REF_NONE, REF_NONE, REF_NONE, REF_NONE};
        uint32_t deepest = 0;
        for (uint32_t k = 0; k < n; k++) {
            if (kids[k].ref & REF_LEAF) {
                refs[k] = kids[k].ref;
            } else {
                uint32_t child_need = 0;
                uint32_t child = emit(kids[k].ref, child_need);
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

WideTree collapse_to_wide(const FastTree &bin) {
    WideTree wt;
    wt.root_ref = bin.root_ref;
    if (bin.root_ref == REF_NONE || (bin.root_ref & REF_LEAF)) { // empty scene or a single leaf
        wt.nodes.emplace_back();
        std::memset(&wt.nodes[0], 0, sizeof(WNode));
        for (int k = 0; k < 4; k++) wt.nodes[0].ref[k] = REF_NONE;
        wt.depth = 0;
        return wt;
    }
    wt.nodes.reserve(bin.fnodes.size() / 2 + 16);
    Collapser c{bin, wt};
    uint32_t need = 0;
    uint32_t root = c.emit(bin.root_ref, need);
    if (need > WREF_NEED_MAX) throw std::length_error("fast tree deeper than 31 wide levels");
    wt.depth = need;
    wt.root_ref = root | (need << WREF_NEED_SHIFT);
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
