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

} // namespace rtd
