// rt_fast_tree.hpp — host-side construction of the tree RT_TRAVERSE_FAST walks.
//
// The fast walk only has to return the same closest hit as the reference walk
// (bvh.rs:97-149), so it is free to use its own hierarchy over the same primitives.
// The builders produce a binary tree in one format (64-byte FNode records holding both
// children's conservative f32 boxes, leaf references to runs of consecutive "fast
// slots", and the permutation between fast slots and reference slots), which
// collapse_to_wide then turns into the 4-wide records the device walks:
//
//   build_fast_tree_sah        binned surface-area-heuristic binary tree (default)
//   build_fast_tree_reference  the reference's own topology (bvh.rs:21-95) with boxes
//                              in place of bounding spheres; kept for A/B measurements
#pragma once

#include <cstdint>
#include <vector>

#include "rt_device.h"

namespace rtd {

struct BoxD {
    double lo[3], hi[3];
};

struct FastTree {
    std::vector<FNode> fnodes;          // never empty (a dummy record when the scene has < 2 leaves)
    std::vector<uint32_t> ref_of_fast;  // fast slot -> reference slot
    uint32_t root_ref = REF_NONE;
    uint32_t max_depth = 1;             // most inner nodes on any root-to-leaf path
    uint32_t top_nodes = 0;             // fnodes[0 .. top_nodes) are the top levels in breadth-first order
};

// Most inner nodes on a root-to-leaf path the SAH builder may produce (it aims for 30 where the primitive
// count allows: 15 wide levels). The walk keeps at least one 6-byte stack entry per wide level in LDS. Near
// the limit the builder falls back to median splits.
constexpr uint32_t FAST_TREE_MAX_DEPTH = 35;

// slot_box[s]: f64 bounds of the primitive in reference slot s. pad: absolute amount every
// box is grown by before rounding outward to f32. top_levels: levels laid out breadth first.
FastTree build_fast_tree_sah(const std::vector<BoxD> &slot_box, double pad, uint32_t top_levels, unsigned threads);
FastTree build_fast_tree_reference(const rt_scene_desc &d, const std::vector<BoxD> &slot_box, double pad);
// Linear BVH built on `device` (rt_lbvh.hip): Morton sort + Karras radix tree + bottom-up fit. Keys are
// shortened until the tree is at most max_depth_wanted deep; if even 6 bits per axis are too deep the
// last attempt is returned and the caller checks max_depth against what the walk's stack can hold.
// The hierarchy the device walks: 128-byte nodes of up to four children (rt_wide_tree.hip).
struct WideTree {
    std::vector<WNode> nodes;   // never empty (one unused record for a scene of fewer than two leaves)
    uint32_t root_ref = REF_NONE; // inner reference with its `need` bits, or a leaf reference, or REF_NONE
    uint32_t depth = 0;         // wide nodes on the longest root-to-leaf path
};
// Collapses every other level of a binary build tree (any of the three builders). Throws
// std::length_error for more than 2^25 wide nodes or more than 31 wide levels.
// depth_wanted: the wide depth the caller can take at no cost; a deeper result of the cheapest collapse is
// replaced by the level-skipping one (at most ceil(binary depth / 2) levels) when that is shallower.
WideTree collapse_to_wide(const FastTree &binary, uint32_t depth_wanted);
// The 64-byte device records of the same nodes: child boxes quantised to 8 bits per plane relative
// to the node's own box, rounded outward.
std::vector<QNode> quantise_wide(const WideTree &wide);
// The box of child k of a quantised node as the device decodes it (lo[3], hi[3]).
void decode_child_box(const QNode &q, int k, double lo[3], double hi[3]);

FastTree build_fast_tree_lbvh(const std::vector<BoxD> &slot_box, double pad, int device, uint32_t max_depth_wanted);

} // namespace rtd
