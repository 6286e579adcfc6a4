// rt_device.h — device-side scene layout and kernel launchers (gfx950).
//
// Layout in HBM (all arrays 256-byte aligned by hipMalloc):
//   nodes   rt_node[n_nodes]   64 B  reference tree, right-first pre-order, f64 shells
//   qnodes  QNode[n_wide]      64 B  the fast walk's hierarchy: one record per node holding the
//                                    conservative boxes (8 bits per plane, relative to the node's
//                                    own box) and the references of up to FOUR children
//   pre     PrimPre[n_slots]   64 B  f32 plane + edge half-spaces for the cheap screen
//   pre2    PrimPre2[n_slots] 128 B  the same for slots s and s+1, interleaved field by field
//   hot     PrimHot[n_slots]   96 B  the fields every exact primitive test reads first
//   cold    PrimCold[n_slots] 128 B  polygon basis + 2-D outline, read by the few
//                                    candidates that pass the bounding-circle test
//   pts2d   double2[...]             outlines of polygons with more than 4 vertices
//   mats    rt_material[...]   64 B
// pre/hot/cold are indexed by FAST slot = position in the fast tree's leaf order, so a
// leaf's primitives are adjacent and no kind/index indirection is needed. rt_node leaves
// name runs of REFERENCE slots (position in the reference's visiting order); the two
// numberings are tied by ref_of_fast / fast_of_ref.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_amd.h"

namespace rtd {

struct alignas(32) PrimHot {
    double nx, ny, nz, radius;        // polygon: normal, bounding_radius | sphere: -, radius
    double p0x, p0y, p0z;             // polygon: points[0]
    uint32_t n_points, points_offset; // n_points == 0 marks a sphere
    double cx, cy, cz;                // polygon: bounding_center | sphere: center
    uint32_t material, prim_id;
};
static_assert(sizeof(PrimHot) == 96, "PrimHot must be 96 bytes");

struct alignas(32) PrimCold {
    double b1[3];
    double b2[3];
    double pts[8]; // up to 4 (x,y) pairs inline
    double pad[2];
};
static_assert(sizeof(PrimCold) == 128, "PrimCold must be 128 bytes");

// f32 screen record of a primitive. Polygon: its plane (normal, normal.p0) and up to three
// half-spaces A.q + C >= 0 that contain it (all three edges of a triangle; three edges of a
// larger convex polygon; "always true" otherwise). Sphere: zero normal, e[0] = (centre, radius).
struct alignas(64) PrimPre {
    float nx, ny, nz, k;
    float e[3][4];
};
static_assert(sizeof(PrimPre) == 64, "PrimPre must be 64 bytes");
// The screen records of slots s and s+1 (of s twice at the last slot) side by side: word 2*j + p holds
// field j (PrimPre order) of primitive p. A leaf visit screens both primitives with packed f32
// operations, one pair of values per instruction, and one 16-byte piece of the record comes in already
// paired the way those instructions want their operands.
struct alignas(128) PrimPre2 {
    float f[32];
};
static_assert(sizeof(PrimPre2) == 128, "PrimPre2 must be 128 bytes");

// child reference: bit 31 clear = index into fnodes; bit 31 set = leaf with
// bits 30..28 = primitive count - 1 (1..8) and bits 27..0 = first slot
constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_CNT_SHIFT = 28;
constexpr uint32_t REF_SLOT_MASK = 0x0FFFFFFFu;
constexpr uint32_t REF_NONE = 0xFFFFFFFFu; // empty scene / no child / ray finished
constexpr uint32_t FAST_LEAF_MAX = 8;
constexpr uint32_t TOP_LEVELS = 8;     // levels of the binary build tree laid out breadth first

// Binary node of the BUILDERS (host side only; the device walks QNode records).
struct alignas(64) FNode {
    float lo0[3], hi0[3]; // child 0: the reference's first-visited (right) child
    float lo1[3], hi1[3]; // child 1: the left child
    uint32_t ref0, ref1;
    uint32_t reserved[2];
};
static_assert(sizeof(FNode) == 64, "FNode must be 64 bytes");

// Wide node as the BUILDERS hold it (host side only: rt_wide_tree.hip collapses the binary tree into these, the
// checks of rt_fast_tree_check run on them, and quantise_wide turns each into the 64-byte QNode below, which is
// the only node record the device ever reads): a cut through the binary subtree below a node, i.e. up to four
// children with f32 boxes, stored plane by plane. A ray needs about half as many DEPENDENT loads as with two
// boxes per record.
//   inner child reference: bits 24..0 index into wnodes, bits 29..25 `need` = number of wide
//     nodes on the longest path from that child down (1 = all its children are leaves);
//   leaf child reference: as above (REF_LEAF | count-1 | first slot); REF_NONE = no child.
// On the walk's stack an inner reference may instead be a GROUP entry (REF_GROUP): "the children
// of node `index` named by the 4-bit mask are still to be visited"; see traverse_fast.
struct alignas(128) WNode {
    float lox[4], loy[4], loz[4];
    float hix[4], hiy[4], hiz[4];
    uint32_t ref[4];
    uint32_t reserved[4];
};
static_assert(sizeof(WNode) == 128, "WNode must be 128 bytes");
// What the device reads: the same node in 64 bytes. A lane's 16-byte load from its own record
// costs the CU's vector cache one cycle whatever the record's size (measured: 0.87 such loads per
// clock and CU at best), so a node step costs as many cycles as its record has 16-byte pieces:
// four here instead of seven. Child k's box along axis a is
//   [org[a] + qlo[a][k] * scale[a], org[a] + qhi[a][k] * scale[a]],
// rounded outward by the host, so it contains the f32 box it was made from.
struct alignas(64) QNode {
    float org[3];
    float scale[3];  // powers of two, one per axis
    uint32_t qlo[3]; // per axis: byte k = child k
    uint32_t qhi[3];
    uint32_t ref[4]; // the children come first: ref[0] and ref[1] are never REF_NONE
};
static_assert(sizeof(QNode) == 64, "QNode must be 64 bytes");
constexpr uint32_t WREF_INDEX_BITS = 25;
constexpr uint32_t WREF_INDEX_MASK = (1u << WREF_INDEX_BITS) - 1u;
constexpr uint32_t WREF_NEED_SHIFT = 25;  // 5 bits
constexpr uint32_t WREF_NEED_MAX = 31;
constexpr uint32_t REF_GROUP = 0x40000000u;
constexpr uint32_t WREF_MASK_SHIFT = 26;  // 4 bits, group entries only

struct Scene {
    const rt_node *nodes;
    const QNode *qnodes;
    const PrimPre *pre;
    const PrimPre2 *pre2;
    const PrimHot *hot;
    const PrimCold *cold;
    const double *pts2d;
    const rt_material *mats;
    const uint32_t *ref_of_fast; // fast slot -> reference slot (ties go to the smaller reference slot)
    const uint32_t *fast_of_ref; // reference slot -> fast slot (the reference-order walk's way to the records)
    uint32_t n_nodes, n_slots, n_inner, stack_depth; // stack_depth: entries per lane of the walk's LDS stack
    uint32_t has_spheres; // a sphere can make a ray's reference answer order dependent: second pass needed
    uint32_t has_gloss;            // some material has gloss > 0: shading needs the lobe sampler
    uint32_t root_ref, reserved; // fast traversal entry
    // The two-rays-per-lane walk (rt_walk2_device.hpp): the same nodes with 20-bit child references, present
    // when the scene is small enough for them (two_rays != 0: bit 0 = the batch launches use that walk, bit 1 =
    // the path tracer's launches do), and that walk's vote thresholds (lanes that have a ray of the kind among
    // their two).
    const QNode *qnodes_c;
    uint32_t root_ref_c, two_rays;
    uint32_t vote2_exact, vote2_leaf, vote2_refill, vote2_exact_paths, vote2_leaf_paths, vote2_refill_paths;
    // The parked-ray walk (rt_walk3_device.hpp): walk3 bit 0 = the batch launches use it, bit 1 = the path tracer's
    // launches do; lanes that must be able to switch to their parked ray / be without a parked ray / have nothing
    // to do and nothing parked before the wavefront runs a switch or a prep round.
    uint32_t walk3, vote3_switch, vote3_prep, vote3_stuck, vote3_switch_paths, vote3_prep_paths, vote3_stuck_paths;
    uint32_t stack_depth3; // that walk's stack entries per lane (one per wide level + 1, at most 15)
    double root_c[3], root_r;    // root shell, for moving far origins next to the scene
    float t_abs_slack;           // absolute slack added to pruning distances
    float pre_m_test, pre_m_p;   // f32 screen: margins on (o-p0).n and on n.dir
    float pre_c1, pre_c2;        // f32 screen: plane-point error = c1 + c2 / |n.dir|
    // lanes that must be waiting before a wavefront runs an exact-test round, a
    // leaf-screen round, or stores finished rays and fetches new ones
    uint32_t vote_exact, vote_leaf, vote_refill;
    // the same three thresholds for the path tracer's launches (costlier loads and stores, other ray mix)
    uint32_t vote_exact_paths, vote_leaf_paths, vote_refill_paths;
    // a node round repeats up to node_iters times while no more than node_drop/16 of its
    // lanes have left the node state
    uint32_t node_iters, node_drop;
#if defined(RT_COUNT_VISITS) || defined(RT_TIME_ROUNDS)
    // instrumented builds only: [0] rays, [1] node steps, [2] primitives screened, [3] exact tests, [4..11]
    // rounds and lanes per kind, [12..19] wave cycles and rounds per kind (RT_TIME_ROUNDS)
    unsigned long long *visits;
#endif
};

// State of the wavefront integrator. The live paths of a bounce are kept DENSE: slot t
// of the current ray buffer is the t-th surviving path, whatever its original index, so
// every kernel reads and writes consecutive addresses; shading compacts the survivors
// into the other ray buffer. Only the per-bounce history that the final fold needs is
// addressed by original path index.
struct Paths {
    double *rays;           // [2][6][capacity] ping-pong: origin xyz, direction xyz by live slot
    uint32_t *ids;          // [2][capacity] original path index of each live slot
    double *hit;            // [3][capacity] closest hit of the current bounce: position
    uint32_t *hit_mat;      // slot of the primitive that was hit, RT_MISS = no hit
    double *cosine;         // [max_depth][capacity] dot(new_dir, normal) per bounce (tracer.rs:49), by path
    uint32_t *material;     // [max_depth][capacity] by path
    uint8_t *length;        // bounces recorded (tracer.rs `i`), by path
    uint8_t *emissive;      // encountered_emissive_feature, by path
    uint32_t *counters;     // [d] = live paths entering bounce d
    uint64_t capacity;
    // A film pass does not copy its camera rays into `rays`: bounce 0 reads path t's ray straight
    // from the film's bundle (ray t % first_period, identity ids). NULL for explicit ray lists.
    const double *first_bundle;
    uint32_t first_period;
};

struct FilmView {
    const uint32_t *owned;  // global pixel index of each owned pixel
    const double *bundle;   // owned-pixel-major ray bundle: [n_owned][5][6]
    double *total[3];       // radiance sums per owned pixel
    uint32_t n_owned, width, height;
};

// ---- launchers (all asynchronous on `stream`) ------------------------------
void launch_intersect(const Scene &sc, int mode, const double *rays, uint64_t n, uint32_t *prim_id,
                      double *distance, double *pos, double *normal, unsigned long long *ticket, hipStream_t stream);
void launch_intersect_brute(const Scene &sc, const double *rays, uint64_t n, uint32_t *prim_id, double *distance,
                            hipStream_t stream);
void launch_intersect_prims(const Scene &sc, const double *rays, const uint32_t *slots, uint64_t n,
                            uint8_t *hit, double *pos, double *normal, hipStream_t stream);
void launch_raygen(const rt_camera &cam, uint32_t width, uint32_t height, uint64_t seed,
                   const uint32_t *owned /* NULL = all pixels in order */, uint32_t n_pixels, double *bundle,
                   hipStream_t stream);
void launch_tone_map(const double *radiance, uint64_t n, uint8_t *rgba, hipStream_t stream);

// paths from explicit rays + keys (rt_trace_paths) or from a film pass
void launch_paths_from_rays(const Paths &p, const double *rays, uint64_t n, hipStream_t stream);
// p.first_bundle / first_period must name f's bundle; only clears the per-path flags
void launch_paths_from_film(const Paths &p, const FilmView &f, uint32_t n_samples, hipStream_t stream);
// One bounce: closest hits of the live paths of `depth` (their exact number is on the
// device; `expected` only sizes the grid, any number is handled), then shading, which
// compacts the survivors into bounce depth+1.
// walk_blocks_per_cu (fast mode): how many of the blocks a CU can hold the persistent grid may occupy (0 = all of
// them); a caller that keeps two launches in flight on two streams gives each its share.
void launch_traverse_paths(const Scene &sc, int mode, const Paths &p, uint32_t depth, uint64_t expected,
                           unsigned long long *ticket, hipStream_t stream, uint32_t walk_blocks_per_cu = 0,
                           bool callers_rays = false);
// the fast walk of launch_traverse_paths (rt_path_walk.hip)
void launch_path_walk_fast(const Scene &sc, const Paths &p, uint32_t depth, uint64_t expected, unsigned long long *ticket,
                           hipStream_t stream, uint32_t walk_blocks_per_cu = 0);
void launch_shade(const Scene &sc, const Paths &p, uint32_t depth, uint64_t expected, uint64_t seed,
                  const uint32_t *keys /* n x 3, or NULL with film */, const FilmView *film,
                  uint32_t first_sample, hipStream_t stream);
// tracer.rs:63-78 fold per path -> radiance[n][3]
void launch_fold_paths(const Scene &sc, const Paths &p, uint64_t n, double *radiance, hipStream_t stream);
// main.rs:58-67: fold + add in (sample, sub-ray) order per owned pixel
void launch_accumulate_film(const Scene &sc, const Paths &p, const FilmView &f, uint32_t n_samples,
                            hipStream_t stream);
// combine of a multi-device render: acc = (first ? 0 : acc) + w * src over n doubles; RGBA8 frames with
// disjoint non-zero pixels summed byte-wise (n_pixels x 4 bytes)
void launch_frame_axpy(double *acc, const double *src, double w, uint64_t n, bool first, hipStream_t stream);
void launch_bytes_add(uint8_t *acc, const uint8_t *src, uint64_t n_pixels, bool first, hipStream_t stream);
// main.rs:68-74 into full-frame buffers (zeros outside the shard)
void launch_resolve(const FilmView &f, uint32_t spp, double *d_radiance, uint8_t *d_rgba, hipStream_t stream);

} // namespace rtd
