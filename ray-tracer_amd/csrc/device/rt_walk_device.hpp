// rt_walk_device.hpp — device code shared by the translation units that hold kernels: the reference-order walk,
// the fast walk (traverse_fast) with its ray sources and sinks, sampling and tone-curve routines, and the host
// helpers that size the persistent grids. Everything sits in an anonymous namespace: each translation unit
// gets its own copy and emits only what its kernels use.
#pragma once
//
// Compiled with -ffp-contract=off: every f64 expression that decides a hit is
// evaluated in the reference's order without fused multiply-adds, so hit
// positions and distances are the same IEEE doubles the reference computes
// (f32 culling code asks for its fused operations explicitly with fmaf).
// Nothing here is a dense contraction, so no MFMA is used; the work is
// f32/f64 vector arithmetic over cache-resident 32..128-byte records, bound by
// lane divergence, vector-memory issue and latency rather than by HBM.
#include <algorithm>
#include <cstdlib>

#include "rt_device.h"

namespace rtd {
namespace {

constexpr double EPSILON = 0.00000001; // geometry.rs:3
constexpr int BLOCK = 256;
// threads per block of the fast-walk kernels
#if defined(RT_TBLOCK)
constexpr int TBLOCK = RT_TBLOCK; // experiment
#else
constexpr int TBLOCK = 256;
#endif
// Wavefronts per SIMD the fast-walk kernels are compiled for (register budget 512 / waves). The walk
// spends more than half of a wavefront's life waiting for dependent loads, so a fifth resident
// wavefront is worth more than the registers it costs (measured: 5 waves with 16 spilled registers
// +5 % on the batch, +2.5 % on the render; 6 waves with 48 spilled registers -25 %).
#if !defined(RT_MIN_WAVES)
#define RT_MIN_WAVES 5
#endif
#define RT_FAST_BOUNDS __launch_bounds__(TBLOCK, RT_MIN_WAVES)
constexpr double PI = 3.14159265358979323846; // std::f64::consts::PI

struct V3 {
    double x, y, z;
};

// ---- geometry.rs:6-46 in the reference's evaluation order ------------------
__device__ __forceinline__ double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(const V3 &a, const V3 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 add(const V3 &a, const V3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 sub(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 scale(double c, const V3 &a) { return {c * a.x, c * a.y, c * a.z}; }
__device__ __forceinline__ double sq(const V3 &a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ double mag(const V3 &a) { return sqrt(sq(a)); }
__device__ __forceinline__ V3 normalize(const V3 &a) { return scale(1.0 / mag(a), a); }

// Marks the operand of a rarely needed square root or division: the empty statement cannot be
// executed speculatively, so the compiler leaves the expensive operation inside its branch instead
// of computing it for every lane ahead of the test that guards it.
__device__ __forceinline__ double rare(double v) {
    asm volatile("" : "+v"(v));
    return v;
}

// sqrt(s) > r and sqrt(s) <= r decided exactly as the correctly rounded sqrt
// would, taking the square root only inside a 1e-12 relative band around r*r.
__device__ __forceinline__ bool sqrt_gt(double s, double r) {
    double r2 = r * r;
    if (s > r2 * (1.0 + 1e-12)) return true;
    if (s < r2 * (1.0 - 1e-12)) return false;
    return sqrt(rare(s)) > r;
}
__device__ __forceinline__ bool sqrt_le(double s, double r) {
    double r2 = r * r;
    if (s < r2 * (1.0 - 1e-12)) return true;
    if (s > r2 * (1.0 + 1e-12)) return false;
    return sqrt(rare(s)) <= r;
}

// ---- geometry.rs:76-106 ------------------------------------------------------
// One edge of the crossing count (geometry.rs:86-103). The reference has one case for an edge that
// goes down across the line y = ty (p1 above, p2 below) and one for an edge that goes up. The second
// case is the first with every intermediate negated: d' = p2y - p1y = -d, w1' = d' - (ty - p1y) = -w1,
// w2' = -w2, so x' = (w1'*p1x + w2'*p2x) / d' = (-(w1*p1x + w2*p2x)) / (-d), and IEEE rounding is
// symmetric in the sign at every step: x' and x are the same double. One body therefore serves both
// cases, which halves the work of a wavefront whose lanes meet edges of both kinds.
__device__ __forceinline__ int edge_crossing(double p1x, double p1y, double p2x, double p2y, double tx, double ty) {
    const bool down = p1y > ty && p2y < ty, up = p1y < ty && p2y > ty;
    if (!(down || up)) return 0;
    const double d = p1y - p2y;
    const double w1 = d - (p1y - ty);
    const double w2 = d - (ty - p2y);
    const double x = (w1 * p1x + w2 * p2x) / d;
    return x >= tx ? 1 : 0;
}

__device__ __forceinline__ bool poly_inclusion_2d(const double *pts, uint32_t n, double tx, double ty) {
    int crossings = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t k = (i + 1 < n) ? i + 1 : 0;
        crossings += edge_crossing(pts[2 * i], pts[2 * i + 1], pts[2 * k], pts[2 * k + 1], tx, ty);
    }
    return (crossings % 2) == 1;
}

// ---- primitives.rs:206-248 ---------------------------------------------------
// Both intersection routines return the hit as the step s along the ray; the hit position is
// o + s*dir evaluated as add(o, scale(s, dir)), exactly the expression of primitives.rs:225-228 and :76-79,
// so a caller that evaluates it again gets the same doubles (the fast walk does: one double instead of
// three stays alive between the test and the comparison with the closest hit so far).
__device__ __forceinline__ V3 hit_position(const V3 &o, const V3 &dir, double s) { return add(o, scale(s, dir)); }

__device__ __forceinline__ bool polygon_intersect(const Scene &sc, uint32_t slot, const PrimHot &h, const V3 &o,
                                                  const V3 &dir, double &step) {
    V3 n = {h.nx, h.ny, h.nz};
    V3 p0 = {h.p0x, h.p0y, h.p0z};
    double p = dot(n, dir);
    double test = dot(sub(o, p0), n);
    if (test <= EPSILON || p >= -EPSILON) return false;
    V3 c = {h.cx, h.cy, h.cz};
    V3 a = sub(c, o);
    double d = fabs(dot(a, n));
    const double s = d / fabs(p);
    V3 pi = hit_position(o, dir, s);
    if (sqrt_gt(sq(sub(pi, c)), h.radius)) return false;

    const PrimCold &k = sc.cold[slot];
    const V3 b1 = {k.b1[0], k.b1[1], k.b1[2]};
    const V3 b2 = {k.b2[0], k.b2[1], k.b2[2]};
    const double q0x = k.pts[0], q0y = k.pts[1], q1x = k.pts[2], q1y = k.pts[3], q2x = k.pts[4], q2y = k.pts[5];
    V3 off = sub(pi, p0);
    double tx = dot(b1, off), ty = dot(b2, off);
    bool inside;
    if (h.n_points <= 4) {
        // triangles and quadrilaterals share one instruction stream: edges 0-1, 1-2, then 2-0 or 2-3,
        // and 3-0 for a quadrilateral only
        const bool tri = h.n_points == 3;
        const double q3x = k.pts[6], q3y = k.pts[7];
        int cr = edge_crossing(q0x, q0y, q1x, q1y, tx, ty) + edge_crossing(q1x, q1y, q2x, q2y, tx, ty) +
                 edge_crossing(q2x, q2y, tri ? q0x : q3x, tri ? q0y : q3y, tx, ty);
        if (!tri) cr += edge_crossing(q3x, q3y, q0x, q0y, tx, ty);
        inside = (cr % 2) == 1;
    } else {
        inside = poly_inclusion_2d(sc.pts2d + 2 * (size_t)h.points_offset, h.n_points, tx, ty);
    }
    if (!inside) return false;
    step = s;
    return true;
}

// ---- primitives.rs:45-81 -----------------------------------------------------
// `behind` is set when the hit lies behind the origin or its position is not a number (origin inside
// or on the sphere): the reference then returns whatever its visiting order makes of it (its prune
// ball around a hit behind the origin can hide closer hits, bvh.rs:106-109, and a NaN distance
// sticks only if it arrives first, bvh.rs:119-124).
__device__ __forceinline__ bool sphere_intersect(const PrimHot &h, const V3 &o, const V3 &dir, double &step, bool &behind) {
    V3 c = {h.cx, h.cy, h.cz};
    V3 leg1 = sub(c, o);
    if (dot(leg1, dir) < 0.0) return false;
    double s = sq(cross(dir, leg1));
    if (sqrt_gt(s, h.radius)) return false;
    double r = sqrt(s);
    double a = mag(leg1);
    double d = sqrt(a * a - r * r) - sqrt(h.radius * h.radius - r * r);
    step = d;
    if (!(d >= 0.0)) behind = true; // sticky: the caller's flag collects every sphere of the ray
    return true;
}

__device__ __forceinline__ bool prim_intersect_step(const Scene &sc, uint32_t slot, const PrimHot &h, const V3 &o,
                                                    const V3 &dir, double &step, bool &behind) {
    if (h.n_points == 0) return sphere_intersect(h, o, dir, step, behind);
    return polygon_intersect(sc, slot, h, o, dir, step);
}
__device__ __forceinline__ bool prim_intersect(const Scene &sc, uint32_t slot, const PrimHot &h, const V3 &o,
                                               const V3 &dir, V3 &pos, bool &behind) {
    double step;
    if (!prim_intersect_step(sc, slot, h, o, dir, step, behind)) return false;
    pos = hit_position(o, dir, step);
    return true;
}
__device__ __forceinline__ bool prim_intersect(const Scene &sc, uint32_t slot, const PrimHot &h, const V3 &o,
                                               const V3 &dir, V3 &pos) {
    bool behind = false;
    return prim_intersect(sc, slot, h, o, dir, pos, behind);
}

// primitives.rs:82-84, 250-252
__device__ __forceinline__ V3 normal_at(const PrimHot &h, const V3 &p) {
    if (h.n_points == 0) return scale(1.0 / h.radius, sub(p, V3{h.cx, h.cy, h.cz}));
    return {h.nx, h.ny, h.nz};
}

// The running closest hit (bvh.rs:100-101). Only the slot is kept while walking; the
// normal (bvh.rs:117), material and primitive id are read once, when the ray is stored.
struct Closest {
    bool have;
    uint32_t slot;
    double dist;
    V3 pos;
};
__device__ __forceinline__ Closest no_hit() { return {false, RT_MISS, 0.0, {0.0, 0.0, 0.0}}; }
__device__ __forceinline__ void take_hit(Closest &best, uint32_t slot, const PrimHot &, const V3 &pos, double d) {
    best.have = true;
    best.slot = slot;
    best.dist = d;
    best.pos = pos;
}
// what the outputs need from the winning primitive: 24 + 8 + 8 bytes of its hot record
struct HitInfo {
    uint32_t prim_id, material;
    V3 nrm;
};
__device__ __forceinline__ HitInfo hit_info(const Scene &sc, const Closest &best) {
    HitInfo r = {RT_MISS, RT_MISS, {0.0, 0.0, 0.0}};
    if (best.have) {
        const PrimHot &h = sc.hot[best.slot];
        r.prim_id = h.prim_id;
        r.material = h.material;
        if (h.n_points == 0) r.nrm = scale(1.0 / h.radius, sub(best.pos, V3{h.cx, h.cy, h.cz}));
        else r.nrm = V3{h.nx, h.ny, h.nz};
    }
    return r;
}

// bvh.rs:152-164
__device__ __forceinline__ bool shell_intersect(const double *shell, const V3 &o, const V3 &dir) {
    V3 leg1 = {shell[0] - o.x, shell[1] - o.y, shell[2] - o.z};
    if (dot(leg1, dir) < 0.0 && sqrt_gt(sq(leg1), shell[3])) return false;
    return sqrt_le(sq(cross(dir, leg1)), shell[3]);
}

// d(centre, closest.pos) - r > closest_distance (bvh.rs:106-109); the square root is only taken
// inside a 1e-9 band around the threshold. A NaN distance prunes nothing, as in the reference.
__device__ __forceinline__ bool shell_pruned(const double *shell, const V3 &best_pos, double best_dist) {
    V3 t = {shell[0] - best_pos.x, shell[1] - best_pos.y, shell[2] - best_pos.z};
    double s = sq(t);
    double lim = best_dist + shell[3];
    if (s > lim * lim * (1.0 + 1e-9)) return true;
    if (s < lim * lim * (1.0 - 1e-9)) return false;
    return sqrt(rare(s)) - shell[3] > best_dist;
}

// bvh.rs:97-149 in the reference's own visiting order. The tree is stored in
// that order, so "push left, push right, pop" becomes: on a shell hit go to the
// next record, otherwise jump over the subtree. No stack is needed.
__device__ __forceinline__ Closest traverse_reference_order(const Scene &sc, const V3 &o, const V3 &dir) {
    Closest best = no_hit();
    uint32_t i = 0;
    const uint32_t end = sc.n_nodes;
    while (i < end) {
        const rt_node &nd = sc.nodes[i];
        double shell[4] = {nd.shell[0], nd.shell[1], nd.shell[2], nd.shell[3]};
        uint32_t skip = nd.skip, sb = nd.slot_begin, cnt = nd.slot_count;
        if (best.have && shell_pruned(shell, best.pos, best.dist)) {
            i = skip;
            continue;
        }
        if (!shell_intersect(shell, o, dir)) {
            i = skip;
            continue;
        }
        for (uint32_t k = 0; k < cnt; k++) {
            uint32_t slot = sc.fast_of_ref[sb + k]; // the records are stored in the fast walk's order
            PrimHot h = sc.hot[slot];
            V3 pos;
            if (!prim_intersect(sc, slot, h, o, dir, pos)) continue;
            double d = mag(sub(pos, o)); // bvh.rs:116
            if (!best.have || d < best.dist) take_hit(best, slot, h, pos, d);
        }
        i = i + 1; // right child first; a leaf's next record is its skip target
    }
    return best;
}

// ---- fast traversal ----------------------------------------------------------
// Conservative f32 slab test of one child box in fused form: t = plane*inv - o*inv.
// The rounding of o*inv and of the fma is worth at most 2^-23*|o| of slab position,
// and the f32 ray itself deviates from the f64 ray by at most 2^-23*(|o| + t); the
// host grows every box by 1.4 times the sum of all such terms (box_pad in rt_api.hip
// lists them), so a box whose contents the f64 ray touches is never missed.
// The boxes are quantised (QNode). A plane byte q sits at slab position q*a + b with a = scale*inv
// and b = org*inv - o*inv per axis. The sign of inv tells which of a box's two planes the ray meets
// first, so the caller hands over the near and the far plane byte of each axis (`near`/`far` words,
// byte k = child k) and no min/max per axis is needed. The host rounds the bytes outward and pads
// the f32 boxes they come from by far more than the rounding of these fused operations.
typedef float f2 __attribute__((ext_vector_type(2)));
struct SlabPair {
    f2 tn, tf;
};
// children (k, k+1) at once with packed f32 operations: two children per instruction
__device__ __forceinline__ SlabPair qslab2(f2 nx, f2 ny, f2 nz, f2 fx, f2 fy, f2 fz, float ax, float ay, float az, float bx,
                                           float by, float bz) {
    const f2 AX = {ax, ax}, AY = {ay, ay}, AZ = {az, az}, BX = {bx, bx}, BY = {by, by}, BZ = {bz, bz};
    f2 tnx = __builtin_elementwise_fma(nx, AX, BX), tfx = __builtin_elementwise_fma(fx, AX, BX);
    f2 tny = __builtin_elementwise_fma(ny, AY, BY), tfy = __builtin_elementwise_fma(fy, AY, BY);
    f2 tnz = __builtin_elementwise_fma(nz, AZ, BZ), tfz = __builtin_elementwise_fma(fz, AZ, BZ);
    SlabPair r;
    r.tn.x = fmaxf(fmaxf(tnx.x, tny.x), fmaxf(tnz.x, 0.0f));
    r.tn.y = fmaxf(fmaxf(tnx.y, tny.y), fmaxf(tnz.y, 0.0f));
    r.tf.x = fminf(fminf(tfx.x, tfy.x), tfz.x) * 1.000002f;
    r.tf.y = fminf(fminf(tfx.y, tfy.y), tfz.y) * 1.000002f;
    return r;
}
// byte k of a word as a float (v_cvt_f32_ubyte0..3)
__device__ __forceinline__ float byte0(uint32_t w) { return (float)(w & 0xFFu); }
__device__ __forceinline__ float byte1(uint32_t w) { return (float)((w >> 8) & 0xFFu); }
__device__ __forceinline__ float byte2(uint32_t w) { return (float)((w >> 16) & 0xFFu); }
__device__ __forceinline__ float byte3(uint32_t w) { return (float)(w >> 24); }

__device__ __forceinline__ float prune_limit(const Scene &sc, double best_dist, double shift) {
    // largest f32 entry distance (measured from the shifted origin) that may still
    // hide a hit closer than best_dist
    float t = __double2float_ru(best_dist - shift);
    return t * 1.000002f + sc.t_abs_slack;
}

// compare-exchange of two packed sort keys, each with the child reference it belongs to
__device__ __forceinline__ void cas(uint32_t &a, uint32_t &b, uint32_t &ra, uint32_t &rb) {
    const bool swap = b < a;
    const uint32_t lo = swap ? b : a, hi = swap ? a : b, rlo = swap ? rb : ra, rhi = swap ? ra : rb;
    a = lo; b = hi; ra = rlo; rb = rhi;
}

// The children of a wide node a ray enters before `reach`, nearest first.
// k[i]: sort key of the i-th nearest such child = the bits of its entry distance (>= 0, so they order
// like the value) with the child's number in the two lowest mantissa bits; 0xFFFFFFFF from the first
// place on that has no child. r[i]: that child's reference. `allow`: bit c clear = leave child c out.
struct NodeHits {
    uint32_t k0, k1, k2, k3, r0, r1, r2, r3;
};
constexpr uint32_t NO_CHILD = 0xFFFFFFFFu;
// `none`: the reference that marks an empty place (REF_NONE, or the compact form's C_NONE).
__device__ __forceinline__ NodeHits node_hits(const QNode *node, float ix, float iy, float iz, float oix, float oiy,
                                              float oiz, float reach, bool masked, uint32_t allow,
                                              const uint32_t none = REF_NONE) {
    const uint4 *q = reinterpret_cast<const uint4 *>(node);
    const uint4 qa = q[0], qb = q[1], qc = q[2], rf = q[3];
    // qa = org x, org y, org z, scale x; qb = scale y, scale z, lo x, lo y; qc = lo z, hi x, hi y, hi z;
    // byte k of a plane word = child k.
    // slab position of plane byte b along an axis: ((org + b*scale) - o) * inv = b * (scale*inv) + (org*inv - o*inv)
    const float ax = __uint_as_float(qa.w) * ix, ay = __uint_as_float(qb.x) * iy, az = __uint_as_float(qb.y) * iz;
    const float bx = fmaf(__uint_as_float(qa.x), ix, -oix);
    const float by = fmaf(__uint_as_float(qa.y), iy, -oiy);
    const float bz = fmaf(__uint_as_float(qa.z), iz, -oiz);
    // the plane a ray meets first along an axis is the low one when it travels upward
    const uint32_t nxw = ix < 0.0f ? qc.y : qb.z, fxw = ix < 0.0f ? qb.z : qc.y;
    const uint32_t nyw = iy < 0.0f ? qc.z : qb.w, fyw = iy < 0.0f ? qb.w : qc.z;
    const uint32_t nzw = iz < 0.0f ? qc.w : qc.x, fzw = iz < 0.0f ? qc.x : qc.w;
    const SlabPair s01 = qslab2(f2{byte0(nxw), byte1(nxw)}, f2{byte0(nyw), byte1(nyw)}, f2{byte0(nzw), byte1(nzw)},
                                f2{byte0(fxw), byte1(fxw)}, f2{byte0(fyw), byte1(fyw)}, f2{byte0(fzw), byte1(fzw)}, ax, ay,
                                az, bx, by, bz);
    const SlabPair s23 = qslab2(f2{byte2(nxw), byte3(nxw)}, f2{byte2(nyw), byte3(nyw)}, f2{byte2(nzw), byte3(nzw)},
                                f2{byte2(fxw), byte3(fxw)}, f2{byte2(fyw), byte3(fyw)}, f2{byte2(fzw), byte3(fzw)}, ax, ay,
                                az, bx, by, bz);
    const float t0 = s01.tn.x, t1 = s01.tn.y, t2 = s23.tn.x, t3 = s23.tn.y;
    // a node has at least two children and they come first: only the last two places can be empty
    const bool h0 = t0 <= fminf(s01.tf.x, reach);
    const bool h1 = t1 <= fminf(s01.tf.y, reach);
    const bool h2 = t2 <= fminf(s23.tf.x, reach) && rf.z != none;
    const bool h3 = t3 <= fminf(s23.tf.y, reach) && rf.w != none;
    NodeHits n;
    n.k0 = h0 ? ((__float_as_uint(t0) & 0x7FFFFFFCu) | 0u) : NO_CHILD;
    n.k1 = h1 ? ((__float_as_uint(t1) & 0x7FFFFFFCu) | 1u) : NO_CHILD;
    n.k2 = h2 ? ((__float_as_uint(t2) & 0x7FFFFFFCu) | 2u) : NO_CHILD;
    n.k3 = h3 ? ((__float_as_uint(t3) & 0x7FFFFFFCu) | 3u) : NO_CHILD;
    if (masked) { // a group entry from the stack: only the children still to visit
        if (!(allow & 1u)) n.k0 = NO_CHILD;
        if (!(allow & 2u)) n.k1 = NO_CHILD;
        if (!(allow & 4u)) n.k2 = NO_CHILD;
        if (!(allow & 8u)) n.k3 = NO_CHILD;
    }
    n.r0 = rf.x; n.r1 = rf.y; n.r2 = rf.z; n.r3 = rf.w;
    cas(n.k0, n.k1, n.r0, n.r1); cas(n.k2, n.k3, n.r2, n.r3); cas(n.k0, n.k2, n.r0, n.r2);
    cas(n.k1, n.k3, n.r1, n.r3); cas(n.k1, n.k2, n.r1, n.r2);
    return n;
}

// Cheap f32 screen in front of the f64 primitive test. It answers "certainly no hit"
// (false) or "maybe" (true); every maybe is settled by the exact f64 routine, so the
// screen can only save work, never change a result. r0 = (normal, normal.p0); e0..e2 =
// half-spaces A.q + C >= 0 containing the polygon; a sphere has a zero normal and
// e0 = (centre, radius).
__device__ __forceinline__ bool prefilter_pass(const Scene &sc, const float4 &rec0, const float4 &e0, const float4 &e1,
                                               const float4 &e2, float ox, float oy, float oz, float dx, float dy,
                                               float dz) {
    if (rec0.x == 0.0f && rec0.y == 0.0f && rec0.z == 0.0f) {
        // primitives.rs:56,65: no hit if the centre lies behind the origin or the ray passes
        // farther from the centre than the radius; the cross-product form keeps the f32
        // error proportional to the distance, not to its square
        float lx = e0.x - ox, ly = e0.y - oy, lz = e0.z - oz;
        float along = fmaf(lx, dx, fmaf(ly, dy, lz * dz));
        float cx = dy * lz - dz * ly, cy = dz * lx - dx * lz, cz = dx * ly - dy * lx;
        float dist2 = fmaf(cx, cx, fmaf(cy, cy, cz * cz));
        float lim = e0.w + sc.pre_c1;
        bool outside = along < -sc.pre_m_test || dist2 > lim * lim; // NaN -> maybe
        return !outside;
    }
    float pf = fmaf(rec0.x, dx, fmaf(rec0.y, dy, rec0.z * dz));          // normal . dir
    float tf = fmaf(rec0.x, ox, fmaf(rec0.y, oy, rec0.z * oz)) - rec0.w; // (o - p0) . normal
    // primitives.rs:217 rejects test <= 1e-8 or p >= -1e-8; stay clear of both by the f32 error
    if (tf < -sc.pre_m_test || pf > sc.pre_m_p) return false;
    float ap = fabsf(pf);
    float inv = __builtin_amdgcn_rcpf(ap);
    float t = fabsf(tf) * inv;
    float qx = fmaf(t, dx, ox), qy = fmaf(t, dy, oy), qz = fmaf(t, dz, oz); // where the ray meets the plane
    // the point is off by at most c1 + c2/|n.dir|; outside only if clearly beyond one edge
    // (primitives.rs:231-247 reject such points in f64 as well: they are outside the outline)
    float m = -2.0f * fmaf(sc.pre_c2, inv, sc.pre_c1);
    float s0 = fmaf(e0.x, qx, fmaf(e0.y, qy, fmaf(e0.z, qz, e0.w)));
    float s1 = fmaf(e1.x, qx, fmaf(e1.y, qy, fmaf(e1.z, qz, e1.w)));
    float s2 = fmaf(e2.x, qx, fmaf(e2.y, qy, fmaf(e2.z, qz, e2.w)));
    bool outside = (s0 < m || s1 < m || s2 < m) && ap > 1e-5f; // NaN -> maybe
    return !outside;
}

// The same screen for two primitives at once (a PrimPre2 record): bit p of the result = primitive p
// may be hit. Every step is the scalar routine's, on pairs of values.
__device__ __forceinline__ uint32_t prefilter_pass2(const Scene &sc, const float4 *rec, float ox, float oy, float oz, float dx,
                                                    float dy, float dz) {
    const float4 v0 = rec[0], v1 = rec[1], v2 = rec[2], v3 = rec[3], v4 = rec[4], v5 = rec[5], v6 = rec[6], v7 = rec[7];
    const f2 nx = {v0.x, v0.y}, ny = {v0.z, v0.w}, nz = {v1.x, v1.y}, nk = {v1.z, v1.w};
    const f2 DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz}, OX = {ox, ox}, OY = {oy, oy}, OZ = {oz, oz};
    const f2 pf = __builtin_elementwise_fma(nx, DX, __builtin_elementwise_fma(ny, DY, nz * DZ));
    const f2 tf = __builtin_elementwise_fma(nx, OX, __builtin_elementwise_fma(ny, OY, nz * OZ)) - nk;
    const float ap0 = fabsf(pf.x), ap1 = fabsf(pf.y);
    const float inv0 = __builtin_amdgcn_rcpf(ap0), inv1 = __builtin_amdgcn_rcpf(ap1);
    const f2 inv = {inv0, inv1};
    const f2 t = {fabsf(tf.x) * inv0, fabsf(tf.y) * inv1};
    const f2 qx = __builtin_elementwise_fma(t, DX, OX), qy = __builtin_elementwise_fma(t, DY, OY),
             qz = __builtin_elementwise_fma(t, DZ, OZ);
    const f2 C1 = {sc.pre_c1, sc.pre_c1}, C2 = {sc.pre_c2, sc.pre_c2}, M2 = {-2.0f, -2.0f};
    const f2 m = M2 * __builtin_elementwise_fma(C2, inv, C1);
    const f2 e0x = {v2.x, v2.y}, e0y = {v2.z, v2.w}, e0z = {v3.x, v3.y}, e0w = {v3.z, v3.w};
    const f2 e1x = {v4.x, v4.y}, e1y = {v4.z, v4.w}, e1z = {v5.x, v5.y}, e1w = {v5.z, v5.w};
    const f2 e2x = {v6.x, v6.y}, e2y = {v6.z, v6.w}, e2z = {v7.x, v7.y}, e2w = {v7.z, v7.w};
    const f2 s0 = __builtin_elementwise_fma(e0x, qx, __builtin_elementwise_fma(e0y, qy, __builtin_elementwise_fma(e0z, qz, e0w)));
    const f2 s1 = __builtin_elementwise_fma(e1x, qx, __builtin_elementwise_fma(e1y, qy, __builtin_elementwise_fma(e1z, qz, e1w)));
    const f2 s2 = __builtin_elementwise_fma(e2x, qx, __builtin_elementwise_fma(e2y, qy, __builtin_elementwise_fma(e2z, qz, e2w)));
    uint32_t pass = 0;
    {
        const bool away = tf.x < -sc.pre_m_test || pf.x > sc.pre_m_p;
        const bool outside = (s0.x < m.x || s1.x < m.x || s2.x < m.x) && ap0 > 1e-5f; // NaN -> maybe
        if (!away && !outside) pass |= 1u;
    }
    {
        const bool away = tf.y < -sc.pre_m_test || pf.y > sc.pre_m_p;
        const bool outside = (s0.y < m.y || s1.y < m.y || s2.y < m.y) && ap1 > 1e-5f;
        if (!away && !outside) pass |= 2u;
    }
    if (sc.has_spheres) { // a sphere's record has a zero normal and e0 = (centre, radius): the scalar routine knows it
        if (v0.x == 0.0f && v0.z == 0.0f && v1.x == 0.0f) {
            const bool ok = prefilter_pass(sc, float4{0.0f, 0.0f, 0.0f, v1.z}, float4{v2.x, v2.z, v3.x, v3.z},
                                           float4{v4.x, v4.z, v5.x, v5.z}, float4{v6.x, v6.z, v7.x, v7.z}, ox, oy, oz, dx, dy, dz);
            pass = (pass & ~1u) | (ok ? 1u : 0u);
        }
        if (v0.y == 0.0f && v0.w == 0.0f && v1.y == 0.0f) {
            const bool ok = prefilter_pass(sc, float4{0.0f, 0.0f, 0.0f, v1.w}, float4{v2.y, v2.w, v3.y, v3.w},
                                           float4{v4.y, v4.w, v5.y, v5.w}, float4{v6.y, v6.w, v7.y, v7.w}, ox, oy, oz, dx, dy, dz);
            pass = (pass & ~2u) | (ok ? 2u : 0u);
        }
    }
    return pass;
}

// The single-precision copy of a ray that the box tests and the screen use.
struct FRay {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz, oix, oiy, oiz;
    double shift; // the f32 origin is o + shift*dir
    bool pre_ok;  // the screen may be used (origin not moved)
    // The direction has unit length, as the interface asks: entry distances (ray parameters) are compared with
    // hit distances, and the screen's and the boxes' margins are derived for |dir| = 1. A ray with any other
    // direction (a NaN included) is handed to the reference-order walk instead, like an order-dependent ray.
    bool unit;
};
__device__ __forceinline__ FRay make_fray(const Scene &sc, const V3 &o, const V3 &dir) {
    FRay f;
    // Start the f32 ray next to the scene so its rounding error is bounded by
    // the scene's size, not by how far away the caller's origin is.
    V3 rc = {sc.root_c[0] - o.x, sc.root_c[1] - o.y, sc.root_c[2] - o.z};
    double shift = dot(rc, dir) - 2.0 * sc.root_r;
    shift = (shift > 0.0) ? shift : 0.0; // NaN -> 0
    f.shift = shift;
    f.pre_ok = shift == 0.0;
    f.unit = fabs(dot(dir, dir) - 1.0) <= 1e-6;
    f.ox = (float)(o.x + shift * dir.x);
    f.oy = (float)(o.y + shift * dir.y);
    f.oz = (float)(o.z + shift * dir.z);
    f.dx = (float)dir.x; f.dy = (float)dir.y; f.dz = (float)dir.z;
    // a zero component would give inf - inf in the fused slab test
    float sx = fabsf(f.dx) < 1e-18f ? copysignf(1e-18f, f.dx) : f.dx;
    float sy = fabsf(f.dy) < 1e-18f ? copysignf(1e-18f, f.dy) : f.dy;
    float sz = fabsf(f.dz) < 1e-18f ? copysignf(1e-18f, f.dz) : f.dz;
    f.ix = 1.0f / sx; f.iy = 1.0f / sy; f.iz = 1.0f / sz;
    f.oix = f.ox * f.ix; f.oiy = f.oy * f.iy; f.oiz = f.oz * f.iz;
    return f;
}

#ifdef RT_COUNT_VISITS
// Counting build: how much of the screen's margins a TRUE hit of a polygon uses (the reference-order walk asks
// this for every hit it finds): the largest s / m over the three edge half-spaces (1 = the screen would have
// thrown the hit away), and the same for the two plane tests. Kept as float bits in visits[92..94] (atomic
// maximum: the values are positive).
__device__ inline void screen_margin_use(const Scene &sc, uint32_t slot, const FRay &f) {
    if (!f.pre_ok) return;
    const PrimPre &q = sc.pre[slot];
    if ((q.nx == 0.0f && q.ny == 0.0f && q.nz == 0.0f) || !(q.nx == q.nx)) return; // a sphere, or no record
    const float pf = fmaf(q.nx, f.dx, fmaf(q.ny, f.dy, q.nz * f.dz));
    const float tf = fmaf(q.nx, f.ox, fmaf(q.ny, f.oy, q.nz * f.oz)) - q.k;
    const float ap = fabsf(pf), inv = __builtin_amdgcn_rcpf(ap), t = fabsf(tf) * inv;
    const float qx = fmaf(t, f.dx, f.ox), qy = fmaf(t, f.dy, f.oy), qz = fmaf(t, f.dz, f.oz);
    const float m = -2.0f * fmaf(sc.pre_c2, inv, sc.pre_c1);
    float use = 0.0f;
    if (ap > 1e-5f)
        for (int k = 0; k < 3; k++) {
            const float sk = fmaf(q.e[k][0], qx, fmaf(q.e[k][1], qy, fmaf(q.e[k][2], qz, q.e[k][3])));
            if (sk < 0.0f) use = fmaxf(use, sk / m);
        }
    atomicMax(reinterpret_cast<unsigned int *>(sc.visits + 92), __float_as_uint(use));
    if (tf < 0.0f) atomicMax(reinterpret_cast<unsigned int *>(sc.visits + 93), __float_as_uint(tf / -sc.pre_m_test));
    if (pf > 0.0f) atomicMax(reinterpret_cast<unsigned int *>(sc.visits + 94), __float_as_uint(pf / sc.pre_m_p));
}
#endif

// The lanes for which `p` holds, as a mask. (The HIP header's __ballot turns the predicate into an integer and
// compares that with zero: two vector instructions per vote; the builtin yields the compare's own lane mask.)
__device__ __forceinline__ unsigned long long lane_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// The rays a wavefront works through: ranges of `per_range` consecutive rays. With a `ticket` counter
// (zero when the launch starts) the ranges are handed out in the order the wavefronts ask for them, so the
// grid stays busy until the rays run out: with a fixed share of ranges per wavefront (first range `first`,
// every further one `stride` later; used when no counter is given) the wavefronts of the persistent grid
// finish up to a fifth of the launch apart (measured: wave residency 78 % of the kernel's duration; handing
// the ranges out on demand made the 1080p render 10 % and the config-5 render 17 % faster).
// I: the type the ray numbers are kept in (32 bits where a launch has fewer than 2^32 rays: six scalar registers
// less through the walk's loop).
template <class I>
struct RaySupplyT {
    I next, end, range, stride, per_range, n;
    unsigned long long *ticket;
    __device__ __forceinline__ I draw() {
        unsigned long long r = 0;
        if (lane_rank(~0ull) == 0) r = atomicAdd(ticket, 1ull);
        if (sizeof(I) == 4) return (I)__builtin_amdgcn_readfirstlane((uint32_t)r);
        return (I)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(r >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)r));
    }
    __device__ __forceinline__ void init(I first, I stride_, I per_range_, I n_,
                                         unsigned long long *ticket_ = nullptr) {
        range = first; stride = stride_; per_range = per_range_; n = n_; ticket = ticket_;
        if (ticket) range = draw();
        open();
    }
    __device__ __forceinline__ void open() {
        next = range * per_range;
        if (next >= n) {
            next = end = n;
        } else {
            end = next + per_range < n ? next + per_range : n;
        }
    }
    __device__ __forceinline__ I available() {
        if (next == end && end != n) {
            range = ticket ? draw() : range + stride;
            open();
        }
        return end - next;
    }
};
using RaySupply = RaySupplyT<uint64_t>;

// Closest hits for the rays a wavefront draws from `supply`.
//
// A ray alternates between three kinds of work of very different cost: a node step
// (f32, one 64-byte record), a leaf screen (f32, one 64-byte record per primitive)
// and an exact primitive test (f64). Run as nested loops, each kind executes with
// only the few lanes that happen to need it at that moment. Instead every lane
// carries its own state and the wavefront votes, once per round, on which kind of
// work to run next: the expensive kinds wait until enough lanes have queued up for
// them (sc.vote_*), node steps run otherwise. Lanes that finish a ray store it and
// take the next unprocessed ray, so a wave stays full although individual rays need
// anything from 1 to 100+ steps.
// stack_ref/stack_t: per-lane LDS stacks (element stride TBLOCK) of children still to visit
// and their entry distances, 4 + 2 bytes per entry.
// Index: the type a lane keeps its ray's number in (uint32_t where a launch has fewer than 2^32 rays: one
// register less through every round).
template <class Index = uint64_t, class IO>
__device__ __forceinline__ void traverse_fast(const Scene &sc, const IO &io, RaySupplyT<Index> supply, uint32_t *stack_ref,
                                              uint16_t *stack_t, double *best_pos, const uint32_t vote_exact,
                                              const uint32_t vote_leaf, const uint32_t vote_refill) {
    bool active = false;
    Index ri = 0;
    // The f64 ray is NOT kept in registers between rounds: only the exact test needs it (0.9 times per ray),
    // which reads it again from the ray array. Twelve registers less through every node step.
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, oix = 0, oiy = 0, oiz = 0;
    float tlimit = 0;
    // pre_ok: the origin was not moved (make_fray), i.e. the f32 ray starts where the f64 ray does: shift == 0. The
    // shift of a moved origin is not kept (two registers): the exact round works it out again where it needs it.
    bool pre_ok = false;
    bool best_have = false; // the running closest hit (bvh.rs:100-101); its position sits in LDS (best_pos)
    uint32_t best_slot = RT_MISS;
    double best_dist = 0.0;
    int sp = 0;
    uint32_t cur = REF_NONE; // inner node, leaf, or REF_NONE = ray finished
    uint32_t cand = 0;       // primitives of leaf `cur` that passed the screen and await the exact test
    // A sphere returned a hit behind the origin (or a NaN): the reference's answer for this ray depends on
    // its visiting order, so the ray is only marked here (REWALK in place of its result) and a second,
    // small kernel walks the marked rays in reference order. A sphere that can do this contains the
    // origin, so its box is entered at distance 0 and the walk cannot cull it.
    bool order_dependent = false;
#ifdef RT_COUNT_VISITS
    uint32_t n_steps = 0, n_screened = 0, n_tested = 0;
#define RT_VISIT(x) (x)
    // wave-level rounds of each kind and the lanes that took part: [4..5] exact, [6..7] leaf,
    // [8..9] node iterations, [10..11] refill rounds and rays handed out
    auto tally = [&](int k, bool lane_takes_part) {
        unsigned long long m = lane_mask(lane_takes_part);
        if (lane_rank(~0ull) == 0) {
            atomicAdd(sc.visits + k, 1ull);
            atomicAdd(sc.visits + k + 1, (unsigned long long)__popcll(m));
        }
        // census of what every lane's ray waits for while this round runs: [100 + 2k + j], j = 0 nothing (no
        // ray, or a finished one that waits to be stored), 1 node step, 2 leaf screen, 3 exact test
        if (k < 10) {
            const bool c_idle = !active || cur == REF_NONE;
            const bool c_exact = !c_idle && cand != 0, c_leaf = !c_idle && cand == 0 && (cur & REF_LEAF);
            const bool c_node = !c_idle && !(cur & REF_LEAF);
            const unsigned long long mi = lane_mask(c_idle), mn = lane_mask(c_node), ml = lane_mask(c_leaf), me = lane_mask(c_exact);
            if (lane_rank(~0ull) == 0) {
                atomicAdd(sc.visits + 100 + 2 * k + 0, (unsigned long long)__popcll(mi));
                atomicAdd(sc.visits + 100 + 2 * k + 1, (unsigned long long)__popcll(mn));
                atomicAdd(sc.visits + 100 + 2 * k + 2, (unsigned long long)__popcll(ml));
                atomicAdd(sc.visits + 100 + 2 * k + 3, (unsigned long long)__popcll(me));
            }
        }
    };
#else
#define RT_VISIT(x) ((void)0)
#endif
#ifdef RT_TIME_ROUNDS
    // timing build only: wave cycles (s_memtime) spent in rounds of each kind, summed in registers and
    // added to sc.visits once per wavefront: [12] refill, [13] exact, [14] leaf, [15] node, then the rounds
    unsigned long long lap_t = __builtin_amdgcn_s_memtime(), lap_acc[4] = {0, 0, 0, 0}, lap_n[4] = {0, 0, 0, 0};
    unsigned long long sub_t[2] = {0, 0}, sub_acc[4] = {0, 0, 0, 0};
    // [32 + 8 * s ...], s = the launch's ticket counter modulo 16: latest start (as 2^62 - start, so that zero
    // is "none"), latest end, sum of starts, sum of ends, wavefronts; in ticks of the 100 MHz wall clock
    const unsigned long long wall_start = wall_clock64();
    auto lap = [&](int kind) {
        unsigned long long now = __builtin_amdgcn_s_memtime();
        lap_acc[kind] += now - lap_t;
        lap_n[kind] += 1;
        lap_t = now;
    };
#define RT_LAP(k) lap(k)
#else
#define RT_LAP(k) ((void)0)
#endif

    // Entry distances are kept as the upper 16 bits of the float, i.e. rounded toward zero:
    // a stored distance never exceeds the true one, so a pop can only keep too much.
    auto pop = [&]() -> uint32_t {
        while (sp > 0) {
            sp -= 1;
            if (__uint_as_float((uint32_t)stack_t[sp * TBLOCK] << 16) <= tlimit) return stack_ref[sp * TBLOCK];
        }
        return REF_NONE;
    };

    for (;;) {
        // A lane without a ray has cur == REF_NONE (which carries the leaf bit) and cand == 0, and cand != 0 only
        // while cur names a leaf; all 64 lanes of the wavefront are alive. The votes are therefore taken on three
        // plain comparisons and combined as 64-bit masks (a vote on a combination of predicates costs two more
        // vector instructions each).
        const bool at_rest = cur == REF_NONE, leaf_bit = (int32_t)cur < 0;
        bool is_exact = cand != 0;
        bool is_leaf = leaf_bit && !at_rest && !is_exact;
        bool is_node = !leaf_bit;
        const unsigned long long m_exact = lane_mask(is_exact), m_leaf_bit = lane_mask(leaf_bit);
        const unsigned long long idle = lane_mask(at_rest); // no ray, or a finished one
        uint32_t n_exact = __popcll(m_exact);
        uint32_t n_leaf = __popcll(m_leaf_bit & ~idle & ~m_exact);
        uint32_t n_node = __popcll(~m_leaf_bit);

        // ---- store finished rays, hand new rays to idle lanes -----------------
        bool finished = active && at_rest;
        uint32_t n_idle = __popcll(idle);
        if (n_idle >= vote_refill || n_exact + n_leaf + n_node == 0) {
            // The next rays are requested first and the finished ones stored while those loads are in
            // flight: the two trips to memory (the winning primitive's record for the stores, the new
            // rays) overlap instead of following each other.
            const Index avail = supply.available();
            const uint32_t rank = lane_rank(idle);
            const bool take = at_rest && rank < avail;
            const Index ri_new = supply.next + rank;
            V3 o_new = {0.0, 0.0, 0.0}, dir_new = {0.0, 0.0, 0.0};
            if (take) io.load(ri_new, o_new, dir_new);
            if (finished) {
                if (order_dependent) {
                    io.store_flag(ri); // k_rewalk_* repeats this ray in reference order
                } else {
                    Closest best = {best_have, best_have ? best_slot : RT_MISS, best_have ? best_dist : 0.0, {0.0, 0.0, 0.0}};
                    if (best_have) best.pos = V3{best_pos[0], best_pos[TBLOCK], best_pos[2 * TBLOCK]};
                    io.store(sc, ri, best);
                }
                active = false;
#ifdef RT_COUNT_VISITS
                atomicAdd(sc.visits + 0, 1ull);
                atomicAdd(sc.visits + 1, (unsigned long long)n_steps);
                atomicAdd(sc.visits + 2, (unsigned long long)n_screened);
                atomicAdd(sc.visits + 3, (unsigned long long)n_tested);
                n_steps = n_screened = n_tested = 0;
#endif
            }
            if (avail != 0) {
#ifdef RT_COUNT_VISITS
                tally(10, take);
#endif
                if (take) {
                    ri = (Index)ri_new;
                    const FRay fr = make_fray(sc, o_new, dir_new);
                    pre_ok = fr.pre_ok;
                    ox = fr.ox; oy = fr.oy; oz = fr.oz; dx = fr.dx; dy = fr.dy; dz = fr.dz;
                    ix = fr.ix; iy = fr.iy; iz = fr.iz; oix = fr.oix; oiy = fr.oiy; oiz = fr.oiz;
                    tlimit = __builtin_inff();
                    best_have = false;
                    order_dependent = !fr.unit;
                    sp = 0;
                    cand = 0;
                    cur = fr.unit ? sc.root_ref : REF_NONE;
                    active = true;
                }
                supply.next += (n_idle < avail) ? n_idle : avail;
                RT_LAP(0);
                continue; // vote again with the new rays on board
            }
            if (lane_mask(active) == 0ull) { // supply exhausted and every ray stored
#ifdef RT_TIME_ROUNDS
                if (lane_rank(~0ull) == 0)
                    for (int k = 0; k < 4; k++) {
                        atomicAdd(sc.visits + 12 + k, lap_acc[k]);
                        atomicAdd(sc.visits + 16 + k, lap_n[k]);
                    }
                for (int k = 0; k < 4; k++) atomicAdd(sc.visits + 28 + k, sub_acc[k]);
                if (lane_rank(~0ull) == 0) {
                    const unsigned long long wall_end = wall_clock64();
                    unsigned long long *w = sc.visits + 32 + 8 * (((uintptr_t)supply.ticket >> 3) & 15u);
                    atomicMax(w + 0, (1ull << 62) - wall_start);
                    atomicMax(w + 1, wall_end);
                    atomicAdd(w + 2, wall_start);
                    atomicAdd(w + 3, wall_end);
                    atomicAdd(w + 4, 1ull);
                }
#endif
                break;
            }
            if (n_exact + n_leaf + n_node == 0) continue;
        }

        if (n_exact >= vote_exact || (n_exact != 0 && n_leaf == 0 && n_node == 0)) {
            // ---- exact f64 test of one queued primitive per lane -------------------
#ifdef RT_COUNT_VISITS
            tally(4, is_exact);
#endif
            if (is_exact) {
                uint32_t k = __ffs(cand) - 1;
                cand &= cand - 1;
                uint32_t slot = (cur & REF_SLOT_MASK) + k;
                PrimHot h = sc.hot[slot];
                V3 o, dir;
                double step;
                io.load(ri, o, dir);
                RT_VISIT(n_tested++);
#ifdef RT_TIME_ROUNDS
                asm volatile("" ::"v"(h.n_points), "v"(h.cz), "v"(o.x), "v"(dir.z)); // records and ray have arrived
                sub_t[0] = __builtin_amdgcn_s_memtime();
#endif
                bool is_hit = prim_intersect_step(sc, slot, h, o, dir, step, order_dependent);
#ifdef RT_TIME_ROUNDS
                asm volatile("" ::"v"(is_hit ? step : 0.0));
                sub_t[1] = __builtin_amdgcn_s_memtime();
#endif
                if (is_hit) {
                    const V3 pos = hit_position(o, dir, step);
                    double dd = mag(sub(pos, o));
                    RT_VISIT(atomicAdd(sc.visits + 90, 1ull)); // exact tests that hit, [91] those that became the closest
                    // strict '<' keeps the first hit in reference visiting order; reference slots
                    // are numbered in that order, so an exact tie goes to the smaller one
                    if (!best_have || dd < best_dist ||
                        (dd == best_dist && sc.ref_of_fast[slot] < sc.ref_of_fast[best_slot])) {
                        RT_VISIT(atomicAdd(sc.visits + 91, 1ull));
                        best_have = true;
                        best_slot = slot;
                        best_dist = dd;
                        best_pos[0] = pos.x; best_pos[TBLOCK] = pos.y; best_pos[2 * TBLOCK] = pos.z;
                        double shift = 0.0;
                        if (!pre_ok) { // make_fray's expression
                            const V3 rc = {sc.root_c[0] - o.x, sc.root_c[1] - o.y, sc.root_c[2] - o.z};
                            shift = dot(rc, dir) - 2.0 * sc.root_r;
                            shift = (shift > 0.0) ? shift : 0.0;
                        }
                        tlimit = prune_limit(sc, dd, shift);
                    }
                }
                if (cand == 0) cur = pop();
            }
#ifdef RT_TIME_ROUNDS
            // [28] waiting for the records and the ray, [29] the f64 test, [30] the update and the pop
            if (is_exact) { // per lane: a lane only knows the times of the rounds it took part in
                sub_acc[0] += sub_t[0] - lap_t;
                sub_acc[1] += sub_t[1] - sub_t[0];
                sub_acc[2] += __builtin_amdgcn_s_memtime() - sub_t[1];
                sub_acc[3] += 1;
            }
#endif
            RT_LAP(1);
        } else if (n_leaf >= vote_leaf || (n_leaf != 0 && n_node == 0)) {
            // ---- leaf: f32 screen of its primitives ---------------------------------
#ifdef RT_COUNT_VISITS
            tally(6, is_leaf);
#endif
            if (is_leaf) {
                uint32_t sb = cur & REF_SLOT_MASK;
                uint32_t cnt = ((cur >> REF_CNT_SHIFT) & 7u) + 1u;
                uint32_t c = (1u << cnt) - 1u;
                RT_VISIT(n_screened += cnt);
                if (pre_ok) {
                    // two primitives per step; the builder's leaves hold one or two, so the loop is for the
                    // reference topology's larger leaves only (and stays out of the common path's block, where
                    // the compiler can then fold the ray's broadcast operands into the packed instructions)
                    const uint32_t first = prefilter_pass2(sc, reinterpret_cast<const float4 *>(sc.pre2 + sb), ox, oy, oz, dx, dy, dz);
                    c = cnt > 1u ? first : first & 1u;
                    if (cnt > 2u)
                        for (uint32_t k = 2; k < cnt; k += 2) {
                            const uint32_t two = prefilter_pass2(sc, reinterpret_cast<const float4 *>(sc.pre2 + (sb + k)), ox,
                                                                 oy, oz, dx, dy, dz);
                            c |= (k + 1 < cnt ? two : two & 1u) << k;
                        }
                }
                cand = c;
                if (cand == 0) cur = pop();
            }
            RT_LAP(2);
        } else {
            // ---- node steps: up to four children of a node come in one 64-byte record ----
            // The hit children are sorted by entry distance; the walk descends into the nearest and
            // the others wait on the stack. The stack holds stack_depth entries per lane and cannot
            // overflow: a child reference carries `need`, the number of wide nodes on the longest
            // path below it, and the siblings are pushed one by one only while that leaves one free
            // entry for every level still to come. Otherwise ONE group entry (node, mask of the
            // children still to visit, distance of the nearest of them) takes their place; popping
            // it repeats the node step for those children. One entry per level always fits.
            const uint32_t cap = sc.stack_depth;
            uint32_t keep = n_node - (n_node * sc.node_drop >> 4);
            for (uint32_t it = 0; it < sc.node_iters; it++) {
#ifdef RT_COUNT_VISITS
                tally(8, is_node);
#endif
                if (is_node) {
                    const bool grp = (cur & REF_GROUP) != 0u;
                    const uint32_t idx = cur & WREF_INDEX_MASK;
                    const uint32_t need = (cur >> WREF_NEED_SHIFT) & 31u; // meaningless for a group entry
                    RT_VISIT(n_steps++);
                    const NodeHits h = node_hits(sc.qnodes + idx, ix, iy, iz, oix, oiy, oiz, tlimit, grp,
                                                 (cur >> WREF_MASK_SHIFT) & 0xFu);
                    if (h.k1 != NO_CHILD) { // more than one child: the others wait on the stack
                        // (three free entries are asked for, however many of the three are pushed)
                        if (!grp && (uint32_t)sp + 3u + need <= cap + 1u) {
                            if (h.k3 != NO_CHILD) { // farthest first, so that the nearest is popped first
                                stack_ref[sp * TBLOCK] = h.r3;
                                stack_t[sp * TBLOCK] = (uint16_t)(h.k3 >> 16);
                                sp += 1;
                            }
                            if (h.k2 != NO_CHILD) {
                                stack_ref[sp * TBLOCK] = h.r2;
                                stack_t[sp * TBLOCK] = (uint16_t)(h.k2 >> 16);
                                sp += 1;
                            }
                            stack_ref[sp * TBLOCK] = h.r1;
                            stack_t[sp * TBLOCK] = (uint16_t)(h.k1 >> 16);
                            sp += 1;
                        } else {
                            uint32_t rest = (1u << (h.k1 & 3u)) | (h.k2 != NO_CHILD ? 1u << (h.k2 & 3u) : 0u) |
                                            (h.k3 != NO_CHILD ? 1u << (h.k3 & 3u) : 0u);
                            stack_ref[sp * TBLOCK] = REF_GROUP | (rest << WREF_MASK_SHIFT) | idx;
                            stack_t[sp * TBLOCK] = (uint16_t)(h.k1 >> 16);
                            sp += 1;
                        }
                    }
                    cur = (h.k0 != NO_CHILD) ? h.r0 : pop();
                    is_node = !(cur & REF_LEAF);
                }
                if ((uint32_t)__popcll(lane_mask(is_node)) < keep) break;
            }
            RT_LAP(3);
        }
    }
}

// ---- the reference's answer for an order-dependent ray ----------------------------------------
// Walking one such ray in reference order costs tens of milliseconds in a large scene (thousands
// of dependent node visits by a single lane). The same answer follows from much less work:
//  1. every primitive the ray hits at all, found with the fast hierarchy (no distance culling);
//  2. the reference's decisions replayed on those few hits in its visiting order (reference slot
//     order): a hit counts only if no node on the way from the root to its leaf is skipped, and a
//     node is skipped if its shell is missed (bvh.rs:110) or pruned against the closest hit the
//     reference holds at the moment it pops that node (bvh.rs:106-109), i.e. the last accepted hit
//     among the primitives that come before the node's first slot (rt_node.reserved[0]).
struct RefHit {
    uint32_t rslot, fslot;
    double dist;
    V3 pos;
};
constexpr int REWALK_MAX_HITS = 32;
constexpr int REWALK_STACK = 48;

// One pass over the fast hierarchy for all_hits. ORIGIN_ONLY: only boxes that contain the ray's
// origin are entered and nothing is recorded; the pass reports the largest distance of a hit
// behind the origin and whether some hit has a NaN distance. Otherwise every hit within `reach`
// (and every hit in a box around the origin) is recorded; `reach` shrinks to the closest hit in
// front of the origin found so far, but never below `keep`.
// false: more hits than REWALK_MAX_HITS or a stack overflow.
template <bool ORIGIN_ONLY>
__device__ bool hits_pass(const Scene &sc, const FRay &f, const V3 &o, const V3 &dir, double keep, RefHit *hits,
                          int &n_hits, double &far_behind, bool &any_nan) {
    uint32_t stack[REWALK_STACK];
    float stack_t[REWALK_STACK];
    int sp = 0;
    uint32_t cur = sc.root_ref;
    float cur_t = 0.0f;
    double reach = __builtin_inf();
    float tlimit = __builtin_inff();
    while (cur != REF_NONE) {
        if (cur_t <= tlimit) {
            if (cur & REF_LEAF) {
                const uint32_t sb = cur & REF_SLOT_MASK, cnt = ((cur >> REF_CNT_SHIFT) & 7u) + 1u;
                for (uint32_t k = 0; k < cnt; k++) {
                    const uint32_t slot = sb + k;
                    if (f.pre_ok) {
                        const float4 *pr = reinterpret_cast<const float4 *>(sc.pre + slot);
                        if (!prefilter_pass(sc, pr[0], pr[1], pr[2], pr[3], f.ox, f.oy, f.oz, f.dx, f.dy, f.dz)) continue;
                    }
                    PrimHot h = sc.hot[slot];
                    V3 pos;
                    bool behind = false;
                    if (!prim_intersect(sc, slot, h, o, dir, pos, behind)) continue;
                    const double dist = mag(sub(pos, o));
                    if (ORIGIN_ONLY) {
                        if (!(dist == dist)) any_nan = true;
                        else if (behind && dist > far_behind) far_behind = dist;
                        continue;
                    }
                    if (n_hits == REWALK_MAX_HITS) return false;
                    hits[n_hits++] = RefHit{sc.ref_of_fast[slot], slot, dist, pos};
                    if (!behind && dist < reach) {
                        reach = dist > keep ? dist : keep;
                        tlimit = prune_limit(sc, reach, f.shift);
                    }
                }
            } else {
                // ORIGIN_ONLY: the ray must start inside the box (entry distance 0)
                const NodeHits h = node_hits(sc.qnodes + (cur & WREF_INDEX_MASK), f.ix, f.iy, f.iz, f.oix, f.oiy, f.oiz,
                                             ORIGIN_ONLY ? 0.0f : tlimit, false, 0xFu);
                if (sp + 4 > REWALK_STACK) return false;
                // the farthest child is pushed first, so that the nearest is looked at first and the reach
                // shrinks early
                if (h.k3 != NO_CHILD) { stack[sp] = h.r3; stack_t[sp] = __uint_as_float(h.k3 & ~3u); sp++; }
                if (h.k2 != NO_CHILD) { stack[sp] = h.r2; stack_t[sp] = __uint_as_float(h.k2 & ~3u); sp++; }
                if (h.k1 != NO_CHILD) { stack[sp] = h.r1; stack_t[sp] = __uint_as_float(h.k1 & ~3u); sp++; }
                if (h.k0 != NO_CHILD) { stack[sp] = h.r0; stack_t[sp] = __uint_as_float(h.k0 & ~3u); sp++; }
            }
        }
        if (sp) {
            sp--;
            cur = stack[sp];
            cur_t = stack_t[sp];
        } else {
            cur = REF_NONE;
        }
    }
    return true;
}

// Every hit of the ray that can take part in the reference's decisions. A hit in front of the origin
// never hides a closer one (the reference's prune ball around it reaches back to the origin), so hits
// in front matter only up to the closest of them, or up to the farthest hit BEHIND the origin if that
// is larger (a hit behind can hide the closer ones and let a farther one through). A NaN distance
// wins only if nothing at all precedes it, so then every hit is needed.
// false: too many hits (the caller falls back to the plain reference-order walk).
__device__ bool all_hits(const Scene &sc, const V3 &o, const V3 &dir, RefHit *hits, int &n_hits) {
    const FRay f = make_fray(sc, o, dir);
    if (!f.unit) return false; // the fast hierarchy is of no use for such a ray (FRay::unit)
    double far_behind = 0.0;
    bool any_nan = false;
    n_hits = 0;
    if (!hits_pass<true>(sc, f, o, dir, 0.0, hits, n_hits, far_behind, any_nan)) return false;
    return hits_pass<false>(sc, f, o, dir, any_nan ? __builtin_inf() : far_behind, hits, n_hits, far_behind, any_nan);
}

__device__ inline Closest reference_answer(const Scene &sc, const V3 &o, const V3 &dir) {
    RefHit hits[REWALK_MAX_HITS];
    int n = 0;
#ifdef RT_TIME_ROUNDS
    // timing build: [20] marked rays, [21] fallbacks, [22..24] cycles in the two passes and the replay (sum),
    // [25..27] the same, largest single ray
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    if (!all_hits(sc, o, dir, hits, n)) {
#ifdef RT_TIME_ROUNDS
        atomicAdd(sc.visits + 21, 1ull);
#endif
        return traverse_reference_order(sc, o, dir);
    }
#ifdef RT_TIME_ROUNDS
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
    for (int a = 1; a < n; a++) { // by reference slot: the order in which the reference meets them
        RefHit key = hits[a];
        int b = a - 1;
        while (b >= 0 && hits[b].rslot > key.rslot) {
            hits[b + 1] = hits[b];
            b--;
        }
        hits[b + 1] = key;
    }
    uint8_t accepted[REWALK_MAX_HITS]; // indices of the hits that became the closest hit, in order
    int n_acc = 0;
    for (int k = 0; k < n; k++) {
        const uint32_t s = hits[k].rslot;
        bool reached = true;
        for (uint32_t i = 0;;) { // from the root to the leaf that holds slot s
            const rt_node &nd = sc.nodes[i];
            double shell[4] = {nd.shell[0], nd.shell[1], nd.shell[2], nd.shell[3]};
            const uint32_t first = nd.reserved[0], left = nd.left;
            int b = -1; // the closest hit the reference holds when it pops node i
            for (int a = n_acc - 1; a >= 0; a--)
                if (hits[accepted[a]].rslot < first) {
                    b = accepted[a];
                    break;
                }
            if ((b >= 0 && shell_pruned(shell, hits[b].pos, hits[b].dist)) || !shell_intersect(shell, o, dir)) {
                reached = false;
                break;
            }
            if (left == 0) break;
            i = (s >= sc.nodes[left].reserved[0]) ? left : i + 1;
        }
        if (!reached) continue;
        // bvh.rs:116-127: the first hit, or strictly closer than the closest so far
        if (n_acc == 0 || hits[k].dist < hits[accepted[n_acc - 1]].dist) accepted[n_acc++] = (uint8_t)k;
    }
#ifdef RT_TIME_ROUNDS
    {
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        atomicAdd(sc.visits + 20, 1ull);
        atomicAdd(sc.visits + 22, t1 - t0);
        atomicAdd(sc.visits + 24, t2 - t1);
        atomicMax(sc.visits + 25, t1 - t0);
        atomicMax(sc.visits + 27, t2 - t1);
        atomicAdd(sc.visits + 23, (unsigned long long)n);
    }
#endif
    if (n_acc == 0) return no_hit();
    const RefHit &w = hits[accepted[n_acc - 1]];
    return Closest{true, w.fslot, w.dist, w.pos};
}

// In place of a ray's primitive id (material, for paths): "walk this ray again in reference order".
// Never a valid id or material index (both stay below 2^28).
constexpr uint32_t REWALK = 0xFFFFFFFEu;

// rays as n x 6 doubles in, ids / distances / positions / normals out
struct BatchIO {
    const double *rays;
    uint32_t *prim_id;
    double *distance, *pos, *normal;
    uint32_t *marked; // may be NULL; set to 1 when a ray is marked for the re-walk: the re-walk of a launch that marked none has nothing to scan
    // rays and results stream through once: non-temporal accesses leave the caches to the scene
    __device__ __forceinline__ void load(uint64_t i, V3 &o, V3 &dir) const {
        const double *r = rays + 6 * i;
        o = V3{__builtin_nontemporal_load(r), __builtin_nontemporal_load(r + 1), __builtin_nontemporal_load(r + 2)};
        dir = V3{__builtin_nontemporal_load(r + 3), __builtin_nontemporal_load(r + 4), __builtin_nontemporal_load(r + 5)};
    }
    __device__ __forceinline__ void store_flag(uint64_t i) const {
        prim_id[i] = REWALK;
        if (marked) *marked = 1u;
    }
    __device__ __forceinline__ bool flagged(uint64_t i) const { return prim_id[i] == REWALK; }
    __device__ __forceinline__ bool needs_pos() const { return pos != nullptr || normal != nullptr; }
    __device__ __forceinline__ void store(const Scene &sc, uint64_t i, const Closest &best) const {
        // ids and distances are what most callers ask for: the winner's record is then read for its id only
        __builtin_nontemporal_store(best.have ? sc.hot[best.slot].prim_id : RT_MISS, prim_id + i);
        if (distance) __builtin_nontemporal_store(best.dist, distance + i);
        if (pos) {
            pos[3 * i] = best.pos.x; pos[3 * i + 1] = best.pos.y; pos[3 * i + 2] = best.pos.z;
        }
        if (normal) {
            HitInfo h = hit_info(sc, best);
            normal[3 * i] = h.nrm.x; normal[3 * i + 1] = h.nrm.y; normal[3 * i + 2] = h.nrm.z;
        }
    }
};

// the live paths of one bounce of the wavefront integrator
struct PathIO {
    const double *bundle; // bounce 0 of a film pass: rays as n x 6, repeating every `period` paths; else NULL
    uint32_t period;
    const double *ray; // this bounce's ray buffer: 6 planes of `cap` doubles
    double *hit;       // 3 planes: hit position
    uint32_t *hit_mat; // slot of the primitive that was hit, RT_MISS, or REWALK
    uint64_t cap;
    __device__ __forceinline__ void load(uint64_t t, V3 &o, V3 &dir) const {
        if (bundle) { // every sample of a pass re-reads the same camera rays: leave them cacheable
            const double *r = bundle + 6 * (size_t)((uint32_t)t % period);
            o = V3{r[0], r[1], r[2]};
            dir = V3{r[3], r[4], r[5]};
            return;
        }
        o = V3{__builtin_nontemporal_load(ray + t), __builtin_nontemporal_load(ray + cap + t),
               __builtin_nontemporal_load(ray + 2 * cap + t)};
        dir = V3{__builtin_nontemporal_load(ray + 3 * cap + t), __builtin_nontemporal_load(ray + 4 * cap + t),
                 __builtin_nontemporal_load(ray + 5 * cap + t)};
    }
    __device__ __forceinline__ void store_flag(uint64_t t) const { hit_mat[t] = REWALK; }
    __device__ __forceinline__ bool flagged(uint64_t t) const { return hit_mat[t] == REWALK; }
    __device__ __forceinline__ bool needs_pos() const { return true; }
    // The walk hands over WHERE it hit and WHAT (the primitive's slot); normal and material are read from the
    // primitive's record by the shading kernel, where every lane has a hit to look up, instead of here, where a
    // refill round would wait for that record with half of its lanes idle.
    __device__ __forceinline__ void store(const Scene &, uint64_t t, const Closest &best) const {
        __builtin_nontemporal_store(best.have ? best.slot : RT_MISS, hit_mat + t);
        if (best.have) {
            __builtin_nontemporal_store(best.pos.x, hit + t);
            __builtin_nontemporal_store(best.pos.y, hit + cap + t);
            __builtin_nontemporal_store(best.pos.z, hit + 2 * cap + t);
        }
    }
};
__device__ __forceinline__ PathIO path_io(const Paths &p, uint32_t depth) {
    return {depth == 0 ? p.first_bundle : nullptr, p.first_period, p.rays + (size_t)(depth & 1) * 6 * p.capacity, p.hit,
            p.hit_mat, p.capacity};
}

// ---- Philox4x32-10 counter-based generator -----------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
constexpr uint32_t STREAM_BSDF = 0u, STREAM_LENS = 1u, STREAM_GLOSS = 2u;
constexpr uint32_t GLOSS_MAX_BLOCKS = 64u;

// two uniforms in [0,1) with 53 random bits each, like rand's gen::<f64>()
__device__ __forceinline__ void uniform2(uint64_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d, double &u0,
                                         double &u1) {
    uint32_t r[4];
    philox4x32_10(a, b, c, d, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    uint64_t x = ((uint64_t)r[0] << 32) | r[1];
    uint64_t y = ((uint64_t)r[2] << 32) | r[3];
    u0 = (double)(x >> 11) * (1.0 / 9007199254740992.0);
    u1 = (double)(y >> 11) * (1.0 / 9007199254740992.0);
}

// brdf.rs:14-29 with geometry.rs:109-130
__device__ __forceinline__ V3 brdf_sample(const V3 &normal, double theta_raw, double phi_raw) {
    double theta = PI * 2.0 * theta_raw;
    // geometry.rs:113-117 takes phi = acos(phi_raw) and then sin(phi) and cos(phi). With phi_raw in
    // [0, 1) these are sqrt(1 - phi_raw^2) and phi_raw themselves; evaluated this way they lie within
    // the last-bit spread that two math libraries already show on acos/sin/cos, at a tenth of the
    // instructions. sincos shares the argument reduction of the two calls made on theta.
    double sp = sqrt((1.0 - phi_raw) * (1.0 + phi_raw)), cp = phi_raw, st, ct;
    sincos(theta, &st, &ct);
    V3 out_raw = {ct * sp, st * sp, cp};
    // unit_perp_vec
    V3 v = {1.0, 0.0, 0.0};
    V3 other = {0.0, 1.0, 0.0};
    if (fabs(dot(normal, v)) > fabs(dot(normal, other))) v = other;
    V3 new_x = normalize(cross(normal, v));
    V3 new_y = cross(normal, new_x);
    return add(scale(out_raw.x, new_x), add(scale(out_raw.y, new_y), scale(out_raw.z, normal)));
}

// The variable-glossy lobe of the reference's JavaScript tracer (geometry.js:268-272,300-318):
// the mirror direction perturbed by a Gaussian angle of width (2n)^2, n = 1 - gloss, redrawn
// until it leaves the surface. Uniforms: Philox blocks (pixel, sample, j | STREAM_GLOSS << 8,
// depth | block << 8), one per candidate point of the ball, one per Gaussian (gauss-random's
// Box-Muller form). Operation order as in the CPU oracle.
__device__ __forceinline__ V3 glossy_sample(double gloss, const V3 &in_dir, const V3 &normal, uint64_t seed,
                                            uint32_t pixel, uint32_t sample, uint32_t j, uint32_t depth) {
    V3 ref = sub(in_dir, scale(2.0, scale(dot(in_dir, normal) / dot(normal, normal), normal)));
    ref = normalize(ref);
    double n = 1.0 - gloss;
    double lobe = (2.0 * n) * (2.0 * n);
    uint32_t block = 0;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    while (block + 1 < GLOSS_MAX_BLOCKS) {
        uint32_t r[4];
        philox4x32_10(pixel, sample, j | (STREAM_GLOSS << 8), depth | (block << 8), k0, k1, r);
        block++;
        V3 rd = {(double)r[0] * (1.0 / 4294967296.0) - 0.5, (double)r[1] * (1.0 / 4294967296.0) - 0.5,
                 (double)r[2] * (1.0 / 4294967296.0) - 0.5};
        if (mag(rd) > 0.5) continue;
        V3 perp = sub(rd, scale(dot(rd, ref) / dot(ref, ref), ref));
        double pm = mag(perp);
        if (pm == 0.0) continue;
        perp = scale(1.0 / pm, perp);
        philox4x32_10(pixel, sample, j | (STREAM_GLOSS << 8), depth | (block << 8), k0, k1, r);
        block++;
        uint64_t x = ((uint64_t)r[0] << 32) | r[1];
        uint64_t y = ((uint64_t)r[2] << 32) | r[3];
        double u = (double)((x >> 11) + 1) * (1.0 / 9007199254740992.0);
        double v = (double)(y >> 11) * (1.0 / 9007199254740992.0);
        double g = sqrt(-2.0 * log(u)) * cos(2.0 * PI * v);
        double ang = g * lobe;
        V3 fin = add(scale(cos(ang), ref), scale(sin(ang), perp));
        if (dot(fin, normal) >= 0.0) return fin;
    }
    return ref;
}

// LDS of a fast-walk block: the lanes' stacks, references first, then the 16-bit distances
// and, in front of both, the position of every lane's closest hit so far (three planes of doubles): it is
// written when a hit is accepted and read when the ray is stored, so it need not occupy six registers
// through every node step
struct FastLds {
    uint32_t *sref;
    uint16_t *st;
    double *pos;
};
__device__ __forceinline__ FastLds fast_lds(const Scene &sc, uint32_t *lds) {
    FastLds f;
    f.pos = reinterpret_cast<double *>(lds) + threadIdx.x;
    uint32_t *stacks = lds + 6 * TBLOCK;
    f.sref = stacks + threadIdx.x;
    f.st = reinterpret_cast<uint16_t *>(stacks + (size_t)sc.stack_depth * TBLOCK) + threadIdx.x;
    return f;
}

// ---- host helpers: grid sizes ---------------------------------------------------
inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }
// two LDS stacks per lane (child references, entry distances) and the position of the closest hit
inline size_t stack_bytes(const Scene &sc) {
    size_t b = (size_t)sc.stack_depth * TBLOCK * (sizeof(uint32_t) + sizeof(uint16_t)) + 3 * sizeof(double) * TBLOCK;
    // experiment hook: extra LDS per block lowers the number of resident wavefronts
    static const size_t extra = [] {
        const char *e = getenv("RT_TUNE_LDS_EXTRA");
        return e ? (size_t)atol(e) : (size_t)0;
    }();
    return b + extra;
}
// The fast-walk kernels run as a persistent grid: exactly as many wavefronts as the device
// keeps resident, each working through many short ranges of rays (RaySupply), so that a
// wavefront drains only once, at the very end, instead of once per range. Measured on the
// 16.7 M-ray batch: +15 % over one range per wavefront, +20 % on the path tracer's launches.
// blocks_per_cu_limit != 0: at most that many of the blocks a CU holds (a launch that shares the device with
// another persistent launch on a second stream).
template <class K>
inline uint64_t resident_waves(K kernel, size_t lds_bytes, uint32_t blocks_per_cu_limit = 0) {
    struct Entry {
        int device;
        size_t lds;
        int per_cu, cus;
        const void *kernel;
    };
    static thread_local Entry cache[4] = {{-1, 0, 0, 0, nullptr}, {-1, 0, 0, 0, nullptr}, {-1, 0, 0, 0, nullptr}, {-1, 0, 0, 0, nullptr}};
    static thread_local int next_entry = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 4096;
    int per_cu = 0, cus = 0;
    for (const Entry &e : cache)
        if (e.device == dev && e.lds == lds_bytes && e.kernel == reinterpret_cast<const void *>(kernel)) {
            per_cu = e.per_cu;
            cus = e.cus;
        }
    if (per_cu == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, TBLOCK, lds_bytes) != hipSuccess || per_cu < 1)
            per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        // The occupancy call answers 5 blocks for 31.5 KB of LDS per block (5 x 31.5 KB = 157.5 KB of the CU's 160 KB),
        // but only 4 become resident (measured: the fifth block of every CU then runs after the others, and the launch
        // takes 40 % longer than with the 4 that fit). Count LDS in whole KB against 158 KB.
        if (lds_bytes != 0) per_cu = std::min<int>(per_cu, std::max<int>(1, (int)((158u * 1024u) / ((lds_bytes + 1023u) / 1024u * 1024u))));
        cache[next_entry] = {dev, lds_bytes, per_cu, cus, reinterpret_cast<const void *>(kernel)};
        next_entry = (next_entry + 1) % 4;
    }
    if (blocks_per_cu_limit != 0) per_cu = std::min<int>(per_cu, (int)blocks_per_cu_limit);
    uint64_t waves = (uint64_t)cus * per_cu * (TBLOCK / 64);
    if (const char *e = getenv("RT_TUNE_MAX_WAVES")) // experiment hook
        if (atol(e) > 0) waves = (uint64_t)atol(e);
    return waves;
}
// Rays per range: about 32 ranges per resident wavefront, so that a static, strided
// assignment of ranges evens out cheap and expensive regions of the batch.
inline uint32_t rays_per_wave_for(uint64_t n, uint64_t waves) {
    static const uint64_t ranges_per_wave = [] {
        const char *e = getenv("RT_TUNE_RANGES"); // experiment hook
        return e && atol(e) > 0 ? (uint64_t)atol(e) : (uint64_t)32;
    }();
    uint64_t target = waves * ranges_per_wave;
    uint64_t r = (n + target - 1) / target;
    r = (r + 63) / 64 * 64;
    if (r < 128) r = 128;
    if (r > 4096) r = 4096;
    return (uint32_t)r;
}
inline unsigned blocks_for_waves(uint64_t n, uint32_t rpw, uint64_t max_waves) {
    uint64_t waves = (n + rpw - 1) / rpw;
    if (waves > max_waves) waves = max_waves;
    uint64_t blocks = (waves + (TBLOCK / 64) - 1) / (TBLOCK / 64);
    return (unsigned)(blocks ? blocks : 1);
}

// blocks of 1024 threads need more than the default 64 KB of dynamic LDS
template <class K>
inline void allow_large_lds(K kernel) {
#if defined(RT_TBLOCK)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#else
    (void)kernel;
#endif
}

} // namespace
} // namespace rtd
