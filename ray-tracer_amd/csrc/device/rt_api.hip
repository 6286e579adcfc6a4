// rt_api.hip — the C ABI of rt_amd.h on top of the gfx950 kernels.
// There is no CPU fallback anywhere in this file: without a HIP device every
// compute entry point fails with RT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../host/rt_error.hpp"
#include "rt_amd.h"
#include "rt_device.h"
#include "rt_fast_tree.hpp"

namespace rt {
namespace {
thread_local std::string g_last_error;
}
rt_status fail(rt_status st, const std::string &msg) {
    g_last_error = msg;
    return st;
}
const char *last_error_cstr() { return g_last_error.c_str(); }
} // namespace rt

using rt::fail;

namespace {

struct HipFail {
    rt_status st;
    std::string msg;
};

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            rt_status s_ = (e_ == hipErrorOutOfMemory) ? RT_ERR_OUT_OF_MEMORY                               \
                           : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? RT_ERR_NO_DEVICE      \
                                                                                     : RT_ERR_DEVICE;        \
            throw HipFail{s_, std::string(#expr) + ": " + hipGetErrorString(e_)};                            \
        }                                                                                                    \
    } while (0)

template <class F> rt_status guarded(F &&f) {
    try {
        return f();
    } catch (const HipFail &h) {
        return fail(h.st, h.msg);
    } catch (const rt::DeviceFailure &f) {
        return fail(f.status, f.what());
    } catch (const std::bad_alloc &) {
        return fail(RT_ERR_OUT_OF_MEMORY, "host allocation failed");
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID_ARGUMENT, e.what());
    }
}

// RAII device buffer
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 256;
        HIP_TRY(hipMalloc(&p, n));
        bytes = n;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        HIP_TRY(hipGetDevice(&prev));
        if (prev != dev) HIP_TRY(hipSetDevice(dev));
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    EventPair() {
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
    }
    ~EventPair() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
    float ms() {
        float t = 0.f;
        HIP_TRY(hipEventSynchronize(b));
        HIP_TRY(hipEventElapsedTime(&t, a, b));
        return t;
    }
};

int usable_devices() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

} // namespace

// ============================================================================
// scene
// ============================================================================
struct rt_scene {
    int device = 0;
    rtd::Scene view{};
    DevBuf nodes, qnodes, qnodes_c, pre, pre2, hot, cold, pts2d, mats, ref_of_fast, fast_of_ref;
    // One counter per fast-walk launch: the launch's wavefronts draw their ranges of rays from it (and one more word
    // behind it, which a batch launch sets when it marks a ray for the re-walk). Live
    // launches take theirs from a ring of TICKETS counters, so that launches on several streams never share
    // one: a counter comes round again after TICKETS launches on this scene. A launch recorded into a HIP
    // graph keeps its counter for as long as the graph lives (the counter's address and the memset in front
    // of the kernel are baked into the graph, and every replay uses them again), so captured launches take
    // theirs from a range of CAPTURED_TICKETS counters that are handed out once and never again: a replay
    // can then overlap live launches, and other graphs, on other streams without sharing a counter.
    static constexpr uint32_t TICKETS = 4096, CAPTURED_TICKETS = 4096;
    DevBuf tickets;
    std::atomic<uint32_t> ticket_turn{0}, captured_turn{0};
    unsigned long long *next_ticket(hipStream_t stream) {
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(stream, &capture) != hipSuccess) {
            (void)hipGetLastError();
            capture = hipStreamCaptureStatusNone;
        }
        if (capture != hipStreamCaptureStatusNone) {
            const uint32_t k = captured_turn.fetch_add(1, std::memory_order_relaxed);
            if (k >= CAPTURED_TICKETS)
                throw HipFail{RT_ERR_UNSUPPORTED, "more than 4,096 fast-walk launches have been recorded into HIP graphs on "
                                                  "this scene: create another scene handle for further captures"};
            return tickets.as<unsigned long long>() + 2 * (size_t)(TICKETS + k);
        }
        return tickets.as<unsigned long long>() + 2 * (size_t)(ticket_turn.fetch_add(1, std::memory_order_relaxed) % TICKETS);
    }
#if defined(RT_COUNT_VISITS) || defined(RT_TIME_ROUNDS)
    DevBuf visits;
#endif
    std::vector<uint32_t> slot_of_prim; // prim_id -> slot (host side, for rt_intersect_primitives)
    rt_scene_info info{};
    unsigned char *static_scratch = nullptr; // this device's copy of g_static_scratch (rt_trace_paths_static)
};

// ============================================================================
// scratch of the entry point that allocates nothing (rt_trace_paths_static)
// ============================================================================
// One copy of this array exists on every GPU the library runs on, from the moment the code
// object is loaded there until the process ends. In host code the array's NAME is only a
// stand-in: the memory is reached through the per-device address that hipGetSymbolAddress
// returns (static_scratch_of), and the kernels get that address as an ordinary argument.
static __device__ __align__(256) unsigned char g_static_scratch[RT_STATIC_SCRATCH_BYTES];

namespace {

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[MAX_DEVICES] = {};

// Address of g_static_scratch on `device`, which must be the current device. Looked up once per
// device (at scene creation, so that the entry point itself only reads the cached value).
unsigned char *static_scratch_of(int device) {
    if (device < 0 || device >= MAX_DEVICES) throw HipFail{RT_ERR_UNSUPPORTED, "device index beyond 63"};
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch)));
        if (!p) throw HipFail{RT_ERR_DEVICE, "hipGetSymbolAddress returned NULL for the static scratch"};
        g_scratch_ptr[device] = static_cast<unsigned char *>(p);
    }
    return g_scratch_ptr[device];
}

} // namespace

namespace {

constexpr uint32_t FAST_STACK_LIMIT = 40;   // binary build trees deeper than this are not accepted
// Most entries per lane a fast-walk block can hold: entries x 256 lanes x 6 bytes + 6 KB of hit positions must
// fit the 64 KB of dynamic LDS a block may ask for: 38 entries = 64,512 bytes.
constexpr uint32_t FAST_STACK_MAX = (65536u - 3u * 8u * 256u) / (256u * 6u);
static_assert(FAST_STACK_MAX == 38, "stack limit of the fast walk");
constexpr uint32_t PIXEL_BLOCK = 1; // side of the pixel blocks a film lists its pixels by
constexpr uint32_t FAST_STACK_ENTRIES = 16; // x 256 lanes x 6 bytes + 6 KB of hit positions = 30 KB of LDS per block: five blocks per CU (31.5 KB no longer fit five)
using Box = rtd::BoxD;
using rtd::TOP_LEVELS;
Box empty_box() {
    Box b;
    for (int k = 0; k < 3; k++) {
        b.lo[k] = INFINITY;
        b.hi[k] = -INFINITY;
    }
    return b;
}

// (bounding_center - p0) . normal: zero for a planar polygon
double polygon_plane_offset(const rt_polygon &p) {
    return (p.bounding_center[0] - p.p0[0]) * p.normal[0] + (p.bounding_center[1] - p.p0[1]) * p.normal[1] +
           (p.bounding_center[2] - p.p0[2]) * p.normal[2];
}

// f64 bounds of everything the primitive in reference slot s can report as a hit
// `circle`: RT_FAST_BOUNDS_CIRCLE, i.e. everything primitives.rs:231 lets through: the hit point lies within
// bounding_radius of bounding_center (on the plane through the centre with the polygon's normal when the polygon
// is planar; anywhere in that ball when it is warped, see below).
Box slot_bounds(const rt_scene_desc &d, uint32_t s, bool circle = false) {
    Box bx = empty_box();
    uint32_t v = d.slots[s], idx = v & ~RT_SLOT_SPHERE;
    if (circle && !(v & RT_SLOT_SPHERE)) {
        const rt_polygon &p = d.polygons[idx];
        const double r = std::fabs(p.bounding_radius) * (1.0 + 1e-12);
        const double nn = p.normal[0] * p.normal[0] + p.normal[1] * p.normal[1] + p.normal[2] * p.normal[2];
        const bool planar = std::fabs(polygon_plane_offset(p)) <= r * 1e-9 && nn > 0.0 && std::isfinite(nn);
        for (int a = 0; a < 3; a++) {
            // half extent of a disc of radius r with unit normal n along axis a: r sqrt(1 - n_a^2)
            double h = r;
            if (planar) h = r * std::sqrt(std::max(0.0, 1.0 - p.normal[a] * p.normal[a] / nn)) + r * 1e-6;
            bx.lo[a] = p.bounding_center[a] - h;
            bx.hi[a] = p.bounding_center[a] + h;
        }
        return bx;
    }
    if (v & RT_SLOT_SPHERE) {
        const rt_sphere &sp = d.spheres[idx];
        for (int k = 0; k < 3; k++) {
            bx.lo[k] = sp.center[k] - std::fabs(sp.radius);
            bx.hi[k] = sp.center[k] + std::fabs(sp.radius);
        }
    } else {
        // The reference puts a hit on the plane through bounding_center (primitives.rs:221-228)
        // and accepts it when its (basis_1, basis_2) coordinates relative to p0 lie inside the
        // outline (primitives.rs:236-243). For a planar polygon that is the polygon itself. For a
        // warped one the centre plane lies off = (c - p0).n above the outline's plane, and an
        // origin between the two planes is mirrored by the |.| of primitives.rs:223 to as much as
        // 2|off| below the centre plane: every possible hit lies in the prism spanned by the
        // outline at heights off - 2|off| and off over the p0 plane.
        const rt_polygon &p = d.polygons[idx];
        const double *q = d.points2d + 2 * (size_t)p.points_offset;
        const double off = polygon_plane_offset(p);
        const double heights[2] = {off - 2.0 * std::fabs(off), off};
        for (uint32_t k = 0; k < p.n_points; k++)
            for (double hgt : heights)
                for (int a = 0; a < 3; a++) {
                    double vtx = p.p0[a] + q[2 * k] * p.basis_1[a] + q[2 * k + 1] * p.basis_2[a] + hgt * p.normal[a];
                    bx.lo[a] = std::min(bx.lo[a], vtx);
                    bx.hi[a] = std::max(bx.hi[a], vtx);
                }
    }
    return bx;
}

// The f32 screen's view of the primitive in reference slot s (see PrimPre).
// `plane_only` (RT_FAST_BOUNDS_CIRCLE): no edge half-spaces, the screen only asks which way the plane faces.
void fill_screen_record(const rt_scene_desc &d, uint32_t s, const rtd::PrimHot &h, double extent, rtd::PrimPre &q,
                        bool plane_only = false) {
    std::memset(&q, 0, sizeof q);
    uint32_t v = d.slots[s], idx = v & ~RT_SLOT_SPHERE;
    if (v & RT_SLOT_SPHERE) {
        q.e[0][0] = (float)h.cx; q.e[0][1] = (float)h.cy; q.e[0][2] = (float)h.cz;
        q.e[0][3] = std::nextafterf((float)std::fabs(h.radius), INFINITY);
        return;
    }
    // The screen places the plane point on the outline's own plane (through p0). A warped
    // polygon is hit on the plane through its bounding centre instead (slot_bounds): when the two
    // planes are farther apart than a small part of the screen's margins, the record is all NaN,
    // which every comparison of the screen answers with "maybe".
    if (!(std::fabs(polygon_plane_offset(d.polygons[idx])) <= extent * 1e-8)) {
        float *w = &q.nx;
        for (int k = 0; k < 16; k++) w[k] = NAN;
        return;
    }
    q.nx = (float)h.nx; q.ny = (float)h.ny; q.nz = (float)h.nz;
    q.k = (float)(h.nx * h.p0x + h.ny * h.p0y + h.nz * h.p0z);
    for (int k = 0; k < 3; k++) q.e[k][3] = INFINITY; // "always true" until an edge is filled in
    if (plane_only) return;
    const rt_polygon &p = d.polygons[idx];
    const double *pt = d.points2d + 2 * (size_t)p.points_offset;
    const uint32_t n = p.n_points;
    double area2 = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t j = (i + 1) % n;
        area2 += pt[2 * i] * pt[2 * j + 1] - pt[2 * j] * pt[2 * i + 1];
    }
    if (!(std::fabs(area2) > 0.0) || !std::isfinite(area2)) return;
    const double orient = area2 > 0.0 ? 1.0 : -1.0; // +1: counter-clockwise in the (basis_1, basis_2) frame
    struct Edge {
        double a, b, c, len;
    };
    std::vector<Edge> edges;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t j = (i + 1) % n;
        double dx = pt[2 * j] - pt[2 * i], dy = pt[2 * j + 1] - pt[2 * i + 1];
        double len = std::sqrt(dx * dx + dy * dy);
        if (!(len > 0.0)) return;
        double a = -dy / len * orient, b = dx / len * orient; // unit normal pointing inside
        double c = -(a * pt[2 * i] + b * pt[2 * i + 1]);
        // convex only if every vertex lies inside (or on) this edge's half-plane
        for (uint32_t m = 0; m < n; m++)
            if (a * pt[2 * m] + b * pt[2 * m + 1] + c < -1e-9 * len) return;
        edges.push_back({a, b, c, len});
    }
    std::sort(edges.begin(), edges.end(), [](const Edge &x, const Edge &y) { return x.len > y.len; });
    for (size_t k = 0; k < edges.size() && k < 3; k++) {
        double A[3];
        for (int c3 = 0; c3 < 3; c3++) A[c3] = edges[k].a * p.basis_1[c3] + edges[k].b * p.basis_2[c3];
        double C0 = edges[k].c - (A[0] * p.p0[0] + A[1] * p.p0[1] + A[2] * p.p0[2]);
        q.e[k][0] = (float)A[0]; q.e[k][1] = (float)A[1]; q.e[k][2] = (float)A[2];
        q.e[k][3] = (float)C0;
    }
}

double scene_extent(const rt_scene_desc &d) {
    const double *root = d.nodes[0].shell;
    double extent = std::max({std::fabs(root[0]), std::fabs(root[1]), std::fabs(root[2])}) + 4.0 * root[3];
    if (!(extent > 0.0) || !std::isfinite(extent)) extent = 1.0;
    return extent;
}

// Bound on every coordinate the f32 screen computes with. The screen only runs for rays whose origin was not moved
// (make_fray: dot(centre - o, dir) <= 2 R for the root shell (centre, R)), and it only matters for rays that hit
// something: their line passes within R of the centre and the hit lies at most R behind or beyond it along the ray,
// so the origin is within sqrt(5) R of the centre, the hit within R, and the distance between them is at most 3 R.
double screen_extent(const rt_scene_desc &d) {
    const double *root = d.nodes[0].shell;
    double extent = std::max({std::fabs(root[0]), std::fabs(root[1]), std::fabs(root[2])}) + 2.25 * root[3];
    if (!(extent > 0.0) || !std::isfinite(extent)) extent = 1.0;
    return extent;
}

// How far every primitive's box is grown before the hierarchy is built over the boxes, so that the f32 slab test
// (node_hits) never misses a box whose contents the f64 ray touches. In units of u E (u = 2^-24, E = screen_extent:
// origins of rays that hit anything lie within sqrt(5) R of the root shell's centre, also after make_fray moved them,
// box planes within R, hits at most 3 R along the ray), as a displacement of the plane along its axis:
//   the f32 ray against the f64 ray at the hit's parameter: origin 1, direction 1.33
//   inv = 1 / dir good to one unit in the last place: 2 |plane - o| = 2.88
//   scale * inv 0.89 (node extent 2 R), o * inv 1, org * inv - o * inv 1.44, byte * a + b 1.44
// = 10 u E; the pad is 1.4 times that. Growing a box by the displacement makes up for it whatever the direction: a
// plane moved by p is met p / |dir| earlier or later, which is what the displacement does to the computed distance.
// (The factor 1.000002 on the far distances and on the prune distance covers the f32 direction's length.)
// Scenes of small primitives feel the pad: with 2^-19 of max|centre| + 4 R (55 u E) the triangles of the 11.9 M
// primitive scene, 6e-4 across, carried 3.5e-5 of it on every side: 11.1 instead of 7 screened primitives per ray.
double box_pad(const rt_scene_desc &d) {
    double units = 14.0;
    if (const char *tune = std::getenv("RT_TUNE_PAD")) units = std::max(0.0, std::atof(tune)); // experiments
    return screen_extent(d) * units * 0x1p-24;
}

// Structural checks so that no kernel can index outside its arrays.
std::string validate(const rt_scene_desc &d, uint32_t &depth_out, uint32_t &leaves_out, uint32_t &max_leaf_out) {
    if (d.struct_size != sizeof(rt_scene_desc)) return "rt_scene_desc.struct_size mismatch";
    if ((d.fast_tree & RT_FAST_TREE_KIND_MASK) > RT_FAST_TREE_LBVH || (d.fast_tree & ~(RT_FAST_TREE_KIND_MASK | RT_FAST_BOUNDS_CIRCLE)))
        return "unknown rt_scene_desc.fast_tree";
    if (d.n_nodes == 0 || !d.nodes) return "scene has no nodes";
    if (d.n_slots != d.n_polygons + d.n_spheres) return "n_slots != n_polygons + n_spheres";
    if (d.n_slots >= (1u << 28)) return "more than 2^28-1 primitives";
    if (d.n_slots && !d.slots) return "slots is NULL";
    if (d.n_polygons && (!d.polygons || !d.points2d)) return "polygons/points2d is NULL";
    if (d.n_spheres && !d.spheres) return "spheres is NULL";
    if (d.n_slots && (d.n_materials == 0 || !d.materials)) return "scene has primitives but no materials";
    for (uint32_t i = 0; i < d.n_polygons; i++) {
        const rt_polygon &p = d.polygons[i];
        if (p.n_points < 3) return "polygon with fewer than 3 points";
        if ((uint64_t)p.points_offset + p.n_points > d.n_points2d) return "polygon points out of range";
        if (p.material >= d.n_materials) return "polygon material out of range";
    }
    for (uint32_t i = 0; i < d.n_spheres; i++)
        if (d.spheres[i].material >= d.n_materials) return "sphere material out of range";
    {   // prim_id is the index in the list given to BVH::from_list: a permutation of 0 .. n_slots-1
        std::vector<uint8_t> seen_id(d.n_slots, 0);
        auto take = [&](uint32_t id) {
            if (id >= d.n_slots || seen_id[id]) return false;
            seen_id[id] = 1;
            return true;
        };
        for (uint32_t i = 0; i < d.n_polygons; i++)
            if (!take(d.polygons[i].prim_id)) return "prim_id out of range or used twice";
        for (uint32_t i = 0; i < d.n_spheres; i++)
            if (!take(d.spheres[i].prim_id)) return "prim_id out of range or used twice";
    }
    for (uint32_t i = 0; i < d.n_materials; i++)
        if (!(d.materials[i].gloss >= 0.0 && d.materials[i].gloss <= 1.0)) return "material gloss outside [0, 1]";
    std::vector<uint8_t> seen_poly(d.n_polygons, 0), seen_sph(d.n_spheres, 0);
    for (uint32_t s = 0; s < d.n_slots; s++) {
        uint32_t v = d.slots[s], idx = v & ~RT_SLOT_SPHERE;
        if (v & RT_SLOT_SPHERE) {
            if (idx >= d.n_spheres || seen_sph[idx]) return "slot references a sphere out of range or twice";
            seen_sph[idx] = 1;
        } else {
            if (idx >= d.n_polygons || seen_poly[idx]) return "slot references a polygon out of range or twice";
            seen_poly[idx] = 1;
        }
    }
    // pre-order walk with an explicit stack of subtree ends
    uint32_t next_slot = 0, depth = 0, max_depth = 0, leaves = 0, max_leaf = 0;
    std::vector<uint32_t> ends;
    for (uint32_t i = 0; i < d.n_nodes; i++) {
        while (!ends.empty() && ends.back() == i) ends.pop_back();
        const rt_node &n = d.nodes[i];
        uint32_t limit = ends.empty() ? d.n_nodes : ends.back();
        if (n.skip <= i || n.skip > limit) return "node skip link out of range";
        depth = (uint32_t)ends.size() + 1;
        max_depth = std::max(max_depth, depth);
        if (n.left == 0) {
            if (n.skip != i + 1) return "leaf skip link must be the next node";
            if (n.slot_count == 0 && d.n_slots != 0) return "empty leaf in a scene with primitives";
            if (n.slot_begin != next_slot || (uint64_t)n.slot_begin + n.slot_count > d.n_slots)
                return "leaf slot range out of order or out of range";
            next_slot += n.slot_count;
            leaves++;
            max_leaf = std::max(max_leaf, n.slot_count);
        } else {
            if (n.slot_count != 0) return "inner node with primitives";
            if (i + 1 >= d.n_nodes || n.left <= i + 1 || n.left >= n.skip) return "inner node child links out of range";
            if (d.nodes[i + 1].skip != n.left) return "left child must follow the right subtree";
            if (d.nodes[n.left].skip != n.skip) return "left subtree must end at the node's skip link";
            ends.push_back(n.skip);
        }
    }
    if (next_slot != d.n_slots) return "leaves do not cover every slot";
    if (d.nodes[0].skip != d.n_nodes) return "root skip link must be n_nodes";
    depth_out = max_depth;
    leaves_out = leaves;
    max_leaf_out = max_leaf;
    return std::string();
}

} // namespace

extern "C" {

const char *rt_last_error(void) { return rt::last_error_cstr(); }
uint32_t rt_abi_version(void) { return 3; }

rt_status rt_device_count(int *count) {
    if (!count) return fail(RT_ERR_INVALID_ARGUMENT, "rt_device_count: count is NULL");
    *count = usable_devices();
    if (*count == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device is visible; this library has no CPU fallback");
    return RT_OK;
}

rt_status rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out) {
    if (!desc || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: NULL argument");
    *out = nullptr;
    return guarded([&]() -> rt_status {
        const rt_scene_desc &d = *desc;
        uint32_t depth = 0, leaves = 0, max_leaf = 0;
        std::string why = validate(d, depth, leaves, max_leaf);
        if (!why.empty()) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: " + why);
        int ndev = usable_devices();
        if (ndev == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device is visible; this library has no CPU fallback");
        if (device < 0 || device >= ndev) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: device out of range");

        const double extent = scene_extent(d);
        bool circle_bounds = (d.fast_tree & RT_FAST_BOUNDS_CIRCLE) != 0u;
        if (const char *e = std::getenv("RT_FAST_BOUNDS")) circle_bounds = std::strcmp(e, "circle") == 0; // A/B measurements
        // repack primitives by slot and compute their boxes
        std::vector<rtd::PrimHot> hot(d.n_slots);
        std::vector<rtd::PrimCold> cold(d.n_slots);
        std::vector<rtd::PrimPre> pre(d.n_slots);
        std::vector<Box> slot_box(d.n_slots);
        uint32_t max_prim_id = 0;
        for (uint32_t s = 0; s < d.n_slots; s++) {
            rtd::PrimHot &h = hot[s];
            rtd::PrimCold &c = cold[s];
            std::memset(&h, 0, sizeof h);
            std::memset(&c, 0, sizeof c);
            Box &bx = slot_box[s];
            uint32_t v = d.slots[s], idx = v & ~RT_SLOT_SPHERE;
            if (v & RT_SLOT_SPHERE) {
                const rt_sphere &sp = d.spheres[idx];
                h.radius = sp.radius;
                h.cx = sp.center[0]; h.cy = sp.center[1]; h.cz = sp.center[2];
                h.n_points = 0;
                h.material = sp.material;
                h.prim_id = sp.prim_id;
            } else {
                const rt_polygon &p = d.polygons[idx];
                h.nx = p.normal[0]; h.ny = p.normal[1]; h.nz = p.normal[2];
                h.radius = p.bounding_radius;
                h.p0x = p.p0[0]; h.p0y = p.p0[1]; h.p0z = p.p0[2];
                h.n_points = p.n_points;
                h.points_offset = p.points_offset;
                h.cx = p.bounding_center[0]; h.cy = p.bounding_center[1]; h.cz = p.bounding_center[2];
                h.material = p.material;
                h.prim_id = p.prim_id;
                std::memcpy(c.b1, p.basis_1, 24);
                std::memcpy(c.b2, p.basis_2, 24);
                const double *q = d.points2d + 2 * (size_t)p.points_offset;
                for (uint32_t k = 0; k < p.n_points && k < 4; k++) {
                    c.pts[2 * k] = q[2 * k];
                    c.pts[2 * k + 1] = q[2 * k + 1];
                }
            }
            bx = slot_bounds(d, s, circle_bounds);
            max_prim_id = std::max(max_prim_id, h.prim_id);
            fill_screen_record(d, s, h, extent, pre[s], circle_bounds);
        }

        auto sc = std::make_unique<rt_scene>();
        sc->device = device;
        // f32 boxes must cover the f64 geometry despite rounding of the ray and of the slab arithmetic (box_pad)
        const double *root = d.nodes[0].shell;
        // The fast walk's own hierarchy (rt_scene_desc.fast_tree; the environment variable
        // RT_FAST_TREE=sah|reference|lbvh overrides it for A/B measurements).
        const double pad = box_pad(d);
        uint32_t kind = d.fast_tree & RT_FAST_TREE_KIND_MASK;
        if (const char *e = std::getenv("RT_FAST_TREE")) {
            if (std::strcmp(e, "reference") == 0) kind = RT_FAST_TREE_REFERENCE;
            else if (std::strcmp(e, "lbvh") == 0) kind = RT_FAST_TREE_LBVH;
            else if (std::strcmp(e, "sah") == 0) kind = RT_FAST_TREE_SAH;
        }
        unsigned threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        rtd::FastTree fb;
        if (kind == RT_FAST_TREE_LBVH) {
            fb = rtd::build_fast_tree_lbvh(slot_box, pad, device, FAST_STACK_LIMIT - 5);
            if (fb.max_depth > FAST_STACK_LIMIT) kind = RT_FAST_TREE_SAH; // deeper than the walk's stack
        }
        if (kind == RT_FAST_TREE_REFERENCE) fb = rtd::build_fast_tree_reference(d, slot_box, pad);
        else if (kind == RT_FAST_TREE_SAH) fb = rtd::build_fast_tree_sah(slot_box, pad, TOP_LEVELS, threads);
        const bool reference_topology = kind == RT_FAST_TREE_REFERENCE;
        std::vector<Box>().swap(slot_box);
        if (fb.max_depth > 2 * rtd::WREF_NEED_MAX)
            return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: this hierarchy is deeper than 62 levels; use "
                                            "RT_FAST_TREE_SAH");
        rtd::WideTree wide;
        try {
            uint32_t depth_wanted = FAST_STACK_ENTRIES - 1u;
            if (const char *tune = std::getenv("RT_TUNE_WIDE_DEPTH")) depth_wanted = (uint32_t)std::atoi(tune); // experiments
            wide = rtd::collapse_to_wide(fb, depth_wanted);
        } catch (const std::length_error &e) {
            return fail(RT_ERR_UNSUPPORTED, std::string("rt_scene_create: ") + e.what());
        } catch (const std::logic_error &e) { // a builder broke its contract
            return fail(RT_ERR_INTERNAL, std::string("rt_scene_create: ") + e.what());
        }
        std::vector<rtd::FNode>().swap(fb.fnodes); // the device walks the wide nodes only
        std::vector<rtd::QNode> qnodes = rtd::quantise_wide(wide);
        std::vector<rtd::WNode>().swap(wide.nodes);
        const uint32_t root_ref = wide.root_ref;
        // primitive records in fast-slot order, and the two slot maps
        std::vector<uint32_t> fast_of_ref(d.n_slots);
        for (uint32_t f = 0; f < d.n_slots; f++) fast_of_ref[fb.ref_of_fast[f]] = f;
        if (!reference_topology) {
            auto permute = [&](auto &v) {
                typename std::remove_reference<decltype(v)>::type out(v.size());
                for (uint32_t f = 0; f < d.n_slots; f++) out[f] = v[fb.ref_of_fast[f]];
                v.swap(out);
            };
            permute(hot);
            permute(cold);
            permute(pre);
        }
        if (d.n_slots) {
            sc->slot_of_prim.assign((size_t)max_prim_id + 1, RT_MISS);
            for (uint32_t f = 0; f < d.n_slots; f++) sc->slot_of_prim[hot[f].prim_id] = f;
        }

        DeviceGuard g(device);
        sc->static_scratch = static_scratch_of(device);
        auto upload = [](DevBuf &b, const void *src, size_t bytes) {
            b.alloc(bytes);
            if (bytes) HIP_TRY(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        };
        {   // the device copy of the reference tree also says where each subtree's primitives begin:
            // reserved[0] = first reference slot below the node (nodes are in visiting order, so this is
            // the number of primitives in the leaves stored before it)
            std::vector<rt_node> nodes(d.nodes, d.nodes + d.n_nodes);
            uint32_t next_slot = 0;
            for (uint32_t i = 0; i < d.n_nodes; i++) {
                nodes[i].reserved[0] = next_slot;
                if (nodes[i].left == 0) next_slot += nodes[i].slot_count;
            }
            upload(sc->nodes, nodes.data(), sizeof(rt_node) * d.n_nodes);
        }
        upload(sc->qnodes, qnodes.data(), sizeof(rtd::QNode) * qnodes.size());
        // The two-rays-per-lane walk keeps a stack entry in one word, which takes child references of 20 bits:
        // leaf = 1 | count-1 (3) | slot (16), inner = 0 | 0 | need (4) | index (14), none = 0xFFFFF. A scene
        // with fewer than 65,528 primitives, at most 16,384 wide nodes and at most 15 levels gets a second copy
        // of its nodes with such references (RT_TWO_RAYS=0 keeps the one-ray walk for A/B measurements).
        // Measured on the hippo scene (MI355X): the 16.7 M-ray batch gains 3-6 % with two rays per lane, the
        // path tracer's launches lose 1.5 %, so by default only the batch entry points use it
        // (RT_TWO_RAYS = 0 none, 1 batch, 2 paths, 3 both).
        uint32_t two_rays_mask = 1u;
        if (const char *e = std::getenv("RT_TWO_RAYS")) two_rays_mask = (uint32_t)std::atoi(e) & 3u;
        uint32_t walk3_mask = 0u; // parked-ray walk (rt_walk3_device.hpp): RT_WALK3 = 0 none, 1 batch, 2 paths, 3 both
        if (const char *e = std::getenv("RT_WALK3")) walk3_mask = (uint32_t)std::atoi(e) & 3u;
        const bool two_rays = (two_rays_mask | walk3_mask) != 0u && d.n_slots < 65528u && qnodes.size() <= 16384u && wide.depth <= 15u;
        uint32_t root_ref_c = 0xFFFFFu;
        if (two_rays) {
            auto compact = [](uint32_t r) -> uint32_t {
                if (r == rtd::REF_NONE) return 0xFFFFFu;
                if (r & rtd::REF_LEAF) return (1u << 19) | (((r >> rtd::REF_CNT_SHIFT) & 7u) << 16) | (r & 0xFFFFu);
                return (((r >> rtd::WREF_NEED_SHIFT) & 15u) << 14) | (r & 0x3FFFu);
            };
            std::vector<rtd::QNode> qc(qnodes);
            for (auto &q : qc)
                for (int k = 0; k < 4; k++) q.ref[k] = compact(q.ref[k]);
            root_ref_c = compact(root_ref);
            upload(sc->qnodes_c, qc.data(), sizeof(rtd::QNode) * qc.size());
        }
        upload(sc->pre, pre.data(), sizeof(rtd::PrimPre) * pre.size());
        {
            std::vector<rtd::PrimPre2> pairs(pre.size());
            for (size_t i = 0; i < pre.size(); i++) {
                const float *a0 = &pre[i].nx, *a1 = &pre[i + 1 < pre.size() ? i + 1 : i].nx;
                for (int j = 0; j < 16; j++) {
                    pairs[i].f[2 * j] = a0[j];
                    pairs[i].f[2 * j + 1] = a1[j];
                }
            }
            upload(sc->pre2, pairs.data(), sizeof(rtd::PrimPre2) * pairs.size());
        }
        upload(sc->hot, hot.data(), sizeof(rtd::PrimHot) * hot.size());
        upload(sc->cold, cold.data(), sizeof(rtd::PrimCold) * cold.size());
        upload(sc->ref_of_fast, fb.ref_of_fast.data(), sizeof(uint32_t) * fb.ref_of_fast.size());
        upload(sc->fast_of_ref, fast_of_ref.data(), sizeof(uint32_t) * fast_of_ref.size());
        upload(sc->pts2d, d.points2d, sizeof(double) * 2 * d.n_points2d);
        upload(sc->mats, d.materials, sizeof(rt_material) * d.n_materials);
        sc->tickets.alloc((rt_scene::TICKETS + rt_scene::CAPTURED_TICKETS) * 2 * sizeof(unsigned long long)); // two words per launch

        rtd::Scene &v = sc->view;
        v.nodes = sc->nodes.as<rt_node>();
        v.qnodes = sc->qnodes.as<rtd::QNode>();
        v.pre = sc->pre.as<rtd::PrimPre>();
        v.pre2 = sc->pre2.as<rtd::PrimPre2>();
        v.hot = sc->hot.as<rtd::PrimHot>();
        v.cold = sc->cold.as<rtd::PrimCold>();
        v.ref_of_fast = sc->ref_of_fast.as<uint32_t>();
        v.fast_of_ref = sc->fast_of_ref.as<uint32_t>();
        v.has_spheres = d.n_spheres != 0;
        v.has_gloss = 0;
        for (uint32_t m = 0; m < d.n_materials; m++)
            if (d.materials[m].gloss > 0.0) v.has_gloss = 1;
        v.pts2d = sc->pts2d.as<double>();
        v.mats = sc->mats.as<rt_material>();
        v.n_nodes = d.n_nodes;
        v.n_slots = d.n_slots;
        v.n_inner = (uint32_t)qnodes.size();
        // Entries per lane of the walk's stack: one per wide level is enough for correctness (group
        // entries), three per level are never exceeded, and up to FAST_STACK_ENTRIES cost no occupancy.
        v.stack_depth = std::max(wide.depth + 1u, std::min(3u * std::max(1u, wide.depth), FAST_STACK_ENTRIES));
        if (const char *tune = std::getenv("RT_TUNE_STACK")) { // experiments: never below one entry per level
            unsigned e = (unsigned)std::atoi(tune);
            if (e >= wide.depth + 1u && e <= FAST_STACK_MAX) v.stack_depth = e;
        }
        v.root_ref = root_ref;
        v.qnodes_c = two_rays ? sc->qnodes_c.as<rtd::QNode>() : nullptr;
        v.root_ref_c = root_ref_c;
        v.two_rays = two_rays ? two_rays_mask : 0u;
        // lanes (of 64) that must have a ray of the kind among their two before the wavefront runs that kind
        // (hippo, exact / leaf / refill: batch 16,24,32 -> 5.17, 24,24,32 -> 5.24, 32,32,24 -> 5.10, 24,32,40 -> 4.86,
        // 32,32,48 -> 3.81 Grays/s; render 20,28,40 -> 7.09, 32,32,40 -> 7.06, 24,24,32 -> 7.01, 24,32,56 -> 6.41 G queries/s)
        v.vote2_exact = 24; v.vote2_leaf = 24; v.vote2_refill = 32;
        v.vote2_exact_paths = 20; v.vote2_leaf_paths = 28; v.vote2_refill_paths = 40;
        if (const char *tune = std::getenv("RT_TUNE_VOTES2")) { // "exact,leaf,refill" for experiments
            unsigned e = 0, l = 0, r = 0;
            if (std::sscanf(tune, "%u,%u,%u", &e, &l, &r) == 3 && e >= 1 && e <= 64 && l >= 1 && l <= 64 && r >= 1 && r <= 64) {
                v.vote2_exact = e; v.vote2_leaf = l; v.vote2_refill = r;
            }
        }
        // parked-ray walk: RT_WALK3 = 0 none, 1 batch, 2 paths, 3 both; thresholds "switch,prep,stuck"
        v.walk3 = (two_rays && wide.depth <= 14u) ? walk3_mask : 0u; // 15 one-word stack entries fit beside the parked ray
        v.stack_depth3 = 15u;
        v.vote3_switch = 8; v.vote3_prep = 32; v.vote3_stuck = 8;
        v.vote3_switch_paths = 8; v.vote3_prep_paths = 32; v.vote3_stuck_paths = 8;
        for (int which = 0; which < 2; which++)
            if (const char *tune = std::getenv(which ? "RT_TUNE_VOTES3_PATHS" : "RT_TUNE_VOTES3")) {
                unsigned a = 0, b = 0, c = 0;
                if (std::sscanf(tune, "%u,%u,%u", &a, &b, &c) == 3 && a >= 1 && a <= 64 && b >= 1 && b <= 64 && c >= 1 && c <= 64) {
                    if (which) { v.vote3_switch_paths = a; v.vote3_prep_paths = b; v.vote3_stuck_paths = c; }
                    else { v.vote3_switch = a; v.vote3_prep = b; v.vote3_stuck = c; }
                }
            }
        if (const char *tune = std::getenv("RT_TUNE_VOTES2_PATHS")) {
            unsigned e = 0, l = 0, r = 0;
            if (std::sscanf(tune, "%u,%u,%u", &e, &l, &r) == 3 && e >= 1 && e <= 64 && l >= 1 && l <= 64 && r >= 1 && r <= 64) {
                v.vote2_exact_paths = e; v.vote2_leaf_paths = l; v.vote2_refill_paths = r;
            }
        }
        v.root_c[0] = root[0]; v.root_c[1] = root[1]; v.root_c[2] = root[2];
        v.root_r = root[3];
        v.t_abs_slack = (float)(extent * 0x1p-18);
        // f32 screen margins, each several times the worst rounding of its expression for
        // coordinates up to `extent` (f32 epsilon 2^-24 ~ 6e-8)
        v.pre_m_test = (float)(extent * 2e-6);
        v.pre_m_p = 2e-6f;
        {
            // The screen rejects a polygon when the plane point lies more than 2 (c2 / |n.dir| + c1) outside an edge.
            // Worst case of the f32 arithmetic for a true hit, with u = 2^-24, E = screen_extent (every coordinate
            // and the hit distance are below 1.34 E, every position's norm below sqrt(3) E), c = |n.dir| > 1e-5:
            //   (o - p0).n           15 u E   (rounded inputs 4.5, three fused steps 5.2, n.p0 1.7, the difference 3.5)
            //   n.dir                 5.7 u
            //   t = |..| / |n.dir|   (15 E + 8 t) 1.1 u / c + 3.5 u t  <=  28.3 u E / c + 4.7 u E
            //   q = o + t dir        that + 4.8 u E
            //   A.q + C              that + 18.6 u E (rounded A and C 5.6, three fused steps over sums of up to 4.4 E 13)
            //   plane through the bounding centre instead of p0 (fill_screen_record): 0.34 u E / c
            // = 28.7 u E / c + 28.1 u E = (1.71 / c + 1.68) 1e-6 E; the margin below is 1.4 times that. The sphere
            // screen's distance from the centre to the ray is off by at most 12 u E = 0.72e-6 E.
            double c1 = 1.2, c2 = 1.2;
            if (const char *tune = std::getenv("RT_TUNE_SCREEN")) { // "c1,c2" in 1e-6 of the extent, for experiments
                double a = 0, b = 0;
                if (std::sscanf(tune, "%lf,%lf", &a, &b) == 2 && a > 0 && b > 0) c1 = a, c2 = b;
            }
            const double se = screen_extent(d);
            v.pre_c1 = (float)(se * c1 * 1e-6);
            v.pre_c2 = (float)(se * c2 * 1e-6);
        }
#if defined(RT_COUNT_VISITS) || defined(RT_TIME_ROUNDS)
        sc->visits.alloc(160 * sizeof(unsigned long long));
        HIP_TRY(hipMemset(sc->visits.p, 0, 160 * sizeof(unsigned long long)));
        v.visits = sc->visits.as<unsigned long long>();
#endif
        v.vote_exact = 16; // measured on the 16.7 M ray batch: 16,20,16 -> 4.07, 16,16,12 -> 4.17, 12,12,12 -> 4.12, 24,28,32 -> 2.44 Grays/s
        v.vote_leaf = 16;
        v.vote_refill = 12;
        v.vote_exact_paths = 16;
        v.vote_leaf_paths = 20;
        v.vote_refill_paths = 32; // measured on the 1080p render: 16 -> 6.17, 24 -> 6.33, 28 -> 6.42, 32 -> 6.45, 36 -> 6.44, 40 -> 6.34 G queries/s
        // A scene whose tree and screen records do not fit the eight 4 MB L2 caches answers a node step from
        // HBM more often than not, and a lane that waits for a vote then waits for longer: such scenes render
        // faster with the batch's lower thresholds (11.9 M primitives, 1080p: 16,20,32 -> 2.28, 16,16,24 -> 2.44,
        // 16,16,12 -> 2.53, 12,12,12 -> 2.49 G queries/s).
        if (sc->qnodes.bytes + sc->pre2.bytes > (32u << 20)) {
            v.vote_leaf_paths = 16;
            v.vote_refill_paths = 12;
        }
        if (const char *tune = std::getenv("RT_TUNE_VOTES_PATHS")) { // the same for the path tracer's launches
            unsigned e = 0, l = 0, r = 0;
            if (std::sscanf(tune, "%u,%u,%u", &e, &l, &r) == 3 && e >= 1 && e <= 64 && l >= 1 && l <= 64 && r >= 1 &&
                r <= 64) {
                v.vote_exact_paths = e;
                v.vote_leaf_paths = l;
                v.vote_refill_paths = r;
            }
        }
        v.node_iters = 8; // re-measured with 4-wide nodes: 6/4 -> 8/6 is +1 % on the render, everything nearby within 1.5 %
        v.node_drop = 6;
        if (const char *tune = std::getenv("RT_TUNE_NODE")) { // "iterations,drop sixteenths" for experiments
            unsigned it = 0, dr = 0;
            if (std::sscanf(tune, "%u,%u", &it, &dr) == 2 && it >= 1 && it <= 64 && dr <= 16) {
                v.node_iters = it;
                v.node_drop = dr;
            }
        }
        if (const char *tune = std::getenv("RT_TUNE_VOTES")) { // "exact,leaf,refill" for experiments
            unsigned e = 0, l = 0, r = 0;
            if (std::sscanf(tune, "%u,%u,%u", &e, &l, &r) == 3 && e >= 1 && e <= 64 && l >= 1 && l <= 64 && r >= 1 &&
                r <= 64) {
                v.vote_exact = e;
                v.vote_leaf = l;
                v.vote_refill = r;
            }
        }

        rt_scene_info &inf = sc->info;
        inf.n_nodes = d.n_nodes;
        inf.n_leaves = leaves;
        inf.depth = depth;
        inf.n_prims = d.n_slots;
        inf.n_polygons = d.n_polygons;
        inf.n_spheres = d.n_spheres;
        inf.n_materials = d.n_materials;
        inf.max_leaf = max_leaf;
        inf.device = device;
        inf.fast_tree = kind | (circle_bounds ? RT_FAST_BOUNDS_CIRCLE : 0u);
        inf.fast_depth = wide.depth;
        inf.fast_nodes = (uint32_t)qnodes.size();
        inf.device_bytes = sc->nodes.bytes + sc->qnodes.bytes + (two_rays ? sc->qnodes_c.bytes : 0) + sc->pre.bytes + sc->pre2.bytes + sc->hot.bytes + sc->cold.bytes + sc->pts2d.bytes +
                           sc->mats.bytes + sc->ref_of_fast.bytes + sc->fast_of_ref.bytes;
        *out = sc.release();
        return RT_OK;
    });
}

rt_status rt_scene_destroy(rt_scene *scene) {
    if (!scene) return RT_OK;
    return guarded([&]() -> rt_status {
        {
            DeviceGuard g(scene->device);
#if defined(RT_COUNT_VISITS) || defined(RT_TIME_ROUNDS)
            unsigned long long c[160] = {0};
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(c, scene->visits.p, sizeof c, hipMemcpyDeviceToHost);
#endif
#ifdef RT_COUNT_VISITS
            std::fprintf(stderr, "{\"fast_walk_rays\": %llu, \"node_steps_per_ray\": %.3f, \"screened_per_ray\": %.3f, "
                         "\"exact_tests_per_ray\": %.3f}\n", c[0], c[1] / (double)std::max(c[0], 1ull),
                         c[2] / (double)std::max(c[0], 1ull), c[3] / (double)std::max(c[0], 1ull));
            if (c[92] | c[93] | c[94]) {
                auto as_float = [](unsigned long long v) { uint32_t u = (uint32_t)v; float f; std::memcpy(&f, &u, 4); return f; };
                std::fprintf(stderr, "  largest share of the f32 screen's margins used by a hit of the reference-order walk: edges %.3f, "
                             "(o - p0).n %.3f, n.dir %.3f (1 = the screen would have lost the hit)\n", as_float(c[92]), as_float(c[93]), as_float(c[94]));
            }
            if (c[90])
                std::fprintf(stderr, "  exact tests that hit %.3f per ray, that became the closest hit %.3f per ray\n",
                             c[90] / (double)std::max(c[0], 1ull), c[91] / (double)std::max(c[0], 1ull));
            const char *kinds[4] = {"exact", "leaf", "node", "refill"};
            for (int k = 0; k < 4; k++)
                std::fprintf(stderr, "  %-6s rounds %12llu  lanes per round %6.2f  rounds per 64 rays %8.2f\n", kinds[k],
                             c[4 + 2 * k], c[5 + 2 * k] / (double)std::max(c[4 + 2 * k], 1ull),
                             64.0 * c[4 + 2 * k] / (double)std::max(c[0], 1ull));
            if (c[96])
                std::fprintf(stderr, "  switch rounds %12llu  lanes per round %6.2f  rounds per 64 rays %8.2f\n", c[96],
                             c[97] / (double)c[96], 64.0 * c[96] / (double)std::max(c[0], 1ull));
            for (int k = 0; k < 3; k++) // what the 64 lanes' rays wait for while a round of this kind runs (one-ray walk)
                if (c[108 + 4 * k] + c[109 + 4 * k] + c[110 + 4 * k] + c[111 + 4 * k])
                    std::fprintf(stderr, "  during %-6s rounds: lanes without a ray %5.2f, waiting for a node step %5.2f, leaf screen %5.2f, "
                                 "exact test %5.2f\n", kinds[k], c[108 + 4 * k] / (double)std::max(c[4 + 2 * k], 1ull),
                                 c[109 + 4 * k] / (double)std::max(c[4 + 2 * k], 1ull), c[110 + 4 * k] / (double)std::max(c[4 + 2 * k], 1ull),
                                 c[111 + 4 * k] / (double)std::max(c[4 + 2 * k], 1ull));
#endif
#ifdef RT_TIME_ROUNDS
            {
                double total = (double)(c[12] + c[13] + c[14] + c[15]);
                const char *names[4] = {"refill", "exact", "leaf", "node"};
                for (int k = 0; k < 4; k++)
                    std::fprintf(stderr, "  %-6s %5.1f %% of wave time, %8.0f cycles per round, %12llu rounds\n", names[k],
                                 100.0 * (double)c[12 + k] / std::max(total, 1.0),
                                 (double)c[12 + k] / (double)std::max(c[16 + k], 1ull), c[16 + k]);
                std::fprintf(stderr, "  exact round: %.0f cycles until records and ray arrive, %.0f in the f64 test, %.0f in update and pop\n",
                             (double)c[28] / (double)std::max(c[31], 1ull), (double)c[29] / (double)std::max(c[31], 1ull),
                             (double)c[30] / (double)std::max(c[31], 1ull));
            }
#endif
#ifdef RT_TIME_ROUNDS
            for (int k = 0; k < 16; k++) { // the last 16 fast-walk launches: when their wavefronts started and ended
                const unsigned long long *w = c + 32 + 8 * k;
                if (!w[4]) continue;
                const double first = (double)((1ull << 62) - w[0]), n = (double)w[4];
                std::fprintf(stderr, "  launch slot %2d: %5llu wavefronts, first start to last end %8.1f us, mean start %6.1f us "
                             "after the first, mean end %6.1f us before the last\n", k, w[4], ((double)w[1] - first) / 100.0,
                             ((double)w[2] / n - first) / 100.0, ((double)w[1] - (double)w[3] / n) / 100.0);
            }
            if (c[20] || c[21])
                std::fprintf(stderr, "  re-walk: %llu marked rays, %llu fallbacks, %.1f hits per ray; cycles per ray: enumerate %.0f "
                             "(max %llu), replay %.0f (max %llu)\n", c[20], c[21], (double)c[23] / (double)std::max(c[20], 1ull),
                             (double)c[22] / (double)std::max(c[20], 1ull), c[25], (double)c[24] / (double)std::max(c[20], 1ull), c[27]);
#endif
#if defined(RT_COUNT_VISITS) || defined(RT_TIME_ROUNDS)
            scene->visits.release();
#endif
            scene->nodes.release(); scene->qnodes.release(); scene->qnodes_c.release(); scene->pre.release(); scene->pre2.release(); scene->tickets.release(); scene->hot.release();
            scene->cold.release(); scene->pts2d.release(); scene->mats.release();
            scene->ref_of_fast.release(); scene->fast_of_ref.release();
        }
        delete scene;
        return RT_OK;
    });
}

rt_status rt_scene_get_info(const rt_scene *scene, rt_scene_info *out) {
    if (!scene || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_get_info: NULL argument");
    *out = scene->info;
    return RT_OK;
}

rt_status rt_fast_tree_check(const rt_scene_desc *desc, int reference_topology, uint32_t threads, uint32_t out[8]) {
    if (!desc || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_fast_tree_check: NULL argument");
    return guarded([&]() -> rt_status {
        const rt_scene_desc &d = *desc;
        uint32_t depth = 0, leaves = 0, max_leaf = 0;
        std::string why = validate(d, depth, leaves, max_leaf);
        if (!why.empty()) return fail(RT_ERR_INVALID_ARGUMENT, "rt_fast_tree_check: " + why);
        std::vector<Box> slot_box(d.n_slots);
        for (uint32_t s = 0; s < d.n_slots; s++) slot_box[s] = slot_bounds(d, s);
        const double pad = box_pad(d);
        rtd::FastTree ft = reference_topology ? rtd::build_fast_tree_reference(d, slot_box, pad)
                                              : rtd::build_fast_tree_sah(slot_box, pad, TOP_LEVELS, std::max(1u, threads));
        auto bad = [&](const char *what) { return fail(RT_ERR_INTERNAL, std::string("rt_fast_tree_check: ") + what); };
        if (ft.ref_of_fast.size() != d.n_slots) return bad("slot map has the wrong size");
        std::vector<uint8_t> seen(d.n_slots, 0);
        for (uint32_t r : ft.ref_of_fast) {
            if (r >= d.n_slots || seen[r]) return bad("slot map is not a permutation");
            seen[r] = 1;
        }
        // the checks run on what the device walks: the binary tree collapsed into 4-wide nodes
        rtd::WideTree wide;
        try {
            wide = rtd::collapse_to_wide(ft, FAST_STACK_ENTRIES - 1u);
        } catch (const std::length_error &e) {
            return fail(RT_ERR_UNSUPPORTED, std::string("rt_fast_tree_check: ") + e.what());
        } catch (const std::logic_error &e) {
            return fail(RT_ERR_INTERNAL, std::string("rt_fast_tree_check: ") + e.what());
        }
        std::vector<uint8_t> covered(d.n_slots, 0), visited(wide.nodes.size(), 0);
        uint32_t n_leaves = 0, leaf_max = 0, deepest = 0;
        uint64_t leaf_depth_sum = 0;
        // expected visits of a random line through the root box: sum of box area over the root's area
        double node_area = 0.0, leaf_area = 0.0, root_area = 0.0;
        auto area_of = [](const float *lo, const float *hi) {
            double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        struct Item {
            uint32_t ref, depth;
            float lo[3], hi[3];
        };
        std::vector<Item> stack;
        if (wide.root_ref != rtd::REF_NONE) {
            Item it{wide.root_ref, 0, {-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}};
            stack.push_back(it);
        }
        std::vector<uint32_t> need_of(wide.nodes.size(), 0); // filled bottom-up below
        for (size_t i = wide.nodes.size(); i-- > 0;) {       // children are emitted after their parent
            uint32_t deepest_child = 0;
            for (int k = 0; k < 4; k++) {
                uint32_t r = wide.nodes[i].ref[k];
                if (r == rtd::REF_NONE || (r & rtd::REF_LEAF)) continue;
                uint32_t c = r & rtd::WREF_INDEX_MASK;
                if (c <= i || c >= wide.nodes.size()) return bad("wide child index out of order");
                if (((r >> rtd::WREF_NEED_SHIFT) & 31u) != need_of[c]) return bad("need bits do not match the subtree");
                deepest_child = std::max(deepest_child, need_of[c]);
            }
            need_of[i] = deepest_child + 1;
        }
        while (!stack.empty()) {
            Item it = stack.back();
            stack.pop_back();
            if (it.ref & rtd::REF_LEAF) {
                uint32_t begin = it.ref & rtd::REF_SLOT_MASK, cnt = ((it.ref >> rtd::REF_CNT_SHIFT) & 7u) + 1u;
                if ((uint64_t)begin + cnt > d.n_slots) return bad("leaf range out of bounds");
                n_leaves++;
                leaf_area += area_of(it.lo, it.hi) * cnt;
                leaf_max = std::max(leaf_max, cnt);
                leaf_depth_sum += it.depth;
                for (uint32_t f = begin; f < begin + cnt; f++) {
                    if (covered[f]) return bad("primitive in two leaves");
                    covered[f] = 1;
                    const Box &b = slot_box[ft.ref_of_fast[f]];
                    for (int k = 0; k < 3; k++)
                        if (!((double)it.lo[k] <= b.lo[k] - pad && (double)it.hi[k] >= b.hi[k] + pad))
                            return bad("leaf box does not contain its primitive");
                }
                continue;
            }
            if (it.ref & rtd::REF_GROUP) return bad("group flag in a stored reference");
            uint32_t idx = it.ref & rtd::WREF_INDEX_MASK;
            if (idx >= wide.nodes.size() || visited[idx]) return bad("node reference out of range or shared");
            visited[idx] = 1;
            deepest = std::max(deepest, it.depth + 1);
            const rtd::WNode &w = wide.nodes[idx];
            if (it.depth == 0) { // the root has no stored box: the union of its children
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (int c = 0; c < 4; c++) {
                    if (w.ref[c] == rtd::REF_NONE) continue;
                    const float cl[3] = {w.lox[c], w.loy[c], w.loz[c]}, ch[3] = {w.hix[c], w.hiy[c], w.hiz[c]};
                    for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], cl[k]), hi[k] = std::max(hi[k], ch[k]);
                }
                root_area = area_of(lo, hi);
                node_area += root_area;
            } else {
                node_area += area_of(it.lo, it.hi);
            }
            uint32_t used = 0;
            for (int c = 0; c < 4; c++) {
                if (w.ref[c] == rtd::REF_NONE) continue;
                if ((uint32_t)c != used) return bad("wide node with an empty place before a child"); // the walk relies on it
                used++;
                Item ch{w.ref[c], it.depth + 1, {w.lox[c], w.loy[c], w.loz[c]}, {w.hix[c], w.hiy[c], w.hiz[c]}};
                for (int k = 0; k < 3; k++)
                    if (!(ch.lo[k] >= it.lo[k] && ch.hi[k] <= it.hi[k])) return bad("child box sticks out of its parent's");
                stack.push_back(ch);
            }
            if (used < 2) return bad("wide node with fewer than two children");
        }
        for (uint32_t f = 0; f < d.n_slots; f++)
            if (!covered[f]) return bad("primitive in no leaf");
        if (deepest != wide.depth) return bad("depth bookkeeping is off");
        // the 64-byte records the device reads: same references, boxes that contain the f32 boxes and are
        // at most two quantisation steps larger
        std::vector<rtd::QNode> qn = rtd::quantise_wide(wide);
        if (qn.size() != wide.nodes.size()) return bad("quantised node count");
        for (size_t i = 0; i < qn.size(); i++) {
            const rtd::WNode &w = wide.nodes[i];
            const float *wlo[3] = {w.lox, w.loy, w.loz}, *whi[3] = {w.hix, w.hiy, w.hiz};
            for (int c = 0; c < 4; c++) {
                if (qn[i].ref[c] != w.ref[c]) return bad("quantised node has other references");
                if (w.ref[c] == rtd::REF_NONE) continue;
                double lo[3], hi[3];
                rtd::decode_child_box(qn[i], c, lo, hi);
                for (int a = 0; a < 3; a++) {
                    const double step = (double)qn[i].scale[a];
                    if (!(lo[a] <= (double)wlo[a][c] && hi[a] >= (double)whi[a][c])) return bad("quantised box too small");
                    if (!(lo[a] > (double)wlo[a][c] - step && hi[a] < (double)whi[a][c] + step))
                        return bad("quantised box more than one step too large");
                }
            }
        }
        if (wide.depth > std::max(FAST_STACK_ENTRIES - 1u, (ft.max_depth + 1) / 2) && !(wide.root_ref & rtd::REF_LEAF))
            return bad("wide tree deeper than both the stack's free depth and half the binary depth");
        if (!reference_topology && ft.max_depth > rtd::FAST_TREE_MAX_DEPTH) return bad("tree deeper than the builder's limit");
        out[0] = (uint32_t)wide.nodes.size();
        out[1] = wide.depth;
        out[2] = ft.max_depth; // of the binary build tree
        out[3] = n_leaves;
        out[4] = leaf_max;
        out[5] = n_leaves ? (uint32_t)(100 * leaf_depth_sum / n_leaves) : 0; // mean leaf depth x 100
        // surface area estimate of the walk's work per line through the scene, x 1000
        out[6] = root_area > 0.0 ? (uint32_t)std::min(4.0e9, 1000.0 * node_area / root_area) : 0;
        out[7] = root_area > 0.0 ? (uint32_t)std::min(4.0e9, 1000.0 * leaf_area / root_area) : 0;
        return RT_OK;
    });
}

// ============================================================================
// closest hit
// ============================================================================
static rt_status check_mode(const rt_scene *scene, rt_traverse_mode mode) {
    if (mode != RT_TRAVERSE_FAST && mode != RT_TRAVERSE_REFERENCE_ORDER)
        return fail(RT_ERR_INVALID_ARGUMENT, "unknown traversal mode");
    if (mode == RT_TRAVERSE_FAST && scene->view.stack_depth > FAST_STACK_MAX) // the block's stacks must fit 64 KB of LDS
        return fail(RT_ERR_UNSUPPORTED, "fast tree deeper than 37 wide levels: use RT_TRAVERSE_REFERENCE_ORDER");
    return RT_OK;
}

rt_status rt_intersect_batch_device(rt_scene *scene, const double *d_rays, uint64_t n, rt_traverse_mode mode,
                                    uint32_t *d_prim_id, double *d_distance, double *d_pos, double *d_normal,
                                    void *stream) {
    if (!scene || (n && (!d_rays || !d_prim_id)))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_intersect_batch_device: NULL argument");
    if (n > (1ull << 40)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_intersect_batch_device: n too large");
    rt_status st = check_mode(scene, mode);
    if (st != RT_OK) return st;
    return guarded([&]() -> rt_status {
        DeviceGuard g(scene->device);
        rtd::launch_intersect(scene->view, mode, d_rays, n, d_prim_id, d_distance, d_pos, d_normal,
                              scene->next_ticket(static_cast<hipStream_t>(stream)), static_cast<hipStream_t>(stream));
        HIP_TRY(hipGetLastError());
        return RT_OK;
    });
}

rt_status rt_intersect_batch(rt_scene *scene, const double *rays, uint64_t n, rt_traverse_mode mode,
                             uint32_t *prim_id, double *distance, double *pos, double *normal, rt_stats *stats) {
    if (!scene || (n && (!rays || !prim_id))) return fail(RT_ERR_INVALID_ARGUMENT, "rt_intersect_batch: NULL argument");
    rt_status st = check_mode(scene, mode);
    if (st != RT_OK) return st;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return RT_OK;
    return guarded([&]() -> rt_status {
        DeviceGuard g(scene->device);
        DevBuf d_rays, d_id, d_dist, d_pos, d_nrm;
        d_rays.alloc(n * 48);
        d_id.alloc(n * 4);
        if (distance) d_dist.alloc(n * 8);
        if (pos) d_pos.alloc(n * 24);
        if (normal) d_nrm.alloc(n * 24);
        EventPair total, trav;
        HIP_TRY(hipEventRecord(total.a, nullptr));
        HIP_TRY(hipMemcpy(d_rays.p, rays, n * 48, hipMemcpyHostToDevice));
        HIP_TRY(hipEventRecord(trav.a, nullptr));
        rtd::launch_intersect(scene->view, mode, d_rays.as<double>(), n, d_id.as<uint32_t>(), d_dist.as<double>(),
                              d_pos.as<double>(), d_nrm.as<double>(), scene->next_ticket(nullptr), nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(trav.b, nullptr));
        HIP_TRY(hipMemcpy(prim_id, d_id.p, n * 4, hipMemcpyDeviceToHost));
        if (distance) HIP_TRY(hipMemcpy(distance, d_dist.p, n * 8, hipMemcpyDeviceToHost));
        if (pos) HIP_TRY(hipMemcpy(pos, d_pos.p, n * 24, hipMemcpyDeviceToHost));
        if (normal) HIP_TRY(hipMemcpy(normal, d_nrm.p, n * 24, hipMemcpyDeviceToHost));
        HIP_TRY(hipEventRecord(total.b, nullptr));
        if (stats) {
            stats->rays = n;
            stats->traverse_ms = trav.ms();
            stats->total_ms = total.ms();
            stats->traverse_launches = 1;
        }
        return RT_OK;
    });
}

rt_status rt_intersect_brute_force(rt_scene *scene, const double *rays, uint64_t n, uint32_t *prim_id,
                                   double *distance) {
    if (!scene || (n && (!rays || !prim_id || !distance)))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_intersect_brute_force: NULL argument");
    if (n == 0) return RT_OK;
    return guarded([&]() -> rt_status {
        DeviceGuard g(scene->device);
        DevBuf d_rays, d_ids, d_dist;
        d_rays.alloc(n * 48);
        d_ids.alloc(n * 4);
        d_dist.alloc(n * 8);
        HIP_TRY(hipMemcpy(d_rays.p, rays, n * 48, hipMemcpyHostToDevice));
        rtd::launch_intersect_brute(scene->view, d_rays.as<double>(), n, d_ids.as<uint32_t>(), d_dist.as<double>(),
                                    nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(prim_id, d_ids.p, n * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(distance, d_dist.p, n * 8, hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

rt_status rt_intersect_primitives(rt_scene *scene, const double *rays, const uint32_t *prim_ids, uint64_t n,
                                  uint8_t *hit, double *pos, double *normal) {
    if (!scene || (n && (!rays || !prim_ids || !hit)))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_intersect_primitives: NULL argument");
    if (n == 0) return RT_OK;
    return guarded([&]() -> rt_status {
        std::vector<uint32_t> slots(n);
        for (uint64_t i = 0; i < n; i++) {
            uint32_t id = prim_ids[i];
            if (id >= scene->slot_of_prim.size() || scene->slot_of_prim[id] == RT_MISS)
                return fail(RT_ERR_INVALID_ARGUMENT, "rt_intersect_primitives: unknown prim_id");
            slots[i] = scene->slot_of_prim[id];
        }
        DeviceGuard g(scene->device);
        DevBuf d_rays, d_slots, d_hit, d_pos, d_nrm;
        d_rays.alloc(n * 48);
        d_slots.alloc(n * 4);
        d_hit.alloc(n);
        if (pos) d_pos.alloc(n * 24);
        if (normal) d_nrm.alloc(n * 24);
        HIP_TRY(hipMemcpy(d_rays.p, rays, n * 48, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_slots.p, slots.data(), n * 4, hipMemcpyHostToDevice));
        rtd::launch_intersect_prims(scene->view, d_rays.as<double>(), d_slots.as<uint32_t>(), n, d_hit.as<uint8_t>(),
                                    d_pos.as<double>(), d_nrm.as<double>(), nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(hit, d_hit.p, n, hipMemcpyDeviceToHost));
        if (pos) HIP_TRY(hipMemcpy(pos, d_pos.p, n * 24, hipMemcpyDeviceToHost));
        if (normal) HIP_TRY(hipMemcpy(normal, d_nrm.p, n * 24, hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

} // extern "C"

// ============================================================================
// wavefront integrator state
// ============================================================================
namespace {

constexpr uint32_t N_COUNTERS = 16;

struct PathPool {
    DevBuf f64s, u32s, u8s, counters;
    rtd::Paths view{};
    uint32_t max_depth = 0;

    // The old buffers are given back first (a pass is sized to what the device can hold), so a
    // failure leaves the pool EMPTY (capacity 0), never pointing at freed memory.
    void alloc(uint64_t capacity, uint32_t depth) {
        view = rtd::Paths{};
        max_depth = 0;
        f64s.release(); u32s.release(); u8s.release(); counters.release();
        uint64_t cap = (capacity + 63) & ~63ull;
        try {
            f64s.alloc(cap * 8 * (15 + depth)); // 2 x 6 ray + 3 hit position + history
            u32s.alloc(cap * 4 * (3 + depth));  // 2 x id + hit material + history
            u8s.alloc(cap * 2);
            counters.alloc(N_COUNTERS * 4);
        } catch (...) {
            f64s.release(); u32s.release(); u8s.release(); counters.release();
            throw;
        }
        max_depth = depth;
        double *f = f64s.as<double>();
        view.rays = f;
        view.hit = f + cap * 12;
        view.cosine = f + cap * 15;
        uint32_t *u = u32s.as<uint32_t>();
        view.ids = u;
        view.hit_mat = u + cap * 2;
        view.material = u + cap * 3;
        view.length = u8s.as<uint8_t>();
        view.emissive = u8s.as<uint8_t>() + cap;
        view.counters = counters.as<uint32_t>();
        view.capacity = cap;
        view.first_bundle = nullptr;
        view.first_period = 1;
    }
};

// Runs tracer.rs:30-55 for the n paths already initialised in `pool`. `expected[d]`
// (may be NULL) is the caller's guess of how many paths are still alive at bounce d; it
// only sizes the grids, the kernels read the true count on the device.
void run_bounces(rt_scene *scene, PathPool &pool, uint64_t n, uint32_t max_depth, rt_traverse_mode mode,
                 uint64_t seed, const uint32_t *d_keys, const rtd::FilmView *film, uint32_t first_sample,
                 hipStream_t stream, hipEvent_t *trav_events /* 2*max_depth or NULL */,
                 const uint64_t *expected = nullptr, uint32_t walk_blocks_per_cu = 0 /* 0 = all that fit */) {
    for (uint32_t depth = 0; depth < max_depth; depth++) {
        uint64_t guess = expected && expected[depth] ? std::min<uint64_t>(n, expected[depth]) : n;
        if (trav_events) HIP_TRY(hipEventRecord(trav_events[2 * depth], stream));
        rtd::launch_traverse_paths(scene->view, mode, pool.view, depth, guess, scene->next_ticket(stream), stream,
                                   walk_blocks_per_cu, /* the first segments are the caller's rays */ depth == 0 && !film);
        if (trav_events) HIP_TRY(hipEventRecord(trav_events[2 * depth + 1], stream));
        rtd::launch_shade(scene->view, pool.view, depth, guess, seed, d_keys, film, first_sample, stream);
    }
    HIP_TRY(hipGetLastError());
}

} // namespace

extern "C" {

rt_status rt_trace_paths(rt_scene *scene, const double *rays, const uint32_t *keys, uint64_t n, uint64_t seed,
                         uint32_t max_depth, rt_traverse_mode mode, double *radiance, rt_stats *stats) {
    if (!scene || (n && (!rays || !keys || !radiance)))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_paths: NULL argument");
    if (max_depth == 0 || max_depth > N_COUNTERS - 2)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_paths: max_depth must be 1..14");
    if (n >= (1ull << 31)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_paths: n too large");
    rt_status st = check_mode(scene, mode);
    if (st != RT_OK) return st;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return RT_OK;
    return guarded([&]() -> rt_status {
        DeviceGuard g(scene->device);
        PathPool pool;
        pool.alloc(n, max_depth);
        DevBuf d_rays, d_keys, d_rad;
        d_rays.alloc(n * 48);
        d_keys.alloc(n * 12);
        d_rad.alloc(n * 24);
        HIP_TRY(hipMemcpy(d_rays.p, rays, n * 48, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_keys.p, keys, n * 12, hipMemcpyHostToDevice));
        EventPair total;
        HIP_TRY(hipEventRecord(total.a, nullptr));
        HIP_TRY(hipMemsetAsync(pool.counters.p, 0, N_COUNTERS * 4, nullptr));
        rtd::launch_paths_from_rays(pool.view, d_rays.as<double>(), n, nullptr);
        run_bounces(scene, pool, n, max_depth, mode, seed, d_keys.as<uint32_t>(), nullptr, 0, nullptr, nullptr);
        rtd::launch_fold_paths(scene->view, pool.view, n, d_rad.as<double>(), nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(total.b, nullptr));
        HIP_TRY(hipMemcpy(radiance, d_rad.p, n * 24, hipMemcpyDeviceToHost));
        if (stats) {
            uint32_t c[N_COUNTERS];
            HIP_TRY(hipMemcpy(c, pool.counters.p, sizeof c, hipMemcpyDeviceToHost));
            for (uint32_t k = 0; k < max_depth; k++) stats->rays += c[k];
            stats->paths = n;
            stats->total_ms = total.ms();
            stats->traverse_launches = max_depth;
        }
        return RT_OK;
    });
}

// bytes of path state for `cap` paths of up to `depth` bounces, counters included
static uint64_t static_scratch_need(uint64_t cap, uint32_t depth) {
    return 256 + cap * (8ull * (15 + depth) + 4ull * (3 + depth) + 2ull);
}

rt_status rt_trace_paths_static(rt_scene *scene, const double *d_rays, const uint32_t *d_keys, uint64_t n,
                                uint64_t seed, uint32_t max_depth, rt_traverse_mode mode, double *d_radiance,
                                void *stream_) {
    if (!scene || (n && (!d_rays || !d_keys || !d_radiance)))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_paths_static: NULL argument");
    if (max_depth == 0 || max_depth > N_COUNTERS - 2)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_paths_static: max_depth must be 1..14");
    const uint64_t cap = (n + 63) & ~63ull;
    if (n > (1ull << 31) || static_scratch_need(cap, max_depth) > RT_STATIC_SCRATCH_BYTES)
        return fail(RT_ERR_INVALID_ARGUMENT,
                    "rt_trace_paths_static: " + std::to_string(n) + " paths of depth " + std::to_string(max_depth) +
                        " need " + std::to_string(static_scratch_need(cap, max_depth)) + " bytes of path state, the " +
                        "static scratch holds " + std::to_string((uint64_t)RT_STATIC_SCRATCH_BYTES) + " (" +
                        std::to_string(RT_STATIC_MAX_PATHS) + " paths at depth 5); use rt_trace_paths");
    rt_status st = check_mode(scene, mode);
    if (st != RT_OK) return st;
    if (n == 0) return RT_OK;
    if (!scene->static_scratch) return fail(RT_ERR_INTERNAL, "rt_trace_paths_static: scene has no scratch address");
    return guarded([&]() -> rt_status {
        hipStream_t stream = static_cast<hipStream_t>(stream_);
        DeviceGuard g(scene->device);
        // carve the path state out of this device's scratch: counters, then f64, u32 and u8 planes
        unsigned char *base = scene->static_scratch;
        PathPool pool; // used as a view only: its buffers stay empty, nothing is allocated or freed
        rtd::Paths &v = pool.view;
        v.counters = reinterpret_cast<uint32_t *>(base);
        double *f = reinterpret_cast<double *>(base + 256);
        v.rays = f;
        v.hit = f + cap * 12;
        v.cosine = f + cap * 15;
        uint32_t *u = reinterpret_cast<uint32_t *>(f + cap * (15 + max_depth));
        v.ids = u;
        v.hit_mat = u + cap * 2;
        v.material = u + cap * 3;
        uint8_t *b = reinterpret_cast<uint8_t *>(u + cap * (3 + max_depth));
        v.length = b;
        v.emissive = b + cap;
        v.capacity = cap;
        v.first_bundle = nullptr;
        v.first_period = 1;
        pool.max_depth = max_depth;
        // the array keeps the previous call's contents: the live-path counters must start at zero
        HIP_TRY(hipMemsetAsync(v.counters, 0, N_COUNTERS * 4, stream));
        rtd::launch_paths_from_rays(v, d_rays, n, stream);
        run_bounces(scene, pool, n, max_depth, mode, seed, d_keys, nullptr, 0, stream, nullptr);
        rtd::launch_fold_paths(scene->view, v, n, d_radiance, stream);
        HIP_TRY(hipGetLastError());
        return RT_OK;
    });
}

rt_status rt_ray_bundle(const rt_camera *cam, uint32_t width, uint32_t height, uint64_t seed, int device,
                        double *rays) {
    if (!cam || !rays) return fail(RT_ERR_INVALID_ARGUMENT, "rt_ray_bundle: NULL argument");
    if (width == 0 || height == 0 || (uint64_t)width * height >= (1ull << 31))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_ray_bundle: bad image size");
    return guarded([&]() -> rt_status {
        int ndev = usable_devices();
        if (ndev == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device is visible; this library has no CPU fallback");
        if (device < 0 || device >= ndev) return fail(RT_ERR_INVALID_ARGUMENT, "rt_ray_bundle: device out of range");
        DeviceGuard g(device);
        uint64_t npix = (uint64_t)width * height;
        DevBuf d;
        d.alloc(npix * 30 * 8);
        rtd::launch_raygen(*cam, width, height, seed, nullptr, (uint32_t)npix, d.as<double>(), nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(rays, d.p, npix * 30 * 8, hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

rt_status rt_tone_map(const double *radiance, uint64_t n, int device, uint8_t *rgba) {
    if (n && (!radiance || !rgba)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_tone_map: NULL argument");
    if (n == 0) return RT_OK;
    return guarded([&]() -> rt_status {
        int ndev = usable_devices();
        if (ndev == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device is visible; this library has no CPU fallback");
        if (device < 0 || device >= ndev) return fail(RT_ERR_INVALID_ARGUMENT, "rt_tone_map: device out of range");
        DeviceGuard g(device);
        DevBuf a, b;
        a.alloc(n * 24);
        b.alloc(n * 4);
        HIP_TRY(hipMemcpy(a.p, radiance, n * 24, hipMemcpyHostToDevice));
        rtd::launch_tone_map(a.as<double>(), n, b.as<uint8_t>(), nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(rgba, b.p, n * 4, hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

rt_status rt_tone_map_device(const double *d_radiance, uint64_t n, uint8_t *d_rgba, void *stream) {
    if (n && (!d_radiance || !d_rgba)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_tone_map_device: NULL argument");
    return guarded([&]() -> rt_status {
        if (usable_devices() == 0)
            return fail(RT_ERR_NO_DEVICE, "no HIP device is visible; this library has no CPU fallback");
        rtd::launch_tone_map(d_radiance, n, d_rgba, static_cast<hipStream_t>(stream));
        HIP_TRY(hipGetLastError());
        return RT_OK;
    });
}

} // extern "C"

// ============================================================================
// film
// ============================================================================
struct rt_film {
    rt_scene *scene = nullptr;
    uint32_t width = 0, height = 0, shard_index = 0, shard_count = 1, tile_size = 32;
    std::vector<uint32_t> owned_host;
    DevBuf owned, bundle, totals;
    // A render call splits its samples into passes and keeps up to TWO passes in flight, one per lane: lane 0
    // on the caller's stream, lane 1 on a stream of the film's own. The walk is bound by vector-instruction
    // issue and the kernels around it (shading, film accumulation) by memory streaming, and every launch of the
    // persistent walk ends with a tail in which its wavefronts drain: with two passes in flight one lane's
    // streaming kernels and tails run under the other lane's walk. The passes' film accumulations are chained
    // by events in sample order, so the sums (and the frame) are bit-identical to one pass after the other.
    struct Lane {
        PathPool pool;
        std::vector<hipEvent_t> events;    // [2*depth] traversal brackets of the lane's last pass, + 2 for the pass
        struct Pending { uint32_t depth; bool live = false; uint64_t paths = 0; } pending;
        DevBuf counter_snap;               // counters of the lane's last pass (copied on its stream)
        hipEvent_t acc_done = nullptr;     // recorded after the lane's last k_accumulate_film
    } lane[2];
    uint32_t lanes = 1;                    // 1 = one pass after the other on the caller's stream; 2 = two in flight
    hipStream_t side = nullptr;            // lane 1's stream
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t last_acc = nullptr;         // the most recent accumulation: the next one waits for it
    bool have_rays = false;
    uint32_t spp = 0;
    rtd::FilmView view{};
    // statistics
    uint64_t rays = 0, paths = 0, launches = 0;
    double traverse_ms = 0.0, total_ms = 0.0;
    // Recorded on the NULL stream after rt_film_set_camera / set_rays / clear; render and resolve make
    // their stream wait for it, so a caller's non-blocking stream sees the bundle and the cleared sums.
    hipEvent_t ready = nullptr;
    double alive_frac[N_COUNTERS] = {}; // live fraction per bounce seen in the last pass (0 = unknown)
};

namespace {

// Target number of paths per pass (about 200 bytes of device memory each: 40 GB of the 288 GB; a device with
// less free memory gets smaller passes, see the allocation loop). Long launches matter: the later bounces of a
// pass carry a third of its paths, and every launch of the persistent walk ends with a tail in which its
// wavefronts drain (they are resident for 91 % of a launch of 52 M rays). Measured on the 1080p hippo render
// (16 samples per call, so 166 M paths at most): 44 M paths per pass 6.95, 88 Mi 7.37, 166 M 7.57 G queries/s.
constexpr uint64_t PASS_PATHS = 192u << 20;

void film_collect(rt_film *f, rt_film::Lane &l) {
    // fold the lane's last pass (event timings, counters) into the totals (synchronises with that pass)
    if (!l.pending.live) return;
    uint32_t depth = l.pending.depth;
    HIP_TRY(hipEventSynchronize(l.events[2 * depth + 1]));
    for (uint32_t k = 0; k < depth; k++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, l.events[2 * k], l.events[2 * k + 1]));
        f->traverse_ms += ms;
    }
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, l.events[2 * depth], l.events[2 * depth + 1]));
    f->total_ms += ms;
    uint32_t c[N_COUNTERS];
    HIP_TRY(hipMemcpy(c, l.counter_snap.p, sizeof c, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < depth; k++) {
        f->rays += c[k];
        if (l.pending.paths) f->alive_frac[k] = (double)c[k] / (double)l.pending.paths;
    }
    f->launches += depth;
    l.pending.live = false;
}
void film_collect(rt_film *f) {
    for (auto &l : f->lane) film_collect(f, l);
}

} // namespace

extern "C" {

rt_status rt_film_create(rt_scene *scene, uint32_t width, uint32_t height, uint32_t shard_index,
                         uint32_t shard_count, uint32_t tile_size, rt_film **out) {
    if (!scene || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_create: NULL argument");
    *out = nullptr;
    if (width == 0 || height == 0 || (uint64_t)width * height >= (1ull << 28))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_create: bad image size");
    if (shard_count == 0 || shard_index >= shard_count || tile_size == 0)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_create: bad shard arguments");
    return guarded([&]() -> rt_status {
        auto f = std::make_unique<rt_film>();
        f->scene = scene;
        f->width = width; f->height = height;
        f->shard_index = shard_index; f->shard_count = shard_count; f->tile_size = tile_size;
        uint32_t tiles_x = (width + tile_size - 1) / tile_size;
        // The owned pixels are listed block by block (PIXEL_BLOCK x PIXEL_BLOCK pixels, row-major inside), not row
        // by row: consecutive paths of a pass are consecutive list entries, so the 64 camera rays of a wavefront
        // then cover a small square of the image instead of a strip one pixel high, and so do the rays that
        // bounce off what they hit. The order decides nothing but which lane works on which pixel.
        uint32_t pb = PIXEL_BLOCK;
        if (const char *e = std::getenv("RT_TUNE_PIXEL_BLOCK")) // experiments; 1 = row by row
            if (std::atoi(e) >= 1 && std::atoi(e) <= 64) pb = (uint32_t)std::atoi(e);
        for (uint32_t by = 0; by < height; by += pb)
            for (uint32_t bx = 0; bx < width; bx += pb)
                for (uint32_t y = by; y < std::min(by + pb, height); y++)
                    for (uint32_t x = bx; x < std::min(bx + pb, width); x++) {
                        uint32_t tile = (y / tile_size) * tiles_x + x / tile_size;
                        if (tile % shard_count == shard_index) f->owned_host.push_back(y * width + x);
                    }
        uint32_t n_owned = (uint32_t)f->owned_host.size();
        DeviceGuard g(scene->device);
        f->owned.alloc((size_t)n_owned * 4);
        if (n_owned) HIP_TRY(hipMemcpy(f->owned.p, f->owned_host.data(), (size_t)n_owned * 4, hipMemcpyHostToDevice));
        f->bundle.alloc((size_t)n_owned * 30 * 8);
        f->totals.alloc((size_t)n_owned * 3 * 8);
        HIP_TRY(hipMemset(f->totals.p, 0, f->totals.bytes));
        for (auto &l : f->lane) {
            l.counter_snap.alloc(N_COUNTERS * 4);
            l.events.resize(2 * (N_COUNTERS) + 2, nullptr);
            for (auto &e : l.events) HIP_TRY(hipEventCreate(&e));
            HIP_TRY(hipEventCreateWithFlags(&l.acc_done, hipEventDisableTiming));
        }
        HIP_TRY(hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&f->fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&f->join, hipEventDisableTiming));
        if (const char *e = std::getenv("RT_FILM_LANES")) // experiments: 2 = two passes in flight
            f->lanes = std::atoi(e) == 2 ? 2u : 1u;
        f->view.owned = f->owned.as<uint32_t>();
        f->view.bundle = f->bundle.as<double>();
        for (int k = 0; k < 3; k++) f->view.total[k] = f->totals.as<double>() + (size_t)n_owned * k;
        f->view.n_owned = n_owned;
        f->view.width = width;
        f->view.height = height;
        HIP_TRY(hipEventCreateWithFlags(&f->ready, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(f->ready, nullptr)); // orders the memset above
        *out = f.release();
        return RT_OK;
    });
}

rt_status rt_film_destroy(rt_film *film) {
    if (!film) return RT_OK;
    return guarded([&]() -> rt_status {
        {
            DeviceGuard g(film->scene->device);
            (void)hipDeviceSynchronize();
            for (auto &l : film->lane) {
                for (auto &e : l.events)
                    if (e) (void)hipEventDestroy(e);
                if (l.acc_done) (void)hipEventDestroy(l.acc_done);
                l.counter_snap.release();
                l.pool.f64s.release(); l.pool.u32s.release(); l.pool.u8s.release(); l.pool.counters.release();
            }
            if (film->side) (void)hipStreamDestroy(film->side);
            if (film->fork) (void)hipEventDestroy(film->fork);
            if (film->join) (void)hipEventDestroy(film->join);
            if (film->ready) (void)hipEventDestroy(film->ready);
            film->owned.release(); film->bundle.release(); film->totals.release();
        }
        delete film;
        return RT_OK;
    });
}

rt_status rt_film_set_camera(rt_film *film, const rt_camera *cam, uint64_t seed) {
    if (!film || !cam) return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_set_camera: NULL argument");
    return guarded([&]() -> rt_status {
        DeviceGuard g(film->scene->device);
        rtd::launch_raygen(*cam, film->width, film->height, seed, film->view.owned, film->view.n_owned,
                           film->bundle.as<double>(), nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(film->ready, nullptr));
        film->have_rays = true;
        return RT_OK;
    });
}

rt_status rt_film_set_rays(rt_film *film, const double *rays) {
    if (!film || !rays) return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_set_rays: NULL argument");
    return guarded([&]() -> rt_status {
        DeviceGuard g(film->scene->device);
        std::vector<double> gathered((size_t)film->view.n_owned * 30);
        for (uint32_t k = 0; k < film->view.n_owned; k++)
            std::memcpy(&gathered[(size_t)k * 30], rays + (size_t)film->owned_host[k] * 30, 30 * sizeof(double));
        if (!gathered.empty())
            HIP_TRY(hipMemcpy(film->bundle.p, gathered.data(), gathered.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipEventRecord(film->ready, nullptr));
        film->have_rays = true;
        return RT_OK;
    });
}

rt_status rt_film_clear(rt_film *film) {
    if (!film) return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_clear: NULL film");
    return guarded([&]() -> rt_status {
        DeviceGuard g(film->scene->device);
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemset(film->totals.p, 0, film->totals.bytes));
        HIP_TRY(hipEventRecord(film->ready, nullptr));
        film->spp = 0;
        film->rays = film->paths = film->launches = 0;
        film->traverse_ms = film->total_ms = 0.0;
        for (auto &l : film->lane) l.pending.live = false;
        film->last_acc = nullptr;
        for (double &a : film->alive_frac) a = 0.0;
        return RT_OK;
    });
}

rt_status rt_film_render(rt_film *film, uint32_t first_sample, uint32_t n_samples, uint64_t seed,
                         uint32_t max_depth, rt_traverse_mode mode, void *stream_) {
    if (!film) return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_render: NULL film");
    if (!film->have_rays) return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_render: no camera or rays set");
    if (max_depth == 0 || max_depth > N_COUNTERS - 2)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_render: max_depth must be 1..14");
    rt_status st = check_mode(film->scene, mode);
    if (st != RT_OK) return st;
    if (n_samples == 0 || film->view.n_owned == 0) {
        film->spp += n_samples;
        return RT_OK;
    }
    return guarded([&]() -> rt_status {
        hipStream_t stream = static_cast<hipStream_t>(stream_);
        DeviceGuard g(film->scene->device);
        HIP_TRY(hipStreamWaitEvent(stream, film->ready, 0));
        uint64_t per_sample = (uint64_t)film->view.n_owned * 5;
        static const uint64_t pass_paths = [] {
            const char *e = std::getenv("RT_TUNE_PASS_PATHS"); // experiment hook
            return e && std::atoll(e) > 0 ? (uint64_t)std::atoll(e) : PASS_PATHS;
        }();
        // Two lanes when there are at least two samples to share out; a pass holds up to pass_paths paths.
        const uint32_t lanes = (film->lanes >= 2 && n_samples >= 2) ? 2u : 1u;
        uint32_t per_pass = (uint32_t)std::max<uint64_t>(1, pass_paths / per_sample);
        per_pass = std::min(per_pass, (n_samples + lanes - 1) / lanes);
        for (uint32_t li = 0; li < lanes; li++) {
            PathPool &pool = film->lane[li].pool;
            if (pool.view.capacity >= per_sample * per_pass && pool.max_depth >= max_depth) continue;
            HIP_TRY(hipDeviceSynchronize());
            for (;;) { // a pass as large as the device's free memory allows
                try {
                    pool.alloc(per_sample * per_pass, max_depth);
                    break;
                } catch (const HipFail &h) {
                    if (h.st != RT_ERR_OUT_OF_MEMORY || per_pass == 1) throw;
                    (void)hipGetLastError();
                    per_pass = (per_pass + 1) / 2;
                }
            }
        }
        for (uint32_t li = 0; li < lanes; li++)
            per_pass = (uint32_t)std::min<uint64_t>(per_pass, film->lane[li].pool.view.capacity / per_sample);
        // How many of the blocks a CU holds each lane's persistent walk may take: both lanes together fill the CU,
        // so that neither walk keeps the other lane's kernels off the device.
        static const uint32_t share[2] = {[] { const char *e = std::getenv("RT_TUNE_WALK_SHARE0"); return e ? (uint32_t)std::atoi(e) : 3u; }(),
                                          [] { const char *e = std::getenv("RT_TUNE_WALK_SHARE1"); return e ? (uint32_t)std::atoi(e) : 2u; }()};
        if (lanes == 2) { // lane 1's stream starts where the caller's stream stands
            HIP_TRY(hipEventRecord(film->fork, stream));
            HIP_TRY(hipStreamWaitEvent(film->side, film->fork, 0));
        }
        uint32_t done = 0, turn = 0;
        while (done < n_samples) {
            const uint32_t li = lanes == 2 ? (turn++ & 1u) : 0u;
            rt_film::Lane &lane = film->lane[li];
            hipStream_t st = li == 0 ? stream : film->side;
            uint32_t s = std::min(per_pass, n_samples - done);
            uint64_t n = per_sample * s;
            film_collect(film, lane); // the lane's events below are about to be reused
            HIP_TRY(hipEventRecord(lane.events[2 * max_depth], st));
            HIP_TRY(hipMemsetAsync(lane.pool.counters.p, 0, N_COUNTERS * 4, st));
            lane.pool.view.first_bundle = film->view.bundle;
            lane.pool.view.first_period = (uint32_t)per_sample;
            rtd::launch_paths_from_film(lane.pool.view, film->view, s, st);
            // the previous pass tells how many paths survive to each bounce: size the grids for that
            uint64_t expected[N_COUNTERS] = {};
            for (uint32_t k = 0; k < max_depth; k++)
                if (film->alive_frac[k] > 0.0) expected[k] = (uint64_t)(film->alive_frac[k] * 1.05 * (double)n) + 4096;
            run_bounces(film->scene, lane.pool, n, max_depth, mode, seed, nullptr, &film->view,
                        first_sample + done, st, lane.events.data(), expected, lanes == 2 ? share[li] : 0u);
            // main.rs:66 adds the samples in order: this pass's sums follow the previous pass's
            if (film->last_acc && film->last_acc != lane.acc_done) HIP_TRY(hipStreamWaitEvent(st, film->last_acc, 0));
            rtd::launch_accumulate_film(film->scene->view, lane.pool.view, film->view, s, st);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(lane.acc_done, st));
            film->last_acc = lane.acc_done;
            HIP_TRY(hipMemcpyAsync(lane.counter_snap.p, lane.pool.counters.p, N_COUNTERS * 4,
                                   hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipEventRecord(lane.events[2 * max_depth + 1], st));
            lane.pending.depth = max_depth;
            lane.pending.paths = n;
            lane.pending.live = true;
            film->paths += n;
            done += s;
        }
        if (lanes == 2) { // the caller's stream continues after both lanes
            HIP_TRY(hipEventRecord(film->join, film->side));
            HIP_TRY(hipStreamWaitEvent(stream, film->join, 0));
        }
        film->spp += n_samples;
        return RT_OK;
    });
}

rt_status rt_film_resolve_device(rt_film *film, double *d_radiance, uint8_t *d_rgba, void *stream_) {
    if (!film) return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_resolve_device: NULL film");
    return guarded([&]() -> rt_status {
        hipStream_t stream = static_cast<hipStream_t>(stream_);
        DeviceGuard g(film->scene->device);
        size_t npix = (size_t)film->width * film->height;
        HIP_TRY(hipStreamWaitEvent(stream, film->ready, 0));
        if (d_radiance) HIP_TRY(hipMemsetAsync(d_radiance, 0, npix * 24, stream));
        if (d_rgba) HIP_TRY(hipMemsetAsync(d_rgba, 0, npix * 4, stream));
        rtd::launch_resolve(film->view, std::max(1u, film->spp), d_radiance, d_rgba, stream);
        HIP_TRY(hipGetLastError());
        return RT_OK;
    });
}

rt_status rt_film_resolve(rt_film *film, double *radiance, uint8_t *rgba) {
    if (!film) return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_resolve: NULL film");
    return guarded([&]() -> rt_status {
        DeviceGuard g(film->scene->device);
        size_t npix = (size_t)film->width * film->height;
        DevBuf a, b;
        if (radiance) a.alloc(npix * 24);
        if (rgba) b.alloc(npix * 4);
        HIP_TRY(hipDeviceSynchronize());
        rt_status st = rt_film_resolve_device(film, a.as<double>(), b.as<uint8_t>(), nullptr);
        if (st != RT_OK) return st;
        if (radiance) HIP_TRY(hipMemcpy(radiance, a.p, npix * 24, hipMemcpyDeviceToHost));
        if (rgba) HIP_TRY(hipMemcpy(rgba, b.p, npix * 4, hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

rt_status rt_film_get_stats(rt_film *film, rt_stats *out) {
    if (!film || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_film_get_stats: NULL argument");
    return guarded([&]() -> rt_status {
        DeviceGuard g(film->scene->device);
        film_collect(film);
        std::memset(out, 0, sizeof *out);
        out->rays = film->rays;
        out->paths = film->paths;
        out->traverse_ms = film->traverse_ms;
        out->total_ms = film->total_ms;
        out->traverse_launches = film->launches;
        return RT_OK;
    });
}

rt_status rt_render(rt_scene *scene, const rt_camera *cam, uint32_t width, uint32_t height, uint32_t spp,
                    uint64_t seed, rt_traverse_mode mode, double *radiance, uint8_t *rgba, rt_stats *stats) {
    if (!scene || !cam) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: NULL argument");
    rt_film *film = nullptr;
    rt_status st = rt_film_create(scene, width, height, 0, 1, 32, &film);
    if (st != RT_OK) return st;
    st = rt_film_set_camera(film, cam, seed);
    if (st == RT_OK) st = rt_film_render(film, 0, spp, seed, RT_MAX_RAY_DEPTH, mode, nullptr);
    if (st == RT_OK) st = rt_film_resolve(film, radiance, rgba);
    if (st == RT_OK && stats) st = rt_film_get_stats(film, stats);
    std::string keep = st == RT_OK ? std::string() : std::string(rt_last_error());
    rt_film_destroy(film);
    if (st != RT_OK) return fail(st, keep);
    return RT_OK;
}

// The thread fan-out and merge of main.rs:34-104 over a list of devices. One host thread per shard: the scene
// is replicated (one copy per distinct device), every shard renders its tiles or its sample range with no
// data exchanged on the way, and ONE combine per frame sums the shards' frames on the first device.
// The combine uses peer copies (hipMemcpyPeerAsync: every other device sends its frame straight over its own
// xGMI link to the first device) and a sum kernel there rather than an RCCL reduce: the shards live in one
// process, a ring or tree reduce of a 50-400 MB frame is bound by a single link anyway, and a repeated device
// index (several shards on one GPU, which is how a single-GPU box rehearses N shards) is something an RCCL
// communicator does not accept.
rt_status rt_render_multi(const rt_scene_desc *desc, const int *devices, uint32_t n_devices, const rt_camera *cam,
                          uint32_t width, uint32_t height, uint32_t spp, uint64_t seed, rt_traverse_mode mode,
                          rt_shard_mode shard, double *radiance, uint8_t *rgba, rt_stats *stats) {
    if (!desc || !devices || !cam) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_multi: NULL argument");
    if (n_devices == 0 || n_devices > 64) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_multi: n_devices must be 1..64");
    if (shard != RT_SHARD_TILES && shard != RT_SHARD_SAMPLES)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_multi: unknown shard mode");
    if (mode != RT_TRAVERSE_FAST && mode != RT_TRAVERSE_REFERENCE_ORDER)
        return fail(RT_ERR_INVALID_ARGUMENT, "unknown traversal mode");
    if (width == 0 || height == 0 || (uint64_t)width * height >= (1ull << 28))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_multi: bad image size");
    const int ndev = usable_devices();
    if (ndev == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device is visible; this library has no CPU fallback");
    for (uint32_t k = 0; k < n_devices; k++)
        if (devices[k] < 0 || devices[k] >= ndev)
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_multi: device " + std::to_string(devices[k]) + " out of range");
    if (stats) std::memset(stats, 0, sizeof *stats);

    struct Shard {
        int device = 0;
        rt_scene *scene = nullptr; // shared by the shards of one device; owned by the first of them
        bool owns_scene = false;
        rt_film *film = nullptr;
        hipStream_t stream = nullptr;
        DevBuf d_rad, d_rgba;
        uint32_t first_sample = 0, n_samples = 0;
        rt_status st = RT_OK;
        std::string err;
        rt_stats stats{};
    };
    std::vector<Shard> shards(n_devices);
    const auto t_begin = std::chrono::steady_clock::now();
    const size_t npix = (size_t)width * height;
    const bool tiles = shard == RT_SHARD_TILES;
    for (uint32_t k = 0; k < n_devices; k++) {
        shards[k].device = devices[k];
        if (tiles) {
            shards[k].first_sample = 0;
            shards[k].n_samples = spp;
        } else { // contiguous sample ranges, the first spp % n shards one sample longer
            const uint32_t base = spp / n_devices, extra = spp % n_devices;
            shards[k].first_sample = k * base + std::min(k, extra);
            shards[k].n_samples = base + (k < extra ? 1u : 0u);
        }
    }
    auto run_all = [&](auto &&body) { // body(k) on one host thread per shard; errors are kept per shard
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < n_devices; k++)
            th.emplace_back([&, k] {
                Shard &s = shards[k];
                if (s.st != RT_OK) return;
                s.st = body(k);
                if (s.st != RT_OK) s.err = rt_last_error();
            });
        for (auto &t : th) t.join();
    };
    // 1. one scene per distinct device, built side by side
    for (uint32_t k = 0; k < n_devices; k++) {
        shards[k].owns_scene = true;
        for (uint32_t j = 0; j < k; j++)
            if (devices[j] == devices[k]) shards[k].owns_scene = false;
    }
    run_all([&](uint32_t k) -> rt_status {
        return shards[k].owns_scene ? rt_scene_create(desc, shards[k].device, &shards[k].scene) : RT_OK;
    });
    for (uint32_t k = 0; k < n_devices; k++)
        if (!shards[k].owns_scene)
            for (uint32_t j = 0; j < k; j++)
                if (devices[j] == devices[k] && shards[j].owns_scene) {
                    shards[k].scene = shards[j].scene;
                    if (shards[j].st != RT_OK) { shards[k].st = shards[j].st; shards[k].err = shards[j].err; }
                }
    // 2. every shard renders its part and resolves it into a full-size frame on its own device
    run_all([&](uint32_t k) -> rt_status {
        Shard &s = shards[k];
        return guarded([&]() -> rt_status {
            DeviceGuard g(s.device);
            HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
            rt_status st = tiles ? rt_film_create(s.scene, width, height, k, n_devices, 32, &s.film)
                                 : rt_film_create(s.scene, width, height, 0, 1, 32, &s.film);
            if (st != RT_OK) return st;
            st = rt_film_set_camera(s.film, cam, seed);
            if (st == RT_OK) st = rt_film_render(s.film, s.first_sample, s.n_samples, seed, RT_MAX_RAY_DEPTH, mode, s.stream);
            if (st != RT_OK) return st;
            s.d_rad.alloc(npix * 24);
            if (tiles) s.d_rgba.alloc(npix * 4);
            st = rt_film_resolve_device(s.film, s.d_rad.as<double>(), tiles ? s.d_rgba.as<uint8_t>() : nullptr, s.stream);
            if (st != RT_OK) return st;
            HIP_TRY(hipStreamSynchronize(s.stream));
            return rt_film_get_stats(s.film, &s.stats);
        });
    });
    // 3. the one combine per frame, on the first device
    rt_status st = RT_OK;
    std::string err;
    for (const Shard &s : shards)
        if (s.st != RT_OK && st == RT_OK) { st = s.st; err = s.err; }
    if (st == RT_OK)
        st = guarded([&]() -> rt_status {
            Shard &s0 = shards[0];
            DeviceGuard g(s0.device);
            DevBuf acc_rad, acc_rgba, stage_rad, stage_rgba;
            acc_rad.alloc(npix * 24);
            acc_rgba.alloc(npix * 4);
            if (n_devices > 1) {
                stage_rad.alloc(npix * 24);
                if (tiles) stage_rgba.alloc(npix * 4);
            }
            for (uint32_t k = 0; k < n_devices; k++) {
                Shard &s = shards[k];
                const double *src_rad = s.d_rad.as<double>();
                const uint8_t *src_rgba = s.d_rgba.as<uint8_t>();
                if (k > 0) { // straight from the shard's device into the first device's staging buffers
                    HIP_TRY(hipMemcpyPeerAsync(stage_rad.p, s0.device, s.d_rad.p, s.device, npix * 24, s0.stream));
                    src_rad = stage_rad.as<double>();
                    if (tiles) {
                        HIP_TRY(hipMemcpyPeerAsync(stage_rgba.p, s0.device, s.d_rgba.p, s.device, npix * 4, s0.stream));
                        src_rgba = stage_rgba.as<uint8_t>();
                    }
                }
                // tiles: disjoint pixels, plain sums; sample ranges: means weighted by the shard's share of spp
                const double w = tiles ? 1.0 : (double)s.n_samples / (double)std::max(1u, spp);
                rtd::launch_frame_axpy(acc_rad.as<double>(), src_rad, w, npix * 3, k == 0, s0.stream);
                if (tiles) rtd::launch_bytes_add(acc_rgba.as<uint8_t>(), src_rgba, npix, k == 0, s0.stream);
                HIP_TRY(hipGetLastError());
            }
            if (!tiles) rtd::launch_tone_map(acc_rad.as<double>(), npix, acc_rgba.as<uint8_t>(), s0.stream); // camera.rs:80, once
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s0.stream));
            if (radiance) HIP_TRY(hipMemcpy(radiance, acc_rad.p, npix * 24, hipMemcpyDeviceToHost));
            if (rgba) HIP_TRY(hipMemcpy(rgba, acc_rgba.p, npix * 4, hipMemcpyDeviceToHost));
            return RT_OK;
        });
    if (st != RT_OK && err.empty()) err = rt_last_error();
    // 4. give everything back
    for (Shard &s : shards) {
        if (s.film) (void)rt_film_destroy(s.film);
        try {
            DeviceGuard g(s.device);
            s.d_rad.release();
            s.d_rgba.release();
            if (s.stream) (void)hipStreamDestroy(s.stream);
        } catch (const HipFail &) {
        }
    }
    for (Shard &s : shards)
        if (s.owns_scene && s.scene) (void)rt_scene_destroy(s.scene);
    if (st != RT_OK) return fail(st, "rt_render_multi: " + err);
    if (stats) {
        for (const Shard &s : shards) {
            stats->rays += s.stats.rays;
            stats->paths += s.stats.paths;
            stats->traverse_ms += s.stats.traverse_ms;
            stats->traverse_launches += s.stats.traverse_launches;
        }
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return RT_OK;
}

} // extern "C"
