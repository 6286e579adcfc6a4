// rt_lbvh.hip — the fast walk's hierarchy built ON the GPU (opt-in, RT_FAST_TREE_LBVH).
//
// The reference builds its tree on the host, recursively, recomputing every primitive's
// bounding sphere four times per level (bvh.rs:21-95). This builder is the opposite end:
// a linear BVH in a handful of kernels (Morton codes of the box centres, one radix sort,
// Karras' parallel radix-tree construction, bottom-up box fitting), milliseconds instead of
// seconds, at the price of a tree that needs more steps per ray than the host's
// surface-area-heuristic build. Same output format as rt_fast_tree.hip.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../host/rt_error.hpp"
#include "rt_fast_tree.hpp"

namespace rtd {
namespace {

constexpr int LB = 256;
constexpr uint32_t LEAF_BIT = 0x80000000u; // child index names a primitive of the sorted order, not a node
constexpr uint32_t LBVH_LEAF_MAX = 2;      // subtrees of this many primitives or fewer become one leaf

struct F6 {
    float lo[3], hi[3];
};

void hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return;
    rt_status st = e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY
                   : (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? RT_ERR_NO_DEVICE
                                                                          : RT_ERR_DEVICE;
    throw rt::DeviceFailure(st, std::string("LBVH build, ") + what + ": " + hipGetErrorString(e));
}

template <class T>
struct Dev {
    T *p = nullptr;
    explicit Dev(size_t n) { hip_ok(hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T)), "hipMalloc"); }
    ~Dev() {
        if (p) (void)hipFree(p);
    }
    Dev(const Dev &) = delete;
    Dev &operator=(const Dev &) = delete;
};

__device__ __forceinline__ uint32_t spread10(uint32_t v) { // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void __launch_bounds__(LB) k_morton(const F6 *__restrict__ boxes, uint32_t n, float ox, float oy, float oz,
                                               float sx, float sy, float sz, float cells,
                                               uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t i = blockIdx.x * LB + threadIdx.x;
    if (i >= n) return;
    F6 b = boxes[i];
    float cx = (0.5f * (b.lo[0] + b.hi[0]) - ox) * sx;
    float cy = (0.5f * (b.lo[1] + b.hi[1]) - oy) * sy;
    float cz = (0.5f * (b.lo[2] + b.hi[2]) - oz) * sz;
    uint32_t x = (uint32_t)fminf(fmaxf(cx, 0.0f), cells - 1.0f);
    uint32_t y = (uint32_t)fminf(fmaxf(cy, 0.0f), cells - 1.0f);
    uint32_t z = (uint32_t)fminf(fmaxf(cz, 0.0f), cells - 1.0f);
    keys[i] = (spread10(x) << 2) | (spread10(y) << 1) | spread10(z);
    vals[i] = i;
}

// length of the common prefix of the (key, index) pairs i and j; -1 outside the array
__device__ __forceinline__ int prefix(const uint32_t *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint32_t a = keys[i], b = keys[j];
    if (a != b) return __clz(a ^ b);
    return 32 + __clz((uint32_t)i ^ (uint32_t)j);
}

// Karras, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees" (HPG 2012):
// internal node i of the binary radix tree over n sorted keys, computed independently.
__global__ void __launch_bounds__(LB) k_radix_tree(const uint32_t *__restrict__ keys, int n, uint32_t *__restrict__ left,
                                                   uint32_t *__restrict__ right, uint32_t *__restrict__ first,
                                                   uint32_t *__restrict__ last, uint32_t *__restrict__ parent_of_node,
                                                   uint32_t *__restrict__ parent_of_leaf) {
    int i = blockIdx.x * LB + threadIdx.x;
    if (i >= n - 1) return;
    int d = prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    int dmin = prefix(keys, n, i, i - d);
    int lmax = 2;
    while (prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = prefix(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    int gamma = i + s * d + (d < 0 ? -1 : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    uint32_t lc = (lo == gamma) ? (LEAF_BIT | (uint32_t)gamma) : (uint32_t)gamma;
    uint32_t rc = (hi == gamma + 1) ? (LEAF_BIT | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    left[i] = lc;
    right[i] = rc;
    first[i] = (uint32_t)lo;
    last[i] = (uint32_t)hi;
    if (lc & LEAF_BIT) parent_of_leaf[gamma] = (uint32_t)i; else parent_of_node[gamma] = (uint32_t)i;
    if (rc & LEAF_BIT) parent_of_leaf[gamma + 1] = (uint32_t)i; else parent_of_node[gamma + 1] = (uint32_t)i;
    if (i == 0) parent_of_node[0] = 0xFFFFFFFFu;
}

// Bottom-up: every leaf climbs; the second thread to reach a node owns both children's boxes.
__global__ void __launch_bounds__(LB) k_fit(const F6 *__restrict__ boxes, const uint32_t *__restrict__ vals, int n,
                                            const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                            const uint32_t *__restrict__ parent_of_node,
                                            const uint32_t *__restrict__ parent_of_leaf, uint32_t *arrived,
                                            F6 *node_box) {
    int i = blockIdx.x * LB + threadIdx.x;
    if (i >= n) return;
    uint32_t node = parent_of_leaf[i];
    while (node != 0xFFFFFFFFu) {
        __threadfence();
        if (atomicAdd(&arrived[node], 1u) == 0u) return; // the sibling subtree is not finished yet
        __threadfence();
        uint32_t c[2] = {left[node], right[node]};
        F6 out;
        for (int k = 0; k < 3; k++) {
            out.lo[k] = INFINITY;
            out.hi[k] = -INFINITY;
        }
        for (int s = 0; s < 2; s++) {
            const F6 *src = (c[s] & LEAF_BIT) ? &boxes[vals[c[s] & ~LEAF_BIT]] : &node_box[c[s]];
            // the sibling's box was written by another thread: the fence after the atomic
            // (acquire, device scope) has dropped this CU's stale L1 lines
            for (int k = 0; k < 3; k++) {
                out.lo[k] = fminf(out.lo[k], src->lo[k]);
                out.hi[k] = fmaxf(out.hi[k], src->hi[k]);
            }
        }
        node_box[node] = out;
        node = parent_of_node[node];
    }
}

__device__ __forceinline__ uint32_t leaf_ref_dev(uint32_t begin, uint32_t count) {
    return REF_LEAF | ((count - 1) << REF_CNT_SHIFT) | begin;
}

// One FNode per radix-tree node; subtrees of <= LBVH_LEAF_MAX primitives collapse into leaf references
// (their own records stay unused). Also the depth of every record that is in use.
__global__ void __launch_bounds__(LB) k_emit(const F6 *__restrict__ boxes, const uint32_t *__restrict__ vals, int n,
                                             const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                             const uint32_t *__restrict__ first, const uint32_t *__restrict__ last,
                                             const uint32_t *__restrict__ parent_of_node, const F6 *__restrict__ node_box,
                                             FNode *__restrict__ out, uint32_t *__restrict__ max_depth) {
    int i = blockIdx.x * LB + threadIdx.x;
    if (i >= n - 1) return;
    FNode f;
    memset(&f, 0, sizeof f);
    uint32_t c[2] = {left[i], right[i]};
    for (int s = 0; s < 2; s++) {
        uint32_t ref;
        const F6 *b;
        if (c[s] & LEAF_BIT) {
            uint32_t k = c[s] & ~LEAF_BIT;
            ref = leaf_ref_dev(k, 1);
            b = &boxes[vals[k]];
        } else {
            uint32_t cnt = last[c[s]] - first[c[s]] + 1;
            ref = cnt <= LBVH_LEAF_MAX ? leaf_ref_dev(first[c[s]], cnt) : c[s];
            b = &node_box[c[s]];
        }
        float *lo = s == 0 ? f.lo0 : f.lo1, *hi = s == 0 ? f.hi0 : f.hi1;
        for (int k = 0; k < 3; k++) {
            lo[k] = b->lo[k];
            hi[k] = b->hi[k];
        }
        if (s == 0) f.ref0 = ref; else f.ref1 = ref;
    }
    out[i] = f;
    if (last[i] - first[i] + 1 > LBVH_LEAF_MAX) { // in use: count the records from here to the root
        uint32_t depth = 1, p = parent_of_node[i];
        while (p != 0xFFFFFFFFu) {
            depth++;
            p = parent_of_node[p];
        }
        atomicMax(max_depth, depth);
    }
}

} // namespace

FastTree build_fast_tree_lbvh(const std::vector<BoxD> &slot_box, double pad, int device,
                              uint32_t max_depth_wanted) {
    FastTree ft;
    const uint32_t n = (uint32_t)slot_box.size();
    ft.ref_of_fast.resize(n);
    auto dummy = [&]() {
        ft.fnodes.emplace_back();
        std::memset(&ft.fnodes[0], 0, sizeof(FNode));
    };
    if (n <= LBVH_LEAF_MAX) {
        for (uint32_t s = 0; s < n; s++) ft.ref_of_fast[s] = s;
        if (n) ft.root_ref = REF_LEAF | ((n - 1) << REF_CNT_SHIFT);
        dummy();
        return ft;
    }
    // conservative f32 boxes and the scene's bounds, on the host (one pass over data already there)
    std::vector<F6> boxes(n);
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t s = 0; s < n; s++) {
        for (int k = 0; k < 3; k++) {
            boxes[s].lo[k] = std::nextafterf((float)(slot_box[s].lo[k] - pad), -INFINITY);
            boxes[s].hi[k] = std::nextafterf((float)(slot_box[s].hi[k] + pad), INFINITY);
            float c = 0.5f * (boxes[s].lo[k] + boxes[s].hi[k]);
            blo[k] = std::min(blo[k], c);
            bhi[k] = std::max(bhi[k], c);
        }
    }

    int prev = 0;
    hip_ok(hipGetDevice(&prev), "hipGetDevice");
    hip_ok(hipSetDevice(device), "hipSetDevice");
    struct Restore {
        int d;
        ~Restore() { (void)hipSetDevice(d); }
    } restore{prev};

    Dev<F6> d_boxes(n), d_node_box(n);
    Dev<uint32_t> k_in(n), k_out(n), v_in(n), v_out(n), left(n), right(n), first(n), last(n), pnode(n), pleaf(n),
        arrived(n), d_depth(1);
    Dev<FNode> d_fnodes(n);
    hip_ok(hipMemcpy(d_boxes.p, boxes.data(), sizeof(F6) * n, hipMemcpyHostToDevice), "upload boxes");
    const unsigned grid = (n + LB - 1) / LB;
    // A radix tree is as deep as the keys are long, plus the index bits that separate primitives
    // sharing a key. Start with 10 bits per axis; if the tree is deeper than max_depth_wanted,
    // build again with shorter keys (each attempt is a few milliseconds).
    uint32_t depth = 0;
    for (int bits = 10; bits >= 6; bits--) {
        const float cells = (float)(1 << bits);
        float scale[3];
        for (int k = 0; k < 3; k++) {
            float w = bhi[k] - blo[k];
            scale[k] = (w > 0.f && std::isfinite(w)) ? cells * (1.0f - 1e-6f) / w : 0.f;
        }
        hip_ok(hipMemset(arrived.p, 0, sizeof(uint32_t) * n), "memset");
        hip_ok(hipMemset(d_depth.p, 0, sizeof(uint32_t)), "memset");
        hipLaunchKernelGGL(k_morton, dim3(grid), dim3(LB), 0, 0, d_boxes.p, n, blo[0], blo[1], blo[2], scale[0],
                           scale[1], scale[2], cells, k_in.p, v_in.p);
        size_t temp_bytes = 0;
        hip_ok(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, k_in.p, k_out.p, v_in.p, v_out.p, (int)n, 0,
                                                  3 * bits, 0),
               "radix sort (size query)");
        Dev<unsigned char> temp(temp_bytes);
        hip_ok(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, k_in.p, k_out.p, v_in.p, v_out.p, (int)n, 0,
                                                  3 * bits, 0),
               "radix sort");
        hipLaunchKernelGGL(k_radix_tree, dim3(grid), dim3(LB), 0, 0, k_out.p, (int)n, left.p, right.p, first.p, last.p,
                           pnode.p, pleaf.p);
        hipLaunchKernelGGL(k_fit, dim3(grid), dim3(LB), 0, 0, d_boxes.p, v_out.p, (int)n, left.p, right.p, pnode.p,
                           pleaf.p, arrived.p, d_node_box.p);
        hipLaunchKernelGGL(k_emit, dim3(grid), dim3(LB), 0, 0, d_boxes.p, v_out.p, (int)n, left.p, right.p, first.p,
                           last.p, pnode.p, d_node_box.p, d_fnodes.p, d_depth.p);
        hip_ok(hipGetLastError(), "LBVH kernels");
        hip_ok(hipMemcpy(&depth, d_depth.p, sizeof depth, hipMemcpyDeviceToHost), "download depth");
        if (depth <= max_depth_wanted) break;
    }
    ft.fnodes.resize(n - 1);
    hip_ok(hipMemcpy(ft.fnodes.data(), d_fnodes.p, sizeof(FNode) * (n - 1), hipMemcpyDeviceToHost), "download nodes");
    hip_ok(hipMemcpy(ft.ref_of_fast.data(), v_out.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost), "download order");
    ft.max_depth = depth ? depth : 1;
    ft.root_ref = 0;
    ft.top_nodes = 0;
    return ft;
}

} // namespace rtd
