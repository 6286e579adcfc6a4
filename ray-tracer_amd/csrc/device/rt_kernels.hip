// rt_kernels.hip — CDNA4 (gfx950) kernels of the wavefront path tracer.
//
// The path tracer's fast-walk kernel lives in rt_path_walk.hip (compiled with another instruction-scheduling
// strategy); everything the two files share is in rt_walk_device.hpp.
#include "rt_walk2_device.hpp"
#include "rt_walk3_device.hpp"

namespace rtd {
namespace {

// ============================================================================
// kernels
// ============================================================================

// RT_TRAVERSE_REFERENCE_ORDER: one ray per lane, stackless walk of the f64 tree.
__global__ void __launch_bounds__(BLOCK) k_intersect_ref(Scene sc, BatchIO io, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    V3 o, dir;
    io.load(i, o, dir);
    const Closest best = traverse_reference_order(sc, o, dir);
#ifdef RT_COUNT_VISITS
    if (best.have) screen_margin_use(sc, best.slot, make_fray(sc, o, dir)); // how close the f32 screen came to losing it
#endif
    io.store(sc, i, best);
}

// RT_TRAVERSE_FAST: every wavefront works through ranges of rays_per_wave consecutive rays.
__global__ void RT_FAST_BOUNDS k_intersect_fast(Scene sc, BatchIO io, uint64_t n, uint32_t rays_per_wave,
                                                unsigned long long *ticket) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    uint64_t wave = (uint64_t)blockIdx.x * (TBLOCK / 64) + (threadIdx.x >> 6);
    RaySupply supply;
    supply.init(wave, (uint64_t)gridDim.x * (TBLOCK / 64), rays_per_wave, n, ticket);
    FastLds l = fast_lds(sc, lds_stack);
    traverse_fast(sc, io, supply, l.sref, l.st, l.pos, sc.vote_exact, sc.vote_leaf, sc.vote_refill);
}

// The same with two rays per lane (scenes whose references fit the compact form; rt_walk2_device.hpp).
__global__ void RT_FAST2_BOUNDS k_intersect_fast2(Scene sc, BatchIO io, uint64_t n, uint32_t rays_per_wave,
                                                  unsigned long long *ticket) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    uint64_t wave = (uint64_t)blockIdx.x * (TBLOCK / 64) + (threadIdx.x >> 6);
    RaySupply supply;
    supply.init(wave, (uint64_t)gridDim.x * (TBLOCK / 64), rays_per_wave, n, ticket);
    traverse_fast2(sc, io, supply, lds_stack, sc.vote2_exact, sc.vote2_leaf, sc.vote2_refill);
}

// The same with a parked next ray per lane (rt_walk3_device.hpp).
__global__ void RT_FAST_BOUNDS k_intersect_fast3(Scene sc, BatchIO io, uint64_t n, uint32_t rays_per_wave,
                                                 unsigned long long *ticket) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    // readfirstlane: the compiler then knows that the ray supply, and with it every exit of the walk's loop, is the same in all lanes
    uint64_t wave = (uint64_t)blockIdx.x * (TBLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    RaySupply supply;
    supply.init(wave, (uint64_t)gridDim.x * (TBLOCK / 64), rays_per_wave, n, ticket);
    traverse_fast3(sc, io, supply, lds_stack, sc.vote_exact, sc.vote_leaf, sc.vote3_switch, sc.vote3_prep, sc.vote3_stuck);
}

// Second pass of RT_TRAVERSE_FAST in scenes with spheres: the rays the fast walk marked.
// Marked rays are rare (one in 300 in the 11.9 M primitive scene) and each costs a long chain of
// dependent loads (about 170 us there), so what matters is how many of those chains a wavefront runs
// one after the other. Every wavefront therefore scans a range of REWALK_RANGE consecutive rays,
// collects the marked ones in an LDS queue and answers them 64 at a time, one per lane, side by side.
// (A grid-stride loop that answers each marked ray where it finds it runs six chains in a row per
// wavefront on that scene: 5.4 ms for 16.7 M rays.)
#if defined(RT_REWALK_RANGE)
constexpr uint32_t REWALK_RANGE = RT_REWALK_RANGE; // experiment
#else
constexpr uint32_t REWALK_RANGE = 4096; // rays per wavefront; a multiple of 64
#endif
template <class IO>
__device__ __forceinline__ void rewalk_marked(const Scene &sc, const IO &io, uint64_t n, uint32_t *queue) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *q = queue + (threadIdx.x >> 6) * 128u; // 128 entries per wavefront: < 64 waiting + <= 64 new
    const uint64_t n_ranges = (n + REWALK_RANGE - 1) / REWALK_RANGE;
    const uint64_t n_waves = (uint64_t)gridDim.x * (BLOCK / 64);
    for (uint64_t range = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); range < n_ranges; range += n_waves) {
        const uint64_t begin = range * REWALK_RANGE;
        const uint64_t end = begin + REWALK_RANGE < n ? begin + REWALK_RANGE : n;
        uint32_t queued = 0; // the same in every lane
        for (uint64_t base = begin; base < end || queued != 0; base += 64) {
            const uint64_t t = base + lane;
            const bool marked = base < end && t < end && io.flagged(t);
            const unsigned long long mask = lane_mask(marked);
            if (marked) q[queued + lane_rank(mask)] = (uint32_t)(t - begin);
            queued += (uint32_t)__popcll(mask);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (queued >= 64u || (base + 64 >= end && queued != 0u)) {
                const uint32_t take = queued < 64u ? queued : 64u;
                const uint32_t mine = lane < take ? q[lane] : 0u;
                const uint32_t moved = lane + 64u < queued ? q[lane + 64u] : 0u;
                __builtin_amdgcn_wave_barrier();
                if (lane + 64u < queued) q[lane] = moved; // what is left moves to the front
                queued -= take;
                if (lane < take) {
                    const uint64_t i = begin + mine;
                    V3 o, dir;
                    io.load(i, o, dir);
                    io.store(sc, i, reference_answer(sc, o, dir));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

__global__ void __launch_bounds__(BLOCK) k_rewalk_batch(Scene sc, BatchIO io, uint64_t n) {
    __shared__ uint32_t queue[(BLOCK / 64) * 128];
    if (io.marked && *io.marked == 0u) return; // the walk marked no ray
    rewalk_marked(sc, io, n, queue);
}

// Closest hit by testing EVERY primitive (no hierarchy): the independent check of both walks.
// Of two equidistant hits the one with the smaller primitive id wins (list order, strict '<').
__global__ void __launch_bounds__(BLOCK) k_intersect_brute(Scene sc, const double *__restrict__ rays, uint64_t n,
                                                            uint32_t *__restrict__ prim_id,
                                                            double *__restrict__ distance) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double *r = rays + 6 * i;
    V3 o = {r[0], r[1], r[2]}, dir = {r[3], r[4], r[5]};
    bool have = false;
    uint32_t best_id = RT_MISS;
    double best = 0.0;
    for (uint32_t slot = 0; slot < sc.n_slots; slot++) { // every lane reads the same record
        PrimHot h = sc.hot[slot];
        V3 p;
        if (!prim_intersect(sc, slot, h, o, dir, p)) continue;
        double d = mag(sub(p, o));
        if (!have || d < best || (d == best && h.prim_id < best_id)) {
            have = true;
            best = d;
            best_id = h.prim_id;
        }
    }
    prim_id[i] = best_id;
    distance[i] = best;
}

__global__ void __launch_bounds__(BLOCK) k_intersect_prims(Scene sc, const double *__restrict__ rays,
                                                            const uint32_t *__restrict__ slots, uint64_t n,
                                                            uint8_t *__restrict__ hit, double *__restrict__ pos,
                                                            double *__restrict__ normal) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double *r = rays + 6 * i;
    V3 o = {r[0], r[1], r[2]}, dir = {r[3], r[4], r[5]};
    uint32_t slot = slots[i];
    PrimHot h = sc.hot[slot];
    V3 p = {0.0, 0.0, 0.0}, nr = {0.0, 0.0, 0.0};
    bool ok = prim_intersect(sc, slot, h, o, dir, p);
    if (ok) nr = normal_at(h, p);
    hit[i] = ok ? 1 : 0;
    if (pos) {
        pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z;
    }
    if (normal) {
        normal[3 * i] = nr.x; normal[3 * i + 1] = nr.y; normal[3 * i + 2] = nr.z;
    }
}

// camera.rs:16-78 for one (pixel, sub-ray); width x height at the reference's pixel pitch
struct CamConst {
    double loc[3], dir[9], focal_length, sensor_to_lens, aperture_radius, sensor_width, sensor_height;
    double lens_center[3];
};

__device__ __forceinline__ V3 mat_mult_3(const double *A, const V3 &x) { // geometry.rs:68-74
    return {A[0] * x.x + A[1] * x.y + A[2] * x.z, A[3] * x.x + A[4] * x.y + A[5] * x.z,
            A[6] * x.x + A[7] * x.y + A[8] * x.z};
}

__global__ void __launch_bounds__(BLOCK) k_raygen(CamConst cam, uint32_t width, uint32_t height, uint64_t seed,
                                                   const uint32_t *__restrict__ owned, uint32_t n_pixels,
                                                   double *__restrict__ bundle) {
    uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (uint64_t)n_pixels * 5) return;
    uint32_t local = (uint32_t)(t / 5), j = (uint32_t)(t % 5);
    uint32_t pixel = owned ? owned[local] : local;
    uint32_t col = pixel / width, row = pixel % width; // reference's (col,row) = image (y,x)
    const double third2 = 2.0 / 3.0;
    double offset_x = (j == 0) ? 0.0 : ((j == 1 || j == 2) ? -third2 : third2);
    double offset_y = (j == 0) ? 0.0 : ((j == 1 || j == 4) ? -third2 : third2);
    double ux = -(2.0 * ((double)col + offset_x) / (double)height - 1.0);
    double uy = (2.0 * ((double)row + offset_y) / (double)width - 1.0);
    V3 raw = {-ux * cam.sensor_height / 2.0, -uy * cam.sensor_width / 2.0, 0.0};
    V3 loc = {cam.loc[0], cam.loc[1], cam.loc[2]};
    V3 lens_center = {cam.lens_center[0], cam.lens_center[1], cam.lens_center[2]};
    V3 sensor_loc = add(loc, mat_mult_3(cam.dir, raw));
    V3 naive = normalize(sub(lens_center, sensor_loc));
    V3 image_loc = add(lens_center, scale(cam.focal_length, naive));
    double u0, u1;
    uniform2(seed, pixel, 0u, j | (STREAM_LENS << 8), 0u, u0, u1);
    // geometry.rs:56-65
    double r = cam.aperture_radius * sqrt(u0);
    double theta = PI * 2.0 * u1;
    V3 lens_raw = {r * cos(theta), r * sin(theta), cam.sensor_to_lens};
    V3 lens_pt = add(loc, mat_mult_3(cam.dir, lens_raw));
    V3 rd = normalize(sub(image_loc, lens_pt));
    double *out = bundle + t * 6;
    out[0] = lens_pt.x; out[1] = lens_pt.y; out[2] = lens_pt.z;
    out[3] = rd.x; out[4] = rd.y; out[5] = rd.z;
}

// ---- camera.rs:80-90 ---------------------------------------------------------
// The reference computes E.powf(-2.5*x) with the host's libm, whose pow is correctly rounded in all
// but about one case in 2^15. The device library's pow is not (1-2 ulp), and floor(255*f) turns one ulp
// into a whole byte wherever 255*f sits next to an integer. So the power is evaluated here in
// double-double arithmetic (about 100 bits) and rounded once: the correctly rounded pow(E, y), where
// E is the DOUBLE nearest to e, i.e. exp(y * ln E) with ln E = 1 - 5.318...e-17.
struct DD {
    double hi, lo;
};
__device__ __forceinline__ DD two_sum(double a, double b) {
    double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ DD quick_two_sum(double a, double b) { // |a| >= |b|
    double s = a + b;
    return {s, b - (s - a)};
}
__device__ __forceinline__ DD two_prod(double a, double b) {
    double p = a * b;
    return {p, __builtin_fma(a, b, -p)};
}
__device__ __forceinline__ DD dd_add(DD a, DD b) {
    DD s = two_sum(a.hi, b.hi), t = two_sum(a.lo, b.lo);
    s = quick_two_sum(s.hi, s.lo + t.hi);
    return quick_two_sum(s.hi, s.lo + t.lo);
}
__device__ __forceinline__ DD dd_mul(DD a, DD b) {
    DD p = two_prod(a.hi, b.hi);
    return quick_two_sum(p.hi, p.lo + (a.hi * b.lo + a.lo * b.hi));
}
__device__ __forceinline__ DD dd_mul_d(DD a, double b) {
    DD p = two_prod(a.hi, b);
    return quick_two_sum(p.hi, p.lo + a.lo * b);
}
__device__ __forceinline__ DD dd_div_small(DD a, double n) { // n a small integer
    double q1 = a.hi / n;
    DD p = two_prod(q1, n);
    double r = ((a.hi - p.hi) - p.lo) + a.lo;
    return quick_two_sum(q1, r / n);
}

// pow(2.718281828459045 (the double), y), correctly rounded
__device__ double pow_e_double(double y) {
    if (!(y == y)) return y;
    if (y > 709.8) return __builtin_inf();
    if (y < -745.2) return 0.0;
    const double LN_E_MINUS_1 = -0x1.ea8556644e4cdp-55; // ln(double(e)) - 1
    const DD LN2 = {0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56};
    DD z = two_sum(y, y * LN_E_MINUS_1); // y * ln E
    double k = rint(z.hi * 0x1.71547652b82fep+0);
    DD r = dd_add(z, dd_mul_d(LN2, -k)); // |r| <= 0.35
    r.hi *= 1.0 / 256.0;                 // exact
    r.lo *= 1.0 / 256.0;
    // expm1(r) by Horner, degree 10: r * (1 + r/2 * (1 + r/3 * (... (1 + r/10))))
    DD t = {1.0, 0.0};
    for (int n = 10; n >= 2; n--) t = dd_add(DD{1.0, 0.0}, dd_mul(dd_div_small(r, (double)n), t));
    DD s = dd_mul(r, t);
    for (int i = 0; i < 8; i++) s = dd_add(dd_mul_d(s, 2.0), dd_mul(s, s)); // (1+s)^2 - 1 = 2s + s^2
    DD e = dd_add(DD{1.0, 0.0}, s);
    return ldexp(e.hi, (int)k); // e.hi is hi+lo rounded once; the scaling is exact above the subnormals
}

__device__ __forceinline__ uint8_t tone_channel(double x) {
    double f = 2.0 / (1.0 + pow_e_double(-2.5 * x)) - 1.0;
    double v = floor(f * 255.0);
    if (!(v == v) || v <= 0.0) return 0; // Rust `as u8`: NaN -> 0, saturating
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

__global__ void __launch_bounds__(BLOCK) k_tone_map(const double *__restrict__ radiance, uint64_t n,
                                                     uint8_t *__restrict__ rgba) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uchar4 c;
    c.x = tone_channel(radiance[3 * i]);
    c.y = tone_channel(radiance[3 * i + 1]);
    c.z = tone_channel(radiance[3 * i + 2]);
    c.w = 255;
    reinterpret_cast<uchar4 *>(rgba)[i] = c;
}

// ---- wavefront integrator ----------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_paths_from_rays(Paths p, const double *__restrict__ rays, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) p.counters[0] = (uint32_t)n;
    if (i >= n) return;
    const double *r = rays + 6 * i;
#pragma unroll
    for (int k = 0; k < 6; k++) p.rays[(size_t)k * p.capacity + i] = r[k];
    p.ids[i] = (uint32_t)i;
    p.length[i] = 0;
    p.emissive[i] = 0;
}

// path index = (sample_local * n_owned + pixel_local) * 5 + sub_ray
__global__ void __launch_bounds__(BLOCK) k_paths_from_film(Paths p, FilmView f, uint32_t n_samples) {
    uint64_t n = (uint64_t)f.n_owned * 5 * n_samples;
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) p.counters[0] = (uint32_t)n;
    if (i >= n) return;
    // the rays themselves stay in the film's bundle (Paths::first_bundle)
    p.length[i] = 0;
    p.emissive[i] = 0;
}

__global__ void __launch_bounds__(BLOCK) k_traverse_paths_ref(Scene sc, Paths p, uint32_t depth) {
    uint64_t n = p.counters[depth];
    PathIO io = path_io(p, depth);
    for (uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; t < n; t += (uint64_t)gridDim.x * BLOCK) {
        V3 o, dir;
        io.load(t, o, dir);
        io.store(sc, t, traverse_reference_order(sc, o, dir));
    }
}


__global__ void __launch_bounds__(BLOCK) k_rewalk_paths(Scene sc, Paths p, uint32_t depth) {
    __shared__ uint32_t queue[(BLOCK / 64) * 128];
    rewalk_marked(sc, path_io(p, depth), p.counters[depth], queue);
}

// tracer.rs:37-52 for every live path, then compaction of the survivors into the other
// ray buffer. All survivors of a 512-thread block take their slots from ONE atomic on
// the live counter (ballot -> rank inside the wave, wave totals summed through LDS): a
// single global word takes only about 90 atomics per microsecond, so one atomic per
// wavefront would cost more than the shading itself.
// paths per thread of the diffuse shading kernel; measured (k_shade per launch of 105 M paths): 1 -> 2.64 ms,
// 2 -> 2.48 ms, 4 -> 3.03 ms
constexpr int SHADE_PT = 2;
constexpr int SHADE_BLOCK = 512; // measured: 256 -> 3.55, 512 -> 3.74, 1024 -> 3.67 G queries/s in the 1080p render
// GLOSS = false leaves the glossy branch (f64 log/sin/cos, many registers) out of the kernel that
// scenes with diffuse materials only (everything brdf.rs can express) run.
// PT: paths per thread. A thread's PT paths lie SHADE_BLOCK apart (a block works on a tile of PT x SHADE_BLOCK
// consecutive paths); their first loads are requested together, so that more of the dependent reads of a path
// (slot -> primitive record -> material) are in flight per resident wavefront.
template <bool GLOSS, int PT>
__global__ void __launch_bounds__(SHADE_BLOCK) k_shade(Scene sc, Paths p, uint32_t depth, uint64_t seed,
                                                        const uint32_t *__restrict__ keys, FilmView f,
                                                        uint32_t first_sample, int from_film) {
    constexpr int WAVES = SHADE_BLOCK / 64;
    __shared__ uint32_t wave_total[PT * WAVES];
    __shared__ uint32_t block_base;
    const uint64_t n = p.counters[depth];
    const uint64_t cap = p.capacity;
    const uint32_t *id_cur = p.ids + (size_t)(depth & 1) * cap;
    uint32_t *id_nxt = p.ids + (size_t)((depth & 1) ^ 1) * cap;
    double *ray_nxt = p.rays + (size_t)((depth & 1) ^ 1) * 6 * cap;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool from_bundle = depth == 0 && p.first_bundle != nullptr;
    // whole blocks step together so that the barriers below are reached by every thread
    for (uint64_t base = (uint64_t)blockIdx.x * (SHADE_BLOCK * PT); base < n; base += (uint64_t)gridDim.x * (SHADE_BLOCK * PT)) {
        bool alive[PT];
        uint32_t i[PT], slot[PT], rank[PT];
        V3 pos[PT], nd[PT];
        // Path index and hit position are requested before the slot says whether the path goes on (most
        // do): one trip to memory less. Path state streams through once per bounce: keep it out of the
        // caches the walk uses.
#pragma unroll
        for (int k = 0; k < PT; k++) {
            const uint64_t t = base + (uint64_t)k * SHADE_BLOCK + threadIdx.x;
            alive[k] = false;
            i[k] = 0;
            slot[k] = RT_MISS;
            pos[k] = V3{0, 0, 0};
            nd[k] = V3{0, 0, 0};
            if (t < n) {
                slot[k] = p.hit_mat[t]; // the slot of the primitive that was hit (PathIO::store)
                i[k] = from_bundle ? (uint32_t)t : id_cur[t];
                pos[k] = V3{__builtin_nontemporal_load(p.hit + t), __builtin_nontemporal_load(p.hit + cap + t),
                            __builtin_nontemporal_load(p.hit + 2 * cap + t)};
            }
        }
#pragma unroll
        for (int k = 0; k < PT; k++) {
            const uint64_t t = base + (uint64_t)k * SHADE_BLOCK + threadIdx.x;
            if (slot[k] != RT_MISS) {
                const PrimHot &h = sc.hot[slot[k]];
                const uint32_t mat = h.material;
                const V3 nrm = normal_at(h, pos[k]); // bvh.rs:117
                if (sc.mats[mat].emissivity > 0.0) p.emissive[i[k]] = 1;
                uint32_t pixel, sample, j;
                if (from_film) {
                    // path indices are 32-bit, so is n_owned * 5 (a pass never holds more paths): 32-bit division
                    const uint32_t per_sample = f.n_owned * 5u;
                    const uint32_t s_local = i[k] / per_sample, within = i[k] - s_local * per_sample;
                    pixel = f.owned[within / 5u];
                    j = within % 5u;
                    sample = first_sample + s_local;
                } else {
                    pixel = keys[3 * (size_t)i[k]]; sample = keys[3 * (size_t)i[k] + 1]; j = keys[3 * (size_t)i[k] + 2];
                }
                const double gloss = sc.mats[mat].gloss;
                if (GLOSS && gloss > 0.0) { // the JavaScript tracer's lobe; brdf.rs has only the diffuse branch
                    V3 in_dir;
                    if (from_bundle) {
                        const double *r = p.first_bundle + 6 * (size_t)((uint32_t)t % p.first_period);
                        in_dir = V3{r[3], r[4], r[5]};
                    } else {
                        const double *ray_cur = p.rays + (size_t)(depth & 1) * 6 * cap;
                        in_dir = V3{ray_cur[3 * cap + t], ray_cur[4 * cap + t], ray_cur[5 * cap + t]};
                    }
                    nd[k] = glossy_sample(gloss, in_dir, nrm, seed, pixel, sample, j, depth);
                } else {
                    double u0, u1;
                    uniform2(seed, pixel, sample, j | (STREAM_BSDF << 8), depth, u0, u1);
                    nd[k] = brdf_sample(nrm, u0, u1);
                }
                __builtin_nontemporal_store(dot(nd[k], nrm), p.cosine + (size_t)depth * cap + i[k]);
                __builtin_nontemporal_store(mat, p.material + (size_t)depth * cap + i[k]);
                p.length[i[k]] = (uint8_t)(depth + 1);
                alive[k] = true;
            }
        }
        // survivors keep their order: the k-th stretch of the tile comes after the (k-1)-th
#pragma unroll
        for (int k = 0; k < PT; k++) {
            const unsigned long long mask = lane_mask(alive[k]);
            rank[k] = lane_rank(mask);
            if (lane == 0) wave_total[k * WAVES + wave] = __popcll(mask);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sum = 0;
            for (int w = 0; w < PT * WAVES; w++) {
                uint32_t c = wave_total[w];
                wave_total[w] = sum; // exclusive prefix
                sum += c;
            }
            block_base = sum ? atomicAdd(&p.counters[depth + 1], sum) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PT; k++)
            if (alive[k]) {
                uint64_t w = (uint64_t)block_base + wave_total[k * WAVES + wave] + rank[k];
                __builtin_nontemporal_store(pos[k].x, ray_nxt + w); // tracer.rs:44
                __builtin_nontemporal_store(pos[k].y, ray_nxt + cap + w);
                __builtin_nontemporal_store(pos[k].z, ray_nxt + 2 * cap + w);
                __builtin_nontemporal_store(nd[k].x, ray_nxt + 3 * cap + w); // tracer.rs:42
                __builtin_nontemporal_store(nd[k].y, ray_nxt + 4 * cap + w);
                __builtin_nontemporal_store(nd[k].z, ray_nxt + 5 * cap + w);
                __builtin_nontemporal_store(i[k], id_nxt + w);
            }
        __syncthreads(); // wave_total / block_base are rewritten by the next round
    }
}

// tracer.rs:57-80
__device__ __forceinline__ V3 fold_path(const Scene &sc, const Paths &p, uint64_t i) {
    V3 sr = {0.0, 0.0, 0.0};
    const uint8_t lit = p.emissive[i];
    int len = p.length[i]; // requested together with the flag: one trip to memory instead of two
    if (!lit) return sr;
    while (len >= 1) {
        len -= 1;
        const rt_material &m = sc.mats[p.material[(size_t)len * p.capacity + i]];
        sr.x *= m.reflection_spectrum[0];
        sr.y *= m.reflection_spectrum[1];
        sr.z *= m.reflection_spectrum[2];
        sr = scale(p.cosine[(size_t)len * p.capacity + i], sr);
        sr = add(sr, scale(m.emissivity, V3{m.reflection_spectrum[0], m.reflection_spectrum[1],
                                            m.reflection_spectrum[2]}));
    }
    return sr;
}

__global__ void __launch_bounds__(BLOCK) k_fold_paths(Scene sc, Paths p, uint64_t n, double *__restrict__ radiance) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    V3 sr = fold_path(sc, p, i);
    radiance[3 * i] = sr.x; radiance[3 * i + 1] = sr.y; radiance[3 * i + 2] = sr.z;
}

// main.rs:58-67: total += trace_ray(...) in (sample, sub-ray) order.
// The folds are independent, only the additions are ordered: one lane per (pixel, sub-ray) folds its
// path of the sample in hand (neighbouring lanes hold neighbouring paths, so the path state is read in
// whole lines and many folds are in flight), and the first lane of each pixel adds the five values in
// sub-ray order, fetched from its neighbours. A wavefront holds FILM_WAVE_PIXELS pixels (60 of its 64 lanes)
// and needs no barrier, so wavefronts whose paths are short do not wait for the others.
constexpr uint32_t FILM_WAVE_PIXELS = 12;
__global__ void __launch_bounds__(BLOCK) k_accumulate_film(Scene sc, Paths p, FilmView f, uint32_t n_samples) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const uint32_t q = lane / 5u, j = lane - 5u * q;
    const uint32_t px = wave * FILM_WAVE_PIXELS + q;
    const bool folds = q < FILM_WAVE_PIXELS && px < f.n_owned;
    const bool sums = folds && j == 0;
    const uint64_t per_sample = (uint64_t)f.n_owned * 5;
    const uint64_t within = (uint64_t)px * 5 + j;
    V3 total = {0.0, 0.0, 0.0};
    if (sums) total = V3{f.total[0][px], f.total[1][px], f.total[2][px]};
    for (uint32_t s = 0; s < n_samples; s++) {
        V3 sr = {0.0, 0.0, 0.0};
        if (folds) sr = fold_path(sc, p, (uint64_t)s * per_sample + within);
        total = add(total, sr); // sub-ray 0 in the lane that sums; the other lanes' totals are not used
#pragma unroll
        for (int d = 1; d < 5; d++)
            total = add(total, V3{__shfl_down(sr.x, d), __shfl_down(sr.y, d), __shfl_down(sr.z, d)});
    }
    if (sums) {
        f.total[0][px] = total.x; f.total[1][px] = total.y; f.total[2][px] = total.z;
    }
}

// main.rs:68-74
__global__ void __launch_bounds__(BLOCK) k_resolve(FilmView f, uint32_t spp, double *__restrict__ radiance,
                                                    uint8_t *__restrict__ rgba) {
    uint32_t px = blockIdx.x * BLOCK + threadIdx.x;
    if (px >= f.n_owned) return;
    double s = 1.0 / (5.0 * (double)spp);
    V3 mean = scale(s, V3{f.total[0][px], f.total[1][px], f.total[2][px]});
    size_t g = f.owned[px];
    if (radiance) {
        radiance[3 * g] = mean.x; radiance[3 * g + 1] = mean.y; radiance[3 * g + 2] = mean.z;
    }
    if (rgba) {
        uchar4 c;
        c.x = tone_channel(mean.x); c.y = tone_channel(mean.y); c.z = tone_channel(mean.z); c.w = 255;
        reinterpret_cast<uchar4 *>(rgba)[g] = c;
    }
}

// The combine of a multi-device render (main.rs:87-100): acc += w * src over a frame of doubles, and the
// byte-wise sum of two RGBA8 frames whose non-zero pixels are disjoint.
__global__ void __launch_bounds__(BLOCK) k_frame_axpy(double *__restrict__ acc, const double *__restrict__ src, double w,
                                                       uint64_t n, int first) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double v = w == 1.0 ? src[i] : w * src[i];
    acc[i] = first ? v : acc[i] + v;
}
__global__ void __launch_bounds__(BLOCK) k_bytes_add(uint32_t *__restrict__ acc, const uint32_t *__restrict__ src, uint64_t n_words,
                                                      int first) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_words) return;
    // disjoint non-zero pixels: no byte carries into its neighbour
    acc[i] = first ? src[i] : acc[i] + src[i];
}

// one wavefront per REWALK_RANGE rays, up to a grid that every device keeps resident many times over
inline unsigned rewalk_blocks(uint64_t n) {
    static const unsigned cap = [] {
        const char *e = getenv("RT_TUNE_REWALK_BLOCKS"); // experiment hook
        return e && atol(e) > 0 ? (unsigned)atol(e) : 2048u;
    }();
    const uint64_t ranges = (n + REWALK_RANGE - 1) / REWALK_RANGE;
    const uint64_t blocks = (ranges + BLOCK / 64 - 1) / (BLOCK / 64);
    return (unsigned)std::min<uint64_t>(blocks ? blocks : 1, cap);
}

} // namespace


// ============================================================================
// launchers
// ============================================================================
// `ticket` (fast mode; may be NULL): two device words of the caller's that no other launch in flight uses (the
// counter the ranges of rays are handed out with, and the "some ray was marked" word). They are zeroed on `stream`
// in front of the kernel.
void launch_intersect(const Scene &sc, int mode, const double *rays, uint64_t n, uint32_t *prim_id,
                      double *distance, double *pos, double *normal, unsigned long long *ticket, hipStream_t stream) {
    if (n == 0) return;
    // the word behind the counter: "a ray of this launch was marked for the re-walk"
    BatchIO io = {rays, prim_id, distance, pos, normal, ticket ? reinterpret_cast<uint32_t *>(ticket + 1) : nullptr};
    if (mode == RT_TRAVERSE_REFERENCE_ORDER) {
        hipLaunchKernelGGL(k_intersect_ref, dim3(blocks_for(n)), dim3(BLOCK), 0, stream, sc, io, n);
    } else {
        if (ticket) (void)hipMemsetAsync(ticket, 0, 2 * sizeof *ticket, stream);
        if ((sc.walk3 & 1u) && n < (1ull << 32)) {
            allow_large_lds(k_intersect_fast3);
            uint64_t waves = resident_waves(k_intersect_fast3, stack3_bytes(sc));
            uint32_t rpw = rays_per_wave_for(n, waves);
            hipLaunchKernelGGL(k_intersect_fast3, dim3(blocks_for_waves(n, rpw, waves)), dim3(TBLOCK), stack3_bytes(sc),
                               stream, sc, io, n, rpw, ticket);
        } else if ((sc.two_rays & 1u) && ticket && n < (1ull << 32)) {
            uint64_t waves = resident_waves(k_intersect_fast2, stack2_bytes(sc));
            uint32_t rpw = rays_per_wave_for(n, waves);
            hipLaunchKernelGGL(k_intersect_fast2, dim3(blocks_for_waves(n, rpw, waves)), dim3(TBLOCK), stack2_bytes(sc),
                               stream, sc, io, n, rpw, ticket);
        } else {
            allow_large_lds(k_intersect_fast);
            uint64_t waves = resident_waves(k_intersect_fast, stack_bytes(sc));
            uint32_t rpw = rays_per_wave_for(n, waves);
            hipLaunchKernelGGL(k_intersect_fast, dim3(blocks_for_waves(n, rpw, waves)), dim3(TBLOCK), stack_bytes(sc),
                               stream, sc, io, n, rpw, ticket);
        }
        // marked rays: order-dependent ones (spheres only) and the caller's rays whose direction is no unit vector
        hipLaunchKernelGGL(k_rewalk_batch, dim3(rewalk_blocks(n)), dim3(BLOCK), 0, stream, sc, io, n);
    }
}

void launch_intersect_brute(const Scene &sc, const double *rays, uint64_t n, uint32_t *prim_id, double *distance,
                            hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_intersect_brute, dim3(blocks_for(n)), dim3(BLOCK), 0, stream, sc, rays, n, prim_id, distance);
}

void launch_intersect_prims(const Scene &sc, const double *rays, const uint32_t *slots, uint64_t n, uint8_t *hit,
                            double *pos, double *normal, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_intersect_prims, dim3(blocks_for(n)), dim3(BLOCK), 0, stream, sc, rays, slots, n, hit, pos,
                       normal);
}

void launch_raygen(const rt_camera &cam, uint32_t width, uint32_t height, uint64_t seed, const uint32_t *owned,
                   uint32_t n_pixels, double *bundle, hipStream_t stream) {
    if (n_pixels == 0) return;
    CamConst c;
    for (int k = 0; k < 3; k++) c.loc[k] = cam.loc[k];
    for (int k = 0; k < 9; k++) c.dir[k] = cam.dir[k];
    c.focal_length = cam.focal_length;
    c.sensor_to_lens = cam.sensor_to_lens;
    c.aperture_radius = cam.aperture_radius;
    // camera.rs:22-27, evaluated once on the host in the same order
    c.sensor_width = cam.sensor_diagonal / __builtin_sqrt(2.0);
    c.sensor_height = c.sensor_width * ((double)height / (double)width);
    const double *A = cam.dir;
    double v[3] = {0.0, 0.0, cam.sensor_to_lens};
    for (int r = 0; r < 3; r++) {
        double m = A[3 * r] * v[0] + A[3 * r + 1] * v[1] + A[3 * r + 2] * v[2];
        c.lens_center[r] = cam.loc[r] + m;
    }
    hipLaunchKernelGGL(k_raygen, dim3(blocks_for((uint64_t)n_pixels * 5)), dim3(BLOCK), 0, stream, c, width, height,
                       seed, owned, n_pixels, bundle);
}

void launch_tone_map(const double *radiance, uint64_t n, uint8_t *rgba, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_tone_map, dim3(blocks_for(n)), dim3(BLOCK), 0, stream, radiance, n, rgba);
}

void launch_paths_from_rays(const Paths &p, const double *rays, uint64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_paths_from_rays, dim3(blocks_for(n ? n : 1)), dim3(BLOCK), 0, stream, p, rays, n);
}

void launch_paths_from_film(const Paths &p, const FilmView &f, uint32_t n_samples, hipStream_t stream) {
    uint64_t n = (uint64_t)f.n_owned * 5 * n_samples;
    hipLaunchKernelGGL(k_paths_from_film, dim3(blocks_for(n ? n : 1)), dim3(BLOCK), 0, stream, p, f, n_samples);
}

void launch_traverse_paths(const Scene &sc, int mode, const Paths &p, uint32_t depth, uint64_t expected,
                           unsigned long long *ticket, hipStream_t stream, uint32_t walk_blocks_per_cu,
                           bool callers_rays) {
    if (expected == 0) expected = 1;
    if (mode == RT_TRAVERSE_REFERENCE_ORDER) {
        hipLaunchKernelGGL(k_traverse_paths_ref, dim3(blocks_for(expected)), dim3(BLOCK), 0, stream, sc, p, depth);
    } else {
        launch_path_walk_fast(sc, p, depth, expected, ticket, stream, walk_blocks_per_cu);
        // marked rays: order-dependent ones (spheres only), and directions that are no unit vectors: the rays the
        // library makes itself (camera, bounces) are normalised, a caller's first segments need not be
        if (sc.has_spheres || callers_rays)
            hipLaunchKernelGGL(k_rewalk_paths, dim3(rewalk_blocks(expected)), dim3(BLOCK), 0, stream, sc, p, depth);
    }
}

void launch_shade(const Scene &sc, const Paths &p, uint32_t depth, uint64_t expected, uint64_t seed,
                  const uint32_t *keys, const FilmView *film, uint32_t first_sample, hipStream_t stream) {
    if (expected == 0) expected = 1;
    FilmView f = {};
    if (film) f = *film;
    static const int per_thread = [] { // experiment hook: paths per thread of the diffuse kernel
        const char *e = getenv("RT_TUNE_SHADE_PT");
        return e ? atoi(e) : SHADE_PT;
    }();
    if (sc.has_gloss) {
        unsigned blocks = (unsigned)((expected + SHADE_BLOCK - 1) / SHADE_BLOCK);
        hipLaunchKernelGGL((k_shade<true, 1>), dim3(blocks), dim3(SHADE_BLOCK), 0, stream, sc, p, depth, seed, keys, f,
                           first_sample, film ? 1 : 0);
    } else if (per_thread == 2) {
        unsigned blocks = (unsigned)((expected + 2 * SHADE_BLOCK - 1) / (2 * SHADE_BLOCK));
        hipLaunchKernelGGL((k_shade<false, 2>), dim3(blocks), dim3(SHADE_BLOCK), 0, stream, sc, p, depth, seed, keys, f,
                           first_sample, film ? 1 : 0);
    } else {
        unsigned blocks = (unsigned)((expected + SHADE_BLOCK - 1) / SHADE_BLOCK);
        hipLaunchKernelGGL((k_shade<false, 1>), dim3(blocks), dim3(SHADE_BLOCK), 0, stream, sc, p, depth, seed, keys, f,
                           first_sample, film ? 1 : 0);
    }
}

void launch_fold_paths(const Scene &sc, const Paths &p, uint64_t n, double *radiance, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_fold_paths, dim3(blocks_for(n)), dim3(BLOCK), 0, stream, sc, p, n, radiance);
}

void launch_accumulate_film(const Scene &sc, const Paths &p, const FilmView &f, uint32_t n_samples,
                            hipStream_t stream) {
    if (f.n_owned == 0) return;
    const uint32_t waves = (f.n_owned + FILM_WAVE_PIXELS - 1) / FILM_WAVE_PIXELS;
    hipLaunchKernelGGL(k_accumulate_film, dim3((waves + BLOCK / 64 - 1) / (BLOCK / 64)), dim3(BLOCK), 0, stream, sc, p, f,
                       n_samples);
}

void launch_frame_axpy(double *acc, const double *src, double w, uint64_t n, bool first, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_frame_axpy, dim3(blocks_for(n)), dim3(BLOCK), 0, stream, acc, src, w, n, first ? 1 : 0);
}

void launch_bytes_add(uint8_t *acc, const uint8_t *src, uint64_t n_pixels, bool first, hipStream_t stream) {
    if (n_pixels == 0) return;
    hipLaunchKernelGGL(k_bytes_add, dim3(blocks_for(n_pixels)), dim3(BLOCK), 0, stream, reinterpret_cast<uint32_t *>(acc),
                       reinterpret_cast<const uint32_t *>(src), n_pixels, first ? 1 : 0);
}

void launch_resolve(const FilmView &f, uint32_t spp, double *d_radiance, uint8_t *d_rgba, hipStream_t stream) {
    if (f.n_owned == 0) return;
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for(f.n_owned)), dim3(BLOCK), 0, stream, f, spp, d_radiance, d_rgba);
}

} // namespace rtd
